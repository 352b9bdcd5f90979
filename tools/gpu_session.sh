#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the whole GPU suite.
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -6
