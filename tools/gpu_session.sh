#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the parity module (new: scratch pool test).
export TMPDIR=/tmp
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -15
