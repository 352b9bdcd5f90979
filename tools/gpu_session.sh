#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): scene2pset tests.
export TMPDIR=/tmp
timeout -s KILL 600 python -m pytest tests/test_scene2pset.py tests/test_pointset.py -x -q -m gpu 2>&1 | tail -12
