#!/bin/bash
export TMPDIR=/tmp
timeout -s KILL 1500 python -m pytest tests/test_gpu_fullsize.py -q -k config5_full_size -s --durations=3 2>&1 | tail -12
