#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): parity tests + the C3 team / determinism test on the final binary.
export TMPDIR=/tmp
timeout -s KILL 100 python -m pytest tests/test_gpu_parity.py "tests/test_gpu_fullsize.py::test_c3_deterministic" -x -q 2>&1 | tail -2
