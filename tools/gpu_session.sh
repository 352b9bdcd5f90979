#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session L -- views registered without pixels (MI_DMRECON_ENOIMAGE) in the library
# and the shim's lazy-failure semantics against the reference binary; then the whole GPU suite.
export TMPDIR=/tmp
O=gpurun_out/r5l
mkdir -p $O
timeout -s KILL 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin_app.py -m gpu -x -q -k "without_pixels or cannot_be_loaded" > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
timeout -s KILL 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -4 $O/pytest.log
