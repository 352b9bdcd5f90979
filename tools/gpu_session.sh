#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the drop-in tests and the parity tests on the rebuilt binaries.
export TMPDIR=/tmp
timeout -s KILL 200 python -m pytest tests/test_gpu_dropin_app.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
