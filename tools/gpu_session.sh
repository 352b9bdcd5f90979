#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): round 5, session S -- how often, and where, a team run differs (tools/team_flake.py)
export TMPDIR=/tmp
timeout -s KILL 200 python tools/team_flake.py 60 0:7 1 2>&1 | tail -12
timeout -s KILL 100 python tools/team_flake.py 40 none 1 2>&1 | tail -6
timeout -s KILL 100 python tools/team_flake.py 40 none none 2>&1 | tail -6
