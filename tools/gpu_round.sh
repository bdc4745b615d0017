#!/bin/bash
# One GPU-box call: GPU parity tests, then the default bench line.  A step that was killed at its time limit stops the script
# (no further GPU step after a hang); an ordinary test failure does not.
# usage: tools/gpu_round.sh <tag> [pytest args...]
tag=${1:-run}; shift
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x "$@" > gpurun_out/${tag}_tests.log 2>&1
rc=$?
tail -5 gpurun_out/${tag}_tests.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "tests killed at the time limit: stopping"; exit $rc; fi
timeout -k 10 600 python bench.py --steps 30 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rc2=$?
tail -c 600 gpurun_out/${tag}_bench.json; tail -5 gpurun_out/${tag}_bench.err
exit $(( rc != 0 ? rc : rc2 ))
