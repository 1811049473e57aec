#!/bin/bash
# One GPU-box call: the GPU test suite, then the default bench line (all legs). Outputs under gpurun_out/<tag>/.
TAG=${1:-r02a}
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -x -q -s ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 1500 $O/bench_default.json; tail -5 $O/bench_default.err
