#!/bin/bash
# GPU call 2 of round 2: GPU tests, pm with/without the cull, the C2 bench line with its in-run PMC passes.
O=gpurun_out/r02_ab2
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -8 $O/pytest_gpu.log
( time timeout 600 python tools/ab_probe.py pm --steps 2 "nocull:MCRT_FLAT_CULL=0" "cull:" ) > $O/ab_pm.log 2>&1
tail -5 $O/ab_pm.log
( time timeout 900 python bench.py --workload c2 --no-secondary ) > $O/bench_c2.json 2> $O/bench_c2.err
tail -c 3000 $O/bench_c2.json; tail -5 $O/bench_c2.err
