#!/bin/bash
# GPU call 1 of round 2: flat-kernel A/B (FP32 cull, split estimate, workgroup sizes), pm with/without the cull, GPU tests.
O=gpurun_out/r02_ab1
mkdir -p $O
export TMPDIR=/tmp
( time timeout 600 python tools/ab_probe.py c2 --steps 2 \
    "old_like:MCRT_FLAT_CULL=0" \
    "cull_1024:" \
    "cull_768:MCRT_FLAT_BLOCK=768" \
    "cull_512:MCRT_FLAT_BLOCK=512" \
    "split_1024:MCRT_FLAT_SPLIT=1" \
    "split_768:MCRT_FLAT_SPLIT=1,MCRT_FLAT_BLOCK=768" \
    "split_512:MCRT_FLAT_SPLIT=1,MCRT_FLAT_BLOCK=512" ) > $O/ab_c2.log 2>&1
cat $O/ab_c2.log | tail -12
( time timeout 600 python tools/ab_probe.py pm --steps 2 "nocull:MCRT_FLAT_CULL=0" "cull:" ) > $O/ab_pm.log 2>&1
tail -5 $O/ab_pm.log
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -8 $O/pytest_gpu.log
