#!/bin/bash
# round 6, call 3: (a) GPU tests of what is new: the sharded multi-context photon pass, the drop-in on it, the tolerance build at 1e-4,
# the photon-mapped pipeline with the kNN launch's batched pops; (b) exact against tolerance library on C2 / C3 / C5 probes;
# (c) C5 through the pipeline again (kernel trace), after the batched pops.
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_octree_build.py tests/test_gpu_dropin.py tests/test_gpu_tolerance_build.py tests/test_abi.py -m gpu -x -q -rA 2>&1 | tail -40 > gpurun_out/r06/pytest_call3.log
tail -5 gpurun_out/r06/pytest_call3.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "photon or pm or wavefront" 2>&1 | tail -5 | tee -a gpurun_out/r06/pytest_call3.log
L=gpurun_out/r06/ab_tolerance.log
: > $L
WORKLOAD=c2 STEPS=3 tools/ab_builds.sh exact tol exact tol 2>&1 | sed "s/^/c2 /" | tee -a $L
WORKLOAD=c3 SQRTSPP=8 tools/ab_builds.sh exact tol exact tol 2>&1 | sed "s/^/c3 /" | tee -a $L
WORKLOAD=c5 SQRTSPP=8 EMISSIONS=1e7 tools/ab_builds.sh exact tol exact tol 2>&1 | sed "s/^/c5 /" | tee -a $L
python tools/ab_probe.py c5 --sqrtspp 8 --emissions 1e7 "mega:" "wf:MCRT_KERNEL=wf" "mega:" "wf:MCRT_KERNEL=wf" 2>&1 | grep variant | cut -c1-250 | tee gpurun_out/r06/ab_c5_pipeline.log
R=$PWD; O=$R/gpurun_out/r06/c5_wf2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export MCRT_KERNEL=wf
timeout 900 rocprofv3 --kernel-trace --stats -d $O/wf/kt -- python $R/bench.py --child-frame --workload c5 --sqrtspp 8 --emissions 1e7 > $O/wf.kt.log 2>&1
python $R/tools/summarize_rocprof.py $O/wf > $O/wf.md 2>&1
grep -h child_frame $O/wf.kt.log | tail -1 >> $O/wf.md
rm -rf $O/wf
head -14 $O/wf.md | cut -c1-200
