#!/bin/bash
# round 5, call 26: the evidence pass on the final library (renderKernelFlatK at 768 lanes by default): GPU tier
# (-rA), the default bench line as the driver runs it, a rocprofv3 kernel trace of the same command without the CPU / counter legs
TAG=r05_evidence5
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
date
( time timeout 1200 python -m pytest tests -m gpu -q -rA ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err
wc -c $O/bench_default.json; head -c 600 $O/bench_default.json; echo; tail -4 $O/bench_default.err | cut -c1-300
cp -r $R/gpurun_out/bench_profiles $O/ 2>/dev/null
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats -d $O/kt_default -- python $R/bench.py --no-cpu --no-counters > $O/kt_default.json 2> $O/kt_default.err
cd $R
python tools/summarize_rocprof.py $O/kt_default > $O/rocprof_kernel_trace.md 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete; find $O -type d -empty -delete
head -14 $O/rocprof_kernel_trace.md
date
