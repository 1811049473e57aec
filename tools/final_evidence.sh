#!/bin/bash
# Round-end evidence pass on the GPU box: GPU tests, the default bench line (with its per-leg counter summaries), a rocprofv3
# kernel trace of the same command. Outputs (summaries only) under gpurun_out/<tag>/; copy what is to be judged into profiles/.
#   tools/final_evidence.sh r03_final
TAG=${1:-final}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( time timeout 1500 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json; tail -4 $O/bench_default.err
cp -r $R/gpurun_out/bench_profiles $O/ 2>/dev/null
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats -d $O/kt_default -- python $R/bench.py --no-cpu --no-counters > $O/kt_default.json 2> $O/kt_default.err
cd $R
python tools/summarize_rocprof.py $O/kt_default > $O/rocprof_kernel_trace.md 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete; find $O -type d -empty -delete
head -30 $O/rocprof_kernel_trace.md
