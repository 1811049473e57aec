#!/bin/bash
# Round-2 final evidence pass on the GPU box: GPU tests, the default bench line, rocprofv3 kernel-trace/--stats of the same
# command, full-size C4 / C5 frames. Outputs under gpurun_out/r02_final/ (summaries only).
R=$PWD
O=$R/gpurun_out/r02_final
mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.json; tail -3 $O/bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt_default -- python $R/bench.py --no-cpu --no-counters > $O/kt_default.json 2> $O/kt_default.err
cd $R
python tools/summarize_rocprof.py $O > $O/rocprof_summary.md 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete; find $O -type d -empty -delete
head -40 $O/rocprof_summary.md
( python tools/ab_probe.py c4 --steps 1 "c4_full:"; python bench.py --workload c5 --no-secondary --no-counters --no-cpu --steps 1 --emissions 1e7 | tail -1 ) > $O/full_size_frames.log 2>&1
grep "variant\|photon_pass" $O/full_size_frames.log | cut -c1-400
