#!/bin/bash
# Round-end evidence pass on the GPU box: GPU tests, the default bench line (with its per-leg counter summaries), a rocprofv3
# kernel trace of the same command. Outputs (summaries only) under gpurun_out/<tag>/; copy what is to be judged into profiles/.
#   tools/final_evidence.sh r03_final
TAG=${1:-final}
R=$PWD
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q -rA ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( time timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err  # as the driver runs it
tail -c 400 $O/bench_default.json; tail -4 $O/bench_default.err
cp -r $R/gpurun_out/bench_profiles $O/ 2>/dev/null
cd /tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/kt_default -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-counters --no-tolerance > $O/kt_default.json 2> $O/kt_default.err
cd $R
python tools/summarize_rocprof.py $O/kt_default > $O/rocprof_kernel_trace.md 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete; find $O -type d -empty -delete
head -30 $O/rocprof_kernel_trace.md
# one-GPU strong-scaling rehearsal of the configurations BASELINE names (tools/shard_probe.py), and the trace kernel's per-iteration statistics at full size
# (C4 at full size takes minutes per shard: profiles/r05_shard_probe_c4.json stands)
for w in "c2 --sqrtspp 16" "c3" "c5"; do set -- $w; timeout 900 python tools/shard_probe.py $@ --reps 2 > $O/shard_$1.json 2> $O/shard_$1.err; python - <<PY
import json
try:
    r = json.load(open("$O/shard_$1.json"))
    print("$1", {n: (v["slowest_shard_ms"], v["predicted_speedup"], v.get("predicted_speedup_with_photon_pass")) for n, v in r["N"].items()})
except Exception as e:
    print("$1 shard probe failed", e)
PY
done
MCRT_COUNT_TESTS=1 timeout 300 python tools/ab_probe.py c3 --steps 1 "count:" > $O/trace_stats_c3_full.log 2>&1; grep "mcrt trace" $O/trace_stats_c3_full.log | head -1
