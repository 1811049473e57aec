#!/bin/bash
# round 5, call 1: (a) the compact bench line on the GPU (headline only, with the counter passes incl. the typed FP64 counters),
# (b) MCRT_FLAT_SHARE=1 against the default on C2 and C2-GGX (the decision the last review asked for), (c) the GPU tests that the
# round's first changes touch (LOCKSTEP fence, shared flat form, RCCL rehearsal)
mkdir -p gpurun_out/r05
date
timeout 600 python bench.py --no-secondary --steps 3 --warmup 1 > gpurun_out/r05/bench_headline.out 2> gpurun_out/r05/bench_headline.err
tail -c 6000 gpurun_out/r05/bench_headline.out; echo; wc -c gpurun_out/r05/bench_headline.out
cp gpurun_out/bench_profiles/bench_full.json gpurun_out/r05/bench_headline_full.json 2>/dev/null
cp gpurun_out/bench_profiles/pmc_c2.md gpurun_out/r05/pmc_c2_first.md 2>/dev/null
for wl in c2 c2_ggx; do
  timeout 300 python tools/ab_probe.py $wl --steps 3 "base:" "share:MCRT_FLAT_SHARE=1" "base:" "share:MCRT_FLAT_SHARE=1" 2>&1 | grep '^{' | cut -c1-200 | tee -a gpurun_out/r05/ab_c2_flat_share.log
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multiprocess.py tests/test_knn_large_k.py tests/test_photon_emission.py -m gpu -q -x 2>&1 | tail -5 | tee gpurun_out/r05/pytest_call1.log
date
