#!/bin/bash
O=gpurun_out/r04g; mkdir -p $O
timeout 600 python tools/ab_probe.py c3 --sqrtspp 8 --steps 3 "base:" "items32:MCRT_WF_LEAF=64,MCRT_WF_LEAF_ITEMS=32" "items40:MCRT_WF_LEAF=64,MCRT_WF_LEAF_ITEMS=40" "items48:MCRT_WF_LEAF=64,MCRT_WF_LEAF_ITEMS=48" "items56:MCRT_WF_LEAF=64,MCRT_WF_LEAF_ITEMS=56" "items64:MCRT_WF_LEAF=64,MCRT_WF_LEAF_ITEMS=64" "lanes8:MCRT_WF_LEAF=8" "lanes16:MCRT_WF_LEAF=16" "lanes20:MCRT_WF_LEAF=20" "refill8:MCRT_WF_REFILL=8" "refill24:MCRT_WF_REFILL=24" "base2:" 2>&1 | cut -c1-130 | tee $O/ab_gate.log
MCRT_COUNT_TESTS=1 timeout 300 python tools/ab_probe.py c3 --steps 1 "count:" "count_items48:MCRT_WF_LEAF=64,MCRT_WF_LEAF_ITEMS=48" > $O/trace_stats.log 2>&1; grep "mcrt trace" $O/trace_stats.log | sort -u
