#!/bin/bash
# round 4, GPU call 1: the GPU test tier on the new default trace kernel (shared leaf steps), then A/B of the forms
O=gpurun_out/r04a; mkdir -p $O
python __graft_entry__.py smoke > $O/smoke.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 600 python tools/ab_probe.py c3 --sqrtspp 8 --steps 3 "share0:MCRT_WF_SHARE=0" "share12:" "share8:MCRT_WF_LEAF=8" "share16:MCRT_WF_LEAF=16" \
  "share24:MCRT_WF_LEAF=24" "share4:MCRT_WF_LEAF=4" "share12_mi16:MCRT_WF_MININNER=16" "share12_mi4:MCRT_WF_MININNER=4" "share12_rf8:MCRT_WF_REFILL=8" \
  "halves2:MCRT_WF_HALVES=2" "share0b:MCRT_WF_SHARE=0" > $O/ab_c3.log 2>&1; cat $O/ab_c3.log | cut -c1-200
timeout 300 python tools/ab_probe.py c3 --sqrtspp 4 --steps 1 "count_share0:MCRT_COUNT_TESTS=1,MCRT_WF_SHARE=0" "count_share:MCRT_COUNT_TESTS=1" "count_share4:MCRT_COUNT_TESTS=1,MCRT_WF_LEAF=4" > $O/count_c3.log 2>&1; grep "mcrt trace\|variant" $O/count_c3.log | cut -c1-400
timeout 600 python tools/ab_probe.py c4 --sqrtspp 4 --steps 2 "share0:MCRT_WF_SHARE=0" "share12:" "share8:MCRT_WF_LEAF=8" "share0b:MCRT_WF_SHARE=0" > $O/ab_c4.log 2>&1; cat $O/ab_c4.log | cut -c1-200
