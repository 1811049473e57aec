#!/bin/bash
# round 4, GPU call 2: the GPU tier with the tightened parity tests (-rA: every test's printed figures), full-size C3 A/B of pool size and
# the two-stream split, then the default bench line
O=gpurun_out/r04b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -rA > $O/pytest_rA.log 2>&1; tail -3 $O/pytest_rA.log; grep -c PASSED $O/pytest_rA.log; grep "FAILED\|ERROR" $O/pytest_rA.log | head -20
timeout 900 python tools/ab_probe.py c3 --steps 2 "slots8M:" "slots16M:MCRT_WF_SLOTS=16777216" "slots32M:MCRT_WF_SLOTS=33554432" "halves2:MCRT_WF_HALVES=2" "halves2_16M:MCRT_WF_HALVES=2,MCRT_WF_SLOTS=16777216" "share0:MCRT_WF_SHARE=0" > $O/ab_c3_full.log 2>&1; cut -c1-190 $O/ab_c3_full.log
(time python bench.py --steps 5 > $O/bench_default.json 2> $O/bench_default.err) 2>&1 | tail -3; tail -c 600 $O/bench_default.err; python - <<'PY'
import json
try:
    r=json.load(open('gpurun_out/r04b/bench_default.json'))
except Exception as e:
    print('bench json error', e); raise SystemExit
def show(n, x):
    rf=x.get('roofline',{})
    print(n, 'value %.1f ms %.1f' % (x['value'], x['ms_per_step']), 'bound', rf.get('bound'), 'frac', rf.get('frac'), 'hbm', rf.get('hbm_frac_measured'), 'valu', rf.get('valu_issue_frac'), 'nec', rf.get('frac_necessary'), 'alg', rf.get('algorithmic_frac'), 'parity', x.get('parity',{}).get('bit_identical'), x.get('frame_with_photon_pass_ms'))
show('c2', r)
for k,v in r.get('secondary',{}).items():
    if 'error' in v: print(k, v['error'])
    else: show(k, v)
PY
cp gpurun_out/bench_profiles/*.md $O/ 2>/dev/null
