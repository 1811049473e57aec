#!/bin/bash
# where a small frame's time goes in the pipeline: kernel trace of spaceship 1080p @ 1 spp (2 M path samples) forced through it
O=$PWD/gpurun_out/r04w; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
MCRT_KERNEL=wf timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -- python $GRAFT_REPO_ROOT/tools/ab_probe.py spaceship --sqrtspp 1 --steps 5 "wf:MCRT_KERNEL=wf" > $O/probe.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/summarize_rocprof.py $O/kt > $O/kernel_trace_small_frame.md 2>&1
grep '^{' $O/probe.log | cut -c1-150
head -14 $O/kernel_trace_small_frame.md
python - <<PY
import sqlite3, glob
db = glob.glob("$O/kt/**/*.db", recursive=True)
if db:
    c = sqlite3.connect(db[0])
    names = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = [n for n in names if 'kernel_dispatch' in n]
    print(t[:3])
PY
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete
