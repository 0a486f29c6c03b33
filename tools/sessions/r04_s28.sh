#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s28
mkdir -p $O
export TMPDIR=/tmp
for F in 0 1; do
(cd /tmp && RENET_FORCE_REDUCER=$F timeout 600 rocprofv3 --hip-trace -d $R/$O/ht$F -o ht -- python $R/tools/dp_overhead.py 40 > $R/$O/ht$F.log 2>&1)
python - <<PY
import sqlite3, glob
db=glob.glob('$O/ht$F/*results.db')[0]
c=sqlite3.connect(db).cursor()
tabs=[r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print('forced=$F tables:', [t for t in tabs if 'region' in t.lower() or 'api' in t.lower()][:12])
for t in tabs:
    if t.lower() in ('regions','regions_and_samples'):
        cols=[r[1] for r in c.execute('pragma table_info(%s)' % t)]
        print(t, cols)
try:
    rows=c.execute("select name, count(*), sum(end-start)/1e6 from regions group by name order by 3 desc limit 14").fetchall()
    for r in rows: print('   %-40s n=%6d total %.1f ms' % (str(r[0])[:40], r[1], r[2]))
except Exception as e:
    print('query failed', e)
PY
done
find $O -name "*.db" -delete
