#!/bin/bash
# round 5, session 12: what a ONE-rank RCCL group adds to the step (RENET_FORCE_REDUCER=1: 3.39 vs 2.73 ms): kernel + memory-copy trace
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r5s12
mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && RENET_FORCE_REDUCER=1 timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/$O/rccl1 -o t -- python $R/bench.py --steps 20 --warmup 3 --plain > $R/$O/rccl1.log 2>&1)
find $O/rccl1 -name "*stats*.csv" | head; for f in $(find $O/rccl1 -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f | cut -c1-200; grep -i "nccl\|rccl" $f | cut -c1-250; done
for f in $(find $O/rccl1 -name "*memory_copy_stats.csv"); do echo "== $f"; cat $f | cut -c1-250 | head -12; done
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r5s12/rccl1/**/*memory_copy_trace.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), 'copies')
    if rows:
        print(list(rows[0].keys()))
        import collections
        agg = collections.defaultdict(lambda: [0, 0.0, 0])
        for r in rows:
            k = r.get('Direction') or r.get('Name') or '?'
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            b = int(r.get('Bytes') or r.get('Size') or 0) if (r.get('Bytes') or r.get('Size')) else 0
            agg[k][0] += 1; agg[k][1] += d; agg[k][2] += b
        for k, (n, us, b) in agg.items():
            print('%-30s %6d copies  %10.1f us total  %8.2f us avg  %10.1f MB' % (k, n, us, us / n, b / 1e6))
PY
find $O -name "*.db" -delete; find $O -name "*trace.csv" -size +20M -delete
tail -2 $O/rccl1.log | cut -c1-300
