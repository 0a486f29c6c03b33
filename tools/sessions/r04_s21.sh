#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=gpurun_out/r4s21
mkdir -p $O
export TMPDIR=/tmp
for w in 8 4 3 2; do
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && RENET_GEMM_TILE_ORDER=$w timeout 300 rocprofv3 --pmc $C --kernel-trace -d $R/$O/p${w}_$C -o pmc -- python $R/tools/gemm_split_probe.py one base 2048,23033,600,0,1 23033,600,2048,1,0 16000,600,800,0,1 > $R/$O/p${w}_$C.log 2>&1)
  done
  F=$(find $O/p${w}_FETCH_SIZE -name "*results.db" | head -1); W=$(find $O/p${w}_WRITE_SIZE -name "*results.db" | head -1)
  echo "== panel $w"
  python tools/pmc_traffic.py "$F" "$W" $O/traffic_p$w.json bf16x6 > /dev/null
  python - <<PY
import json
j=json.load(open('$O/traffic_p$w.json'))
for k,v in j['kernels'].items():
    if 'gemm' in k:
        f=v['FETCH_SIZE_KB'] or 0; w_=v['WRITE_SIZE_KB'] or 0
        print('   %-36s n=%3d fetch(x2) %7.1f MB write %7.1f MB' % (k[:36], v['dispatches'], 2*f/1024, w_/1024))
PY
done
find $O -name "*.db" -delete
