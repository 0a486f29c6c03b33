#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4s20
mkdir -p $O
for w in 8 1 2 3 4 6 16; do
  echo "== panel $w"
  RENET_GEMM_TILE_ORDER=$w timeout 600 python tools/gemm_split_probe.py one base 2048,23033,600,0,1 2048,600,23033,0,0,6 23033,600,2048,1,0 16000,600,800,0,1 16000,600,600,0,0 2>&1 | grep -v amdgpu.ids
done > $O/panel.txt 2>&1
cat $O/panel.txt
