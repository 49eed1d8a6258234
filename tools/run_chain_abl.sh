#!/bin/bash
# chain1x1 kernel ablations (same box): full, no DMA, no MFMA, no stores, DMA only
mkdir -p gpurun_out
for d in ${CHAIN_ABL:-0 1 2 4 6 3 14 22}; do
  BSY_FUSE_CHAIN=1 BSY_CHAIN_DBG=$d timeout -k 10 200 python tools/gpu_explore.py time 64 > gpurun_out/chain_abl_$d.txt 2>&1 || exit $?
  echo "== dbg $d"; grep "kind 21" gpurun_out/chain_abl_$d.txt | cut -c1-100
done
