#!/bin/bash
# Diagnostic: phase timings of fuse_chain_kernel for the ablation builds build/stamps/abl<N>.so
for n in "$@"; do
  echo "== ablation $n"
  SFM_STAMPS_LIB=build/stamps/abl$n.so timeout -k 10 200 python3 tools/chain_stamps.py 9 > gpurun_out/abl_$n.txt 2>&1 || { tail -5 gpurun_out/abl_$n.txt; exit 1; }
  grep -A7 "keyframe +16" gpurun_out/abl_$n.txt | grep -E "survivors|barrier passed|A done|B done|C done"
  tail -1 gpurun_out/abl_$n.txt
done
