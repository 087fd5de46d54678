#!/bin/bash
# usage: flake.sh <lib or ""> <n>
lib=$1; n=$2; ok=0; bad=0
for i in $(seq 1 $n); do
  if [ -n "$lib" ]; then export PSGSDF_ENGINE_LIB=$PWD/$lib; else unset PSGSDF_ENGINE_LIB; fi
  PSGSDF_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 GLOO_SOCKET_IFNAME=lo timeout 300 python bench.py --gpus 8 --steps 3 --warmup 1 --reps 2 --grid 64 --frames 8 --configs4 48:70 > /tmp/flake_out.json 2> /tmp/flake_err.log
  if [ $? -eq 0 ]; then ok=$((ok+1)); else bad=$((bad+1)); grep -h "gave up\|fallback\|status" /tmp/flake_err.log | head -3; cp /tmp/flake_err.log gpurun_out/flake_err_$i.log; fi
done
echo "lib=${lib:-current} ok=$ok bad=$bad"
