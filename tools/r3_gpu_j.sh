#!/bin/bash
set -x
O=gpurun_out/r3j; mkdir -p $O
for i in 1 2 3 4 5; do timeout 600 python -m pytest tests/test_slab_gpu.py tests/test_bench_gpu.py -m gpu -x -q > $O/pytest_slab_$i.log 2>&1; echo "slab $i rc=$?"; tail -1 $O/pytest_slab_$i.log; done
# two ranks sharing the GPU: collectives per step with and without the cross-rank persistent solve (functional numbers, not a measurement)
for xr in 1 0; do
  port=$((29600 + xr))
  for r in 0 1; do
    PSGSDF_XR=$xr PSGSDF_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 GLOO_SOCKET_IFNAME=lo RANK=$r LOCAL_RANK=$r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port \
      timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --grid 128 --frames 20 --no-breakdown > $O/share_xr${xr}_r$r.json 2> $O/share_xr${xr}_r$r.err &
  done
  wait
  python - <<EOF
import json
d=json.load(open("$O/share_xr${xr}_r0.json"))
print("xr=$xr", d["value"], d["ms_per_step"], d["config"].get("collectives_per_step"), d["config"].get("band_rows_per_rank"))
EOF
done
