#!/bin/bash
# diagnostics for the two-ranks-on-one-GPU hang
mkdir -p gpurun_out/diag
D=gpurun_out/diag
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; hostname; cat /etc/hosts; ip -o addr 2>/dev/null; } > $D/box.txt 2>&1
run() { # name, extra env...
  name=$1; shift
  env PSGSDF_BENCH_SHARE_GPU=1 PSGSDF_FAULT_DUMP=60 "$@" timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
     bench.py --gpus 2 --steps 3 --warmup 1 --grid 64 --frames 8 > $D/$name.out 2> $D/$name.err
  echo "$name rc=$? $(date +%s)" >> $D/summary.txt
}
date +%s > $D/summary.txt
run noif1
run noif2
run noif3
run lo1 GLOO_SOCKET_IFNAME=lo
