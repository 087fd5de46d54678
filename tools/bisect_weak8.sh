export PSGSDF_BENCH_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 GLOO_SOCKET_IFNAME=lo PSGSDF_FAULT_DUMP=150 PSGSDF_BENCH_NO_FULL_CHECK=1 PSGSDF_XWAIT_LOG2=20
run() { name=$1; shift; t0=$(date +%s); ( "$@" > gpurun_out/bis_$name.out 2> gpurun_out/bis_$name.err ); rc=$?; echo "$name rc=$rc $(( $(date +%s) - t0 ))s nan=$(grep -c 'NaN came back' gpurun_out/bis_$name.err) stale=$(grep -c 'does not show' gpurun_out/bis_$name.err) $(head -c 100 gpurun_out/bis_$name.out)"; grep -h "NaN came back\|does not show" gpurun_out/bis_$name.err | sed 's/.*PsgsdfError: //' | sort | uniq -c | head -12; }
B="timeout 200 python bench.py --weak --grid 64 --steps 2 --warmup 1 --reps 1 --no-extra --no-breakdown"
for i in 1 2 3; do run w8f8_$i $B --gpus 8 --frames 8; done
for i in 1 2 3; do run w8f4_$i $B --gpus 8 --frames 4; done
run w8f40 $B --gpus 8 --frames 40 --width 160 --height 120
run w4f8 $B --gpus 4 --frames 8
