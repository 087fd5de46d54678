#!/bin/bash
# `voxelPS --gpus N` repeated: every run of a configuration must write the same bytes (the ranks place their shares by an exclusive scan; the engine's multi-rank
# loop is bit-reproducible).  usage: tools/soak_voxelps_ranks.sh [runs per configuration, default 6]   -> one JSON line
cd "$(dirname "$0")/.."
RUNS=${1:-6}; GOLD=$PWD/tests/golden/sokrates_small; EXE=$PWD/psgradientsdf_amd/host/voxelPS; OUT=/tmp/svr; NCU=256
rm -rf $OUT; mkdir -p $OUT
res=""
for cfg in "2 SH1 128 false" "4 SH1 64 true" "8 SH1 96 false" "3 LED 96 false"; do
  set -- $cfg; N=$1; MODEL=$2; GRID=$3; UPS=$4
  M=""; for ((r=0;r<N;r++)); do M="$M${M:+,}$((r*NCU/N)):$(((r+1)*NCU/N))"; done
  first=""; same=0; failed=0
  for ((i=0;i<RUNS;i++)); do
    d=$OUT/run; rm -rf $d; mkdir -p $d
    cat > $d/config.json <<J
{"input": "$GOLD/", "output": "$d/", "pose filename": "pose.txt", "datatype": "multiview", "first": 0, "last": 7, "voxel size": 0.004,
 "truncation factor": 5, "zmin": 0.5, "zmax": 3.5, "sharpness threshold": 0.0, "model type": "$MODEL", "loss function": "cauchy",
 "reg albedo": 0.0, "reg norm": 10.0, "reg laplacian": 0.0, "max iter": 7, "damping": $( [ $UPS = true ] && echo 10.0 || echo 1.0 ), "converge threshold": 1e-9, "lambda": 0.2,
 "upsample": $UPS, "--light": true, "--albedo": true, "--distance": true, "--pose": true, "grid dim": $GRID}
J
    VOXELPS_SHARE_GPU=1 VOXELPS_CU_MASKS=$M timeout 200 $EXE --config_file $d/config.json --gpus $N --transport sockets > $d/log.txt 2>&1 || failed=$((failed+1))
    h=$(cd $d && ls | grep -v -e config.json -e log.txt | sort | xargs md5sum | md5sum | cut -c1-16)
    if [ -z "$first" ]; then first=$h; fi
    [ "$h" = "$first" ] && same=$((same+1))
  done
  res="$res${res:+, }{\"ranks\": $N, \"model\": \"$MODEL\", \"grid\": $GRID, \"upsample\": $UPS, \"runs\": $RUNS, \"identical_to_the_first\": $same, \"failed\": $failed, \"digest\": \"$first\"}"
done
echo "{\"tool\": \"soak_voxelps_ranks\", \"configurations\": [$res]}"
