#!/bin/bash
# scratch: voxelPS on 1 process vs --gpus N --transport sockets on one GPU (tests/test_voxelps_ranks_gpu.py is the real check)
set -u
cd "$(dirname "$0")/.."
GOLD=$PWD/tests/golden/sokrates_small
EXE=$PWD/psgradientsdf_amd/host/voxelPS
OUT=${1:-/tmp/vr}; N=${2:-2}
rm -rf $OUT; mkdir -p $OUT/one $OUT/many
for d in one many; do
cat > $OUT/$d/config.json <<J
{"input": "$GOLD/", "output": "$OUT/$d/", "pose filename": "pose.txt", "datatype": "multiview", "first": 0, "last": 7, "voxel size": 0.004,
 "truncation factor": 5, "zmin": 0.5, "zmax": 3.5, "sharpness threshold": 0.0, "model type": "SH1", "loss function": "cauchy",
 "reg albedo": 0.0, "reg norm": 10.0, "reg laplacian": 0.0, "max iter": 7, "damping": 1.0, "converge threshold": 1e-9, "lambda": 0.2,
 "upsample": false, "--light": true, "--albedo": true, "--distance": true, "--pose": true, "grid dim": 128}
J
done
timeout 200 $EXE --config_file $OUT/one/config.json > $OUT/one.log 2>&1; echo one rc=$?
NCU=256; M=""; for ((r=0;r<N;r++)); do M="$M${M:+,}$((r*NCU/N)):$(((r+1)*NCU/N))"; done
VOXELPS_SHARE_GPU=1 VOXELPS_CU_MASKS=$M timeout 300 $EXE --config_file $OUT/many/config.json --gpus $N --transport sockets > $OUT/many.log 2>&1; echo many rc=$?
tail -5 $OUT/many.log
ls -la $OUT/one $OUT/many | head -60
for f in $(ls $OUT/one); do cmp -s $OUT/one/$f $OUT/many/$f && echo "SAME $f" || echo "DIFF $f $(wc -l < $OUT/one/$f) $(wc -l < $OUT/many/$f)"; done
