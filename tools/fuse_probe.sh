#!/bin/bash
# scratch: where does the fuse stage's time go on the 21-frame demo (prefetch window, host-writers path)
cd "$(dirname "$0")/.."
OUT=/tmp/fp; rm -rf $OUT; mkdir -p $OUT
cat > $OUT/config.json <<J
{"input": "$PWD/tests/golden/sokrates_21/", "output": "$OUT/", "pose filename": "pose.txt", "datatype": "multiview", "first": 0, "last": 20, "voxel size": 0.004,
 "truncation factor": 5, "zmin": 0.5, "zmax": 3.5, "sharpness threshold": 0.0, "model type": "SH1", "loss function": "cauchy",
 "reg albedo": 0.0, "reg norm": 10.0, "reg laplacian": 0.0, "max iter": 2, "damping": 1.0, "converge threshold": 5e-3, "lambda": 0.2,
 "upsample": false, "--light": true, "--albedo": true, "--distance": true, "--pose": true}
J
show() { python3 -c "
import json,sys; d=json.load(open('$OUT/t.json'))['stages_s']
print('$1', {k.split(':')[0]: round(v,4) for k,v in d.items() if k.startswith(('fuse','decode','keyframe'))}, json.load(open('$OUT/t.json'))['total_s'])"; }
for rep in 1 2; do
psgradientsdf_amd/host/voxelPS --config_file $OUT/config.json --timing $OUT/t.json > /dev/null 2>&1; show default
VOXELPS_PREFETCH=1 psgradientsdf_amd/host/voxelPS --config_file $OUT/config.json --timing $OUT/t.json > /dev/null 2>&1; show prefetch1
VOXELPS_PREFETCH=8 psgradientsdf_amd/host/voxelPS --config_file $OUT/config.json --timing $OUT/t.json > /dev/null 2>&1; show prefetch8
psgradientsdf_amd/host/voxelPS --config_file $OUT/config.json --timing $OUT/t.json --host-writers > /dev/null 2>&1; show host-writers
done
