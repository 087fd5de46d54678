#!/bin/bash
set -x
O=gpurun_out/r3e; mkdir -p $O
for p in 1 0; do PSGSDF_PACKED=$p timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_packed$p.json 2> $O/bench_packed$p.err; python - <<EOF
import json
d=json.load(open("$O/bench_packed$p.json"))
print("packed=$p", d["value"], d["iterate_value"], d["kernels"])
EOF
done
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q 2>&1 | tail -5
