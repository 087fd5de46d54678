#!/bin/bash
# bench throughput against the workgroup cap of the fused PCG pass (PSGSDF_PCG_BLOCKS); usage on the GPU box: tools/pcg_block_sweep.sh 560 620 768
for b in "$@"; do
  PSGSDF_PCG_BLOCKS=$b timeout 200 python bench.py --no-cpu-baseline --no-breakdown --steps 40 2>/dev/null | B=$b python -c "import json,sys,os; d=json.loads(sys.stdin.read()); print(os.environ['B'], round(d['value']), round(1e3*d['roofline']['avg_launch_ms'],2))"
done
