#!/bin/bash
# round 3, GPU call A: the micro-test of mapped-memory visibility, the soak, then the GPU suite
set -x
O=gpurun_out/r3a; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/mapped_visibility.hip -o /tmp/mapped_visibility && timeout 300 /tmp/mapped_visibility 20000 > $O/mapped_visibility.jsonl 2> $O/mapped_visibility.err
cat $O/mapped_visibility.jsonl
timeout 1500 python tools/soak.py --models LED,SH1,SH2 --n 64 --frames 8 --procs 4 --reps 8 --out $O/soak > $O/soak.log 2>&1
tail -c 6000 $O/soak.log
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $O/pytest.log
