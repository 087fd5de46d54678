#!/bin/bash
# round 3, GPU call B: GPU suite, big soak at 64^3 x 8, soak without the check words (round-2 behaviour), soak at the headline size
set -x
O=gpurun_out/r3b; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 $O/pytest.log
timeout 1500 python tools/soak.py --models LED,SH1,SH2 --n 64 --frames 8 --procs 8 --reps 25 --variants default,fold0,poll0,persist0 --out $O/soak > $O/soak.log 2>&1; echo "soak rc=$?"
timeout 1500 python tools/soak.py --models LED,SH1 --n 64 --frames 8 --procs 8 --reps 100 --variants nocheck --out $O/soak_nocheck > $O/soak_nocheck.log 2>&1; echo "soak_nocheck rc=$?"
timeout 1500 python tools/soak.py --models SH1,LED --n 256 --frames 50 --width 640 --height 480 --procs 2 --reps 2 --variants default,fold0,persist0 --timeout 1200 --out $O/soak_big > $O/soak_big.log 2>&1; echo "soak_big rc=$?"
