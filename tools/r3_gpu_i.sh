#!/bin/bash
set -x
O=gpurun_out/r3i; mkdir -p $O
timeout 1200 python -m pytest tests/test_slab_gpu.py -m gpu -x -q > $O/pytest_slab.log 2>&1; echo "pytest_slab rc=$?"; tail -40 $O/pytest_slab.log | cut -c1-400
