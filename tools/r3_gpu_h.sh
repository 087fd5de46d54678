#!/bin/bash
set -x
O=gpurun_out/r3h; mkdir -p $O
export PSGSDF_COMMIT=ed3fccb
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_$i.log 2>&1; echo "pytest $i rc=$?"; tail -2 $O/pytest_$i.log; done
bash tools/profile_round.sh r03a > $O/profile.log 2>&1; tail -30 $O/profile.log
