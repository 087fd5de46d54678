#!/bin/bash
# Collect PMC counters for the bench kernels in separate passes (gpurun refuses --pmc together with tracing).
# usage: tools/pmc_run.sh <outdir> [bench args]
set -u
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
run() { name=$1; shift; rocprofv3 --pmc "$@" -d "$out/$name" -o pmc --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-breakdown ${BENCH_ARGS:-} > "$out/$name.json" 2> "$out/$name.err"; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
ls -R "$out" | head -40
