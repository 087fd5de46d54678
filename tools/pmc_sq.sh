#!/bin/bash
# SQ counters of the default bench command (VALU instruction counts vs kernel durations): gpurun_out/pmc_sq_<tag>.md
set -u
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmcsq_$tag; mkdir -p $out
for c in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr TA_TA_BUSY_sum" \
         "SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32" \
         "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_VMEM_TA_ADDR_FIFO_FULL SQ_ACTIVE_INST_SCA" \
         "TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 240 rocprofv3 --pmc $c -d $out/$n -o pmc --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-breakdown --no-extra > $out/$n.json 2> $out/$n.err
done
python - "$out" <<'PY'
import sys, glob, csv, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(set(glob.glob(out + "/**/*counter_collection.csv", recursive=True))):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = re.sub(r'.*psg::(k_\w+(<[^>]*>)?).*', r'\1', r["Kernel_Name"]); c = r["Counter_Name"]
        agg[k][c] += float(r["Counter_Value"]); seen[(k, c)] += 1
    for (k, c), n in seen.items(): agg[k]["n_" + c] = n
cols = sorted({c for k in agg for c in agg[k] if not c.startswith("n_")})
print("| kernel | launches | " + " | ".join(cols) + " |"); print("|---|---|" + "---|" * len(cols))
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_VALU", 0)):
    n = max(agg[k].get("n_" + c, 1) for c in cols)
    print(f"| {k} | {int(n)} | " + " | ".join(f"{agg[k].get(c, 0) / max(agg[k].get('n_' + c, 1), 1):.4g}" for c in cols) + " |")
PY
rm -rf $out/*/
