#!/bin/bash
# One profiling pass for profiles/: kernel trace + stats of the default bench command, then FETCH_SIZE / WRITE_SIZE PMC passes.
# usage (on the GPU box): tools/profile_round.sh <tag>     -> gpurun_out/prof_<tag>/
set -u
export PSGSDF_BENCH_LIVE_PMC=0      # (the passes below ARE the counter passes: bench.py must not spawn its own under a profiler)
tag=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag; mkdir -p $out
PSGSDF_BENCH_NO_SOLVE_MODEL=1 rocprofv3 --kernel-trace --stats -d $out/trace -o kt -- python bench.py --no-cpu-baseline --no-extra > $out/bench_traced.json 2> $out/trace.err
python tools/rocpd_stats.py $(find $out/trace -name '*results.db' | head -1) $out/kernel_stats.md $out/kernel_stats_summary.json > /dev/null 2>> $out/trace.err
python tools/gap_stats.py $(find $out/trace -name '*results.db' | head -1) > $out/gaps.txt 2>> $out/trace.err
for c in FETCH_SIZE WRITE_SIZE; do
  PSGSDF_BENCH_NO_SOLVE_MODEL=1 rocprofv3 --pmc $c -d $out/pmc_$c -o pmc --output-format csv -- python bench.py --steps 4 --warmup 1 --reps 1 --no-cpu-baseline --no-breakdown --no-extra > $out/pmc_$c.json 2> $out/pmc_$c.err
done
f=$(find $out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1); w=$(find $out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py "$f" "$w" > $out/pmc_summary.json
cp $out/pmc_summary.json profiles/pmc_summary.json; cp $out/kernel_stats_summary.json profiles/kernel_stats_summary.json      # (the line below quotes them; copy them into profiles/ at home too)
env -u PSGSDF_BENCH_LIVE_PMC python bench.py > $out/bench.json 2> $out/bench.err
find $out/trace -name "*.db" -delete; rm -rf $out/pmc_*/    # keep the summaries only (the databases are tens of MB)
head -20 $out/kernel_stats.md; cat $out/pmc_summary.json | head -30; cut -c 1-300 $out/bench.json
