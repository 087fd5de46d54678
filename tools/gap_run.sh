cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/gapx; mkdir -p $out
rocprofv3 --kernel-trace -d $out/trace -o kt -- python bench.py --no-cpu-baseline --no-breakdown > $out/b1.json 2> $out/err1
python tools/gap_stats.py $out/trace/kt_results.db 6
PSGSDF_NO_WATCH=1 rocprofv3 --kernel-trace -d $out/trace2 -o kt -- python bench.py --no-cpu-baseline --no-breakdown > $out/b2.json 2> $out/err2
python tools/gap_stats.py $out/trace2/kt_results.db 6
rm -rf $out/trace $out/trace2
