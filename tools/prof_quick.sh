# one-stream kernel trace of the stage-1 step (3 steps, no warm-up): per-kernel times comparable with profiles/rNN/kernel_stats_*_one_stream.csv
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-prof_quick}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AFI_BENCH_OTHER_DTYPES=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -o step -- python3 $R/bench.py --steps 3 --warmup 0 --no-interp --no-cpu-baseline --one-stream > $O/bench_trace_serial.log 2>&1
cp $(find $O/trace_serial -name "*kernel_stats.csv" | head -1) $O/kernel_stats_one_stream.csv
grep '^{' $O/bench_trace_serial.log > $O/bench_line.json || true
rm -rf $O/trace_serial
head -30 $O/kernel_stats_one_stream.csv | cut -c1-160
