# two-stream kernel trace of the stage-1 step (default schedule): for the per-stream timeline analysis of tools/stream_timeline.py
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_q2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AFI_BENCH_OTHER_DTYPES=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o step -- python3 $R/bench.py --steps 3 --warmup 1 --no-interp --no-cpu-baseline --profile-timed > $O/bench_trace.log 2>&1
echo done
