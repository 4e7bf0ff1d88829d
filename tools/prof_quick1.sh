set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_q1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AFI_BENCH_OTHER_DTYPES=0
S="--no-interp --no-cpu-baseline --profile-timed"
export AFI_PROFILE_DUMP=$O/launches_one_stream.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -o step -- python3 $R/bench.py --steps 3 --warmup 0 $S --one-stream > $O/bench_trace_serial.log 2>&1
echo done
