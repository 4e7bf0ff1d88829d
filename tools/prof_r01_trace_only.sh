set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r01k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o runc -- python3 $R/bench.py --steps 3 --warmup 0 --no-interp --no-cpu-baseline > $O/bench_trace.log 2>&1
cd $R && timeout -k 10 400 python bench.py > $O/bench_default.log 2>&1
echo done
