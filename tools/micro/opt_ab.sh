# A/B of one bench.py argument on one box: bash tools/micro/opt_ab.sh "--option g_batch_growth_grads=0"   (or "--no-guide-prefetch", "--one-stream", ...)
cd $GRAFT_REPO_ROOT
export AFI_BENCH_OTHER_DTYPES=0
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"],2), round(d["roofline"]["avg_launch_us"],1))'
for i in 1 2 3; do
python bench.py --steps 8 --warmup 3 --no-interp --no-cpu-baseline 2>/dev/null | python -c "$P" default
python bench.py --steps 8 --warmup 3 --no-interp --no-cpu-baseline $1 2>/dev/null | python -c "$P" "$1"
done
