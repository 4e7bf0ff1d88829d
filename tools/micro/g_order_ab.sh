# A/B of the G-phase backward order on one box: smallest level first (default) against level order
cd $GRAFT_REPO_ROOT
export AFI_BENCH_OTHER_DTYPES=0
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"],2), round(d["roofline"]["avg_launch_us"],1))'
for i in 1 2 3; do
python bench.py --steps 8 --warmup 3 --no-interp --no-cpu-baseline 2>/dev/null | python -c "$P" small-first
AFI_BENCH_G_BWD_ORDER=level-order python bench.py --steps 8 --warmup 3 --no-interp --no-cpu-baseline 2>/dev/null | python -c "$P" level-order
done
