cd $GRAFT_REPO_ROOT
export AFI_BENCH_OTHER_DTYPES=0
for i in 1 2; do
python bench.py --steps 8 --warmup 3 --no-interp --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('prefetch', d['config']['guide_prefetch'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['losses_last_step']['d_loss_p2'], d['losses_last_step']['g_loss_p2'])"
python bench.py --steps 8 --warmup 3 --no-interp --no-cpu-baseline --no-guide-prefetch 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('prefetch', d['config']['guide_prefetch'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['losses_last_step']['d_loss_p2'], d['losses_last_step']['g_loss_p2'])"
done
