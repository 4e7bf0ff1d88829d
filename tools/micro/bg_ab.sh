cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_modules.py tests/test_gpu_fullsize.py tests/test_gpu_stage1.py tests/test_gpu_fpn.py tests/test_gpu_pafpn.py -x -q 2>&1 | tail -4
export AFI_BENCH_OTHER_DTYPES=0
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], round(d["ms_per_step"],2), round(d["roofline"]["avg_launch_us"],1), d["losses_last_step"]["g_loss_p2"])'
for i in 1 2 3; do
python bench.py --steps 8 --warmup 3 --no-interp --no-cpu-baseline 2>/dev/null | python -c "$P" batched
python bench.py --steps 8 --warmup 3 --no-interp --no-cpu-baseline --option g_batch_growth_grads=0 2>/dev/null | python -c "$P" per-conv
done
