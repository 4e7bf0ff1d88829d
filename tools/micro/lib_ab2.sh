set -e
R=$GRAFT_REPO_ROOT
for round in 1 2 3; do
  for v in A B; do
    ms=$(AFI_LIB_PATH=$R/tools/micro/lib_$v.so AFI_BENCH_OTHER_DTYPES=0 python3 $R/bench.py --no-interp --no-cpu-baseline --steps 8 --warmup 3 2>/dev/null | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],2))")
    echo "round $round lib_$v $ms ms/step"
  done
done
