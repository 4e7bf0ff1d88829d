# A/B of bench.py argument strings (context options: --option name=value; --dtype) on ONE box: the stage-1 step (two-stream wall time, 3 alternating rounds) for each argument string.
# Usage: bash tools/micro/knob_ab.sh "<bench args A>" "<bench args B>" ...
set -e
R=${GRAFT_REPO_ROOT:-.}
for round in 1 2 3; do
  for a in "$@"; do
    ms=$(AFI_BENCH_OTHER_DTYPES=0 python3 $R/bench.py --no-interp --no-cpu-baseline --steps 8 --warmup 3 $a 2>/dev/null | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'],2))")
    echo "round $round [$a] $ms ms/step"
  done
done
