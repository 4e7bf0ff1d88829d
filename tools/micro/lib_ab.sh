# A/B of two builds of the library on ONE box (box-to-box differences of ~5 % hide the gains of a kernel change): runs the command three
# times per build, alternating, with AFI_LIB_PATH pointing at each build -- the library in the tree is never overwritten.
# Usage: bash tools/micro/lib_ab.sh <lib_A.so> <lib_B.so> <command ...>   (paths relative to the repo root; the command should print one number)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
A=$1; B=$2; shift 2
for round in 1 2 3; do
  for v in "$A" "$B"; do
    echo "round $round [$v] $(AFI_LIB_PATH=$R/$v "$@" 2>/dev/null | tail -1)"
  done
done
