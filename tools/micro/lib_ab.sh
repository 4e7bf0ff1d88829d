# A/B of two builds of the library on ONE box (box-to-box differences of 5 % and more hide small gains): tools/micro/lib_A.so and lib_B.so are
# copied over afigan_amd/csrc/libafigan_hip.so in turn (A B A B) and the command given as arguments runs on each.
#   bash tools/micro/lib_ab.sh python tools/cfg1_loop.py 200 g
set -e
cd "$(dirname "$0")/../.."
cp afigan_amd/csrc/libafigan_hip.so /tmp/lib_keep.so
for v in A B A B; do
  cp tools/micro/lib_$v.so afigan_amd/csrc/libafigan_hip.so
  echo "== $v"
  "$@"
done
cp /tmp/lib_keep.so afigan_amd/csrc/libafigan_hip.so
