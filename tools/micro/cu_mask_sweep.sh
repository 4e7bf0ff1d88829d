set -e
cd $GRAFT_REPO_ROOT
for cfg in "all all" "0-128 128-256" "even odd" "m4_012 m4_3" "all m4_3" "all 0-64" "0-192 192-256" "all all"; do
  timeout -k 10 150 python tools/micro/cu_mask_step.py $cfg 2>&1 | grep "A=\|rror\|Assert" || echo "failed: $cfg"
done
