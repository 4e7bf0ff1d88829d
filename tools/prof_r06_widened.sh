#!/bin/bash
# rocprofv3 kernel-trace summaries of the widened rows: the FPN top-down merge and the BiFPN training pass.
# One gpurun call; outputs under gpurun_out/r6prof/, copied into profiles/r06/ by hand.
set -e
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6prof
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/fpn -o fpn -- python3 $R/tools/fpn_loop.py fpn 10 > $O/fpn_loop.txt 2>$O/fpn.err
rocprofv3 --kernel-trace --stats -d $O/bifpn -o bifpn -- python3 $R/tools/bifpn_train_loop.py 5 > $O/bifpn_train_loop.txt 2>$O/bifpn.err
find $O -name "*kernel_stats.csv" | while read f; do cp $f $O/$(basename $(dirname $f))_$(basename $f); done
ls $O
