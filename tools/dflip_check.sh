#!/bin/bash
# Gradient deviation of the discriminator (fwd+bwd, train-mode BN) from an fp64 evaluation, per conv algorithm; $1 $2 = H W
set -e
H=${1:-100}; W=${2:-168}
python tools/dflip_check.py torch64 $H $W
python tools/dflip_check.py torch32 $H $W
python tools/dflip_check.py direct $H $W
python tools/dflip_check.py f2fwd $H $W
python tools/dflip_check.py f4fwd $H $W
python tools/dflip_check.py f2all $H $W
python tools/dflip_cmp.py torch32 direct f2all f2fwd f4fwd
rm -f gpurun_out/dflip_*.pt
