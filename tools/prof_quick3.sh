# SQ counter pass (one stream) of the stage-1 step: MFMA duty, held clock, LDS conflicts per kernel
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_q3; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AFI_BENCH_OTHER_DTYPES=0
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq -o step -- python3 $R/bench.py --steps 1 --warmup 0 --no-interp --no-cpu-baseline --profile-timed --one-stream > $O/bench_sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o step -- python3 $R/bench.py --steps 1 --warmup 0 --no-interp --no-cpu-baseline --profile-timed --one-stream > $O/bench_tr.log 2>&1
echo done
