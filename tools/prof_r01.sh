set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r01r; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o runc -- $B --steps 3 --warmup 0 --no-interp --no-cpu-baseline > $O/bench_trace.log 2>&1
echo trace done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o runc -- $B --steps 1 --warmup 0 --no-interp --no-cpu-baseline > $O/bench_fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o runc -- $B --steps 1 --warmup 0 --no-interp --no-cpu-baseline > $O/bench_write.log 2>&1
echo write done
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -o runc -- $B --steps 1 --warmup 0 --no-interp --no-cpu-baseline > $O/bench_sq.log 2>&1
echo sq done
cd $R && timeout -k 10 400 python bench.py > $O/bench_default.log 2>&1
echo default done
ls $O $O/trace | head -30
