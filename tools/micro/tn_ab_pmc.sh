set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tn_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AB_PMC=1
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
timeout -k 10 200 rocprofv3 --pmc $SQ --output-format csv -d $O/sq -o ab -- python3 $R/tools/gemm_tn_ab.py > $O/sq.log 2>&1
SQ2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"
timeout -k 10 200 rocprofv3 --pmc $SQ2 --output-format csv -d $O/sq2 -o ab -- python3 $R/tools/gemm_tn_ab.py > $O/sq2.log 2>&1 || echo "sq2 failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o ab -- python3 $R/tools/gemm_tn_ab.py > $O/trace.log 2>&1
python3 $R/profiles/summarize_pmc.py $(find $O/sq -name '*counter_collection.csv') > $O/sq_summary.csv
python3 $R/profiles/summarize_pmc.py $(find $O/sq2 -name '*counter_collection.csv') > $O/sq2_summary.csv || true
cp $(find $O/trace -name '*kernel_stats.csv') $O/kernel_stats.csv
