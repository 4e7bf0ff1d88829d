# FETCH_SIZE / WRITE_SIZE and time of the batched Winograd NT GEMM on its largest shapes, for two library builds (tools/micro/lib_A.so, lib_B.so).
# Each build is selected with AFI_LIB_PATH (afigan_amd/_lib.py): the library in the tree is never overwritten.  rocprofv3 runs python3 directly
# (no env / shell hop between the profiler and the program), the variable is exported around it.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nt_ab; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in A B; do
  export AFI_LIB_PATH=$R/tools/micro/lib_$v.so
  python3 $R/tools/gemm_nt_dtype.py 16 33664 1024 1024 36 8448 1024 1024 > $O/time_$v.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch_$v -o nt -- python3 $R/tools/gemm_nt_dtype.py 16 33664 1024 1024 36 8448 1024 1024 > $O/fetch_$v.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write_$v -o nt -- python3 $R/tools/gemm_nt_dtype.py 16 33664 1024 1024 36 8448 1024 1024 > $O/write_$v.log 2>&1
  python3 $R/profiles/summarize_pmc.py $(find $O/fetch_$v -name "*counter_collection.csv") $(find $O/write_$v -name "*counter_collection.csv") > $O/pmc_$v.csv
  echo "== $v"; grep "f16x3\|bf16x6" $O/time_$v.log; grep "afi_gemm_nt" $O/pmc_$v.csv | cut -c1-400
done
unset AFI_LIB_PATH
