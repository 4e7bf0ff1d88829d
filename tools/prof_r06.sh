# Round-6 profiling passes (one gpurun call): the stage-1 step (metric 2) and the config-1 interpolator loop (metric 1).
# PMC passes are separate runs without any trace domain, as the pool requires, and run the step on ONE stream (--one-stream) so that a
# dispatch's counters are its own.  profiles/make_r06.sh turns gpurun_out/prof_r06 into profiles/r06/.
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_r06; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export AFI_BENCH_OTHER_DTYPES=0      # the profiled runs hold the default arithmetic only (the default run at the end reports every setting)
B="python3 $R/bench.py"
L="python3 $R/tools/cfg1_loop.py"
S="--no-interp --no-cpu-baseline --profile-timed"   # (the library's event brackets inside the timed region: the traces hold exactly the timed launches)
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
export AFI_PROFILE_DUMP=$O/launches_two_stream.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o step -- $B --steps 3 --warmup 0 $S > $O/bench_trace.log 2>&1
export AFI_PROFILE_DUMP=$O/launches_one_stream.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_serial -o step -- $B --steps 3 --warmup 0 $S --one-stream > $O/bench_trace_serial.log 2>&1
unset AFI_PROFILE_DUMP
echo step traces done
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o step -- $B --steps 1 --warmup 0 $S --one-stream > $O/bench_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o step -- $B --steps 1 --warmup 0 $S --one-stream > $O/bench_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_sq_serial -o step -- $B --steps 1 --warmup 0 $S --one-stream > $O/bench_sq_serial.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fp32 -o step -- $B --steps 3 --warmup 0 $S --dtype fp32 > $O/bench_trace_fp32.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bf16x6 -o step -- $B --steps 3 --warmup 0 $S --dtype bf16x6 --one-stream > $O/bench_trace_bf16x6.log 2>&1
echo step pmc done
python3 $R/tools/stream_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) > $O/stream_timeline_two_stream.txt 2>&1 || true
python3 $R/tools/gemm_nt_dtype.py > $O/gemm_dtypes.txt 2>&1 || true
bash $R/tools/micro/knob_ab.sh "" "--option f16_local_sums=0" "--option winograd_f4_forward=8 --option f16_local_sums=0" "--option winograd_f4_forward=0" "--option winograd_f4_forward=1 --option f16_local_sums=1" "--option d_fold_bn_apply=1" "--dtype bf16x6" > $O/knob_ab.txt 2>&1 || true
python3 $R/tools/micro/host_enqueue_probe.py 6 > $O/host_enqueue_probe.txt 2>&1 || true
python3 $R/tools/micro/guide_overlap_probe.py 5 > $O/guide_overlap_probe.txt 2>&1 || true
echo micro done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg1_trace -o cfg1 -- $L 100 > $O/cfg1_trace.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/cfg1_fetch -o cfg1 -- $L 20 > $O/cfg1_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/cfg1_write -o cfg1 -- $L 20 > $O/cfg1_write.log 2>&1
timeout -k 10 200 rocprofv3 --pmc $SQ --output-format csv -d $O/cfg1_sq -o cfg1 -- $L 20 > $O/cfg1_sq.log 2>&1
echo cfg1 passes done
# the widened rows: FPN top-down merge and the BiFPN training pass, per kernel
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fpn_trace -o fpn -- python3 $R/tools/fpn_loop.py fpn 10 > $O/fpn_loop.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pafpn_trace -o pafpn -- python3 $R/tools/fpn_loop.py pafpn 10 > $O/pafpn_loop.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bifpn_trace -o bifpn -- python3 $R/tools/bifpn_train_loop.py 5 > $O/bifpn_train_loop.txt 2>&1
python3 $R/tools/interp_sweep.py 1 25 42 1 50 84 1 56 88 2 50 68 8 25 34 1 64 128 1 100 168 > $O/interp_sweep_default.txt 2>&1 || true
python3 $R/tools/interp_sweep.py g_smallmap6_max_pixels=8192 1 50 84 1 56 88 2 50 68 8 25 34 1 64 128 > $O/interp_sweep_smallmap6_8192.txt 2>&1 || true
python3 $R/tools/gflip_check.py 52 84 > $O/gflip_interpolator_forwards.txt 2>&1 || true
echo widened rows done
unset AFI_BENCH_OTHER_DTYPES
cd $R && timeout -k 10 200 $L 200 g > $O/cfg1_default.log 2>&1
cd $R && timeout -k 10 500 python bench.py > $O/bench_default.log 2> $O/bench_default.err
echo default done
find $O -name "*.csv" | head -40
