#!/bin/bash
# Rebuild profiles/r02/ from gpurun_out/prof_r02 (produced by tools/prof_r02.sh on the GPU box).
set -e
P=${1:-gpurun_out/prof_r02}
D=profiles/r02
mkdir -p $D
cp $(find $P/trace -name "*kernel_stats.csv" | head -1) $D/kernel_stats_bench_steps3_warmup0.csv
grep '^{' $P/bench_trace.log > $D/bench_line_under_rocprof_steps3_warmup0.json
cp $(find $P/trace_serial -name "*kernel_stats.csv" | head -1) $D/kernel_stats_bench_steps3_warmup0_one_stream.csv
grep '^{' $P/bench_trace_serial.log > $D/bench_line_under_rocprof_steps3_warmup0_one_stream.json
python profiles/summarize_pmc.py $(find $P/pmc_sq_serial -name "*counter_collection.csv" | head -1) > $D/pmc_sq_steps1_one_stream.csv
cp $(find $P/trace_fp32 -name "*kernel_stats.csv" | head -1) $D/kernel_stats_bench_steps3_warmup0_dtype_fp32.csv
grep '^{' $P/bench_trace_fp32.log > $D/bench_line_under_rocprof_steps3_warmup0_dtype_fp32.json
grep '^{' $P/bench_default.log > $D/bench_line_default_run.json
python profiles/summarize_pmc.py $(find $P/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $P/pmc_write -name "*counter_collection.csv" | head -1) > $D/pmc_hbm_fetch_write_steps1.csv
python profiles/summarize_pmc.py $(find $P/pmc_sq -name "*counter_collection.csv" | head -1) > $D/pmc_sq_steps1.csv
cp $(find $P/cfg1_trace -name "*kernel_stats.csv" | head -1) $D/kernel_stats_cfg1_interpolator_100iters.csv
python profiles/summarize_pmc.py $(find $P/cfg1_fetch -name "*counter_collection.csv" | head -1) $(find $P/cfg1_write -name "*counter_collection.csv" | head -1) > $D/pmc_hbm_fetch_write_cfg1_20iters.csv
python profiles/summarize_pmc.py $(find $P/cfg1_sq -name "*counter_collection.csv" | head -1) > $D/pmc_sq_cfg1_20iters.csv
grep -h "ms_eager" $P/cfg1_trace.log $P/cfg1_default.log > $D/cfg1_loop_lines.txt || true
python - <<'PY'
import json, csv, subprocess
D = 'profiles/r02'
rows = list(csv.DictReader(open(f'{D}/pmc_hbm_fetch_write_steps1.csv')))
d = json.loads(open(f'{D}/bench_line_under_rocprof_steps3_warmup0.json').read())
dom = max((x for x in rows if 'afi_gemm_nt' in x['kernel']), key=lambda x: float(x['FETCH_SIZE_sum']))
fetch = float(dom['FETCH_SIZE_per_dispatch']) * 1024 * 2
write = float(dom['WRITE_SIZE_per_dispatch']) * 1024
head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
out = {"kernel": dom['kernel'], "dispatches_in_pass": int(dom['dispatches']), "measured_at": head,
       "fetch_size_kb_per_launch_raw": float(dom['FETCH_SIZE_per_dispatch']), "write_size_kb_per_launch_raw": float(dom['WRITE_SIZE_per_dispatch']),
       "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0 --no-interp --no-cpu-baseline`; "
               "FETCH_SIZE x2 (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B), both x1024 (KB units); average over the "
               "launches of the dominant kernel in that step"}
# clock the chip holds under the dominant kernel and its matrix-pipe duty there (SQ pass of the same command): SQ_BUSY_CYCLES is summed over
# the 32 shader engines, SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
sq = {x['kernel']: x for x in csv.DictReader(open(f'{D}/pmc_sq_steps1_one_stream.csv'))}.get(dom['kernel'])
ks_all = {x['Name']: x for x in csv.DictReader(open(f'{D}/kernel_stats_bench_steps3_warmup0_one_stream.csv'))}
if sq and dom['kernel'] in ks_all:
    cyc = float(sq['SQ_BUSY_CYCLES_sum']) / 32 / int(sq['dispatches'])
    out["held_clock_ghz"] = round(cyc / float(ks_all[dom['kernel']]['AverageNs']), 3)
    out["mfma_busy_at_held_clock"] = round(float(sq['SQ_VALU_MFMA_BUSY_CYCLES_sum']) / (float(sq['SQ_BUSY_CYCLES_sum']) / 32 * 1024), 4)
    out["clock_note"] = "one-stream passes (AFI_D_OVERLAP=0: the kernel alone on the chip): SQ_BUSY_CYCLES / 32 shader engines / launches / rocprofv3 average duration; SQ_VALU_MFMA_BUSY_CYCLES / (those cycles x 1024 SIMDs)"
json.dump(out, open(f'{D}/traffic_dominant_kernel.json', 'w'), indent=1)
r = d['roofline']
print('step  live HIP events :', r['kernel'], r['launches'], 'launches, avg', round(r['avg_launch_us'], 1), 'us,', round(r['achieved'], 1), 'TFLOP/s')
ks = list(csv.DictReader(open(f'{D}/kernel_stats_bench_steps3_warmup0.csv')))[0]
print('step  rocprofv3 stats :', ks['Name'][:60], ks['Calls'], 'calls, avg', round(float(ks['AverageNs']) / 1e3, 1), 'us')
print('step  traffic         :', round(out['hbm_bytes_per_launch'] / 1e9, 3), 'GB per launch')
ks = list(csv.DictReader(open(f'{D}/kernel_stats_cfg1_interpolator_100iters.csv')))
tot = sum(float(k['TotalDurationNs']) for k in ks)
IT = 130   # 10 warm-up + 100 timed + 20 with the library's HIP-event brackets (interp_bench's roofline leg)
print('cfg1  kernel time per iteration (130 iterations in the run):', round(tot / IT / 1e3, 1), 'us;', round(sum(int(k["Calls"]) for k in ks) / IT, 1), 'launches')
for k in ks[:4]:
    print('      ', k['Name'][:70], 'calls/iter', round(int(k['Calls']) / 130, 1), 'avg', round(float(k['AverageNs']) / 1e3, 1), 'us')
PY
