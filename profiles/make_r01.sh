#!/bin/bash
# Rebuild profiles/r01/ from a gpurun_out/prof_* directory produced by the rocprofv3 commands in profiles/README.md
set -e
P=${1:?usage: make_r01.sh gpurun_out/prof_dir}
D=profiles/r01
mkdir -p $D
cp $(find $P/trace -name "*kernel_stats.csv" | head -1) $D/kernel_stats_bench_steps3_warmup0.csv
grep '^{' $P/bench_trace.log > $D/bench_line_under_rocprof_steps3_warmup0.json
python profiles/summarize_pmc.py $(find $P/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $P/pmc_write -name "*counter_collection.csv" | head -1) > $D/pmc_hbm_fetch_write_steps1.csv
python profiles/summarize_pmc.py $(find $P/pmc_sq -name "*counter_collection.csv" | head -1) > $D/pmc_sq_steps1.csv
python - <<'PY'
import json, csv
rows = list(csv.DictReader(open('profiles/r01/pmc_hbm_fetch_write_steps1.csv')))
d = json.loads(open('profiles/r01/bench_line_under_rocprof_steps3_warmup0.json').read())
dom = max((x for x in rows if ('afi_pix_gemm_kernel<128, 128, 2, 2, false' in x['kernel'] or 'afi_gemm_nt_kernel' in x['kernel'])), key=lambda x: float(x['FETCH_SIZE_sum']))
fetch = float(dom['FETCH_SIZE_per_dispatch']) * 1024 * 2
write = float(dom['WRITE_SIZE_per_dispatch']) * 1024
out = {"kernel": dom['kernel'], "dispatches_in_pass": int(dom['dispatches']),
       "fetch_size_kb_per_launch_raw": float(dom['FETCH_SIZE_per_dispatch']), "write_size_kb_per_launch_raw": float(dom['WRITE_SIZE_per_dispatch']),
       "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 1 --warmup 0 --no-interp --no-cpu-baseline`; "
               "FETCH_SIZE x2 (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B), both x1024 (KB units); average over the "
               "launches of the dominant kernel in that step"}
json.dump(out, open('profiles/r01/traffic_dominant_kernel.json', 'w'), indent=1)
r = d['roofline']
print('live HIP events :', r['kernel'], r['launches'], 'launches, avg', round(r['avg_launch_us'], 1), 'us,', round(r['achieved'], 1), 'TFLOP/s')
ks = list(csv.DictReader(open('profiles/r01/kernel_stats_bench_steps3_warmup0.csv')))[0]
print('rocprofv3 stats :', ks['Name'][:60], ks['Calls'], 'calls, avg', round(float(ks['AverageNs']) / 1e3, 1), 'us')
print('traffic         :', round(out['hbm_bytes_per_launch'] / 1e9, 3), 'GB per launch')
PY
