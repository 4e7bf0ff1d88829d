#!/bin/bash
# Rebuild profiles/r06/ from gpurun_out/prof_r06 (produced by tools/prof_r06.sh on the GPU box).  Run from the repo root, on the tree the
# passes were taken on: the traffic records are stamped with the sha256 of the kernel source they were measured on.
set -e
P=${1:-gpurun_out/prof_r06}
D=profiles/r06
mkdir -p $D
cp $(find $P/trace -name "*kernel_stats.csv" | head -1) $D/kernel_stats_bench_steps3_warmup0.csv
grep '^{' $P/bench_trace.log > $D/bench_line_under_rocprof_steps3_warmup0.json
cp $(find $P/trace_serial -name "*kernel_stats.csv" | head -1) $D/kernel_stats_bench_steps3_warmup0_one_stream.csv
grep '^{' $P/bench_trace_serial.log > $D/bench_line_under_rocprof_steps3_warmup0_one_stream.json
cp $P/launches_one_stream.csv $D/launches_bench_steps3_one_stream.csv
python profiles/summarize_pmc.py $(find $P/pmc_sq_serial -name "*counter_collection.csv" | head -1) > $D/pmc_sq_steps1_one_stream.csv
cp $(find $P/trace_fp32 -name "*kernel_stats.csv" | head -1) $D/kernel_stats_bench_steps3_warmup0_dtype_fp32.csv
cp $(find $P/trace_bf16x6 -name "*kernel_stats.csv" | head -1) $D/kernel_stats_bench_steps3_warmup0_one_stream_dtype_bf16x6.csv
grep '^{' $P/bench_trace_bf16x6.log > $D/bench_line_under_rocprof_steps3_warmup0_one_stream_dtype_bf16x6.json
cp $(find $P/fpn_trace -name "*kernel_stats.csv" | head -1) $D/kernel_stats_fpn_topdown_10iters.csv
cp $(find $P/pafpn_trace -name "*kernel_stats.csv" | head -1) $D/kernel_stats_pafpn_10iters.csv
cp $(find $P/bifpn_trace -name "*kernel_stats.csv" | head -1) $D/kernel_stats_bifpn_train_5iters.csv
for f in stream_timeline_two_stream.txt gemm_dtypes.txt knob_ab.txt host_enqueue_probe.txt guide_overlap_probe.txt fpn_loop.txt pafpn_loop.txt bifpn_train_loop.txt interp_sweep_default.txt interp_sweep_smallmap6_8192.txt gflip_interpolator_forwards.txt; do [ -f $P/$f ] && grep -v "amdgpu.ids" $P/$f > $D/$f || true; done
grep '^{' $P/bench_trace_fp32.log > $D/bench_line_under_rocprof_steps3_warmup0_dtype_fp32.json
grep '^{' $P/bench_default.log > $D/bench_line_default_run.json
python profiles/summarize_pmc.py $(find $P/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $P/pmc_write -name "*counter_collection.csv" | head -1) > $D/pmc_hbm_fetch_write_steps1_one_stream.csv
cp $(find $P/cfg1_trace -name "*kernel_stats.csv" | head -1) $D/kernel_stats_cfg1_interpolator_100iters.csv
python profiles/summarize_pmc.py $(find $P/cfg1_fetch -name "*counter_collection.csv" | head -1) $(find $P/cfg1_write -name "*counter_collection.csv" | head -1) > $D/pmc_hbm_fetch_write_cfg1_20iters.csv
python profiles/summarize_pmc.py $(find $P/cfg1_sq -name "*counter_collection.csv" | head -1) > $D/pmc_sq_cfg1_20iters.csv
grep -h "ms_eager" $P/cfg1_trace.log $P/cfg1_default.log > $D/cfg1_loop_lines.txt || true
python - <<'PY'
import csv, hashlib, json, subprocess
D = 'profiles/r06'
head = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
sha = lambda p: hashlib.sha256(open(p, 'rb').read()).hexdigest()
NOTE = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `%s`; FETCH_SIZE x2 (gfx950 tallies the 128-B requests of wide "
        "coalesced reads at 64 B), both x1024 (KB units); average over the launches of the dominant kernel in that pass")


def record(pmc_csv, pick, source, cmd):
    rows = list(csv.DictReader(open(pmc_csv)))
    dom = max((x for x in rows if pick(x['kernel'])), key=lambda x: float(x['FETCH_SIZE_sum']) + float(x['WRITE_SIZE_sum']))
    fetch = float(dom['FETCH_SIZE_per_dispatch']) * 1024 * 2
    write = float(dom['WRITE_SIZE_per_dispatch']) * 1024
    return dom, {"kernel": dom['kernel'], "dispatches_in_pass": int(dom['dispatches']), "measured_at": head, "kernel_source": source, "kernel_source_sha256": sha(source),
                 "fetch_size_kb_per_launch_raw": float(dom['FETCH_SIZE_per_dispatch']), "write_size_kb_per_launch_raw": float(dom['WRITE_SIZE_per_dispatch']),
                 "hbm_read_bytes_per_launch": fetch, "hbm_write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write, "note": NOTE % cmd}


# ---- the stage-1 step: the batched Winograd NT GEMM
dom, out = record(f'{D}/pmc_hbm_fetch_write_steps1_one_stream.csv', lambda k: 'afi_gemm_nt' in k, 'afigan_amd/csrc/afi_gemm_f16.h',
                  'bench.py --steps 1 --warmup 0 --no-interp --no-cpu-baseline --one-stream')
# algorithmic bytes per launch from the per-launch shapes of the one-stream trace pass (afi_profile_dump): A rows x K fp32 read, the
# pre-split weights planes x N x K x 2 bytes x parts read, C rows x N fp32 written
# (the f16x3 NT GEMM runs as four kernels -- the 256 x 256 tile on the large shapes and the 128 x 128 tile on the rest, each on fp32 or on pre-split A planes; the per-launch dump's `split` column says which --: the record is the
#  one with the larger traffic, and its algorithmic bytes are those of ITS launches)
kn = dom['kernel']
w16, pre, local = 'w16' in kn, ('<true' in kn or 'true>' in kn), '<true, true>' in kn     # (w16 kernel: <APRE, LOCAL>; 128-tile kernel: <MINW, ABL, APRE>)
code = 7 if (w16 and local) else {(False, False): 2, (True, False): 3, (True, True): 4, (False, True): 5}[(w16, pre)]   # (igemm.hip: the dump's `split` column names the kernel)
ln = [r for r in csv.DictReader(open(f'{D}/launches_bench_steps3_one_stream.csv')) if r['kind'].startswith('gemm_nt_f16x3') and int(r['split']) == code]
if ln:
    alg = [4 * int(r['rows']) * int(r['k']) + 4 * int(r['planes']) * int(r['cols']) * int(r['k']) + 4 * int(r['rows']) * int(r['cols']) for r in ln]
    out["algorithmic_bytes_per_launch"] = sum(alg) / len(alg)
    out["algorithmic_note"] = "mean over the %d launches of this kernel in 3 steps: 4*rows*K (A: fp32, or two fp16 pieces) + 4*planes*N*K (weights as two fp16 pieces) + 4*rows*N (C)" % len(alg)
sq = {x['kernel']: x for x in csv.DictReader(open(f'{D}/pmc_sq_steps1_one_stream.csv'))}.get(dom['kernel'])
ks_all = {x['Name'][:110]: x for x in csv.DictReader(open(f'{D}/kernel_stats_bench_steps3_warmup0_one_stream.csv'))}
if sq and dom['kernel'] in ks_all:
    cyc = float(sq['SQ_BUSY_CYCLES_sum']) / 32 / int(sq['dispatches'])
    out["held_clock_ghz"] = round(cyc / float(ks_all[dom['kernel']]['AverageNs']), 3)
    out["mfma_busy_at_held_clock"] = round(float(sq['SQ_VALU_MFMA_BUSY_CYCLES_sum']) / (float(sq['SQ_BUSY_CYCLES_sum']) / 32 * 1024), 4)
    out["clock_note"] = ("one-stream passes (the kernel alone on the chip): SQ_BUSY_CYCLES / 32 shader engines / launches / rocprofv3 average duration; "
                         "SQ_VALU_MFMA_BUSY_CYCLES / (those cycles x 1024 SIMDs)")
json.dump(out, open(f'{D}/traffic_dominant_kernel.json', 'w'), indent=1)
# the default-run line of the same call was printed BEFORE this record existed (bench.py looks the committed record up): put the record of
# these very passes into it, and say so
try:
    dl = json.loads(open(f'{D}/bench_line_default_run.json').read())
    dl['roofline']['traffic'] = {k: out.get(k) for k in ('hbm_bytes_per_launch', 'algorithmic_bytes_per_launch', 'measured_at', 'kernel_source', 'kernel_source_sha256',
                                                          'held_clock_ghz', 'mfma_busy_at_held_clock')}
    dl['roofline']['traffic']['source'] = 'profiles/r06/traffic_dominant_kernel.json, written by profiles/make_r06.sh from the PMC passes of the same gpurun call as this line (re-emitted: the line was printed before the record existed)'
    open(f'{D}/bench_line_default_run.json', 'w').write(json.dumps(dl) + '\n')
except (OSError, ValueError, KeyError) as e:
    print('default line not patched:', e)
# the same three figures for the weight-gradient GEMM, for DESIGN 4b
tn = next((x for x in csv.DictReader(open(f'{D}/pmc_sq_steps1_one_stream.csv')) if 'afi_gemm_tn' in x['kernel']), None)
if tn and tn['kernel'] in ks_all:
    cyc = float(tn['SQ_BUSY_CYCLES_sum']) / 32 / int(tn['dispatches'])
    print('TN   held clock', round(cyc / float(ks_all[tn['kernel']]['AverageNs']), 3), 'GHz, MFMA busy', round(float(tn['SQ_VALU_MFMA_BUSY_CYCLES_sum']) / (float(tn['SQ_BUSY_CYCLES_sum']) / 32 * 1024), 3))

# ---- the config-1 interpolator loop: its dominant GEMM kernel
ks = list(csv.DictReader(open(f'{D}/kernel_stats_cfg1_interpolator_100iters.csv')))
top = next(k for k in ks if any(t in k['Name'] for t in ('pix_gemm', 'wgrad', 'gemm_nt', 'gemm_tn')))     # (first row = largest total time)
src = 'afigan_amd/csrc/smallmap.hip' if any(t in top['Name'] for t in ('_wk_', '_wk6_', '_sk_', 'group', 'wgrad6')) else 'afigan_amd/csrc/igemm.hip'
dom1, out1 = record(f'{D}/pmc_hbm_fetch_write_cfg1_20iters.csv', lambda k: k == top['Name'][:110], src, 'tools/cfg1_loop.py 20')
out1["avg_launch_us_trace"] = float(top['AverageNs']) / 1e3
json.dump(out1, open(f'{D}/traffic_cfg1_dominant_kernel.json', 'w'), indent=1)

d = json.loads(open(f'{D}/bench_line_under_rocprof_steps3_warmup0.json').read())
r = d['roofline']
print('step  live HIP events :', r['kernel'], r['launches'], 'launches, avg', round(r['avg_launch_us'], 1), 'us,', round(r['achieved'], 1), 'TFLOP/s')
k0 = list(csv.DictReader(open(f'{D}/kernel_stats_bench_steps3_warmup0.csv')))[0]
print('step  rocprofv3 stats :', k0['Name'][:60], k0['Calls'], 'calls, avg', round(float(k0['AverageNs']) / 1e3, 1), 'us')
k1 = list(csv.DictReader(open(f'{D}/kernel_stats_bench_steps3_warmup0_one_stream.csv')))[0]
print('step  one stream      :', k1['Name'][:60], k1['Calls'], 'calls, avg', round(float(k1['AverageNs']) / 1e3, 1), 'us')
print('step  traffic         :', round(out['hbm_bytes_per_launch'] / 1e9, 3), 'GB per launch; algorithmic', round(out.get('algorithmic_bytes_per_launch', 0) / 1e9, 3), 'GB; clock', out.get('held_clock_ghz'), 'busy', out.get('mfma_busy_at_held_clock'))
tot = sum(float(k['TotalDurationNs']) for k in ks)
IT = 130   # 10 warm-up + 100 timed + 20 with the library's HIP-event brackets (interp_bench's roofline leg)
print('cfg1  kernel time per iteration (130 iterations in the run):', round(tot / IT / 1e3, 1), 'us;', round(sum(int(k["Calls"]) for k in ks) / IT, 1), 'launches')
for k in ks[:4]:
    print('      ', k['Name'][:70], 'calls/iter', round(int(k['Calls']) / 130, 1), 'avg', round(float(k['AverageNs']) / 1e3, 1), 'us')
print('cfg1  traffic         :', out1['kernel'][:60], round(out1['hbm_bytes_per_launch'] / 1e6, 3), 'MB per launch')
PY
