"""One iteration of the config-1 interpolator loop as a timeline: kernel, duration, gap to the previous kernel's end (from a rocprofv3
--kernel-trace CSV).  Usage: python tools/cfg1_timeline.py <kernel_trace.csv> [iteration index from the end, default 5]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
# an iteration starts at the image kernel that follows a tail kernel (or the first image kernel)
starts = [i for i, r in enumerate(rows) if "afi_wk6_image_kernel" in r["Kernel_Name"] and (i == 0 or "tail" in rows[i - 1]["Kernel_Name"] or "unpack" in rows[i - 1]["Kernel_Name"])]
if len(starts) < back + 2:
    print("too few iterations", len(starts)); sys.exit(1)
a, b = starts[-back - 1], starts[-back]
prev_end = None
tot_k = tot_g = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{r['Kernel_Name'][:70]:70s} grid {r.get('Grid_Size_X', '?'):>8s} wg {r.get('Workgroup_Size_X', '?'):>5s}  {(e - s) / 1e3:7.2f} us  gap {gap:6.2f} us")
    tot_k += (e - s) / 1e3; tot_g += gap
    prev_end = e
print(f"{b - a} launches, kernel time {tot_k:.1f} us, gaps {tot_g:.1f} us, span {(int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3:.1f} us")
