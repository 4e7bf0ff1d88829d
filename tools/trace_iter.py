"""Summarise one steady-state iteration out of a rocprofv3 --kernel-trace CSV: kernels, busy time, span, gaps.
usage: python tools/trace_iter.py <kernel_trace.csv> <kernels_per_iter or 0=auto> [iter_index_from_end]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 0
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if n == 0:   # period = distance between the last two occurrences of the rarest kernel name
    cnt = collections.Counter(names)
    rare = min(cnt, key=lambda k: cnt[k])
    idx = [i for i, k in enumerate(names) if k == rare]
    n = idx[-1] - idx[-2] if len(idx) > 1 else len(rows)
it = rows[len(rows) - back * n: len(rows) - (back - 1) * n]
t0 = int(it[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in it)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in it)
print(f"kernels/iter {n}  span {1e-3*(t1-t0):.1f} us  sum of kernel durations {1e-3*busy:.1f} us")
agg = collections.OrderedDict()
for r in it:
    k = r["Kernel_Name"][:90]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(k, [0, 0]); a[0] += 1; a[1] += d
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{c:4d} x {1e-3*d/c:8.1f} us = {1e-3*d:8.1f} us  {k}")
if "-v" in sys.argv:
    prev = t0
    for r in it:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"  +{1e-3*(s-t0):8.1f} gap {1e-3*(s-prev):6.1f} dur {1e-3*(e-s):6.1f}  q{r.get('Queue_Id','?')} {r['Kernel_Name'][:70]}")
        prev = max(prev, e)
