"""Per-stream timeline of a rocprofv3 kernel trace of bench.py (two-stream stage-1 step): for the second-to-last step in the trace (the last
one issues no guide forwards for a following batch when the run has no warm-up), each stream's busy
time, the wall time, and the time during which only ONE stream has a kernel running, by kernel family.
Usage: python tools/stream_timeline.py <kernel_trace.csv> [steps_in_trace]"""
import csv
import sys
from collections import defaultdict


def fam(n):
    for k in ("gemm_nt", "gemm_tn", "wino4_input", "wino_input", "wino4_dy", "wino_dy", "wino4_output", "wino_output", "bn_", "colred", "wk6", "pix_gemm", "wgrad", "Cijk", "fillBuffer", "sgd", "split", "absmax", "wino4_weight", "wino_weight"):
        if k in n:
            return k
    return "other"


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"] if "Stream_Id" in r else r["Queue_Id"], r["Kernel_Name"]) for r in rows))
    # the steps are separated by the optimizer's SGD launches: take the window between the last two G-optimizer steps
    sgd = [e for e in ev if "sgd" in e[3]]
    if len(sgd) < 5:
        print("not enough steps in the trace"); return
    back = 2 if len(sgd) >= 5 else 0          # two SGD launches per step (D, G)
    t1 = sgd[-1 - back][1]; t0 = sgd[-3 - back][1]
    win = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    wall = (t1 - t0) / 1e6
    busy = defaultdict(float)
    for s, e, q, n in win:
        busy[q] += (e - s) / 1e6
    print(f"step window {wall:.2f} ms; kernels {len(win)}; busy per stream (ms): " + ", ".join(f"{q}: {b:.1f}" for q, b in sorted(busy.items(), key=lambda kv: -kv[1])))
    # sweep: time with 0 / 1 / >= 2 kernels running, and who runs alone
    pts = []
    for i, (s, e, q, n) in enumerate(win):
        pts.append((s, 1, i)); pts.append((e, -1, i))
    pts.sort()
    active = set(); last = t0
    t_n = defaultdict(float); alone = defaultdict(float); idle_after = defaultdict(float)
    for t, d, i in pts:
        dt = (t - last) / 1e6
        k = len(active)
        t_n[min(k, 2)] += dt
        if k == 1:
            alone[fam(win[next(iter(active))][3])] += dt
        last = t
        if d > 0: active.add(i)
        else: active.discard(i)
    print(f"time with 0 / 1 / >=2 kernels running: {t_n[0]:.2f} / {t_n[1]:.2f} / {t_n[2]:.2f} ms")
    print("running ALONE (ms): " + ", ".join(f"{k}: {v:.2f}" for k, v in sorted(alone.items(), key=lambda kv: -kv[1])[:14]))
    tot = defaultdict(float)
    for s, e, q, n in win:
        tot[fam(n)] += (e - s) / 1e6
    print("kernel time by family (ms): " + ", ".join(f"{k}: {v:.2f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:16]))


if __name__ == "__main__":
    main()
