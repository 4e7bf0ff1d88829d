"""config-1 AF-interpolator fwd+bwd under several library switches, one subprocess each (the switches are read once per process).
Usage: python tools/cfg1_ab.py "AFI_SK=0 AFI_WG_GROUP=0" "AFI_SK=1" ...   ("" = defaults)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import json, sys
sys.path.insert(0, %r)
import torch, bench
import afigan_amd as amd
r = bench.interp_bench(amd, torch, 1, 25, 34, iters=200, warmup=20)
print("RESULT " + json.dumps({k: r[k] for k in ("ms", "ms_eager", "ms_graph", "ms_host_enqueue", "tflops")}))
''' % ROOT
for spec in (sys.argv[1:] or [""]):
    env = dict(os.environ)
    for kv in spec.split():
        k, v = kv.split("=", 1)
        env[k] = v
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    print(f"[{spec or 'defaults'}] " + (line[0][7:] if line else "FAILED\n" + out.stdout[-2000:] + out.stderr[-3000:]), flush=True)
