"""Experiment: the stage-1 step with its two streams confined to CU subsets (hipExtStreamCreateWithCUMask).
usage: cu_mask_step.py <maskA> <maskB>   each 'all' or a spec 'lo-hi' (bit range of the 256-bit CU mask) or 'even'/'odd'/'q0'..'q3' (bit i with i%4==k)"""
import ctypes, os, sys, json, io, contextlib, runpy
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch

hip = ctypes.CDLL("libamdhip64.so")


def mask_words(spec):
    bits = [0] * 256
    if spec == "all":
        bits = [1] * 256
    elif spec in ("even", "odd"):
        for i in range(256):
            bits[i] = 1 if (i % 2 == (0 if spec == "even" else 1)) else 0
    elif spec.startswith("m4_"):                 # m4_012: bits with i%4 in {0,1,2}
        ks = {int(c) for c in spec[3:]}
        for i in range(256):
            bits[i] = 1 if (i % 4) in ks else 0
    else:
        lo, hi = map(int, spec.split("-"))
        for i in range(lo, hi):
            bits[i] = 1
    words = (ctypes.c_uint32 * 8)()
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= (1 << (i % 32))
    return words, sum(bits)


def make_stream(spec):
    if spec == "all":
        return torch.cuda.Stream()
    words, n = mask_words(spec)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def main():
    a, b = sys.argv[1], sys.argv[2]
    torch.cuda.init()
    sa = make_stream(a)
    sb = make_stream(b)
    orig = torch.cuda.Stream
    made = []

    def fake(*args, **kw):
        made.append(1)
        return sb
    torch.cuda.Stream = fake
    os.environ["AFI_BENCH_OTHER_DTYPES"] = "0"
    sys.argv = ["bench.py", "--steps", "8", "--warmup", "3", "--no-interp", "--no-cpu-baseline"]
    buf = io.StringIO()
    with torch.cuda.stream(sa), contextlib.redirect_stdout(buf):
        runpy.run_path(os.path.join(R, "bench.py"), run_name="__main__")
    torch.cuda.Stream = orig
    line = [l for l in buf.getvalue().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    print(f"A={a} B={b} (second-stream objects handed out: {len(made)}): {d['ms_per_step']:.2f} ms/step, NT avg {d['roofline']['avg_launch_us']:.1f} us", flush=True)


if __name__ == "__main__":
    main()
