import sys, torch
ref = torch.load("gpurun_out/dflip_torch64.pt")
for tag in sys.argv[1:]:
    o = torch.load(f"gpurun_out/dflip_{tag}.pt")
    # the conv biases ahead of train-mode BatchNorm have an identically zero gradient (fp64 shows ~1e-17 noise): skip them
    live = [k for k in ref if ref[k].norm() > 1e-9 * ref[k].numel() ** 0.5]
    worst = max(((o[k] - ref[k]).norm() / ref[k].norm()).item() for k in live)
    dx = ((o["dx"] - ref["dx"]).norm() / ref["dx"].norm()).item()
    print(f"{tag:10s} vs fp64: dx rel-L2 {dx:.3e}   worst tensor rel-L2 {worst:.3e}", flush=True)
