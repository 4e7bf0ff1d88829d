import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import afigan_amd as amd
from oracle import afigan_oracle as orc
torch.manual_seed(0)
for shape in [(2,256,13,21)]:
    D = amd.Discriminator().cuda(); dp = orc.closed_form_discriminator_params(); D.load_state_dict(dp); D.train()
    x_cpu = torch.randn(shape, generator=torch.Generator().manual_seed(11))
    R = torch.randn((shape[0],1,shape[2],shape[3]), generator=torch.Generator().manual_seed(12))
    x = x_cpu.cuda().requires_grad_(True)
    logits = D(x); (logits*R.cuda()).sum().backward()
    p = {k:(v.clone().requires_grad_(True) if v.dtype.is_floating_point and 'running' not in k else v.clone()) for k,v in dp.items()}
    xr = x_cpu.clone().requires_grad_(True)
    # double precision oracle
    p64 = {k:(v.double().clone().requires_grad_(True) if v.dtype.is_floating_point and 'running' not in k else (v.double() if v.dtype.is_floating_point else v.clone())) for k,v in dp.items()}
    x64 = x_cpu.double().clone().requires_grad_(True)
    l64,_ = orc.discriminator_forward(x64, p64); (l64*R.double()).sum().backward()
    lr,_ = orc.discriminator_forward(xr, p); (lr*R).sum().backward()
    def rel(a,b): return ((a.double().cpu()-b.double()).norm()/b.double().norm()).item()
    print('logits gpu-vs-64', rel(logits.detach(), l64.detach()), ' cpu32-vs-64', rel(lr.detach(), l64.detach()))
    print('dx     gpu-vs-64', rel(x.grad, x64.grad), ' cpu32-vs-64', rel(xr.grad, x64.grad))
    for k,q in D.named_parameters():
        print(k, 'gpu-vs-64 %.2e'%rel(q.grad.contiguous(), p64[k].grad), ' cpu32-vs-64 %.2e'%rel(p[k].grad, p64[k].grad))
