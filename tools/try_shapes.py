"""Dev probe: run the plan on unusual geometries and compare with the CPU oracle (small ones only)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import ssie; ssie.load()
from ssie_amd import hostlib as H
from oracle import ssie_oracle as O

def run(n, bands, h, w, backward=True, check=True):
    table, total = H.param_table(bands)
    P = O.closed_form_params(bands)
    flat = torch.zeros(total, device="cuda")
    for name, off, shape in table:
        flat[off:off + P[name].numel()] = P[name].reshape(-1).cuda()
    g = torch.zeros_like(flat)
    t0 = time.time()
    plan = H.Plan(n, bands, h, w, O.JYU_COEFS, flat, g)
    x = O.synthetic_patches(n, bands, h, w)
    xc = x.cuda()
    if backward:
        plan.loss_fwd_bwd(xc, backward=True)
    else:
        plan.enhance_fwd(xc)
    torch.cuda.synchronize()
    t1 = time.time()
    S = plan.nchw("S").cpu()
    msg = f"n{n} b{bands} {h}x{w} bwd={backward}: gpu {t1-t0:.2f}s finite={bool(torch.isfinite(S).all())}"
    if check:
        t2 = time.time()
        P64 = {k: v.double() for k, v in P.items()}
        if backward:
            vals, grads, outs = O.loss_and_grads(P64, x.double(), O.JYU_COEFS)
            Sx = outs[3]
            gw = {}
            for name, off, shape in table:
                gg = g[off:off + grads[name].numel()].cpu().double().reshape(grads[name].shape)
                gw[name] = (gg - grads[name]).norm().item() / max(grads[name].norm().item(), 1e-30)
            worst = sorted(gw.items(), key=lambda kv: -kv[1])[:3]
            msg += f" | oracle loss {vals['total'] if 'total' in vals else list(vals.values())[0]:.6e} worst grads {worst}"
        else:
            Sx = O.enhance_forward(P64, x.double())[3]
        err = (S.double() - Sx).abs().max().item()
        msg += f" | S maxabs err {err:.2e} oracle {time.time()-t2:.1f}s"
        if backward:
            sc = plan.loss_scalars()
            msg += f" | loss gpu {sc}"
    print(msg, flush=True)

if __name__ == "__main__":
    run(1, 31, 128, 128)
    run(1, 256, 64, 64)
    run(1, 31, 200, 264, backward=False)
    run(1, 31, 1024, 1024, backward=False, check=False)
