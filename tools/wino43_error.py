"""Dev tool (CPU): fp32 error of Winograd F(4x4,3x3) against F(2x2,3x3) and the direct fp32 convolution on the REAL operands of the
network's stride-1 3x3 layers (oracle activations, closed-form and PyTorch-default weights), vs the fp64 direct convolution.
VERDICT r3 item 3: build an F(4x4,3x3) kernel only if outputs stay <= 1e-5 (of the tensor maximum).
usage: python tools/wino43_error.py"""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ssie_oracle as O

BT4 = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
G4 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float64)
AT4 = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
BT2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def wino(x, w, BT, G, AT, m, dt):
    """x (N,C,H,W), w (O,C,3,3), H, W multiples of m; all arithmetic in dtype dt"""
    n, c, h, wd = x.shape
    a = m + 2
    xp = F.pad(x.to(dt), (1, 1, 1, 1))
    pt = xp.unfold(2, a, m).unfold(3, a, m)                         # (N,C,th,tw,a,a)
    BTd, Gd, ATd = BT.to(dt), G.to(dt), AT.to(dt)
    V = torch.einsum("ij,nctujk,lk->nctuil", BTd, pt, BTd)          # B^T d B
    U = torch.einsum("ij,ocjk,lk->ocil", Gd, w.to(dt), Gd)          # G g G^T
    M = torch.einsum("nctuil,ocil->notuil", V, U)
    Y = torch.einsum("ij,notujk,lk->notuil", ATd, M, ATd)           # (N,O,th,tw,m,m)
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], h, wd)


def report(name, x, w):
    ref = F.conv2d(x.double(), w.double(), padding=1)
    sc = ref.abs().max().item()
    out = {}
    out["direct fp32"] = F.conv2d(x.float(), w.float(), padding=1).double()
    out["F(2x2,3x3) fp32"] = wino(x, w, BT2, G2, AT2, 2, torch.float32).double()
    out["F(4x4,3x3) fp32"] = wino(x, w, BT4, G4, AT4, 4, torch.float32).double()
    line = f"{name:34s} max|ref| {sc:9.3e} "
    for k, v in out.items():
        e = (v - ref).abs().max().item() / sc
        r = (v - ref).norm().item() / ref.norm().item()
        line += f"| {k}: max/scale {e:.2e} relL2 {r:.2e} "
    print(line)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    bands, hw = 31, 64
    for label, P in (("closed-form", O.closed_form_params(bands)),):
        x = O.synthetic_patches(2, bands, hw, hw)
        tr = {}
        with torch.no_grad():
            O.enhance_forward({k: v.double() for k, v in P.items()}, x.double(), tr)
        d = "decomposition_net."; i = "illum_adjust_net."
        layers = [("conv1 64->64", tr["sh_1"], P[d + "conv1.0.weight"]), ("conv3 128->128 (32x32)", tr["c2_1"], P[d + "conv3.0.weight"]),
                  ("conv5 128->64", torch.cat([tr["dc_1"], tr["c1_1"]], 1), P[d + "conv5.0.weight"]),
                  ("conv7 96->64", torch.cat([tr["c5_1"], tr["c0_1"]], 1), P[d + "conv7.0.weight"]),
                  ("recon 64->32", tr["c7_1"], P[d + "recon.weight"]), ("illum conv0 32->64", torch.cat([tr["R"], tr["I"]], 1), P[i + "conv0.0.weight"])]
        for nm, a, w in layers:
            report(f"[{label}] {nm}", a.detach(), w)
    # PyTorch default init (what bench.py times) on unit-scale random activations, and a data-gradient-like operand (heavy-tailed)
    for cin, cout in ((64, 64), (128, 128)):
        w = (torch.rand(cout, cin, 3, 3, dtype=torch.float64) * 2 - 1) / np.sqrt(cin * 9)
        a = torch.relu(torch.randn(2, cin, 64, 64, dtype=torch.float64))
        report(f"[default init] relu(randn) {cin}->{cout}", a, w)
        g = torch.randn(2, cin, 64, 64, dtype=torch.float64) * torch.rand(2, cin, 64, 64, dtype=torch.float64) ** 8 * 1e-3
        report(f"[default init] gradient-like {cin}->{cout}", g, w)


if __name__ == "__main__":
    main()
