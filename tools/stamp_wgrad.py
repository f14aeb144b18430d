"""Diagnostic (dev tool): -DSSIE_STAMP build of the library in /tmp; run one conv weight-gradient and print where wave 0
of each workgroup spends its cycles (staging / barriers / MFMA loop / slab write)."""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssie
ssie.load()
from ssie_amd import build, hostlib as H

def main():
    cin, cout, k, hw, N = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (64, 64, 3, 128, 32)))
    out = os.environ.get("LIB", "/tmp/libssie_stamp_v2.so")
    if not os.path.exists(out):
        subprocess.check_call([build.hipcc(), *build.FLAGS, "-DSSIE_STAMP", "-shared", "-o", out, *build.sources()])
    L = C.CDLL(out)
    L.ssie_op_workspace_bytes.restype = C.c_size_t
    dev = "cuda"
    cs = (cin + 3) // 4 * 4
    data = os.environ.get("DATA", "randn")
    x = torch.randn(N, hw, hw, cs, device=dev); g = torch.randn(N, hw, hw, cout, device=dev)
    if data == "zeros":
        x.zero_(); g.zero_()
    elif data == "relu":      # what the plan feeds: post-ReLU activations (half zeros), small smooth gradients
        x = torch.relu(x); g = g * 1e-3 * (torch.rand_like(g) > 0.5)
    dw = torch.zeros(cout, cin, k, k, device=dev); db = torch.zeros(cout, device=dev)
    ws = torch.zeros(L.ssie_op_workspace_bytes(max(cin, 64), max(cout, 64), k) // 4 + 1, device=dev)
    s = H.src_of(x, cin)
    def run():
        return L.ssie_conv2d_wgrad(C.byref(s), N, hw, hw, H.ptr(g), cout, 0, cout, k, 1, cin, 0, H.ptr(dw), H.ptr(db), 0,
                                   H.ptr(ws), C.c_size_t(ws.numel() * 4), None)
    for _ in range(10):
        assert run() == 0
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        run()
    t1.record(); torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 20 * 1e3
    fl = 2.0 * N * hw * hw * cout * cin * k * k
    print(f"[{data}] wgrad {cin}->{cout} k{k} {hw}x{hw} N{N}: avg call (wgrad + reduce) {us:.1f} us -> {fl/us/1e6:.1f} TF")
    if not hasattr(L, "ssie_debug_set_stamp_buffer") or os.environ.get("NOSTAMP"):
        return
    nwg = 512
    stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    assert L.ssie_debug_set_stamp_buffer(C.c_void_p(stamps.data_ptr())) == 0
    assert run() == 0; torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(nwg, 8).astype(np.float64)
    st = st[st[:, 6] > 0]
    tot = st[:, 3] - st[:, 0]; nt = st[:, 6]
    print(f"  WGs {len(st)}  tiles/WG {nt.mean():.2f}  total cycles mean {tot.mean():.0f} max {tot.max():.0f}")
    print(f"  shader clock inside the kernel (s_memtime / s_memrealtime x 100 MHz): median {np.median(tot / st[:, 4]) * 100:.0f} MHz")
    names = {1: "staging (loads + LDS writes)", 2: "barriers", 7: "MFMA loop", 5: "slab write (+K-split reduce)"}
    for kx, nm in names.items():
        print(f"  {nm:34s} {np.mean(st[:, kx] / nt):9.0f} cycles/tile  {100 * st[:, kx].sum() / tot.sum():5.1f} %")

if __name__ == "__main__":
    main()
