"""Diagnostic (dev tool): build a -DSSIE_STAMP copy of the library into /tmp, run one stride-1 conv through the v2
(DMA, one workgroup per CU) fprop kernel and print where wave 0 of each workgroup spends its cycles."""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssie
ssie.load()
from ssie_amd import build, hostlib as H

def main():
    cin, cout, k, hw, N = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (64, 64, 3, 128, 32)))
    stride = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    out = "/tmp/libssie_stamp_v2.so"
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in build.sources()):
        subprocess.check_call([build.hipcc(), *build.FLAGS, "-DSSIE_STAMP", "-shared", "-o", out, *build.sources()])
    L = C.CDLL(out)
    L.ssie_op_workspace_bytes.restype = C.c_size_t
    dev = "cuda"
    cs = (cin + 3) // 4 * 4
    x = torch.randn(N, hw, hw, cs, device=dev); x[..., cin:] = 0; w = torch.randn(cout, cin, k, k, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    o = torch.zeros(N, hw // stride, hw // stride, cout, device=dev)
    ws = torch.zeros(L.ssie_op_workspace_bytes(cin, cout, k) // 4 + 1, device=dev)
    nwg = 256
    stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    arr = (H.SrcT * 1)(H.src_of(x, cs))
    def run():
        return L.ssie_conv2d_fwd(arr, 1, N, hw, hw, H.ptr(w), cin, H.ptr(b), cout, k, stride, 1, None, None, H.ptr(o), cout, 0,
                                 H.ptr(ws), C.c_size_t(ws.numel() * 4), None)
    rc = run(); assert rc == 0, f"ssie_conv2d_fwd rc={rc}"
    for _ in range(20):
        assert run() == 0
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        run()
    t1.record(); torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 20 * 1e3
    fl = 2.0 * N * hw * hw * cout * cin * k * k
    print(f"conv {cin}->{cout} k{k} {hw}x{hw} N{N}: avg call (pack + conv) {us:.1f} us -> {fl/us/1e6:.1f} TF")
    assert L.ssie_debug_set_stamp_buffer_v2(C.c_void_p(stamps.data_ptr())) == 0
    assert run() == 0; torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(nwg, 8).astype(np.float64)
    s = s[s[:, 6] > 0]
    tot = s[:, 3] - s[:, 0]
    nt = s[:, 6]
    print(f"  WGs {len(s)}  tiles/WG {nt.mean():.2f} (min {nt.min():.0f} max {nt.max():.0f})  total cycles mean {tot.mean():.0f} max {tot.max():.0f}")
    names = {1: "barrier wait, first step of tile", 2: "barrier wait, other steps", 4: "DMA issue", 7: "MFMA tap loops", 5: "epilogue + tile bookkeeping"}
    for kx, nm in names.items():
        print(f"  {nm:34s} {np.mean(s[:, kx] / nt):9.0f} cycles/tile  {100 * s[:, kx].sum() / tot.sum():5.1f} %")
    acc = sum(s[:, kx] for kx in names)
    print(f"  unaccounted {100 * (1 - acc.sum() / tot.sum()):.1f} %")

if __name__ == "__main__":
    main()
