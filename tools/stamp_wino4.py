"""Diagnostic (dev tool): build a -DSSIE_STAMP copy of the library into /tmp, run one stride-1 3x3 conv through the Winograd
kernel and print where one wave of each workgroup spends its cycles.  WINO4=1: the F(4x4,3x3) kernel (conv_wino4.hip; SSIE_STAMP_FLAGS="-DSSIE_STAMP_WAVE=4" stamps wave 4)."""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssie
ssie.load()
from ssie_amd import build, hostlib as H

def main():
    cin, cout, k, hw, N = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (64, 64, 3, 128, 32)))
    extra = os.environ.get("SSIE_STAMP_FLAGS", "").split()          # e.g. -DSSIE_X_NOSTORE: ablation builds
    out = "/tmp/libssie_stamp_wino" + "".join(f.replace("-D", "_") for f in extra) + ".so"
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(s) for s in build.sources()):
        subprocess.check_call([build.hipcc(), *build.FLAGS, "-DSSIE_STAMP", *extra, "-shared", "-o", out, *build.sources()])
    L = C.CDLL(out)
    L.ssie_op_workspace_bytes.restype = C.c_size_t
    L.ssie_debug_set_wino_min_tiles(1)
    w4 = os.environ.get("WINO4", "0") == "1"
    L.ssie_debug_set_wino4_min_tiles(1 if w4 else 1 << 30)
    dev = "cuda"
    cs = (cin + 3) // 4 * 4
    x = torch.randn(N, hw, hw, cs, device=dev); x[..., cin:] = 0; w = torch.randn(cout, cin, k, k, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    o = torch.zeros(N, hw, hw, cout, device=dev)
    ws = torch.zeros(L.ssie_op_workspace_bytes(cin, cout, k) // 4 + 1, device=dev)
    nwg = 256
    stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    arr = (H.SrcT * 1)(H.src_of(x, cs))
    def run():
        return L.ssie_conv2d_fwd(arr, 1, N, hw, hw, H.ptr(w), cin, H.ptr(b), cout, k, 1, 1, None, None, H.ptr(o), cout, 0,
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
    assert (L.ssie_debug_set_stamp_buffer_wino4 if w4 else L.ssie_debug_set_stamp_buffer_wino)(C.c_void_p(stamps.data_ptr())) == 0
    assert run() == 0; torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(nwg, 8).astype(np.float64)
    s = s[s[:, 6] > 0]
    tot = s[:, 3] - s[:, 0]
    nt = s[:, 6]
    print(f"  WGs {len(s)}  tiles/WG {nt.mean():.2f} (min {nt.min():.0f} max {nt.max():.0f})  total cycles mean {tot.mean():.0f} max {tot.max():.0f}")
    names = {1: "wait for the step's DMA (vmcnt)", 2: "wait at the barrier", 7: "step bodies (LDS reads, transforms, MFMAs" + ("" if w4 else ", DMA issue") + ")", 5: "epilogue + tile bookkeeping"}
    if w4:
        names[4] = "DMA issue"
    for kx, nm in names.items():
        print(f"  {nm:34s} {np.mean(s[:, kx] / nt):9.0f} cycles/tile  {100 * s[:, kx].sum() / tot.sum():5.1f} %")
    acc = sum(s[:, kx] for kx in names)
    print(f"  unaccounted {100 * (1 - acc.sum() / tot.sum()):.1f} %")

if __name__ == "__main__":
    main()
