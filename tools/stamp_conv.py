"""Diagnostic (dev tool): build a -DSSIE_STAMP copy of the library into /tmp, run one conv fprop and print the
per-workgroup phase durations (prologue / main loop / epilogue) and concurrency from s_memtime stamps."""
import ctypes as C, os, subprocess, sys, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ssie
ssie.load()
from ssie_amd import build, hostlib as H

def main():
    cin, cout, k, stride, hw, N = (int(a) for a in (sys.argv[1:7] if len(sys.argv) > 6 else (64, 64, 3, 1, 128, 32)))
    abl = [a for a in os.environ.get("ABL", "").split(",") if a]
    out = "/tmp/libssie_stamp_%s.so" % "_".join(abl)
    srcs = build.sources()
    subprocess.check_call([build.hipcc(), *build.FLAGS, "-DSSIE_STAMP", *["-DABL_" + a for a in abl], "-shared", "-o", out, *srcs])
    print("ablations:", abl)
    L = C.CDLL(out)
    L.ssie_op_workspace_bytes.restype = C.c_size_t
    dev = "cuda"
    x = torch.randn(N, hw, hw, cin, device=dev); w = torch.randn(cout, cin, k, k, device=dev) * 0.05; b = torch.randn(cout, device=dev)
    ho = hw // stride
    o = torch.zeros(N, ho, ho, cout, device=dev)
    ws = torch.zeros(L.ssie_op_workspace_bytes(cin, cout, k) // 4 + 1, device=dev)
    nwg = min(N * ((ho + 7) // 8) * ((ho + 15) // 16) * max(1, (cout + 63) // 64), 768)
    stamps = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
    arr = (H.SrcT * 1)(H.src_of(x, cin))
    def run():
        return L.ssie_conv2d_fwd(arr, 1, N, hw, hw, H.ptr(w), cin, H.ptr(b), cout, k, stride, 1, None, None, H.ptr(o), cout, 0,
                                 H.ptr(ws), C.c_size_t(ws.numel() * 4), None)
    for _ in range(30):
        assert run() == 0
    torch.cuda.synchronize()
    if os.environ.get("SSIE_WGS"):
        L.ssie_debug_set_fprop_wgs_per_cu(int(os.environ["SSIE_WGS"]))
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        run()
    t1.record(); torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 20 * 1e3
    print(f"  avg call (pack + conv) {us:.1f} us  -> {2.0*N*ho*ho*cout*cin*k*k/us/1e6:.1f} TF")
    assert L.ssie_debug_set_stamp_buffer(C.c_void_p(stamps.data_ptr())) == 0
    assert run() == 0; torch.cuda.synchronize()
    s = stamps.cpu().numpy().reshape(nwg, 8)
    tot = s[:, 3] - s[:, 0]
    print(f"persistent WGs {nwg}; per-WG (cycles): total mean {tot.mean():.0f} max {tot.max()}  first-commit {np.mean(s[:,1]-s[:,0]):.0f}")
    print(f"  tiles/WG mean {s[:,6].mean():.2f}  main-loop per tile {np.mean(s[:,2]/np.maximum(s[:,6],1)):.0f}  epilogue+boundary per tile {np.mean(s[:,5]/np.maximum(s[:,6],1)):.0f}")
    clk = tot / np.maximum(s[:, 7], 1) * 100.0
    print(f"  in-kernel shader clock (d s_memtime / d s_memrealtime x 100 MHz): median {np.median(clk):.0f} MHz  p10 {np.percentile(clk,10):.0f} p90 {np.percentile(clk,90):.0f}")
    for q in (5, 6):
        sel = s[:, 6] == q
        if sel.any():
            print(f"  WGs with {q} tiles: {sel.sum()}  total mean {tot[sel].mean():.0f}  loop/tile {np.mean(s[sel,2]/q):.0f}  epi/tile {np.mean(s[sel,5]/q):.0f}")
    for x in range(8):
        sel = s[:, 4] == x
        if sel.any():
            span = s[sel, 3].max() - s[sel, 0].min()
            print(f"  XCD {x}: {sel.sum()} WGs, span {span} cycles")

if __name__ == "__main__":
    main()
