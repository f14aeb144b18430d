"""Micro-benchmark of the MFMA conv operators (dev tool, GPU only): TFLOP/s per layer shape."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssie
ssie.load()
from ssie_amd import hostlib as H

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    only = sys.argv[2] if len(sys.argv) > 2 else None
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    dev = "cuda"
    L = H.lib()
    if os.environ.get('SSIE_V2') is not None:
        L.ssie_debug_set_fprop_v2(int(os.environ['SSIE_V2']))
    if os.environ.get('SSIE_WGS'):
        L.ssie_debug_set_fprop_wgs_per_cu(int(os.environ['SSIE_WGS']))
    for (name, cin, cout, k, stride, hw) in [("conv1 64->64 3x3", 64, 64, 3, 1, 128), ("shallow 32->64 9x9", 32, 64, 9, 1, 128),
                                              ("conv0 32->32 3x3", 32, 32, 3, 1, 128), ("conv3 128->128 @64", 128, 128, 3, 1, 64),
                                              ("conv2 64->128 s2", 64, 128, 3, 2, 128), ("conv5 128->64", 128, 64, 3, 1, 128),
                                              ("fusion 192->64 1x1", 192, 64, 1, 1, 128)]:
        if only and not name.startswith(only):
            continue
        x = torch.randn(N, hw, hw, cin, device=dev)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        b = torch.randn(cout, device=dev)
        ho = hw // stride
        g = torch.randn(N, ho, ho, cout, device=dev)
        out = torch.zeros(N, ho, ho, cout, device=dev)
        gx = torch.zeros(N, hw, hw, cin, device=dev)
        dw = torch.zeros_like(w); db = torch.zeros_like(b)
        ws = H.workspace(max(cin, 64), max(cout, 64), k, dev)
        wsb = C.c_size_t(ws.numel() * 4)
        src = H.src_of(x, cin)
        arr = (H.SrcT * 1)(src)
        flops = 2.0 * N * ho * ho * cout * cin * k * k
        f = lambda: H.check(L.ssie_conv2d_fwd(arr, 1, N, hw, hw, H.ptr(w), cin, H.ptr(b), cout, k, stride, 1, None, None, H.ptr(out), cout, 0, H.ptr(ws), wsb, H.stream_ptr()), "fwd")
        d = lambda: H.check(L.ssie_conv2d_dgrad(H.ptr(g), cout, 0, N, ho, ho, cout, H.ptr(w), cin, 0, cin, k, stride, H.ptr(gx), hw, hw, cin, 0, None, 0, 0, H.ptr(ws), wsb, H.stream_ptr()), "dgrad")
        wg = lambda: H.check(L.ssie_conv2d_wgrad(C.byref(src), N, hw, hw, H.ptr(g), cout, 0, cout, k, stride, cin, 0, H.ptr(dw), H.ptr(db), 0, H.ptr(ws), wsb, H.stream_ptr()), "wgrad")
        tf, td, tw = timeit(f, iters), timeit(d, iters), timeit(wg, iters)
        print(f"{name:22s} N={N} fprop {tf*1e6:8.1f} us {flops/tf/1e12:6.1f} TF | dgrad {td*1e6:8.1f} us {flops/td/1e12:6.1f} TF | wgrad {tw*1e6:8.1f} us {flops/tw/1e12:6.1f} TF", flush=True)

if __name__ == "__main__":
    main()
