"""Dev tool (GPU): F(4x4,3x3) against F(2x2,3x3) and the direct kernel on one stride-1 3x3 layer: time per call (pack + conv) and the
error of each against fp64 F.conv2d.  usage: wino4_ab.py [cin cout hw N]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import ssie
ssie.load()
from ssie_amd import hostlib as H

shapes = [tuple(int(a) for a in sys.argv[1:5])] if len(sys.argv) > 4 else [(64, 64, 128, 32), (128, 128, 64, 32), (128, 64, 128, 32), (64, 32, 128, 32), (32, 64, 128, 32)]
L = H.lib()
dev = "cuda"
for cin, cout, hw, N in shapes:
    torch.manual_seed(0)
    x = torch.relu(torch.randn(N, hw, hw, cin, device=dev)); w = (torch.rand(cout, cin, 3, 3, device=dev) * 2 - 1) / (cin * 9) ** 0.5; b = torch.randn(cout, device=dev)
    o = torch.zeros(N, hw, hw, cout, device=dev)
    ws = H.workspace(max(cin, 64), max(cout, 64), 3, dev)
    arr = (H.SrcT * 1)(H.src_of(x, cin))
    ref = F.conv2d(x[:2].permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu(), padding=1).permute(0, 2, 3, 1)
    def run():
        H.check(L.ssie_conv2d_fwd(arr, 1, N, hw, hw, H.ptr(w), cin, H.ptr(b), cout, 3, 1, 0, None, None, H.ptr(o), cout, 0, H.ptr(ws),
                                  C.c_size_t(ws.numel() * 4), H.stream_ptr()), "fwd")
    for name, w4, w2 in (("F(4x4,3x3)", 1, 1), ("F(2x2,3x3)", 1 << 30, 1), ("direct", 1 << 30, 1 << 30)):
        L.ssie_debug_set_wino4_min_tiles(w4); L.ssie_debug_set_wino_min_tiles(w2)
        o.zero_()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record(); torch.cuda.synchronize()
        err = (o[:2].double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"conv {cin}->{cout} {hw}x{hw} N{N} {name:11s}: {us:7.1f} us per call (pack + conv)  {2.0 * N * hw * hw * cin * cout * 9 / us / 1e6:6.1f} TFLOP/s direct-equivalent  max err / max|ref| {err:.2e}", flush=True)
    L.ssie_debug_set_wino4_min_tiles(-1); L.ssie_debug_set_wino_min_tiles(-1)               # the library's default
