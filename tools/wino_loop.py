"""Dev tool (GPU): a few forward launches of one stride-1 3x3 conv through the Winograd kernel (or the direct one: WINO=0), for
rocprofv3 runs.  usage: wino_loop.py [cin cout hw N reps]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssie
ssie.load()
from ssie_amd import hostlib as H

cin, cout, hw, N, reps = (int(a) for a in (sys.argv[1:6] if len(sys.argv) > 5 else (64, 64, 128, 32, 5)))
L = H.lib()
L.ssie_debug_set_wino(int(os.environ.get("WINO", "1")))
L.ssie_debug_set_wino_min_tiles(1)
L.ssie_debug_set_wino4_min_tiles(1 if os.environ.get("WINO4", "0") == "1" else 1 << 30)     # WINO4=1: the F(4x4,3x3) kernel
dev = "cuda"
x = torch.randn(N, hw, hw, cin, device=dev); w = torch.randn(cout, cin, 3, 3, device=dev) * 0.05; b = torch.randn(cout, device=dev)
o = torch.zeros(N, hw, hw, cout, device=dev)
ws = H.workspace(max(cin, 64), max(cout, 64), 3, dev)
arr = (H.SrcT * 1)(H.src_of(x, cin))
def run():
    H.check(L.ssie_conv2d_fwd(arr, 1, N, hw, hw, H.ptr(w), cin, H.ptr(b), cout, 3, 1, 1, None, None, H.ptr(o), cout, 0, H.ptr(ws),
                              C.c_size_t(ws.numel() * 4), H.stream_ptr()), "fwd")
for _ in range(reps):
    run()
torch.cuda.synchronize()
if os.environ.get("TIME"):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('SSIE_HIP_LIB', 'default')}: conv {cin}->{cout} {hw}x{hw} N{N}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (pack + conv)")
print("done")
