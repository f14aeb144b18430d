"""Dev tool (GPU): BASELINE.json configs[4] geometry - enhance-only forward (model.py:229-234) on one 1 x 31 x 1024 x 1024 cube.
Prints images/s and the per-launch device time of the forward op list."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssie
ssie.load()
from ssie_amd import hostlib as H, model
import bench

def main():
    hw = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    bf16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
    torch.manual_seed(41)
    net = model.LowLightEnhance(input_channels=31, lr=1e-3, **bench.JYU).to("cuda")
    x = bench.synth(1, 31, hw, 41, "cuda")
    ref = None
    if bf16:
        with torch.no_grad():
            ref = [t.clone() for t in net(x)]
        net.bf16_inference = True
    with torch.no_grad():
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 10
        for _ in range(n):
            net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    if ref is not None:
        with torch.no_grad():
            out = net(x)
        for nm, a, b in zip(("R_low", "I_low", "I_delta", "S"), out, ref):
            mse = ((a.double() - b.double()) ** 2).mean().item()
            print(f"  bf16 vs fp32 {nm}: max abs {float((a - b).abs().max()):.3e}  PSNR {10 * __import__('math').log10(1.0 / max(mse, 1e-30)):.1f} dB")
    print(f"enhance-only forward 1x31x{hw}x{hw} {'bf16' if bf16 else 'fp32'}: {dt*1e3:.2f} ms/image = {1/dt:.2f} images/s "
          f"({2*579.0*(hw/1024)**2/dt/1e3:.1f} TFLOP/s at 579 GMAC per 1024^2 image)")
    plan = net._plan_for(x)
    L = H._proto()
    cap = 1024
    ms = (C.c_double * cap)(); fl = (C.c_double * cap)(); kinds = (C.c_int * cap)(); tags = C.create_string_buffer(1 << 16)
    L.ssie_plan_profile_list.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.c_void_p, C.c_int, C.POINTER(C.c_double),
                                         C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int]
    nops = L.ssie_plan_profile_list(plan.h, x.data_ptr(), plan._strides(x), torch.cuda.current_stream().cuda_stream, int(bf16), ms, fl, kinds, cap, tags, 1 << 16)
    assert nops > 0, nops
    names = tags.value.decode().split("\n")
    rows = [(ms[i], fl[i], H.Plan.KINDS[kinds[i]], names[i]) for i in range(nops)]
    tot = sum(r[0] for r in rows)
    print(f"forward op list: {nops} launches, {tot:.2f} ms")
    for t, f, k, nm in sorted(rows, key=lambda r: -r[0])[:int(os.environ.get("TOP", "18"))]:
        print(f"{t:8.3f} ms {100*t/tot:5.1f}%  {f/(t*1e-3)/1e12 if f else 0:6.1f} TF  {k:22s} {nm}")

if __name__ == "__main__":
    main()
