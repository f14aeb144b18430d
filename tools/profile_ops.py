"""Dev tool (GPU): per-launch device time of one train step, sorted; shows which layers sit far from the MFMA roof."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssie
ssie.load()
from ssie_amd import hostlib as H, model
import bench

def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    bands = int(sys.argv[2]) if len(sys.argv) > 2 else 31           # usage: profile_ops.py [N [bands]]
    torch.manual_seed(41)
    net = model.LowLightEnhance(input_channels=bands, lr=1e-3, **bench.JYU).to("cuda")
    x = bench.synth(N, bands, 128, 41, "cuda")
    for _ in range(3):
        net.train_step(x)
    plan = net._plan_for(x)
    cap = 1024
    ms = (C.c_double * cap)(); fl = (C.c_double * cap)(); kinds = (C.c_int * cap)(); tags = C.create_string_buffer(1 << 16)
    L = H._proto()
    L.ssie_plan_profile_ops.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_long), C.c_void_p, C.POINTER(C.c_double),
                                        C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_int]
    acc = None
    reps = 3
    for _ in range(reps):
        n = L.ssie_plan_profile_ops(plan.h, x.data_ptr(), plan._strides(x), torch.cuda.current_stream().cuda_stream, ms, fl, kinds, cap, tags, 1 << 16)
        assert n > 0, n
        cur = [ms[i] for i in range(n)]
        acc = cur if acc is None else [a + b for a, b in zip(acc, cur)]
    names = tags.value.decode().split("\n")
    rows = [(acc[i] / reps, fl[i], kinds[i], names[i]) for i in range(n)]
    tot = sum(r[0] for r in rows)
    print(f"{n} launches, {tot:.2f} ms per step")
    if os.environ.get("RAW"):
        for i, (t, f, k, nm) in enumerate(rows):
            if os.environ["RAW"] in nm:
                print(f"  #{i:3d} {t:7.3f} ms {f / (t * 1e-3) / 1e12 if f > 0 else 0:6.1f} TF  {nm}")
    agg = {}
    for t, f, k, nm in rows:
        key = (H.Plan.KINDS[k], nm)
        a = agg.setdefault(key, [0.0, 0.0, 0]); a[0] += t; a[1] += f; a[2] += 1
    for (kind, nm), (t, f, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
        tf = f / (t * 1e-3) / 1e12 if f > 0 else 0
        print(f"{t:8.3f} ms {100*t/tot:5.1f}%  x{c:<2d} {tf:6.1f} TF  {kind:22s} {nm}")

if __name__ == "__main__":
    main()
