"""Diagnostic (dev tool): -DSSIE_STAMP build of the library; runs the bf16 enhance-only forward of one 1x31xHWxHW cube and prints
where wave 0 of each workgroup of ONE wide-kernel launch (index among the wide bf16 launches of the forward) spends its cycles.
usage: python tools/stamp_bf16.py [hw] [launch_index ...]"""
import ctypes as C, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
extra = os.environ.get("SSIE_STAMP_FLAGS", "").split()
out = "/tmp/libssie_stamp_bf16" + "".join(f.replace("-D", "_") for f in extra) + ".so"
import ssie
ssie.load()
from ssie_amd import build
if not os.path.exists(out):
    subprocess.check_call([build.hipcc(), *build.FLAGS, "-DSSIE_STAMP", *extra, "-shared", "-o", out, *build.sources()])
os.environ["SSIE_DEBUG"] = "1"; os.environ["SSIE_HIP_LIB"] = out
import numpy as np, torch
from ssie_amd import hostlib as H, model
import bench

def main():
    hw = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    idxs = [int(a) for a in sys.argv[2:]] or [0]
    torch.manual_seed(41)
    net = model.LowLightEnhance(input_channels=31, lr=1e-3, **bench.JYU).to("cuda")
    net.bf16_inference = True
    x = bench.synth(1, 31, hw, 41, "cuda")
    L = H.lib()
    with torch.no_grad():
        for _ in range(5):
            net._forward_views(x)
        torch.cuda.synchronize()
        for idx in idxs:
            stamps = torch.zeros(256 * 12, dtype=torch.int64, device="cuda")
            L.ssie_debug_set_stamp_buffer_h(C.c_void_p(stamps.data_ptr()), idx)
            net._forward_views(x); torch.cuda.synchronize()
            L.ssie_debug_set_stamp_buffer_h(None, -1)
            s = stamps.cpu().numpy().reshape(256, 12).astype(np.float64)
            s = s[s[:, 6] > 0]
            if not len(s):
                print(f"launch {idx}: no stamps"); continue
            tot = s[:, 3] - s[:, 0]; nt = s[:, 6]
            print(f"wide bf16 launch {idx}: WGs {len(s)}  tiles/WG {nt.mean():.2f} (min {nt.min():.0f} max {nt.max():.0f})  total cycles mean {tot.mean():.0f} max {tot.max():.0f}"
                  f"  start spread {s[:, 0].max() - s[:, 0].min():.0f}  end spread {s[:, 3].max() - s[:, 3].min():.0f}  first start -> last end {s[:, 3].max() - s[:, 0].min():.0f}")
            names = {1: "barrier wait, first step of tile", 2: "barrier wait, other steps", 4: "DMA issue", 7: "MFMA tap loops", 8: "end of MFMA loop -> epilogue", 9: "epilogue proper",
                     5: "after epilogue: bookkeeping, next tile's first step head"}
            prod = {4: "producer: DMA issue", 10: "producer: wait for landing (vmcnt)", 11: "producer: barrier wait"}
            if s[:, 10].sum() > 0:
                for kx, nm in prod.items():
                    print(f"  {nm:34s} {np.mean(s[:, kx] / nt):9.0f} cycles/tile")
                names.pop(4)
            for kx, nm in names.items():
                print(f"  {nm:34s} {np.mean(s[:, kx] / nt):9.0f} cycles/tile  {100 * s[:, kx].sum() / tot.sum():5.1f} %")
            acc = sum(s[:, kx] for kx in names)
            print(f"  unaccounted {100 * (1 - acc.sum() / tot.sum()):.1f} %")

if __name__ == "__main__":
    main()
