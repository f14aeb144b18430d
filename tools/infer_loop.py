"""Dev tool (GPU): a few enhance-only forwards of one 1x31xHWxHW cube, for rocprofv3 runs.  usage: infer_loop.py [hw] [bf16|f32] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ssie
ssie.load()
from ssie_amd import model
import bench

hw = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bf16 = (sys.argv[2] if len(sys.argv) > 2 else "bf16") == "bf16"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.manual_seed(41)
net = model.LowLightEnhance(input_channels=31, lr=1e-3, **bench.JYU).to("cuda")
net.bf16_inference = bf16
x = bench.synth(1, 31, hw, 41, "cuda")
with torch.no_grad():
    for _ in range(reps):
        net._forward_views(x)
torch.cuda.synchronize()
print("done")
