#!/usr/bin/env python3
"""Benchmark of the hot path: full self-supervised train step (forward, six losses, hand-derived backward,
fused Adam) on synthetic 128x128x31 hyperspectral patches, fp32, batch 32 per GPU (BASELINE.json configs[1];
configs[3] = the same per GPU over N GPUs with one RCCL all-reduce of the flat gradient buffer).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Rank 0 prints ONE JSON line.  `value` = patches/s over all ranks, inputs resident in HBM, max-over-ranks time
around exactly K steps.  `roofline` = the dominant kernel class by device time, measured with HIP events after every
launch in a profiled pass of the same step that follows the timed region (events inside the timed region would
perturb `value`); `cpu_baseline` = the CPU oracle (a PyTorch port of the reference, which cannot travel to the GPU
box) timed on this box's host cores on a bounded sample (batch 2 = the reference config's batch).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_PATCH = {(31, 128): 87.6, (31, 64): 21.9, (64, 128): 122.8, (256, 128): 327.6}   # SURVEY §8(d), full train step
PEAK_F32_TFLOPS = 157.3            # MI355X fp32 matrix (= vector) peak, MI355X_MICROARCH.md
JYU = dict(c_loss_reconstruction=10, c_loss_r_fidelity=1, c_loss_i_smooth_low=1, c_loss_i_smooth_delta=2000,
           c_loss_fourier=20, c_loss_spectral_cons=1, alpha_i_smooth_low=1, alpha_i_smooth_delta=10)   # config_outdoor_jyu.yml:24-31


def synth(n, bands, hw, seed, device):
    """low-light cubes in [0, 0.3]: smooth illumination x band-correlated reflectance + noise; channels_last like the
    reference loader (model.py:301,312)"""
    import torch
    g = torch.Generator(device="cpu").manual_seed(seed)
    hh = torch.arange(hw, dtype=torch.float32).view(1, 1, hw, 1)
    ww = torch.arange(hw, dtype=torch.float32).view(1, 1, 1, hw)
    cc = torch.arange(bands, dtype=torch.float32).view(1, bands, 1, 1)
    ph = torch.rand(n, 1, 1, 1, generator=g) * 6.28
    illum = 0.55 + 0.35 * torch.sin(0.11 * hh + ph) * torch.cos(0.07 * ww - 0.5 * ph)
    refl = 0.5 + 0.3 * torch.sin(0.45 * cc + 0.05 * hh - 0.04 * ww + ph) + 0.15 * torch.cos(0.9 * cc - 0.13 * ww + 0.21 * hh)
    x = (illum * refl + 0.01 * (torch.rand(n, bands, hw, hw, generator=g) - 0.5)).clamp_(0, 1) * 0.3
    return x.contiguous(memory_format=torch.channels_last).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="patches per GPU")
    ap.add_argument("--bands", type=int, default=31)
    ap.add_argument("--hw", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import torch
    import ssie
    ssie.load()
    from ssie_amd import hostlib, model

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with WORLD_SIZE={args.gpus} (got {world})")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    assert hostlib.lib().ssie_device_ok() == 1, "bench.py needs a gfx950 (MI355X) device"

    torch.manual_seed(41)                                   # reference default seed_value (main.py:19); same init on every rank
    net = model.LowLightEnhance(input_channels=args.bands, lr=1e-3, **JYU).to(dev)
    x = synth(args.batch, args.bands, args.hw, 41 + rank, dev)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        net.train_step(x, world)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        net.train_step(x, world)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    losses = net._plan_for(x).loss_scalars().cpu().tolist()
    value = world * args.batch * args.steps / dt

    out = {
        "metric": "HSI patches/sec (train step), 128x128x31", "value": round(value, 2), "unit": "patches/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"full self-supervised train step (fwd + 6 losses + bwd + Adam), batch {args.batch}/GPU of "
                               f"{args.hw}x{args.hw}x{args.bands} patches, fp32, JYU loss coefficients",
                   "global_batch": world * args.batch, "parallelism": f"dp{world}" if world > 1 else "single",
                   "weights": "random init (PyTorch default), seed 41"},
        "final_total_loss": losses[0],
    }
    gf = GFLOP_PER_PATCH.get((args.bands, args.hw))
    if gf:
        out["step_tflops_per_gpu"] = round(value / world * gf / 1e3, 2)
        out["step_frac_of_f32_peak"] = round(value / world * gf / 1e3 / PEAK_F32_TFLOPS, 4)

    if rank == 0 and not args.no_roofline:
        plan = net._plan_for(x)
        agg = None
        reps = 3
        for _ in range(reps):
            pr = plan.profile_step(x)
            agg = pr if agg is None else {k: (agg[k][0] + v[0], agg[k][1] + v[1], agg[k][2] + v[2]) for k, v in pr.items()}
        dom = max(agg, key=lambda k: agg[k][0])
        ms, fl, cnt = agg[dom]
        ach = fl / (ms * 1e-3) / 1e12
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process; the per-launch figure
        # comes from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (profiles/)
        traffic = None
        try:
            import glob
            tj = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))[-1]))["kernels"]
            cand = [v["hbm_bytes_per_launch"] * v["launches_sampled"] for k, v in tj.items() if k.startswith("conv_fprop")]
            nl = sum(v["launches_sampled"] for k, v in tj.items() if k.startswith("conv_fprop"))
            traffic = int(sum(cand) / nl) if nl else None
        except Exception:
            pass
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_F32_TFLOPS, 4), "traffic": traffic,
                           "launches_per_step": cnt // reps, "avg_launch_ms": round(ms / cnt, 4),
                           "algorithmic_gflop_per_step": round(fl / reps / 1e9, 1)}
        out["kernel_classes"] = {k: {"ms_per_step": round(v[0] / reps, 3), "tflops": round(v[1] / (v[0] * 1e-3) / 1e12, 2) if v[1] > 0 and v[0] > 0 else None,
                                     "launches": v[2] // reps} for k, v in agg.items()}
    if world > 1:
        sync_all()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # TEST-INFRASTRUCTURE import: the CPU oracle, used here only as the timed baseline and the PSNR checker
        from collections import OrderedDict
        from oracle import ssie_oracle as O
        # the box gives one GPU job a 16-core CPU share; more threads than that only oversubscribe
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        ncpu = max(1, min(ncpu, int(os.environ.get("SSIE_CPU_THREADS", "16"))))
        torch.set_num_threads(ncpu)
        P = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
        xb = x[:2].cpu()
        co = dict(c_rec=10.0, c_rf=1.0, c_il=1.0, c_id=2000.0, c_f=20.0, c_sp=1.0, alpha_low=1.0, alpha_delta=10.0)
        with torch.no_grad():
            So = O.enhance_forward(P, xb)[3]
            Sh = net(x[:2].contiguous(memory_format=torch.channels_last))[3].cpu()
        out["parity"] = {"psnr_enhanced_vs_cpu_oracle_db": round(O.psnr(Sh, So), 1),
                         "max_abs_S": float((Sh - So).abs().max())}
        st = O.AdamState(P)
        O.train_step(P, xb, co, st)                          # warm-up
        t0 = time.perf_counter(); n = 0
        while n < 2 or (time.perf_counter() - t0 < 12.0 and n < 400):
            P, *_ = O.train_step(P, xb, co, st); n += 1
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(2 * n / cdt, 3), "unit": "patches/s", "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{n} full train steps of batch 2 (reference config batch) 128x128x{args.bands} on the CPU oracle "
                                         f"(plain PyTorch restatement of the reference), {cdt:.1f} s"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
