"""Training / evaluation / test harness around the HIP hot path.

Own counterpart of the reference's host orchestration — `train_model` (model.py:236-341), `evaluate_model` (:343-404),
`test_model` (:406-443), `.mat` IO + normalisation (utils.py:36-57, 171-178) and `calc_metrics` (metrics.py:101-141) —
redesigned for the GPU path:
  * cubes are normalised once on the host exactly like `load_hsi` (including its double normalisation:
    `(x-min)/(max-min)`, clamp <0, then divide by the cube's own max, utils.py:45-47,57) and kept RESIDENT on the device;
  * every batch is cropped + augmented on the device (`ssie_assemble_batch`, bit-exact vs the reference's numpy path);
    only the crop coordinates come from the host RNG, drawn in the reference's order
    (`np.random.randint(0, h-patch)`, `(0, w-patch)`, `(0, 8)` per sample, model.py:306-308);
  * the train step is the fused `LowLightEnhance.train_step` (no autograd, one RCCL all-reduce when world > 1);
    the loss scalars are read back with a one-step lag through a pinned ring (`LaggedScalars`): the same per-batch print as
    the reference's `loss.item()`, without its per-step synchronisation;
  * the Linux-only defects of the reference are fixed: checkpoints are written to AND read from
    `.../Decomposition_<timestamp>` (main.py:87 looks for `decomposition_`), evaluation reads the key it wrote
    (`data`, model.py:375 vs :395), file names are split with os.path.basename (metrics.py:111 splits on '\\\\').
mlflow / torchinfo / matplotlib are optional and never required.
"""
from __future__ import annotations

import glob
import math
import os
import time

import numpy as np
import torch

from . import dp
from . import hostlib as H

LOSS_KEYS = H.LOSS_KEYS


# ---- .mat IO + normalisation (utils.py) -----------------------------------------------------------
def load_hsi(path, mat_key="data", normalization=None, max_val=None, min_val=None) -> np.ndarray:
    import scipy.io as sio
    x = np.array(sio.loadmat(path)[mat_key], dtype="float32")
    if normalization is None:
        return x.astype("float32")
    if normalization == "self":
        x = x / np.max(x)
    elif normalization == "global_normalization":
        if max_val is None:
            raise ValueError("max value is not provided for normalization")
        lo = 0.0 if min_val is None else min_val
        if lo > max_val:
            raise ValueError("min value cannot be larger than the max value for normalization")
        x = (x - lo) / (max_val - lo)
        x[x < 0] = 0.0
    elif normalization == "per_channel_normalization":
        mn = np.min(x, axis=(0, 1), keepdims=True); mx = np.max(x, axis=(0, 1), keepdims=True)
        x = (x - mn) / np.where(mx > mn, mx - mn, 1)
    elif normalization == "per_channel_standardization":
        mu = np.mean(x, axis=(0, 1), keepdims=True); sd = np.std(x, axis=(0, 1), keepdims=True)
        x = (x - mu) / np.where(sd > 0, sd, 1)
    else:
        raise NotImplementedError(normalization + " is not implemented")
    return x.astype("float32") / np.max(x)            # the reference's second normalisation (utils.py:57)


def save_hsi(path, data, postfix=None, key="data"):
    import scipy.io as sio
    base = path[:-4]
    if postfix is not None:
        base += postfix
    sio.savemat(base + ".mat", {key: data})


# ---- metrics (metrics.py:13-34, 101-141): own definitions; torchmetrics is absent => parity unpinned -----------------
def psnr(pred: torch.Tensor, target: torch.Tensor, data_range=None) -> torch.Tensor:
    """10 log10(data_range^2 / MSE) over all elements (torchmetrics peak_signal_noise_ratio with scalar data_range)."""
    if data_range is None:
        data_range = float(target.max() - target.min())
    mse = torch.mean((pred.double() - target.double()) ** 2)
    return 10.0 * torch.log10(torch.tensor(float(data_range) ** 2, dtype=torch.float64) / mse)


def _gauss(size=11, sigma=1.5):
    d = torch.arange(size, dtype=torch.float64) - (size - 1) / 2
    g = torch.exp(-(d ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def ssim(pred_hwc: torch.Tensor, target_hwc: torch.Tensor, data_range=None) -> torch.Tensor:
    """The reference feeds the (H,W,C) cube as a (1,H,W,C) NCHW image (metrics.py:16-19), i.e. H plays the channel role;
    Gaussian 11x11 sigma 1.5 window, k1=0.01, k2=0.03, reflect padding, mean over the valid map (torchmetrics defaults)."""
    import torch.nn.functional as F
    x = pred_hwc.double().unsqueeze(0); y = target_hwc.double().unsqueeze(0)       # (1, H, W, C): "channels" = H
    if data_range is None:
        data_range = float(max(x.max() - x.min(), y.max() - y.min()))
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    ch = x.shape[1]
    g = _gauss()
    win = (g[:, None] * g[None, :]).expand(ch, 1, 11, 11)
    pad = 5
    xp = F.pad(x, (pad, pad, pad, pad), mode="reflect"); yp = F.pad(y, (pad, pad, pad, pad), mode="reflect")
    cat = torch.cat([xp, yp, xp * xp, yp * yp, xp * yp])
    out = F.conv2d(cat, win, groups=ch)
    mx, my, sxx, syy, sxy = out[0:1], out[1:2], out[2:3], out[3:4], out[4:5]
    vx, vy, cxy = sxx - mx * mx, syy - my * my, sxy - mx * my
    m = ((2 * mx * my + c1) * (2 * cxy + c2)) / ((mx * mx + my * my + c1) * (vx + vy + c2))
    return m[..., pad:-pad, pad:-pad].mean()


def sam(pred_hwc: torch.Tensor, target_hwc: torch.Tensor) -> torch.Tensor:
    """mean spectral angle (radians) over pixels (torchmetrics spectral_angle_mapper, reduction='elementwise_mean')."""
    a = pred_hwc.double(); b = target_hwc.double()
    dot = (a * b).sum(-1)
    den = a.norm(dim=-1) * b.norm(dim=-1)
    return torch.acos(torch.clamp(dot / den, -1, 1)).mean()


def calc_metrics(im_glob, label_dir, data_max=None, key_pred="data", key_gt="data"):
    tot = np.zeros(3); n = 0
    for item in sorted(glob.glob(im_glob)):
        if not item.endswith(".mat"):
            continue
        name = os.path.basename(item)
        gt_path = os.path.join(label_dir, name)
        if not os.path.exists(gt_path):
            continue
        p = torch.from_numpy(load_hsi(item, key_pred)); t = torch.from_numpy(load_hsi(gt_path, key_gt))
        s = (float(psnr(p, t, data_max)), float(ssim(p, t, data_max)), float(sam(p, t)))
        print(f"\n===> {name} | PSNR : {s[0]:.4f}\n===> {name} | SSIM : {s[1]:.4f}\n===> {name} | SAM  : {s[2]:.4f}")
        tot += s; n += 1
    if n <= 0:
        raise ValueError("Number of files must be greater than 0")
    return tuple(tot / n)


# ---- training / evaluation / test -----------------------------------------------------------------
def _to_device_cubes(cubes, device):
    return [torch.from_numpy(np.ascontiguousarray(c)).to(device) for c in cubes]


def ctypes_sizeof_crop() -> int:
    import ctypes
    return ctypes.sizeof(H.CropT)


def draw_crops(n_cubes, shapes, batch_id, batch_size, patch, rng=np.random):
    """(cube index, x0, y0, mode) per sample, in the reference's RNG order (model.py:304-308)."""
    out = []
    for i in range(batch_size):
        idx = (batch_id * batch_size + i) % n_cubes
        h, w = shapes[idx][:2]
        if h <= patch or w <= patch:
            raise ValueError(f"cube {idx} ({h}x{w}) must be strictly larger than the patch ({patch}) (model.py:306-307)")
        x0 = int(rng.randint(0, h - patch)); y0 = int(rng.randint(0, w - patch)); mode = int(rng.randint(0, 8))
        out.append((idx, x0, y0, mode))
    return out


def resolve_dp_mode(dp_mode: str, batch_size: int, world: int) -> str:
    """"shard": ONE global batch of `batch_size` patches, drawn identically on every rank (same seed), each rank taking its
    contiguous slice - bit-comparable with a single-GPU run at the same batch_size, needs batch_size % world == 0.
    "per_rank" (SURVEY 8(e)): `batch_size` patches PER RANK from the rank's own RNG stream (seed + rank), global batch =
    batch_size x world - what lets the reference's batch 1-2 configurations use 8 GPUs.  "auto" = shard when it divides, else
    per_rank."""
    if dp_mode not in ("auto", "shard", "per_rank"):
        raise ValueError(f"dp_mode must be auto, shard or per_rank, got {dp_mode!r}")
    if world <= 1:
        return "shard"
    if dp_mode == "auto":
        return "shard" if batch_size % world == 0 else "per_rank"
    if dp_mode == "shard" and batch_size % world:
        raise ValueError("dp_mode=shard: batch_size must be divisible by the number of ranks (or use dp_mode=per_rank)")
    return dp_mode


def batches_per_epoch(n_cubes: int, batch_size: int, world: int, mode: str) -> int:
    """model.py:292: len(train) // batch_size; per_rank mode consumes batch_size x world samples per step (at least one step)"""
    if mode == "per_rank" and world > 1:
        return max(1, n_cubes // (batch_size * world))
    return n_cubes // batch_size


def rank_crops(n_cubes, shapes, batch_id, batch_size, patch, rank, world, mode, rng):
    """this rank's (cube index, x0, y0, mode) records of global step `batch_id`.
    shard: the reference's draw for the whole batch (model.py:304-308) on the shared stream, then this rank's slice;
    per_rank: the rank's own stream, sample i of the rank being sample (batch_id * world + rank) * batch_size + i of the epoch."""
    if mode == "per_rank" and world > 1:
        return draw_crops(n_cubes, shapes, batch_id * world + rank, batch_size, patch, rng)
    crops = draw_crops(n_cubes, shapes, batch_id, batch_size, patch, rng)
    return [crops[i] for i in dp.shard_range(batch_size, rank, world)]


class LaggedScalars:
    """Loss read-back that does not stall the launch loop (SURVEY §8(f) N1 "async loss logging"; the reference blocks on seven
    `.item()` calls per step, model.py:566-574).  After every step the 7 device scalars are copied into a slot of a pinned ring
    on the compute stream and an event is recorded; the host consumes step i only after step i+1 has been enqueued, so it
    waits for a step that is already finished (or finishing) while the device has the next one queued.  Values, order and
    count of what the caller sees are exactly those of the blocking read - only one step later."""

    def __init__(self, depth: int = 2):
        self.depth = max(2, int(depth))
        self.host = torch.empty(self.depth, 8, dtype=torch.float32, pin_memory=True)
        self.events = [torch.cuda.Event() for _ in range(self.depth)]
        self.pending = []                     # [(slot, tag)] oldest first
        self.n = 0

    def _take(self):
        slot, tag = self.pending.pop(0)
        self.events[slot].synchronize()
        return tag, self.host[slot, :7].numpy().copy()

    def push(self, scalars: torch.Tensor, tag):
        """enqueue the read-back of this step; returns the [(tag, values)] that are now due (everything but the newest)"""
        out = []
        while len(self.pending) >= self.depth:            # never overwrite a slot that has not been consumed
            out.append(self._take())
        slot = self.n % self.depth; self.n += 1
        self.host[slot, :7].copy_(scalars.detach().reshape(-1)[:7], non_blocking=True)
        self.events[slot].record()
        self.pending.append((slot, tag))
        while len(self.pending) > 1:
            out.append(self._take())
        return out

    def drain(self):
        out = []
        while self.pending:
            out.append(self._take())
        return out


def train_model(net, train_data_path, eval_data_path, batch_size, patch_size, num_epochs, ckpt_dir, eval_result_dir,
                eval_every_epoch, label_dir, mat_key="data", normalization="global_normalization", log=print,
                dp_mode="auto", seed=41):
    """model.py:236-341 on the device.  Returns the checkpoint directory.  Multi-GPU (one process per GPU): see resolve_dp_mode;
    rank 0 logs the losses of ITS shard (every loss is a batch mean, so they estimate the same quantity)."""
    rank, world, _ = dp.init_from_env()
    dev = next(net.parameters()).device
    ckpt_dir = os.path.join(ckpt_dir, "Decomposition_" + str(net.time_stamp))
    if rank == 0:
        os.makedirs(ckpt_dir, exist_ok=True); os.makedirs(eval_result_dir, exist_ok=True)
    train_files = sorted(glob.glob(os.path.join(train_data_path, "*.mat")))
    eval_files = sorted(glob.glob(os.path.join(eval_data_path, "*.mat")))
    load = lambda f: load_hsi(f, mat_key, normalization, net.global_max, net.global_min)
    train_np = [load(f) for f in train_files]
    eval_np = [load(f) for f in eval_files]
    if not train_np:
        raise ValueError(f"no .mat files under {train_data_path}")
    cubes = _to_device_cubes(train_np, dev)                    # resident in HBM for the whole run
    shapes = [c.shape for c in train_np]
    mode = resolve_dp_mode(dp_mode, batch_size, world)
    num_batches = batches_per_epoch(len(train_np), batch_size, world, mode)
    per_rank = batch_size if mode == "per_rank" else len(dp.shard_range(batch_size, rank, world))
    # shard: the process-global numpy stream main.py seeded identically on every rank (the reference's own draws);
    # per_rank: this rank's own stream (seed + rank)
    rng = np.random.RandomState(dp.rank_seed(seed, rank)) if mode == "per_rank" and world > 1 else np.random
    dp.broadcast_flat_(net.flat_parameters(), world)
    # no per-step host<->device synchronisation: crop records go up through a pinned two-slot ring (slot i is reused at step
    # i + 2, after the lagged read-back below has waited for step i), loss scalars come back one step late
    lag = LaggedScalars(2)
    nrec = max(per_rank, 1) * ctypes_sizeof_crop()
    staging = [torch.empty(nrec, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
    step_no = 0
    for epoch in range(num_epochs):
        # DecompositionNet freeze / unfreeze (model.py:274-288)
        frozen = getattr(net, "freeze_decom_epochs", 0) > 0 and epoch < net.freeze_decom_epochs
        net.set_decomposition_frozen(frozen)
        sums = np.zeros(7); count = 0

        def consume(items):
            nonlocal sums, count
            for (ep, bb), vals in items:
                sums += vals; count += 1
                if rank == 0:
                    log(f"Epoch [{ep + 1}/{num_epochs}] Batch [{bb + 1}/{num_batches}] Loss: {vals[0]:.6f}")

        for b in range(num_batches):
            crops = rank_crops(len(cubes), shapes, b, batch_size, patch_size, rank, world, mode, rng)
            x = H.assemble_batch(cubes, crops, patch_size, net.input_channels, staging=staging[step_no & 1])
            step_no += 1
            scal = net.train_step(x, world)
            consume(lag.push(scal, (epoch, b)))
        consume(lag.drain())                                  # epoch boundary: means, evaluation and checkpoints see every step
        means = sums / max(count, 1)
        for k, v in zip(LOSS_KEYS, means):
            net.all_epoch_losses[k].append(float(v))
        if rank == 0 and (epoch + 1) % eval_every_epoch == 0:
            evaluate_model(net, eval_np, eval_files, eval_result_dir, epoch + 1, label_dir, log=log)
            net.save_checkpoint(os.path.join(ckpt_dir, f"model_epoch_{epoch + 1}.pth"), epoch + 1)
            net.save_checkpoint(os.path.join(ckpt_dir, "model_epoch_latest.pth"), epoch + 1)
        if net.adaptive_lr:
            net.scheduler.step()
        if rank == 0:
            log(f"Epoch [{epoch + 1}/{num_epochs}] Average Loss: {means[0]:.6f}")
    return ckpt_dir


def _enhance_whole(net, cube_hwc: np.ndarray):
    """whole-image forward (no tiling, batch 1) like model.py:363-366 / :416-418"""
    dev = next(net.parameters()).device
    x = torch.from_numpy(np.ascontiguousarray(cube_hwc)).to(dev).unsqueeze(0).permute(0, 3, 1, 2)
    with torch.no_grad():
        R, I, D, S = net._forward_views(x)          # copied to host right below: no need for owned device tensors
        to_np = lambda t: t.squeeze(0).permute(1, 2, 0).cpu().numpy()
        return to_np(R), to_np(I), to_np(D), to_np(S)


def _save_outputs(net, out_dir, filename, R, I, D, S, save_r, save_i, save_d):
    if net.global_min is not None and net.global_max is not None:
        S = S * (net.global_max - net.global_min) + net.global_min          # model.py:371-372
    save_hsi(os.path.join(out_dir, filename), S)
    art = os.path.join(out_dir, "artifacts"); os.makedirs(art, exist_ok=True)
    stem = filename.split(".")[0]
    if save_r:
        save_hsi(os.path.join(art, stem + "_R_low.mat"), R)
    if save_i:
        save_hsi(os.path.join(art, stem + "_I_low.mat"), I)
    if save_d:
        save_hsi(os.path.join(art, stem + "_I_delta.mat"), D)


def evaluate_model(net, eval_np, eval_files, eval_result_dir, epoch, label_dir, log=print):
    if len(eval_np) <= 0:
        log(f"--- No files found for evaluation. Skipping evaluation for epoch {epoch} ---"); return None
    out_dir = os.path.join(eval_result_dir, f"epoch_{epoch}"); os.makedirs(out_dir, exist_ok=True)
    for cube, f in zip(eval_np, eval_files):
        R, I, D, S = _enhance_whole(net, cube)
        _save_outputs(net, out_dir, os.path.basename(f), R, I, D, S, net.save_reflectance, net.save_illumination, net.save_i_delta)
    try:
        m = calc_metrics(os.path.join(out_dir, "*.mat"), label_dir, data_max=net.global_max)
        net.eval_metrics[epoch] = {"psnr": m[0], "ssim": m[1], "sam": m[2]}
        return m
    except ValueError:
        return None                                                            # no labels for the eval split


def test_model(net, model_dir, test_np, test_files, save_dir, save_r=False, save_i=False, save_d=False, log=print):
    net.load_checkpoint(os.path.join(model_dir, "model_epoch_latest.pth"))
    os.makedirs(save_dir, exist_ok=True)
    total = 0.0
    for cube, f in zip(test_np, test_files):
        name = os.path.basename(f)
        log(f"Processing {name}")
        torch.cuda.synchronize(); t0 = time.time()
        R, I, D, S = _enhance_whole(net, cube)
        torch.cuda.synchronize(); dt = time.time() - t0; total += dt
        _save_outputs(net, save_dir, name, R, I, D, S, save_r, save_i, save_d)
        log(f"Processed {name} in {dt:.4f} seconds.")
    log(f"Average run time: {total / max(len(test_np), 1):.4f} seconds.")
