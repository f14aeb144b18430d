#!/usr/bin/env python3
"""`python main.py --config <yml> [--key value ...]` — the reference's train / test / train_and_test entry
(/root/reference/main.py:16-90, 147-281) on the MI355X hot path.

Same keys, same priority (CLI > YAML > default), same derived directories.  Differences, all deliberate:
  * `use_gpu` must be 1: the product has no CPU path;
  * mlflow logging is used only when mlflow is importable;
  * `train_and_test` works on Linux (the reference writes `Decomposition_<ts>` but reads `decomposition_<ts>`, main.py:87);
  * in `test` phase the checkpoint timestamp comes from `--timestamp` (the reference hard-codes a literal, main.py:78-80);
  * exceptions propagate with a non-zero exit code (the reference swallows them, main.py:266-270);
  * multi-GPU: launch under `python -m torch.distributed.run --nproc-per-node N main.py ...`; `--dp_mode shard` splits ONE batch of
    `batch_size` over the ranks (bit-comparable with one GPU), `per_rank` gives every rank `batch_size` patches from its own RNG
    stream (seed + rank; the reference's batch 1-2 on 8 GPUs), `auto` (default) shards when batch_size divides by the rank count.
"""
import argparse
import glob
import os
import random
import sys
from datetime import datetime

import numpy as np
import torch
import yaml

import ssie

DEFAULTS = {
    "use_gpu": 1, "seed_value": 41, "gpu_idx": "0", "gpu_mem": 0.8, "decom": 0, "mat_key": "data", "channels": 64,
    "global_min": 0., "global_max": 1., "normalization": "global_normalization", "batch_size": 1, "patch_size": 128,
    "start_lr": 0.001, "lr_update_factor": 1, "lr_update_period": 400, "train_data": "./data/train/low",
    "eval_data": "./data/eval/low", "test_data": "./data/test/low", "label_dir": "./data/test/high",
    "phase": "train_and_test", "epoch": 400, "eval_every_epoch": 200, "plot_every_epoch": 200,
    "c_loss_reconstruction": 10., "c_loss_r_fidelity": 1., "c_loss_i_smooth_low": 1., "c_loss_i_smooth_delta": 20.,
    "c_loss_fourier": 0.2, "c_loss_spectral_cons": 1., "alpha_i_smooth_low": 1., "alpha_i_smooth_delta": 10.,
    "save_reflectance": False, "save_illumination": False, "save_i_delta": False, "bf16_inference": 0, "model_name": "no_name_model",
    "pretrained_model": "", "freeze_decom_epochs": 0,
    # extra key (the reference is single-process): how `batch_size` is read under torch.distributed.run - "shard" = one global batch
    # split over the ranks, "per_rank" = batch_size patches per rank from per-rank RNG streams, "auto" = shard when it divides
    "dp_mode": "auto",
}


def _flag_type(v):
    if isinstance(v, bool):
        return lambda s: str(s).lower() in ("1", "true", "yes", "y")
    return type(v)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Parse config from YAML and command-line.")
    ap.add_argument("--config", type=str, default="./config/config_outdoor_jyu.yml")
    ap.add_argument("--timestamp", type=str, default=None, help="checkpoint timestamp to test (phase=test)")
    ap.add_argument("--preset", type=str, default=None, help="preset name inside a presets file (config/presets.yml)")
    for k, v in DEFAULTS.items():
        ap.add_argument(f"--{k}", type=_flag_type(v), default=None)
    args = ap.parse_args(argv)
    with open(args.config, "r") as f:
        cfg = yaml.safe_load(f) or {}
    if "presets" in cfg:                                  # config/presets.yml: base file + the keys a preset changes
        name = args.preset or "outdoor_jyu"
        fold = None
        if name not in cfg["presets"] and name[:-1] in cfg["presets"] and name[-1].isdigit():
            name, fold = name[:-1], name[-1]              # indoor_li_et_al_cv3 -> preset indoor_li_et_al_cv, fold 3
        if name not in cfg["presets"]:
            raise SystemExit(f"unknown preset {args.preset!r}; available: {sorted(cfg['presets'])}")
        with open(os.path.join(os.path.dirname(os.path.abspath(args.config)), cfg["base"]), "r") as f:
            merged = yaml.safe_load(f) or {}
        for k, v in (cfg["presets"][name] or {}).items():
            merged[k] = v.format(fold=fold) if isinstance(v, str) and fold is not None else v
        cfg = merged
    for k, dv in DEFAULTS.items():                       # CLI > YAML > default
        if getattr(args, k) is None:
            setattr(args, k, cfg.get(k, dv))
    ts = f"{datetime.now():%Y%m%d_%H%M%S}"
    postfix = ""
    if args.phase == "test":
        if not args.timestamp:
            raise SystemExit("phase=test needs --timestamp <YYYYmmdd_HHMMSS> of the checkpoint to load")
        postfix = "_test_" + ts
        ts = args.timestamp
    args.timestamp = ts
    args.full_model_name = args.model_name + "_" + ts + postfix
    args.model_ckpt_dir = "./checkpoint/" + args.model_name
    args.eval_result_dir = "./results/eval_results_" + args.full_model_name
    args.test_result_dir = "./results/test_results_" + args.full_model_name
    args.test_model_dir = "./checkpoint/" + args.model_name + "/Decomposition_" + ts
    args.log_file_path = "./logs/" + args.full_model_name + ".log"
    return args


def build_model(args, device):
    ssie.load()
    from ssie_amd.model import LowLightEnhance
    net = LowLightEnhance(
        input_channels=args.channels, lr=args.start_lr, lr_update_factor=args.lr_update_factor,
        lr_update_period=args.lr_update_period, time_stamp=args.timestamp,
        c_loss_reconstruction=args.c_loss_reconstruction, c_loss_r_fidelity=args.c_loss_r_fidelity,
        c_loss_i_smooth_low=args.c_loss_i_smooth_low, c_loss_i_smooth_delta=args.c_loss_i_smooth_delta,
        c_loss_fourier=args.c_loss_fourier, c_loss_spectral_cons=args.c_loss_spectral_cons,
        alpha_i_smooth_low=args.alpha_i_smooth_low, alpha_i_smooth_delta=args.alpha_i_smooth_delta, device=device,
        global_min=args.global_min, global_max=args.global_max, save_reflectance=args.save_reflectance,
        save_illumination=args.save_illumination, save_i_delta=args.save_i_delta)
    net = net.to(device)
    net.bf16_inference = bool(int(args.bf16_inference))          # extra key (not in the reference): bf16 test/eval forward, fp32 training
    if args.pretrained_model:                                            # main.py:196-212
        ck = torch.load(args.pretrained_model, map_location=device, weights_only=True)
        net.load_state_dict(ck["model_state_dict"] if "model_state_dict" in ck else ck)
        net.freeze_decom_epochs = args.freeze_decom_epochs
    return net


def main(args):
    if not args.use_gpu or not torch.cuda.is_available():
        raise SystemExit("this build runs only on an MI355X (use_gpu must be 1 and a GPU must be visible)")
    random.seed(args.seed_value); np.random.seed(args.seed_value); torch.manual_seed(args.seed_value)
    ssie.load()
    from ssie_amd import dp, harness
    rank, world, local = dp.init_from_env()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    net = build_model(args, device)
    if args.phase in ("train", "train_and_test"):
        harness.train_model(net, args.train_data, args.eval_data, args.batch_size, args.patch_size, args.epoch,
                            args.model_ckpt_dir, args.eval_result_dir, args.eval_every_epoch, args.label_dir,
                            mat_key=args.mat_key, normalization=args.normalization, dp_mode=args.dp_mode, seed=args.seed_value)
    if args.phase in ("test", "train_and_test") and rank == 0:
        files = sorted(glob.glob(os.path.join(args.test_data, "*.*")))
        print("Found test files:", files)
        cubes = [harness.load_hsi(f, args.mat_key, args.normalization, args.global_max, args.global_min) for f in files]
        harness.test_model(net, args.test_model_dir, cubes, files, args.test_result_dir, args.save_reflectance,
                           args.save_illumination, args.save_i_delta)
        try:
            p, s, a = harness.calc_metrics(os.path.join(args.test_result_dir, "*.mat"), args.label_dir, data_max=args.global_max)
            print(f"PSNR_dB {p:.4f}  SSIM {s:.4f}  SAM {a:.4f}")
        except ValueError as e:
            print("metrics skipped:", e)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(parse_args())
