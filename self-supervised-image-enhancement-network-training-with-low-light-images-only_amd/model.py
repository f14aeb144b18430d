"""Drop-in `LowLightEnhance` for the MI355X hot path.

Mirrors the reference module's public surface (/root/reference/model.py:177-234, :544-575):
same constructor keywords, `forward(input_low) -> (R_low, I_low, I_delta, S)`,
`compute_loss(input_low) -> (total_loss, dict)`, `.optimizer`, `.decomposition_net`,
`.illum_adjust_net`, and the same 46 state-dict keys — but every FLOP runs in the hand-written HIP
kernels of libssie_hip.so through the plan executor (hostlib.Plan).  There is no PyTorch-operator
or CPU fallback: tensors must live on a gfx950 device.

All parameters are views into ONE flat fp32 buffer (and their gradients into one flat gradient
buffer), so Adam is a single fused kernel and data-parallel training needs exactly one RCCL
all-reduce per step (`train_step`).
"""
from __future__ import annotations

import os

import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import dp
from . import hostlib as H

LOSS_KEYS = H.LOSS_KEYS


class FusedAdam(torch.optim.Optimizer):
    """torch.optim.Adam defaults (model.py:213) as one HIP kernel over the flat parameter buffer."""

    def __init__(self, owner: "LowLightEnhance", lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self._owner = owner
        super().__init__(list(owner.parameters()), dict(lr=lr, betas=betas, eps=eps))
        self.step_count = 0
        self.exp_avg = None
        self.exp_avg_sq = None

    def _buffers(self):
        flat = self._owner._flat
        if self.exp_avg is None or self.exp_avg.device != flat.device:
            self.exp_avg = torch.zeros_like(flat) if self.exp_avg is None else self.exp_avg.to(flat.device)
            self.exp_avg_sq = torch.zeros_like(flat) if self.exp_avg_sq is None else self.exp_avg_sq.to(flat.device)
        return self.exp_avg, self.exp_avg_sq

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0):
        """torch.optim.Adam.step semantics: a parameter whose .grad is None (frozen, model.py:274-279) is skipped entirely -
        no moment decay, no update.  Parameters with gradients are stepped in contiguous runs of the flat buffer (one launch
        when nothing is frozen)."""
        o = self._owner
        o._ensure_device_layout()
        flat, gflat = o._flat, o._gflat
        runs = []                                     # [start, end) float ranges of the flat buffer that have gradients
        for (name, off, shape), p in zip(o._table, o._plist):
            n = p.numel()
            if p.grad is None:
                continue
            if p.grad.data_ptr() != gflat.data_ptr() + 4 * off:      # autograd materialised it outside the flat buffer
                gflat[off:off + n].copy_(p.grad.reshape(-1))
            end = (off + n + 3) // 4 * 4                               # tensors are 16-byte aligned: pad floats are zero
            if runs and runs[-1][1] == off:
                runs[-1][1] = end
            else:
                runs.append([off, end])
        if not runs:
            return
        m, v = self._buffers()
        g = self.param_groups[0]
        self.step_count += 1
        for a, b in runs:
            b = min(b, flat.numel())
            H.adam_step(flat[a:b], gflat[a:b], m[a:b], v[a:b], self.step_count, float(g["lr"]), grad_scale,
                        g["betas"][0], g["betas"][1], g["eps"])

    def reset_state(self):
        """what re-creating torch.optim.Adam does (model.py:284): moments and step count start over"""
        self.step_count = 0
        if self.exp_avg is not None:
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()

    # checkpoint format == torch.optim.Adam's (model.py:595-607): per-parameter 'step' / 'exp_avg' / 'exp_avg_sq'
    def state_dict(self):
        o = self._owner
        state = {}
        if self.exp_avg is not None and self.step_count > 0:
            for i, ((name, off, shape), p) in enumerate(zip(o._table, o._plist)):
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + n].view(shape).detach().clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(shape).detach().clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(o._plist)))}
        if "initial_lr" in g:
            group["initial_lr"] = g["initial_lr"]
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        o = self._owner
        o._ensure_device_layout()
        m, v = self._buffers()
        m.zero_(); v.zero_(); self.step_count = 0
        for i, st in sd.get("state", {}).items():
            name, off, shape = o._table[int(i)]
            n = o._plist[int(i)].numel()
            m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(m.device))
            v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(v.device))
            self.step_count = max(self.step_count, int(float(st["step"])))
        groups = sd.get("param_groups") or []
        if groups:
            self.param_groups[0]["lr"] = groups[0].get("lr", self.param_groups[0]["lr"])


class _LossFn(torch.autograd.Function):
    """The HIP plan computed loss AND every parameter gradient in one pass; backward hands them out."""

    @staticmethod
    def forward(ctx, owner, loss_scalar, *params):
        ctx.owner = owner
        return loss_scalar.clone()

    @staticmethod
    def backward(ctx, gout):
        o = ctx.owner
        grads = []
        for (name, off, shape), p in zip(o._table, o._plist):
            if not p.requires_grad:
                grads.append(None)
                continue
            g = o._gflat[off:off + p.numel()].view(shape)
            grads.append(g * gout)
        return (None, None, *grads)


class LowLightEnhance(nn.Module):
    def __init__(self, input_channels=64, lr=1e-3, lr_update_factor=1, lr_update_period=None, time_stamp=None,
                 c_loss_reconstruction=10, c_loss_r_fidelity=1, c_loss_i_smooth_low=1, c_loss_i_smooth_delta=20,
                 c_loss_fourier=0.2, c_loss_spectral_cons=1, alpha_i_smooth_low=1, alpha_i_smooth_delta=10,
                 device=torch.device("cpu"), global_min=None, global_max=None,
                 save_reflectance=False, save_illumination=False, save_i_delta=False):
        super().__init__()
        self.input_channels = input_channels
        self.device = device
        self.time_stamp = time_stamp
        self.c_loss_reconstruction = c_loss_reconstruction
        self.c_loss_r_fidelity = c_loss_r_fidelity
        self.c_loss_i_smooth_low = c_loss_i_smooth_low
        self.c_loss_i_smooth_delta = c_loss_i_smooth_delta
        self.c_loss_fourier = c_loss_fourier
        self.c_loss_spectral_cons = c_loss_spectral_cons
        self.alpha_i_smooth_low = alpha_i_smooth_low
        self.alpha_i_smooth_delta = alpha_i_smooth_delta
        self.lr = lr
        self.lr_update_factor = lr_update_factor
        self.lr_update_period = lr_update_period
        self.adaptive_lr = abs(lr_update_factor - 1) > 1e-6          # model.py:207-208
        self.global_min, self.global_max = global_min, global_max
        self.save_reflectance, self.save_illumination, self.save_i_delta = save_reflectance, save_illumination, save_i_delta
        self.eval_metrics = {}
        self.freeze_decom_epochs = 0
        # opt-in (not a reference kwarg): forward() under torch.no_grad() uses bf16 storage + bf16 MFMA; outputs stay fp32
        self.bf16_inference = False
        self.max_cached_plans = 4
        self._warned_forward_grad = False
        self.all_epoch_losses = {k: [] for k in LOSS_KEYS}

        self._table, total = H.param_table(input_channels)
        self._flat = torch.zeros(total, dtype=torch.float32)
        self._gflat = None
        self._plist = []
        self._plans = {}
        self._decomp_frozen = False
        self._illum_off = next(off for (name, off, shape) in self._table if name.startswith("illum_adjust_net."))
        self._build_tree()
        self._default_init()
        self.optimizer = FusedAdam(self, lr=lr)
        if self.adaptive_lr:
            self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=lr_update_period, gamma=lr_update_factor)

    # ---- parameter plumbing -------------------------------------------------------------------
    def _build_tree(self):
        """Register every parameter under the reference's module path so state-dict keys match."""
        for name, off, shape in self._table:
            parts = name.split(".")
            mod = self
            for part in parts[:-1]:
                if part not in mod._modules:
                    mod.add_module(part, nn.Module())
                mod = mod._modules[part]
            n = int(math.prod(shape))
            p = nn.Parameter(self._flat[off:off + n].view(shape))
            mod.register_parameter(parts[-1], p)
            self._plist.append(p)

    @torch.no_grad()
    def _default_init(self):
        """PyTorch default init of Conv2d / ConvTranspose2d / Linear: U(-1/sqrt(fan_in), 1/sqrt(fan_in))."""
        fan = {}
        for name, off, shape in self._table:
            if name.endswith(".weight"):
                fan[name[:-7]] = int(math.prod(shape[1:]))
        for (name, off, shape), p in zip(self._table, self._plist):
            bound = 1.0 / math.sqrt(fan[name.rsplit(".", 1)[0]])
            p.uniform_(-bound, bound)

    def _ensure_device_layout(self):
        """(Re)establish 'all parameters are views of one flat buffer' after .to()/.cuda()/load_state_dict."""
        dev = self._plist[0].device
        ok = self._flat.device == dev
        if ok:
            base = self._flat.data_ptr()
            ok = all(p.data_ptr() == base + 4 * off for (n_, off, s_), p in zip(self._table, self._plist))
        if not ok:
            flat = torch.zeros(self._flat.numel(), dtype=torch.float32, device=dev)
            for (name, off, shape), p in zip(self._table, self._plist):
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(shape)
            self._flat = flat
            self._gflat = None
            self._plans = {}
        if self._gflat is None or self._gflat.device != dev:
            self._gflat = torch.zeros_like(self._flat)
            self._plans = {}

    def coefs(self):
        return dict(c_rec=self.c_loss_reconstruction, c_rf=self.c_loss_r_fidelity, c_il=self.c_loss_i_smooth_low,
                    c_id=self.c_loss_i_smooth_delta, c_f=self.c_loss_fourier, c_sp=self.c_loss_spectral_cons,
                    alpha_low=self.alpha_i_smooth_low, alpha_delta=self.alpha_i_smooth_delta)

    def _plan_for(self, x):
        if not x.is_cuda:
            raise H.SsieError("LowLightEnhance runs only on the MI355X HIP path: move the model and input to a cuda device")
        if x.dim() != 4 or x.shape[1] != self.input_channels:
            raise H.SsieError(f"expected (N, {self.input_channels}, H, W) input, got {tuple(x.shape)}")
        if self._plist[0].device != x.device:
            raise H.SsieError("model parameters and input are on different devices")
        self._ensure_device_layout()
        key = (x.shape[0], x.shape[2], x.shape[3])
        plan = self._plans.pop(key, None)
        if plan is None:
            # a plan owns its workspace (4 GiB for 32 training patches, 17 GiB for one 1024 x 1024 cube): evaluating many
            # differently sized images must not keep them all, so only the most recently used few stay alive
            while len(self._plans) >= self.max_cached_plans:
                self._plans.pop(next(iter(self._plans)))
            plan = H.Plan(x.shape[0], self.input_channels, x.shape[2], x.shape[3], self.coefs(), self._flat, self._gflat)
            # train steps of this plan replay one hipGraph (development: SSIE_DEBUG=1 SSIE_GRAPH=0 runs them eagerly)
            plan.set_graph(not (H.debug_enabled() and os.environ.get("SSIE_GRAPH") == "0"))
            self._plans[key] = plan
            plan._coefs = tuple(self.coefs().values())
        else:
            self._plans[key] = plan                       # re-insert: dict order = recency
            if plan._coefs != tuple(self.coefs().values()):
                plan.set_coefs(self.coefs()); plan._coefs = tuple(self.coefs().values())
        return plan

    @staticmethod
    def _f32(x):
        return x if x.dtype == torch.float32 else x.float()

    # ---- reference API ------------------------------------------------------------------------
    def forward(self, input_low):
        """model.py:229-234 -> (R_low, I_low, I_delta, S), tensors the caller owns (like the reference's).
        The four outputs carry NO autograd graph: the backward pass of this build is the hand-derived one inside
        `compute_loss` / `train_step`.  Calling forward with grad enabled on trainable parameters therefore warns once."""
        R, I, D, S = self._forward_views(input_low)
        if torch.is_grad_enabled() and not self._warned_forward_grad and any(p.requires_grad for p in self._plist):
            import warnings
            warnings.warn("LowLightEnhance.forward() returns tensors without an autograd graph: a loss built on them yields no "
                          "parameter gradients.  Use compute_loss()/train_step() (hand-derived backward), or call forward "
                          "under torch.no_grad().", stacklevel=2)
            self._warned_forward_grad = True
        return R.clone(), I.clone(), D.clone(), S.clone()

    def _forward_views(self, input_low):
        """forward without the copies: views into the plan workspace, valid until the next call on this input shape
        (internal: harness and tests)."""
        x = self._f32(input_low)
        plan = self._plan_for(x)
        # mixed-precision inference (BASELINE.json configs[4]): only outside autograd, training always runs fp32
        bf16 = bool(self.bf16_inference) and not torch.is_grad_enabled() and plan.has_bf16()
        plan.enhance_fwd(x, bf16=bf16)
        b = self.input_channels
        return plan.nchw("RL_1", 0, b), plan.nchw("RL_1", b, b + 1), plan.nchw("D", 0, 1), plan.nchw("S", 0, b)

    def compute_loss(self, input_low):
        """model.py:544-575.  With grad enabled the backward pass runs here too (fused) and
        `total_loss.backward()` only distributes the already-computed gradients."""
        x = self._f32(input_low)
        plan = self._plan_for(x)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._plist)
        plan.loss_fwd_bwd(x, backward=need_grad)
        scal = plan.loss_scalars()
        vals = scal.detach().cpu().tolist()                    # ONE device->host sync (reference: 7 .item() calls)
        losses = dict(zip(LOSS_KEYS, vals))
        total = _LossFn.apply(self, scal[0], *self._plist) if need_grad else scal[0].clone()
        return total, losses

    # ---- fused fast path (bench / own training harness) -----------------------------------------
    def train_step(self, input_low, world_size: int = 1):
        """zero_grad -> compute_loss -> backward -> (RCCL all-reduce) -> Adam.step  (model.py:313-316)
        without autograd, host syncs or per-parameter launches.  Returns the 7 loss scalars (device tensor)."""
        x = self._f32(input_low)
        plan = self._plan_for(x)
        plan.loss_fwd_bwd(x, backward=True)
        return self._finish_step(world_size, plan.loss_scalars())

    def _finish_step(self, world_size, scalars=None):
        """tail of a train step on the flat gradient buffer: frozen range -> (RCCL) all-reduce -> fused Adam with the 1/world
        scale.  Frozen DecompositionNet (model.py:274-279): the reference leaves those parameters with grad None, so
        torch's Adam skips them - no update AND no moment decay; here Adam runs only on the illumination-net range."""
        gscale = dp.allreduce_flat_(self._gflat, world_size)   # one flat fp32 buffer over RCCL / xGMI
        opt = self.optimizer
        m, v = opt._buffers()
        g = opt.param_groups[0]
        opt.step_count += 1
        opt._opt_called = True                                   # lets torch's StepLR know a step happened
        a = self._illum_off if self._decomp_frozen else 0
        H.adam_step(self._flat[a:], self._gflat[a:], m[a:], v[a:], opt.step_count, float(g["lr"]), gscale,
                    g["betas"][0], g["betas"][1], g["eps"])
        return scalars

    def set_decomposition_frozen(self, frozen: bool):
        """freeze_decom_epochs semantics (model.py:274-288): frozen => DecompositionNet gets no updates; on the
        frozen -> unfrozen transition the reference re-creates Adam AND StepLR with the current lr (model.py:284-286),
        i.e. all optimiser state and the lr-decay period start over."""
        frozen = bool(frozen)
        if self._decomp_frozen and not frozen:
            self.optimizer.reset_state()
            if self.adaptive_lr:
                g = self.optimizer.param_groups[0]
                g.pop("initial_lr", None)                          # a fresh optimiser has none: StepLR restarts from the current lr
                self.scheduler = torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=self.lr_update_period,
                                                                 gamma=self.lr_update_factor)
        self._decomp_frozen = frozen
        for p in self.decomposition_net.parameters():
            p.requires_grad = not frozen

    # ---- the reference's harness methods (model.py:236, :343, :406), same argument names -----------------------------
    def train_model(self, train_data_path, eval_data_path, batch_size, patch_size, num_epochs, start_lr, ckpt_dir,
                    eval_result_dir, eval_every_epoch, label_dir, plot_every_epoch=10):
        """model.py:236-341 (called at main.py:92-105).  `start_lr` is only logged by the reference (model.py:258);
        `plot_every_epoch` drives matplotlib curves that are out of scope here."""
        from . import harness
        return harness.train_model(self, train_data_path, eval_data_path, batch_size, patch_size, num_epochs, ckpt_dir,
                                   eval_result_dir, eval_every_epoch, label_dir)

    def evaluate_model(self, eval_low_data, eval_files, eval_result_dir, epoch, label_dir):
        """model.py:343-404"""
        from . import harness
        return harness.evaluate_model(self, eval_low_data, eval_files, eval_result_dir, epoch, label_dir)

    def test_model(self, model_dir, test_low_data, test_low_data_names, save_dir, save_reflectance=False,
                   save_illumination=False, save_i_delta=False):
        """model.py:406-443 (called at main.py:120-128)"""
        from . import harness
        return harness.test_model(self, model_dir, test_low_data, test_low_data_names, save_dir, save_reflectance,
                                  save_illumination, save_i_delta)

    def flat_parameters(self):
        self._ensure_device_layout()
        return self._flat

    def flat_gradients(self):
        self._ensure_device_layout()
        return self._gflat

    def load_named(self, named: "OrderedDict[str, torch.Tensor]"):
        with torch.no_grad():
            for (name, off, shape), p in zip(self._table, self._plist):
                p.copy_(named[name].to(p.device, torch.float32))

    def save_checkpoint(self, path, epoch):
        torch.save({"epoch": epoch, "model_state_dict": self.state_dict(),
                    "optimizer_state_dict": self.optimizer.state_dict()}, path)
        print(f"Checkpoint saved at {path}")

    def load_checkpoint(self, path):
        """model.py:603-607; tensors-only loader (never unpickles code)"""
        ck = torch.load(path, map_location=self._plist[0].device, weights_only=True)
        self.load_state_dict(ck["model_state_dict"])
        if "optimizer_state_dict" in ck:
            self.optimizer.load_state_dict(ck["optimizer_state_dict"])
        print(f"Loaded checkpoint from {path}")
