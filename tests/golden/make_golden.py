#!/usr/bin/env python3
"""Generate golden fixtures by running the REAL reference (build container only).

Imports /root/reference/model.py (read-only; never copied, never shipped), drives
`LowLightEnhance` with the closed-form parameters / inputs of `oracle.ssie_oracle`, and
writes reference OUTPUTS (data only) to tests/golden/*.npz:

  case_b5_16   N=1 B=5  16x16   default coefs : full R,I,D,S,E, 7 losses, grads (full <=40000 elems, else ::53), params after 1 & 3 Adam steps (::97)
  case_b31_32  N=2 B=31 32x32   JYU coefs     : strided sub-samples + fp64 checksums of R,I,D,S,E, 7 losses, grad norms + small grads (others ::53), params after 1 & 3 steps
  case_b31_64  N=2 B=31 64x64   JYU coefs     : strided sub-samples + fp64 checksums, 7 losses, grad norms
  case_b64_32  N=2 B=64 32x32   JYU coefs     : the reference's own band count (constructor default model.py:178 and `channels: 64`
               in all eight shipped configs): strided sub-samples + checksums, 7 losses, grad norms + small grads, params after 1 & 3 steps
  aux          Fourier masks (16/64/128), nearest-upsample index vectors, crop+augment patches
  hsi_raw.mat  a small raw cube written by the reference's utils.save_hsi (utils.py:171-178)
  io           utils.load_hsi (utils.py:36-57) of that file in every normalisation mode (None, self, global with min 238 /
               max 4095 - values below the min clamp to 0 -, global with min None, per-channel normalisation / standardisation)

The logging/plot dependencies the reference imports at module top (mlflow, torchinfo,
torchmetrics, skimage) are not installed here and are not on the hot path; empty module
objects are registered for them so `import model` succeeds.

Run:  python tests/golden/make_golden.py         (requires /root/reference)
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import ssie_oracle as O  # noqa: E402


def import_reference():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m
    nop = lambda *a, **k: None
    mod("torchinfo", summary=nop)
    mod("mlflow", set_experiment=nop, start_run=nop, log_param=nop, log_params=nop,
        log_metric=nop, log_metrics=nop, log_artifact=nop)
    tm = mod("torchmetrics"); tf = mod("torchmetrics.functional")
    ti = mod("torchmetrics.functional.image", peak_signal_noise_ratio=nop,
             structural_similarity_index_measure=nop, spectral_angle_mapper=nop)
    tm.functional = tf; tf.image = ti
    sk = mod("skimage"); sk.metrics = mod("skimage.metrics", peak_signal_noise_ratio=nop, structural_similarity=nop)
    sys.path.insert(0, REF)
    import model as ref_model          # noqa
    import utils as ref_utils          # noqa
    return ref_model, ref_utils


def build_ref(ref_model, bands, coefs, lr=1e-3):
    torch.manual_seed(0)
    net = ref_model.LowLightEnhance(
        input_channels=bands, lr=lr,
        c_loss_reconstruction=coefs["c_rec"], c_loss_r_fidelity=coefs["c_rf"],
        c_loss_i_smooth_low=coefs["c_il"], c_loss_i_smooth_delta=coefs["c_id"],
        c_loss_fourier=coefs["c_f"], c_loss_spectral_cons=coefs["c_sp"],
        alpha_i_smooth_low=coefs["alpha_low"], alpha_i_smooth_delta=coefs["alpha_delta"])
    P = O.closed_form_params(bands)
    sd = net.state_dict()
    assert list(sd.keys()) == list(P.keys()), "state-dict key order differs from oracle table"
    for k in sd:
        assert tuple(sd[k].shape) == tuple(P[k].shape), k
    net.load_state_dict(P)
    return net


def run_case(ref_model, name, n, bands, hw, coefs, full, steps=3):
    torch.set_num_threads(8)
    net = build_ref(ref_model, bands, coefs)
    x = O.synthetic_patches(n, bands, hw, hw)
    out = {"meta_torch": np.array(torch.__version__), "n": n, "bands": bands, "hw": hw}
    net.eval()
    with torch.no_grad():
        R, I, D, S = net(x)
        E, _ = net.decomposition_net(S)
    net.train()
    tensors = dict(R=R, I=I, D=D, S=S, E=E)
    for k, v in tensors.items():
        v = v.contiguous()
        out["sum64_" + k] = np.float64(v.double().sum().item())
        out["abs64_" + k] = np.float64(v.double().abs().sum().item())
        if full:
            out[k] = v.numpy()
        else:
            out[k + "_sub"] = v[:, ::5, ::7, ::9].contiguous().numpy()
    # train steps
    for step in range(1, steps + 1):
        net.optimizer.zero_grad()
        loss, ld = net.compute_loss(x)
        loss.backward()
        if step == 1:
            out["losses"] = np.array([ld[k] for k in O.LOSS_KEYS], dtype=np.float64)
            names = []
            norms = []
            for k, p in net.named_parameters():
                names.append(k); norms.append(p.grad.double().norm().item())
                if p.grad.numel() <= (40000 if full else 4096):
                    out["grad/" + k] = p.grad.detach().numpy().copy()
                else:
                    out["grad_sub/" + k] = p.grad.detach().flatten()[::53].numpy().copy()
            out["grad_names"] = np.array(names)
            out["grad_norms"] = np.array(norms, dtype=np.float64)
        net.optimizer.step()
        if step in (1, steps):
            for k, p in net.named_parameters():
                pd = p.detach()
                out[f"psum{step}/" + k] = np.float64(pd.double().sum().item())
                if pd.numel() <= 4096:
                    out[f"param{step}/" + k] = pd.numpy().copy()
                else:
                    out[f"param{step}_sub/" + k] = pd.flatten()[::97].numpy().copy()
        if step == steps:
            out["losses_step%d" % steps] = np.array([ld[k] for k in O.LOSS_KEYS], dtype=np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, "%.1f KB" % (os.path.getsize(path) / 1024), "losses", out["losses"])


def run_aux(ref_model, ref_utils):
    out = {}
    # Fourier masks exactly as model.py:460-464 builds them
    for hw in (16, 64, 128):
        y = torch.linspace(-1, 1, hw); x = torch.linspace(-1, 1, hw)
        Y, X = torch.meshgrid(y, x, indexing="ij")
        out["mask%d" % hw] = np.packbits((torch.sqrt(X ** 2 + Y ** 2) >= 0.1).numpy().astype(np.uint8))
    # rectangular mask
    y = torch.linspace(-1, 1, 32); x = torch.linspace(-1, 1, 48)
    Y, X = torch.meshgrid(y, x, indexing="ij")
    out["mask32x48"] = np.packbits((torch.sqrt(X ** 2 + Y ** 2) >= 0.1).numpy().astype(np.uint8))
    # nearest-upsample source indices (F.interpolate mode='nearest'), incl. odd sizes
    for (i, o) in ((16, 32), (17, 33), (13, 25), (4, 16), (5, 18)):
        src = torch.arange(i, dtype=torch.float32).reshape(1, 1, 1, i)
        out["nearest_%d_%d" % (i, o)] = torch.nn.functional.interpolate(
            src, size=(1, o), mode="nearest").flatten().numpy().astype(np.int32)
    # crop + augmentation (model.py:306-309 + utils.py:7-34): cube, (x, y, mode) -> patch
    cube = O.synthetic_patches(1, 6, 24, 20)[0].permute(1, 2, 0).contiguous().numpy()   # H,W,C host cube
    out["aug_cube"] = cube
    crops = []
    for mode in range(8):
        x0, y0 = (3 + mode) % 8, (5 * mode) % 4
        patch = ref_utils.data_augmentation(cube[x0:x0 + 16, y0:y0 + 16, :], mode)
        out["aug_patch_%d" % mode] = np.ascontiguousarray(patch)
        crops.append((x0, y0, mode))
    out["aug_crops"] = np.array(crops, dtype=np.int32)
    path = os.path.join(HERE, "aux.npz")
    np.savez_compressed(path, **out)
    print("aux ->", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def run_io(ref_utils):
    """N2: the on-disk format either side of the path.  Raw cube = integer-hash values in [100, 4500): below global_min 238
    (negative after the shift -> clamped) and above global_max 4095 (> 1 before the second normalisation)."""
    raw = (100.0 + 4400.0 * O._hash_uniform(14 * 12 * 6, 777).reshape(14, 12, 6)).astype("float32")
    raw[3, 4, :] = raw[3, 4, 0]                      # a constant pixel; channel 5 constant -> per-channel range/std guards
    raw[:, :, 5] = 1234.0
    path = os.path.join(HERE, "hsi_raw.mat")
    ref_utils.save_hsi(path, raw)                    # reference writer
    out = {"raw": raw}
    out["none"] = ref_utils.load_hsi(path)
    out["self"] = ref_utils.load_hsi(path, "data", "self")
    out["global_238_4095"] = ref_utils.load_hsi(path, "data", "global_normalization", 4095.0, 238.0)
    out["global_none_4095"] = ref_utils.load_hsi(path, "data", "global_normalization", 4095.0, None)
    out["per_channel_normalization"] = ref_utils.load_hsi(path, "data", "per_channel_normalization")
    out["per_channel_standardization"] = ref_utils.load_hsi(path, "data", "per_channel_standardization")
    # postfix form of save_hsi (model.py:431-439 artifacts): file name only
    tmp = os.path.join(HERE, "_tmp_io.mat")
    ref_utils.save_hsi(tmp, raw[:2, :2], postfix="_R_low", key="data")
    assert os.path.exists(tmp[:-4] + "_R_low.mat")
    os.remove(tmp[:-4] + "_R_low.mat")
    p = os.path.join(HERE, "io.npz")
    np.savez_compressed(p, **out)
    print("io ->", p, "%.1f KB" % (os.path.getsize(p) / 1024), {k: (v.dtype, float(v.max())) for k, v in out.items()})


if __name__ == "__main__":
    ref_model, ref_utils = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "io":
        run_io(ref_utils)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "b64":       # round 4: added without re-writing the older fixtures
        run_case(ref_model, "case_b64_32", 2, 64, 32, O.JYU_COEFS, full=False)
        sys.exit(0)
    run_case(ref_model, "case_b5_16", 1, 5, 16, O.DEFAULT_COEFS, full=True)
    run_case(ref_model, "case_b31_32", 2, 31, 32, O.JYU_COEFS, full=False)
    run_case(ref_model, "case_b31_64", 2, 31, 64, O.JYU_COEFS, full=False)
    run_case(ref_model, "case_b64_32", 2, 64, 32, O.JYU_COEFS, full=False)
    run_aux(ref_model, ref_utils)
    run_io(ref_utils)
