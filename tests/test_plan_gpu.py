"""GPU parity of the whole hot path (plan executor, through the C-ABI) against the CPU oracle.

Stage-by-stage: every forward activation, the seven loss scalars, the loss cotangents, a set of
intermediate gradients and all 46 parameter gradients, then Adam.  Tolerances:
outputs abs <= 1e-5, loss scalars rel <= 1e-5 vs the fp32 reference fixtures (5e-5 vs the fp64 oracle: the reference's own
fp32 rounding of these ~1e-4 means is ~1e-5), gradients rel-L2 <= 1e-3 per tensor
(`k_linear.bias` excluded: analytically zero), enhanced-cube PSNR vs oracle > 100 dB.
The golden fixtures (reference outputs) are checked too, so the chain reference -> oracle -> HIP is closed.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ssie_oracle as O

pytestmark = pytest.mark.gpu

CASES = {
    "b5_16": (1, 5, 16, 16, O.DEFAULT_COEFS),
    "b31_32": (2, 31, 32, 32, O.JYU_COEFS),
    "b31_64": (2, 31, 64, 64, O.JYU_COEFS),
    "b64_32": (2, 64, 32, 32, O.JYU_COEFS),          # the reference's own band count (model.py:178; `channels: 64` in every shipped config):
    "b64_64": (2, 64, 64, 64, O.JYU_COEFS),          # R|I of 65 channels in a 68-float pixel, recon with a ragged third 32-channel block,
    "b64_128": (2, 64, 128, 128, O.JYU_COEFS),       # conv0 of the illumination net at Cin = 65; b64_128 = config_outdoor_jyu.yml:7,11-12 (batch 2, 128 x 128 x 64)
    "b8_32x64": (3, 8, 32, 64, O.JYU_COEFS),
    "b31_128": (1, 31, 128, 128, O.JYU_COEFS),       # BASELINE.json configs[1] geometry (one patch of the bench batch)
    "b256_64": (1, 256, 64, 64, O.JYU_COEFS),        # BASELINE.json configs[2]: 256-band cubes
    "b8_24x40": (2, 8, 24, 40, O.JYU_COEFS),         # patch size that is not a power of two: direct-DFT Fourier loss
    "b5_96": (1, 5, 96, 96, O.DEFAULT_COEFS),
    "b5_256": (1, 5, 256, 256, O.JYU_COEFS),         # patch_size 256 (model.py:456-473 takes any): three-pass Fourier loss
    "b4_160x288": (1, 4, 160, 288, O.JYU_COEFS),     # above the LDS plane limit and not powers of two: three-pass direct DFT
}
GRAD_FLOOR = {"b256_64": 1e-2}     # 256 bands: cotangent noise 5-7e-2, deep-layer gradients see 3-7e-3 of it (which signs flip moves with
                                   # every 1e-7 change of the forward; the noise-free pin of this case is tests/test_backward_gpu.py at 2e-5)


@pytest.fixture(scope="module")
def pkg():
    import ssie
    ssie.load()
    from ssie_amd import hostlib, model
    assert hostlib.lib().ssie_device_ok() == 1
    return hostlib, model


def build_plan(H, n, bands, h, w, coefs, P=None):
    table, total = H.param_table(bands)
    P = P or O.closed_form_params(bands)
    flat = torch.zeros(total, device="cuda")
    for name, off, shape in table:
        flat[off:off + P[name].numel()] = P[name].reshape(-1).cuda()
    gflat = torch.zeros_like(flat)
    return H.Plan(n, bands, h, w, coefs, flat, gflat), table, flat, gflat, P


def rel_l2(a, b):
    return (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-30)


@pytest.fixture(params=[None, "dma_kernels", "winograd", "winograd4"])
def forced_kernels(pkg, request):
    """None = launch heuristics (8x16 tiles at these small batches).  "dma_kernels" forces the kernels the bench-size
    layers run - 16x16 and, where eligible, 16x32 tiles (conv_fprop_v2_kernel / conv_fprop_v2w_kernel) - through the same
    parity cases, so the whole fwd + bwd chain is checked on them too."""
    L = pkg[0].lib()
    if request.param in ("winograd", "winograd4"):  # every stride-1 3x3 forward / data-gradient launch on conv_wino_kernel (F(2x2,3x3)) -
        # "winograd4": on conv_wino4_kernel (F(4x4,3x3)) wherever it is eligible (sources at their own resolution, >= 48 columns)
        L.ssie_debug_set_wino4_min_tiles(1 if request.param == "winograd4" else 1 << 30)
        L.ssie_debug_set_wino_min_tiles(1)
        L.ssie_debug_set_wgrad_wino_min_tiles(1)
        L.ssie_debug_set_tconv_min_tiles(1)       # and the one-launch transposed convolution (conv_tconv.hip)
        L.ssie_debug_set_wino_half_below(0 if request.param == "winograd4" else 256)   # F(2x2) workgroups: 32 channels wide in the "winograd4" runs, 16 (under-filled launches) otherwise
    elif request.param:
        L.ssie_debug_set_wino4_min_tiles(1 << 30)
        L.ssie_debug_set_fprop_min_tiles16(0)
        L.ssie_debug_set_fprop_wide_min_tiles(1)
        L.ssie_debug_set_fprop_v2_split_min_tiles(1)
        L.ssie_debug_set_wino_min_tiles(1 << 30)
        L.ssie_debug_set_wgrad_wino_min_tiles(1 << 30)  # (the Winograd kernel is forced in its own fixture value below)
        L.ssie_debug_set_skinny_final(0)          # and final_conv (64 -> 1) on the MFMA tiles instead of the VALU kernels
        L.ssie_debug_set_spectral9(0)             # and the 9 x 9 convolution on the direct MFMA kernels instead of the frequency domain
    yield request.param
    L.ssie_debug_set_wino_half_below(256)
    L.ssie_debug_set_tconv_min_tiles(-1)
    L.ssie_debug_set_wino4_min_tiles(-1)              # the library's default
    L.ssie_debug_set_wino_min_tiles(-1)               # the library's default
    L.ssie_debug_set_wgrad_wino_min_tiles(-1)
    L.ssie_debug_set_skinny_final(1)
    L.ssie_debug_set_spectral9(1)
    L.ssie_debug_set_fprop_min_tiles16(256)
    L.ssie_debug_set_fprop_wide_min_tiles(512)
    L.ssie_debug_set_fprop_v2_split_min_tiles(1024)


@pytest.mark.parametrize("case", list(CASES))
def test_stagewise_parity(pkg, case, forced_kernels):
    H, _ = pkg
    if forced_kernels and (case in ("b5_16", "b5_96", "b5_256", "b4_160x288", "b64_32") or (case == "b64_128" and forced_kernels != "winograd4") or
                           (case == "b256_64" and forced_kernels not in ("winograd", "winograd4"))):
        pytest.skip("forced-kernel variant runs on the mid-size cases only (time); 256 bands: the Winograd / tconv kernels only")
    n, bands, h, w, coefs = CASES[case]
    plan, table, flat, gflat, P = build_plan(H, n, bands, h, w, coefs)
    x = O.synthetic_patches(n, bands, h, w)
    plan.loss_fwd_bwd(x.cuda(), backward=True)
    torch.cuda.synchronize()

    P64 = {k: v.double() for k, v in P.items()}
    tr = {}
    vals, grads, outs = O.loss_and_grads(P64, x.double(), coefs, tr)
    # the reference arithmetic itself (plain PyTorch fp32) vs fp64: L1 losses make gradients sign-like, so a
    # handful of sg() flips near zero move rel-L2 by up to a few 1e-2 on cotangents; HIP must be no worse than
    # 2x that intrinsic fp32 noise (or 1e-3)
    tr32 = {}
    _, grads32, _ = O.loss_and_grads(P, x, coefs, tr32)
    report, bad = [], []

    # Every parameter gradient is a linear functional of the loss cotangents, and those carry the sg() flip noise directly
    # (d/dD below: 2-7e-2 in BOTH fp32 evaluations).  One fp32 oracle run gives a noisy per-tensor estimate of how much of
    # it survives the averaging into a deep-layer gradient, so the floor is tied to the cotangent noise itself: 0.15 x
    # the fp32 oracle's own d/dD deviation (never below 1e-3; which signs flip differs between two valid fp32 forwards - direct,
    # frequency-domain and Winograd convolutions - and the worst tensor seen is 0.102 x).  The noise-free pin of the same
    # launches is tests/test_backward_gpu.py (fixed 2e-5).
    noise_D = rel_l2(tr32["D"].grad, tr["D"].grad)
    floor = max(GRAD_FLOOR.get(case, 1e-3), 0.15 * noise_D)

    def gtol(ref32, ref64):
        return max(floor, 2.0 * rel_l2(ref32, ref64))

    def chk(label, got, ref, tol, kind="abs"):
        got = got.detach().double().cpu(); ref = ref.detach().double()
        err = (got - ref).abs().max().item() if kind == "abs" else rel_l2(got, ref)
        report.append(f"{label:34s} {kind} {err:.3e} (tol {tol:.3e})")
        if not (err <= tol):
            bad.append(report[-1])

    B = bands
    fwd = ["c0_1", "sh_1", "c1_1", "c2_1", "c3_1", "dc_1", "c5_1", "c7_1", "a0", "a1", "a2", "a3", "ao", "f1", "t3",
           "u1", "d1", "u2", "d2", "u3", "d3", "f", "c0_2", "sh_2", "c1_2", "c2_2", "c3_2", "dc_2", "c5_2", "c7_2"]
    for name in fwd:
        ref = tr[name]
        scale = max(1.0, ref.abs().max().item())
        chk(name, plan.nchw(name), ref, 2e-5 * scale)
    R, I, D, S, E = outs
    chk("R_low", plan.nchw("RL_1", 0, B), R, 1e-5); chk("I_low", plan.nchw("RL_1", B, B + 1), I, 1e-5)
    chk("I_delta", plan.nchw("D", 0, 1), D, 1e-5); chk("S", plan.nchw("S", 0, B), S, 1e-5)
    chk("R_enh", plan.nchw("RL_2", 0, B), E, 1e-5)
    psnr = O.psnr(plan.nchw("S", 0, B).cpu(), S)
    report.append(f"PSNR(S_hip, S_oracle) = {psnr:.1f} dB")
    if psnr < 100:
        bad.append(report[-1])

    got_l = plan.loss_scalars().cpu().double().numpy()
    ref_l = np.array([vals[k] for k in O.LOSS_KEYS])
    for k, g, r in zip(O.LOSS_KEYS, got_l, ref_l):
        e = abs(g - r) / max(abs(r), 1e-30)
        report.append(f"{k:34s} rel {e:.3e} (tol 5e-05 vs fp64 oracle)  hip={g:.8e} oracle={r:.8e}")
        if e > 5e-5:
            bad.append(report[-1])

    # intermediate gradients (state at the END of the step = pass-1 backward)
    inter = [("gS", "S", B), ("gD", "D", 1), ("G8", "c8_1", B + 1), ("G7", "c7_1", 64), ("Gsh", "sh_1", 64),
             ("Gf", "f", 64), ("gd3", "a0", 64)]
    # The cotangents ARE the flip noise: which signs flip differs between any two fp32 forwards, and ONE fp32 oracle run is a noisy
    # estimate of a tensor's own share of it - b4_160x288's d/dS measured 6.665e-3 against 2 x 3.311e-3 (2.01 x, identical bits with
    # every executor / reduction mode, so not a kernel difference).  Same rule as the q/k gradients below: never tighter than 0.3 x the
    # d/dD deviation of the fp32 oracle (2.87e-2 here: d/dS is 0.23 x).  The noise-free pins are tests/test_backward_gpu.py (2e-5).
    ctol = lambda r32, r64: max(gtol(r32, r64), 0.3 * noise_D)
    for buf, key, c in inter:
        chk("d/d " + key + " [" + buf + "]", plan.nchw(buf, 0, c), tr[key].grad, ctol(tr32[key].grad, tr[key].grad), "rel")
    gRI = torch.cat([tr["R"].grad, tr["I"].grad], 1)
    gRI32 = torch.cat([tr32["R"].grad, tr32["I"].grad], 1)
    chk("d/d (R,I) [gRL]", plan.nchw("gRL", 0, B + 1), gRI, ctol(gRI32, gRI), "rel")

    for name, off, shape in table:
        if name.endswith("k_linear.bias"):
            continue
        g = gflat[off:off + int(np.prod(shape))].view(shape)
        tol = gtol(grads32[name], grads[name])
        if ".q_linear." in name or ".k_linear." in name:
            # dS = P (dP - delta) cancels almost completely for these fixtures (|dq|,|dk| ~ 1e-4 |dv|), so the upstream
            # fp32 sign-flip noise in dO (~7e-4) is amplified; the kernel itself is pinned to 2e-5 in test_attention_gpu.py
            tol = max(tol, 5e-3, 0.3 * noise_D)
        chk("grad " + name, g, grads[name], tol, "rel")

    print("\n".join(report))
    assert not bad, "parity failures:\n" + "\n".join(bad)


def test_side_stream_overlap_is_bit_identical(pkg):
    """The slab reductions of the weight gradients can run on a side stream (ssie_debug_set_overlap(1): per-op slab flags, events
    both ways); the default is launch order on the caller's stream.  Same kernels, same summation orders: the two executors must
    give bit-identical gradients and losses (a missing slab dependency shows up as a wrong gradient, as it did in round 2)."""
    H, _ = pkg
    L = H.lib()
    n, bands, h, w, coefs = 2, 31, 64, 64, O.JYU_COEFS
    x = O.synthetic_patches(n, bands, h, w).cuda()
    out = []
    try:
        for mode in (0, 1):
            L.ssie_debug_set_overlap(mode)
            plan, table, flat, gflat, P = build_plan(H, n, bands, h, w, coefs)
            for _ in range(2):                              # twice: the second run reuses slab areas the first one filled
                gflat.zero_()
                plan.loss_fwd_bwd(x, backward=True)
            torch.cuda.synchronize()
            out.append((gflat.clone(), plan.loss_scalars().clone()))
    finally:
        L.ssie_debug_set_overlap(0)
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])


def test_batched_slab_reduction_is_bit_identical(pkg):
    """Default: every weight-gradient slab reduction of a backward pass in ONE launch at its end (each producer owns its slab region);
    ssie_debug_set_batched_reduce(0): one launch per layer right behind its producer.  Same body, same per-element summation order:
    identical bits, and the batched list must be the shorter one by (layers - 1) launches."""
    H, _ = pkg
    L = H.lib()
    n, bands, h, w, coefs = 2, 31, 64, 64, O.JYU_COEFS
    x = O.synthetic_patches(n, bands, h, w).cuda()
    out, launches = [], []
    try:
        for mode in (1, 0):
            L.ssie_debug_set_batched_reduce(mode)
            plan, table, flat, gflat, P = build_plan(H, n, bands, h, w, coefs)
            for _ in range(2):
                gflat.zero_()
                plan.loss_fwd_bwd(x, backward=True)
            torch.cuda.synchronize()
            out.append((gflat.clone(), plan.loss_scalars().clone()))
            launches.append(plan.profile_step(x)["wgrad_reduce_kernel"][2])
    finally:
        L.ssie_debug_set_batched_reduce(1)
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    assert launches[0] == 1 and launches[1] > 10, launches


# Full small gradients vs the reference's fp32 values: two independent fp32 evaluations, each with its own sg() flips.  At 64 bands on
# 32 x 32 planes the REFERENCE's own fp32 gradients sit up to 8.9e-3 (conv3.0.bias; shallow_conv.0.bias 4.1e-3) from the fp64 oracle
# (at 31 bands: 1.8e-4) - measured in the build container - so that case gets 2e-2; the noise-free pins of the same launches are the
# b64_64 / b64_128 injected chains (tests/test_backward_gpu.py, fixed 2e-5).
GOLDEN_GRAD_TOL = {"b64_32": 2e-2}


@pytest.mark.parametrize("case", ["b5_16", "b31_32", "b31_64", "b64_32"])
def test_golden_reference_outputs(pkg, golden_dir, case):
    """HIP path vs the reference's own outputs (fixtures made by tests/golden/make_golden.py)."""
    H, _ = pkg
    n, bands, h, w, coefs = CASES[case]
    g = np.load(os.path.join(golden_dir, "case_" + case + ".npz"))
    plan, table, flat, gflat, P = build_plan(H, n, bands, h, w, coefs)
    x = O.synthetic_patches(n, bands, h, w)
    plan.loss_fwd_bwd(x.cuda(), backward=True)
    torch.cuda.synchronize()
    B = bands
    got = dict(R=plan.nchw("RL_1", 0, B), I=plan.nchw("RL_1", B, B + 1), D=plan.nchw("D", 0, 1),
               S=plan.nchw("S", 0, B), E=plan.nchw("RL_2", 0, B))
    for key, t in got.items():
        t = t.cpu()
        if key in g.files:
            assert np.abs(t.numpy() - g[key]).max() <= 1e-5, key
        else:
            assert np.abs(t[:, ::5, ::7, ::9].numpy() - g[key + "_sub"]).max() <= 1e-5, key
    ref = g["losses"]; mine = plan.loss_scalars().cpu().double().numpy()
    assert np.all(np.abs(mine - ref) <= 1e-5 * np.abs(ref) + 1e-9), (mine, ref)
    names = [str(s) for s in g["grad_names"]]
    for (name, off, shape), ref_norm in zip(table, g["grad_norms"]):
        if name.endswith("k_linear.bias"):
            continue
        gr = gflat[off:off + int(np.prod(shape))]
        # two independent fp32 evaluations (reference on CPU, HIP): each carries its own sg() flips, see above
        assert abs(gr.double().norm().item() - ref_norm) <= 5e-3 * ref_norm + 1e-12, name
        if "grad/" + name in g.files:
            r = torch.from_numpy(g["grad/" + name]).reshape(-1)
            assert rel_l2(gr.cpu(), r) <= GOLDEN_GRAD_TOL.get(case, 6e-3), name


@pytest.mark.parametrize("fused_tail", [1, 0])
@pytest.mark.parametrize("n,bands,h,w", [(1, 31, 200, 264), (2, 31, 50, 38), (1, 256, 128, 128), (3, 8, 32, 64), (1, 64, 90, 70)])
def test_enhance_only_ragged_sizes(pkg, n, bands, h, w, fused_tail):
    """Enhance-only path (test/inference entry, model.py:229-234) on sizes that are not multiples of the 16-pixel
    tiles or of 8 (odd pyramid levels: the nearest up-sampling reads ceil-sized levels), and on full-size 256-band cubes.
    fused_tail = 1: feature_fusion + final_conv + S as one launch (tail_kernels.hip); 0: the three separate launches."""
    H, _ = pkg
    H.lib().ssie_debug_set_fused_tail(fused_tail)
    try:
        plan, table, flat, gflat, P = build_plan(H, n, bands, h, w, O.JYU_COEFS)
    finally:
        H.lib().ssie_debug_set_fused_tail(1)
    x = O.synthetic_patches(n, bands, h, w)
    plan.enhance_fwd(x.cuda())
    torch.cuda.synchronize()
    R, I, D, S = O.enhance_forward({k: v.double() for k, v in P.items()}, x.double())
    B = bands
    assert (plan.nchw("RL_1", 0, B).cpu().double() - R).abs().max() <= 1e-5
    assert (plan.nchw("RL_1", B, B + 1).cpu().double() - I).abs().max() <= 1e-5
    assert (plan.nchw("D", 0, 1).cpu().double() - D).abs().max() <= 1e-5
    assert (plan.nchw("S", 0, B).cpu().double() - S).abs().max() <= 1e-5
    assert O.psnr(plan.nchw("S", 0, B).cpu(), S) > 100


def test_module_api_and_adam(pkg):
    """Drop-in nn.Module: forward 4-tuple, compute_loss + backward + optimizer.step == oracle train steps."""
    H, M = pkg
    n, bands, h, w, coefs = CASES["b5_16"]
    net = M.LowLightEnhance(input_channels=bands, lr=1e-3, c_loss_reconstruction=coefs["c_rec"], c_loss_r_fidelity=coefs["c_rf"],
                            c_loss_i_smooth_low=coefs["c_il"], c_loss_i_smooth_delta=coefs["c_id"], c_loss_fourier=coefs["c_f"],
                            c_loss_spectral_cons=coefs["c_sp"], alpha_i_smooth_low=coefs["alpha_low"],
                            alpha_i_smooth_delta=coefs["alpha_delta"])
    assert list(net.state_dict().keys()) == list(O.param_shapes(bands).keys())
    P = O.closed_form_params(bands)
    net.load_state_dict(P)
    net = net.to("cuda")
    x = O.synthetic_patches(n, bands, h, w)
    xc = x.cuda()
    with torch.no_grad():
        R, I, D, S = net(xc)
        assert R.shape == (n, bands, h, w) and I.shape == (n, 1, h, w) and D.shape == (n, 1, h, w) and S.shape == (n, bands, h, w)
        Ro, Io, Do, So = O.enhance_forward(P, x)
        assert (S.cpu() - So).abs().max() <= 1e-5 and (R.cpu() - Ro).abs().max() <= 1e-5
    st = O.AdamState(P)
    from collections import OrderedDict
    Pw = OrderedDict(P)
    for step in range(3):
        net.optimizer.zero_grad()
        loss, ld = net.compute_loss(xc)
        loss.backward()
        net.optimizer.step()
        Pw, vals, grads, _ = O.train_step(Pw, x, coefs, st, lr=1e-3)
        assert abs(ld["total_loss"] - vals["total_loss"]) <= 2e-4 * abs(vals["total_loss"]), (step, ld, vals)
        assert abs(float(loss.detach()) - ld["total_loss"]) < 1e-6 * abs(ld["total_loss"]) + 1e-7
    sd = net.state_dict()
    bad = tot = 0
    for k, ref in Pw.items():
        if k.endswith("k_linear.bias"):
            continue
        d = (sd[k].cpu() - ref).abs()
        bad += int((d > 2e-5).sum()); tot += d.numel()
    assert bad <= 2e-3 * tot, (bad, tot)       # Adam's sign-like first steps amplify ~0-gradient sign flips
    # fused fast path == autograd-style path
    net2 = M.LowLightEnhance(input_channels=bands, lr=1e-3, **{k: v for k, v in dict(
        c_loss_reconstruction=coefs["c_rec"], c_loss_r_fidelity=coefs["c_rf"], c_loss_i_smooth_low=coefs["c_il"],
        c_loss_i_smooth_delta=coefs["c_id"], c_loss_fourier=coefs["c_f"], c_loss_spectral_cons=coefs["c_sp"]).items()})
    net2.load_state_dict(P); net2 = net2.to("cuda")
    for step in range(3):
        scal = net2.train_step(xc)
    torch.cuda.synchronize()
    for k, v in net2.state_dict().items():
        assert torch.equal(v, sd[k]), k
