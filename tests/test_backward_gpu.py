"""The hand-derived backward chain of the plan executor, pinned WITHOUT the sg()-flip noise of the L1 losses
(VERDICT r1 weak 1 / ADVICE r1): /root/reference/model.py:315 (`loss.backward()`).

1. `test_backward_chain_injected`: the oracle's fp64 direct cotangents dL/d{R, I, D, S, E} (oracle/loss_cotangents.py) are
   written into the plan buffers and ONLY the backward schedule runs (`ssie_plan_backward_from_cotangents`,
   include/ssie_debug.h).  The chain is then linear in fixed cotangents, so all 45 parameter gradients and the
   intermediate data gradients are held to a FIXED rel-L2 (no tolerance derived from any noise estimate).  The oracle side
   is autograd of the matching linear surrogate (`O.grads_from_cotangents`, fp64).
2. `test_one_hot_coefficients`: full loss+backward with one loss coefficient at a time (`ssie_plan_set_coefs`): the terms
   whose sg() arguments sit away from 0 (reconstruction, spectral TV) hold the fixed 1e-3 end to end; the other four get a
   measured, documented bound.
3. `test_timed_configuration_n32`: the configuration bench.py times (N = 32, 128 x 128 x 31: 16x32 tiles, 512 wgrad slices,
   paired [pass 1; pass 2] weight gradients, slab reductions on the side stream) - seven losses, S, gradient norms, and
   the injected-cotangent chain at the same fixed tolerance.
"""
import numpy as np
import pytest
import torch

from oracle import loss_cotangents as LC
from oracle import ssie_oracle as O

pytestmark = pytest.mark.gpu

def rel_l2(a, b):
    return (a.double() - b.double()).norm().item() / max(b.double().norm().item(), 1e-300)


FIXED_TOL = 2e-5          # 50x tighter than SURVEY §8(c)'s 1e-3; measured 3.7e-7 ... 1.7e-6 on MI355X (fp32 accumulation order only)
# The attention's q/k gradients are a near-total cancellation dS = P (dP - delta) for these fixtures (|dq|, |dk| ~ 1e-4 ... 1e-6
# of |dv|, shrinking with the token count): fp32 rounding of the O(1) summands is amplified accordingly (measured 3e-6 ... 6e-4
# at 4 ... 256 tokens, 8e-3 at 1024).  They pass at SURVEY §8(c)'s 1e-3 OR when the ABSOLUTE error is below 2e-5 of the
# v_linear weight gradient's norm - the same contraction over the same tokens without the cancellation, i.e. the fixed
# 2e-5 bar applied to the magnitude the rounding actually scales with.  The attention kernel itself is pinned to 2e-5 on
# well-conditioned inputs in tests/test_attention_gpu.py.
QK_TOL = 1e-3


def qk_ok(name, got, ref, grads):
    if rel_l2(got, ref) <= QK_TOL:
        return True
    yard = grads["illum_adjust_net.attn.v_linear.weight"].double().norm().item()
    return (got.double() - ref.double()).norm().item() <= FIXED_TOL * yard
RELU_BUFFERS = ["c0_1", "c1_1", "c2_1", "c3_1", "dc_1", "c5_1", "c0_2", "c1_2", "c2_2", "c3_2", "dc_2", "c5_2",
                "a1", "a2", "a3", "f1", "u1", "u2", "u3"]

CASES = {
    "b5_16": (1, 5, 16, 16, O.DEFAULT_COEFS),
    "b31_32": (2, 31, 32, 32, O.JYU_COEFS),
    "b31_64": (2, 31, 64, 64, O.JYU_COEFS),
    "b64_64": (2, 64, 64, 64, O.JYU_COEFS),           # the reference's own band count (model.py:178; every shipped config)
    "b64_128": (2, 64, 128, 128, O.JYU_COEFS),        # config_outdoor_jyu.yml:7,11-12: batch 2 of 128 x 128 x 64, real launch heuristics
    "b8_32x64": (3, 8, 32, 64, O.JYU_COEFS),
    "b8_24x40": (2, 8, 24, 40, O.JYU_COEFS),
    "b8_20x28": (2, 8, 20, 28, O.JYU_COEFS),          # pyramid 20x28 -> 10x14 -> 5x7 -> 3x4: the odd levels take the general up-sampling adjoint
    "b31_128": (1, 31, 128, 128, O.JYU_COEFS),
    "b256_64": (1, 256, 64, 64, O.JYU_COEFS),
    "b5_256": (1, 5, 256, 256, O.JYU_COEFS),
}


@pytest.fixture(scope="module")
def pkg():
    import ssie
    ssie.load()
    from ssie_amd import hostlib
    assert hostlib.lib().ssie_device_ok() == 1
    return hostlib


def build_plan(H, n, bands, h, w, coefs):
    table, total = H.param_table(bands)
    P = O.closed_form_params(bands)
    flat = torch.zeros(total, device="cuda")
    for name, off, shape in table:
        flat[off:off + P[name].numel()] = P[name].reshape(-1).cuda()
    gflat = torch.zeros_like(flat)
    return H.Plan(n, bands, h, w, coefs, flat, gflat), table, flat, gflat, P




def direct_cotangents64(P, x, coefs):
    """fp64 forward + hand-derived direct cotangents, rounded to fp32 (the values both sides then treat as exact)"""
    P64 = {k: v.double() for k, v in P.items()}
    with torch.no_grad():
        R, I, D, S = O.enhance_forward(P64, x.double())
        E, _ = O.decomposition(P64, S)
    cot = LC.direct_cotangents(x.double(), R, I, D, S, E, coefs)
    return {k: v.float() for k, v in cot.items()}, (R, I, D, S, E)


def relu_masks(plan):
    """the ReLU decisions the HIP forward took (stored activations > 0), handed to the fp64 oracle so that both sides
    differentiate the SAME piecewise-linear function (a pre-activation within 1e-7 of 0 may round to the other side)"""
    return {name: (plan.nchw(name) > 0).cpu() for name in RELU_BUFFERS}


def inject_and_run(plan, bands, cot, E):
    B = bands
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().cuda()
    plan.buffer("gRL")[..., :B].copy_(nhwc(cot["gR"])); plan.buffer("gRL")[..., B:B + 1].copy_(nhwc(cot["gI"]))
    plan.buffer("gD")[..., :1].copy_(nhwc(cot["gD"]))
    plan.buffer("gS")[..., :B].copy_(nhwc(cot["gS"]))
    g8 = (cot["gE"].double() * E * (1.0 - E)).float()             # through pass 2's sigmoid (model.py:68)
    plan.buffer("G8_2")[..., :B].copy_(nhwc(g8)); plan.buffer("G8_2")[..., B:B + 1].zero_()
    plan.backward_from_cotangents()
    torch.cuda.synchronize()


def check_chain(H, case, n, bands, h, w, coefs, forced=None):
    plan, table, flat, gflat, P = build_plan(H, n, bands, h, w, coefs)
    x = O.synthetic_patches(n, bands, h, w)
    plan.loss_fwd_bwd(x.cuda(), backward=True)                    # fills every activation the backward reads
    cot, (R, I, D, S, E) = direct_cotangents64(P, x, coefs)
    inject_and_run(plan, bands, cot, E)
    tr = {}
    masks = relu_masks(plan)
    grads, _ = O.grads_from_cotangents({k: v.double() for k, v in P.items()}, x.double(), cot, tr, masks)
    report, bad = [], []
    inter = [("G7", "c7_1", 64), ("Gsh", "sh_1", 64), ("Gf", "f", 64), ("gd3", "a0", 64), ("G7_2", "c7_2", 64), ("Gsh_2", "sh_2", 64)]
    for buf, key, c in inter:
        e = rel_l2(plan.nchw(buf, 0, c).cpu(), tr[key].grad)
        report.append(f"d/d {key:6s} [{buf}] rel {e:.2e}")
        if not e <= FIXED_TOL:
            bad.append(report[-1])
    worst = 0.0
    for name, off, shape in table:
        if name.endswith("k_linear.bias"):
            continue                                              # analytically zero (softmax shift invariance)
        g = gflat[off:off + int(np.prod(shape))].view(shape).cpu()
        e = rel_l2(g, grads[name])
        qk = ".q_linear." in name or ".k_linear." in name
        report.append(f"grad {name:48s} rel {e:.2e} (tol {'q/k rule' if qk else FIXED_TOL})")
        if not qk:
            worst = max(worst, e)
        if not (qk_ok(name, g, grads[name], grads) if qk else e <= FIXED_TOL):
            bad.append(report[-1])
    print(f"[{case}{' ' + forced if forced else ''}] worst non-q/k gradient rel-L2 = {worst:.2e}")
    print("\n".join(report))
    assert not bad, "backward-chain parity failures:\n" + "\n".join(bad)


@pytest.fixture(params=[None, "dma_kernels", "winograd", "winograd4"])
def forced_kernels(pkg, request):
    """None = launch heuristics (8x16 tiles at these small batches); "dma_kernels" = the kernels the bench-size layers
    run (16x16 / 16x32 DMA tiles, split 32-channel workgroups) forced through the same cases (include/ssie_debug.h);
    "winograd" = the Winograd F(2x2,3x3) kernel for every stride-1 3x3 forward / data-gradient launch."""
    L = pkg.lib()
    if request.param in ("winograd", "winograd4"):  # every stride-1 3x3 forward / data-gradient launch on conv_wino_kernel (F(2x2,3x3)) -
        # "winograd4": on conv_wino4_kernel (F(4x4,3x3)) wherever it is eligible (sources at their own resolution, >= 48 columns)
        L.ssie_debug_set_wino4_min_tiles(1 if request.param == "winograd4" else 1 << 30)
        L.ssie_debug_set_wino_min_tiles(1)
        L.ssie_debug_set_wgrad_wino_min_tiles(1)
        L.ssie_debug_set_tconv_min_tiles(1)       # and the one-launch transposed convolution (conv_tconv.hip)
    elif request.param:
        L.ssie_debug_set_wino4_min_tiles(1 << 30)
        L.ssie_debug_set_wino_min_tiles(1 << 30)
        L.ssie_debug_set_wgrad_wino_min_tiles(1 << 30)
        L.ssie_debug_set_fprop_min_tiles16(0)
        L.ssie_debug_set_fprop_wide_min_tiles(1)
        L.ssie_debug_set_fprop_v2_split_min_tiles(1)
        L.ssie_debug_set_skinny_final(0)          # and final_conv (64 -> 1) on the MFMA tiles instead of the VALU kernels
        L.ssie_debug_set_spectral9(0)             # and the 9 x 9 convolution on the direct MFMA kernels instead of the frequency domain
    yield request.param
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
def test_backward_chain_injected(pkg, case, forced_kernels):
    if forced_kernels and (case in ("b5_16", "b5_256") or (case == "b64_128" and forced_kernels != "winograd4") or
                           (case == "b256_64" and forced_kernels not in ("winograd", "winograd4"))):
        pytest.skip("forced-kernel variant runs on the mid-size cases only (time); 256 bands: the Winograd / tconv kernels only")
    n, bands, h, w, coefs = CASES[case]
    check_chain(pkg, case, n, bands, h, w, coefs, forced_kernels)


def test_backward_chain_256_bands_bench_kernels(pkg):
    """BASELINE configs[2] (128 x 128 x 256) with the REAL launch heuristics at a batch that keeps >= 256 tiles per layer
    (N = 8: 512 16x32 tiles at full resolution), i.e. the kernels the N = 32 plan selects: conv_wino / conv_wgrad_wino with
    256 -> 32, 64 -> 257 (ragged last output-channel block) and 257 -> 64 operands, the one-launch transposed convolution and
    the frequency-domain 9 x 9 at 256 input channels.  Same fixed 2e-5 as every other chain case (model.py:315)."""
    check_chain(pkg, "b256_128_n8", 8, 256, 128, 128, O.JYU_COEFS)


ONE_HOT = ["c_rec", "c_rf", "c_il", "c_id", "c_f", "c_sp"]
# measured on MI355X (tests print the worst value): reconstruction 7e-8 ... 1.2e-4 and spectral TV 1.5e-7 ... 1.7e-7 (their sg()
# arguments sit away from 0: at most a single sign / ReLU decision differs between two valid fp32 forwards, e.g. the direct and
# the frequency-domain 9 x 9 convolution) are held to the fixed 1e-3 of SURVEY 8(c); R fidelity 2e-4 ... 1.0e-3, I_low smoothness 6e-5 ... 1.1e-3, Fourier 8e-6 ...
# 8.8e-3 (32 x 32 planes: one flipped sg(|F(S)| - |F(x)|) bin is 1/1024 of a plane) and I_delta smoothness 1e-5 ... 6.2e-3 (I_delta is very smooth here: median |dx D| = 2e-5, so the HIP forward's 1e-7
# differences flip a fraction of a percent of sg(dx D)) get ~3-5x the worst value seen.  The arithmetic behind those four is
# pinned elementwise in tests/test_loss_op_gpu.py and the chain behind them in test_backward_chain_injected.
ONE_HOT_TOL = {"c_rec": 1e-3, "c_sp": 1e-3, "c_rf": 5e-3, "c_il": 5e-3, "c_id": 2e-2, "c_f": 2e-2}


@pytest.mark.parametrize("case", ["b31_32", "b31_64", "b8_24x40", "b64_64"])
@pytest.mark.parametrize("term", ONE_HOT)
def test_one_hot_coefficients(pkg, case, term):
    H = pkg
    n, bands, h, w, coefs = CASES[case]
    plan, table, flat, gflat, P = build_plan(H, n, bands, h, w, coefs)
    one = dict(coefs, **{k: 0.0 for k in ONE_HOT}); one[term] = coefs[term]
    plan.set_coefs(one)
    x = O.synthetic_patches(n, bands, h, w)
    plan.loss_fwd_bwd(x.cuda(), backward=True)
    torch.cuda.synchronize()
    vals, grads, _ = O.loss_and_grads({k: v.double() for k, v in P.items()}, x.double(), one)
    got = plan.loss_scalars().cpu().double().tolist()
    assert abs(got[0] - vals["total_loss"]) <= 5e-5 * abs(vals["total_loss"]), (got[0], vals["total_loss"])
    worst, worst_name = 0.0, ""
    for name, off, shape in table:
        if name.endswith("k_linear.bias") or ".q_linear." in name or ".k_linear." in name:
            continue
        if term == "c_id" and name.endswith("final_conv.bias"):
            continue             # = sum over pixels of the adjoint of a difference: telescopes to 0 analytically
        ref = grads[name]
        if ref.norm().item() == 0.0:
            assert gflat[off:off + int(np.prod(shape))].abs().max().item() == 0.0, name
            continue
        e = rel_l2(gflat[off:off + int(np.prod(shape))].view(shape).cpu(), ref)
        if e > worst:
            worst, worst_name = e, name
    print(f"[one-hot {term} {case}] worst gradient rel-L2 = {worst:.2e} ({worst_name})")
    assert worst <= ONE_HOT_TOL[term], (term, worst, worst_name)


def test_timed_configuration_n32(pkg):
    """N = 32, 128 x 128 x 31, JYU coefficients = BASELINE.json configs[1], the plan bench.py times."""
    H = pkg
    n, bands, h, w, coefs = 32, 31, 128, 128, O.JYU_COEFS
    plan, table, flat, gflat, P = build_plan(H, n, bands, h, w, coefs)
    x = O.synthetic_patches(n, bands, h, w)
    plan.loss_fwd_bwd(x.cuda(), backward=True)
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, torch.get_num_threads()))
    vals, grads, outs = O.loss_and_grads(P, x, coefs)              # the reference arithmetic: plain PyTorch fp32
    got = plan.loss_scalars().cpu().double().tolist()
    for k, g in zip(O.LOSS_KEYS, got):
        assert abs(g - vals[k]) <= 1e-5 * abs(vals[k]), (k, g, vals[k])
    S = plan.nchw("S", 0, bands).cpu()
    assert (S - outs[3]).abs().max().item() <= 1e-5
    assert (plan.nchw("RL_1", 0, bands).cpu() - outs[0]).abs().max().item() <= 1e-5
    assert (plan.nchw("RL_2", 0, bands).cpu() - outs[4]).abs().max().item() <= 1e-5
    assert O.psnr(S, outs[3]) > 100.0
    for name, off, shape in table:
        if name.endswith("k_linear.bias"):
            continue
        gn = gflat[off:off + int(np.prod(shape))].double().norm().item()
        rn = grads[name].double().norm().item()
        # two independent fp32 evaluations, each with its own sg() flips: norms agree to 5e-3 (as vs the reference fixtures)
        tol = 2e-2 if (".q_linear." in name or ".k_linear." in name) else 5e-3
        assert abs(gn - rn) <= tol * rn + 1e-12, (name, gn, rn)
    del grads, outs
    # and the backward chain of THIS plan at the fixed tolerance (fp64 oracle of the linear surrogate)
    cot, (R, I, D, S64, E) = direct_cotangents64(P, x, coefs)
    inject_and_run(plan, bands, cot, E)
    g64, _ = O.grads_from_cotangents({k: v.double() for k, v in P.items()}, x.double(), cot, None, relu_masks(plan))
    worst = 0.0
    for name, off, shape in table:
        if name.endswith("k_linear.bias"):
            continue
        g = gflat[off:off + int(np.prod(shape))].view(shape).cpu()
        e = rel_l2(g, g64[name])
        if ".q_linear." in name or ".k_linear." in name:
            assert qk_ok(name, g, g64[name], g64), (name, e)
            continue
        worst = max(worst, e)
        assert e <= FIXED_TOL, (name, e)
    print(f"[N=32 128x128x31] worst non-q/k gradient rel-L2 on injected cotangents = {worst:.2e}")
