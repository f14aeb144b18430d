"""GPU: boundary semantics of the drop-in `LowLightEnhance` (reference: /root/reference/model.py:177-234, :236-443).

  * forward() returns tensors the caller owns (the reference's do not alias anything either, model.py:229-234)
  * the three harness methods exist ON the module with the reference's argument names (model.py:236, :343, :406), which is
    what the reference's own main.py calls (main.py:92-128)
  * freeze_decom_epochs semantics (model.py:274-288): frozen parameters do not move even with non-zero Adam moments
    (torch's Adam skips parameters whose grad is None), and un-freezing re-creates Adam + StepLR
  * bf16_inference on a band count without a bf16 list uses the fp32 path instead of failing
"""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import ssie_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import ssie
    ssie.load()
    from ssie_amd import hostlib, model
    assert hostlib.lib().ssie_device_ok() == 1
    return model


def make_net(M, bands, coefs=O.JYU_COEFS, **kw):
    net = M.LowLightEnhance(input_channels=bands, lr=1e-3, c_loss_reconstruction=coefs["c_rec"], c_loss_r_fidelity=coefs["c_rf"],
                            c_loss_i_smooth_low=coefs["c_il"], c_loss_i_smooth_delta=coefs["c_id"], c_loss_fourier=coefs["c_f"],
                            c_loss_spectral_cons=coefs["c_sp"], alpha_i_smooth_low=coefs["alpha_low"],
                            alpha_i_smooth_delta=coefs["alpha_delta"], **kw)
    net.load_state_dict(O.closed_form_params(bands))
    return net.to("cuda")


def test_graph_replay_is_bit_identical(M, monkeypatch):
    """A plan made by the module replays its train step as ONE hipGraph from the third call on (ssie_plan_set_graph; SSIE_GRAPH=0
    turns it off).  Same kernels in the same order: parameters and losses after five steps must be bit-identical to the eager
    launches; new loss coefficients rebuild the op lists and must drop the captured graph."""
    res = []
    for flag in ("0", "1"):
        monkeypatch.setenv("SSIE_DEBUG", "1")                       # development switches are honoured only with it (hostlib.py)
        monkeypatch.setenv("SSIE_GRAPH", flag)
        net = make_net(M, 31)
        x = O.synthetic_patches(2, 31, 64, 64).cuda()
        scal = [net.train_step(x).clone() for _ in range(5)]
        net.c_loss_fourier = 0.25                                   # coefficient change between steps (set_coefs -> rebuild)
        scal += [net.train_step(x).clone() for _ in range(3)]
        torch.cuda.synchronize()
        res.append((torch.stack(scal), net._flat.clone()))
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1], res[1][1])


def test_forward_returns_owned_tensors(M):
    net = make_net(M, 5)
    x1 = O.synthetic_patches(1, 5, 16, 16, seed=1).cuda(); x2 = O.synthetic_patches(1, 5, 16, 16, seed=2).cuda()
    with torch.no_grad():
        S1 = net(x1)[3]
        keep = S1.clone()
        S2 = net(x2)[3]
    assert S1.data_ptr() != S2.data_ptr()
    assert torch.equal(S1, keep) and not torch.equal(S1, S2)
    with pytest.warns(UserWarning, match="autograd graph"):
        out = net(x1)
    assert not any(t.requires_grad for t in out)


def test_bf16_flag_runs_bf16_for_any_band_count(M):
    """round 3 built the bf16 list only when B and B + 1 padded to multiples of 8 and `bf16_inference = True` silently ran fp32
    otherwise (9 bands, and 64 / 256 - everything the reference ships); now every band count takes it (model.py:229-234)"""
    net9 = make_net(M, 9)                     # 9 bands: fp32 pixel strides 12 / 12, bf16 strides 16 / 16
    x = O.synthetic_patches(1, 9, 16, 16).cuda()
    with torch.no_grad():
        a = net9(x)[3]
        net9.bf16_inference = True
        b = net9(x)[3]
    assert not torch.equal(a, b)              # the bf16 list ran ...
    assert (a - b).abs().max() <= 5e-3        # ... and stays inside the bf16 bar of tests/test_bf16_infer_gpu.py


def test_freeze_semantics(M):
    bands, n, hw = 5, 1, 16
    coefs = O.DEFAULT_COEFS
    net = make_net(M, bands, coefs, lr_update_factor=0.5, lr_update_period=3)
    x = O.synthetic_patches(n, bands, hw, hw)
    xc = x.cuda()
    # non-zero optimiser state, as after load_checkpoint(): two ordinary steps first
    net.train_step(xc); net.train_step(xc)
    torch.cuda.synchronize()
    P0 = OrderedDict((k, v.detach().cpu().clone()) for k, v in net.state_dict().items())
    m0 = net.optimizer.exp_avg.cpu().clone(); v0 = net.optimizer.exp_avg_sq.cpu().clone()
    assert m0[:net._illum_off].abs().max() > 0
    net.set_decomposition_frozen(True)
    net.train_step(xc)
    torch.cuda.synchronize()
    off = net._illum_off
    for k, v in net.state_dict().items():
        if k.startswith("decomposition_net."):
            assert torch.equal(v.cpu(), P0[k]), k                      # bit-unchanged despite non-zero moments
    assert torch.equal(net.optimizer.exp_avg.cpu()[:off], m0[:off]) and torch.equal(net.optimizer.exp_avg_sq.cpu()[:off], v0[:off])
    # illumination net: one Adam step (t = 3) from the loaded moments, gradients of the oracle
    _, grads, _ = O.loss_and_grads(P0, x, coefs)
    for (name, o, shape), p in zip(net._table, net._plist):
        if not name.startswith("illum_adjust_net.") or name.endswith("k_linear.bias"):
            continue
        k = p.numel()
        g = grads[name].reshape(-1).double()
        m = 0.9 * m0[o:o + k].double() + 0.1 * g
        v = 0.999 * v0[o:o + k].double() + 0.001 * g * g
        ref = P0[name].reshape(-1).double() - (1e-3 / (1 - 0.9 ** 3)) * m / (v.sqrt() / np.sqrt(1 - 0.999 ** 3) + 1e-8)
        d = (p.detach().cpu().reshape(-1).double() - ref).abs()
        assert (d > 2e-5).double().mean().item() <= 5e-3, name        # Adam's sign-like steps amplify ~0-gradient flips
    # autograd-style path while frozen: frozen parameters come back with grad None and stay put
    net.optimizer.zero_grad()
    loss, _ = net.compute_loss(xc); loss.backward()
    assert all(p.grad is None for p in net.decomposition_net.parameters())
    before = net.state_dict()["decomposition_net.conv1.0.weight"].clone()
    net.optimizer.step()
    assert torch.equal(net.state_dict()["decomposition_net.conv1.0.weight"], before)
    # un-freeze: Adam and StepLR start over from the CURRENT lr (model.py:284-286)
    for _ in range(4):
        net.scheduler.step()                                            # lr has decayed once by now (period 3)
    lr_now = net.optimizer.param_groups[0]["lr"]
    assert lr_now == pytest.approx(5e-4)
    net.set_decomposition_frozen(False)
    assert net.optimizer.step_count == 0 and float(net.optimizer.exp_avg.abs().max()) == 0.0
    assert all(p.requires_grad for p in net.parameters())
    assert net.optimizer.param_groups[0]["lr"] == pytest.approx(lr_now)
    net.train_step(xc)
    for _ in range(2):
        net.scheduler.step()
    assert net.optimizer.param_groups[0]["lr"] == pytest.approx(lr_now)        # period restarted: no decay after 2 epochs
    net.scheduler.step()
    assert net.optimizer.param_groups[0]["lr"] == pytest.approx(lr_now * 0.5)


def _cube(seed, h, w, c):
    x = O.synthetic_patches(1, c, h, w, seed=seed)[0].permute(1, 2, 0).numpy()
    return (238.0 + x / 0.3 * (4095.0 - 238.0) * 0.6).astype("float32")


def test_reference_method_names(M, tmp_path, monkeypatch):
    """train_model / evaluate_model / test_model called exactly the way /root/reference/main.py:92-128 calls them."""
    import scipy.io as sio
    import ssie
    ssie.load()
    from ssie_amd import harness
    bands = 8                       # the reference's SSIM treats the band axis as image width: needs more than 5 bands (11-tap window)
    for split in ("train", "eval", "test", "high"):
        os.makedirs(tmp_path / split)
    sio.savemat(str(tmp_path / "train" / "a.mat"), {"data": _cube(1, 40, 48, bands)})
    sio.savemat(str(tmp_path / "train" / "b.mat"), {"data": _cube(2, 44, 40, bands)})
    sio.savemat(str(tmp_path / "eval" / "e.mat"), {"data": _cube(3, 32, 32, bands)})
    sio.savemat(str(tmp_path / "test" / "t.mat"), {"data": _cube(4, 32, 48, bands)})
    sio.savemat(str(tmp_path / "high" / "t.mat"), {"data": _cube(4, 32, 48, bands) * 1.5})
    monkeypatch.chdir(tmp_path)
    np.random.seed(41)
    net = make_net(M, bands, O.DEFAULT_COEFS, time_stamp="ts0", global_min=238.0, global_max=4095.0, save_reflectance=True)
    P0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net.train_model(train_data_path=str(tmp_path / "train"), eval_data_path=str(tmp_path / "eval"), batch_size=2,
                    patch_size=32, num_epochs=2, start_lr=1e-3, ckpt_dir=str(tmp_path / "ck"),
                    eval_result_dir=str(tmp_path / "evalres"), eval_every_epoch=1, label_dir=str(tmp_path / "high"),
                    plot_every_epoch=1)
    model_dir = tmp_path / "ck" / "Decomposition_ts0"
    assert (model_dir / "model_epoch_latest.pth").exists() and (model_dir / "model_epoch_2.pth").exists()
    assert (tmp_path / "evalres" / "epoch_2" / "e.mat").exists()
    assert len(net.all_epoch_losses["total_loss"]) == 2
    assert any(not torch.equal(v, P0[k]) for k, v in net.state_dict().items())
    names = sorted(str(p) for p in (tmp_path / "test").glob("*.*"))
    data = [harness.load_hsi(f, "data", "global_normalization", 4095.0, 238.0) for f in names]
    net.test_model(model_dir=str(model_dir), test_low_data=data, test_low_data_names=names, save_dir=str(tmp_path / "out"),
                   save_reflectance=True, save_illumination=False, save_i_delta=False)
    S = sio.loadmat(str(tmp_path / "out" / "t.mat"))["data"]
    assert S.shape == (32, 48, bands) and np.isfinite(S).all()
    assert (tmp_path / "out" / "artifacts" / "t_R_low.mat").exists()
    # evaluate_model positional, as model.py:328 calls it
    net.evaluate_model(data, names, str(tmp_path / "evalres2"), 7, str(tmp_path / "high"))
    assert (tmp_path / "evalres2" / "epoch_7" / "t.mat").exists() and 7 in net.eval_metrics


def test_lagged_loss_readback(M):
    """harness.LaggedScalars (async loss logging): the values, order and count of the blocking read-back
    (model.py:566-574), one step late, with no slot overwritten before it was consumed; and the pinned crop-record staging of
    ssie_assemble_batch gives the same batch as the pageable path."""
    import ssie
    ssie.load()
    from ssie_amd import harness, hostlib
    lag = harness.LaggedScalars(2)
    dev = torch.device("cuda", 0)
    scal = torch.zeros(16, device=dev)                      # ONE device buffer rewritten every step, like the plan's
    seen = []
    for i in range(7):
        scal[:7] = torch.arange(7, device=dev, dtype=torch.float32) + 100.0 * i
        due = lag.push(scal, ("e", i))
        assert len(due) == (0 if i == 0 else 1)             # exactly the previous step becomes due
        seen += due
    seen += lag.drain()
    assert [t for t, _ in seen] == [("e", i) for i in range(7)]
    for i, (_, v) in enumerate(seen):
        assert v.shape == (7,) and np.array_equal(v, np.arange(7, dtype=np.float32) + 100.0 * i)
    assert lag.drain() == []
    cube = torch.rand(40, 48, 8, device=dev)
    crops = [(0, 3, 5, 2), (0, 7, 1, 7), (0, 0, 15, 0)]
    a = hostlib.assemble_batch([cube], crops, 32, 8)
    st = torch.empty(len(crops) * harness.ctypes_sizeof_crop(), dtype=torch.uint8, pin_memory=True)
    b = hostlib.assemble_batch([cube], crops, 32, 8, staging=st)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    with pytest.raises(hostlib.SsieError):
        hostlib.assemble_batch([cube], crops, 32, 8, staging=torch.empty(4, dtype=torch.uint8, pin_memory=True))
