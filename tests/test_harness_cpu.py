"""CPU: host-side harness logic: config priority, load_hsi normalisation incl. the reference's
double-normalisation quirk, crop RNG order, metrics definitions, checkpoint format."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import ssie
    ssie.load()
    from ssie_amd import harness, model
    return harness, model


def test_config_priority_cli_over_yaml_over_default(tmp_path, monkeypatch):
    import importlib.util
    spec = importlib.util.spec_from_file_location("ssie_main", os.path.join(ROOT, "main.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    cfg = tmp_path / "c.yml"
    cfg.write_text("channels: 31\nbatch_size: 2\nc_loss_fourier: 20\nsave_reflectance: true\n")
    a = m.parse_args(["--config", str(cfg), "--batch_size", "4", "--phase", "train"])
    assert a.channels == 31 and a.batch_size == 4 and a.c_loss_fourier == 20 and a.patch_size == 128
    assert a.save_reflectance is True and a.save_i_delta is False
    assert a.test_model_dir.endswith("Decomposition_" + a.timestamp) and a.model_ckpt_dir == "./checkpoint/no_name_model"
    b = m.parse_args(["--config", str(cfg), "--save_reflectance", "false"])
    assert b.save_reflectance is False
    with pytest.raises(SystemExit):
        m.parse_args(["--config", str(cfg), "--phase", "test"])


def test_presets_cover_the_reference_config_files():
    """config/presets.yml + --preset reproduce the key sets of the reference's eight config/*.yml (values checked for the keys
    that differ between them: coefficients, sensor range, batch, LR schedule, fold paths)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ssie_main", os.path.join(ROOT, "main.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    pre = os.path.join(ROOT, "config", "presets.yml")
    a = m.parse_args(["--config", pre, "--phase", "train"])
    assert a.c_loss_i_smooth_delta == 2000 and a.c_loss_fourier == 20 and a.global_min == 238.0 and a.batch_size == 2
    assert a.lr_update_factor == 0.1 and a.lr_update_period == 250
    b = m.parse_args(["--config", pre, "--preset", "indoor_li_et_al_cv4", "--phase", "train"])
    assert b.c_loss_i_smooth_delta == 20 and b.c_loss_fourier == 0.2 and b.batch_size == 1 and b.save_i_delta is True
    assert abs(b.global_max - 1.6697606) < 1e-9 and b.lr_update_factor == 1 and b.lr_update_period == 400
    assert b.train_data.endswith("train_fold_4/low") and b.label_dir.endswith("test_fold_4/high")
    c = m.parse_args(["--config", pre, "--preset", "indoor_jyu", "--batch_size", "8", "--phase", "train"])
    assert "jyu_indoor" in c.test_data and c.batch_size == 8 and c.c_loss_fourier == 20
    with pytest.raises(SystemExit):
        m.parse_args(["--config", pre, "--preset", "nope", "--phase", "train"])


def test_load_hsi_double_normalisation_and_roundtrip(pkg, tmp_path):
    harness, _ = pkg
    rng = np.random.RandomState(0)
    cube = (rng.rand(12, 10, 6) * 3000 + 200).astype("float32")
    p = str(tmp_path / "a.mat")
    harness.save_hsi(p, cube)
    raw = harness.load_hsi(p)
    assert np.array_equal(raw, cube)
    x = harness.load_hsi(p, "data", "global_normalization", 4095.0, 238.0)
    ref = (cube - 238.0) / (4095.0 - 238.0); ref[ref < 0] = 0; ref = ref.astype("float32") / ref.max()
    assert np.allclose(x, ref, atol=1e-7) and abs(x.max() - 1.0) < 1e-7           # utils.py:45-47,57
    y = harness.load_hsi(p, "data", "per_channel_normalization")
    assert y.shape == cube.shape and abs(y.max() - 1.0) < 1e-6


def test_load_hsi_matches_reference_fixtures(pkg, golden_dir, tmp_path):
    """N2 pinned: `hsi_raw.mat` was written by the reference's utils.save_hsi and `io.npz` holds the reference's
    utils.load_hsi outputs for it in every mode (tests/golden/make_golden.py, utils.py:36-109,171-178) - bit-exact."""
    import scipy.io as sio
    harness, _ = pkg
    g = np.load(os.path.join(golden_dir, "io.npz"))
    raw_path = os.path.join(golden_dir, "hsi_raw.mat")
    calls = {"none": (), "self": ("data", "self"), "global_238_4095": ("data", "global_normalization", 4095.0, 238.0),
             "global_none_4095": ("data", "global_normalization", 4095.0, None),
             "per_channel_normalization": ("data", "per_channel_normalization"),
             "per_channel_standardization": ("data", "per_channel_standardization")}
    for key, a in calls.items():
        got = harness.load_hsi(raw_path, *a)
        assert got.dtype == np.float32 and got.shape == g[key].shape, key
        assert np.array_equal(got, g[key]), (key, np.abs(got - g[key]).max())
    assert (g["global_238_4095"] == 0).sum() > 0 and g["global_238_4095"].max() == 1.0       # clamp + second normalisation present
    with pytest.raises(NotImplementedError):
        harness.load_hsi(raw_path, "data", "nope")
    # writer: same file content as the reference's writer produced (key 'data', postfix before the extension)
    p = str(tmp_path / "w.mat")
    harness.save_hsi(p, g["raw"])
    assert np.array_equal(sio.loadmat(p)["data"], sio.loadmat(raw_path)["data"])
    harness.save_hsi(p, g["raw"][:2, :2], postfix="_R_low")
    assert os.path.exists(str(tmp_path / "w_R_low.mat"))


def test_crop_draw_order_matches_reference_loop(pkg):
    harness, _ = pkg
    shapes = [(96, 80, 5), (70, 90, 5)]
    np.random.seed(41)
    got = harness.draw_crops(2, shapes, batch_id=1, batch_size=3, patch=64)
    np.random.seed(41)
    exp = []
    for i in range(3):                                        # model.py:303-308
        idx = (1 * 3 + i) % 2
        h, w, _ = shapes[idx]
        x = np.random.randint(0, h - 64); y = np.random.randint(0, w - 64); mode = np.random.randint(0, 8)
        exp.append((idx, x, y, mode))
    assert got == exp
    with pytest.raises(ValueError):
        harness.draw_crops(1, [(64, 64, 5)], 0, 1, 64)


def test_metrics_definitions(pkg):
    harness, _ = pkg
    g = torch.Generator().manual_seed(0)
    t = torch.rand(24, 20, 16, generator=g)      # spectral axis must exceed the 11-tap SSIM window (torchmetrics crops it)
    assert float(harness.psnr(t, t + 0.1, 1.0)) == pytest.approx(20.0, abs=1e-4)
    assert float(harness.ssim(t, t, 1.0)) == pytest.approx(1.0, abs=1e-9)
    assert float(harness.ssim(t, torch.rand(24, 20, 16, generator=g), 1.0)) < 0.2
    assert float(harness.sam(t, 3.0 * t)) == pytest.approx(0.0, abs=1e-6)
    a = torch.tensor([[[1.0, 0.0]]]); b = torch.tensor([[[0.0, 1.0]]])
    assert float(harness.sam(a, b)) == pytest.approx(np.pi / 2, abs=1e-9)


def test_checkpoint_format_is_torch_adam_compatible(pkg, tmp_path):
    """state dict keys = the reference's 46 keys; optimizer state loads into a plain torch.optim.Adam and back."""
    _, model = pkg
    net = model.LowLightEnhance(input_channels=5)
    sd = net.optimizer.state_dict()
    assert sd["param_groups"][0]["params"] == list(range(46)) and sd["state"] == {}
    ref_opt = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in net.parameters()], lr=1e-3)
    ref_opt.load_state_dict(sd)                                    # reference-side loader accepts it
    for p in ref_opt.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    ref_opt.step()
    path = str(tmp_path / "ck.pth")
    torch.save({"epoch": 1, "model_state_dict": net.state_dict(), "optimizer_state_dict": ref_opt.state_dict()}, path)
    ck = torch.load(path, weights_only=True)
    assert list(ck["model_state_dict"].keys()) == [n for n, _ in net.named_parameters()]
    st = ck["optimizer_state_dict"]["state"]
    assert len(st) == 46 and float(st[0]["step"]) == 1.0 and st[0]["exp_avg"].shape == net._plist[0].shape
