"""GPU: the `main.py --config ... ` train_and_test entry end to end (BASELINE config 1: config_outdoor_jyu keys,
channels=31, batch 1, one 64x64 patch per step, 1 epoch) on synthetic .mat cubes, with the first train step checked
against the CPU oracle on the identical crop (same seeds, reference RNG order)."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import ssie_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cube(seed, h=96, w=80, c=31):
    x = O.synthetic_patches(1, c, h, w, seed=seed)[0].permute(1, 2, 0).numpy()      # (H, W, C) in [0, 0.3]
    return (238.0 + x / 0.3 * (4095.0 - 238.0) * 0.6).astype("float32")             # raw sensor-like range


def test_train_and_test_entry(tmp_path, monkeypatch):
    import scipy.io as sio
    import ssie
    ssie.load()
    from ssie_amd import harness, model
    spec = importlib.util.spec_from_file_location("ssie_main", os.path.join(ROOT, "main.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    for split in ("train", "eval", "test", "high"):
        os.makedirs(tmp_path / split)
    sio.savemat(str(tmp_path / "train" / "a.mat"), {"data": _cube(1)})
    sio.savemat(str(tmp_path / "eval" / "e.mat"), {"data": _cube(2, 64, 64)})
    sio.savemat(str(tmp_path / "test" / "t.mat"), {"data": _cube(3, 64, 80)})
    sio.savemat(str(tmp_path / "high" / "t.mat"), {"data": _cube(3, 64, 80) * 1.5})
    monkeypatch.chdir(tmp_path)
    argv = ["--config", os.path.join(ROOT, "config", "config_outdoor_jyu.yml"), "--channels", "31", "--batch_size", "1",
            "--patch_size", "64", "--epoch", "1", "--eval_every_epoch", "1", "--model_name", "t",
            "--train_data", str(tmp_path / "train"), "--eval_data", str(tmp_path / "eval"),
            "--test_data", str(tmp_path / "test"), "--label_dir", str(tmp_path / "high")]
    args = m.parse_args(argv)
    m.main(args)
    ck = os.path.join("checkpoint", "t", "Decomposition_" + args.timestamp, "model_epoch_latest.pth")
    assert os.path.exists(ck)
    out = os.path.join(args.test_result_dir, "t.mat")
    assert os.path.exists(out)
    S = sio.loadmat(out)["data"]
    assert S.shape == (64, 80, 31) and np.isfinite(S).all()
    assert os.path.exists(os.path.join(args.test_result_dir, "artifacts", "t_R_low.mat"))
    saved = torch.load(ck, weights_only=True)
    assert list(saved["model_state_dict"].keys()) == list(O.param_shapes(31).keys())
    assert len(saved["optimizer_state_dict"]["state"]) == 46

    # first train step vs the oracle on the identical crop + identical initial weights
    import random
    random.seed(41); np.random.seed(41); torch.manual_seed(41)
    ref_net = model.LowLightEnhance(input_channels=31, lr=1e-3)              # same default init stream as main.build_model
    P = {k: v.detach().clone() for k, v in ref_net.state_dict().items()}
    cube = harness.load_hsi(str(tmp_path / "train" / "a.mat"), "data", "global_normalization", 4095.0, 238.0)
    (idx, x0, y0, mode), = harness.draw_crops(1, [cube.shape], 0, 1, 64)
    patch = cube[x0:x0 + 64, y0:y0 + 64, :]
    patch = [lambda a: a, np.flipud, np.rot90, lambda a: np.flipud(np.rot90(a)), lambda a: np.rot90(a, 2),
             lambda a: np.flipud(np.rot90(a, 2)), lambda a: np.rot90(a, 3), lambda a: np.flipud(np.rot90(a, 3))][mode](patch)
    x = torch.from_numpy(np.ascontiguousarray(patch)).unsqueeze(0).permute(0, 3, 1, 2)
    _, vals, _ = O.compute_loss(P, x, O.JYU_COEFS)
    ck_first = torch.load(os.path.join("checkpoint", "t", "Decomposition_" + args.timestamp, "model_epoch_1.pth"), weights_only=True)
    assert ck_first["epoch"] == 1
    # the harness stored the epoch-mean losses on the live model; re-run the first step to compare numerically
    net = m.build_model(m.parse_args(argv), torch.device("cuda"))
    net.load_state_dict(P)
    scal = net.train_step(x.cuda()).cpu().double().numpy()
    ref = np.array([vals[k] for k in O.LOSS_KEYS])
    assert np.all(np.abs(scal - ref) <= 2e-5 * np.abs(ref) + 1e-9), (scal, ref)


def test_test_entry_bf16_inference_matches_fp32(tmp_path, monkeypatch):
    """`--bf16_inference 1`: the test/eval forward runs the bf16 mixed-precision path; the written cubes must agree with the
    fp32 run's to bf16 accuracy (PSNR >= 55 dB on the [0, 1]-normalised enhanced cube)."""
    import scipy.io as sio
    import ssie
    ssie.load()
    from ssie_amd import harness
    spec = importlib.util.spec_from_file_location("ssie_main", os.path.join(ROOT, "main.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    os.makedirs(tmp_path / "test")
    sio.savemat(str(tmp_path / "test" / "t.mat"), {"data": _cube(5, 72, 88)})
    monkeypatch.chdir(tmp_path)
    outs = {}
    for flag in ("0", "1"):
        argv = ["--config", os.path.join(ROOT, "config", "config_outdoor_jyu.yml"), "--channels", "31", "--model_name", "b" + flag,
                "--test_data", str(tmp_path / "test"), "--bf16_inference", flag]
        args = m.parse_args(argv)
        torch.manual_seed(7)
        net = m.build_model(args, torch.device("cuda"))
        assert net.bf16_inference == (flag == "1")
        cube = harness.load_hsi(str(tmp_path / "test" / "t.mat"), "data", "global_normalization", 4095.0, 238.0)
        R, I, D, S = harness._enhance_whole(net, cube)              # what test_model runs per file (model.py:418-421)
        outs[flag] = np.asarray(S, dtype=np.float64)
    a, b = outs["0"], outs["1"]
    assert a.shape == b.shape == (72, 88, 31)
    mse = np.mean((a - b) ** 2)                 # cubes are normalised to [0, 1]
    assert 10 * np.log10(1.0 / max(mse, 1e-30)) >= 55.0
    assert not np.array_equal(a, b)            # the bf16 path really ran
