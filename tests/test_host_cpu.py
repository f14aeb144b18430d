"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol declared in
include/ssie_hip.h, the C-side parameter table equals the reference's state-dict layout, the host
replica of the Fourier mask equals the reference's, and the drop-in module has the reference's keys.
No kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import ssie_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def pkg():
    import ssie
    ssie.load()
    from ssie_amd import hostlib, model
    return hostlib, model


def test_library_exports_every_declared_symbol(pkg):
    H, _ = pkg
    lib = H.lib()
    hdr = open(os.path.join(ROOT, "include", "ssie_hip.h")).read() + open(os.path.join(ROOT, "include", "ssie_debug.h")).read()
    names = sorted(set(re.findall(r"\b(ssie_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 40 and "ssie_selfsup_loss_fwd_bwd" in names and "ssie_plan_backward_from_cotangents" in names
    for n in names:
        assert hasattr(lib, n), n
    assert b"gfx950" in lib.ssie_version()


@pytest.mark.parametrize("bands", [5, 31, 64, 256])
def test_param_table_matches_reference_state_dict(pkg, bands):
    H, _ = pkg
    table, total = H.param_table(bands)
    spec = O.param_shapes(bands)
    assert [t[0] for t in table] == list(spec.keys())
    for (name, off, shape), ref in zip(table, spec.values()):
        assert tuple(shape) == tuple(ref), name
        assert off % 4 == 0
    assert total >= sum(int(np.prod(s)) for s in spec.values())


def test_fourier_mask_host_replica(pkg, golden_dir):
    H, _ = pkg
    g = np.load(os.path.join(golden_dir, "aux.npz"))
    for hw in (16, 64, 128):
        assert np.array_equal(np.packbits(H.fourier_mask(hw, hw)), g["mask%d" % hw])
    assert np.array_equal(np.packbits(H.fourier_mask(32, 48)), g["mask32x48"])
    for (h, w) in ((8, 8), (256, 128), (1024, 1024)):
        assert np.array_equal(H.fourier_mask(h, w), O.fourier_mask(h, w).numpy().astype(np.uint8)), (h, w)


def test_plan_shapes_and_errors(pkg):
    H, _ = pkg
    L = H._proto()
    cf = (ctypes.c_float * 8)(*[1.0] * 8)
    assert not L.ssie_plan_create(1, 31, 129, 128, cf)        # odd H is rejected (model.py:59 would fail too)
    assert not L.ssie_plan_create(0, 31, 128, 128, cf)
    h = L.ssie_plan_create(2, 31, 64, 64, cf)
    assert h
    assert L.ssie_plan_workspace_bytes(h) > 0
    off = ctypes.c_size_t(); d = (ctypes.c_int * 5)()
    assert L.ssie_plan_buffer(h, b"RL_1", ctypes.byref(off), d) == 0 and list(d) == [2, 64, 64, 32, 32]
    assert L.ssie_plan_buffer(h, b"a3", ctypes.byref(off), d) == 0 and list(d)[:4] == [2, 8, 8, 64]
    assert L.ssie_plan_buffer(h, b"nope", ctypes.byref(off), d) != 0
    strides = (ctypes.c_long * 4)(1, 1, 1, 1)
    assert L.ssie_plan_enhance_fwd(h, 1, strides, None) != 0     # not bound -> error, nothing launched
    L.ssie_plan_destroy(h)


def test_module_surface_matches_reference(pkg):
    H, M = pkg
    net = M.LowLightEnhance(input_channels=31, lr=1e-3, lr_update_factor=0.1, lr_update_period=250, c_loss_fourier=20)
    assert list(net.state_dict().keys()) == list(O.param_shapes(31).keys())
    assert sum(p.numel() for p in net.parameters()) == 923297
    assert net.adaptive_lr and hasattr(net, "scheduler") and hasattr(net, "optimizer")
    assert hasattr(net, "decomposition_net") and hasattr(net, "illum_adjust_net") and net.freeze_decom_epochs == 0
    P = O.closed_form_params(31)
    net.load_state_dict(P)
    for k, v in net.state_dict().items():
        assert torch.equal(v, P[k])
    with pytest.raises(H.SsieError):
        net(torch.zeros(1, 31, 16, 16))          # CPU tensors: loud failure, no fallback
