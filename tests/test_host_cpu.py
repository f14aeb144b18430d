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


def test_32bit_offset_guard(pkg):
    """The kernels index activations with 32-bit element offsets: a plan / operator whose largest tensor reaches 2^31 floats
    must be refused on the host (NULL / SSIE_E_SHAPE) instead of wrapping into a GPU fault.  No kernel is launched: geometry
    is rejected before anything is enqueued (the pointers below are fakes)."""
    H, _ = pkg
    L = H._proto()
    cf = (ctypes.c_float * 8)(*[1.0] * 8)
    # BASELINE sizes are far inside the limit: N = 32 at 128 x 128 x 256 is 2 * 32 * 128 * 128 * 260 = 2.7e8 floats
    for ok in ((32, 31, 128, 128), (32, 256, 128, 128), (1, 31, 1024, 1024), (200, 31, 128, 128)):
        h = L.ssie_plan_create(*ok, cf)
        assert h, ok
        L.ssie_plan_destroy(h)
    # 2N * H * W * 64 >= 2^31  <=>  N >= 1024 at 128 x 128; one 4096 x 4096 image is 2 * 4096^2 * 64 = 2^31 exactly
    for bad in ((1024, 31, 128, 128), (1, 31, 4096, 4096), (64, 256, 256, 256), (1 << 20, 31, 128, 128)):
        assert not L.ssie_plan_create(*bad, cf), bad
    assert L.ssie_plan_create(1023, 31, 128, 128, cf)
    # granular operators: SSIE_E_SHAPE (2) from the geometry builder, before any launch
    fake = ctypes.c_void_p(1 << 20)
    src = H.SrcT(1 << 20, 64, 64, 0, 128, 128)
    n_bad = 2048 + 1                                              # 2049 * 128 * 128 * 64 > 2^31 - 1
    ws_bytes = ctypes.c_size_t(1 << 30)
    rc = L.ssie_conv2d_fwd(ctypes.byref(src), 1, n_bad, 128, 128, fake, 64, fake, 64, 3, 1, 0, None, None, fake, 64, 0,
                           fake, ws_bytes, None)
    assert rc == 2, rc
    rc = L.ssie_conv2d_wgrad(ctypes.byref(src), n_bad, 128, 128, fake, 64, 0, 64, 3, 1, 64, 0, fake, fake, 0, fake, ws_bytes, None)
    assert rc == 2, rc
    rc = L.ssie_conv2d_dgrad(fake, 64, 0, n_bad, 128, 128, 64, fake, 64, 0, 64, 3, 1, fake, 128, 128, 64, 0, None, 0, 0,
                             fake, ws_bytes, None)
    assert rc == 2, rc


def test_polluted_environment_does_not_reach_the_product_loader(pkg, monkeypatch):
    """stray SSIE_* variables must not change what the product loads or how it launches.  Without
    SSIE_DEBUG=1 the loader ignores SSIE_HIP_LIB and applies none of the eighteen development switches (a fresh load with a
    polluted environment resolves the in-tree library and calls no ssie_debug_set_*); the module's plans keep their hipGraph."""
    H, M = pkg
    called = []

    class Spy:
        def __init__(self, real):
            object.__setattr__(self, "_real", real)

        def __getattr__(self, name):
            f = getattr(self._real, name)
            if name.startswith("ssie_debug_set_"):
                called.append(name)
            return f

        def __setattr__(self, name, v):
            setattr(self._real, name, v)

    for env, _ in H._DEBUG_ENV:
        monkeypatch.setenv(env, "0")
    monkeypatch.setenv("SSIE_HIP_LIB", "/nonexistent/libssie_other.so")
    monkeypatch.delenv("SSIE_DEBUG", raising=False)
    real_cdll = H.C.CDLL
    loaded = []
    monkeypatch.setattr(H.C, "CDLL", lambda path: (loaded.append(path), Spy(real_cdll(path)))[1])
    monkeypatch.setattr(H, "_LIB", None)
    L = H.lib()
    assert loaded == [H._build.LIB] and called == []
    assert b"gfx950" in L.ssie_version()
    assert not H.debug_enabled()
    # the same environment WITH the opt-in: the alternate path is taken (and, not existing, refused loudly)
    monkeypatch.setenv("SSIE_DEBUG", "1")
    monkeypatch.setattr(H, "_LIB", None)
    with pytest.raises(H.SsieError):
        H.lib()
    monkeypatch.setenv("SSIE_HIP_LIB", H._build.LIB)
    for env, _ in H._DEBUG_ENV:                     # one harmless switch only (its default value): the setters are process-global
        monkeypatch.delenv(env, raising=False)
    monkeypatch.setenv("SSIE_GRAPH", "0")
    monkeypatch.setattr(H, "_LIB", None)
    H.lib()
    assert called == ["ssie_debug_set_graph"]
    monkeypatch.setattr(H, "_LIB", None)
