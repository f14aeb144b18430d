"""CPU check of the two Winograd identities the HIP kernels are built on (conv_wino.hip: F(2x2, 3x3) forward / data gradient;
conv_wgrad_wino.hip: F(3x3, 2x2) weight gradient), with exactly the matrices and the deferred 1/2 scaling quoted in the kernel
headers, against the direct correlation in float64.  (The kernels themselves are pinned on the GPU: tests/test_conv_ops_gpu.py in
"winograd" mode and the fixed-2e-5 backward chain of tests/test_backward_gpu.py.)"""
import numpy as np


def test_f2x2_3x3_forward_identity():
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], float)
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
    rng = np.random.default_rng(0)
    for _ in range(20):
        d = rng.standard_normal((4, 4)); g = rng.standard_normal((3, 3))
        ref = np.array([[sum(g[r, s] * d[y + r, x + s] for r in range(3) for s in range(3)) for x in range(2)] for y in range(2)])
        got = AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T
        assert np.abs(got - ref).max() < 1e-12


def test_f2x2_3x3_bias_is_transform_position_5():
    """A constant added to all four outputs of a tile is a constant added to transform position (1, 1)."""
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
    M = np.zeros((4, 4)); M[1, 1] = 3.25
    assert np.array_equal(AT @ M @ AT.T, np.full((2, 2), 3.25))


def test_f3x3_2x2_weight_gradient_identity_with_deferred_scaling():
    AT = np.array([[1, 1, 1, 0], [0, 1, -1, 0], [0, 1, 1, 1]], float)
    Gp = np.array([[1, 0], [1, 1], [1, -1], [0, 1]], float)          # G without its 1/2 factors (the kernel's H')
    BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, -1, 0, 1]], float)
    sc = np.array([1, .5, .5, 1]); S = np.outer(sc, sc)                # applied once, in wgrad_wino_out_kernel
    rng = np.random.default_rng(1)
    acc = np.zeros((4, 4)); ref = np.zeros((3, 3))
    for _ in range(12):                                                # the sum over tiles is taken BEFORE the output transform
        d = rng.standard_normal((4, 4)); g = rng.standard_normal((2, 2))
        acc += (Gp @ g @ Gp.T) * (BT @ d @ BT.T)
        ref += np.array([[sum(g[a, b] * d[a + k, b + l] for a in range(2) for b in range(2)) for l in range(3)] for k in range(3)])
    got = AT @ (acc * S) @ AT.T
    assert np.abs(got - ref).max() < 1e-12


def test_dma_slot_decodes():
    """The mul-shift divisions used when decoding DMA slots: x // 17 (conv_wino.hip, conv_tconv.hip), x // 18 (conv_wgrad_wino.hip)."""
    for x in range(306):
        assert (x * 241) >> 12 == x // 17
    for x in range(384):                                              # 3 DMA rounds of 512 slots = 384 pixels
        assert (x * 3856) >> 16 == x // 17
    for x in range(192):
        assert (x * 3641) >> 16 == x // 18
