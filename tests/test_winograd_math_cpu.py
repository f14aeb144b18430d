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


def _bt6(d):
    """the 1-D input transform of F(4x4, 3x3) exactly as conv_wino4.hip evaluates it (two independent output triples)"""
    d0, d1, d2, d3, d4, d5 = d
    r0 = d0 * 4 + (d4 - d2 * 5)
    s, t = d4 - d2 * 4, d3 - d1 * 4
    r1, r2 = s + t, s - t
    s2, t2 = d4 - d2, d3 - d1
    r3, r4 = t2 * 2 + s2, s2 - t2 * 2
    r5 = d1 * 4 + (d5 - d3 * 5)
    return np.array([r0, r1, r2, r3, r4, r5])


def _at6(m):
    """the 1-D output transform of F(4x4, 3x3) as the kernel's epilogue evaluates it"""
    m0, m1, m2, m3, m4, m5 = m
    s1, d1, s2, d2 = m1 + m2, m1 - m2, m3 + m4, m3 - m4
    return np.array([m0 + s1 + s2, d1 + 2 * d2, s1 + 4 * s2, d1 + 8 * d2 + m5])


def test_f4x4_3x3_forward_identity_as_coded():
    """conv_wino4.hip: Y = A^T [(G g G^T) (.) (B^T d B)] A with the matrices of its header, the input transform done horizontally
    first (rows of the patch) and then vertically, xi = 6 i + j with i the vertical index - against the direct correlation."""
    BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
    G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], float)
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)
    rng = np.random.default_rng(4)
    for _ in range(10):
        d = rng.standard_normal((6, 6)); g = rng.standard_normal((3, 3))
        ref = np.array([[sum(g[r, s] * d[y + r, x + s] for r in range(3) for s in range(3)) for x in range(4)] for y in range(4)])
        assert np.abs(AT @ ((G @ g @ G.T) * (BT @ d @ BT.T)) @ AT.T - ref).max() < 1e-11
        rt = np.array([_bt6(d[a]) for a in range(6)])                  # horizontal: rt[a][j]
        V = np.array([_bt6(rt[:, j]) for j in range(6)]).T             # vertical:   V[i][j]
        assert np.abs(V - BT @ d @ BT.T).max() < 1e-11
        M = (G @ g @ G.T) * V
        t = np.array([_at6(M[:, j]) for j in range(6)]).T              # over i: t[y][j]
        Y = np.array([_at6(t[y]) for y in range(4)])                   # over j: Y[y][x]
        assert np.abs(Y - ref).max() < 1e-11


def test_f4x4_dma_slot_decode():
    """conv_wino4.hip's halo layout [shift group (col >> 2) & 1][half][phase col & 3][row][col >> 3], group 1 shifted by 8 bytes: the
    slot decode of its DMA table covers every (row, column) of the 18 x 66 halo tile exactly once per channel half; the reader's
    address of patch element (a, b) of tile (wm, tx) - two per-lane bases + one compile-time offset - is that slot; and the 16 lanes of a
    k-group hit 16 different 8-byte bank pairs for every (a, b) (r2 // 9 == (r2 * 7282) >> 16 on its range)."""
    V_HPH, V_IDX = 18, 9
    V_PLANE = V_HPH * V_IDX; V_HSL = 4 * V_PLANE; V_SG = (2 * V_HSL + 63) // 64 * 64; V_SLOTS = 2 * V_SG; V_SHIFT = V_SG * 16 + 8
    assert (V_PLANE, V_HSL, V_SG, V_SLOTS) == (162, 648, 1344, 2688)
    byte_of = {}
    for sid in range(6 * 512):
        sg = int(sid >= V_SG); r = sid - sg * V_SG
        h = int(r >= V_HSL); r1 = r - h * V_HSL
        ph = (r1 >= V_PLANE) + (r1 >= 2 * V_PLANE) + (r1 >= 3 * V_PLANE); r2 = r1 - ph * V_PLANE
        hy = (r2 * 7282) >> 16
        if r2 < V_PLANE:
            assert hy == r2 // 9
        hx = 8 * (r2 - hy * V_IDX) + 4 * sg + ph
        real = sid < V_SLOTS and r < 2 * V_HSL and hx < 66
        if real:
            assert 0 <= hy < 18 and (h, hy, hx) not in byte_of
            piece = sid // 64
            byte_of[(h, hy, hx)] = piece * 1024 + (8 if piece >= 21 else 0) + (sid % 64) * 16      # where the DMA puts the slot
    assert len(byte_of) == 2 * 18 * 66
    aoff = lambda a, b: (b >> 2) * V_SHIFT + (((b & 3) * V_HPH + a) * V_IDX) * 16
    for wm in range(4):
        for g in range(4):
            for a in range(6):
                for b in range(6):
                    banks = set()
                    for tx in range(16):
                        u, e = tx >> 1, tx & 1
                        acommon = ((((g >> 1) * 4) * V_HPH + 4 * wm) * V_IDX + u) * 16 + (g & 1) * 8
                        base = acommon + e * V_SHIFT if b < 4 else acommon + e * (16 - V_SHIFT)
                        addr = base + aoff(a, b)
                        assert addr == byte_of[(g >> 1, 4 * wm + a, 4 * tx + b)] + (g & 1) * 8, (wm, g, a, b, tx)
                        banks.add((addr // 8) % 16)
                    assert len(banks) == 16, (wm, g, a, b)
