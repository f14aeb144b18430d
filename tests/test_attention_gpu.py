"""GPU parity of the attention core (C-ABI ssie_attention_fwd/bwd) vs a PyTorch CPU float64 reference of
TransformerBlock's attention lines (model.py:107-114).  Tolerance 2e-5 * max|ref| (fp32 softmax)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    import ssie
    ssie.load()
    from ssie_amd import hostlib
    return hostlib


def ref_attention(qkv):
    n, t, _ = qkv.shape
    q, k, v = (qkv[..., i * 64:(i + 1) * 64].reshape(n, t, 4, 16).permute(0, 2, 1, 3) for i in range(3))
    att = torch.softmax(q @ k.transpose(-2, -1) / 4.0, dim=-1)
    return (att @ v).permute(0, 2, 1, 3).reshape(n, t, 64)


@pytest.mark.parametrize("n,t,scale", [(2, 4, 3.0), (3, 256, 2.0), (2, 300, 1.0), (1, 1024, 4.0)])
def test_attention_fwd_bwd(H, n, t, scale):
    g = torch.Generator().manual_seed(t)
    qkv = (torch.rand(n, t, 192, generator=g, dtype=torch.float64) * 2 - 1) * scale
    go = torch.rand(n, t, 64, generator=g, dtype=torch.float64) * 2 - 1
    x = qkv.clone().requires_grad_(True)
    o_ref = ref_attention(x)
    (o_ref * go).sum().backward()
    qd = qkv.float().cuda()
    out, lse = H.attention_fwd(qd)
    gq = H.attention_bwd(qd, out, go.float().cuda(), lse)
    torch.cuda.synchronize()
    assert (out.double().cpu() - o_ref.detach()).abs().max() <= 2e-5 * o_ref.abs().max()
    for i, nm in enumerate("qkv"):
        ref = x.grad[..., i * 64:(i + 1) * 64]; got = gq[..., i * 64:(i + 1) * 64].double().cpu()
        assert (got - ref).abs().max() <= 2e-5 * ref.abs().max(), nm


@pytest.mark.parametrize("prepass", [True, False])
@pytest.mark.parametrize("n,t", [(1, 256), (2, 300), (1, 1024)])
def test_attention_bf16_first_block_far_below_zero(H, n, t, prepass):
    """bf16 attention of the enhance-only path (attn_fwd_bf16p_kernel from 256 tokens on: lazy running maximum).  The first 32
    keys are strongly anti-aligned with every query (logit * log2(e) / 4 far below -128): the first block's re-basing must not
    rescale the still-empty state by exp2(+large) = inf (0 * inf = NaN for that query).  Bar = the bf16 path's own
    (tests/test_bf16_infer_gpu.py): finite, |err| <= 2e-2 of the output range (bf16 probabilities and values)."""
    g = torch.Generator().manual_seed(7 + t)
    qkv = torch.rand(n, t, 192, generator=g, dtype=torch.float64) * 2 - 1
    qkv[:, :, 0:64] = 8.0 + 0.5 * qkv[:, :, 0:64]               # queries: every component in [7.5, 8.5]
    qkv[:, :, 64:128] *= 0.25                                   # keys: small and zero-mean (logits of a few units) ...
    qkv[:, :32, 64:128] = -3.75 + qkv[:, :32, 64:128]           # ... except the first 32: q.k <= -16 * 7.5 * 3.5 = -420 -> * log2(e) / 4 < -150
    ref = ref_attention(qkv)
    qd = qkv.float().cuda()
    out = H.attention_fwd_bf16(qd, prepass=prepass).float().double().cpu()
    assert torch.isfinite(out).all()
    assert (out - ref).abs().max() <= 2e-2 * ref.abs().max()
    out32, _ = H.attention_fwd(qd)                              # the fp32 kernel (eager running maximum) on the same input, at its own bar
    assert (out32.double().cpu() - ref).abs().max() <= 2e-5 * ref.abs().max()
