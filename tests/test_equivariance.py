"""RC-equivariance properties of the reference's test-suite (/root/reference/caduceus/tests/test_rcps.py) on the HIP
engine.  Because every flip is an integer index map feeding the SAME floating-point operation order, the properties hold
BIT-EXACTLY here (torch.equal), not just to tolerance -- the reference's tolerances are kept as a fallback note only."""
import pytest
import torch

from caduceus_amd import CaduceusConfig, CaduceusForMaskedLM, CaduceusMixerModel, RCPSEmbedding, RCPSLMHead, create_block
from caduceus_amd import ops
from test_model_parity import build_model

# test_rcps.py fixture constants (:41-57)
STR_TO_ID = {"[CLS]": 0, "[MASK]": 1, "A": 2, "C": 3, "G": 4, "T": 5, "N": 6}
COMP12 = {0: 0, 1: 1, 2: 5, 3: 4, 4: 3, 5: 2, 6: 6, 7: 7, 8: 8, 9: 9, 10: 10, 11: 11}
SSM = dict(d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0,
           dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True)


def rc_tensor(x):
    return torch.flip(x, dims=[-2, -1])


def _comp16():
    return torch.tensor([COMP12.get(i, i) for i in range(16)])


def test_rcps_embedding(backend):
    """test_rcps.py:27-73."""
    _, dev = backend
    torch.manual_seed(0)
    comp = {**COMP12, **{i: i for i in range(12, 16)}}
    emb = RCPSEmbedding(16, 256, comp).to(dev)
    ids = torch.randint(1, 7, (4, 512), device=dev)
    rc_ids = emb.rc(ids)
    out, out_rc = emb(ids), emb(rc_ids)
    assert out.shape == (4, 512, 512)
    assert torch.equal(out, rc_tensor(out_rc))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_rcps_wrapper_generic_submodule(backend, dtype):
    """test_rcps.py:76-109, restated directly: RCPSWrapper around a GENERIC submodule (the reference's Linear / ReLU stack, which has no
    t-frame fast path), batch 2 x seqlen 1024 x 2 * 128 channels.  The wrapper's own work is index maps, so the property holds
    bit-exactly wherever the submodule is row-wise deterministic; the reference's tolerances are the stated bound."""
    _, dev = backend
    torch.manual_seed(0)
    d_model = 128
    from caduceus_amd import RCPSWrapper
    x = torch.randn(2, 1024, 2 * d_model, device=dev, dtype=dtype)
    module = torch.nn.Sequential(torch.nn.Linear(d_model, d_model, bias=False), torch.nn.ReLU(),
                                 torch.nn.Linear(d_model, 2 * d_model, bias=True), torch.nn.ReLU(),
                                 torch.nn.Linear(2 * d_model, d_model, bias=True)).to(dev).to(dtype)
    wrapped = RCPSWrapper(module).to(dev)
    out, rc_out = wrapped(x), rc_tensor(wrapped(rc_tensor(x)))
    assert out.shape == x.shape and rc_out.shape == x.shape
    rtol, atol = (6e-4, 2e-3) if dtype == torch.float32 else (3e-3, 5e-3)  # test_rcps.py:83
    torch.testing.assert_close(out.detach(), rc_out.detach(), rtol=rtol, atol=atol)
    assert torch.equal(out, rc_out)  # (the same rows go through the same GEMMs: exact here)


@pytest.mark.parametrize("prenorm", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_rcps_add_norm_wrapper_direct(backend, prenorm, dtype):
    """test_rcps.py:112-146, restated directly: RCPSAddNormWrapper(RMSNorm) on a random (2, 1024, 2 * 128) input, prenorm on / off (the
    reference runs float16; bf16 / fp32 are the kernels' own instantiations).  One add + norm launch over both strands as rows: exact."""
    _, dev = backend
    torch.manual_seed(0)
    d_model = 128
    from caduceus_amd import RCPSAddNormWrapper
    from caduceus_amd.mamba import RMSNorm
    x = torch.randn(2, 1024, 2 * d_model, device=dev, dtype=dtype)
    norm = RMSNorm(d_model, eps=1e-5).to(dev)
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.1 * torch.randn(d_model))
    wrapped = RCPSAddNormWrapper(norm).to(dev)
    out, out_rc = wrapped(x, prenorm=prenorm), wrapped(rc_tensor(x), prenorm=prenorm)
    pairs = list(zip(out, out_rc)) if prenorm else [(out, out_rc)]
    assert len(pairs) == (2 if prenorm else 1)
    for f, r in pairs:
        assert f.shape == x.shape and r.shape == x.shape
        assert torch.equal(f, rc_tensor(r))
    # ... and the values are the un-fused reference computation (modeling_rcps.py:107-130): each strand normed by the same RMSNorm
    y = pairs[0][0].float()
    xf = x.float()
    ref = torch.cat([t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True) + 1e-5) * norm.weight.float()
                     for t in (xf[..., :d_model], xf[..., d_model:])], dim=-1)
    # (strand 2 is normed in the flipped frame: the weight meets its channels reversed)
    ref[..., d_model:] = (xf[..., d_model:] * torch.rsqrt(xf[..., d_model:].pow(2).mean(-1, keepdim=True) + 1e-5)
                          * norm.weight.float().flip(0))
    rtol, atol = (6e-4, 2e-3) if dtype == torch.float32 else ((3e-3, 5e-3) if dtype == torch.float16 else (3e-2, 5e-2))
    torch.testing.assert_close(y, ref, rtol=rtol, atol=atol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("bidirectional", [True, False])
def test_rcps_mamba_block(backend, dtype, fused, bidirectional):
    """test_rcps.py:149-202 (residual None and residual = x)."""
    _, dev = backend
    torch.manual_seed(0)
    blk = create_block(64, ssm_cfg=SSM, norm_epsilon=1e-5, rms_norm=True, fused_add_norm=fused, layer_idx=0,
                       bidirectional=bidirectional, rcps=True).to(dev)
    x = torch.randn(2, 200, 128, device=dev).to(dtype)
    for res in (None, x.float()):
        h, r = blk(x, res)
        h2, r2 = blk(rc_tensor(x), None if res is None else rc_tensor(res))
        assert torch.equal(h, rc_tensor(h2)) and torch.equal(r, rc_tensor(r2))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rcps_lm_head(backend, dtype):
    """test_rcps.py:205-260."""
    _, dev = backend
    torch.manual_seed(0)
    comp = {**COMP12, **{i: i for i in range(12, 16)}}
    head = RCPSLMHead(true_dim=128, vocab_size=16, complement_map=comp).to(dev)
    x = torch.randn(2, 300, 256, device=dev).to(dtype)
    a, b = head(x), head(rc_tensor(x))
    cm = _comp16().to(dev)
    assert torch.equal(a, torch.flip(b[..., cm], dims=[1]))


@pytest.mark.parametrize("name", ["ps_fused", "ps_unfused", "ps_fused_ewmul", "ps_fused_untied", "ps_fused_layernorm"])
@pytest.mark.parametrize("amp", [False, True])
def test_rcps_mamba_lm_and_backbone(backend, name, amp):
    """test_rcps.py:263-419: backbone and LM logits / softmax equivariance, plus collapse invariance (:422-490)."""
    _, dev = backend
    model, cfg, sd, rec = build_model(name, dev)
    comp = sd["lm_head.complement_map"].to(dev)
    ids = rec["input_ids"].to(dev)
    rc_ids = comp[torch.flip(ids, dims=[-1])]
    with torch.autocast(dev.type, dtype=torch.bfloat16, enabled=amp):
        a, b = model(ids), model(rc_ids)
        ha, hb = model.caduceus(ids).last_hidden_state, model.caduceus(rc_ids).last_hidden_state
    assert torch.equal(a.logits, torch.flip(b.logits[..., comp], dims=[1]))
    sa, sb = a.logits.softmax(-1), b.logits.softmax(-1)
    # softmax is torch's (its denominator is summed in vocabulary order): reference tolerance, not bit-exact
    torch.testing.assert_close(sa, torch.flip(sb[..., comp], dims=[1]), rtol=6e-4, atol=2e-3)
    assert torch.equal(ha, rc_tensor(hb))
    D = cfg["d_model"]
    col_a = (ha[..., :D] + torch.flip(ha[..., D:], dims=[1, 2])) / 2
    col_b = (hb[..., :D] + torch.flip(hb[..., D:], dims=[1, 2])) / 2
    # the collapsed representation is RC-INVARIANT (test_rcps.py:422-490); exact because a + b == b + a
    assert torch.equal(col_a, col_b)


@pytest.mark.parametrize("L", [64, 1000, 1024, 2500])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_and_conv_mirror_exact(backend, L, dtype):
    """A right-to-left row must be the exact mirror of a left-to-right row on flipped data (any L, any chunk tail)."""
    _, dev = backend
    g = torch.Generator().manual_seed(L)
    E, N = 5, 16
    r = lambda *s: torch.randn(*s, generator=g).to(dev).to(dtype)
    u, delta, z, Bm, Cm = r(E, 1, L), r(E, 1, L), r(E, 1, L), r(N, 1, L), r(N, 1, L)
    A = -(0.5 + 15.5 * torch.rand(E, N, generator=g)).to(dev)
    D, bias = torch.randn(E, generator=g).to(dev), (torch.randn(E, generator=g) - 3).to(dev)
    f = lambda t: t.flip(-1).contiguous()
    fwd_on_flipped = ops.selective_scan(f(u), f(delta), A, f(Bm), f(Cm), D, f(z), bias, 1, 0, 0)
    rev = ops.selective_scan(u, delta, A, Bm, Cm, D, z, bias, 1, 1, 1)
    assert torch.equal(rev, f(fwd_on_flipped))
    w = (0.5 * torch.randn(E, 1, 4, generator=g)).to(dev)
    cb = (0.2 * torch.randn(E, generator=g)).to(dev)
    assert torch.equal(ops.causal_conv1d(u, w, cb, 1, 1, 1), f(ops.causal_conv1d(f(u), w, cb, 1, 0, 0)))
