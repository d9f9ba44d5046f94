"""The hand-written MFMA projection kernels (csrc/gemm.hip) against fp32 matrix products of the same bf16 operands, on
the host emulator (MFMA / LDS-DMA semantics restated in tests/emu + cad_common.h) and on the GPU."""
import pytest
import torch

from caduceus_amd import ops


def _bf(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(torch.bfloat16)


@pytest.mark.parametrize("M,K,T", [(48, 32, 70), (128, 64, 200), (80, 256, 130), (40, 512, 96), (1024, 128, 64)])
def test_proj_wxT(backend, M, K, T):
    name, dev = backend
    W, X = _bf(M, K, seed=1), _bf(T, K, seed=2)
    out = ops.proj_wxT(W.to(dev), X.to(dev))
    ref = W.float() @ X.float().t()
    assert out.shape == (M, T) and out.dtype == torch.bfloat16
    torch.testing.assert_close(out.float().cpu(), ref, rtol=1e-2, atol=1e-2 * float(ref.abs().max()) / 8)
    # the rounding is that of ONE fp32-accumulated product rounded to bf16
    torch.testing.assert_close(out.float().cpu(), ref.to(torch.bfloat16).float(), rtol=2e-2, atol=2e-2)


def test_proj_wxT_is_position_independent(backend):
    """A token's projection does not depend on where the token sits (block, sub-block, lane): permuting the tokens
    permutes the output columns bit for bit -- what keeps the t-frame strands / directions bit-identical."""
    name, dev = backend
    M, K, T = 96, 64, 333
    W, X = _bf(M, K, seed=3), _bf(T, K, seed=4)
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(5))
    a = ops.proj_wxT(W.to(dev), X.to(dev)).cpu()
    b = ops.proj_wxT(W.to(dev), X[perm].contiguous().to(dev)).cpu()
    assert torch.equal(a[:, perm], b)


def test_proj_wxT_strided_views(backend):
    name, dev = backend
    M, K, T = 64, 64, 150
    Wb, Xb = _bf(M, K + 8, seed=6), _bf(T, K + 16, seed=7)
    outb = torch.zeros(M, T + 8, dtype=torch.bfloat16, device=dev)
    W, X = Wb[:, :K].to(dev), Xb[:, :K].to(dev)
    Wd, Xd = Wb.to(dev)[:, :K], Xb.to(dev)[:, :K]
    ops.proj_wxT(Wd, Xd, out=outb[:, :T])
    ref = (Wb[:, :K].float() @ Xb[:, :K].float().t()).to(torch.bfloat16)
    torch.testing.assert_close(outb[:, :T].float().cpu(), ref.float(), rtol=2e-2, atol=2e-2)
    assert float(outb[:, T:].abs().max()) == 0.0


@pytest.mark.parametrize("M,K,T", [(512, 16, 256), (96, 48, 200), (64, 8, 72), (130, 40, 64), (512, 48, 192)])
@pytest.mark.parametrize("acc", [False, True])
def test_proj_wx(backend, M, K, T, acc):
    """Thin-K channel-major product (dt_proj; the x_proj input gradient with its addend), transposing LDS reads."""
    name, dev = backend
    W, X = _bf(M, K, seed=11), _bf(K, T, seed=12)
    A = _bf(M, T, seed=13) if acc else None
    out = ops.proj_wx(W.to(dev), X.to(dev), acc=None if A is None else A.to(dev))
    ref = W.float() @ X.float()
    if acc:
        ref = ref.to(torch.bfloat16).float() + A.float()
    torch.testing.assert_close(out.float().cpu(), ref.to(torch.bfloat16).float(), rtol=2e-2, atol=2e-2)
    if acc:  # in place
        buf = A.clone().to(dev)
        ops.proj_wx(W.to(dev), X.to(dev), out=buf, acc=buf)
        assert torch.equal(buf.cpu(), out.cpu())


@pytest.mark.parametrize("M,K,T", [(48, 512, 384), (16, 512, 200), (33, 128, 136), (64, 256, 1032), (5, 192, 64)])
def test_proj_wx_thin_m_deep_k(backend, M, K, T):
    """x_proj / d(dt_lr): thin M, K walked in 64-row chunks through the DMA ring, several 128-token blocks per workgroup."""
    name, dev = backend
    assert ops.proj_wx_supported(_bf(1).to(dev), K, T, M=M)
    W, X = _bf(M, K, seed=21) * 0.2, _bf(K, T, seed=22)
    out = ops.proj_wx(W.to(dev), X.to(dev))
    ref = W.float() @ X.float()
    assert out.shape == (M, T)
    torch.testing.assert_close(out.float().cpu(), ref.to(torch.bfloat16).float(), rtol=2e-2, atol=2e-2 * float(ref.abs().max()) / 4)
    # strided output rows (the x_proj gradient operand is assembled in place) and position independence
    buf = torch.zeros(M + 3, T + 8, dtype=torch.bfloat16, device=dev)
    ops.proj_wx(W.to(dev), X.to(dev), out=buf[1:M + 1, :T])
    assert torch.equal(buf[1:M + 1, :T].cpu(), out.cpu()) and float(buf[0].abs().max()) == 0 and float(buf[:, T:].abs().max()) == 0
    perm = torch.randperm(T // 8, generator=torch.Generator().manual_seed(3))
    Xp = X.view(K, T // 8, 8)[:, perm].reshape(K, T).contiguous()
    outp = ops.proj_wx(W.to(dev), Xp.to(dev))
    assert torch.equal(outp.view(M, T // 8, 8).cpu(), out.view(M, T // 8, 8)[:, perm].cpu())


@pytest.mark.parametrize("M,K,T", [(64, 1024, 264), (48, 256, 136)])
def test_proj_wx_thin_m_two_k_halves(backend, M, K, T):
    """x_proj at d_inner 1024 (configs[4]): 64 x 1024 of W_x do not fit LDS next to the X ring, so the mixer runs two K halves, the second
    with the first as its (aliasing) addend -- fp32 sums of the half + the widened addend, rounded once.  Against the fp32 product; the halves
    are views (row stride K of W, rows K/2.. of X), as the mixer passes them."""
    name, dev = backend
    W, X = (_bf(M, K, seed=31) * 0.2).to(dev), _bf(K, T, seed=32).to(dev)
    if K == 1024:
        assert not ops.proj_wx_supported(X, K, T, M=M)  # the case the split exists for
    assert ops.proj_wx_supported(X, K // 2, T, M=M)
    out = ops.proj_wx(W[:, :K // 2], X[:K // 2])
    first = out.clone()
    ops.proj_wx(W[:, K // 2:], X[K // 2:], out=out, acc=out)
    ref = W.float().cpu() @ X.float().cpu()
    torch.testing.assert_close(out.float().cpu(), ref.to(torch.bfloat16).float(), rtol=2e-2, atol=2e-2 * float(ref.abs().max()) / 4)
    # exactly bf16(first half as stored + fp32 sums of the second half): at most one bf16 ulp from a reference assembled the same way
    ref2 = (first.float().cpu() + W[:, K // 2:].float().cpu() @ X[K // 2:].float().cpu()).to(torch.bfloat16).float()
    ulp = torch.maximum(ref2.abs(), torch.tensor(1e-30)) * 2.0 ** -7
    assert bool(((out.float().cpu() - ref2).abs() <= ulp).all())
    # a separate addend buffer gives the same bits as the aliasing one
    out2 = ops.proj_wx(W[:, K // 2:], X[K // 2:], acc=first)
    assert torch.equal(out2.cpu(), out.cpu())


@pytest.mark.parametrize("M,K,T", [(512, 16, 256), (70, 24, 136)])
def test_proj_wx_softplus_bias_epilogue(backend, M, K, T):
    """dt_proj + delta_bias + softplus in one pass (fp32 evaluation, one rounding to bf16)."""
    name, dev = backend
    W, X = _bf(M, K, seed=31), _bf(K, T, seed=32)
    bias = torch.randn(M, generator=torch.Generator().manual_seed(33)) - 2.0
    out = ops.proj_wx(W.to(dev), X.to(dev), softplus_bias=bias.to(dev))
    ref = torch.nn.functional.softplus(W.float() @ X.float() + bias[:, None], threshold=20.0)
    torch.testing.assert_close(out.float().cpu(), ref.to(torch.bfloat16).float(), rtol=1.6e-2, atol=1e-6)
    with pytest.raises(Exception):
        ops.proj_wx(W.to(dev), X.to(dev), acc=out, softplus_bias=bias.to(dev))


@pytest.mark.parametrize("case", [(256, 300, 130), (512, 2048 + 40, 200), (256, 1024, 64)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_fp8_in_proj_quantisation_and_product(backend, case, dtype):
    """configs[4]'s fp8 projection: cad_quant_rows_fp8 (per-token e4m3 quantisation) is held bit-for-bit to torch's own
    float8_e4m3fn cast of x / scale, and cad_proj_wxT_fp8 to the fp64 product of the DE-QUANTISED operands (the only
    difference left is fp32 accumulation order + the bf16 output rounding); against the UN-quantised product the relative error
    norm stays below 6e-2 -- the stated tolerance of the fp8 path: e4m3 keeps 3 mantissa bits, i.e. an RMS rounding error of
    2^-4 / sqrt(3) = 3.6 % per operand, 5 % per product, and a sum of K such products keeps that relative error."""
    name, dev = backend
    K, M, T = case
    g = torch.Generator().manual_seed(K + T)
    x = torch.randn(T, K, generator=g)
    x[3] *= 30.0      # rows of very different magnitude: the per-token scale absorbs them
    x[5] = 0.0        # an all-zero row (scale 1, zeros)
    W = torch.randn(M, K, generator=g) / K ** 0.5
    xd = x.to(dtype).to(dev)
    q, sx = ops.quant_rows_fp8(xd)
    xf = xd.float().cpu()
    want_s = torch.where(xf.abs().amax(1) > 0, xf.abs().amax(1) * (1.0 / 448.0), torch.ones(T))
    torch.testing.assert_close(sx.cpu(), want_s, rtol=1e-6, atol=0)
    want_q = (xf * (1.0 / sx.cpu())[:, None]).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q.cpu(), want_q)
    Wq, sw = ops.quant_weight_fp8(W.to(dev))
    out = ops.proj_wxT_fp8(Wq, sw, q, sx)
    deq_w = Wq.cpu().view(torch.float8_e4m3fn).double() * sw.cpu().double()[:, None]
    deq_x = q.cpu().view(torch.float8_e4m3fn).double() * sx.cpu().double()[:, None]
    ref = deq_w @ deq_x.t()
    torch.testing.assert_close(out.float().cpu().double(), ref, rtol=8e-3, atol=8e-3 * float(ref.abs().max()) / 16)
    full = W.double() @ xf.double().t()
    rel = float((out.float().cpu().double() - full).norm() / full.norm())
    assert rel < 6e-2, rel


@pytest.mark.parametrize("M,K,T", [(16, 512, 128 * 5), (16, 256, 128 * 3), (32, 512, 128 * 2), (16, 512, 128 * 300)])
def test_proj_wx_wgrad_fused(backend, M, K, T):
    """cad_proj_wx_wgrad: d(dt_lr) = W . X and dW = X . Y^T from one pass over X -- the product is bit-identical to the plain thin
    kernel, the weight gradient equals the fp32 product of the bf16 operands (fp32 accumulation over T in a fixed order)."""
    name, dev = backend
    if name == "emu" and T > 128 * 8:
        pytest.skip("long stream: device only")
    W, X, Y = _bf(M, K, seed=5) * 0.2, _bf(K, T, seed=6), _bf(M, T, seed=7)
    out, dW = ops.proj_wx_wgrad(W.to(dev), X.to(dev), Y.to(dev))
    plain = ops.proj_wx(W.to(dev), X.to(dev))
    assert torch.equal(out, plain)
    ref = X.float().double() @ Y.float().double().t()
    assert dW.shape == (K, M) and dW.dtype == torch.float32
    torch.testing.assert_close(dW.cpu().double(), ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))


@pytest.mark.parametrize("M,K,T", [(48, 512, 128 * 70), (40, 256, 128 * 5), (24, 512, 128 * 300), (64, 256, 128 * 9), (7, 512, 128), (64, 512, 256)])
def test_proj_wgrad_only(backend, M, K, T):
    """cad_proj_wx_wgrad with W == NULL: dW = Y . X^T alone (dW_x = d(dbc) . xc^T of the x_proj backward), any M <= 64, against the
    fp64 product of the same bf16 operands."""
    dev = backend[1]
    g = torch.Generator().manual_seed(M * 7 + K)
    X = torch.randn(K, T, generator=g).to(torch.bfloat16)
    Y = torch.randn(M, T, generator=g).to(torch.bfloat16)
    assert ops.proj_wgrad_only_supported(X.to(dev), M, K, T)
    assert not ops.proj_wgrad_only_supported(X.to(dev), 65, K, T)
    dW = ops.proj_wgrad_only(X.to(dev), Y.to(dev))
    ref = Y.double() @ X.double().t()
    assert dW.shape == (M, K) and dW.dtype == torch.float32
    torch.testing.assert_close(dW.cpu().double(), ref, rtol=1e-4, atol=1e-3 * (T / 128) ** 0.5)


def test_mixer_layer_own_wgrad_kernels_match_the_library_path(backend, monkeypatch):
    """One BiMamba mixer layer (d_model 256: E = 512, dt_rank 16, T = 256 tokens), bf16, forward + backward with dW_dt / dW_x from the
    own kernels (partial slots of both parameter sets folded by one sum) against the GEMM-library path: same gradients."""
    from caduceus_amd import mixer
    from caduceus_amd.mamba import Mamba
    name, dev = backend
    torch.manual_seed(0)
    mf, mr = Mamba(256, device=dev), Mamba(256, device=dev)
    mr.in_proj.weight = mf.in_proj.weight
    mr.out_proj.weight = mf.out_proj.weight
    hn = torch.randn(2, 1, 128, 256, device=dev).to(torch.bfloat16)
    g = torch.randn(2, 1, 128, 256, device=dev).to(torch.bfloat16)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(mixer, "_FUSED_WGRAD", on)
        monkeypatch.setattr(mixer, "_OWN_DWX", on)
        for p in list(mf.parameters()) + list(mr.parameters()):
            p.grad = None
        x = hn.clone().requires_grad_(True)
        mixer.prepare_step_cache([(mf, mr)], torch.bfloat16)
        out = mixer.bimamba_mixer(x, mf, mr, 1)
        out.backward(g)
        res[on] = {n: p.grad.detach().float().cpu().clone() for m, tag in ((mf, "f"), (mr, "r"))
                   for n, p in ((tag + "." + k, v) for k, v in m.named_parameters())}
        res[on]["x"] = x.grad.float().cpu()
    assert ops.proj_wx_wgrad_supported(hn, 16, 512, 256) and ops.proj_wgrad_only_supported(hn, 48, 512, 256)
    for k in res[True]:
        a, b = res[True][k], res[False][k]
        rel = float((a - b).norm() / b.norm().clamp_min(1e-20))
        assert rel < (2e-2 if ("x_proj" in k or "dt_proj.weight" in k) else 1e-2), (k, rel)


@pytest.mark.parametrize("M,K,T,panels", [(256, 512, 384, 2), (256, 512, 200, 1), (128, 256, 136, 2), (256, 256, 128, 2), (128, 512, 1000, 2)])
def test_proj_xTw_token_major_out_proj(backend, M, K, T, panels):
    """cad_proj_xTw: out (T, M) token-major = (X1 + X2)^T W^T with fp32 accumulation over BOTH panels (out_proj of the tied BiMamba
    mixer on y_f, y_r) -- against the fp32 product, and bit-for-bit against one rounding of it where the sum is exact."""
    name, dev = backend
    W, X1 = _bf(M, K, seed=31), _bf(K, T, seed=32)
    X2 = _bf(K, T, seed=33) if panels == 2 else None
    assert ops.proj_xTw_supported(X1.to(dev), M, K, T)
    out = ops.proj_xTw(W.to(dev), X1.to(dev), None if X2 is None else X2.to(dev))
    assert out.shape == (T, M) and out.dtype == torch.bfloat16
    ref = X1.float().t() @ W.float().t()
    if X2 is not None:
        ref = ref + X2.float().t() @ W.float().t()
    torch.testing.assert_close(out.float().cpu(), ref, rtol=1e-2, atol=1e-2 * float(ref.abs().max()) / 8)
    torch.testing.assert_close(out.float().cpu(), ref.to(torch.bfloat16).float(), rtol=2e-2, atol=2e-2 * float(ref.abs().max()) / 8)
    # position independence: permuting the tokens permutes the output rows bit for bit (RC-equivariance of the t-frame)
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(5))
    out_p = ops.proj_xTw(W.to(dev), X1[:, perm].contiguous().to(dev), None if X2 is None else X2[:, perm].contiguous().to(dev))
    assert torch.equal(out.cpu()[perm], out_p.cpu())


def test_mixer_layer_own_out_proj_matches_the_library_path(backend, monkeypatch):
    """One BiMamba mixer layer forward + backward with out_proj on cad_proj_xTw against the hipBLASLt / torch path on [y_f ; y_r]."""
    from caduceus_amd import mixer
    from caduceus_amd.mamba import Mamba
    name, dev = backend
    torch.manual_seed(3)
    D, Lq = 128, 256
    mf, mr = Mamba(D, device=dev), Mamba(D, device=dev)
    mr.in_proj.weight = mf.in_proj.weight
    mr.out_proj.weight = mf.out_proj.weight
    hn0 = torch.randn(2, 1, Lq, D, device=dev).to(torch.bfloat16)
    g = torch.randn(2, 1, Lq, D, device=dev).to(torch.bfloat16)
    res = {}
    for own in (True, False):
        monkeypatch.setattr(mixer, "_OWN_OUT_PROJ", own)
        for p in list(mf.parameters()) + list(mr.parameters()):
            p.grad = None
        hn = hn0.clone().requires_grad_(True)
        out = mixer.bimamba_mixer(hn, mf, mr, 1)
        out.backward(g)
        res[own] = (out.detach().float().cpu(), hn.grad.float().cpu(), mf.out_proj.weight.grad.float().cpu())
    for a, b in zip(res[True], res[False]):
        torch.testing.assert_close(a, b, rtol=2e-2, atol=2e-2 * max(1.0, float(b.abs().max())))


@pytest.mark.parametrize("M,N,T,slices", [(256, 256, 128, None), (512, 256, 32 * 12, None), (256, 512, 64, None), (1024, 256, 32 * 5, 1)])
def test_gemm_stream_weight_gradient_partials(backend, monkeypatch, M, N, T, slices):
    """cad_gemm_stream / CAD_GEMM_PARTIALS: (M, N) fp32 = a (M, T) channel-major @ b (T, N) token-major over ALL tokens, fp32
    accumulation (dW_in / dW_out of the mixer), against the fp32 product of the same bf16 operands -- for several K-slice counts, on
    operands that are views with a row pitch."""
    name, dev = backend
    if slices is not None:
        monkeypatch.setattr(ops, "_cu_count", lambda: slices * (M // 256) * (N // 256))
    a_full, b_full = _bf(M, T + 16, seed=41), _bf(T, N + 8, seed=42)
    a, b = a_full.to(dev)[:, 8:8 + T], b_full.to(dev)[:, :N]
    out = ops.wgrad_cm_tm(a, b)
    assert out is not None and out.shape == (M, N) and out.dtype == torch.float32
    ref = a_full[:, 8:8 + T].float() @ b_full[:, :N].float()
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-4 * float(ref.abs().max()))
    # shapes the kernel does not serve are refused, not mis-computed
    assert ops.wgrad_cm_tm(a[:200], b) is None and ops.wgrad_cm_tm(a, b[:, :100]) is None


@pytest.mark.parametrize("M,K,T", [(256, 1024, 256), (256, 512, 768), (512, 64, 256), (256, 32, 512)])
def test_gemm_stream_token_major_input_gradient(backend, M, K, T):
    """cad_gemm_stream / CAD_GEMM_OUT_T_BF16: out (T, M) token-major bf16 = X (K, T)^T @ Wt (M, K)^T with both operands streamed
    (d(x2d) = dxz^T W_in, K = 2 d_inner) -- against the fp32 product and one bf16 rounding of it, several tiles per workgroup."""
    name, dev = backend
    Wt, X = _bf(M, K, seed=43), _bf(K, T, seed=44)
    out = ops.proj_xTw_stream(Wt.to(dev), X.to(dev))
    assert out is not None and out.shape == (T, M) and out.dtype == torch.bfloat16
    ref = X.float().t() @ Wt.float().t()
    torch.testing.assert_close(out.float().cpu(), ref, rtol=1e-2, atol=1e-2 * float(ref.abs().max()) / 8)
    torch.testing.assert_close(out.float().cpu(), ref.to(torch.bfloat16).float(), rtol=2e-2, atol=2e-2 * float(ref.abs().max()) / 8)
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(5))
    out_p = ops.proj_xTw_stream(Wt.to(dev), X[:, perm].contiguous().to(dev))
    assert torch.equal(out.cpu()[perm], out_p.cpu())  # position independence (RC-equivariance of the t-frame)
    assert ops.proj_xTw_stream(Wt.to(dev)[:100], X.to(dev)) is None


def test_mixer_layer_own_streaming_gemms_match_the_library_path(backend, monkeypatch):
    """One d_model 256 BiMamba mixer layer backward with d(x2d), dW_in and dW_out on cad_gemm_stream against the torch.mm / K-split bmm
    (hipBLASLt) path: the input gradient within one bf16 rounding of the same fp32-accumulated product (bit-identical on the MI355X
    at the production shape, tools/gemm_stream_bench.py), the weight gradients within the rounding of the library's bf16 partial products."""
    from caduceus_amd import mixer
    from caduceus_amd.mamba import Mamba
    name, dev = backend
    torch.manual_seed(4)
    D, Lq = 256, 256
    mf, mr = Mamba(D, device=dev), Mamba(D, device=dev)
    mr.in_proj.weight = mf.in_proj.weight
    mr.out_proj.weight = mf.out_proj.weight
    hn0 = torch.randn(2, 1, Lq, D, device=dev).to(torch.bfloat16)
    g = torch.randn(2, 1, Lq, D, device=dev).to(torch.bfloat16)
    res = {}
    for own in (True, False):
        monkeypatch.setattr(mixer, "_OWN_GEMM", own)
        for p in list(mf.parameters()) + list(mr.parameters()):
            p.grad = None
        hn = hn0.clone().requires_grad_(True)
        out = mixer.bimamba_mixer(hn, mf, mr, 1)
        out.backward(g)
        res[own] = (hn.grad.float().cpu(), mf.in_proj.weight.grad.float().cpu(), mf.out_proj.weight.grad.float().cpu())
    assert ops.proj_xTw_stream(mf.in_proj.weight.detach().to(torch.bfloat16).t().contiguous(),
                               torch.zeros(4 * D, 2 * Lq, dtype=torch.bfloat16, device=dev)) is not None
    assert ops.wgrad_cm_tm(torch.zeros(4 * D, 2 * Lq, dtype=torch.bfloat16, device=dev),
                           torch.zeros(2 * Lq, D, dtype=torch.bfloat16, device=dev)) is not None
    for a, b in zip(res[True], res[False]):
        assert float((a - b).norm() / b.norm()) < 5e-3


@pytest.mark.parametrize("Lq", [1104, 512])
def test_mixer_layer_bf16_production_scans_against_the_generic_fp32_path(backend, monkeypatch, Lq):
    """One weight-tied BiMamba mixer layer (d_model 128: E = 256, dt_rank 8, d_state 16; both strands; chunks 512 + 512 + 80 with
    out-of-row lanes in the tail) through the hand-scheduled bf16 path -- dt from the dt_proj epilogue, so the backward scan is the lean
    production instantiation with the SHARED gate gradient (out2) -- against the same layer on the generic per-op autograd path in fp32
    (engine._bimamba_tframe: the generic kernel instantiations, which tests/test_kernels.py holds to the oracle): output, input gradient and
    every parameter gradient by relative error norm, at the bf16 class."""
    from caduceus_amd import engine, mixer
    from caduceus_amd.mamba import Mamba
    name, dev = backend
    torch.manual_seed(7)
    D = 128
    mf, mr = Mamba(D, device=dev), Mamba(D, device=dev)
    mr.in_proj.weight = mf.in_proj.weight
    mr.out_proj.weight = mf.out_proj.weight
    hn0 = torch.randn(2, 1, Lq, D, device=dev).to(torch.bfloat16)
    g = torch.randn(2, 1, Lq, D, device=dev).to(torch.bfloat16)
    params = {tag + "." + k: v for m, tag in ((mf, "f"), (mr, "r")) for k, v in m.named_parameters()}

    def run(fast):
        for p in params.values():
            p.grad = None
        hn = (hn0 if fast else hn0.float()).clone().requires_grad_(True)
        if fast:
            assert mixer.can_use(mf, mr, "add") and ops.proj_wx_supported(hn0, mf.dt_rank, 2 * Lq)  # fused softplus -> delta_is_dt
            out = mixer.bimamba_mixer(hn, mf, mr, 1)
        else:
            monkeypatch.setattr(mixer, "can_use", lambda *a, **k: False)
            out = engine._bimamba_tframe(hn, mf, mr, "add", True)
            monkeypatch.undo()
        out.backward(g if fast else g.float())
        res = {k: p.grad.detach().float().cpu().clone() for k, p in params.items()}
        res["out"], res["x"] = out.detach().float().cpu(), hn.grad.float().cpu()
        return res

    fast, ref = run(True), run(False)
    for k in ref:
        rel = float((fast[k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-20))
        assert rel < 3e-2, (k, rel)


def test_mixer_layer_d512_runs_without_a_library_gemm(backend, monkeypatch):
    """configs[4]'s width (d_model 512: E = 1024, dt_rank 32, x_proj 64 x 1024), one weight-tied BiMamba mixer layer, bf16, forward + backward:
    EVERY dense product of the hand-scheduled path is an own kernel -- in_proj / d(y) / out_proj / d(x2d) / dW_in / dW_out on cad_gemm_stream,
    x_proj as two K halves of cad_proj_wx, the thin products and their weight gradients on the cad_proj_wx family -- held by making the
    torch matrix products raise while it runs; results against the same layer on the generic fp32 path."""
    from caduceus_amd import engine, mixer
    from caduceus_amd.mamba import Mamba
    name, dev = backend
    torch.manual_seed(11)
    D, Lq = 512, 256
    mf, mr = Mamba(D, device=dev), Mamba(D, device=dev)
    mr.in_proj.weight = mf.in_proj.weight
    mr.out_proj.weight = mf.out_proj.weight
    assert mf.d_inner == 1024 and mf.dt_rank == 32
    hn0 = torch.randn(2, 1, Lq, D, device=dev).to(torch.bfloat16)
    g = torch.randn(2, 1, Lq, D, device=dev).to(torch.bfloat16)
    params = {tag + "." + k: v for m, tag in ((mf, "f"), (mr, "r")) for k, v in m.named_parameters()}

    def boom(*a, **k):
        raise AssertionError("a library matrix product on the d_model 512 mixer path")

    def run(fast):
        for p in params.values():
            p.grad = None
        hn = (hn0 if fast else hn0.float()).clone().requires_grad_(True)
        if fast:
            assert mixer.can_use(mf, mr, "add")
            with monkeypatch.context() as mp:
                for fn in ("mm", "bmm", "addmm", "matmul", "baddbmm", "einsum"):
                    mp.setattr(torch, fn, boom)
                mp.setattr(torch.Tensor, "__matmul__", boom)
                mp.setattr(torch.nn.functional, "linear", boom)
                out = mixer.bimamba_mixer(hn, mf, mr, 1)
                out.backward(g)
        else:
            with monkeypatch.context() as mp:
                mp.setattr(mixer, "can_use", lambda *a, **k: False)
                out = engine._bimamba_tframe(hn, mf, mr, "add", True)
            out.backward(g.float())
        res = {k: p.grad.detach().float().cpu().clone() for k, p in params.items()}
        res["out"], res["x"] = out.detach().float().cpu(), hn.grad.float().cpu()
        return res

    fast, ref = run(True), run(False)
    for k in ref:
        rel = float((fast[k] - ref[k]).norm() / ref[k].norm().clamp_min(1e-20))
        assert rel < 3e-2, (k, rel)


@pytest.mark.parametrize("M,K,T", [(512, 512, 512), (256, 64, 768), (1024, 512, 256)])
def test_gemm_out_t_as_a_projection_with_column_tiles_fastest(backend, M, K, T):
    """cad_gemm_stream / CAD_GEMM_OUT_T_BF16 with col_fastest = 1 as the d_model 512 in_proj / d(y): out (M, T) channel-major =
    W (M, K) . X (T, K)^T computed as (X @ W^T)^T with BOTH operands streamed -- against the fp32 product, against the W-stationary
    cad_proj_wxT on the same operands (same products, fp32 accumulation: within one bf16 rounding), identical to the row-tiles-fastest
    placement (placement only), and position-independent (the RC-equivariance property of the t-frame)."""
    name, dev = backend
    W, X = _bf(M, K, seed=51), _bf(T, K, seed=52)
    Wt = W.t().contiguous()
    out = ops.gemm_out_t(X.to(dev), Wt.to(dev))
    assert out is not None and out.shape == (M, T) and out.dtype == torch.bfloat16
    ref = W.float() @ X.float().t()
    torch.testing.assert_close(out.float().cpu(), ref, rtol=1e-2, atol=1e-2 * float(ref.abs().max()) / 8)
    assert torch.equal(out.cpu(), ops.proj_xTw_stream(X.to(dev), Wt.to(dev), col_fastest=False).cpu())
    if ops.proj_supported(X.to(dev), K):
        wxT = ops.proj_wxT(W.to(dev), X.to(dev))
        torch.testing.assert_close(out.float().cpu(), wxT.float().cpu(), rtol=2e-2, atol=2e-2 * float(ref.abs().max()) / 8)
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(6))
    out_p = ops.gemm_out_t(X[perm].contiguous().to(dev), Wt.to(dev))
    assert torch.equal(out.cpu()[:, perm], out_p.cpu())


# ---- cad_gemm_f32: the fp32 path's dense products on the fp32 matrix core (csrc/gemm_f32.hip) ---------------------------------------
def _f32(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g)


@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (40, 70, 128), (128, 128, 16), (130, 257, 33), (512, 96, 8), (48, 300, 256), (200, 64, 1000)])
@pytest.mark.parametrize("ta,tb", [(False, False), (True, False), (False, True), (True, True)])
def test_gemm_f32_matches_fp64_product(backend, M, N, K, ta, tb):
    """Every combination of plain / transposed operand views (the mixer's W . X^T, X^T . W^T, Y . X^T), ragged sizes: tile edges in all
    three dimensions.  fp32 products accumulated in fp32: the bound is the fp32 class of the reference (rtol 6e-4 is ITS parity bound;
    measured ~1e-6 relative to the result's scale)."""
    name, dev = backend
    a = (_f32(K, M, seed=1).t() if ta else _f32(M, K, seed=1))
    b = (_f32(N, K, seed=2).t() if tb else _f32(K, N, seed=2))
    out = ops.mm_f32(a.to(dev) if not ta else a.t().contiguous().to(dev).t(), b.to(dev) if not tb else b.t().contiguous().to(dev).t())
    ref = a.double() @ b.double()
    assert out.shape == (M, N) and out.dtype == torch.float32
    scale = float(ref.abs().max()) + 1e-30
    assert float((out.cpu().double() - ref).abs().max()) / scale < 2e-6 * max(1.0, K ** 0.5 / 8)


@pytest.mark.parametrize("ta,tb", [(False, False), (True, True)])
def test_gemm_f32_large_tile_configuration(backend, ta, tb):
    """Enough 128 x 128 tiles to fill the chip (>= 2 per CU): the launcher takes the 4 x 4-tiles-per-wave configuration -- ragged edges in
    every dimension, plain and transposed views."""
    name, dev = backend
    M, N, K = 2048 + 40, 4096 + 5, 21
    a = (_f32(K, M, seed=8).t() if ta else _f32(M, K, seed=8))
    b = (_f32(N, K, seed=9).t() if tb else _f32(K, N, seed=9))
    ad = a.t().contiguous().to(dev).t() if ta else a.to(dev)
    bd = b.t().contiguous().to(dev).t() if tb else b.to(dev)
    out = ops.mm_f32(ad, bd)
    ref = a.double() @ b.double()
    assert float((out.cpu().double() - ref).abs().max()) / float(ref.abs().max()) < 2e-6


def test_gemm_f32_addend_out_views_and_batches(backend):
    name, dev = backend
    M, N, K, n = 70, 90, 50, 3
    a, b, c = _f32(M, K, seed=3), _f32(K, N, seed=4), _f32(M, N, seed=5)
    # addend + a padded out view (token-major column slice): the bytes outside the view stay untouched
    buf = torch.full((M, N + 6), 7.0, device=dev)
    add = torch.zeros((M, N + 6), device=dev)
    add[:, :N] = c.to(dev)
    ops.mm_f32(a.to(dev), b.to(dev), out=buf[:, :N], addend=add[:, :N])
    torch.testing.assert_close(buf[:, :N].cpu(), c + a @ b, rtol=1e-5, atol=1e-4)
    assert float((buf[:, N:] - 7.0).abs().max()) == 0.0
    # in-place accumulate (addend is out)
    acc = c.clone().to(dev)
    ops.mm_f32(a.to(dev), b.to(dev), out=acc, addend=acc)
    torch.testing.assert_close(acc.cpu(), c + a @ b, rtol=1e-5, atol=1e-4)
    # a transposed OUT view: D^T written through strides
    outT = torch.empty(N, M, device=dev)
    ops.mm_f32(a.to(dev), b.to(dev), out=outT.t())
    torch.testing.assert_close(outT.cpu(), (a @ b).t(), rtol=1e-5, atol=1e-4)
    # batches = the K slices of a weight gradient: (n, M, Kc) permuted views of channel-major operands, as mixer._wgrad_cm_cm builds them
    Kc = 64
    y, x = _f32(M, n * Kc, seed=6), _f32(N, n * Kc, seed=7)
    part = ops.bmm_f32(y.to(dev).view(M, n, Kc).permute(1, 0, 2), x.to(dev).view(N, n, Kc).permute(1, 2, 0))
    assert part.shape == (n, M, N)
    torch.testing.assert_close(part.sum(0).cpu(), y @ x.t(), rtol=1e-5, atol=2e-4)
    with pytest.raises(ValueError):
        ops.mm_f32(a.to(dev).to(torch.bfloat16), b.to(dev))
    with pytest.raises(ValueError):
        ops.mm_f32(a.to(dev), b.to(dev)[:-1])
