"""Per-kernel parity of the C-ABI ops against the CPU oracle, on the host emulator (CPU) and on the GPU (-m gpu).
Edge cases follow the reference tests' spirit: ragged / tiny / multi-chunk lengths, both directions, odd d_state."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from caduceus_amd import ops
from oracle import oracle_model as om

def leaf(t, dev, dtype=None):
    """Fresh leaf copy on the backend device (never aliases the CPU master tensor)."""
    t = t.detach().clone().to(dev)
    return (t if dtype is None else t.to(dtype)).requires_grad_(True)


FP32 = dict(rtol=6e-4, atol=2e-3)   # reference tolerance triple (caduceus/tests/test_rcps.py:34-36)
BF16 = dict(rtol=3e-2, atol=5e-2)


def _rows_oracle(fn, tensors_rowdim1, split, rev_lo, rev_hi):
    """Apply a (1, C, L)-shaped oracle `fn` row by row with the row's direction realised by explicit flips."""
    SB = tensors_rowdim1[0].shape[1]
    outs = []
    for sb in range(SB):
        rev = rev_lo if sb < split else rev_hi
        f = (lambda t: t.flip(-1)) if rev else (lambda t: t)
        outs.append(f(fn(*[f(t[:, sb]).unsqueeze(0) for t in tensors_rowdim1])[0]))
    return torch.stack(outs, 1)


SCAN_CASES = [  # E, SB, L, N, split, rev_lo, rev_hi
    (4, 1, 64, 16, 1, 0, 0),
    (6, 2, 100, 16, 1, 0, 1),
    (3, 2, 37, 8, 1, 1, 0),
    (5, 1, 1100, 16, 0, 0, 1),
    (2, 1, 1, 3, 1, 0, 0),
    (9, 3, 1040, 16, 2, 1, 0),  # > 8 channels, three rows, forward chunks 1024 + 16, backward chunks 512 + 512 + 16
]


def _scan_inputs(E, SB, L, N, seed, device, dtype):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    t = dict(u=r(E, SB, L), delta=r(E, SB, L), A=-(0.5 + 15.5 * torch.rand(E, N, generator=g)), B=r(N, SB, L),
             C=r(N, SB, L), D=r(E), z=r(E, SB, L), bias=r(E) - 3.0, w=r(E, SB, L))
    for k in ("u", "delta", "B", "C", "z", "w"):
        t[k] = t[k].to(dtype).float()  # oracle sees exactly the values the kernel sees
    return t


@pytest.mark.parametrize("case", SCAN_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_selective_scan_fwd_bwd(backend, case, dtype):
    name, dev = backend
    E, SB, L, N, split, rl, rh = case
    t = _scan_inputs(E, SB, L, N, 11, dev, dtype)
    order = ("u", "delta", "A", "B", "C", "D", "z", "bias")
    act = {"u", "delta", "B", "C", "z"}
    ins = [leaf(t[k], dev, dtype if k in act else torch.float32) for k in order]
    out = ops.selective_scan(*ins, split, rl, rh)
    (out.float() * t["w"].to(dev)).sum().backward()
    ref_ins = [leaf(t[k], 'cpu') for k in order]
    u, d, A, B, C, D, z, b = ref_ins
    ref = _rows_oracle(lambda u_, d_, B_, C_, z_: om.selective_scan(u_, d_, A, B_, C_, D, z_, b), [u, d, B, C, z],
                       split, rl, rh)
    (ref * t["w"]).sum().backward()
    tol = FP32 if dtype == torch.float32 else BF16
    torch.testing.assert_close(out.float().cpu(), ref.detach(), **tol)
    for k, a, r in zip(order, ins, ref_ins):
        scale = max(1.0, float(r.grad.abs().max()))
        torch.testing.assert_close(a.grad.float().cpu(), r.grad, rtol=tol["rtol"], atol=tol["atol"] * scale,
                                   msg=lambda m, k=k: f"d{k}: {m}")


@pytest.mark.parametrize("shape", ["1x64x64x16", "2x32x200x16", "1x16x37x8"])
def test_selective_scan_golden(backend, shape, golden_dir):
    """Against the committed third-party (HF mamba torch path) vectors, batch-major -> channel-major."""
    name, dev = backend
    z = {k: torch.from_numpy(v) for k, v in np.load(f"{golden_dir}/scan_op_{shape}.npz").items()}
    cm = lambda t: t.permute(1, 0, 2).contiguous()
    ins = [cm(z["u"]), cm(z["delta"]), z["A"], cm(z["B"]), cm(z["C"]), z["D"], cm(z["z"]), z["delta_bias"]]
    ins = [leaf(t, dev) for t in ins]
    out = ops.selective_scan(*ins, ins[0].shape[1], 0, 0)
    torch.testing.assert_close(cm(out.cpu().detach()), z["out"], **FP32)
    (out * cm(z["dout"]).to(dev)).sum().backward()
    for a, k, is_cm in zip(ins, ("du", "ddelta", "dA", "dB", "dC", "dD", "dz", "ddelta_bias"),
                           (1, 1, 0, 1, 1, 0, 1, 0)):
        got = a.grad.cpu()
        got = cm(got) if is_cm else got
        torch.testing.assert_close(got, z[k], rtol=6e-4, atol=2e-3 * max(1.0, float(z[k].abs().max())))


@pytest.mark.parametrize("case", [(5, 2, 75, 4, 1, 0, 1), (3, 1, 8, 4, 1, 1, 1), (4, 3, 2100, 3, 1, 0, 1),
                                  (2, 1, 1, 4, 0, 0, 0), (6, 2, 4096, 2, 2, 1, 0), (2, 2, 17000, 4, 1, 0, 1),
                                  (3, 70, 1024, 4, 30, 0, 1), (2, 45, 200, 4, 45, 1, 0)])  # many short rows per channel
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_causal_conv1d(backend, case, dtype):
    name, dev = backend
    E, SB, L, K, split, rl, rh = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(E, SB, L, generator=g).to(dtype).float()
    w = 0.5 * torch.randn(E, 1, K, generator=g)
    b = 0.2 * torch.randn(E, generator=g)
    dout = torch.randn(E, SB, L, generator=g)
    ins = [leaf(x, dev, dtype), leaf(w, dev), leaf(b, dev)]
    out = ops.causal_conv1d(*ins, split, rl, rh)
    (out.float() * dout.to(dev)).sum().backward()
    rx, rw, rb = leaf(x, 'cpu'), leaf(w, 'cpu'), leaf(b, 'cpu')
    ref = _rows_oracle(lambda x_: om.causal_conv1d_silu(x_, rw.squeeze(1), rb), [rx], split, rl, rh)
    (ref * dout).sum().backward()
    tol = FP32 if dtype == torch.float32 else BF16
    torch.testing.assert_close(out.float().cpu(), ref.detach(), **tol)
    torch.testing.assert_close(ins[0].grad.float().cpu(), rx.grad, **tol)
    torch.testing.assert_close(ins[1].grad.cpu(), rw.grad, rtol=tol["rtol"], atol=tol["atol"] * max(1, L / 64))
    torch.testing.assert_close(ins[2].grad.cpu(), rb.grad, rtol=tol["rtol"], atol=tol["atol"] * max(1, L / 64))


def test_causal_conv1d_golden(backend, golden_dir):
    name, dev = backend
    z = {k: torch.from_numpy(v) for k, v in np.load(f"{golden_dir}/conv_op.npz").items()}
    cm = lambda t: t.permute(1, 0, 2).contiguous()
    x, w, b = leaf(cm(z["x"]), dev), leaf(z["w"].unsqueeze(1), dev), leaf(z["b"], dev)
    out = ops.causal_conv1d(x, w, b, x.shape[1], 0, 0)
    torch.testing.assert_close(cm(out.detach().cpu()), z["out"], **FP32)
    (out * cm(z["dout"]).to(dev)).sum().backward()
    torch.testing.assert_close(cm(x.grad.cpu()), z["dx"], **FP32)
    torch.testing.assert_close(w.grad.cpu().squeeze(1), z["dw"], **FP32)
    torch.testing.assert_close(b.grad.cpu(), z["db"], **FP32)


def _ref_add_norm(x, res, w, b, eps, is_rms, swap_flip):
    """Reference-frame restatement on the t-frame tensors: strands swap and channels flip when `swap_flip`."""
    y, s = om.add_norm(x, w if not swap_flip else w, b, res, eps, is_rms)
    if swap_flip:
        # kernel: out[1-s][..., D-1-c] = w[D-1-c] * xhat[s][..., c]
        xh, _ = om.add_norm(x, torch.ones_like(w), None, res, eps, is_rms)
        y = xh.flip(0).flip(-1) * w
        if b is not None:
            y = y + b
        s = s.flip(0).flip(-1)
    return y, s


@pytest.mark.parametrize("S,R,D", [(2, 37, 32), (1, 64, 256), (2, 5, 130), (2, 9, 512), (2, 16, 1024)])
@pytest.mark.parametrize("is_rms", [True, False])
@pytest.mark.parametrize("swap_flip", [False, True])
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (torch.float32, torch.bfloat16),
                                     (torch.bfloat16, torch.bfloat16)])
def test_add_norm(backend, S, R, D, is_rms, swap_flip, xdt, ydt):
    name, dev = backend
    if swap_flip and S == 1:
        pytest.skip("swap needs two strands")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(S, 1, R, D, generator=g).to(xdt).float()
    for has_res in (False, True):
        res = torch.randn(S, 1, R, D, generator=g) if has_res else None
        w = 1 + 0.2 * torch.randn(D, generator=g)
        b = None if is_rms else 0.1 * torch.randn(D, generator=g)
        gy, gr = torch.randn(S, 1, R, D, generator=g), torch.randn(S, 1, R, D, generator=g)
        ins = [leaf(x, dev, xdt), None if res is None else leaf(res, dev), leaf(w, dev),
               None if b is None else leaf(b, dev)]
        y, s = ops.add_norm(ins[0], ins[1], ins[2], ins[3], 1e-5, is_rms, swap_flip, ydt)
        ((y.float() * gy.to(dev)).sum() + (s * gr.to(dev)).sum()).backward()
        rins = [leaf(x, 'cpu'), None if res is None else leaf(res, 'cpu'), leaf(w, 'cpu'),
                None if b is None else leaf(b, 'cpu')]
        ry, rs = _ref_add_norm(rins[0], rins[1], rins[2], rins[3], 1e-5, is_rms, swap_flip)
        ((ry * gy).sum() + (rs * gr).sum()).backward()
        tol = FP32 if ydt == torch.float32 else BF16
        torch.testing.assert_close(y.float().cpu(), ry.detach(), **tol)
        torch.testing.assert_close(s.cpu(), rs.detach(), **FP32)
        for a, r in zip(ins, rins):
            if a is not None:
                sc = max(1.0, float(r.grad.abs().max()))
                torch.testing.assert_close(a.grad.float().cpu(), r.grad, rtol=tol["rtol"], atol=tol["atol"] * sc)


@pytest.mark.parametrize("S,R,D,xdt", [(2, 40001, 64, torch.bfloat16), (1, 300007, 30, torch.float32)])
def test_add_norm_many_rows(backend, S, R, D, xdt):
    """Row counts at which the backward walks more than its minimum of 8 rows per wave (the launcher aims at ~1024 workgroups: 19 rows
    per wave for 80002 rows, the maximum of 64 for 300007) with a ragged last workgroup; vector and scalar kernels."""
    name, dev = backend
    g = torch.Generator().manual_seed(11)
    x = torch.randn(S, 1, R, D, generator=g).to(xdt).float()
    res = torch.randn(S, 1, R, D, generator=g)
    w = 1 + 0.2 * torch.randn(D, generator=g)
    gy, gr = torch.randn(S, 1, R, D, generator=g), torch.randn(S, 1, R, D, generator=g)
    swap = S == 2
    ins = [leaf(x, dev, xdt), leaf(res, dev), leaf(w, dev)]
    y, s = ops.add_norm(ins[0], ins[1], ins[2], None, 1e-5, True, swap, xdt)
    ((y.float() * gy.to(dev)).sum() + (s * gr.to(dev)).sum()).backward()
    rins = [leaf(x, 'cpu'), leaf(res, 'cpu'), leaf(w, 'cpu')]
    ry, rs = _ref_add_norm(rins[0], rins[1], rins[2], None, 1e-5, True, swap)
    ((ry * gy).sum() + (rs * gr).sum()).backward()
    tol = FP32 if xdt == torch.float32 else BF16
    torch.testing.assert_close(y.float().cpu(), ry.detach(), **tol)
    for a, r in zip(ins[:2], rins[:2]):
        sc = max(1.0, float(r.grad.abs().max()))
        torch.testing.assert_close(a.grad.float().cpu(), r.grad, rtol=tol["rtol"], atol=tol["atol"] * sc)
    # the weight gradient sums R * S rows: tolerance relative to its size
    sc = float(rins[2].grad.abs().max())
    torch.testing.assert_close(ins[2].grad.cpu(), rins[2].grad, rtol=2e-3 if xdt == torch.float32 else 2e-2, atol=2e-3 * sc)


@pytest.mark.parametrize("n_strands", [1, 2])
@pytest.mark.parametrize("D", [48, 256, 50])
def test_embed(backend, n_strands, D):
    """D % 4 == 0: the vector kernels (a wave per token, per-wave LDS tables in the backward); D = 50: the scalar kernels."""
    name, dev = backend
    g = torch.Generator().manual_seed(0)
    V, B, L = 16, 3, 700
    ids = torch.randint(0, V, (B, L), generator=g)
    comp = torch.tensor([0, 1, 2, 3, 4, 5, 6, 10, 9, 8, 7, 11, 12, 13, 14, 15])
    W = torch.randn(V, D, generator=g)
    w = leaf(W, dev)
    out = ops.embed(ids.to(dev), w, comp.to(dev) if n_strands == 2 else None, n_strands)
    ref = torch.stack([W[ids]] + ([W[comp[ids]]] if n_strands == 2 else []), 0)
    assert torch.equal(out.cpu(), ref)  # integer index path: exact
    gout = torch.randn(out.shape, generator=g)
    (out * gout.to(dev)).sum().backward()
    rw = leaf(W, 'cpu')
    rref = torch.stack([rw[ids]] + ([rw[comp[ids]]] if n_strands == 2 else []), 0)
    (rref * gout).sum().backward()
    torch.testing.assert_close(w.grad.cpu(), rw.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("n_strands", [1, 2])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("V,D,B,L", [(16, 40, 2, 300), (16, 256, 2, 301), (12, 128, 1, 75), (16, 256, 1, 9000), (16, 512, 1, 200)])
def test_lm_head_and_loss(backend, n_strands, dtype, V, D, B, L):
    """D = 40: the general kernel; D = 128 / 256: the matrix-core kernel (fp32 MFMA, 16-token tiles -- ragged last tile, a vocabulary
    smaller than the tile, enough tiles for several per wave); D = 512: the backward as two launches over blocks of 256 channels."""
    name, dev = backend
    g = torch.Generator().manual_seed(2)
    comp = torch.tensor([0, 1, 2, 3, 4, 5, 6, 10, 9, 8, 7, 11, 12, 13, 14, 15])[:V]
    W = torch.randn(V, D, generator=g)
    h = torch.randn(n_strands, B, L, D, generator=g).to(dtype).float()
    labels = torch.randint(0, V, (B, L), generator=g)
    labels[torch.rand(B, L, generator=g) < 0.8] = 4
    hd, wd = leaf(h, dev, dtype), leaf(W, dev)
    logits, loss = ops.lm_head(hd, wd, comp.to(dev) if n_strands == 2 else None, labels.to(dev), 4)
    (loss + 0.01 * logits.square().mean()).backward()
    rh, rw = leaf(h, 'cpu'), leaf(W, 'cpu')
    rl = F.linear(rh[0], rw) + (F.linear(rh[1], rw[comp]) if n_strands == 2 else 0)
    rloss = om.cross_entropy(rl, labels, 4)
    (rloss + 0.01 * rl.square().mean()).backward()
    tol = FP32 if dtype == torch.float32 else BF16
    torch.testing.assert_close(logits.cpu(), rl.detach(), **FP32)
    torch.testing.assert_close(loss.cpu(), rloss.detach(), **FP32)
    torch.testing.assert_close(hd.grad.float().cpu(), rh.grad, **tol)
    torch.testing.assert_close(wd.grad.cpu(), rw.grad, rtol=tol["rtol"], atol=tol["atol"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(5, 2, 75, 4), (3, 3, 2100, 3), (6, 2, 4096, 4), (2, 2, 34000, 4), (2, 50, 1024, 4)])
def test_causal_conv1d_two_sets(backend, case, dtype):
    """cad_conv1d_fwd_multi / cad_conv1d_bwd_multi (two parameter sets on one x, opposite directions, dx summed) are
    identical to two single-set launches."""
    from caduceus_amd import mixer
    name, dev = backend
    E, SB, L, K = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn(E, SB, L, generator=g).to(dev).to(dtype)
    params = [(torch.randn(E, K, generator=g).to(dev), torch.randn(E, generator=g).to(dev)) for _ in range(2)]
    douts = [torch.randn(E, SB, L, generator=g).to(dev).to(dtype) for _ in range(2)]
    dirs, split = ((0, 1), (1, 0)), 1
    outs = mixer._conv_fwd2(x, params, split, dirs)
    dx = torch.empty_like(x)
    grads = mixer._conv_bwd2(x, params, douts, dx, split, dirs)
    dx_ref = torch.zeros(E, SB, L, device=dev)
    for i in range(2):
        xi = x.clone().requires_grad_(True)
        wi, bi = (p.clone().requires_grad_(True) for p in params[i])
        oi = ops.causal_conv1d(xi, wi.view(E, 1, K), bi, split, *dirs[i])
        assert torch.equal(oi, outs[i])
        oi.backward(douts[i])
        dx_ref += xi.grad.float()
        tol = FP32 if dtype == torch.float32 else BF16
        torch.testing.assert_close(grads[i][0], wi.grad.view(E, K), rtol=1e-4, atol=1e-4 * max(1.0, float(wi.grad.abs().max())))
        torch.testing.assert_close(grads[i][1], bi.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(bi.grad.abs().max())))
    torch.testing.assert_close(dx.float(), dx_ref, **(FP32 if dtype == torch.float32 else BF16))
    # the C-ABI's accumulate flag: set 0 written, set 1 added to it (one launch each)
    from caduceus_amd import _lib as CL
    dx2 = torch.empty_like(x)
    for i in range(2):
        wf, bf = params[i]
        dw, db = torch.zeros_like(wf), torch.zeros_like(bf)
        stream = CL.stream_and_check(x, wf, bf, douts[i], dx2, dw, db)
        a = CL.Conv1dBwdArgs(CL.ptr(x), CL.ptr(wf), CL.ptr(bf), CL.ptr(douts[i]), CL.ptr(dx2), CL.ptr(dw), CL.ptr(db), SB, L, split,
                             E, K, dirs[i][0], dirs[i][1], CL.dtype_code(x.dtype), i)
        CL.check(CL.get_lib().cad_conv1d_bwd(ctypes.byref(a), stream), "cad_conv1d_bwd")
        torch.testing.assert_close(dw, grads[i][0], rtol=1e-5, atol=1e-5 * max(1.0, float(dw.abs().max())))
    torch.testing.assert_close(dx2.float(), dx_ref, **(FP32 if dtype == torch.float32 else BF16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cut", [512, 700, 37])
def test_scan_state_carries_chain_segments(backend, dtype, cut):
    """h0 / hT / dhT / dh0 of cad_scan_fwd / cad_scan_bwd: a row scanned as two chained segments (each direction gets
    the segments in its own order) equals the row scanned at once, outputs and every gradient."""
    name, dev = backend
    E, SB, L, N = 5, 2, 1100, 16
    t = _scan_inputs(E, SB, L, N, 31, dev, dtype)
    order = ("u", "delta", "A", "B", "C", "D", "z", "bias")
    act = {"u", "delta", "B", "C", "z"}
    mk = lambda: [leaf(t[k], dev, dtype if k in act else torch.float32) for k in order]
    w = t["w"].to(dev)
    ref_in = mk()
    ref = ops.selective_scan(*ref_in, 1, 0, 1)          # row 0 left-to-right, row 1 right-to-left
    (ref.float() * w).sum().backward()
    ins = mk()
    u, d, A, B, C, D, z, b = ins
    seg = lambda x, s: x[..., s]
    lo, hi = slice(0, cut), slice(cut, L)
    outs = {}
    for row, rev in ((0, 0), (1, 1)):
        first, second = (lo, hi) if rev == 0 else (hi, lo)  # the segment that holds the row's logical start first
        r = slice(row, row + 1)
        f = lambda x, s: x[:, r][..., s]
        o1, h = ops.selective_scan_stateful(f(u, first), f(d, first), A, f(B, first), f(C, first), D, f(z, first), b, None,
                                            1, rev, rev)
        o2, _ = ops.selective_scan_stateful(f(u, second), f(d, second), A, f(B, second), f(C, second), D, f(z, second), b,
                                            h, 1, rev, rev)
        outs[row] = torch.cat([o1, o2], -1) if rev == 0 else torch.cat([o2, o1], -1)
    out = torch.cat([outs[0], outs[1]], 1)
    (out.float() * w).sum().backward()
    tol = FP32 if dtype == torch.float32 else BF16
    torch.testing.assert_close(out.float(), ref.float(), **tol)
    for k, a_, r_ in zip(order, ins, ref_in):
        scale = max(1.0, float(r_.grad.abs().max()))
        torch.testing.assert_close(a_.grad.float(), r_.grad.float(), rtol=tol["rtol"], atol=tol["atol"] * scale,
                                   msg=lambda m, k=k: f"d{k}: {m}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_gate_exact_zero(backend, dtype):
    """z == 0 exactly: out is 0 there, so y cannot be recovered from out / z; the fix-up launch recomputes y and adds
    dout * y / 2 (the true gate gradient).  Zeros are scattered, fill whole positions (a zero in_proj row), and sit in
    several chunks; every other gradient must be unaffected."""
    name, dev = backend
    E, SB, L, N, split, rl, rh = 5, 2, 1100, 16, 1, 0, 1
    t = _scan_inputs(E, SB, L, N, 23, dev, dtype)
    g = torch.Generator().manual_seed(5)
    t["z"][torch.rand(E, SB, L, generator=g) < 0.02] = 0.0
    t["z"][:, :, [0, 7, 511, 512, 700, L - 1]] = 0.0
    t["z"][2, 1, :] = 0.0
    order = ("u", "delta", "A", "B", "C", "D", "z", "bias")
    act = {"u", "delta", "B", "C", "z"}
    ins = [leaf(t[k], dev, dtype if k in act else torch.float32) for k in order]
    out = ops.selective_scan(*ins, split, rl, rh)
    (out.float() * t["w"].to(dev)).sum().backward()
    ref_ins = [leaf(t[k], 'cpu') for k in order]
    u, d, A, B, C, D, z, b = ref_ins
    ref = _rows_oracle(lambda u_, d_, B_, C_, z_: om.selective_scan(u_, d_, A, B_, C_, D, z_, b), [u, d, B, C, z],
                       split, rl, rh)
    (ref * t["w"]).sum().backward()
    tol = FP32 if dtype == torch.float32 else BF16
    zero = (t["z"] == 0)
    assert float(z.grad[zero].abs().max()) > 0.1  # the case is not vacuous
    for k, a, r in zip(order, ins, ref_ins):
        scale = max(1.0, float(r.grad.abs().max()))
        torch.testing.assert_close(a.grad.float().cpu(), r.grad, rtol=tol["rtol"], atol=tol["atol"] * scale,
                                   msg=lambda m, k=k: f"d{k}: {m}")
    torch.testing.assert_close(ins[6].grad.float().cpu()[zero], z.grad[zero], rtol=tol["rtol"],
                               atol=tol["atol"] * max(1.0, float(z.grad[zero].abs().max())))


@pytest.mark.parametrize("D", [40, 128, 512])
def test_lm_head_backward_without_labels(backend, D):
    """Only an upstream gradient of the logits (no loss): cad_lm_head_bwd with labels == NULL (D = 128; D = 512: two launches over blocks of
    256 channels with the row stride 512 -- configs[4]'s head) / the torch path (D = 40)."""
    name, dev = backend
    g = torch.Generator().manual_seed(4)
    S, B, L, V = 2, 1, 333, 16
    comp = torch.tensor([0, 1, 2, 3, 4, 5, 6, 10, 9, 8, 7, 11, 12, 13, 14, 15])
    h, W = torch.randn(S, B, L, D, generator=g), torch.randn(V, D, generator=g)
    up = torch.randn(B, L, V, generator=g)
    hd, wd = leaf(h, dev), leaf(W, dev)
    logits, _ = ops.lm_head(hd, wd, comp.to(dev), None, 4)
    (logits * up.to(dev)).sum().backward()
    rh, rw = leaf(h, 'cpu'), leaf(W, 'cpu')
    rl = F.linear(rh[0], rw) + F.linear(rh[1], rw[comp])
    (rl * up).sum().backward()
    torch.testing.assert_close(logits.cpu(), rl.detach(), **FP32)
    torch.testing.assert_close(hd.grad.cpu(), rh.grad, **FP32)
    torch.testing.assert_close(wd.grad.cpu(), rw.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("D", [32, 256])
def test_lm_head_loss_is_deterministic(backend, D):
    """The fused loss is a two-stage fixed-order sum: repeated runs give identical bits (general and matrix-core kernel)."""
    name, dev = backend
    g = torch.Generator().manual_seed(3)
    S, B, L, V = 2, 2, 700, 16
    hidden = torch.randn(S, B, L, D, generator=g).to(dev)
    w = torch.randn(V, D, generator=g).to(dev)
    comp = torch.tensor([0, 1, 2, 3, 4, 5, 6, 10, 9, 8, 7, 11, 12, 13, 14, 15]).to(dev)
    labels = torch.randint(0, V, (B, L), generator=g)
    labels[torch.rand(B, L, generator=g) < 0.8] = 4
    labels = labels.to(dev)
    losses = [ops.lm_head(hidden, w, comp, labels, 4)[1] for _ in range(4)]
    assert all(torch.equal(losses[0], l) for l in losses[1:])
    logits = ops.lm_head(hidden, w, comp, None, 4)[0]
    ref = F.cross_entropy(logits.reshape(-1, V).cpu().double(), labels.reshape(-1).cpu(), ignore_index=4)
    torch.testing.assert_close(losses[0].cpu().double(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n_strands", [1, 2])
def test_embed_large_vocabulary_backward(backend, n_strands):
    """V * D * 4 > 64 KB (e.g. CaduceusConfig's default vocab_size): the LDS-resident gradient kernel does not apply and
    the op scatter-adds instead; forward stays on the gather kernel."""
    name, dev = backend
    g = torch.Generator().manual_seed(9)
    V, D, B, L = 4096, 64, 2, 50
    w = torch.randn(V, D, generator=g)
    ids = torch.randint(0, V, (B, L), generator=g)
    comp = torch.randperm(V, generator=g)
    wd = leaf(w, dev)
    out = ops.embed(ids.to(dev), wd, comp.to(dev) if n_strands == 2 else None, n_strands)
    up = torch.randn(out.shape, generator=g)
    (out * up.to(dev)).sum().backward()
    wr = leaf(w, "cpu")
    ref = [wr[ids]] + ([wr[comp[ids]]] if n_strands == 2 else [])
    (torch.stack(ref) * up).sum().backward()
    assert torch.equal(out.detach().cpu(), torch.stack(ref).detach())
    torch.testing.assert_close(wd.grad.cpu(), wr.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("case", [(5, 2, 75, 4, 1, 0, 1), (8, 2, 1100, 16, 1, 1, 0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_takes_dt_computed_by_the_projection(backend, case, dtype):
    """delta_is_dt: the scans take dt = softplus(delta_raw + bias) as produced by cad_proj_wx's epilogue.  Same output, and
    the gradients are still those w.r.t. delta_raw and the bias (d dt * sigmoid(raw) = d dt * (1 - exp(-dt)))."""
    name, dev = backend
    E, SB, L, N, split, rl, rh = case
    t = _scan_inputs(E, SB, L, N, 23, dev, dtype)
    order = ("u", "delta", "A", "B", "C", "D", "z", "bias")
    act = {"u", "delta", "B", "C", "z"}
    raw = [leaf(t[k], dev, dtype if k in act else torch.float32) for k in order]
    out_raw = ops.selective_scan(*raw, split, rl, rh)
    (out_raw.float() * t["w"].to(dev)).sum().backward()
    # dt from the RAW (dtype-rounded) delta the first run saw, evaluated in fp32: in fp32 both runs see the same dt bit for bit
    dt = torch.nn.functional.softplus(raw[1].detach().float() + raw[7].detach()[:, None, None], threshold=20.0).to(dtype)
    fused = [leaf(t[k], dev, dtype if k in act else torch.float32) for k in order]
    fused[1] = dt.clone().requires_grad_(True)
    u, d, A, B, C, D, z, b = fused
    out = ops.selective_scan_multi([(u, d, A, B, C, D, b)], z, split, [(rl, rh)], delta_is_dt=True)[0]
    (out.float() * t["w"].to(dev)).sum().backward()
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(out.float(), out_raw.float(), **tol)
    for k, a, r in zip(order, fused, raw):
        scale = max(1.0, float(r.grad.abs().max()))
        torch.testing.assert_close(a.grad.float(), r.grad.float(), rtol=tol["rtol"], atol=tol["atol"] * scale,
                                   msg=lambda m, k=k: f"d{k}: {m}")


@pytest.mark.parametrize("case", [(8, 2, 1100, 16, 1, 1, 0), (16, 1, 2100, 16, 1, 0, 0)])
def test_fused_softplus_rounding_point_against_oracle(backend, case):
    """bf16: the production path rounds dt = softplus(raw + bias) to bf16 (dt_proj epilogue, delta_is_dt) where mamba-ssm rounds
    the RAW delta and evaluates softplus / sigmoid in fp32 inside the scan.  Both HIP paths are held
    to the fp32 oracle on the same bf16-rounded raw delta: output and every gradient -- d(delta), d(bias) in particular -- of the
    fused path stay inside the bf16 tolerance class AND within 1.5x the un-fused path's error norm (+ a floor), so moving the
    rounding point does not cost accuracy."""
    name, dev = backend
    E, SB, L, N, split, rl, rh = case
    dtype = torch.bfloat16
    t = _scan_inputs(E, SB, L, N, 29, dev, dtype)
    t["delta"] = (0.5 * t["delta"] - 1.0).to(dtype).float()  # raw pre-activations around the dt range of a trained model
    order = ("u", "delta", "A", "B", "C", "D", "z", "bias")
    act = {"u", "delta", "B", "C", "z"}
    ref_ins = [leaf(t[k], "cpu") for k in order]
    u, d, A, B, C, D, z, b = ref_ins
    ref = _rows_oracle(lambda u_, d_, B_, C_, z_: om.selective_scan(u_, d_, A, B_, C_, D, z_, b), [u, d, B, C, z],
                       split, rl, rh)
    (ref * t["w"]).sum().backward()
    raw = [leaf(t[k], dev, dtype if k in act else torch.float32) for k in order]
    out_raw = ops.selective_scan(*raw, split, rl, rh)
    (out_raw.float() * t["w"].to(dev)).sum().backward()
    fused = [leaf(t[k], dev, dtype if k in act else torch.float32) for k in order]
    dt = torch.nn.functional.softplus(fused[1].detach().float() + fused[7].detach()[:, None, None], threshold=20.0).to(dtype)
    fused[1] = dt.clone().requires_grad_(True)
    fu, fd, fA, fB, fC, fD, fz, fb = fused
    out = ops.selective_scan_multi([(fu, fd, fA, fB, fC, fD, fb)], fz, split, [(rl, rh)], delta_is_dt=True)[0]
    (out.float() * t["w"].to(dev)).sum().backward()
    torch.testing.assert_close(out.float().cpu(), ref.detach(), **BF16)

    def rel(x, want):
        return float((x.detach().float().cpu() - want).norm() / want.norm().clamp_min(1e-12))

    e_out_f, e_out_r = rel(out, ref.detach()), rel(out_raw, ref.detach())
    assert e_out_f < 1.5 * e_out_r + 2e-3, (e_out_f, e_out_r)
    for k, a, r0, r in zip(order, fused, raw, ref_ins):
        scale = max(1.0, float(r.grad.abs().max()))
        torch.testing.assert_close(a.grad.float().cpu(), r.grad, rtol=BF16["rtol"], atol=BF16["atol"] * scale,
                                   msg=lambda m, k=k: f"d{k} (fused): {m}")
        ef, er = rel(a.grad, r.grad), rel(r0.grad, r.grad)
        assert ef < 1.5 * er + 4e-3, (k, ef, er)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("k", [2, 4])
def test_scan_lsplit_two_pass_matches_unsplit_and_oracle(backend, monkeypatch, dtype, k):
    """L-split scans (ops.lsplit_factor / scan_fwd_launch / scan_bwd_launch): every row cut into k segments that run as rows
    of the same launch -- pass 1 with the `map_only` / `carry_only` kernel modes, composition of the segment maps in each row's
    direction, pass 2 from the true entry states.  Both parameter sets of a BiMamba layer (opposite directions, shared gate):
    output and every gradient equal the un-split launch (fp32: to summation-order rounding) and the oracle."""
    name, dev = backend
    E, SB, L, N, split = 8, 2, 4096, 16, 1
    order = ("u", "delta", "A", "B", "C", "D", "bias")
    act = {"u", "delta", "B", "C"}
    t1, t2 = _scan_inputs(E, SB, L, N, 31, dev, dtype), _scan_inputs(E, SB, L, N, 32, dev, dtype)
    dirs = [(0, 1), (1, 0)]

    def run(kk):
        monkeypatch.setenv("CADUCEUS_AMD_LSPLIT", str(kk))
        sets = [[leaf(t[n], dev, dtype if n in act else torch.float32) for n in order] for t in (t1, t2)]
        z = leaf(t1["z"], dev, dtype)
        o1, o2 = ops.selective_scan_multi([tuple(x) for x in sets], z, split, dirs)
        ((o1.float() * t1["w"].to(dev)).sum() + (o2.float() * t2["w"].to(dev)).sum()).backward()
        return (o1, o2), sets, z

    (a1, a2), sa, za = run(1)
    (b1, b2), sb_, zb = run(k)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else BF16
    torch.testing.assert_close(b1.float(), a1.float(), **tol)
    torch.testing.assert_close(b2.float(), a2.float(), **tol)
    for i in range(2):
        for n, x, y in zip(order, sb_[i], sa[i]):
            scale = max(1.0, float(y.grad.abs().max()))
            torch.testing.assert_close(x.grad.float(), y.grad.float(), rtol=tol["rtol"], atol=tol["atol"] * scale,
                                       msg=lambda m, n=n, i=i: f"set {i} d{n}: {m}")
    torch.testing.assert_close(zb.grad.float(), za.grad.float(), rtol=tol["rtol"], atol=tol["atol"] * max(1.0, float(za.grad.abs().max())))
    # ... and the oracle (set 0, strand rows in their own directions)
    ref_ins = [leaf(t1[n], "cpu") for n in ("u", "delta", "A", "B", "C", "D", "z", "bias")]
    u, d, A, B, C, D, z, b = ref_ins
    ref = _rows_oracle(lambda u_, d_, B_, C_, z_: om.selective_scan(u_, d_, A, B_, C_, D, z_, b), [u, d, B, C, z], split, 0, 1)
    otol = FP32 if dtype == torch.float32 else BF16
    torch.testing.assert_close(b1.float().cpu(), ref.detach(), **otol)


def test_kernel_timer_counts_only_the_enabled_kinds(backend):
    """cad_prof_enable_kinds: launches of a kind whose bit is not set are not timed (bench.py times the two scan kinds only inside its
    timed region -- every timed launch costs two event records on the stream); cad_prof_enable(1) times every kind."""
    from caduceus_amd import _lib as CL
    name, dev = backend
    x = torch.randn(4, 2, 64).to(dev)
    w, b = torch.randn(4, 1, 4).to(dev), torch.randn(4).to(dev)
    h, nw = torch.randn(1, 1, 8, 32).to(dev), torch.ones(32).to(dev)
    CL.prof_enable(False)
    CL.prof_reset()
    try:
        CL.prof_enable(True, kinds=("conv_fwd",))
        ops.causal_conv1d(x, w, b, 2, 0, 0)
        ops.add_norm(h, None, nw, None, 1e-5, True, False, torch.float32)
        CL.prof_enable(False)
        pr = CL.prof_read()
        assert pr["conv_fwd"][1] == 1 and pr["add_norm_fwd"][1] == 0
        CL.prof_reset()
        CL.prof_enable(True)
        ops.causal_conv1d(x, w, b, 2, 0, 0)
        ops.add_norm(h, None, nw, None, 1e-5, True, False, torch.float32)
        CL.prof_enable(False)
        pr = CL.prof_read()
        assert pr["conv_fwd"][1] == 1 and pr["add_norm_fwd"][1] == 1
    finally:
        CL.prof_enable(False)
        CL.prof_reset()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,npart", [(4096, 7), (16 * 3 * 200, 64), (1030, 3)])
def test_reduce_partials_multi_equals_the_single_folds(backend, dtype, n, npart):
    """cad_reduce_partials_multi: up to four folds in one launch, the same summation order as cad_reduce_partials (bit-identical),
    including the ragged fallback (n % 4 != 0)."""
    import ctypes
    from caduceus_amd import _lib as CL
    name, dev = backend
    g = torch.Generator().manual_seed(5)
    srcs = [torch.randn(npart, n, generator=g).to(dev).to(dtype) for _ in range(4)]
    single = [torch.empty(n, dtype=dtype, device=dev) for _ in range(4)]
    multi = [torch.empty(n, dtype=dtype, device=dev) for _ in range(4)]
    stream = CL.stream_and_check(*srcs, *single, *multi)
    jobs = (CL.ReduceJob * 4)()
    for i in range(4):
        CL.check(CL.get_lib().cad_reduce_partials(CL.ptr(srcs[i]), npart, n, CL.ptr(single[i]), CL.dtype_code(dtype), stream), "single")
        jobs[i] = CL.ReduceJob(CL.ptr(srcs[i]), CL.ptr(multi[i]))
    CL.check(CL.get_lib().cad_reduce_partials_multi(jobs, 4, npart, n, CL.dtype_code(dtype), stream), "multi")
    for i in range(4):
        assert torch.equal(single[i], multi[i])
        torch.testing.assert_close(single[i].float().cpu(), srcs[i].float().sum(0).cpu(), rtol=2e-2 if dtype == torch.bfloat16 else 1e-5,
                                   atol=5e-2 if dtype == torch.bfloat16 else 1e-4)


LEAN_CASES = [  # E, SB, L, split, rev_lo, rev_hi -- bf16, d_state 16, L % 8 == 0, delta_is_dt: the production instantiation of the backward
    (9, 3, 1104, 2, 1, 0),   # two workgroups (the second with 7 padding waves), chunks 512 + 512 + 80 (lanes 10.. of the tail outside)
    (8, 2, 512, 1, 0, 1),    # exactly one chunk, both directions
    (16, 1, 8, 1, 1, 1),     # one lane of one chunk
]


@pytest.mark.parametrize("case", LEAN_CASES)
@pytest.mark.parametrize("nsets", [1, 2])
def test_scan_backward_lean_production_instantiation(backend, case, nsets):
    """The unrolled bf16 / d_state 16 / vector-path / delta_is_dt backward (scan_bwd_kernel<bf16, true, false, 8, true>: lean prologue and
    epilogue -- out-of-row lanes silenced by dt = dy = 0 instead of per-item selects, v_perm unpack, dt taken from the (dt, dt u) pairs,
    dA summed in 8-lane groups) against the fp32 oracle on the same bf16-rounded inputs: output and EVERY gradient, with a ragged tail
    chunk, padding waves, both directions, exact-zero gates, one and two parameter sets under a shared gate."""
    name, dev = backend
    E, SB, L, split, rl, rh = case
    N, dtype = 16, torch.bfloat16
    dirs = [(rl, rh), (1 - rl, 1 - rh)][:nsets]
    order = ("u", "delta", "A", "B", "C", "D", "bias")
    act = {"u", "delta", "B", "C"}
    ts = []
    for i in range(nsets):
        t = _scan_inputs(E, SB, L, N, 31 + i, dev, dtype)
        raw = (0.5 * t["delta"] - 1.0)
        t["delta"] = torch.nn.functional.softplus(raw + t["bias"][:, None, None]).to(dtype).float()  # dt as the dt_proj epilogue stores it
        t["delta"][0, 0, : min(L, 24)] = 0.0  # dt == 0 inside the row too (softplus underflow): sigmoid'(..) = 0 there, nothing breaks
        ts.append(t)
    z = ts[0]["z"].clone()
    z[1, 0, 3] = 0.0  # exact-zero gates: the fix-up worklist path
    z[E - 1, SB - 1, L - 1] = 0.0
    w = [t["w"] for t in ts]
    # device run
    zd = leaf(z, dev, dtype)
    dsets = [tuple(leaf(t[k], dev, dtype if k in act else torch.float32) for k in order) for t in ts]
    outs = ops.selective_scan_multi(dsets, zd, split, dirs, delta_is_dt=True)
    sum((o.float() * w[i].to(dev)).sum() for i, o in enumerate(outs)).backward()
    # oracle: raw delta = softplus^-1(dt) (so that its softplus returns the very dt the kernel saw), bias 0: d raw = d dt * sigmoid(raw)
    zr = leaf(z, "cpu")
    rsets = []
    for t in ts:
        tt = dict(t)
        dtv = t["delta"].double()
        tt["delta"] = torch.where(dtv > 0, dtv + torch.log(-torch.expm1(-dtv)), torch.full_like(dtv, -200.0)).float()
        tt["bias"] = torch.zeros_like(t["bias"])
        rsets.append(tuple(leaf(tt[k], "cpu") for k in order))
    refs = []
    for i, (u, d, A, B, C, D, b) in enumerate(rsets):
        refs.append(_rows_oracle(lambda u_, d_, B_, C_, z_: om.selective_scan(u_, d_, A, B_, C_, D, z_, b), [u, d, B, C, zr],
                                 split, *dirs[i]))
    sum((r * w[i]).sum() for i, r in enumerate(refs)).backward()
    for i in range(nsets):
        torch.testing.assert_close(outs[i].float().cpu(), refs[i].detach(), **BF16)
        for k, a, r in zip(order, dsets[i], rsets[i]):
            scale = max(1.0, float(r.grad.abs().max()))
            torch.testing.assert_close(a.grad.float().cpu(), r.grad, rtol=BF16["rtol"], atol=BF16["atol"] * scale,
                                       msg=lambda m, k=k, i=i: f"set {i} d{k}: {m}")
            assert float((a.grad.float().cpu() - r.grad).norm() / r.grad.norm().clamp_min(1e-12)) < 2e-2, (i, k)
    scale = max(1.0, float(zr.grad.abs().max()))
    torch.testing.assert_close(zd.grad.float().cpu(), zr.grad, rtol=BF16["rtol"], atol=BF16["atol"] * scale)
