"""Parity at BASELINE.json's full sequence length (L = 131072, configs[2]) on the MI355X, through the C-ABI.
The torch oracle is too slow there, so these tests use (i) the C/OpenMP oracle on a narrow channel slice, which is exact
for everything except the cross-channel sums dB/dC -- those are checked with E small enough to run all channels --
and (ii) size-independent properties: mirror exactness, linearity in u, RC-equivariance of the whole model."""
import pytest
import torch

from caduceus_amd import ops
from oracle import oracle_ops

pytestmark = pytest.mark.gpu
L_FULL = 131072
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _real_library():
    from caduceus_amd import _lib
    _lib.use_library_for_testing(None)
    assert torch.cuda.is_available() and _lib.is_device_build()
    yield


def _inputs(E, SB, L, N, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g)
    t = dict(u=r(E, SB, L), delta=0.5 * r(E, SB, L), A=-(0.5 + 15.5 * torch.rand(E, N, generator=g)), B=r(N, SB, L),
             C=r(N, SB, L), D=r(E), z=r(E, SB, L), bias=r(E) - 4.0)
    for k in ("u", "delta", "B", "C", "z"):
        t[k] = t[k].to(dtype).float()
    return t


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scan_full_length_vs_c_oracle(dtype):
    """E = 16 channels, 2 rows (one per direction), L = 131072: forward and every gradient against oracle/cad_oracle.c."""
    E, SB, N = 16, 2, 16
    t = _inputs(E, SB, L_FULL, N, dtype)
    order = ("u", "delta", "A", "B", "C", "D", "z", "bias")
    act = {"u", "delta", "B", "C", "z"}
    ins = [t[k].to(DEV).to(dtype if k in act else torch.float32).requires_grad_(True) for k in order]
    out = ops.selective_scan(*ins, 1, 0, 1)
    w = torch.randn(E, SB, L_FULL, generator=torch.Generator().manual_seed(9))
    (out.float() * w.to(DEV)).sum().backward()
    refs = [t[k].clone().requires_grad_(True) for k in order]
    u, d, A, B, C, D, z, b = refs
    rows = []
    for sb in range(SB):  # row 1 runs right-to-left: explicit flips on the oracle side
        f = (lambda x: x.flip(-1)) if sb == 1 else (lambda x: x)
        bm = lambda x: f(x[:, sb]).unsqueeze(0)  # (1, C, L) batch-major
        rows.append(f(oracle_ops.selective_scan_c(bm(u), bm(d), A, bm(B), bm(C), D, bm(z), b)[0]))
    ref = torch.stack(rows, 1)
    (ref * w).sum().backward()
    tol = dict(rtol=6e-4, atol=2e-3) if dtype == torch.float32 else dict(rtol=3e-2, atol=5e-2)
    torch.testing.assert_close(out.float().cpu(), ref.detach(), **tol)
    for k, a, r in zip(order, ins, refs):
        scale = max(1.0, float(r.grad.abs().max()))
        if dtype == torch.bfloat16 and k in ("A", "D", "bias"):
            scale *= 4  # sums of 2.6e5 bf16-rounded terms
        torch.testing.assert_close(a.grad.float().cpu(), r.grad, rtol=tol["rtol"], atol=tol["atol"] * scale,
                                   msg=lambda m, k=k: f"d{k}: {m}")


def test_scan_full_size_properties():
    """configs[2] layer shape (E = 512, 2 rows, L = 131072, N = 16, bf16): mirror exactness and linearity in u."""
    E, SB, N, dtype = 512, 2, 16, torch.bfloat16
    t = _inputs(E, SB, L_FULL, N, dtype, seed=1)
    dv = {k: v.to(DEV) for k, v in t.items()}
    a = {k: dv[k].to(dtype) for k in ("u", "delta", "B", "C", "z")}
    f = lambda x: x.flip(-1).contiguous()
    fwd = ops.selective_scan(a["u"], a["delta"], dv["A"], a["B"], a["C"], dv["D"], a["z"], dv["bias"], SB, 0, 0)
    rev = ops.selective_scan(f(a["u"]), f(a["delta"]), dv["A"], f(a["B"]), f(a["C"]), dv["D"], f(a["z"]), dv["bias"], SB, 1, 1)
    assert torch.equal(fwd, f(rev))
    assert torch.isfinite(fwd.float()).all()
    # linear in u for fixed (delta, B, C, z): scan(2u) == 2 scan(u) exactly (power-of-two scaling commutes with rounding)
    two = ops.selective_scan(2 * a["u"], a["delta"], dv["A"], a["B"], a["C"], dv["D"], a["z"], dv["bias"], SB, 0, 0)
    assert torch.equal(two, 2 * fwd)
    # additivity: inputs on a 1/8 grid so that u1 + u2 is exact in bf16 and only the output rounding differs
    q = lambda x: (x.float().clamp(-4, 4) * 8).round() / 8
    u1, u2 = q(a["u"]).to(dtype), q(torch.randn(E, SB, L_FULL, device=DEV)).to(dtype)
    run = lambda u: ops.selective_scan(u, a["delta"], dv["A"], a["B"], a["C"], dv["D"], a["z"], dv["bias"], SB, 0, 0).float()
    s1, s2, s12 = run(u1), run(u2), run(u1 + u2)
    torch.testing.assert_close(s12, s1 + s2, rtol=3e-2, atol=3e-2 * max(1.0, float(s12.abs().max()) / 8))


def test_model_full_length_rc_equivariance():
    """Caduceus-PS d_model 256, seqlen 131072 (2 layers): logits of the reverse-complement input are the RC of the
    logits, bit-exact, under bf16 autocast; the MLM loss is identical (caduceus/tests/test_rcps.py properties)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import COMP, make_config, synthetic_batch
    from caduceus_amd import CaduceusForMaskedLM
    torch.manual_seed(0)
    model = CaduceusForMaskedLM(make_config(256, 2)).to(DEV).eval()
    ids, labels = synthetic_batch(torch.Generator().manual_seed(3), 1, L_FULL, DEV)
    comp = torch.tensor([COMP.get(i, i) for i in range(16)], device=DEV)
    rc = lambda x: comp[x.flip(-1)]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a = model(ids, labels=labels)
        b = model(rc(ids), labels=rc(labels))
    assert torch.isfinite(a.logits).all()
    assert torch.equal(a.logits, b.logits.flip(1)[..., comp])
    assert abs(float(a.loss) - float(b.loss)) < 1e-4 * abs(float(a.loss))  # same terms, summed in the opposite order
