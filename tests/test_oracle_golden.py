"""The CPU oracle (oracle/oracle_model.py) must reproduce every golden vector generated from the reference's own
classes (oracle/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, MODEL_VARIANTS, load_golden_model
from oracle import oracle_model as om

TOL = dict(rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("name", MODEL_VARIANTS)
def test_oracle_model_matches_reference(name):
    cfg, sd, rec = load_golden_model(name)
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    # tied tensors are stored under every alias in the reference state dict: tie them again
    head, emb = (("lm_head.lm_head.weight", "caduceus.backbone.embeddings.word_embeddings.embedding.weight")
                 if cfg["rcps"] else ("lm_head.weight", "caduceus.backbone.embeddings.word_embeddings.weight"))
    if torch.equal(sd[head], sd[emb]):  # (transformers 5.x did not tie the plain-Linear Ph head in the harness)
        sd[head] = sd[emb]
    if cfg.get("bidirectional", True) and cfg.get("bidirectional_weight_tie", True):
        for k in list(sd):
            if ".mamba_rev.in_proj." in k or ".mamba_rev.out_proj." in k:
                sd[k] = sd[k.replace(".mamba_rev.", ".mamba_fwd.")]
    out = om.masked_lm_forward(sd, rec["input_ids"], cfg, labels=rec["labels"], ignore_index=4, collect=True)
    torch.testing.assert_close(out["logits"], rec["logits"], **TOL)
    torch.testing.assert_close(out["hidden"], rec["hidden"], **TOL)
    torch.testing.assert_close(out["loss"], rec["loss"], **TOL)
    for i, (h, r) in enumerate(out["trace"]):
        if f"trace/{i}/hidden" in rec:
            torch.testing.assert_close(h, rec[f"trace/{i}/hidden"], **TOL)
            torch.testing.assert_close(r, rec[f"trace/{i}/residual"], **TOL)
    out["loss"].backward()
    seen = 0
    for k, g in rec.items():
        if not k.startswith("grad/"):
            continue
        got = sd[k[5:]].grad
        assert got is not None, k
        scale = max(1.0, float(g.abs().max()))
        torch.testing.assert_close(got, g, rtol=2e-4, atol=5e-5 * scale, msg=lambda m, k=k: f"{k}: {m}")
        seen += 1
    assert seen > 10


@pytest.mark.parametrize("shape", ["1x64x64x16", "2x32x200x16", "1x16x37x8"])
def test_oracle_scan_matches_third_party(shape):
    z = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, f"scan_op_{shape}.npz")).items()}
    ins = {k: z[k].clone().requires_grad_(True) for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")}
    out = om.selective_scan(ins["u"], ins["delta"], ins["A"], ins["B"], ins["C"], ins["D"], ins["z"],
                            ins["delta_bias"])
    torch.testing.assert_close(out, z["out"], **TOL)
    (out * z["dout"]).sum().backward()
    for k in ins:
        g = z["d" + k]
        torch.testing.assert_close(ins[k].grad, g, rtol=2e-4, atol=5e-5 * max(1.0, float(g.abs().max())))


def test_oracle_conv_matches_third_party():
    z = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "conv_op.npz")).items()}
    x, w, b = (z[k].clone().requires_grad_(True) for k in ("x", "w", "b"))
    out = om.causal_conv1d_silu(x, w, b)
    torch.testing.assert_close(out, z["out"], **TOL)
    (out * z["dout"]).sum().backward()
    torch.testing.assert_close(x.grad, z["dx"], **TOL)
    torch.testing.assert_close(w.grad, z["dw"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(b.grad, z["db"], rtol=1e-4, atol=1e-4)


def test_reference_equivariance_vectors():
    """The reference's RC-equivariance (caduceus/tests/test_rcps.py:341-419) holds exactly (0.0) on CPU fp32 for the
    committed vectors, and the oracle reproduces both sides."""
    z = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "equivariance.npz")).items()}
    comp = z["comp"]
    back = torch.flip(z["logits_rc"][..., comp], dims=[1])
    assert torch.equal(z["logits"], back)
    cfg, sd, _ = load_golden_model("ps_fused")
    a = om.masked_lm_forward(sd, z["input_ids"], cfg)["logits"]
    b = om.masked_lm_forward(sd, z["rc_input_ids"], cfg)["logits"]
    torch.testing.assert_close(a, z["logits"], **TOL)
    assert torch.equal(a, torch.flip(b[..., comp], dims=[1]))
    assert torch.equal(om.rc_ids(z["input_ids"], comp), z["rc_input_ids"])


@pytest.mark.parametrize("shape", ["1x64x64x16", "2x32x200x16", "1x16x37x8"])
def test_c_oracle_scan_matches_third_party(shape):
    """oracle/cad_oracle.c (used for large sizes and as the cpu_baseline port) against the same committed vectors."""
    from oracle import oracle_ops
    z = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, f"scan_op_{shape}.npz")).items()}
    ins = {k: z[k].clone().requires_grad_(True) for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias")}
    out = oracle_ops.selective_scan_c(ins["u"], ins["delta"], ins["A"], ins["B"], ins["C"], ins["D"], ins["z"],
                                      ins["delta_bias"])
    torch.testing.assert_close(out, z["out"], **TOL)
    (out * z["dout"]).sum().backward()
    for k in ins:
        g = z["d" + k]
        torch.testing.assert_close(ins[k].grad, g, rtol=2e-4, atol=5e-5 * max(1.0, float(g.abs().max())))


def test_c_oracle_conv_matches_third_party():
    from oracle import oracle_ops
    z = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "conv_op.npz")).items()}
    torch.testing.assert_close(oracle_ops.conv_fwd_c(z["x"], z["w"], z["b"]), z["out"], **TOL)
