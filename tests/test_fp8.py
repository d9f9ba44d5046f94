"""BASELINE configs[4] "fp8 MFMA projections" at the model level: the in_proj of every mixer on the e4m3 matrix-core kernel
(caduceus_amd/csrc/gemm_fp8.hip) inside the otherwise unchanged bf16 path."""
import pytest
import torch

from caduceus_amd import CaduceusConfig, CaduceusForMaskedLM, mixer

COMP = {7: 10, 8: 9, 9: 8, 10: 7}


def _model(dev):
    torch.manual_seed(3)
    cfg = CaduceusConfig(d_model=256, n_layer=1, vocab_size=12, bidirectional=True, bidirectional_strategy="add",
                         bidirectional_weight_tie=True, rcps=True, complement_map={i: COMP.get(i, i) for i in range(12)},
                         ssm_cfg=dict(d_state=16, d_conv=4, expand=2), fused_add_norm=True, rms_norm=True, pad_token_id=4)
    return CaduceusForMaskedLM(cfg).to(dev).train()


def test_fp8_in_proj_in_the_model(backend):
    """One PS layer, d_model 256, bf16 autocast: with the fp8 in_proj the logits stay within the stated fp8 tolerance of the bf16
    run, RC-equivariance stays BIT-exact (per-token scales: a token's projection does not depend on its position or strand), and
    the backward (which keeps the bf16 activations) yields finite gradients close to the bf16 run's."""
    name, dev = backend
    model = _model(dev)
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(7, 11, (1, 64), generator=g).to(dev)  # T = 2 x 64 = 128 tokens: the fused d(dt_lr) + dW_dt kernel is on the path
    labels = ids.clone()
    comp = model.lm_head.complement_map  # (vocabulary padded to a multiple of 8)

    def run(x, y):
        model.zero_grad(set_to_none=True)
        with torch.autocast(dev.type, dtype=torch.bfloat16):
            out = model(x, labels=y)
        out.loss.backward()
        return out, {k: p.grad.detach().float().clone() for k, p in model.named_parameters()}

    base, gbase = run(ids, labels)
    try:
        mixer.set_fp8_in_proj(True)
        a, ga = run(ids, labels)
        b, _ = run(comp[ids.flip(-1)], comp[labels.flip(-1)])
    finally:
        mixer.set_fp8_in_proj(False)
    assert torch.equal(a.logits, b.logits.flip(1)[..., comp])
    rel = float(((a.logits - base.logits).norm() / base.logits.norm()).detach())
    assert 0 < rel < 0.1, rel  # > 0: the fp8 kernel really ran
    for k in ga:
        assert torch.isfinite(ga[k]).all(), k
    k = "caduceus.backbone.layers.0.mixer.submodule.mamba_fwd.out_proj.weight"
    e = float((ga[k] - gbase[k]).norm() / gbase[k].norm())
    assert e < 0.2, e


@pytest.mark.parametrize("D,swap", [(256, True), (512, True), (256, False)])
def test_add_norm_writes_the_e4m3_operand_of_the_in_proj(backend, D, swap):
    """configs[4]: the e4m3 activations + per-token scales of the fp8 in_proj are an EPILOGUE of add + norm (cad_add_norm_args.y_fp8 /
    y_scale) -- bit-identical to cad_quant_rows_fp8 of the stored bf16 tensor, strand swap included, with no pass of their own."""
    from caduceus_amd import ops
    name, dev = backend
    g = torch.Generator().manual_seed(5)
    S, rows = (2, 37) if swap else (1, 50)
    x = torch.randn(S, 1, rows, D, generator=g).to(dev).to(torch.bfloat16)
    x[0, 0, 3] = 0  # an all-zero row (scale 1) ...
    res = torch.randn(S, 1, rows, D, generator=g).to(dev)
    res[0, 0, 3] = 0
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(dev)
    try:
        mixer.set_fp8_in_proj(True)
        y, r = ops.add_norm(x, res, w, None, 1e-5, True, swap, torch.bfloat16, want_fp8=True)
        y1, _ = ops.add_norm(x, res, w, None, 1e-5, True, swap, torch.bfloat16)  # not asked for (final norm, un-fused wrapper): not written
    finally:
        mixer.set_fp8_in_proj(False)
    y0, r0 = ops.add_norm(x, res, w, None, 1e-5, True, swap, torch.bfloat16, want_fp8=True)  # (fp8 projection off: not written either)
    assert torch.equal(y, y0) and torch.equal(r, r0) and not hasattr(y0, "_cad_fp8") and not hasattr(y1, "_cad_fp8")
    yq, ys = ops.fp8_operand_of(y)
    q_ref, s_ref = ops.quant_rows_fp8(y.reshape(-1, D))
    assert torch.equal(yq.reshape(-1, D), q_ref) and torch.equal(ys, s_ref)
    assert float(ys.min()) > 0
    # the copy is tied to the contents it was made from: an in-place write to the bf16 tensor (a hook, in-place dropout) retires it, and
    # the mixer then quantises the current contents itself
    y.mul_(2.0)
    assert ops.fp8_operand_of(y) is None


def test_fp8_model_step_takes_the_producer_written_operand(backend, monkeypatch):
    """With the fp8 in_proj on, no separate quantisation pass runs in the model: every mixer picks up what add + norm wrote."""
    from caduceus_amd import ops
    name, dev = backend
    model = _model(dev)
    ids = torch.randint(7, 11, (1, 64), generator=torch.Generator().manual_seed(2)).to(dev)
    calls = []
    real = ops.proj_wxT_fp8
    monkeypatch.setattr(ops, "quant_rows_fp8", lambda *a, **k: (_ for _ in ()).throw(AssertionError("separate quantisation pass")))
    monkeypatch.setattr(ops, "proj_wxT_fp8", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    try:
        mixer.set_fp8_in_proj(True)
        with torch.autocast(dev.type, dtype=torch.bfloat16):
            out = model(ids, labels=ids)
        out.loss.backward()
    finally:
        mixer.set_fp8_in_proj(False)
    assert len(calls) == 1 and torch.isfinite(out.loss)
