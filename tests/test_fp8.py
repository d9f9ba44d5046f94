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
