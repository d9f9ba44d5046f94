"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own classes.

Run in the build container only (needs /root/reference and the installed `transformers`):

    python oracle/gen_golden.py

The reference's `caduceus/*.py` hard-imports the un-vendored `mamba_ssm` package
(/root/reference/caduceus/modeling_caduceus.py:11); `oracle/ref_harness/` supplies it as an adapter around the
installed third-party HF `transformers` MambaMixer torch path.  Nothing from /root/reference is
copied: only numeric inputs / outputs / parameters are written, as compressed .npz.

Fixture contents (all fp32 unless noted):
    model_<variant>.npz : cfg (json), state-dict tensors `sd/<key>`, input_ids, labels, logits, hidden (final
                          backbone output), loss (ignore_index=4), `grad/<key>` for loss.backward(), and for
                          traced variants per-layer `trace/<i>/hidden`, `trace/<i>/residual`.
    scan_op_<shape>.npz : inputs and outputs (+ input grads) of the third-party HF `mamba_selective_scan` torch path.
    conv_op.npz         : inputs/outputs of HF `causal_conv1d_fn` torch path (+ SiLU).
    equivariance.npz    : reference model outputs on x and RC(x) (test_rcps.py-style property, CPU fp32).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_harness"))
sys.path.insert(0, "/root/reference")

import transformers  # noqa: E402
from caduceus.configuration_caduceus import CaduceusConfig  # noqa: E402  (reference)
from caduceus.modeling_caduceus import CaduceusForMaskedLM  # noqa: E402  (reference)
from transformers.models.mamba import modeling_mamba as hf_mamba  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
COMP = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5, 6: 6, 7: 10, 8: 9, 9: 8, 10: 7, 11: 11}  # tokenization_caduceus.py:49-66
SSM_CFG = dict(d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=1e-3, dt_max=0.1, dt_init="random",
               dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True)


class RefMLM(CaduceusForMaskedLM):
    """transformers 5.x passes kwargs to tie_weights; the reference (4.38) signature takes none."""

    def tie_weights(self, *a, **k):
        return CaduceusForMaskedLM.tie_weights(self)


def make_cfg(**over):
    base = dict(d_model=32, n_layer=2, vocab_size=12, ssm_cfg=dict(SSM_CFG), rms_norm=True, residual_in_fp32=False,
                fused_add_norm=True, pad_vocab_size_multiple=8, norm_epsilon=1e-5,
                initializer_cfg=dict(initializer_range=0.02, rescale_prenorm_residual=True, n_residuals_per_layer=1),
                bidirectional=True, bidirectional_strategy="add", bidirectional_weight_tie=True, rcps=True,
                complement_map=dict(COMP))
    base.update(over)
    return base


def randomize_(model, gen):
    """Move every parameter away from its (tiny / structured) init so the vectors are sensitive to every term."""
    seen = set()
    with torch.no_grad():
        for name, p in model.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            r = lambda *s: torch.randn(*s, generator=gen)
            if name.endswith("A_log"):
                p.copy_(torch.log(0.5 + 15.5 * torch.rand(p.shape, generator=gen)))
            elif name.endswith(".D"):
                p.copy_(1.0 + 0.3 * r(*p.shape))
            elif name.endswith("dt_proj.bias"):
                dt = torch.exp(torch.rand(p.shape, generator=gen) * (np.log(0.1) - np.log(1e-3)) + np.log(1e-3))
                p.copy_(dt + torch.log(-torch.expm1(-dt)))
            elif name.endswith("dt_proj.weight"):
                p.copy_(0.5 * r(*p.shape))
            elif name.endswith("conv1d.weight"):
                p.copy_(0.5 * r(*p.shape))
            elif name.endswith("conv1d.bias"):
                p.copy_(0.2 * r(*p.shape))
            elif "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * r(*p.shape))
            elif "norm" in name and name.endswith("bias"):
                p.copy_(0.1 * r(*p.shape))
            elif "embedding" in name or "word_embeddings" in name or name.startswith("lm_head"):
                p.copy_(0.7 * r(*p.shape))
            elif name.endswith("x_proj.weight"):
                p.copy_(r(*p.shape) / np.sqrt(p.shape[1]) * 1.5)
            else:  # in_proj / out_proj
                p.copy_(r(*p.shape) / np.sqrt(p.shape[1]) * 1.5)


def mlm_batch(gen, B, L):
    """Synthetic ids + MLM labels in the spirit of src/dataloaders/utils/mlm.py:4-32 (ignore label = 4)."""
    ids = torch.randint(7, 11, (B, L), generator=gen)
    ids[torch.rand(B, L, generator=gen) < 0.02] = 4  # a few [PAD] (N -> pad, hg38_dataset.py:212)
    labels = ids.clone()
    tgt = torch.rand(B, L, generator=gen) < 0.15
    labels[~tgt] = 4
    mask80 = tgt & (torch.rand(B, L, generator=gen) < 0.8)
    ids = ids.clone()
    ids[mask80] = 3
    rnd = tgt & ~mask80 & (torch.rand(B, L, generator=gen) < 0.5)
    ids[rnd] = torch.randint(0, 12, (int(rnd.sum()),), generator=gen)
    return ids, labels


def gen_model(name, cfg_dict, B=2, L=64, seed=0, trace=False):
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    cfg = CaduceusConfig(**json.loads(json.dumps(cfg_dict), object_hook=_intkeys))
    model = RefMLM(cfg)
    randomize_(model, gen)
    model.tie_weights()
    model.train()
    ids, labels = mlm_batch(gen, B, L)
    rec = {}
    traces = []
    if trace:
        for i, layer in enumerate(model.caduceus.backbone.layers):
            layer.register_forward_hook(lambda m, a, o, i=i: traces.append((i, o[0].detach(), o[1].detach())))
    out = model(ids, output_hidden_states=False, return_dict=True)
    logits = out.logits
    hidden = model.caduceus(ids, return_dict=True).last_hidden_state if not trace else None
    loss = torch.nn.functional.cross_entropy(logits.view(-1, logits.shape[-1]), labels.view(-1), ignore_index=4)
    loss.backward()
    rec["cfg"] = np.frombuffer(json.dumps(cfg_dict).encode(), dtype=np.uint8)
    rec["meta"] = np.frombuffer(json.dumps(dict(torch=torch.__version__, transformers=transformers.__version__,
                                                 seed=seed, generator="oracle/gen_golden.py")).encode(), dtype=np.uint8)
    for k, v in model.state_dict().items():
        rec["sd/" + k] = v.detach().numpy()
    named = dict(model.named_parameters())
    sd_keys = model.state_dict().keys()
    # grads keyed by state-dict key (tied tensors share a grad; report it under every alias)
    ptr2grad = {p.data_ptr(): p.grad for p in named.values() if p.grad is not None}
    for k in sd_keys:
        t = model.state_dict(keep_vars=True)[k]
        if isinstance(t, torch.nn.Parameter) and t.data_ptr() in ptr2grad:
            rec["grad/" + k] = ptr2grad[t.data_ptr()].numpy()
    rec["input_ids"], rec["labels"] = ids.numpy(), labels.numpy()
    rec["logits"], rec["loss"] = logits.detach().numpy(), loss.detach().numpy()
    if trace:
        for i, h, r in traces[:cfg.n_layer]:
            rec[f"trace/{i}/hidden"], rec[f"trace/{i}/residual"] = h.numpy(), r.numpy()
        with torch.no_grad():
            rec["hidden"] = model.caduceus(ids, return_dict=True).last_hidden_state.numpy()
    else:
        rec["hidden"] = hidden.detach().numpy()
    np.savez_compressed(os.path.join(OUT, f"model_{name}.npz"), **rec)
    print(f"model_{name}: loss={float(loss):.6f} |logits|={float(logits.abs().mean()):.4f} keys={len(rec)}")
    return model, ids


def _intkeys(d):
    return {(int(k) if isinstance(k, str) and k.lstrip("-").isdigit() else k): v for k, v in d.items()}


def gen_scan_ops():
    for (b, E, L, N), seed in (((1, 64, 64, 16), 1), ((2, 32, 200, 16), 2), ((1, 16, 37, 8), 3)):
        g = torch.Generator().manual_seed(seed)
        r = lambda *s: torch.randn(*s, generator=g)
        u, delta, z = r(b, E, L).requires_grad_(), r(b, E, L).requires_grad_(), r(b, E, L).requires_grad_()
        A = (-(0.5 + 15.5 * torch.rand(E, N, generator=g))).requires_grad_()
        Bm, Cm = r(b, N, L).requires_grad_(), r(b, N, L).requires_grad_()
        D, bias = r(E).requires_grad_(), (r(E) - 3.0).requires_grad_()
        out = hf_mamba.mamba_selective_scan(u, delta, A, Bm, Cm, D=D, z=z, delta_bias=bias, delta_softplus=True)
        w = r(b, E, L)
        (out * w).sum().backward()
        rec = dict(u=u, delta=delta, A=A, B=Bm, C=Cm, D=D, z=z, delta_bias=bias, out=out, dout=w,
                   du=u.grad, ddelta=delta.grad, dA=A.grad, dB=Bm.grad, dC=Cm.grad, dD=D.grad, dz=z.grad,
                   ddelta_bias=bias.grad)
        np.savez_compressed(os.path.join(OUT, f"scan_op_{b}x{E}x{L}x{N}.npz"),
                            **{k: v.detach().numpy() for k, v in rec.items()})
        print(f"scan_op {b}x{E}x{L}x{N}: |out|={float(out.abs().mean()):.4f}")


def gen_conv_op():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 24, 75, generator=g).requires_grad_()
    w = (0.5 * torch.randn(24, 4, generator=g)).requires_grad_()
    b = (0.2 * torch.randn(24, generator=g)).requires_grad_()
    out = hf_mamba.causal_conv1d_fn(x, w, b, activation="silu")
    dout = torch.randn(2, 24, 75, generator=g)
    (out * dout).sum().backward()
    np.savez_compressed(os.path.join(OUT, "conv_op.npz"), x=x.detach().numpy(), w=w.detach().numpy(),
                        b=b.detach().numpy(), out=out.detach().numpy(), dout=dout.numpy(), dx=x.grad.numpy(),
                        dw=w.grad.numpy(), db=b.grad.numpy())
    print("conv_op ok")


def gen_equivariance(model, ids):
    """test_rcps.py:341-419 style property on the reference itself (CPU fp32): logits(x) vs logits(RC x)."""
    comp = torch.tensor([COMP.get(i, i) for i in range(16)])
    rc_in = comp[torch.flip(ids, dims=[-1])]
    with torch.no_grad():
        a = model(ids).logits
        b = model(rc_in).logits
    b_back = torch.flip(b[..., comp], dims=[1])
    print("reference equivariance max|diff| =", float((a - b_back).abs().max()))
    np.savez_compressed(os.path.join(OUT, "equivariance.npz"), input_ids=ids.numpy(), rc_input_ids=rc_in.numpy(),
                        logits=a.numpy(), logits_rc=b.numpy(), comp=comp.numpy())


def main():
    os.makedirs(OUT, exist_ok=True)
    m, ids = gen_model("ps_fused", make_cfg(), trace=True)
    gen_equivariance(m, ids)
    gen_model("ps_unfused", make_cfg(fused_add_norm=False), trace=True, seed=1)
    gen_model("ph_fused", make_cfg(rcps=False), trace=True, seed=2)
    gen_model("ph_unfused", make_cfg(rcps=False, fused_add_norm=False), seed=3)
    gen_model("ps_fused_ewmul", make_cfg(bidirectional_strategy="ew_multiply"), seed=4)
    gen_model("ps_fused_untied", make_cfg(bidirectional_weight_tie=False), seed=5)
    gen_model("ps_fused_unidir", make_cfg(bidirectional=False), seed=6)
    gen_model("ps_fused_layernorm", make_cfg(rms_norm=False), seed=7)
    gen_model("ps_fused_res32_odd", make_cfg(residual_in_fp32=True, d_model=48, n_layer=3), B=3, L=37, seed=8)
    gen_scan_ops()
    gen_conv_op()


if __name__ == "__main__":
    main()
