"""CPU oracle for the Caduceus forward/backward hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module.
`caduceus_amd/` never does; the product path fails loudly when the HIP library is missing.

What this is: a from-scratch, fp32, torch-CPU *functional* restatement of the reference's algorithm in the
reference's own literal formulation (explicit `flip`/`cat`, two full Mamba invocations per BiMamba, RCPS
wrappers). It is deliberately NOT the flip-free "t-frame" formulation the HIP engine uses, so that agreement
between the two is an independent check of every index map in the engine.

Pinning: `oracle/gen_golden.py` runs the reference's own classes (imported from /root/reference, with the
third-party `mamba_ssm` dependency provided by `oracle/ref_harness/`, an adapter around the installed HF
`transformers` MambaMixer torch path) and commits inputs/outputs under `tests/golden/`.
`tests/test_oracle_golden.py` checks this file against every one of those vectors.  The Mamba arithmetic itself
lives in the un-vendored dependency mamba-ssm==1.2.0.post1 / causal-conv1d==1.2.0.post2
(/root/reference/caduceus_env.yml:46-50); its published algorithm is restated in `mamba_forward` below.

All functions take a flat state dict with the REFERENCE's key names and a plain-dict
config with the fields of `CaduceusConfig` (/root/reference/caduceus/configuration_caduceus.py:41-55).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# --------------------------------------------------------------------------------------------------
# selective scan (upstream `selective_scan_ref` semantics; the reference reaches it through
# mamba_ssm.Mamba.forward, imported at caduceus/modeling_caduceus.py:11 and called at :128,:130)
# --------------------------------------------------------------------------------------------------

_SCAN_BACKEND: Optional[Callable] = None  # optional C-backed autograd function (oracle_ops.selective_scan_c)


def set_scan_backend(fn: Optional[Callable]) -> None:
    """Install a faster (C, OpenMP) implementation of `selective_scan` for long sequences (cpu_baseline)."""
    global _SCAN_BACKEND
    _SCAN_BACKEND = fn


def selective_scan(u: Tensor, delta: Tensor, A: Tensor, B: Tensor, C: Tensor, D: Tensor, z: Tensor,
                   delta_bias: Tensor) -> Tensor:
    """u, delta, z: (b, E, L); A: (E, N); B, C: (b, N, L); D, delta_bias: (E).  Returns (b, E, L).

        dt   = softplus(delta + delta_bias)
        h_l  = exp(dt_l * A) * h_{l-1} + dt_l * B_l * u_l            (fp32 state, h_{-1} = 0)
        y_l  = <C_l, h_l> + D * u_l ;   out_l = y_l * silu(z_l)
    """
    if _SCAN_BACKEND is not None:
        return _SCAN_BACKEND(u, delta, A, B, C, D, z, delta_bias)
    b, E, L = u.shape
    dt = F.softplus(delta + delta_bias[None, :, None])
    dA = torch.exp(dt[:, :, :, None] * A[None, :, None, :])  # (b,E,L,N)
    dBu = dt[:, :, :, None] * B.transpose(1, 2)[:, None, :, :] * u[:, :, :, None]  # (b,E,L,N)
    h = torch.zeros(b, E, A.shape[1], dtype=u.dtype)
    ys = []
    for l in range(L):
        h = dA[:, :, l] * h + dBu[:, :, l]
        ys.append((h * C[:, None, :, l]).sum(-1))
    y = torch.stack(ys, dim=-1) + u * D[None, :, None]
    return y * F.silu(z)


def causal_conv1d_silu(x: Tensor, w: Tensor, bias: Optional[Tensor]) -> Tensor:
    """Depthwise causal conv (zero left pad) + SiLU.  x: (b, E, L); w: (E, K).
    xc[l] = silu(bias + sum_k w[:, k] * x[l - (K-1) + k])          """
    K = w.shape[-1]
    L = x.shape[-1]
    out = F.conv1d(x, w.unsqueeze(1), bias, padding=K - 1, groups=x.shape[1])[..., :L]
    return F.silu(out)


# test hook: a list here receives (prefix, xc (b, E, L), dbc (b, L, R + 2N)) of every mamba_forward call, dbc with retain_grad() -- the
# operands of the x_proj weight gradient, for the tests that localise an error to an operand (tests/test_configs.py)
TRACE: Optional[list] = None


def mamba_forward(sd: Dict[str, Tensor], pfx: str, hidden: Tensor) -> Tensor:
    """One `Mamba.forward` (slow-path math of mamba-ssm 1.2.0 `mamba_inner_fn`).  hidden: (b, L, D)."""
    w_in = sd[pfx + "in_proj.weight"]
    E = w_in.shape[0] // 2
    A = -torch.exp(sd[pfx + "A_log"].float())
    N = A.shape[1]
    w_dt = sd[pfx + "dt_proj.weight"]
    R = w_dt.shape[1]
    xz = F.linear(hidden, w_in, sd.get(pfx + "in_proj.bias")).transpose(1, 2)  # (b, 2E, L)
    x, z = xz[:, :E], xz[:, E:]
    xc = causal_conv1d_silu(x, sd[pfx + "conv1d.weight"].squeeze(1), sd.get(pfx + "conv1d.bias"))
    dbc = F.linear(xc.transpose(1, 2), sd[pfx + "x_proj.weight"])  # (b, L, R+2N)
    if TRACE is not None:
        dbc.retain_grad()
        TRACE.append((pfx, xc.detach(), dbc))
    dt_lr, Bm, Cm = torch.split(dbc, [R, N, N], dim=-1)
    delta = (w_dt @ dt_lr.transpose(1, 2))  # (b, E, L)
    y = selective_scan(xc, delta, A, Bm.transpose(1, 2), Cm.transpose(1, 2), sd[pfx + "D"].float(), z,
                       sd[pfx + "dt_proj.bias"].float())
    return F.linear(y.transpose(1, 2), sd[pfx + "out_proj.weight"], sd.get(pfx + "out_proj.bias"))


def bimamba_forward(sd, pfx: str, hidden: Tensor, cfg: dict) -> Tensor:
    """BiMambaWrapper.forward  (/root/reference/caduceus/modeling_caduceus.py:122-140)."""
    out = mamba_forward(sd, pfx + "mamba_fwd.", hidden)
    if cfg.get("bidirectional", True):
        out_rev = mamba_forward(sd, pfx + "mamba_rev.", hidden.flip(dims=(1,))).flip(dims=(1,))
        strat = cfg.get("bidirectional_strategy") or "add"
        if strat == "add":
            out = out + out_rev
        elif strat == "ew_multiply":
            out = out * out_rev
        else:
            raise NotImplementedError(strat)
    return out


# --------------------------------------------------------------------------------------------------
# norms (upstream rms_norm_fn / layer_norm_fn semantics; call sites modeling_rcps.py:175-195)
# --------------------------------------------------------------------------------------------------

def add_norm(x: Tensor, w: Tensor, b: Optional[Tensor], residual: Optional[Tensor], eps: float, rms: bool):
    res = x if residual is None else x + residual
    if rms:
        y = res * torch.rsqrt(res.pow(2).mean(-1, keepdim=True) + eps) * w
    else:
        mu = res.mean(-1, keepdim=True)
        y = (res - mu) * torch.rsqrt((res - mu).pow(2).mean(-1, keepdim=True) + eps) * w
    if b is not None:
        y = y + b
    return y, res


def rc(x: Tensor) -> Tensor:
    """RCPSWrapper.rc (modeling_rcps.py:80-83): flip length and channel dims."""
    return torch.flip(x, dims=[-2, -1])


def _norm_params(sd, pfx: str, fused: bool, rcps: bool):
    key = pfx + ("norm.weight" if (fused or not rcps) else "norm.submodule.weight")
    bkey = key[:-6] + "bias"
    return sd[key], sd.get(bkey)


def rcps_block_forward(sd, pfx: str, hidden: Tensor, residual: Optional[Tensor], cfg: dict):
    """RCPSMambaBlock.forward (modeling_rcps.py:160-199) incl. the fused-path strand swap."""
    fused, rms, eps = cfg["fused_add_norm"], cfg["rms_norm"], cfg["norm_epsilon"]
    w, b = _norm_params(sd, pfx, fused, True)
    D = hidden.shape[-1] // 2
    if not fused:
        # RCPSAddNormWrapper.forward (modeling_rcps.py:107-130), prenorm=True
        if residual is None:
            residual = hidden
            x_fwd, _ = add_norm(hidden[..., :D], w, b, None, eps, rms)
            x_rc, _ = add_norm(rc(hidden[..., D:]), w, b, None, eps, rms)
            hidden = torch.cat([x_fwd, rc(x_rc)], dim=-1)
        else:
            x_fwd, r_fwd = add_norm(hidden[..., :D], w, b, residual[..., :D], eps, rms)
            x_rc, r_rc = add_norm(rc(hidden[..., D:]), w, b, rc(residual[..., D:]), eps, rms)
            residual = torch.cat([r_fwd, rc(r_rc)], dim=-1)
            hidden = torch.cat([x_fwd, rc(x_rc)], dim=-1)
    else:
        # modeling_rcps.py:175-197: "fwd" norm sees the SECOND half, "rc" norm the FIRST half.
        h_fwd, r_fwd = add_norm(hidden[..., D:], w, b, None if residual is None else residual[..., D:], eps, rms)
        h_rc, r_rc = add_norm(hidden[..., :D].flip(dims=[-2, -1]), w, b,
                              None if residual is None else residual[..., :D].flip(dims=[-2, -1]), eps, rms)
        hidden = torch.cat([h_fwd, h_rc.flip(dims=[-2, -1])], dim=-1)
        residual = torch.cat([r_fwd, r_rc.flip(dims=[-2, -1])], dim=-1)
    # RCPSWrapper.forward (modeling_rcps.py:85-99)
    mpfx = pfx + "mixer.submodule."
    fwd_out = bimamba_forward(sd, mpfx, hidden[..., :D], cfg)
    rc_out = bimamba_forward(sd, mpfx, rc(hidden[..., D:]), cfg)
    return torch.cat([fwd_out, rc(rc_out)], dim=-1), residual


def block_forward(sd, pfx: str, hidden: Tensor, residual: Optional[Tensor], cfg: dict):
    """mamba_ssm Block.forward (Caduceus-Ph; selected at modeling_caduceus.py:64)."""
    w, b = _norm_params(sd, pfx, cfg["fused_add_norm"], False)
    hidden, residual = add_norm(hidden, w, b, residual, cfg["norm_epsilon"], cfg["rms_norm"])
    return bimamba_forward(sd, pfx + "mixer.", hidden, cfg), residual


# --------------------------------------------------------------------------------------------------
# embedding, backbone, head, loss
# --------------------------------------------------------------------------------------------------

def rcps_embedding(emb_w: Tensor, comp: Tensor, ids: Tensor) -> Tensor:
    """RCPSEmbedding.forward/.rc (modeling_rcps.py:46-67)."""
    rc_ids = torch.gather(comp.unsqueeze(0).expand(ids.shape[0], -1), dim=1, index=torch.flip(ids, dims=[-1]))
    return torch.cat([emb_w[ids], torch.flip(emb_w[rc_ids], dims=[-2, -1])], dim=-1)


def backbone_forward(sd, ids: Optional[Tensor], cfg: dict, inputs_embeds: Optional[Tensor] = None,
                     pfx: str = "caduceus.backbone.", collect: bool = False):
    """CaduceusMixerModel.forward (modeling_caduceus.py:216-276).  Returns (hidden, per-layer list)."""
    rcps, fused, rms, eps = cfg["rcps"], cfg["fused_add_norm"], cfg["rms_norm"], cfg["norm_epsilon"]
    if inputs_embeds is not None:
        hidden = inputs_embeds
    elif rcps:
        hidden = rcps_embedding(sd[pfx + "embeddings.word_embeddings.embedding.weight"],
                                sd[pfx + "embeddings.word_embeddings.complement_map"], ids)
    else:
        hidden = sd[pfx + "embeddings.word_embeddings.weight"][ids]
    residual = None
    trace = []
    for i in range(cfg["n_layer"]):
        lp = f"{pfx}layers.{i}."
        if rcps:
            hidden, residual = rcps_block_forward(sd, lp, hidden, residual, cfg)
        else:
            hidden, residual = block_forward(sd, lp, hidden, residual, cfg)
        if collect:
            trace.append((hidden, residual))
    if rcps and not fused:
        w, b = sd[pfx + "norm_f.submodule.weight"], sd.get(pfx + "norm_f.submodule.bias")
    else:
        w, b = sd[pfx + "norm_f.weight"], sd.get(pfx + "norm_f.bias")
    if rcps:
        # both the fused (modeling_caduceus.py:244-262) and un-fused (:234-236) final norms are un-swapped
        D = hidden.shape[-1] // 2
        h_fwd, _ = add_norm(hidden[..., :D], w, b, residual[..., :D], eps, rms)
        h_rc, _ = add_norm(rc(hidden[..., D:]), w, b, rc(residual[..., D:]), eps, rms)
        hidden = torch.cat([h_fwd, rc(h_rc)], dim=-1)
    else:
        hidden, _ = add_norm(hidden, w, b, residual, eps, rms)
    return hidden, trace


def lm_head(sd, hidden: Tensor, cfg: dict) -> Tensor:
    """RCPSLMHead.forward (modeling_rcps.py:233-246) / nn.Linear head (modeling_caduceus.py:407-412)."""
    if cfg["rcps"]:
        w = sd["lm_head.lm_head.weight"]
        comp = sd["lm_head.complement_map"]
        D = hidden.shape[-1] // 2
        return F.linear(hidden[..., :D], w) + F.linear(torch.flip(hidden[..., D:], dims=[-1]), w[comp, :])
    return F.linear(hidden, sd["lm_head.weight"])


def cross_entropy(logits: Tensor, y: Tensor, ignore_index: int = -100) -> Tensor:
    """modeling_caduceus.py:279-283 / src/tasks/metrics.py:181-184."""
    return F.cross_entropy(logits.view(-1, logits.shape[-1]), y.view(-1), ignore_index=ignore_index)


def weighted_cross_entropy(logits: Tensor, y: Tensor, loss_weights: Tensor, ignore_index: int = -100) -> Tensor:
    """modeling_caduceus.py:286-294."""
    logits = logits.view(-1, logits.shape[-1])
    y = y.view(-1)
    ce = F.cross_entropy(logits, y, ignore_index=ignore_index, reduction="none")
    lw = loss_weights.reshape(-1).clone()
    lw[y == ignore_index] = 0.0
    return (ce * (lw / lw.sum())).sum()


def masked_lm_forward(sd, ids: Tensor, cfg: dict, labels: Optional[Tensor] = None, ignore_index: int = -100,
                      collect: bool = False):
    """CaduceusForMaskedLM.forward (modeling_caduceus.py:449-492).  Returns dict(logits, loss, hidden, trace)."""
    hidden, trace = backbone_forward(sd, ids, cfg, collect=collect)
    logits = lm_head(sd, hidden, cfg).float()
    loss = cross_entropy(logits, labels, ignore_index) if labels is not None else None
    return {"logits": logits, "loss": loss, "hidden": hidden, "trace": trace}


# --------------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------------

def padded_vocab(cfg: dict) -> int:
    """Caduceus.__init__ vocab padding (modeling_caduceus.py:352-357)."""
    v, m = cfg["vocab_size"], cfg.get("pad_vocab_size_multiple", 8)
    return v + (m - v % m) % m


def dt_rank_of(d_model: int, ssm_cfg: Optional[dict]) -> int:
    r = (ssm_cfg or {}).get("dt_rank", "auto")
    return math.ceil(d_model / 16) if r == "auto" else int(r)


def rc_ids(ids: Tensor, comp: Tensor) -> Tensor:
    """Reverse-complement token ids (test_rcps.py fixture semantics)."""
    return comp[torch.flip(ids, dims=[-1])]
