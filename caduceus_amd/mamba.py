"""The mixer-level boundary the reference sits on, re-implemented over the HIP engine:
`Mamba`, `Block`, `RMSNorm`, `rms_norm_fn`, `layer_norm_fn` with the constructor / forward signatures of
mamba-ssm 1.2.0 (imported by the reference at /root/reference/caduceus/modeling_caduceus.py:11-27 and
modeling_rcps.py:12-18; SURVEY.md section 8b "lower (mixer) boundary").  Parameter names, shapes and
initialisation follow upstream so that state dicts are interchangeable.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from . import engine, ops


def requested_dtype(x: torch.Tensor) -> torch.dtype:
    """The dtype the caller asked for: the autocast dtype when autocast is on (the reference trains under AMP,
    configs/experiment/hg38/hg38.yaml:20), else the dtype of x."""
    dt = x.dtype
    dev = x.device.type
    try:
        if torch.is_autocast_enabled(dev):
            dt = torch.get_autocast_dtype(dev)
    except (TypeError, RuntimeError):
        pass
    return dt


def act_dtype_of(x: torch.Tensor) -> torch.dtype:
    """Compute dtype of the kernels for input x.  fp32 and bf16 are implemented; a float16 request (the reference's own AMP
    precision, and vep_embeddings.py:352) is COMPUTED BY THE FP32 KERNELS -- at least as accurate as fp16 arithmetic, no fp16
    instantiations of the kernels -- and the module outputs are rounded to float16 (as_requested)."""
    dt = requested_dtype(x)
    if dt == torch.float16:
        return torch.float32
    if dt not in (torch.float32, torch.bfloat16):
        raise NotImplementedError(f"caduceus_amd computes in float32 or bfloat16 (requested {dt}); on MI355X use bf16")
    return dt


def as_requested(t: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Output t of a module called with input x: float16 when float16 was requested (see act_dtype_of), else unchanged."""
    return t.to(torch.float16) if requested_dtype(x) == torch.float16 else t


class Mamba(nn.Module):
    """Parameter container + single-direction forward with the signature of mamba_ssm.modules.mamba_simple.Mamba."""

    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True,
                 layer_idx=None, device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path = use_fast_path
        self.layer_idx = layer_idx
        if d_conv > 4:
            raise NotImplementedError("d_conv <= 4 (same limit as upstream causal_conv1d)")
        self.in_proj = nn.Linear(d_model, self.d_inner * 2, bias=bias, **factory_kwargs)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, bias=conv_bias, kernel_size=d_conv, groups=self.d_inner,
                                padding=d_conv - 1, **factory_kwargs)
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + d_state * 2, bias=False, **factory_kwargs)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)
        # dt_proj init preserves variance at initialisation; bias = softplus^-1(dt), dt ~ logU[dt_min, dt_max]
        dt_init_std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, dt_init_std)
        elif dt_init == "random":
            nn.init.uniform_(self.dt_proj.weight, -dt_init_std, dt_init_std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(self.d_inner, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))
        with torch.no_grad():
            self.dt_proj.bias.copy_(inv_dt)
        self.dt_proj.bias._no_reinit = True
        A = torch.arange(1, d_state + 1, dtype=torch.float32, device=device).repeat(self.d_inner, 1).contiguous()
        self.A_log = nn.Parameter(torch.log(A))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias, **factory_kwargs)

    def forward(self, hidden_states, inference_params=None):
        """hidden_states: (B, L, D) -> (B, L, D), left-to-right."""
        if inference_params is not None:
            raise NotImplementedError("step-wise inference cache is outside the pre-training hot path")
        act = act_dtype_of(hidden_states)
        out = engine.bimamba_tframe(hidden_states.to(act).unsqueeze(0), self, None, None, strand_swap=False)
        return as_requested(out[0], hidden_states)

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        raise NotImplementedError("step-wise inference cache is outside the pre-training hot path")


class RMSNorm(nn.Module):
    """mamba_ssm.ops.triton.layernorm.RMSNorm: weight only, `bias` registered as None."""

    def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.register_parameter("bias", None)

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        return rms_norm_fn(x, self.weight, self.bias, residual=residual, eps=self.eps, prenorm=prenorm,
                           residual_in_fp32=residual_in_fp32)


def _norm_fn(x, weight, bias, residual, prenorm, eps, is_rms):
    act = act_dtype_of(x)
    shape = x.shape
    xs = x.reshape(1, -1, shape[-1])
    rs = None if residual is None else residual.reshape(1, -1, shape[-1]).float()
    if xs.dtype not in (torch.float32, act):
        xs = xs.to(act)
    y, res = ops.add_norm(xs, rs, weight, bias, eps, is_rms, False, act)
    y = as_requested(y.reshape(shape), x)
    return y if not prenorm else (y, res.reshape(shape))


def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
    """Fused add + RMSNorm (single strand).  The residual stream is always kept in fp32 by this engine."""
    return _norm_fn(x, weight, bias, residual, prenorm, eps, True)


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False, is_rms_norm=False):
    return _norm_fn(x, weight, bias, residual, prenorm, eps, is_rms_norm)


def norm_params(norm: nn.Module):
    """(weight, bias, eps, is_rms) of an RMSNorm / nn.LayerNorm module."""
    if isinstance(norm, RMSNorm):
        return norm.weight, norm.bias, norm.eps, True
    if isinstance(norm, nn.LayerNorm):
        return norm.weight, norm.bias, norm.eps, False
    raise TypeError("Only LayerNorm and RMSNorm are supported")


class Block(nn.Module):
    """mamba_ssm.modules.mamba_simple.Block (Caduceus-Ph layers): Add -> Norm -> Mixer, returning (hidden, residual)."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False):
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)

    def forward_tframe(self, hidden: torch.Tensor, residual: Optional[torch.Tensor], act: torch.dtype):
        w, b, eps, is_rms = norm_params(self.norm)
        hn, residual = ops.add_norm(hidden, residual, w, b, eps, is_rms, False, act, want_fp8=True)  # (feeds the mixer's in_proj)
        return self.mixer.forward_tframe(hn, strand_swap=False), residual

    def forward(self, hidden_states, residual=None, inference_params=None):
        act = act_dtype_of(hidden_states)
        h = hidden_states.unsqueeze(0)
        if h.dtype not in (torch.float32, act):
            h = h.to(act)
        r = None if residual is None else residual.unsqueeze(0).float()
        out, res = self.forward_tframe(h, r, act)
        return as_requested(out[0], hidden_states), res[0]

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return self.mixer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)
