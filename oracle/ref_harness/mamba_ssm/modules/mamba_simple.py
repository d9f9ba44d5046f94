"""`Mamba` and `Block` with upstream mamba-ssm 1.2.0 signatures, backed by the installed third-party
`transformers.models.mamba.modeling_mamba.MambaMixer` torch (slow) path — an independent implementation of
the same Mamba-1 arithmetic with identical parameter names/shapes.
Reference import site: caduceus/modeling_caduceus.py:11-15.
"""
import math

import torch
from torch import nn
from transformers.models.mamba.configuration_mamba import MambaConfig
from transformers.models.mamba.modeling_mamba import MambaMixer

from mamba_ssm.ops.triton.layernorm import RMSNorm, layer_norm_fn, rms_norm_fn


class Mamba(MambaMixer):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=True, layer_idx=None, device=None, dtype=None):
        dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        cfg = MambaConfig(hidden_size=d_model, state_size=d_state, conv_kernel=d_conv, expand=expand,
                          time_step_rank=dt_rank, time_step_min=dt_min, time_step_max=dt_max,
                          time_step_init_scheme=dt_init, time_step_scale=dt_scale,
                          time_step_floor=dt_init_floor, use_conv_bias=conv_bias, use_bias=bias,
                          num_hidden_layers=1, vocab_size=8)
        super().__init__(cfg, layer_idx=0)
        self.layer_idx = layer_idx
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner, self.dt_rank = int(expand * d_model), dt_rank
        self.dt_proj.bias._no_reinit = True
        self.A_log._no_weight_decay = True
        self.D._no_weight_decay = True
        if device is not None or dtype is not None:
            self.to(device=device, dtype=dtype)

    def forward(self, hidden_states, inference_params=None):
        assert inference_params is None
        was_training = self.training
        self.training = False  # force the torch path (the `training` branch only probes for hub kernels)
        try:
            return MambaMixer.forward(self, hidden_states)
        finally:
            self.training = was_training


class Block(nn.Module):
    """Upstream mamba_simple.Block semantics (pre-norm residual block) [upstream, from knowledge]."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False):
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)

    def forward(self, hidden_states, residual=None, inference_params=None):
        if not self.fused_add_norm:
            residual = (hidden_states + residual) if residual is not None else hidden_states
            hidden_states = self.norm(residual.to(dtype=self.norm.weight.dtype))
            if self.residual_in_fp32:
                residual = residual.to(torch.float32)
        else:
            fn = rms_norm_fn if isinstance(self.norm, RMSNorm) else layer_norm_fn
            hidden_states, residual = fn(hidden_states, self.norm.weight, self.norm.bias, residual=residual,
                                         prenorm=True, residual_in_fp32=self.residual_in_fp32, eps=self.norm.eps)
        hidden_states = self.mixer(hidden_states, inference_params=inference_params)
        return hidden_states, residual
