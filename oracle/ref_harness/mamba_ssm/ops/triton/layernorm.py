"""Torch restatement of upstream `mamba_ssm.ops.triton.layernorm` (v1.2.0) semantics  [upstream, from knowledge]:

    res   = x + residual            (fp32 arithmetic)
    y     = norm(res) * w (+ b)     (fp32 statistics, output in x.dtype)
    residual_out dtype = residual.dtype if residual given, else fp32 if residual_in_fp32, else x.dtype
    returns y if not prenorm else (y, residual_out)

Call sites in the reference: caduceus/modeling_rcps.py:175-195, caduceus/modeling_caduceus.py:241-273.
"""
import torch
from torch import nn


def _norm_fn(x, weight, bias, residual, prenorm, residual_in_fp32, eps, is_rms):
    x_dtype = x.dtype
    res = x.float()
    if residual is not None:
        res = res + residual.float()
        res_dtype = residual.dtype
    else:
        res_dtype = torch.float32 if residual_in_fp32 else x_dtype
    if is_rms:
        rstd = torch.rsqrt(res.pow(2).mean(-1, keepdim=True) + eps)
        y = res * rstd * weight.float()
    else:
        mean = res.mean(-1, keepdim=True)
        var = (res - mean).pow(2).mean(-1, keepdim=True)
        y = (res - mean) * torch.rsqrt(var + eps) * weight.float()
    if bias is not None:
        y = y + bias.float()
    y = y.to(x_dtype)
    return y if not prenorm else (y, res.to(res_dtype))


def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
    return _norm_fn(x, weight, bias, residual, prenorm, residual_in_fp32, eps, True)


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False,
                  is_rms_norm=False):
    return _norm_fn(x, weight, bias, residual, prenorm, residual_in_fp32, eps, is_rms_norm)


class RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
        self.register_parameter("bias", None)

    def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
        return rms_norm_fn(x, self.weight, self.bias, residual=residual, eps=self.eps, prenorm=prenorm,
                           residual_in_fp32=residual_in_fp32)
