"""Reverse-complement equivariant modules with the reference's names, constructor signatures, state-dict keys and
(B, L, 2*D) call convention (/root/reference/caduceus/modeling_rcps.py), executed flip-free on the HIP engine.

Every module offers `forward(...)` in the reference frame (what the reference's tests in
caduceus/tests/test_rcps.py call) and the model uses the `*_tframe` methods internally (no flips / cats at all).
"""
from collections import OrderedDict
from typing import Optional

import torch
from torch import Tensor, nn

from . import engine, ops
from .mamba import RMSNorm, act_dtype_of, as_requested, norm_params


def _comp_tensor(complement_map: dict) -> Tensor:
    """complement_map[i] by token id i.  Built from the SORTED integer keys: a config that went through HF's JSON
    round trip (to_json_string sorts the string keys lexicographically: '0', '1', '10', '11', '2', ...) must not
    scramble the map (the reference relies on dict order, modeling_rcps.py:32-35, and on the checkpoint overwriting it)."""
    keys = sorted(int(k) for k in complement_map)
    assert keys == list(range(len(keys))), "complement_map must cover token ids 0 .. V-1"
    by_int = {int(k): int(v) for k, v in complement_map.items()}
    return torch.tensor([by_int[k] for k in keys], dtype=torch.long)


class RCPSEmbedding(nn.Module):
    """modeling_rcps.py:21-67."""

    def __init__(self, vocab_size: int, d_model: int, complement_map: dict, **factory_kwargs):
        super().__init__()
        self.register_buffer("complement_map", _comp_tensor(complement_map))
        self.embedding = nn.Embedding(vocab_size, d_model, **factory_kwargs)

    @property
    def weight(self):
        return self.embedding.weight

    def set_weight(self, value):
        self.embedding.weight = value

    def rc(self, x):
        """Reverse-complement a tensor of input ids (integer index path, exact)."""
        return self.complement_map[torch.flip(x, dims=[-1])]

    def forward_tframe(self, input_ids, out_dtype=torch.float32):
        return ops.embed(input_ids, self.embedding.weight, self.complement_map, 2, out_dtype)

    def forward(self, input_ids):
        """(B, L) -> (B, L, 2 * d_model)."""
        w = self.embedding.weight
        return engine.from_tframe(self.forward_tframe(input_ids, w.dtype))


class RCPSWrapper(nn.Module):
    """modeling_rcps.py:70-99.  With a BiMambaWrapper inside, the wrapper is two index maps; for any other submodule
    the reference's literal composition is used (it is then not on the hot path)."""

    def __init__(self, submodule: nn.Module):
        super().__init__()
        self.submodule = submodule

    @staticmethod
    def rc(x):
        return torch.flip(x, dims=[-2, -1])

    def forward_tframe(self, hn: Tensor, **kwargs):
        return self.submodule.forward_tframe(hn, strand_swap=True)

    def forward(self, x, **kwargs):
        if hasattr(self.submodule, "forward_tframe"):
            kwargs.pop("inference_params", None)
            act = act_dtype_of(x)
            return as_requested(engine.from_tframe(self.forward_tframe(engine.to_tframe(x.to(act), True))), x)
        n_channels = x.shape[-1]
        fwd_out = self.submodule(x[..., :n_channels // 2], **kwargs)
        rc_out = self.submodule(self.rc(x[..., n_channels // 2:]), **kwargs)
        return torch.cat([fwd_out, self.rc(rc_out)], dim=-1)

    def allocate_inference_cache(self, *args, **kwargs):
        return self.submodule.allocate_inference_cache(*args, **kwargs)


class RCPSAddNormWrapper(RCPSWrapper):
    """modeling_rcps.py:102-130 (the un-fused add+norm: no strand swap)."""

    def __init__(self, submodule: nn.Module):
        super().__init__(submodule)

    def forward_tframe(self, hidden: Tensor, residual: Optional[Tensor], act: torch.dtype):
        w, b, eps, is_rms = norm_params(self.submodule)
        return ops.add_norm(hidden, residual, w, b, eps, is_rms, False, act)

    def forward(self, x, residual=None, prenorm=False):
        act = act_dtype_of(x)
        t = engine.to_tframe(x, True)
        if t.dtype not in (torch.float32, act):
            t = t.to(act)
        r = None if residual is None else engine.to_tframe(residual.float(), True)
        y, res = self.forward_tframe(t, r, act)
        y = as_requested(engine.from_tframe(y), x)
        return y if not prenorm else (y, engine.from_tframe(res))


class RCPSMambaBlock(nn.Module):
    """modeling_rcps.py:133-206, including the fused-path quirk that the two strands swap every layer
    (SURVEY.md section 0): reproduced by `swap_flip` in the add+norm kernel."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False,
                 device=None, dtype=None):
        super().__init__()
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.mixer = RCPSWrapper(mixer_cls(dim))
        norm_f = norm_cls(dim)
        self.norm = norm_f if fused_add_norm else RCPSAddNormWrapper(norm_f)
        if self.fused_add_norm:
            assert isinstance(self.norm, (nn.LayerNorm, RMSNorm)), \
                "Only LayerNorm and RMSNorm are supported for fused_add_norm"

    def forward_tframe(self, hidden: Tensor, residual: Optional[Tensor], act: torch.dtype):
        if self.fused_add_norm:
            w, b, eps, is_rms = norm_params(self.norm)
            hn, residual = ops.add_norm(hidden, residual, w, b, eps, is_rms, True, act, want_fp8=True)  # (feeds the mixer's in_proj)
        else:
            hn, residual = self.norm.forward_tframe(hidden, residual, act)
        return self.mixer.forward_tframe(hn), residual

    def forward(self, hidden_states: Tensor, residual: Optional[Tensor] = None, inference_params=None):
        """(B, L, 2D) x2 -> (hidden, residual) like the reference (residual is returned in fp32)."""
        act = act_dtype_of(hidden_states)
        h = engine.to_tframe(hidden_states, True)
        if h.dtype not in (torch.float32, act):
            h = h.to(act)
        r = None if residual is None else engine.to_tframe(residual.float(), True)
        out, res = self.forward_tframe(h, r, act)
        return as_requested(engine.from_tframe(out), hidden_states), engine.from_tframe(res)

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return self.mixer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)


class RCPSLMHead(nn.Module):
    """modeling_rcps.py:209-246."""

    def __init__(self, true_dim: int, vocab_size: int, complement_map: dict, **factory_kwargs):
        super().__init__()
        self.register_buffer("complement_map", _comp_tensor(complement_map))
        self.true_dim = true_dim
        self.lm_head = nn.Linear(true_dim, vocab_size, bias=False, **factory_kwargs)

    @property
    def weight(self):
        return self.lm_head.weight

    def set_weight(self, value):
        self.lm_head.weight = value

    def forward_tframe(self, hidden: Tensor, labels=None, ignore_index=-100):
        """hidden (2, B, L, D) -> (fp32 logits (B, L, V), loss | None)."""
        if self.lm_head.bias is not None:
            raise NotImplementedError("RCPSLMHead is bias-free in the reference")
        V = self.lm_head.weight.shape[0]
        if V > 16:  # large vocabularies: a real GEMM (fp32: cad_gemm_f32; bf16: hipBLASLt)
            w = self.lm_head.weight.to(hidden.dtype)
            D = hidden.shape[-1]
            logits = ops.addmm(ops.mm(hidden[0].reshape(-1, D), w.t()), hidden[1].reshape(-1, D), w[self.complement_map].t())
            logits = logits.view(*hidden.shape[1:-1], V).float()
            loss = None
            if labels is not None:
                loss = torch.nn.functional.cross_entropy(logits.view(-1, V), labels.view(-1), ignore_index=ignore_index)
            return logits, loss
        return ops.lm_head(hidden, self.lm_head.weight, self.complement_map, labels, ignore_index)

    def forward(self, x):
        n_channels = x.shape[-1]
        assert n_channels == 2 * self.true_dim, "Input must have 2 * true_dim channels."
        act = act_dtype_of(x)
        logits, _ = self.forward_tframe(engine.to_tframe(x.to(act), True))
        return logits.to(x.dtype)
