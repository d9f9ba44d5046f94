"""Downstream consumers of the backbone used by the reference's fine-tuning harness:

* `DNAEmbeddingModelCaduceus`  -- /root/reference/src/models/sequence/dna_embedding.py:156-195: returns the backbone's hidden
  states as `(B, L, d_model, 2)` strand pairs when the model is RCPS or when conjoining (two passes: forward and RC
  input), plain `(B, L, d_model)` otherwise.  Return value is the reference's `(hidden, None)` tuple.
* `SequenceDecoder`            -- /root/reference/src/tasks/decoders.py:39-161: restrict the sequence to `l_output` positions
  (`last` / `first` / `pool` / `sum` / `ragged`), apply the output transform, and average the two strands when conjoining.

Pure host-side glue over the kernels (the backbone does the work); same constructor arguments, attribute names and error
behaviour as the reference classes, so fine-tuning checkpoints (`caduceus.*`, `output_transform.*`) load unchanged.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .configuration_caduceus import CaduceusConfig
from .modeling_caduceus import Caduceus


class DNAEmbeddingModelCaduceus(nn.Module):
    def __init__(self, config: CaduceusConfig, device=None, dtype=None, conjoin_train=False, conjoin_test=False):
        super().__init__()
        self.config = config
        self.d_model = config.d_model  # read by the decoder
        self.caduceus = Caduceus(config=config, device=device, dtype=dtype)
        self.conjoin_train = conjoin_train
        self.conjoin_test = conjoin_test

    def forward(self, input_ids, position_ids=None, inference_params=None, state=None):
        """`position_ids`, `inference_params`, `state` exist for the harness interface and are ignored, as in the reference."""
        if self.config.rcps:
            hidden = self.caduceus(input_ids, return_dict=False)
            half = hidden.shape[-1] // 2
            # strand 1 back in the forward frame: reverse positions and channels
            return torch.stack([hidden[..., :half], torch.flip(hidden[..., half:], dims=[1, 2])], dim=-1), None
        if self.conjoin_train or (self.conjoin_test and not self.training):
            assert input_ids.ndim == 3, "Input must be 3D tensor, where channels corresponds to forward and rc strands"
            fwd = self.caduceus(input_ids[..., 0], return_dict=False)
            rc = self.caduceus(input_ids[..., 1], return_dict=False)
            return torch.stack([fwd, rc], dim=-1), None
        return self.caduceus(input_ids, return_dict=False), None


class SequenceDecoder(nn.Module):
    def __init__(self, d_model, d_output=None, l_output=None, use_lengths=False, mode="last", conjoin_train=False,
                 conjoin_test=False):
        super().__init__()
        self.output_transform = nn.Identity() if d_output is None else nn.Linear(d_model, d_output)
        if l_output is None:
            self.l_output, self.squeeze = None, False
        elif l_output == 0:  # one position, then squeezed away
            self.l_output, self.squeeze = 1, True
        else:
            assert l_output > 0
            self.l_output, self.squeeze = l_output, False
        self.use_lengths = use_lengths
        self.mode = mode
        if mode == "ragged":
            assert not use_lengths
        self.conjoin_train = conjoin_train
        self.conjoin_test = conjoin_test

    def _restrict(self, x_seq, l_output, lengths):
        mode = self.mode
        if mode == "last":
            return x_seq[..., -l_output:, :]
        if mode == "first":
            return x_seq[..., :l_output, :]
        if mode == "pool":  # running mean: output j = mean of positions [0, L - l_output + j]
            ax = 1 if x_seq.dim() >= 3 else 0  # (B, L, D[, 2]) batches, or one (L, D) sample under use_lengths
            L = x_seq.size(ax)
            prefix = torch.cumsum(x_seq, dim=ax).narrow(ax, L - l_output, l_output)
            count = torch.arange(L - l_output + 1, L + 1, dtype=x_seq.dtype, device=x_seq.device)
            return prefix / count.view(*([1] * ax), -1, *([1] * (x_seq.dim() - ax - 1)))
        if mode == "sum":
            return torch.cumsum(x_seq, dim=-2)[..., -l_output:, :]
        if mode == "ragged":
            assert lengths is not None, "lengths must be provided for ragged mode"
            return x_seq[..., : max(lengths), :]
        raise NotImplementedError("Mode must be ['last' | 'first' | 'pool' | 'sum' | 'ragged']")

    def forward(self, x, state=None, lengths=None, l_output=None):
        """x: (B, L, d_model) or (B, L, d_model, 2) when conjoining.  Returns (B, l_output, d_output)."""
        if self.l_output is None:
            if l_output is not None:
                assert isinstance(l_output, int)
            else:
                l_output = x.size(1)
            squeeze = False
        else:
            l_output, squeeze = self.l_output, self.squeeze
        if self.mode not in ("last", "first", "pool", "sum", "ragged"):
            raise NotImplementedError("Mode must be ['last' | 'first' | 'pool' | 'sum' | 'ragged']")
        if self.use_lengths:
            assert lengths is not None
            x = torch.stack([self._restrict(out[..., :length, :], l_output, lengths)
                             for out, length in zip(torch.unbind(x, dim=0), lengths)], dim=0)
        else:
            x = self._restrict(x, l_output, lengths)
        if squeeze:
            assert x.size(1) == 1
            x = x.squeeze(1)
        if self.conjoin_train or (self.conjoin_test and not self.training):
            x, x_rc = x.chunk(2, dim=-1)
            x = (self.output_transform(x.squeeze()) + self.output_transform(x_rc.squeeze())) / 2
        else:
            x = self.output_transform(x)
        return x

    def step(self, x, state=None):
        x_fwd = self.output_transform(x.mean(dim=1))
        x_rc = self.output_transform(x.flip(dims=[1, 2]).mean(dim=1)).flip(dims=[1])
        return (x_fwd + x_rc) / 2
