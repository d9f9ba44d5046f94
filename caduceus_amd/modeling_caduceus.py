"""Caduceus model for Hugging Face, MI355X-native engine behind the reference's surface.

Same public classes, constructor / forward signatures, module tree and state-dict keys as
/root/reference/caduceus/modeling_caduceus.py, so `train.py`-style loops (`model(input_ids).logits`), the HF `AutoModel`
path and reference checkpoints work unchanged; all arithmetic runs in the flip-free t-frame on the HIP kernels
(caduceus_amd.engine / caduceus_amd.ops).  No dependency on mamba_ssm / causal_conv1d / triton.
"""
import math
from functools import partial
from typing import Optional, Tuple, Union

import torch
from torch import nn
from torch.nn import functional as F
from transformers import PreTrainedModel
from transformers.modeling_outputs import BaseModelOutputWithNoAttention, MaskedLMOutput, SequenceClassifierOutput

from . import engine, ops
from . import mixer as mixer_sched
from .configuration_caduceus import CaduceusConfig
from .mamba import Block, Mamba, RMSNorm, act_dtype_of, as_requested, norm_params
from .modeling_rcps import RCPSAddNormWrapper, RCPSEmbedding, RCPSLMHead, RCPSMambaBlock, RCPSWrapper


def create_block(d_model, ssm_cfg=None, norm_epsilon=1e-5, rms_norm=False, residual_in_fp32=False,
                 fused_add_norm=False, layer_idx=None, bidirectional=True, bidirectional_strategy="add",
                 bidirectional_weight_tie=True, rcps=False, device=None, dtype=None):
    """modeling_caduceus.py:33-84."""
    if ssm_cfg is None:
        ssm_cfg = {}
    factory_kwargs = {"device": device, "dtype": dtype}
    bidirectional_kwargs = {
        "bidirectional": bidirectional,
        "bidirectional_strategy": bidirectional_strategy,
        "bidirectional_weight_tie": bidirectional_weight_tie,
    }
    mixer_cls = partial(BiMambaWrapper, layer_idx=layer_idx, **ssm_cfg, **bidirectional_kwargs, **factory_kwargs)
    norm_cls = partial(nn.LayerNorm if not rms_norm else RMSNorm, eps=norm_epsilon, **factory_kwargs)
    block_cls = RCPSMambaBlock if rcps else Block
    block = block_cls(d_model, mixer_cls, norm_cls=norm_cls, fused_add_norm=fused_add_norm,
                      residual_in_fp32=residual_in_fp32)
    block.layer_idx = layer_idx
    return block


class BiMambaWrapper(nn.Module):
    """modeling_caduceus.py:87-140: weight-tied forward + reverse Mamba; here one shared in_proj, two index-mapped
    scans, one accumulated out_proj."""

    def __init__(self, d_model: int, bidirectional: bool = True, bidirectional_strategy: Optional[str] = "add",
                 bidirectional_weight_tie: bool = True, **mamba_kwargs):
        super().__init__()
        if bidirectional and bidirectional_strategy is None:
            bidirectional_strategy = "add"
        if bidirectional and bidirectional_strategy not in ["add", "ew_multiply"]:
            raise NotImplementedError(f"`{bidirectional_strategy}` strategy for bi-directionality is not implemented!")
        self.bidirectional = bidirectional
        self.bidirectional_strategy = bidirectional_strategy
        self.bidirectional_weight_tie = bool(bidirectional and bidirectional_weight_tie)
        self.mamba_fwd = Mamba(d_model=d_model, **mamba_kwargs)
        if bidirectional:
            self.mamba_rev = Mamba(d_model=d_model, **mamba_kwargs)
            self.retie()
        else:
            self.mamba_rev = None

    def retie(self):
        """Tie in and out projections (where most of param count lies), modeling_caduceus.py:114-118."""
        if self.bidirectional_weight_tie:
            self.mamba_rev.in_proj.weight = self.mamba_fwd.in_proj.weight
            self.mamba_rev.in_proj.bias = self.mamba_fwd.in_proj.bias
            self.mamba_rev.out_proj.weight = self.mamba_fwd.out_proj.weight
            self.mamba_rev.out_proj.bias = self.mamba_fwd.out_proj.bias

    def forward_tframe(self, hn: torch.Tensor, strand_swap: bool) -> torch.Tensor:
        return engine.bimamba_tframe(hn, self.mamba_fwd, self.mamba_rev if self.bidirectional else None,
                                     self.bidirectional_strategy, strand_swap)

    def forward(self, hidden_states, inference_params=None):
        """hidden_states: (B, L, D) -> same shape."""
        if inference_params is not None:
            raise NotImplementedError("step-wise inference cache is outside the pre-training hot path")
        act = act_dtype_of(hidden_states)
        return as_requested(self.forward_tframe(hidden_states.to(act).unsqueeze(0), strand_swap=False)[0], hidden_states)

    def allocate_inference_cache(self, *args, **kwargs):
        raise NotImplementedError("step-wise inference cache is outside the pre-training hot path")


class CaduceusEmbeddings(nn.Module):
    """modeling_caduceus.py:143-163."""

    def __init__(self, config: CaduceusConfig, device=None, dtype=None):
        super().__init__()
        factory_kwargs = {"device": device, "dtype": dtype}
        if config.rcps:
            self.word_embeddings = RCPSEmbedding(config.vocab_size, config.d_model, config.complement_map,
                                                 **factory_kwargs)
        else:
            self.word_embeddings = nn.Embedding(config.vocab_size, config.d_model, **factory_kwargs)

    def forward_tframe(self, input_ids, out_dtype=torch.float32):
        we = self.word_embeddings
        if isinstance(we, RCPSEmbedding):
            return we.forward_tframe(input_ids, out_dtype)
        return ops.embed(input_ids, we.weight, None, 1, out_dtype)

    def forward(self, input_ids):
        return engine.from_tframe(self.forward_tframe(input_ids, self.word_embeddings.weight.dtype))


class CaduceusMixerModel(nn.Module):
    """modeling_caduceus.py:166-276."""

    def __init__(self, config: CaduceusConfig, device=None, dtype=None) -> None:
        super().__init__()
        factory_kwargs = {"device": device, "dtype": dtype}
        self.fused_add_norm = config.fused_add_norm
        self.rcps = config.rcps
        self.residual_in_fp32 = config.residual_in_fp32
        self.embeddings = CaduceusEmbeddings(config, **factory_kwargs)
        self.layers = nn.ModuleList([
            create_block(config.d_model, ssm_cfg=config.ssm_cfg, norm_epsilon=config.norm_epsilon,
                         rms_norm=config.rms_norm, residual_in_fp32=config.residual_in_fp32,
                         fused_add_norm=config.fused_add_norm, layer_idx=i, bidirectional=config.bidirectional,
                         bidirectional_strategy=config.bidirectional_strategy,
                         bidirectional_weight_tie=config.bidirectional_weight_tie, rcps=config.rcps, **factory_kwargs)
            for i in range(config.n_layer)
        ])
        norm_f = (nn.LayerNorm if not config.rms_norm else RMSNorm)(config.d_model, eps=config.norm_epsilon,
                                                                    **factory_kwargs)
        self.norm_f = norm_f if (config.fused_add_norm or not config.rcps) else RCPSAddNormWrapper(norm_f)

    def forward_tframe(self, input_ids, inputs_embeds=None, collect: Optional[list] = None) -> torch.Tensor:
        """Returns the final normed hidden state in the t-frame (S, B, L, D) in the compute dtype."""
        if inputs_embeds is not None:
            act = act_dtype_of(inputs_embeds)
            hidden = engine.to_tframe(inputs_embeds, self.rcps)
            if hidden.dtype not in (torch.float32, act):
                hidden = hidden.to(act)
        else:
            w = self.embeddings.word_embeddings.weight
            act = act_dtype_of(w)
            hidden = self.embeddings.forward_tframe(input_ids, torch.float32 if w.dtype == torch.float32 else act)
        residual = None
        if act != torch.float32:  # one multi-tensor cast of every layer's projection weights (mixer.prepare_step_cache)
            pairs = []
            for layer in self.layers:
                m = layer.mixer.submodule if isinstance(layer.mixer, RCPSWrapper) else layer.mixer
                if isinstance(m, BiMambaWrapper) and m.bidirectional and \
                        mixer_sched.can_use(m.mamba_fwd, m.mamba_rev, m.bidirectional_strategy):
                    pairs.append((m.mamba_fwd, m.mamba_rev))
            mixer_sched.prepare_step_cache(pairs, act)
        for layer in self.layers:
            if collect is not None:
                collect.append(hidden)
            hidden, residual = layer.forward_tframe(hidden, residual, act)
        nf = self.norm_f.submodule if isinstance(self.norm_f, RCPSAddNormWrapper) else self.norm_f
        w, b, eps, is_rms = norm_params(nf)
        # final norm never swaps strands (modeling_caduceus.py:234-262)
        hidden, _ = ops.add_norm(hidden, residual, w, b, eps, is_rms, False, act)
        if collect is not None and self.fused_add_norm:  # reference quirk: appended only in the fused branch (:274-275)
            collect.append(hidden)
        return hidden

    def forward(self, input_ids, inputs_embeds=None, output_hidden_states=False):
        """Mixer forward: returns (hidden_states (B, L, 2D | D), all_hidden_states)."""
        collect = [] if output_hidden_states else None
        hidden = self.forward_tframe(input_ids, inputs_embeds, collect)
        like = inputs_embeds if inputs_embeds is not None else self.embeddings.word_embeddings.weight
        all_hidden_states = [as_requested(engine.from_tframe(h), like) for h in collect] if collect is not None else []
        return as_requested(engine.from_tframe(hidden), like), all_hidden_states


def _first(outputs):
    """Backbone output when `return_dict=False` may be a bare tensor."""
    return outputs if isinstance(outputs, torch.Tensor) else outputs[0]


def cross_entropy(logits, y, ignore_index=-100):
    """modeling_caduceus.py:279-283."""
    logits = logits.view(-1, logits.shape[-1])
    y = y.view(-1)
    return F.cross_entropy(logits, y, ignore_index=ignore_index)


def weighted_cross_entropy(logits, y, loss_weights, ignore_index=-100):
    """modeling_caduceus.py:286-294."""
    logits = logits.view(-1, logits.shape[-1])
    y = y.view(-1)
    ce = F.cross_entropy(logits, y, ignore_index=ignore_index, reduction="none")
    loss_weights = loss_weights.view(-1)
    loss_weights[y == ignore_index] = 0.0
    return (ce * (loss_weights / loss_weights.sum())).sum()


class CaduceusPreTrainedModel(PreTrainedModel):
    """modeling_caduceus.py:297-341."""
    config_class = CaduceusConfig
    base_model_prefix = "caduceus"
    supports_gradient_checkpointing = False
    _no_split_modules = ["BiMambaWrapper"]

    def _init_weights(self, module, initializer_range=0.02, **kwargs):
        n_layer = self.config.n_layer
        initialized_cfg = self.config.initializer_cfg if self.config.initializer_cfg is not None else {}
        rescale_prenorm_residual = initialized_cfg.get("rescale_prenorm_residual", True)
        initializer_range = initialized_cfg.get("initializer_range", initializer_range)
        n_residuals_per_layer = initialized_cfg.get("n_residuals_per_layer", 1)
        if isinstance(module, nn.Linear):
            if module.bias is not None:
                if not getattr(module.bias, "_no_reinit", False):
                    nn.init.zeros_(module.bias)
        elif isinstance(module, nn.Embedding):
            nn.init.normal_(module.weight, std=initializer_range)
        if rescale_prenorm_residual:
            for name, p in module.named_parameters():
                if name in ["out_proj.weight", "fc2.weight"]:
                    nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                    with torch.no_grad():
                        p /= math.sqrt(n_residuals_per_layer * n_layer)

    def _retie_shared(self, missing_keys=None):
        """Re-establish module-level parameter sharing (BiMamba in/out projections) and publish the full alias map in
        the form newer transformers releases use for (de)serialisation; tolerant of loaders that replaced parameters."""
        for m in self.modules():
            if isinstance(m, BiMambaWrapper) and m.bidirectional:
                m.retie()
        first, mapping = {}, {}
        for name, p in self.named_parameters(remove_duplicate=False):
            if id(p) in first:
                mapping[name] = first[id(p)]
            else:
                first[id(p)] = name
        self.all_tied_weights_keys = mapping
        self._tied_weights_keys = dict(mapping)  # what save_pretrained consults for known duplicates
        if missing_keys is not None:
            for k in mapping:
                missing_keys.discard(k)

    def tie_weights(self, missing_keys=None, **kwargs):
        self._retie_shared(missing_keys)

    @property
    def _return_dict_default(self):
        rd = getattr(self.config, "return_dict", None)
        return True if rd is None else rd


class Caduceus(CaduceusPreTrainedModel):
    """modeling_caduceus.py:344-389."""

    def __init__(self, config: CaduceusConfig, device=None, dtype=None, **kwargs):
        super().__init__(config)
        if config.rcps:
            assert config.complement_map is not None, "Complement map must be provided for RCPS."
        if config.vocab_size % config.pad_vocab_size_multiple != 0:
            config.vocab_size += config.pad_vocab_size_multiple - (config.vocab_size % config.pad_vocab_size_multiple)
        if config.complement_map is not None and config.vocab_size > len(config.complement_map):
            for i in range(len(config.complement_map), config.vocab_size):
                config.complement_map[i] = i
        self.config = config
        factory_kwargs = {"device": device, "dtype": dtype}
        self.backbone = CaduceusMixerModel(config, **factory_kwargs, **kwargs)

    def forward(self, input_ids: torch.LongTensor = None, inputs_embeds: Optional[torch.FloatTensor] = None,
                output_hidden_states: Optional[bool] = None, return_dict: Optional[bool] = None,
                ) -> Union[torch.Tensor, Tuple, BaseModelOutputWithNoAttention]:
        output_hidden_states = (output_hidden_states if output_hidden_states is not None
                                else self.config.output_hidden_states)
        return_dict = return_dict if return_dict is not None else self._return_dict_default
        hidden_states, all_hidden_states = self.backbone(input_ids, inputs_embeds=inputs_embeds,
                                                         output_hidden_states=output_hidden_states)
        if return_dict:
            return BaseModelOutputWithNoAttention(last_hidden_state=hidden_states,
                                                  hidden_states=all_hidden_states if output_hidden_states else None)
        elif output_hidden_states:
            return hidden_states, all_hidden_states
        else:
            return hidden_states


class CaduceusForMaskedLM(CaduceusPreTrainedModel):
    """modeling_caduceus.py:392-492."""

    def __init__(self, config: CaduceusConfig, device=None, dtype=None, **kwargs):
        super().__init__(config, **kwargs)
        factory_kwargs = {"device": device, "dtype": dtype}
        self.caduceus = Caduceus(config, **factory_kwargs, **kwargs)
        if config.rcps:
            self.lm_head = RCPSLMHead(complement_map=self.config.complement_map, vocab_size=self.config.vocab_size,
                                      true_dim=config.d_model, dtype=dtype)
        else:
            self.lm_head = nn.Linear(config.d_model, self.config.vocab_size, bias=False, **factory_kwargs)
        self.post_init()

    def get_input_embeddings(self):
        return self.caduceus.backbone.embeddings.word_embeddings

    def set_input_embeddings(self, value):
        if self.config.rcps:
            raise NotImplementedError("Setting input embeddings for RCPS LM is not supported.")
        self.caduceus.backbone.embeddings.word_embeddings = value

    def get_output_embeddings(self):
        return self.lm_head

    def set_output_embeddings(self, new_embeddings):
        if self.config.rcps:
            raise NotImplementedError("Setting output embeddings for RCPS LM is not supported.")
        self.lm_head = new_embeddings

    def tie_weights(self, missing_keys=None, **kwargs):
        """Tie weights, accounting for RCPS (modeling_caduceus.py:434-439).  Accepts the keyword arguments newer
        transformers releases pass (`missing_keys`, `recompute_mapping`)."""
        if getattr(self.config, "tie_word_embeddings", True):
            if self.config.rcps:
                self.lm_head.set_weight(self.get_input_embeddings().weight)
            else:
                self.lm_head.weight = self.get_input_embeddings().weight
        self._retie_shared(missing_keys)

    def get_decoder(self):
        return self.caduceus

    def set_decoder(self, decoder):
        self.caduceus = decoder

    def forward(self, input_ids: torch.LongTensor = None, inputs_embeds: Optional[torch.FloatTensor] = None,
                labels: Optional[torch.LongTensor] = None, loss_weights: Optional[torch.FloatTensor] = None,
                output_hidden_states: Optional[bool] = None, return_dict: Optional[bool] = None,
                ) -> Union[Tuple, MaskedLMOutput]:
        output_hidden_states = (output_hidden_states if output_hidden_states is not None
                                else self.config.output_hidden_states)
        return_dict = return_dict if return_dict is not None else self._return_dict_default
        backbone = self.caduceus.backbone
        collect = [] if output_hidden_states else None
        hidden_t = backbone.forward_tframe(input_ids, inputs_embeds, collect)  # (S, B, L, D), never leaves the t-frame
        ignore_index = getattr(self.config, "pad_token_id", None)
        ignore_index = -100 if ignore_index is None else ignore_index
        fused_loss = labels is not None and loss_weights is None
        if self.config.rcps:
            logits, loss = self.lm_head.forward_tframe(hidden_t, labels if fused_loss else None, ignore_index)
        else:
            w = self.lm_head.weight
            if w.shape[0] <= 16 and getattr(self.lm_head, "bias", None) is None:
                logits, loss = ops.lm_head(hidden_t, w, None, labels if fused_loss else None, ignore_index)
            else:
                h0 = hidden_t[0]
                logits = ops.mm(h0.reshape(-1, h0.shape[-1]), w.to(hidden_t.dtype).t()).view(*h0.shape[:-1], w.shape[0])
                if self.lm_head.bias is not None:
                    logits = logits + self.lm_head.bias.to(logits.dtype)
                logits = logits.float()
                loss = cross_entropy(logits, labels, ignore_index=ignore_index) if fused_loss else None
        if labels is not None and loss_weights is not None:
            loss = weighted_cross_entropy(logits, labels, loss_weights, ignore_index=ignore_index)
        all_hidden = tuple(as_requested(engine.from_tframe(h), logits) for h in collect) if collect is not None else None
        if not return_dict:
            output = (logits,) + ((all_hidden,) if output_hidden_states else ())
            return (loss,) + output if loss is not None else output
        return MaskedLMOutput(loss=loss, logits=logits, hidden_states=all_hidden)


class CaduceusForSequenceClassification(CaduceusPreTrainedModel):
    """Sequence-level head on the backbone (reference: modeling_caduceus.py:495-640), computed on
    the t-frame: the backbone's two strands are separate rows there, so "stack the forward half and the flipped
    reverse-complement half, pool over the sequence, score both, average" needs no flip, split or stack at all --
    pooling over positions is order-free for mean / max, and first <-> last simply exchange roles on the second strand.
    Post-hoc conjoining of a non-RCPS model runs its two inputs as ONE batch.

    Constructor arguments, attribute names (`caduceus`, `score`), the loss selection through `config.problem_type` and the
    output containers are the reference's.  `pooling_strategy` "first" / "last" return the evident intent (the reference's
    own code raises TypeError on them: it hands the tensor itself to `moveaxis`, :541-543; recorded in
    tests/golden/downstream.npz)."""

    POOLINGS = ("mean", "max", "first", "last")

    def __init__(self, config: CaduceusConfig, pooling_strategy: str = "mean", conjoin_train: bool = False,
                 conjoin_eval: bool = False, device=None, dtype=None, **kwargs):
        super().__init__(config, **kwargs)
        if pooling_strategy not in self.POOLINGS:
            raise NotImplementedError(f"Pooling strategy `{pooling_strategy}` not implemented.")
        self.pooling_strategy = pooling_strategy
        self.num_labels = kwargs.get("num_labels", config.num_labels)
        self.caduceus = Caduceus(config, device=device, dtype=dtype, **kwargs)
        self.score = nn.Linear(config.d_model, self.num_labels, bias=False)
        self.conjoin_train, self.conjoin_eval = conjoin_train, conjoin_eval
        self.post_init()
        self.init_scorer()

    def init_scorer(self, initializer_range=0.02):
        cfg = self.config.initializer_cfg or {}
        self.score.weight.data.normal_(std=cfg.get("initializer_range", initializer_range))

    def get_input_embeddings(self):
        return self.caduceus.backbone.embeddings.word_embeddings

    def set_input_embeddings(self, value):
        if self.config.rcps:
            raise NotImplementedError("Setting input embeddings for RCPS LM is not supported.")
        self.caduceus.backbone.embeddings.word_embeddings = value

    def pool_hidden_states(self, hidden_states, sequence_length_dim=1, reversed_positions=False):
        """Pools over the sequence axis.  `reversed_positions`: the rows are stored in the opposite position order to the
        frame the strategy refers to (second t-frame strand), which only matters for first / last."""
        how = self.pooling_strategy
        if how == "mean":
            return hidden_states.mean(dim=sequence_length_dim)
        if how == "max":
            return hidden_states.max(dim=sequence_length_dim).values
        take_last = (how == "last") != reversed_positions
        return hidden_states.select(sequence_length_dim, -1 if take_last else 0)

    def _sequence_loss(self, logits, labels):
        """HF convention (config.problem_type, inferred once from the label dtype / num_labels when unset)."""
        labels = labels.to(logits.device)
        if self.config.problem_type is None:
            if self.num_labels == 1:
                self.config.problem_type = "regression"
            elif labels.dtype in (torch.long, torch.int):
                self.config.problem_type = "single_label_classification"
            else:
                self.config.problem_type = "multi_label_classification"
        kind = self.config.problem_type
        if kind == "regression":
            return F.mse_loss(logits.squeeze(), labels.squeeze()) if self.num_labels == 1 else F.mse_loss(logits, labels)
        if kind == "single_label_classification":
            return F.cross_entropy(logits.view(-1, self.num_labels), labels.view(-1))
        if kind == "multi_label_classification":
            return F.binary_cross_entropy_with_logits(logits, labels)
        return None

    def forward(self, input_ids: torch.LongTensor = None, inputs_embeds: Optional[torch.FloatTensor] = None,
                labels: Optional[torch.LongTensor] = None, output_hidden_states: Optional[bool] = None,
                return_dict: Optional[bool] = None) -> Union[Tuple, SequenceClassifierOutput]:
        return_dict = return_dict if return_dict is not None else self._return_dict_default
        backbone = self.caduceus.backbone
        collect = [] if output_hidden_states else None
        conjoin = (not self.config.rcps) and (self.conjoin_train or (self.conjoin_eval and not self.training))
        if conjoin:
            if input_ids is None:
                raise AssertionError("`input_ids` must be provided for conjoining.")
            if input_ids.ndim != 3:
                raise AssertionError("`input_ids` must be 3D tensor: channels corresponds to forward and rc strands.")
            B = input_ids.shape[0]
            both = torch.cat([input_ids[..., 0], input_ids[..., 1]], dim=0)  # one pass over (2B, L)
            t = backbone.forward_tframe(both, None, collect)[0]               # (2B, L, D)
            pooled = [self.pool_hidden_states(t[:B]), self.pool_hidden_states(t[B:])]
            if collect is not None:  # the reference reports the hidden states of its first (forward-strand) pass
                collect = [h[:, :B] for h in collect]
        else:
            t = backbone.forward_tframe(input_ids, inputs_embeds if self.config.rcps else None, collect)
            pooled = [self.pool_hidden_states(t[0])]
            if self.config.rcps:  # second strand: positions run the other way in the reference's stacked frame
                pooled.append(self.pool_hidden_states(t[1], reversed_positions=True))
        wdt = self.score.weight.dtype
        logits = self.score(pooled[0].to(wdt))
        if len(pooled) == 2:
            logits = (logits + self.score(pooled[1].to(wdt))) / 2
        loss = self._sequence_loss(logits, labels) if labels is not None else None
        hidden = tuple(as_requested(engine.from_tframe(h), logits) for h in collect) if collect is not None else None
        if not return_dict:
            out = (logits,) + ((hidden,) if hidden is not None else ())
            return ((loss,) + out) if loss is not None else out
        return SequenceClassifierOutput(loss=loss, logits=logits, hidden_states=hidden)
