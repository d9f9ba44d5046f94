"""Variant-effect embedding extraction: the forward-only user of the backbone at seqlen 131k
that /root/reference/vep_embeddings.py:172-402 implements.

For every variant the reference embeds the reference-allele and the alternate-allele sequence, takes the mean of the
last hidden states over a 1536-bp window centred on the variant (clamped at the sequence ends), does the same for the
reverse-complement strand (the second channel half flipped back for RCPS models, a second forward on the RC input
otherwise) and stores `concat_avg_ws = [ref | alt]` and `rc_concat_avg_ws`.  Work is sharded over ranks exactly like
`DistributedSampler(shuffle=False, drop_last=True)` + `DataLoader(drop_last=True)`.
Differences by design: bf16 autocast by default (float16 is accepted and computed in fp32), and the four forwards of a Ph model / two of an
RCPS model are batched into one launch sequence per step.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional

import torch

WINDOW_SIZE_BP = 1536  # vep_embeddings.py:26


def find_variant_idx(ref_ids: torch.Tensor, alt_ids: torch.Tensor, rc: bool = False) -> torch.Tensor:
    """vep_embeddings.py:172-195, batched: the centre token if it differs, else the LAST differing token, else -1.
    (B, L) x (B, L) -> (B,) int64.  `rc=True` uses the reverse-complement centre L // 2 - 1."""
    L = ref_ids.shape[1]
    centre = L // 2 - 1 if rc else L // 2
    diff = ref_ids != alt_ids
    pos = torch.arange(L, device=ref_ids.device).expand_as(diff)
    last = torch.where(diff, pos, torch.full_like(pos, -1)).max(dim=1).values
    return torch.where(diff[:, centre], torch.full_like(last, centre), last)


def window_mean(hidden: torch.Tensor, variant_idx: torch.Tensor, window_tokens: int) -> torch.Tensor:
    """Mean of `hidden` (B, L, C) over positions variant_idx + [-w // 2, w // 2], clamped to [0, L - 1]
    (vep_embeddings.py:297-305: clamping REPEATS the edge token, it does not shrink the window)."""
    start, end = -window_tokens // 2, window_tokens // 2 + 1
    idx = torch.arange(start, end, device=hidden.device).unsqueeze(0) + variant_idx.to(hidden.device).unsqueeze(1)
    idx = idx.clamp_(0, hidden.size(1) - 1)
    return torch.gather(hidden, 1, idx.unsqueeze(-1).expand(-1, -1, hidden.size(2))).mean(dim=1)


@torch.no_grad()
def embed_variants(backbone, batch: Dict[str, torch.Tensor], rcps: bool, bp_per_token: int = 1,
                   autocast_dtype: Optional[torch.dtype] = torch.bfloat16) -> Dict[str, torch.Tensor]:
    """One step of vep_embeddings.py:352-392.  `backbone(input_ids)` returns the last hidden states (B, L, C) -- e.g.
    `lambda ids: model.caduceus(ids, return_dict=False)`.  batch: ref_input_ids, alt_input_ids, variant_idx and, for
    non-RCPS models, ref_rc_input_ids / alt_rc_input_ids."""
    dev = next(iter(batch.values())).device if not hasattr(backbone, "parameters") else next(backbone.parameters()).device
    ids = [batch["alt_input_ids"], batch["ref_input_ids"]]
    if not rcps:
        ids += [batch["alt_rc_input_ids"], batch["ref_rc_input_ids"]]
    B = ids[0].shape[0]
    stacked = torch.cat([t.to(dev) for t in ids], 0)  # one launch sequence for all forwards of the step
    enabled = autocast_dtype is not None and dev.type == "cuda"
    with torch.autocast(device_type=dev.type, dtype=autocast_dtype or torch.bfloat16, enabled=enabled):
        out = backbone(stacked)
    alt, ref = out[:B], out[B:2 * B]
    if rcps:
        half = out.size(-1) // 2
        alt_rc, ref_rc = alt[..., half:].flip(dims=[1, 2]), ref[..., half:].flip(dims=[1, 2])
        alt, ref = alt[..., :half], ref[..., :half]
    else:
        alt_rc, ref_rc = out[2 * B:3 * B].flip(dims=[1]), out[3 * B:].flip(dims=[1])
    w = WINDOW_SIZE_BP // bp_per_token
    v = batch["variant_idx"]
    return {"concat_avg_ws": torch.cat([window_mean(ref, v, w), window_mean(alt, v, w)], -1),
            "rc_concat_avg_ws": torch.cat([window_mean(ref_rc, v, w), window_mean(alt_rc, v, w)], -1)}


def shard_batches(n_items: int, rank: int, world: int, batch_size: int) -> List[List[int]]:
    """Indices each rank processes: DistributedSampler(shuffle=False, drop_last=True) followed by a DataLoader with
    drop_last=True (vep_embeddings.py:309-338)."""
    per_rank = n_items // world
    mine = list(range(rank, per_rank * world, world))
    return [mine[i:i + batch_size] for i in range(0, len(mine) - batch_size + 1, batch_size)]


@torch.no_grad()
def dump_embeddings(backbone, dataset: Dict[str, torch.Tensor], rcps: bool, batch_size: int, rank: int = 0, world: int = 1,
                    bp_per_token: int = 1, passthrough: Iterable[str] = ("chromosome", "labels", "distance_to_nearest_tss",
                                                                         "tissue_embed"),
                    autocast_dtype: Optional[torch.dtype] = torch.bfloat16) -> Dict[str, torch.Tensor]:
    """This rank's share of vep_embeddings.py:340-396 as one dict of CPU tensors (the caller saves / gathers it).
    dataset: column name -> tensor with one row per variant."""
    n = next(iter(dataset.values())).shape[0]
    store: Dict[str, List[torch.Tensor]] = {}
    for idx in shard_batches(n, rank, world, batch_size):
        sel = torch.tensor(idx)
        batch = {k: v[sel] for k, v in dataset.items()}
        res = embed_variants(backbone, batch, rcps, bp_per_token, autocast_dtype)
        for k in passthrough:
            if k in batch:
                store.setdefault(k, []).append(batch[k].cpu())
        for k, v in res.items():
            store.setdefault(k, []).append(v.float().cpu())
    return {k: torch.cat(v, 0) for k, v in store.items()}
