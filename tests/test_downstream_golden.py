"""Rows f-1 / f-3 against vectors produced by the REFERENCE's own classes and functions (oracle/gen_golden_downstream.py:
CaduceusForSequenceClassification, DNAEmbeddingModelCaduceus, SequenceDecoder, find_variant_idx).  Where the reference's
own code raises (recorded in the fixture), this package's documented behaviour is asserted instead."""
import json
import os

import numpy as np
import pytest
import torch

from caduceus_amd import (CaduceusConfig, CaduceusForSequenceClassification, DNAEmbeddingModelCaduceus, SequenceDecoder,
                          vep)
from conftest import GOLDEN, _intkeys

Z = np.load(os.path.join(GOLDEN, "downstream.npz"))
FP32 = dict(rtol=6e-4, atol=2e-3)


def _group(prefix):
    return {k[len(prefix):]: Z[k] for k in Z.files if k.startswith(prefix)}


def _meta(g):
    return json.loads(bytes(g["cfg"]).decode(), object_hook=_intkeys)


SEQCLS = ["ps_mean", "ps_max", "ps_regression", "ps_multilabel", "ph_plain", "ph_conjoin_train", "ph_conjoin_eval"]


@pytest.mark.parametrize("name", SEQCLS)
def test_sequence_classification_matches_reference(backend, name):
    _, dev = backend
    g = _group(f"seqcls/{name}/")
    meta = _meta(g)
    cfg = CaduceusConfig(**meta["cfg"], num_labels=meta["num_labels"])
    model = CaduceusForSequenceClassification(cfg, **meta["ctor"])
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    model.load_state_dict(sd, strict=True)
    model = model.to(dev).train(meta["train"])
    ids, labels = torch.from_numpy(g["input_ids"]).to(dev), torch.from_numpy(g["labels"]).to(dev)
    out = model(ids, labels=labels)
    torch.testing.assert_close(out.logits.cpu(), torch.from_numpy(g["logits"]), **FP32)
    torch.testing.assert_close(out.loss.cpu(), torch.from_numpy(g["loss"]), **FP32)
    out.loss.backward()
    named = dict(model.named_parameters())
    for k, want in g.items():
        if k.startswith("grad/"):
            want = torch.from_numpy(want)
            got = named[k[5:]].grad.cpu()
            torch.testing.assert_close(got, want, rtol=6e-4, atol=2e-3 * max(1.0, float(want.abs().max())))
    tup = model(ids, labels=labels, return_dict=False, output_hidden_states=True)
    assert len(tup) == int(g["tuple_len"])
    torch.testing.assert_close(tup[1].cpu(), torch.from_numpy(g["logits"]), **FP32)


def test_sequence_classification_first_last_pooling(backend):
    """The reference raises TypeError for these two strategies (fixture flag); here they return what their names say, on
    the strand pair the reference would have stacked: first/last of the forward half and of the flipped RC half."""
    _, dev = backend
    assert int(Z["seqcls/first_last_raise_typeerror"]) == 1
    g = _group("seqcls/ps_mean/")
    meta = _meta(g)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    ids = torch.from_numpy(g["input_ids"]).to(dev)
    for how in ("first", "last"):
        model = CaduceusForSequenceClassification(CaduceusConfig(**meta["cfg"], num_labels=3), pooling_strategy=how)
        model.load_state_dict(sd, strict=True)
        model = model.to(dev).eval()
        with torch.no_grad():
            logits = model(ids).logits
            h = model.caduceus(ids).last_hidden_state  # (B, L, 2D) reference frame
        D = h.shape[-1] // 2
        stacked = torch.stack([h[..., :D], torch.flip(h[..., D:], dims=[1, 2])], dim=-1)
        pooled = stacked[:, 0 if how == "first" else -1]
        want = (model.score(pooled[..., 0]) + model.score(pooled[..., 1])) / 2
        torch.testing.assert_close(logits, want, rtol=1e-5, atol=1e-6)
    with pytest.raises(NotImplementedError):
        CaduceusForSequenceClassification(CaduceusConfig(**meta["cfg"]), pooling_strategy="median")


@pytest.mark.parametrize("name", ["ps", "ph_conjoin", "ph_plain"])
def test_embedding_model_matches_reference(backend, name):
    _, dev = backend
    g = _group(f"embed/{name}/")
    meta = _meta(g)
    emb = DNAEmbeddingModelCaduceus(CaduceusConfig(**meta["cfg"]), **meta["ctor"])
    emb.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}, strict=True)
    emb = emb.to(dev).train()
    hidden, none = emb(torch.from_numpy(g["input_ids"]).to(dev))
    assert none is None
    torch.testing.assert_close(hidden.detach().cpu(), torch.from_numpy(g["hidden"]), **FP32)


def test_sequence_decoder_matches_reference():
    x, x2 = torch.from_numpy(Z["decoder/x"]), torch.from_numpy(Z["decoder/x2"])
    lens = Z["decoder/lengths"].tolist()

    def load(dec, p):
        with torch.no_grad():
            dec.output_transform.weight.copy_(torch.from_numpy(Z[p + "w"]))
            dec.output_transform.bias.copy_(torch.from_numpy(Z[p + "b"]))
        return dec

    checked = 0
    for mode in ("last", "first", "pool", "sum"):
        for l_out in (0, 3):
            p = f"decoder/{mode}/{l_out}/"
            if p + "y" in Z.files:
                dec = load(SequenceDecoder(6, d_output=4, l_output=l_out, mode=mode), p)
                torch.testing.assert_close(dec(x), torch.from_numpy(Z[p + "y"]), rtol=1e-5, atol=1e-5)
                checked += 1
        p = f"decoder/{mode}/lengths/"
        if p + "y" in Z.files:
            dec = SequenceDecoder(6, d_output=None, l_output=0, mode=mode, use_lengths=True)
            torch.testing.assert_close(dec(x, lengths=lens), torch.from_numpy(Z[p + "y"]), rtol=1e-5, atol=1e-5)
            checked += 1
        p = f"decoder/{mode}/conjoin/"
        if p + "y" in Z.files:
            dec = load(SequenceDecoder(6, d_output=2, l_output=0, mode=mode, conjoin_train=True), p)
            torch.testing.assert_close(dec(x2), torch.from_numpy(Z[p + "y"]), rtol=1e-5, atol=1e-5)
            checked += 1
    dec = load(SequenceDecoder(6, d_output=2, l_output=0, mode="pool", conjoin_train=True), "decoder/step/")
    torch.testing.assert_close(dec.step(x), torch.from_numpy(Z["decoder/step/y"]), rtol=1e-5, atol=1e-5)
    assert checked >= 11


def test_find_variant_idx_matches_reference():
    ref, alt = torch.from_numpy(Z["vep/ref_input_ids"]), torch.from_numpy(Z["vep/alt_input_ids"])
    ref_rc, alt_rc = torch.from_numpy(Z["vep/ref_rc_input_ids"]), torch.from_numpy(Z["vep/alt_rc_input_ids"])
    assert vep.find_variant_idx(ref, alt).tolist() == Z["vep/variant_idx"].tolist()
    assert vep.find_variant_idx(ref_rc, alt_rc, rc=True).tolist() == Z["vep/rc_variant_idx"].tolist()
    assert vep.WINDOW_SIZE_BP == int(Z["vep/window_size_bp"])


@pytest.mark.parametrize("name", ["ps", "ph"])
def test_vep_dump_matches_reference_dump_embeddings(backend, name):
    """`vep.dump_embeddings` (window means, RCPS strand split / the two extra RC forwards, batching with drop_last) against
    the tensors the REFERENCE's own `dump_embeddings` wrote for the same backbone weights and token ids
    (/root/reference/vep_embeddings.py:275-404, executed by oracle/gen_golden_downstream.py::gen_vep_dump)."""
    from caduceus_amd import Caduceus
    _, dev = backend
    g = _group(f"vepdump/{name}/")
    meta = _meta(g)
    cfg = CaduceusConfig(**meta["cfg"])
    model = Caduceus(cfg)
    model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}, strict=True)
    model = model.to(dev).eval()
    backbone = lambda ids: model(ids, return_dict=False)  # noqa: E731
    for split in ("train", "test"):
        data = {k[len(split) + 4:]: torch.from_numpy(v).to(dev) for k, v in g.items() if k.startswith(f"{split}/in/")}
        data["tissue_embed"] = torch.arange(data["labels"].shape[0])  # label encoding is data preparation: passed through
        out = vep.dump_embeddings(backbone, data, rcps=bool(cfg.rcps), batch_size=meta["batch_size"],
                                  bp_per_token=meta["bp_per_token"], autocast_dtype=None)
        n = g[f"{split}/concat_avg_ws"].shape[0]
        assert n == (data["labels"].shape[0] // meta["batch_size"]) * meta["batch_size"]  # drop_last
        for k in ("concat_avg_ws", "rc_concat_avg_ws"):
            torch.testing.assert_close(out[k], torch.from_numpy(g[f"{split}/{k}"]), **FP32)
        for k in ("chromosome", "labels", "distance_to_nearest_tss"):
            assert out[k].tolist() == g[f"{split}/{k}"].tolist()
        assert out["tissue_embed"].tolist() == list(range(n))


def test_vep_rank_sharding_matches_distributed_sampler():
    """`vep.shard_batches` against index lists produced by torch's DistributedSampler(shuffle=False, drop_last=True), the
    sampler the reference builds (vep_embeddings.py:333-338), followed by DataLoader(drop_last=True) batching."""
    for n, world in ((11, 2), (9, 4), (8, 3)):
        for r in range(world):
            want = Z[f"vepdump/shard/{n}_{world}_{r}"].tolist()
            for bs in (1, 2, 3):
                batches = [want[i:i + bs] for i in range(0, len(want) - bs + 1, bs)]
                assert vep.shard_batches(n, r, world, bs) == batches
