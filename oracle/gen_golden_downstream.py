"""Golden vectors for the rows next to the hot path, produced by the REFERENCE's own code.

Run in the build container only (needs /root/reference):   python oracle/gen_golden_downstream.py

What runs:
  * `CaduceusForSequenceClassification`  (/root/reference/caduceus/modeling_caduceus.py:495-640) -- importable like the MLM
    class through oracle/ref_harness (the un-vendored mamba_ssm is served by the HF MambaMixer adapter);
  * `DNAEmbeddingModelCaduceus`          (/root/reference/src/models/sequence/dna_embedding.py:156-195);
  * `SequenceDecoder`                    (/root/reference/src/tasks/decoders.py:39-161);
  * `find_variant_idx`                   (/root/reference/vep_embeddings.py:172-195), on ids from the reference's tokenizer.
The last three live in modules whose IMPORT statements pull packages this image does not have (flash_attn, hydra-based
src.utils, enformer_pytorch, the Hyena backbones).  None of that code is on the path of the functions above, so the
generator registers EMPTY placeholder modules for those names before importing (below, `_placeholder`): they satisfy
`import x` / `from x import Name` only -- every placeholder attribute raises if it is ever called.  The vectors are
therefore still the output of the reference's own function bodies.

Only numbers are stored (tests/golden/downstream.npz): parameters, inputs, outputs, gradients.
"""
import importlib
import importlib.util
import json
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_harness"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, HERE)

from gen_golden import COMP, make_cfg, randomize_, _intkeys  # noqa: E402  (the MLM generator's helpers)

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "downstream.npz")


class _Never:
    def __init__(self, name):
        self._name = name

    def __call__(self, *a, **k):
        raise RuntimeError(f"placeholder {self._name} was called: it is not part of the reference path being recorded")

    def __getattr__(self, item):
        return _Never(f"{self._name}.{item}")


def _placeholder(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__path__ = []  # behaves as a package for sub-imports
    m.__getattr__ = lambda item, _n=name: _Never(f"{_n}.{item}")
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    if "." in name:  # `import a.b.c` then `a.b.c.x`: the attribute chain must exist on the (real or placeholder) parents
        parent, child = name.rsplit(".", 1)
        try:
            pm = sys.modules.get(parent) or importlib.import_module(parent)
            setattr(pm, child, m)
        except Exception:
            pass
    return m


def _install_placeholders():
    _placeholder("flash_attn")
    _placeholder("flash_attn.utils")
    _placeholder("flash_attn.utils.generation", GenerationMixin=type("GenerationMixin", (), {}))
    _placeholder("flash_attn.ops")
    _placeholder("flash_attn.ops.fused_dense")
    _placeholder("mamba_ssm.models")
    _placeholder("mamba_ssm.models.config_mamba", MambaConfig=type("MambaConfig", (), {}))
    _placeholder("mamba_ssm.models.mixer_seq_simple", MixerModel=type("MixerModel", (), {}), _init_weights=_Never("x"))
    _placeholder("src.models.sequence.long_conv_lm", LMBackbone=type("LMBackbone", (), {}), _init_weights=_Never("x"))
    _placeholder("src.models.nn")
    _placeholder("src.models.nn.utils")
    _placeholder("src.utils")
    _placeholder("src.utils.train", get_logger=lambda name=None, **k: logging.getLogger(name or "ref"))
    _placeholder("enformer_pytorch")
    # vep_embeddings.py imports one pure-python helper through the `src.dataloaders` package, whose __init__ pulls every
    # dataset (and their missing third-party packages): load that single reference FILE directly instead
    _placeholder("src.dataloaders")
    _placeholder("src.dataloaders.utils")
    spec = importlib.util.spec_from_file_location("src.dataloaders.utils.rc", "/root/reference/src/dataloaders/utils/rc.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["src.dataloaders.utils.rc"] = mod
    spec.loader.exec_module(mod)
    sys.modules["src.dataloaders.utils"].rc = mod


def _np(t):
    return t.detach().cpu().numpy()


def gen_seqcls(rec):
    from caduceus.configuration_caduceus import CaduceusConfig
    from caduceus.modeling_caduceus import CaduceusForSequenceClassification

    class Ref(CaduceusForSequenceClassification):
        def tie_weights(self, *a, **k):  # transformers 5.x passes kwargs
            return None

    cases = [  # name, cfg overrides, ctor kwargs, num_labels, label kind, train flag, 3-D conjoin input
        ("ps_mean", dict(), dict(pooling_strategy="mean"), 3, "long", True, False),
        ("ps_max", dict(), dict(pooling_strategy="max"), 3, "long", True, False),
        ("ps_regression", dict(), dict(pooling_strategy="mean"), 1, "float1", True, False),
        ("ps_multilabel", dict(), dict(pooling_strategy="mean"), 4, "floatN", True, False),
        ("ph_plain", dict(rcps=False), dict(pooling_strategy="mean"), 3, "long", True, False),
        ("ph_conjoin_train", dict(rcps=False), dict(pooling_strategy="max", conjoin_train=True), 3, "long", True, True),
        ("ph_conjoin_eval", dict(rcps=False), dict(pooling_strategy="mean", conjoin_eval=True), 3, "long", False, True),
    ]
    # "first" / "last" pooling cannot be recorded: the reference's own code raises TypeError there
    # (modeling_caduceus.py:541-543 passes the tensor itself as moveaxis' first argument); recorded as a fact below
    probe = Ref(CaduceusConfig(**json.loads(json.dumps(make_cfg()), object_hook=_intkeys), num_labels=2),
                pooling_strategy="first")
    try:
        probe(torch.randint(7, 11, (1, 8)))
        rec["seqcls/first_last_raise_typeerror"] = np.array(0)
    except TypeError:
        rec["seqcls/first_last_raise_typeerror"] = np.array(1)
    for i, (name, over, kw, nl, kind, train, conj) in enumerate(cases):
        gen = torch.Generator().manual_seed(100 + i)
        torch.manual_seed(100 + i)
        cfg_dict = make_cfg(**over)
        cfg = CaduceusConfig(**json.loads(json.dumps(cfg_dict), object_hook=_intkeys), num_labels=nl)
        model = Ref(cfg, **kw)
        randomize_(model, gen)
        with torch.no_grad():
            model.score.weight.copy_(0.5 * torch.randn(model.score.weight.shape, generator=gen))
        model.train(train)
        B, L = 3, 40
        ids = torch.randint(7, 11, (B, L), generator=gen)
        if conj:
            comp = torch.tensor([COMP.get(j, j) for j in range(16)])
            ids = torch.stack([ids, comp[ids.flip(-1)]], dim=-1)
        labels = {"long": torch.randint(0, max(nl, 2), (B,), generator=gen),
                  "float1": torch.randn(B, generator=gen),
                  "floatN": (torch.rand(B, nl, generator=gen) > 0.5).float()}[kind]
        out = model(ids, labels=labels, return_dict=True)
        out.loss.backward()
        p = f"seqcls/{name}/"
        rec[p + "cfg"] = np.frombuffer(json.dumps(dict(cfg=cfg_dict, ctor=kw, num_labels=nl, train=train)).encode(),
                                       dtype=np.uint8)
        for k, v in model.state_dict().items():
            rec[p + "sd/" + k] = _np(v)
        rec[p + "input_ids"], rec[p + "labels"] = _np(ids), _np(labels)
        rec[p + "logits"], rec[p + "loss"] = _np(out.logits), _np(out.loss)
        rec[p + "grad/score.weight"] = _np(model.score.weight.grad)
        first = "caduceus.backbone.layers.0.mixer." + ("submodule." if cfg.rcps else "") + "mamba_fwd.in_proj.weight"
        rec[p + "grad/" + first] = _np(dict(model.named_parameters())[first].grad)
        tup = model(ids, labels=labels, return_dict=False, output_hidden_states=True)
        rec[p + "tuple_len"] = np.array(len(tup))
        print(f"seqcls {name}: loss={float(out.loss):.6f} problem_type={model.config.problem_type}")


def gen_embedding_and_decoder(rec):
    from caduceus.configuration_caduceus import CaduceusConfig
    emb_mod = importlib.import_module("src.models.sequence.dna_embedding")
    dec_mod = importlib.import_module("src.tasks.decoders")
    comp = torch.tensor([COMP.get(j, j) for j in range(16)])
    for i, (name, over, kw, conj) in enumerate([("ps", dict(), dict(), False),
                                                ("ph_conjoin", dict(rcps=False), dict(conjoin_train=True), True),
                                                ("ph_plain", dict(rcps=False), dict(), False)]):
        gen = torch.Generator().manual_seed(200 + i)
        torch.manual_seed(200 + i)
        cfg_dict = make_cfg(**over)
        cfg = CaduceusConfig(**json.loads(json.dumps(cfg_dict), object_hook=_intkeys))
        emb = emb_mod.DNAEmbeddingModelCaduceus(cfg, **kw)
        randomize_(emb, gen)
        emb.train()
        ids = torch.randint(7, 11, (2, 36), generator=gen)
        if conj:
            ids = torch.stack([ids, comp[ids.flip(-1)]], dim=-1)
        hidden, none = emb(ids)
        assert none is None
        p = f"embed/{name}/"
        rec[p + "cfg"] = np.frombuffer(json.dumps(dict(cfg=cfg_dict, ctor=kw)).encode(), dtype=np.uint8)
        for k, v in emb.state_dict().items():
            rec[p + "sd/" + k] = _np(v)
        rec[p + "input_ids"], rec[p + "hidden"] = _np(ids), _np(hidden)
        print(f"embed {name}: hidden {tuple(hidden.shape)}")
    # decoder: every restriction mode, with / without lengths, conjoined, on fixed random features
    gen = torch.Generator().manual_seed(300)
    x = torch.randn(3, 12, 6, generator=gen)
    x2 = torch.randn(3, 12, 6, 2, generator=gen)
    rec["decoder/x"], rec["decoder/x2"] = _np(x), _np(x2)
    lens = [12, 9, 5]
    rec["decoder/lengths"] = np.array(lens)
    for mode in ("last", "first", "pool", "sum"):
        for l_out in (0, 3):
            torch.manual_seed(310)
            dec = dec_mod.SequenceDecoder(6, d_output=4, l_output=l_out, mode=mode)
            rec[f"decoder/{mode}/{l_out}/w"], rec[f"decoder/{mode}/{l_out}/b"] = \
                _np(dec.output_transform.weight), _np(dec.output_transform.bias)
            try:
                rec[f"decoder/{mode}/{l_out}/y"] = _np(dec(x))
            except Exception as ex:  # e.g. "pool" with l_output > 1 slices the wrong axis in the reference
                rec[f"decoder/{mode}/{l_out}/raises"] = np.frombuffer(type(ex).__name__.encode(), dtype=np.uint8)
        torch.manual_seed(311)
        decl = dec_mod.SequenceDecoder(6, d_output=None, l_output=0, mode=mode, use_lengths=True)
        try:
            rec[f"decoder/{mode}/lengths/y"] = _np(decl(x, lengths=lens))
        except Exception as ex:  # the reference's own restriction indexes the wrong axis on un-batched samples
            rec[f"decoder/{mode}/lengths/raises"] = np.frombuffer(type(ex).__name__.encode(), dtype=np.uint8)
        torch.manual_seed(312)
        decc = dec_mod.SequenceDecoder(6, d_output=2, l_output=0, mode=mode, conjoin_train=True)
        rec[f"decoder/{mode}/conjoin/w"], rec[f"decoder/{mode}/conjoin/b"] = \
            _np(decc.output_transform.weight), _np(decc.output_transform.bias)
        try:  # (B, L, D, 2) strand pairs: only the modes that index the length axis explicitly accept them
            rec[f"decoder/{mode}/conjoin/y"] = _np(decc(x2))
        except Exception as ex:
            rec[f"decoder/{mode}/conjoin/raises"] = np.frombuffer(type(ex).__name__.encode(), dtype=np.uint8)
    torch.manual_seed(313)
    decs = dec_mod.SequenceDecoder(6, d_output=2, l_output=0, mode="pool", conjoin_train=True)
    rec["decoder/step/w"], rec["decoder/step/b"] = _np(decs.output_transform.weight), _np(decs.output_transform.bias)
    rec["decoder/step/y"] = _np(decs.step(x))
    print("decoder ok")


def gen_vep(rec):
    vep = importlib.import_module("vep_embeddings")
    from caduceus.tokenization_caduceus import CaduceusTokenizer
    tok = CaduceusTokenizer(model_max_length=64)
    rng = np.random.default_rng(5)
    L = 48
    refs, alts = [], []
    for j in range(6):
        s = "".join(rng.choice(list("ACGT"), size=L))
        pos = [L // 2, 5, L - 3, L // 2, 11, L // 2][j]
        t = list(s)
        t[pos] = {"A": "C", "C": "G", "G": "T", "T": "A"}[t[pos]]
        if j == 4:  # two differences: the reference keeps the LAST one
            t[30] = {"A": "C", "C": "G", "G": "T", "T": "A"}[t[30]]
        if j == 5:  # no difference at all
            t = list(s)
        refs.append(s)
        alts.append("".join(t))
    # `tokenize_variants` (vep_embeddings.py:134-169) calls tokenizer.batch_encode_plus, which the installed transformers
    # 5.x no longer has; its four columns are rebuilt here with the reference's own tokenizer class and the reference's
    # string_reverse_complement, then handed to the reference's `find_variant_idx`
    rcs = sys.modules["src.dataloaders.utils.rc"].string_reverse_complement
    enc = lambda seqs: [tok(q, add_special_tokens=False, max_length=L, truncation=True)["input_ids"] for q in seqs]
    batch = {"ref_input_ids": enc(refs), "alt_input_ids": enc(alts),
             "ref_rc_input_ids": enc([rcs(q) for q in refs]), "alt_rc_input_ids": enc([rcs(q) for q in alts])}
    rec["vep/ref_seqs"] = np.array(refs)
    rec["vep/alt_seqs"] = np.array(alts)
    for k, v in batch.items():
        rec["vep/" + k] = np.array(v, dtype=np.int64)
    idx, rc_idx = [], []
    for j in range(len(refs)):
        r = vep.find_variant_idx({k: v[j] for k, v in batch.items()})
        idx.append(r["variant_idx"])
        rc_idx.append(r["rc_variant_idx"])
    rec["vep/variant_idx"], rec["vep/rc_variant_idx"] = np.array(idx), np.array(rc_idx)
    rec["vep/window_size_bp"] = np.array(vep.WINDOW_SIZE_BP)
    print("vep ok:", idx, rc_idx)


def gen_vep_dump(rec):
    """The reference's whole `dump_embeddings` (vep_embeddings.py:275-404: DistributedSampler + DataLoader batching, the four
    forwards / the RCPS strand split :363-376, the nested `extract_embeddings` window means :277-307, the per-rank .pt file)
    executed as it stands, on the reference's own `Caduceus` backbone (through oracle/ref_harness), one RCPS and one
    non-RCPS model, world size 1 over gloo.  Only numbers are kept: parameters, token ids, the dumped tensors."""
    import tempfile
    import datasets
    import torch.distributed as dist
    from caduceus.configuration_caduceus import CaduceusConfig
    from caduceus.modeling_caduceus import Caduceus
    vep = importlib.import_module("vep_embeddings")
    comp = torch.tensor([COMP.get(j, j) for j in range(16)])
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29591", rank=0, world_size=1)
    for i, (name, over) in enumerate([("ps", dict()), ("ph", dict(rcps=False))]):
        gen = torch.Generator().manual_seed(400 + i)
        torch.manual_seed(400 + i)
        cfg_dict = make_cfg(**over)
        cfg = CaduceusConfig(**json.loads(json.dumps(cfg_dict), object_hook=_intkeys))
        backbone = Caduceus(cfg)
        randomize_(backbone, gen)
        backbone.eval()
        L = 40
        splits = {}
        for split, n in (("train", 5), ("test", 4)):  # batch size 2, drop_last: the fifth training variant is dropped
            ref = torch.randint(7, 11, (n, L), generator=gen)
            alt = ref.clone()
            pos = torch.tensor([L // 2, 3, L - 2, L // 2, 17][:n])
            alt[torch.arange(n), pos] = 7 + (alt[torch.arange(n), pos] - 7 + 1) % 4
            cols = {"ref_input_ids": ref, "alt_input_ids": alt, "ref_rc_input_ids": comp[ref.flip(-1)],
                    "alt_rc_input_ids": comp[alt.flip(-1)]}
            vi = [vep.find_variant_idx({k: v[j].tolist() for k, v in cols.items()}) for j in range(n)]
            cols = {k: v.tolist() for k, v in cols.items()}
            cols["variant_idx"] = [r["variant_idx"] for r in vi]
            cols["rc_variant_idx"] = [r["rc_variant_idx"] for r in vi]
            cols["chromosome"] = list(range(1, n + 1))
            cols["labels"] = [j % 2 for j in range(n)]
            cols["distance_to_nearest_tss"] = [100 * j + 7 for j in range(n)]
            cols["tissue"] = [["liver", "lung", "brain"][j % 3] for j in range(n)]
            splits[split] = datasets.Dataset.from_dict(cols)
        ds = datasets.DatasetDict(splits)
        p = f"vepdump/{name}/"
        with tempfile.TemporaryDirectory() as tmp:
            args = types.SimpleNamespace(downstream_save_dir=tmp, name="run", embed_dump_batch_size=2, num_workers=0,
                                         rcps=bool(cfg.rcps), model_name_or_path="caduceus-ref", bp_per_token=128)
            model = lambda ids: backbone(ids).last_hidden_state  # noqa: E731  (DNAEmbeddingModel.forward, :56-60)
            vep.dump_embeddings(args, ds, model, torch.device("cpu"))
            for split in ("train", "test"):
                got = torch.load(os.path.join(tmp, "run", f"{split}_embeds_0.pt"))
                for k, v in got.items():
                    rec[p + f"{split}/{k}"] = _np(v)
        rec[p + "cfg"] = np.frombuffer(json.dumps(dict(cfg=cfg_dict, batch_size=2, bp_per_token=128)).encode(), dtype=np.uint8)
        for k, v in backbone.state_dict().items():
            rec[p + "sd/" + k] = _np(v)
        for split in ("train", "test"):
            for k in ("ref_input_ids", "alt_input_ids", "ref_rc_input_ids", "alt_rc_input_ids", "variant_idx", "chromosome",
                      "labels", "distance_to_nearest_tss"):
                rec[p + f"{split}/in/{k}"] = np.array(ds[split][k], dtype=np.int64)
        print(f"vep dump {name}: train {tuple(rec[p + 'train/concat_avg_ws'].shape)} test {tuple(rec[p + 'test/concat_avg_ws'].shape)}")
    # rank sharding of the same pipeline for world sizes > 1: the reference's sampler class itself
    from torch.utils.data import DistributedSampler
    for n, world in ((11, 2), (9, 4), (8, 3)):
        for r in range(world):
            smp = DistributedSampler(list(range(n)), num_replicas=world, rank=r, shuffle=False, drop_last=True)
            rec[f"vepdump/shard/{n}_{world}_{r}"] = np.array(list(iter(smp)), dtype=np.int64)
    dist.destroy_process_group()


def main():
    _install_placeholders()
    rec = {"meta": np.frombuffer(json.dumps(dict(torch=torch.__version__, generator="oracle/gen_golden_downstream.py")
                                            ).encode(), dtype=np.uint8)}
    gen_seqcls(rec)
    gen_embedding_and_decoder(rec)
    gen_vep(rec)
    gen_vep_dump(rec)
    np.savez_compressed(OUT, **rec)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(rec), "arrays")


if __name__ == "__main__":
    main()
