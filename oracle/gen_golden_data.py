"""Generates tests/golden/datapath.npz from the REFERENCE's own data-path code, run in the build container
(the reference cannot travel to the GPU box; only these vectors do).  Run: python oracle/gen_golden_data.py

What is executed from /root/reference (loaded by file path, nothing is copied):
  * src/dataloaders/utils/rc.py            string_reverse_complement
  * src/dataloaders/utils/mlm.py           mlm_getitem            (torch CPU generator, fixed seed)
  * src/dataloaders/datasets/hg38_dataset.py  FastaInterval.__call__ / _compute_interval
  * caduceus/tokenization_caduceus.py      vocabulary (ids of the characters)
`pyfaidx` is not installed here; FastaInterval only uses it as a container (`Fasta(path)[name][start:end]`, `len`), so a
dict-of-strings stand-in is registered under that module name for the duration of this script.  The reference's own
arithmetic (interval shifting, reverse complement, tokenisation, MLM sampling) is what produces every stored value."""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "datapath.npz")


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class _Seq(str):
    def __getitem__(self, k):
        return _Seq(str.__getitem__(self, k))


class _Fasta(dict):
    def __init__(self, path):
        super().__init__()
        name = None
        for line in open(path):
            line = line.strip()
            if line.startswith(">"):
                name = line[1:].split()[0]
                self[name] = ""
            elif name is not None:
                self[name] = self[name] + line
        for k in list(self):
            self[k] = _Seq(self[k])


def main():
    for pkg in ("src", "src.dataloaders", "src.dataloaders.utils", "src.dataloaders.datasets"):
        sys.modules.setdefault(pkg, types.ModuleType(pkg))
    rc = load("src.dataloaders.utils.rc", f"{REF}/src/dataloaders/utils/rc.py")
    mlm = load("src.dataloaders.utils.mlm", f"{REF}/src/dataloaders/utils/mlm.py")
    fake = types.ModuleType("pyfaidx")
    fake.Fasta = _Fasta
    sys.modules["pyfaidx"] = fake
    hg = load("src.dataloaders.datasets.hg38_dataset", f"{REF}/src/dataloaders/datasets/hg38_dataset.py")

    rng = np.random.default_rng(7)
    out = {}
    # 1. reverse complement + tokenisation of strings with lower case, N and foreign characters
    alphabet = np.array(list("ACGTNacgtnRYX-"))
    probs = np.array([6, 6, 6, 6, 1, 2, 2, 2, 2, 0.5, 0.2, 0.2, 0.2, 0.1]); probs = probs / probs.sum()
    strings = ["".join(rng.choice(alphabet, size=n, p=probs)) for n in (1, 7, 64, 257, 1000)]
    out["rc_in"] = np.array(strings)
    out["rc_out"] = np.array([rc.string_reverse_complement(s) for s in strings])
    sys.path.insert(0, f"{REF}/caduceus")
    try:
        tok_mod = load("ref_tokenization_caduceus", f"{REF}/caduceus/tokenization_caduceus.py")
        tok = tok_mod.CaduceusTokenizer(model_max_length=1024)
        vocab = dict(tok.get_vocab())
        enc = [tok(s, add_special_tokens=False)["input_ids"] for s in strings]
    except Exception as ex:  # transformers-version incompatibility of the reference tokenizer
        print("reference tokenizer not constructible here:", repr(ex))
        src = open(f"{REF}/caduceus/tokenization_caduceus.py").read()
        assert '"[CLS]": 0' in src and '"[UNK]": 6' in src  # the vocabulary literal the ids below come from
        vocab = {**{t: i for i, t in enumerate(["[CLS]", "[SEP]", "[BOS]", "[MASK]", "[PAD]", "[RESERVED]", "[UNK]"])},
                 **{c: 7 + i for i, c in enumerate("ACGTN")}}
        enc = None
    out["vocab_keys"] = np.array(list(vocab.keys()))
    out["vocab_vals"] = np.array(list(vocab.values()), dtype=np.int64)
    if enc is not None:
        out["tok_ids"] = np.array([np.array(e, dtype=np.int64) for e in enc], dtype=object)

    # 2. FastaInterval on a synthetic genome (short chromosomes exercise every clamp)
    fa = "/tmp/_golden_genome.fa"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import data_oracle
    chroms = data_oracle.SYNTH_CHROMS
    data_oracle.write_synthetic_genome(fa)
    fi = hg.FastaInterval(fasta_file=fa, rc_aug=False)
    cases, starts, ends, heads, tails, sums = [], [], [], [], [], []
    for name, n in chroms.items():
        for start in (0, 5000, n - 2 ** 20, n - 2 ** 20 + 4000, max(0, n - 2 ** 19)):
            if start < 0:
                continue
            for max_length in (1024, 131072, 2 ** 20):
                shifts = 2 ** 20 // max_length
                for i_shift in sorted({0, shifts // 2, shifts - 1}):
                    s = fi(name, start, start + 2 ** 20, max_length=max_length, i_shift=i_shift)
                    cases.append((list(chroms).index(name), start, max_length, i_shift, len(s)))
                    heads.append(s[:16].ljust(16)), tails.append(s[-16:].rjust(16))
                    sums.append(int(np.frombuffer(s.encode(), dtype=np.uint8).astype(np.int64).sum()))
    out["fa_cases"] = np.array(cases, dtype=np.int64)
    out["fa_heads"], out["fa_tails"], out["fa_sums"] = np.array(heads), np.array(tails), np.array(sums, dtype=np.int64)
    out["fa_chrom_names"] = np.array(list(chroms)); out["fa_chrom_lens"] = np.array(list(chroms.values()), dtype=np.int64)
    out["fa_seed"] = np.array([7])
    # the genome itself is re-generated by the test from the same seed and procedure (3 MB of text is not committed);
    # store its byte checksum to pin that
    out["fa_file_sum"] = np.array([int(np.frombuffer(open(fa, "rb").read(), dtype=np.uint8).astype(np.int64).sum())])
    with pytest_raises(ValueError):
        fi._compute_interval(0, 2 ** 21, 2 ** 21, 0)

    # 3. mlm_getitem statistics with the reference sampler (distribution pin, not a sample-by-sample pin)
    class Tok:
        pad_token_id, mask_token = vocab["[PAD]"], "[MASK]"

        def convert_tokens_to_ids(self, t):
            return vocab[t]

        def __len__(self):
            return 12
    torch.manual_seed(2222)
    seq = torch.randint(7, 11, (400000,))
    data, target = mlm.mlm_getitem(seq, mlm_probability=0.15, contains_eos=False, tokenizer=Tok())
    tgt = target != vocab["[PAD]"]
    out["mlm_rates"] = np.array([tgt.float().mean().item(),                                 # targets
                                 (data[tgt] == vocab["[MASK]"]).float().mean().item(),      # -> [MASK]
                                 ((data[tgt] != vocab["[MASK]"]) & (data[tgt] != seq[tgt])).float().mean().item(),
                                 (data[~tgt] == seq[~tgt]).float().mean().item(),            # untouched elsewhere
                                 (target[tgt] == seq[tgt]).float().mean().item()])           # labels = original ids
    out["mlm_random_word_max"] = np.array([int(data[tgt & (data != vocab["[MASK]"])].max())])
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(cases), "interval cases; mlm rates", out["mlm_rates"])


class pytest_raises:
    def __init__(self, exc):
        self.exc = exc

    def __enter__(self):
        return self

    def __exit__(self, et, ev, tb):
        assert et is not None and issubclass(et, self.exc), "reference did not raise"
        return True


if __name__ == "__main__":
    main()
