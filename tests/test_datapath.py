"""the hg38 data path (FASTA slicing, tokenisation, reverse complement, MLM corruption).
Oracle = oracle/data_oracle.py (numpy), pinned to tests/golden/datapath.npz which oracle/gen_golden_data.py produced by
running the reference's own functions.  Kernel parity is bit-exact (integer path)."""
import os

import numpy as np
import pytest
import torch

from caduceus_amd import data as cdata
from oracle import data_oracle as do

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "datapath.npz"), allow_pickle=True)


@pytest.fixture(scope="module")
def genome(tmp_path_factory):
    path = str(tmp_path_factory.mktemp("genome") / "synthetic.fa")
    seqs = do.write_synthetic_genome(path)
    assert int(np.frombuffer(open(path, "rb").read(), dtype=np.uint8).astype(np.int64).sum()) == int(GOLD["fa_file_sum"][0])
    return path, seqs


# ---- oracle vs the reference's own outputs ------------------------------------------------------------------------------
def test_oracle_reverse_complement_and_tokens_match_reference():
    for s, want in zip(GOLD["rc_in"], GOLD["rc_out"]):
        assert do.reverse_complement(str(s)) == str(want)
    vocab = dict(zip(GOLD["vocab_keys"].tolist(), GOLD["vocab_vals"].tolist()))
    assert {k: vocab[k] for k in do.VOCAB} == do.VOCAB
    if "tok_ids" in GOLD.files:
        for s, ids in zip(GOLD["rc_in"], GOLD["tok_ids"]):
            assert do.tokenize(str(s)) == list(ids)


def test_oracle_interval_matches_reference(genome):
    _, seqs = genome
    names, lens = GOLD["fa_chrom_names"].tolist(), GOLD["fa_chrom_lens"].tolist()
    for (ci, start, ml, sh, n), head, tail, csum in zip(GOLD["fa_cases"], GOLD["fa_heads"], GOLD["fa_tails"], GOLD["fa_sums"]):
        s, e = do.hg38_interval(int(start), int(start) + 2 ** 20, int(ml), int(sh), lens[ci])
        got = seqs[names[ci]][s:e]
        assert len(got) == n and got[:16].ljust(16) == str(head) and got[-16:].rjust(16) == str(tail)
        assert int(np.frombuffer(got.encode(), dtype=np.uint8).astype(np.int64).sum()) == int(csum)
    with pytest.raises(ValueError):
        do.hg38_interval(0, 2 ** 21, 2 ** 21, 0, 10 ** 9)


def test_oracle_mlm_rates_match_reference_sampler():
    """mlm.py:4-32 sampled with torch's generator vs the Philox restatement: same distribution."""
    rng = np.random.default_rng(0)
    seq = "".join(rng.choice(np.array(list("ACGT")), size=400000))
    ids, labels = do.tokenize_mlm([seq], len(seq), seed=5)
    orig = np.asarray(do.tokenize(seq))
    tgt = labels[0] != do.PAD
    rates = [tgt.mean(), (ids[0][tgt] == do.MASK).mean(), ((ids[0][tgt] != do.MASK) & (ids[0][tgt] != orig[tgt])).mean(),
             (ids[0][~tgt] == orig[~tgt]).mean(), (labels[0][tgt] == orig[tgt]).mean()]
    want = [0.15, 0.8 + 0.1 / 12, 0.1 * 10 / 12, 1.0, 1.0]  # a random word is [MASK] or the original id 1/12 of the time each
    np.testing.assert_allclose(rates, want, atol=6e-3)
    np.testing.assert_allclose(rates, GOLD["mlm_rates"], atol=1.5e-2)  # the reference's own sample
    assert ids[0][tgt & (ids[0] != do.MASK)].max() <= int(GOLD["mlm_random_word_max"][0]) <= 11


# ---- host C++: FASTA store and interval arithmetic (no GPU needed) ----------------------------------------------------
def test_fasta_store_and_interval_against_golden(genome):
    path, seqs = genome
    names, lens = GOLD["fa_chrom_names"].tolist(), GOLD["fa_chrom_lens"].tolist()
    fi = cdata.FastaInterval(fasta_file=path)
    assert fi.seqs.keys() == names and [fi.chr_lens[n] for n in names] == lens
    for (ci, start, ml, sh, n), head, tail, csum in zip(GOLD["fa_cases"], GOLD["fa_heads"], GOLD["fa_tails"], GOLD["fa_sums"]):
        got = fi(names[ci], int(start), int(start) + 2 ** 20, max_length=int(ml), i_shift=int(sh))
        assert len(got) == n and got[:16].ljust(16) == str(head) and got[-16:].rjust(16) == str(tail)
        assert int(np.frombuffer(got.encode(), dtype=np.uint8).astype(np.int64).sum()) == int(csum)
    assert fi.seqs.fetch("chrB", 55, 190).decode() == seqs["chrB"][55:190]   # crosses line breaks
    with pytest.raises(ValueError):
        cdata.hg38_interval(0, 2 ** 21, 2 ** 21, 0, 10 ** 9)
    with pytest.raises(KeyError):
        fi.seqs.fetch("chrZ", 0, 1)
    with pytest.raises(RuntimeError):
        fi.seqs.fetch("chrC", 0, 2 ** 20 + 1)
    fi_rc = cdata.FastaInterval(fasta_file=path, rc_aug=True, seed=1)
    outs = {fi_rc("chrC", 0, 2 ** 20, max_length=1024, i_shift=3) for _ in range(12)}
    plain = seqs["chrC"][3072:4096]
    assert outs == {plain, do.reverse_complement(plain)}


def test_fasta_open_errors(tmp_path):
    with pytest.raises(AssertionError):
        cdata.FastaInterval(fasta_file=tmp_path / "missing.fa")
    bad = tmp_path / "bad.fa"
    bad.write_text("ACGT\n")
    with pytest.raises(RuntimeError):
        cdata.FastaStore(bad)


# ---- the kernel: bit-exact against the oracle -------------------------------------------------------------------------
def _rows(B, ld, rng, lens):
    alphabet = np.frombuffer(b"ACGTNacgtnRYX-", dtype=np.uint8)
    probs = np.array([6, 6, 6, 6, 1, 2, 2, 2, 2, 0.5, 0.2, 0.2, 0.2, 0.1]); probs = probs / probs.sum()
    raw = rng.choice(alphabet, size=(B, ld), p=probs)
    return raw, [raw[b, :lens[b]].tobytes().decode() for b in range(B)]


@pytest.mark.parametrize("B,L,ragged", [(1, 1, False), (3, 37, True), (2, 1024, False), (4, 3000, True)])
@pytest.mark.parametrize("mlm", [True, False])
def test_tokenize_mlm_kernel_bit_exact(backend, B, L, ragged, mlm):
    _, dev = backend
    rng = np.random.default_rng(B * 1000 + L)
    lens = rng.integers(0 if L > 1 else 1, L + 1, size=B) if ragged else np.full(B, L)
    raw, seqs = _rows(B, L + 5, rng, lens)
    rc = rng.integers(0, 2, size=B).astype(np.uint8)
    ids, labels = cdata.tokenize_mlm(torch.from_numpy(raw).to(dev), torch.from_numpy(lens.astype(np.int64)).to(dev),
                                     torch.from_numpy(rc).to(dev), L, mlm=mlm, mlm_probability=0.15, seed=0x1234567890AB,
                                     offset=7)
    want_ids, want_labels = do.tokenize_mlm(seqs, L, rc_flags=rc, mlm_probability=0.15, seed=0x1234567890AB, offset=7, mlm=mlm)
    assert np.array_equal(ids.cpu().numpy(), want_ids)
    if mlm:
        assert np.array_equal(labels.cpu().numpy(), want_labels)
    else:
        assert labels is None


def test_tokenize_truncates_on_the_right(backend):
    """pad_max_length < max_length: the reference tokenizer keeps the FIRST pad_max_length tokens (truncation=True,
    truncation_side "right", hg38_dataset.py:190-200), after the reverse complement."""
    _, dev = backend
    rng = np.random.default_rng(77)
    L = 8
    lens = np.array([16, 8, 5, 12])
    raw, seqs = _rows(4, 20, rng, lens)
    raw[0, :16] = np.frombuffer(b"AAAACCCCGGGGTTTT", dtype=np.uint8)
    seqs[0] = "AAAACCCCGGGGTTTT"
    rc = np.array([0, 1, 0, 1], dtype=np.uint8)
    ids, _ = cdata.tokenize_mlm(torch.from_numpy(raw).to(dev), torch.from_numpy(lens.astype(np.int64)).to(dev),
                                torch.from_numpy(rc).to(dev), L, mlm=False)
    want, _ = do.tokenize_mlm(seqs, L, rc_flags=rc, mlm=False)
    assert np.array_equal(ids.cpu().numpy(), want)
    assert ids[0].tolist() == [7, 7, 7, 7, 8, 8, 8, 8]  # AAAACCCC, not GGGGTTTT
    assert ids[3].tolist() == do.tokenize(do.reverse_complement(seqs[3])[:L])


def test_tokenize_mlm_row_streams(backend):
    """row_ids select the random stream of a row: bit-exact vs the oracle, and a sample's corruption does not depend on
    which batch row it occupies."""
    _, dev = backend
    rng = np.random.default_rng(5)
    B, L = 4, 512
    lens = np.full(B, L)
    raw, seqs = _rows(B, L, rng, lens)
    rid = np.array([1000003, 7, 2 ** 31 + 5, 0], dtype=np.int64)
    kw = dict(mlm=True, mlm_probability=0.15, seed=99, offset=3)
    ids, labels = cdata.tokenize_mlm(torch.from_numpy(raw).to(dev), None, None, L, row_ids=torch.from_numpy(rid).to(dev), **kw)
    want_ids, want_labels = do.tokenize_mlm(seqs, L, mlm_probability=0.15, seed=99, offset=3, row_ids=rid)
    assert np.array_equal(ids.cpu().numpy(), want_ids) and np.array_equal(labels.cpu().numpy(), want_labels)
    perm = np.array([2, 0, 3, 1])
    ids_p, labels_p = cdata.tokenize_mlm(torch.from_numpy(raw[perm].copy()).to(dev), None, None, L,
                                         row_ids=torch.from_numpy(rid[perm].copy()).to(dev), **kw)
    assert torch.equal(ids_p, ids[torch.from_numpy(perm).to(dev)]) and torch.equal(labels_p, labels[torch.from_numpy(perm).to(dev)])
    # default stream = the row number
    ids_d, _ = cdata.tokenize_mlm(torch.from_numpy(raw).to(dev), None, None, L, **kw)
    ids_r, _ = cdata.tokenize_mlm(torch.from_numpy(raw).to(dev), None, None, L,
                                  row_ids=torch.arange(B, dtype=torch.int64, device=dev), **kw)
    assert torch.equal(ids_d, ids_r) and not torch.equal(ids_d, ids)


def test_dataset_end_to_end(backend, genome, tmp_path):
    """HG38Dataset over the synthetic genome: reference constructor, (data, target) contract, N -> pad, determinism."""
    _, dev = backend
    path, seqs = genome
    bed = tmp_path / "splits.bed"
    bed.write_text("chrA\t0\t1048576\ttrain\nchrA\t2000000\t3048576\ttrain\nchrB\t17\t1048593\tvalid\nchrC\t0\t1048576\ttrain\n")
    ds = cdata.HG38Dataset("train", bed, path, max_length=1024, mlm=True, mlm_probability=0.15, rc_aug=False, device=dev,
                           seed=3)
    assert len(ds) == 3 * 1024
    idx = [0, 1, 1024 + 5, 2 * 1024 + 1023]
    data, target = ds.batch(idx)
    assert data.shape == (4, 1024) and data.dtype == torch.int64 and data.device.type == dev.type
    ds2 = cdata.HG38Dataset("train", bed, path, max_length=1024, mlm=True, mlm_probability=0.15, device=dev, seed=3)
    d2, t2 = ds2.batch(idx)
    assert torch.equal(data, d2) and torch.equal(target, t2)
    # sample 1024 + 5 = second train row, shift 5: chrA[2000000 + 5 * 1024 : ...]
    plain = seqs["chrA"][2000000 + 5 * 1024: 2000000 + 6 * 1024]
    clean = np.asarray(do.tokenize(plain)); clean[clean == do.N_ID] = do.PAD
    tgt = (target[2] != do.PAD).cpu().numpy()
    assert np.array_equal(target[2].cpu().numpy()[tgt], clean[tgt])
    keep = ~tgt & (clean != do.PAD)  # an N (-> [PAD]) chosen as a target keeps the label [PAD], as in mlm.py
    assert np.array_equal(data[2].cpu().numpy()[keep], clean[keep])
    # the corruption of a sample is keyed on its index, not on its row in the batch
    ds3 = cdata.HG38Dataset("train", bed, path, max_length=1024, mlm=True, mlm_probability=0.15, device=dev, seed=3)
    d3, t3 = ds3.batch(idx[::-1])
    assert torch.equal(d3.flip(0), data) and torch.equal(t3.flip(0), target)
    one_d, one_t = ds2[0]
    assert one_d.shape == (1024,) and one_d.device.type == "cpu"
    clm = cdata.HG38Dataset("valid", bed, path, max_length=2048, mlm=False, add_eos=True, device=dev)
    x, y = clm.batch([0])
    assert x.shape == (1, 2048) and torch.equal(x[0, 1:], y[0, :-1]) and int(y[0, -1]) == 1
    with pytest.raises(ValueError):
        cdata.HG38Dataset("train", bed, path, max_length=2 ** 21)
    with pytest.raises(ValueError):
        cdata.HG38Dataset("train", bed, path, max_length=1024, mlm=True, mlm_probability=0.0)
