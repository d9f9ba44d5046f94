"""numpy restatement of the hg38 data path (TEST INFRASTRUCTURE ONLY -- never imported by the product).

Follows the reference's per-sample Python:
  * tokenisation            -- /root/reference/caduceus/tokenization_caduceus.py:104-110 (upper-case characters, A C G T N ->
                               7..11, anything else -> [UNK] 6)
  * reverse complement      -- /root/reference/src/dataloaders/utils/rc.py:5-26 (A<->T, C<->G, case kept, others unchanged)
  * N -> [PAD], left padding -- /root/reference/src/dataloaders/datasets/hg38_dataset.py:176-212
  * MLM corruption          -- /root/reference/src/dataloaders/utils/mlm.py:4-32, with the uniforms drawn from
                               Philox4x32-10 (key = seed, counter = (pos_lo, pos_hi, row stream id, offset)) exactly as
                               caduceus_amd/csrc/datapath.hip does, so the comparison is bit-exact.
  * interval arithmetic     -- hg38_dataset.py:41-89 (`hg38_interval`)
Pinned by tests/golden/datapath.npz, generated from the reference's own functions by oracle/gen_golden_data.py.
"""
import numpy as np

SPECIALS = ["[CLS]", "[SEP]", "[BOS]", "[MASK]", "[PAD]", "[RESERVED]", "[UNK]"]
VOCAB = {**{t: i for i, t in enumerate(SPECIALS)}, **{c: 7 + i for i, c in enumerate("ACGTN")}}
PAD, MASK, UNK, N_ID = VOCAB["[PAD]"], VOCAB["[MASK]"], VOCAB["[UNK]"], VOCAB["N"]
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "a": "t", "c": "g", "g": "c", "t": "a"}
MAX_ALLOWED_LENGTH = 2 ** 20


def reverse_complement(seq: str) -> str:
    return "".join(_COMP.get(b, b) for b in reversed(seq))


def tokenize(seq: str):
    return [VOCAB.get(ch.upper(), UNK) if ch.upper() in "ACGTN" else UNK for ch in seq]


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c = [np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF for x in (c0, c1, c2, c3)]
    c = list(np.broadcast_arrays(*c))
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & np.uint64(0xFFFFFFFF)
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & np.uint64(0xFFFFFFFF)
        c = [n0, p1 & np.uint64(0xFFFFFFFF), n2, p0 & np.uint64(0xFFFFFFFF)]
        k0 = (k0 + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
        k1 = (k1 + np.uint64(0xBB67AE85)) & np.uint64(0xFFFFFFFF)
    return c


def mlm_threshold(p: float) -> int:
    if not p > 0.0:
        return 0
    t = p * 4294967296.0
    return 0xFFFFFFFF if t >= 4294967295.0 else int(t)


def tokenize_mlm(seqs, L, rc_flags=None, mlm_probability=0.15, seed=0, offset=0, mlm=True, vocab=12, row_ids=None):
    """seqs: list of str (longer than L: the first L tokens are kept).  Returns (input_ids, labels) int64 (B, L); labels None when mlm is False."""
    B = len(seqs)
    ids = np.full((B, L), PAD, dtype=np.int64)
    valid = np.zeros((B, L), dtype=bool)
    for b, s in enumerate(seqs):
        if rc_flags is not None and rc_flags[b]:
            s = reverse_complement(s)
        s = s[:L]  # tokenizer(truncation=True), default truncation_side="right" (hg38_dataset.py:190-200), after the RC
        t = np.asarray(tokenize(s), dtype=np.int64)
        t[t == N_ID] = PAD
        if len(s):
            ids[b, L - len(s):] = t
            valid[b, L - len(s):] = True
    if not mlm:
        return ids, None
    pos = np.arange(L, dtype=np.uint64)[None, :]
    row = (np.arange(B, dtype=np.uint64) if row_ids is None else
           np.asarray(row_ids, dtype=np.int64).astype(np.uint64) & np.uint64(0xFFFFFFFF))[:, None]
    r0, r1, r2, r3 = philox4x32_10(pos & np.uint64(0xFFFFFFFF), pos >> np.uint64(32), row, np.uint64(offset & 0xFFFFFFFF),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    target = valid & (r0 < np.uint64(mlm_threshold(mlm_probability)))
    labels = np.where(target, ids, PAD)
    replaced = target & (r1 < np.uint64(0xCCCCCCCC))
    rand = target & ~replaced & (r2 < np.uint64(0x80000000))
    words = ((r3 * np.uint64(vocab)) >> np.uint64(32)).astype(np.int64)
    out = np.where(replaced, MASK, np.where(rand, words, ids))
    return out.astype(np.int64), labels.astype(np.int64)


def hg38_interval(start, end, max_length, i_shift, chrom_len):
    if max_length > MAX_ALLOWED_LENGTH:
        raise ValueError("`max_length` is too large!")
    if max_length < MAX_ALLOWED_LENGTH:
        assert MAX_ALLOWED_LENGTH % max_length == 0
        start, end = start + i_shift * max_length, start + (i_shift + 1) * max_length
    if end > chrom_len:
        start, end = start - (end - chrom_len), chrom_len
    if start < 0:
        start, end = 0, end - start
    if end > chrom_len:
        start, end = chrom_len - max_length, chrom_len
    return max(start, 0), end


SYNTH_CHROMS = {"chrA": 3 * 2 ** 20 + 12345, "chrB": 2 ** 20 + 17, "chrC": 2 ** 20}


def write_synthetic_genome(path, chroms=None, seed=11, width=60):
    """The synthetic genome behind tests/golden/datapath.npz (re-generated, not committed: 5 MB of text)."""
    chroms = chroms or SYNTH_CHROMS
    rng = np.random.default_rng(seed)
    seqs = {}
    with open(path, "w") as f:
        for name, n in chroms.items():
            seq = "".join(rng.choice(np.array(list("ACGTNacgt")), size=n, p=[.22, .22, .22, .22, .02, .025, .025, .025, .025]))
            seqs[name] = seq
            f.write(f">{name} synthetic\n")
            for i in range(0, n, width):
                f.write(seq[i:i + width] + "\n")
    return seqs
