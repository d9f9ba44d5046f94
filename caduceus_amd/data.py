"""hg38 data path in front of the model, MI355X-first.

The reference prepares every sample in Python inside DataLoader workers: pyfaidx slice -> per-character reverse complement
-> per-character tokenizer -> torch MLM sampling (/root/reference/src/dataloaders/datasets/hg38_dataset.py:17-227,
src/dataloaders/utils/rc.py, utils/mlm.py).  At the model's token rate that is the bottleneck, so here:
  * the host only slices bytes: `FastaStore` (memory-mapped FASTA, C++ in libcaduceus_hip.so) copies the raw bases of a
    batch into one pinned staging buffer;
  * one kernel (`cad_tokenize_mlm`) does upper-casing, tokenisation, reverse complement, N -> [PAD], left padding and the
    MLM corruption for the whole batch on the GPU.
`FastaInterval` and `HG38Dataset` keep the reference's constructor / call signatures; `HG38Dataset.batch()` is the fast
path that returns device tensors for a list of indices (no worker processes needed).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L

MAX_ALLOWED_LENGTH = 2 ** 20  # hg38_dataset.py:15
_RC_TABLE = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


class FastaStore:
    """Memory-mapped FASTA (replaces pyfaidx.Fasta for this path)."""

    def __init__(self, path):
        self._h = C.c_void_p()
        self._lib = L.get_lib()
        L.check(self._lib.cad_fasta_open(str(path).encode(), C.byref(self._h)), f"cad_fasta_open({path})")
        n = self._lib.cad_fasta_num_seqs(self._h)
        self.names = [self._lib.cad_fasta_seq_name(self._h, i).decode() for i in range(n)]
        self.lengths = {nm: int(self._lib.cad_fasta_seq_len(self._h, i)) for i, nm in enumerate(self.names)}
        self._index = {nm: i for i, nm in enumerate(self.names)}

    def keys(self):
        return list(self.names)

    def fetch_into(self, name: str, start: int, end: int, out: np.ndarray) -> None:
        """Bases [start, end) of sequence `name` into the uint8 array `out` (at least end - start long)."""
        if name not in self._index:
            raise KeyError(name)
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.size >= end - start
        L.check(self._lib.cad_fasta_fetch(self._h, self._index[name], start, end, out.ctypes.data_as(C.c_void_p)),
                "cad_fasta_fetch")

    def fetch(self, name: str, start: int, end: int) -> bytes:
        buf = np.empty(end - start, dtype=np.uint8)
        self.fetch_into(name, start, end, buf)
        return buf.tobytes()

    def close(self):
        if self._h:
            self._lib.cad_fasta_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def hg38_interval(start: int, end: int, max_length: int, i_shift: int, chrom_len: int):
    """The window FastaInterval.__call__ reads (hg38_dataset.py:41-89).  ValueError for max_length > 2**20."""
    s, e = C.c_int64(), C.c_int64()
    st = L.get_lib().cad_hg38_interval(start, end, max_length, i_shift, chrom_len, C.byref(s), C.byref(e))
    if st == 2:
        raise ValueError(f"`max_length` {max_length} (> 2^20) is too large!")
    L.check(st, "cad_hg38_interval")
    return int(s.value), int(e.value)


class FastaInterval:
    """Same call contract as the reference class (hg38_dataset.py:17-89): returns the sequence string."""

    def __init__(self, *, fasta_file, return_seq_indices=False, rc_aug=False, seed: Optional[int] = None):
        fasta_file = Path(fasta_file)
        assert fasta_file.exists(), "Path to fasta file must exist!"
        self.seqs = FastaStore(fasta_file)
        self.return_seq_indices = return_seq_indices
        self.rc_aug = rc_aug
        self.chr_lens = dict(self.seqs.lengths)
        self._rng = np.random.default_rng(seed)

    _compute_interval = staticmethod(lambda start, end, max_length, i_shift:
                                     hg38_interval(start, end, max_length, i_shift, 1 << 62))

    def window(self, chr_name, start, end, max_length, i_shift):
        return hg38_interval(start, end, max_length, i_shift, self.chr_lens[chr_name])

    def coin_flip(self) -> bool:
        return bool(self._rng.random() > 0.5)

    def __call__(self, chr_name, start, end, max_length, i_shift, return_augs=False):
        s, e = self.window(chr_name, start, end, max_length, i_shift)
        seq = self.seqs.fetch(chr_name, s, e)
        if self.rc_aug and self.coin_flip():
            seq = seq.translate(_RC_TABLE)[::-1]
        return seq.decode()


def tokenize_mlm(bases: torch.Tensor, lengths: Optional[torch.Tensor], rc_flags: Optional[torch.Tensor], L_out: int, *,
                 mlm: bool = True, mlm_probability: float = 0.15, seed: int = 0, offset: int = 0, pad_id: int = 4,
                 mask_id: int = 3, unk_id: int = 6, n_id: int = 11, vocab: int = 12, base_ids=(7, 8, 9, 10),
                 row_ids: Optional[torch.Tensor] = None):
    """bases: (B, ld) uint8 on the kernel device (raw ASCII).  Returns (input_ids, labels or None), int64 (B, L_out).
    row_ids (B) int64: the random stream of each row (default: the row number)."""
    lib = L.get_lib()
    B, ld = bases.shape
    ids = torch.empty((B, L_out), dtype=torch.int64, device=bases.device)
    labels = torch.empty_like(ids) if mlm else None
    stream = L.stream_and_check(bases, lengths, rc_flags, ids, labels, row_ids)
    a = L.MlmArgs(L.ptr(bases), L.ptr(rc_flags), L.ptr(lengths), L.ptr(ids), L.ptr(labels), B, L_out, ld, seed, offset,
                  lib.cad_mlm_threshold(float(mlm_probability)), pad_id, mask_id, unk_id, n_id, vocab,
                  (C.c_int * 4)(*base_ids), L.ptr(row_ids))
    L.check(lib.cad_tokenize_mlm(C.byref(a), stream), "cad_tokenize_mlm")
    return ids, labels


class HG38Dataset(torch.utils.data.Dataset):
    """Constructor arguments of the reference dataset (hg38_dataset.py:92-147); `tokenizer` may be a CaduceusTokenizer
    (its ids are used) or None (the default character vocabulary).  `__getitem__` returns the reference's
    `(data, target)` LongTensors; `batch(indices)` returns them stacked on the device in one kernel launch."""

    def __init__(self, split, bed_file, fasta_file, max_length, mlm=False, mlm_probability=0.15, pad_max_length=None,
                 tokenizer=None, tokenizer_name=None, add_eos=False, return_seq_indices=False, rc_aug=False,
                 return_augs=False, device=None, seed: int = 2222):
        self.mlm = mlm
        self.mlm_probability = mlm_probability
        if self.mlm and self.mlm_probability <= 0.0:
            raise ValueError(f"`mlm_probability` has to be > 0.0, got {self.mlm_probability}.")
        self.max_length = max_length
        self.pad_max_length = pad_max_length if pad_max_length is not None else max_length
        self.tokenizer_name = tokenizer_name
        self.tokenizer = tokenizer
        self.return_augs = return_augs
        self.add_eos = add_eos
        if max_length <= MAX_ALLOWED_LENGTH:
            assert MAX_ALLOWED_LENGTH % max_length == 0, "`max_length` must be a power of 2!"
            self.shifts = MAX_ALLOWED_LENGTH // max_length
        else:
            raise ValueError(f"`max_length` {max_length} (> 2^20) is too large!")
        if tokenizer_name not in (None, "char"):
            raise NotImplementedError("only the character tokenizer is part of the Caduceus path")
        bed_path = Path(bed_file)
        assert bed_path.exists(), "Path to .bed file must exist!"
        rows = []
        for line in open(bed_path):
            f = line.rstrip("\n").split("\t")
            if len(f) >= 4 and f[3] == split:
                rows.append((f[0], int(f[1])))
        self.rows = rows  # (chr_name, start); every interval is [start, start + 2**20)
        self.fasta = FastaInterval(fasta_file=fasta_file, return_seq_indices=return_seq_indices, rc_aug=rc_aug, seed=seed)
        self.seed = seed
        self._calls = 0
        self.device = torch.device(device) if device is not None else \
            torch.device("cuda", torch.cuda.current_device()) if (L.is_device_build() and torch.cuda.is_available()) \
            else torch.device("cpu")
        v = tokenizer.get_vocab() if tokenizer is not None else None
        g = (lambda t, d: v[t] if v is not None else d)
        self._ids = dict(pad_id=g("[PAD]", 4), mask_id=g("[MASK]", 3), unk_id=g("[UNK]", 6), n_id=g("N", 11),
                         vocab=len(tokenizer) if tokenizer is not None else 12,
                         base_ids=(g("A", 7), g("C", 8), g("G", 9), g("T", 10)))
        self._sep = g("[SEP]", 1)

    def __len__(self):
        return len(self.rows) * self.shifts

    def batch(self, indices: Sequence[int]):
        """(data, target) for the samples `indices`, each (len(indices), L) int64 on self.device."""
        B, ml, Lp = len(indices), self.max_length, self.pad_max_length
        stage = torch.empty((B, ml), dtype=torch.uint8, pin_memory=self.device.type == "cuda")
        lens = torch.empty((B,), dtype=torch.int64)
        rc = torch.zeros((B,), dtype=torch.uint8)
        buf = stage.numpy()
        for j, idx in enumerate(indices):
            chr_name, start = self.rows[idx // self.shifts]
            s, e = self.fasta.window(chr_name, start, start + MAX_ALLOWED_LENGTH, ml, idx % self.shifts)
            self.fasta.seqs.fetch_into(chr_name, s, e, buf[j])
            lens[j] = e - s
            rc[j] = 1 if (self.fasta.rc_aug and self.fasta.coin_flip()) else 0
        dev = self.device
        bases = stage.to(dev, non_blocking=True)
        self._calls += 1  # the mask of a sample is keyed on (its index, how many batches this object has served)
        ids, labels = tokenize_mlm(bases, lens.to(dev), rc.to(dev), Lp, mlm=self.mlm,
                                   mlm_probability=self.mlm_probability, seed=self.seed, offset=self._calls,
                                   row_ids=torch.as_tensor(list(indices), dtype=torch.int64).to(dev), **self._ids)
        if self.mlm:
            return ids, labels  # add_eos is appended and stripped again by the reference (mlm.py:10): no net effect
        if self.add_eos:
            ids = torch.cat([ids, torch.full((B, 1), self._sep, dtype=ids.dtype, device=dev)], 1)
        return ids[:, :-1].clone(), ids[:, 1:].clone()  # next-token targets

    def __getitem__(self, idx):
        data, target = self.batch([idx])
        return data[0].cpu(), target[0].cpu()
