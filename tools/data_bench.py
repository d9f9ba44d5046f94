"""Throughput of the hg38 data path at the BASELINE configs[2] sample shape (131072 bases per sample):
host FASTA slicing (GB/s from the page cache), H2D staging, and the tokenize + RC + MLM kernel (G tokens/s).
Usage on the GPU box: python tools/data_bench.py [--batch 8]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import data as cdata  # noqa: E402
from oracle import data_oracle as do  # noqa: E402  (synthetic genome generator only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seqlen", type=int, default=131072)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    d = tempfile.mkdtemp()
    fa, bed = os.path.join(d, "g.fa"), os.path.join(d, "s.bed")
    do.write_synthetic_genome(fa)
    open(bed, "w").write("chrA\t0\t1048576\ttrain\nchrA\t1048576\t2097152\ttrain\nchrC\t0\t1048576\ttrain\n")
    ds = cdata.HG38Dataset("train", bed, fa, max_length=a.seqlen, mlm=True, rc_aug=True, device="cuda:0")
    rng = np.random.default_rng(0)
    idx = [list(rng.integers(0, len(ds), size=a.batch)) for _ in range(a.reps)]
    ds.batch(idx[0]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in idx:
        ds.batch(i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    res = {"end_to_end_tokens_per_s": a.batch * a.seqlen / dt, "ms_per_batch": dt * 1e3, "batch": a.batch, "seqlen": a.seqlen}
    # kernel alone
    bases = torch.randint(65, 85, (a.batch, a.seqlen), dtype=torch.uint8, device="cuda:0")
    rc = torch.ones(a.batch, dtype=torch.uint8, device="cuda:0")
    cdata.tokenize_mlm(bases, None, rc, a.seqlen); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        cdata.tokenize_mlm(bases, None, rc, a.seqlen)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    res["kernel_ms"] = ms
    res["kernel_tokens_per_s"] = a.batch * a.seqlen / ms * 1e3
    res["kernel_GBps"] = a.batch * a.seqlen * 17 / ms / 1e6  # 1 byte in, 2 x 8 bytes out per position
    # host slicing alone
    buf = np.empty(a.seqlen, dtype=np.uint8)
    t0 = time.perf_counter()
    for k in range(200):
        ds.fasta.seqs.fetch_into("chrA", (k * 4099) % 1000000, (k * 4099) % 1000000 + a.seqlen, buf)
    res["fasta_fetch_GBps"] = 200 * a.seqlen / (time.perf_counter() - t0) / 1e9
    print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}))


if __name__ == "__main__":
    main()
