"""Micro-benchmark of the token-major scan kernels at the BASELINE configs[2] shape (E=512, N=16, SB=2, L=131072, bf16,
both parameter sets of a layer in one launch sequence).  Usage on the GPU box: python tools/scan_tm_bench.py"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops, _lib  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=131072)
    ap.add_argument("--E", type=int, default=512)
    ap.add_argument("--SB", type=int, default=2)
    ap.add_argument("--N", type=int, default=16)
    ap.add_argument("--R", type=int, default=16)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    s = 2 if a.dtype == "bf16" else 4
    E, SB, L, N, R = a.E, a.SB, a.L, a.N, a.R
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    xz = r(SB, L, 2 * E).to(dt)
    u1, z = xz[..., :E], xz[..., E:]           # column slices of the in_proj output, as in the mixer
    u2 = r(SB, L, E).to(dt)
    d1, d2 = r(SB, L, E).to(dt), r(SB, L, E).to(dt)
    xdbl1, xdbl2 = r(SB, L, R + 2 * N), r(SB, L, R + 2 * N)   # fp32 x_proj outputs
    A = -(torch.arange(1, N + 1).float().repeat(E, 1)).to(dev)
    D, bias = torch.ones(E, device=dev), (torch.randn(E, generator=g) - 4).to(dev)
    sets = [(u1, d1, A, xdbl1[..., R:], D, bias), (u2, d2, A, xdbl2[..., R:], D, bias)]
    sp = SB // 2 or SB
    res = {}
    _lib.prof_reset(); _lib.prof_enable(True)
    t = timeit(lambda: ops.scan_tm_forward(sets, z, sp, [(0, 1), (1, 0)]), a.reps)
    _lib.prof_enable(False)
    T = SB * L
    res["tm_fwd2_ms"] = t
    res["tm_fwd2_GBps"] = 2 * (4 * E + 2 * N) * s * T / t / 1e6
    pr = _lib.prof_read()
    res["tm_fwd2_kernel_ms"] = pr["scan_fwd"][0] / max(1, pr["scan_fwd"][1])
    t = timeit(lambda: ops.scan_tm_forward(sets[:1], z, sp, [(0, 1)]), a.reps)
    res["tm_fwd1_ms"] = t
    print(json.dumps({k: round(v, 3) for k, v in res.items()}))


if __name__ == "__main__":
    main()
