"""Kernel-level micro-benchmark of the hot kernels at the BASELINE configs[2] shapes (E=512, N=16, SB=2, L=131072, bf16).
Usage on the GPU box:  python tools/scan_bench.py [--L 131072] [--E 512] [--reps 5]"""
import argparse
import json
import os
import sys
import time

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
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only-scan", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    s = 2 if a.dtype == "bf16" else 4
    E, SB, L, N = a.E, a.SB, a.L, a.N
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *sh: torch.randn(*sh, generator=g).to(dev).to(dt)
    u, delta, z = r(E, SB, L), r(E, SB, L), r(E, SB, L)
    Bm, Cm = r(N, SB, L), r(N, SB, L)
    A = -(torch.arange(1, N + 1).float().repeat(E, 1)).to(dev)
    D, bias = torch.ones(E, device=dev), (torch.randn(E, generator=g) - 4).to(dev)
    w = (0.5 * torch.randn(E, 1, 4, generator=g)).to(dev)
    cb = torch.zeros(E, device=dev)
    res = {}
    T = SB * L
    t = timeit(lambda: ops.selective_scan(u, delta, A, Bm, Cm, D, z, bias, SB // 2 or SB, 0, 1), a.reps)
    res["scan_fwd_ms"] = t
    res["scan_fwd_GBps"] = (4 * E + 2 * N) * s * T / t / 1e6
    ins = [x.clone().requires_grad_(True) for x in (u, delta, A, Bm, Cm, D, z, bias)]
    out = ops.selective_scan(*ins, SB // 2 or SB, 0, 1)
    go = torch.randn_like(out)

    def bwd():
        for x in ins:
            x.grad = None
        out.backward(go, retain_graph=True)
    t = timeit(bwd, a.reps)
    res["scan_bwd_ms"] = t
    res["scan_bwd_GBps"] = (7 * E + 4 * N) * s * T / t / 1e6
    # production path: both parameter sets of a BiMamba layer in one launch
    u2, d2, B2, C2 = r(E, SB, L), r(E, SB, L), r(N, SB, L), r(N, SB, L)
    sets = [(u, delta, A, Bm, Cm, D, bias), (u2, d2, A, B2, C2, D, bias)]
    sp = SB // 2 or SB
    t = timeit(lambda: ops.selective_scan_multi(sets, z, sp, [(0, 1), (1, 0)]), a.reps)
    res["scan_fwd2_ms"] = t
    res["scan_fwd2_GBps"] = 2 * (4 * E + 2 * N) * s * T / t / 1e6
    gsets = [tuple(x.clone().requires_grad_(True) for x in st) for st in sets]
    zg = z.clone().requires_grad_(True)
    o1, o2 = ops.selective_scan_multi(gsets, zg, sp, [(0, 1), (1, 0)])
    g1, g2 = torch.randn_like(o1), torch.randn_like(o2)

    def bwd2():
        torch.autograd.backward([o1, o2], [g1, g2], retain_graph=True)
    _lib.prof_reset(); _lib.prof_enable(True)
    t = timeit(bwd2, a.reps)
    _lib.prof_enable(False)
    pr = _lib.prof_read()
    res["scan_bwd2_kernel_ms"] = pr["scan_bwd"][0] / max(1, pr["scan_bwd"][1])
    res["scan_bwd2_ms"] = t
    res["scan_bwd2_GBps"] = 2 * (7 * E + 4 * N) * s * T / t / 1e6
    if a.only_scan:
        print(json.dumps({k: round(v, 3) for k, v in res.items()}))
        return
    t = timeit(lambda: ops.causal_conv1d(u, w, cb, SB // 2 or SB, 0, 1), a.reps)
    res["conv_fwd_ms"] = t
    res["conv_fwd_GBps"] = 2 * E * s * T / t / 1e6
    D2 = E // 2
    x = torch.randn(2, T // 2, D2, device=dev, dtype=dt)
    rs = torch.randn(2, T // 2, D2, device=dev)
    wn = torch.ones(D2, device=dev)
    t = timeit(lambda: ops.add_norm(x, rs, wn, None, 1e-5, True, True, dt), a.reps)
    res["add_norm_fwd_ms"] = t
    res["add_norm_fwd_GBps"] = (s + 4 + s + 4) * D2 * T / t / 1e6
    # the dense projections at this shape (hipBLASLt through torch.mm)
    Win = torch.randn(2 * E, D2, device=dev, dtype=dt)
    t = timeit(lambda: torch.mm(Win, x.view(T, D2).t()), a.reps)
    res["in_proj_ms"] = t
    res["in_proj_TFLOPs"] = 2 * T * D2 * 2 * E / t / 1e9
    Wx = torch.randn(D2 // 16 + 2 * N, E, device=dev, dtype=dt)
    t = timeit(lambda: torch.mm(Wx, u.view(E, T)), a.reps)
    res["x_proj_ms"] = t
    Wo = torch.randn(D2, E, device=dev, dtype=dt)
    t = timeit(lambda: torch.mm(u.view(E, T).t(), Wo.t()), a.reps)
    res["out_proj_ms"] = t
    print(json.dumps({k: round(v, 3) for k, v in res.items()}))


if __name__ == "__main__":
    main()
