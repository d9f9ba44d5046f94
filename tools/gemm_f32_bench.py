"""cad_gemm_f32 (fp32 matrix core, csrc/gemm_f32.hip) against torch.mm (hipBLASLt) at the fp32 path's projection shapes (GPU box).
usage: python tools/gemm_f32_bench.py [--d-model 256] [--T 262144] [--reps 10]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d-model", type=int, default=256)
    ap.add_argument("--T", type=int, default=262144)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    D, T = a.d_model, a.T
    E, R, N = 2 * D, (D + 15) // 16, 16
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (0.5 * torch.randn(*s, generator=g)).to(dev)
    x2d, w_in = r(T, D), r(2 * E, D)
    xc, w_x = r(E, T), r(R + 2 * N, E)
    dxz = r(2 * E, T)
    shapes = {
        "in_proj   W (2E x D) . X^T (D x T)": (lambda f: f(w_in, x2d.t()), 2.0 * 2 * E * D * T),
        "x_proj    W (R+2N x E) . xc (E x T)": (lambda f: f(w_x, xc), 2.0 * (R + 2 * N) * E * T),
        "d(x2d)    dxz^T (T x 2E) . W (2E x D)": (lambda f: f(dxz.t(), w_in), 2.0 * 2 * E * D * T),
        "dW_in     dxz (2E x T) . X (T x D)  [K = T]": (lambda f: f(dxz, x2d), 2.0 * 2 * E * D * T),
    }
    for name, (call, flop) in shapes.items():
        ref = call(torch.mm)
        own = call(ops.mm_f32)
        err = float((own - ref).abs().max() / ref.abs().max())
        t_lib, t_own = timeit(lambda: call(torch.mm), a.reps), timeit(lambda: call(ops.mm_f32), a.reps)
        print(json.dumps({"product": name, "library_ms": round(t_lib, 3), "own_ms": round(t_own, 3), "own_TFLOPs": round(flop / t_own / 1e9, 1),
                          "library_TFLOPs": round(flop / t_lib / 1e9, 1), "max_rel_diff": err}))


if __name__ == "__main__":
    main()
