"""MFMA projection kernels (csrc/gemm.hip) against hipBLASLt (torch.mm) at the configs[2] layer shapes: time, achieved
GB/s of the algorithmic traffic, TFLOP/s, and the max deviation between the two."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
T, E, D = 262144, 512, 256
r = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


x2d, w_in, w_outT, dout = r(T, D), r(2 * E, D) * 0.06, r(E, D) * 0.06, r(T, D)
res = {}
for name, W, X in (("in_proj W(1024x256) . x^T", w_in, x2d), ("dy W_out^T(512x256) . dout^T", w_outT, dout)):
    M, K = W.shape
    ours = ops.proj_wxT(W, X)
    ref = torch.mm(W, X.t())
    err = float((ours.float() - ref.float()).abs().max())
    t_ours = timeit(lambda: ops.proj_wxT(W, X))
    t_lib = timeit(lambda: torch.mm(W, X.t()))
    by = (T * K + M * K + M * T) * 2
    fl = 2.0 * T * K * M
    res[name] = {"ours_ms": round(t_ours, 4), "hipblaslt_ms": round(t_lib, 4), "ours_GBps": round(by / t_ours / 1e6, 1),
                 "ours_TFLOPs": round(fl / t_ours / 1e9, 1), "hipblaslt_GBps": round(by / t_lib / 1e6, 1),
                 "max_abs_diff_vs_hipblaslt": err}
    print(name, res[name])
print(json.dumps(res))
