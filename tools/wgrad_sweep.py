"""K-chunk sweep of the weight-gradient products (reduction over T = 262144 tokens, caduceus_amd/mixer.py::_wgrad_*): time of the
strided-batch GEMM + fp32 sum for n = 8 .. 512 chunks, per product shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device("cuda:0")
T, E, D, R, N = 262144, 512, 256, 16, 16
r = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def cm_cm(a, b, n):
    M = a.shape[0]
    Kc = T // n
    return torch.sum(torch.bmm(a.view(M, n, Kc).permute(1, 0, 2), b.view(-1, n, Kc).permute(1, 2, 0)), dim=0, dtype=torch.float32)


def cm_tm(a, b, n):
    M = a.shape[0]
    Kc = T // n
    return torch.sum(torch.bmm(a.view(M, n, Kc).permute(1, 0, 2), b.view(n, Kc, -1)), dim=0, dtype=torch.float32)


dxz, x2d, ycat, dout, ddbc, xc, ddelta, dtlr = r(2 * E, T), r(T, D), r(2 * E, T), r(T, D), r(R + 2 * N, T), r(E, T), r(E, T), r(R, T)
cases = {
    "dW_in  dxz(1024xT) . x2d(Tx256)": (cm_tm, dxz, x2d),
    "dW_out ycat(1024xT) . dout(Tx256)": (cm_tm, ycat, dout),
    "dW_x   ddbc(48xT) . xc(512xT)^T": (cm_cm, ddbc, xc),
    "dW_dt  ddelta(512xT) . dt_lr(16xT)^T": (cm_cm, ddelta, dtlr),
}
res = {}
for name, (fn, a, b) in cases.items():
    res[name] = {}
    for n in (8, 16, 32, 64, 128, 256, 512):
        res[name][n] = round(timeit(lambda: fn(a, b, n)), 4)
    print(name, res[name], flush=True)
print(json.dumps(res))
