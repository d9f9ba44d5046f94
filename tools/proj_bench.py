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

# thin-K channel-major products (transposing LDS reads): dt_proj and the x_proj input gradient with its addend
R, N = 16, 16
w_dt, dt_lr = r(E, R) * 0.2, r(R, T)
w_xT, ddbc, du = (r(R + 2 * N, E) * 0.06).t().contiguous(), r(R + 2 * N, T), r(E, T)
for name, W, X, acc in (("dt_proj W(512x16) . dt_lr(16xT)", w_dt, dt_lr, None),
                        ("du += W_x^T(512x48) . ddbc(48xT)", w_xT, ddbc, du)):
    M, K = W.shape
    if not ops.proj_wx_supported(W, K, T):
        continue
    ours = ops.proj_wx(W, X, acc=acc)
    ref = torch.mm(W, X) if acc is None else (torch.mm(W, X) + acc)
    err = float((ours.float() - ref.float()).abs().max())
    if acc is None:
        t_ours = timeit(lambda: ops.proj_wx(W, X))
        t_lib = timeit(lambda: torch.mm(W, X))
    else:
        buf = acc.clone()
        t_ours = timeit(lambda: ops.proj_wx(W, X, out=buf, acc=buf))
        buf2 = acc.clone()
        t_lib = timeit(lambda: buf2.addmm_(W, X))
    by = (T * K + M * K + M * T * (2 if acc is not None else 1)) * 2
    res[name] = {"ours_ms": round(t_ours, 4), "hipblaslt_ms": round(t_lib, 4), "ours_GBps": round(by / t_ours / 1e6, 1),
                 "hipblaslt_GBps": round(by / t_lib / 1e6, 1), "max_abs_diff_vs_hipblaslt": err}
    print(name, res[name])
print(json.dumps(res))

# thin M / deep K (x_proj forward, d(dt_lr) of the backward): the (K, T) activation is the stream
xc, ddelta = r(E, T), r(E, T)
w_x, w_dtT = r(R + 2 * N, E) * 0.06, (r(E, R) * 0.2).t().contiguous()
for name, W, X in (("x_proj W_x(48x512) . xc(512xT)", w_x, xc), ("d(dt_lr) W_dt^T(16x512) . ddelta(512xT)", w_dtT, ddelta)):
    M, K = W.shape
    if not ops.proj_wx_supported(X, K, T, M=M):
        continue
    ours = ops.proj_wx(W, X)
    ref = torch.mm(W, X)
    err = float((ours.float() - ref.float()).abs().max())
    t_ours = timeit(lambda: ops.proj_wx(W, X))
    t_lib = timeit(lambda: torch.mm(W, X))
    by = (T * K + M * K + M * T) * 2
    res[name] = {"ours_ms": round(t_ours, 4), "hipblaslt_ms": round(t_lib, 4), "ours_GBps": round(by / t_ours / 1e6, 1),
                 "hipblaslt_GBps": round(by / t_lib / 1e6, 1), "max_abs_diff_vs_hipblaslt": err}
    print(name, res[name])
print(json.dumps(res))
