"""cad_gemm_stream against the library products it replaces, at the mixer-backward shapes (GPU box).
usage: python tools/gemm_stream_bench.py [--d-model 256] [--T 262144] [--reps 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import mixer, ops  # noqa: E402


def timeit(fn, reps, flush):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        flush.add_(1.0)  # 512 MB through the caches: the operands come from HBM, as in the step
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
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    D, T = a.d_model, a.T
    E2 = 4 * D  # 2 d_inner
    g = torch.Generator().manual_seed(0)
    r = lambda *s: (0.5 * torch.randn(*s, generator=g)).to(torch.bfloat16).to(dev)
    dxz, x2d, w_in = r(E2, T), r(T, D), r(E2, D)
    flush = torch.zeros(128 * 1024 * 1024, device=dev)
    res = {"d_model": D, "T": T}
    mixer._OWN_GEMM = False  # mixer._wgrad_cm_tm = the library path (K-split bmm + fp32 sum)
    # weight gradient (dW_in; dW_out has the same shape)
    ref = mixer._wgrad_cm_tm(dxz, x2d)
    own = ops.wgrad_cm_tm(dxz, x2d)
    res["wgrad_rel_err_vs_library"] = float((own - ref).norm() / ref.norm())
    f64 = (dxz[:, :4096].double() @ x2d[:4096].double())
    res["wgrad_rel_err_fp64_slice"] = float((ops.wgrad_cm_tm(dxz[:, :4096], x2d[:4096]).double() - f64).norm() / f64.norm())
    res["wgrad_library_ms"] = timeit(lambda: mixer._wgrad_cm_tm(dxz, x2d), a.reps, flush)
    res["wgrad_own_ms"] = timeit(lambda: ops.wgrad_cm_tm(dxz, x2d), a.reps, flush)
    # token-major input gradient
    wt = w_in.t().contiguous()
    ref = torch.mm(dxz.t(), w_in)
    own = ops.proj_xTw_stream(wt, dxz)
    res["dx_bit_identical"] = bool(torch.equal(ref, own))
    res["dx_rel_err_vs_library"] = float((own.float() - ref.float()).norm() / ref.float().norm())
    res["dx_library_ms"] = timeit(lambda: torch.mm(dxz.t(), w_in), a.reps, flush)
    res["dx_own_ms"] = timeit(lambda: ops.proj_xTw_stream(w_in.t().contiguous(), dxz), a.reps, flush)
    byt = (E2 * T + T * D) * 2
    for k in ("wgrad_library_ms", "wgrad_own_ms", "dx_library_ms", "dx_own_ms"):
        res[k.replace("_ms", "_TBps")] = round(byt / res[k] / 1e9, 2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
