"""Occupancy experiment on the real scan code (GPU box): the carry-only backward pass (92 VGPRs, 68 KB of LDS -> two workgroups =
four waves per SIMD fit a CU) and the full backward on the SAME total work cut into k segments per row, i.e. 256 * k workgroups:
k = 1 is one workgroup per CU (two waves per SIMD), k = 2 lets two carry-only workgroups share every CU.  Prints ms per launch.
    python tools/carry_bench.py            (optionally CADUCEUS_AMD_LIB=...)"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import _lib as L  # noqa: E402

E, SB, Lq, N = 512, 2, 131072, 16
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *sh: torch.randn(*sh, generator=g).to(dev).to(torch.bfloat16)
lib = L.get_lib()
sets = []
for i in range(2):
    u, z, B, Cm, dout = r(E, SB, Lq), r(E, SB, Lq), r(N, SB, Lq), r(N, SB, Lq), r(E, SB, Lq)
    dt = torch.nn.functional.softplus(torch.randn(E, SB, Lq, generator=g) - 3.0).to(dev).to(torch.bfloat16)
    A = -(torch.arange(1, N + 1).float().repeat(E, 1)).to(dev)
    D, bias = torch.ones(E, device=dev), torch.zeros(E, device=dev)
    sets.append((u, dt, A, B, Cm, D, z, bias, dout))
res = {"lib": os.environ.get("CADUCEUS_AMD_LIB", "default")}
for k in (1, 2, 4):
    args = (L.ScanBwdArgs * 2)()
    keep = []
    for i, (u, dt, A, B, Cm, D, z, bias, dout) in enumerate(sets):
        dh0 = torch.empty((E, SB * k, N), dtype=torch.float32, device=dev)
        a = L.ScanBwdArgs()
        a.u, a.delta, a.A, a.Bm, a.Cm, a.D, a.z, a.delta_bias, a.dout = (L.ptr(u), L.ptr(dt), L.ptr(A), L.ptr(B), L.ptr(Cm),
                                                                         L.ptr(D), L.ptr(z), L.ptr(bias), L.ptr(dout))
        a.SB, a.L, a.split, a.E, a.N = SB * k, Lq // k, k, E, N
        a.rev_lo, a.rev_hi = (0, 1) if i == 0 else (1, 0)
        a.dtype, a.delta_is_dt, a.carry_only, a.dh0 = L.CAD_BF16, 1, 1, L.ptr(dh0)
        args[i] = a
        keep.append(dh0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        L.check(lib.cad_scan_bwd_multi(args, 2, stream), "carry")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        L.check(lib.cad_scan_bwd_multi(args, 2, stream), "carry")
    e1.record()
    torch.cuda.synchronize()
    res[f"carry_only_k{k}_ms"] = round(e0.elapsed_time(e1) / 6, 4)
print(json.dumps(res))
