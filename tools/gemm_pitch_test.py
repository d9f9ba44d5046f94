"""Does the row pitch of the channel-major operand matter to cad_gemm_stream (power-of-two pitch = same channel / bank for every row)?"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops  # noqa: E402
from tools.gemm_stream_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
D, T = 256, 262144
E2 = 4 * D
flush = torch.zeros(128 * 1024 * 1024, device=dev)
g = torch.Generator().manual_seed(0)
x2d = (0.5 * torch.randn(T, D, generator=g)).to(torch.bfloat16).to(dev)
wt = (0.5 * torch.randn(D, E2, generator=g)).to(torch.bfloat16).to(dev)
for pad in (0, 32, 64, 128, 256, 1024, 2048 + 64, 8192 + 128):
    buf = torch.empty(E2, T + pad, dtype=torch.bfloat16, device=dev).normal_()
    dxz = buf[:, :T]
    r = {"pad_elems": pad,
         "wgrad_own_ms": round(timeit(lambda: ops.wgrad_cm_tm(dxz, x2d), 10, flush), 4),
         "dx_own_ms": round(timeit(lambda: ops.proj_xTw_stream(wt, dxz), 10, flush), 4),
         "dx_lib_ms": round(timeit(lambda: torch.mm(dxz.t(), wt.t()), 10, flush), 4)}
    print(json.dumps(r))
