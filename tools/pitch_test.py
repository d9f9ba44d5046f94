"""Row pitch of the channel-major activations: the own projection kernels with a power-of-two pitch (T) against T + pad."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops  # noqa: E402
from tools.gemm_stream_bench import timeit  # noqa: E402

dev = torch.device("cuda:0")
D, T, E, R, N = 256, 262144, 512, 16, 16
flush = torch.zeros(128 * 1024 * 1024, device=dev)
bf = lambda *s: torch.empty(*s, dtype=torch.bfloat16, device=dev).normal_()
x2d, w_in, w_x, w_dt, w_out = bf(T, D), bf(2 * E, D), bf(R + 2 * N, E), bf(E, R), bf(D, E)
for pad in (0, 64, 192):
    P = T + pad
    cm = lambda rows: bf(rows, P)[:, :T]
    xz, xc, dbc, delta, y1, y2 = cm(2 * E), cm(E), cm(R + 2 * N), cm(E), cm(E), cm(E)
    r = {"pad": pad}
    r["in_proj_wxT"] = timeit(lambda: ops.proj_wxT(w_in, x2d, out=xz), 10, flush)
    r["x_proj_thin"] = timeit(lambda: ops.proj_wx(w_x, xc, out=dbc), 10, flush)
    r["dt_proj_wx"] = timeit(lambda: ops.proj_wx(w_dt, dbc[:R], out=delta), 10, flush)
    r["out_proj_xTw"] = timeit(lambda: ops.proj_xTw(w_out, y1, y2), 10, flush)
    r["dy_wxT"] = timeit(lambda: ops.proj_wxT(w_out.t().contiguous(), x2d, out=y1), 10, flush)
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items()}))
