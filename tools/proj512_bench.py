"""configs[4] (d_model 512, T = 2 x 262144 tokens): the K = 512 projections -- in_proj W (2048, 512) . x^T and d(y) W_out^T (1024, 512) . dout^T --
own kernel (cad_proj_wxT) against hipBLASLt, operands evicted from the memory-side cache between launches."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
T, K = 524288, 512
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev).to(torch.bfloat16)
X = r(T, K)
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, reps=6):
    for _ in range(2):
        fn()
    ms = []
    for _ in range(reps):
        big.fill_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    return sorted(ms)[len(ms) // 2]


only = sys.argv[1] if len(sys.argv) > 1 else ""
for M in (2048, 1024):
    W = r(M, K) * 0.05
    out = torch.empty(M, T, dtype=torch.bfloat16, device=dev)
    res = {"M": M, "K": K, "T": T, "lib": os.environ.get("CADUCEUS_AMD_LIB", "default")}
    by = (T * K + M * K + M * T) * 2
    if only != "lib":
        res["own_ms"] = round(timeit(lambda: ops.proj_wxT(W, X, out=out)), 4)
        res["own_TBps"] = round(by / res["own_ms"] / 1e9, 2)
        res["own_TFLOPs"] = round(2.0 * T * K * M / res["own_ms"] / 1e9, 1)
    if only != "lib":  # round 5: both operands streamed through the tiled kernel, column tiles fastest (ops.gemm_out_t)
        Wt = W.t().contiguous()
        for cf in (True, False):
            key = "stream_ms" if cf else "stream_rowfast_ms"
            res[key] = round(timeit(lambda: ops.proj_xTw_stream(X, Wt, col_fastest=cf)), 4)
        res["stream_TBps"] = round(by / res["stream_ms"] / 1e9, 2)
        res["stream_TFLOPs"] = round(2.0 * T * K * M / res["stream_ms"] / 1e9, 1)
    if only != "own":
        res["hipblaslt_ms"] = round(timeit(lambda: torch.mm(W, X.t(), out=out)), 4)
        res["hipblaslt_TBps"] = round(by / res["hipblaslt_ms"] / 1e9, 2)
    print(json.dumps(res), flush=True)
