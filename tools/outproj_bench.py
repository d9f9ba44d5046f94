"""out_proj of the tied BiMamba mixer at the configs[2] layer shape: cad_proj_xTw (own kernel, two panels through one set of W fragments)
against the library GEMM on the concatenation [y_f ; y_r] with a doubled weight (what it replaced)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
T, E, D = 262144, 512, 256
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev).to(torch.bfloat16)
ycat, w_out = r(2 * E, T), r(D, E) * 0.06
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)  # evicts the operands from the 256 MB memory-side cache between repetitions


def timeit(fn, reps=8):
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


own = lambda: ops.proj_xTw(w_out, ycat[:E], ycat[E:])
lib = lambda: torch.mm(ycat.t(), torch.cat([w_out, w_out], 1).t())
a, b = own(), lib()
by = (2 * E * T + D * E + T * D) * 2
res = {"lib": os.environ.get("CADUCEUS_AMD_LIB", "default"), "own_ms": round(timeit(own), 4), "hipblaslt_ms": round(timeit(lib), 4),
       "max_abs_diff": float((a.float() - b.float()).abs().max())}
res["own_TBps"] = round(by / res["own_ms"] / 1e9, 2)
res["hipblaslt_TBps"] = round(by / res["hipblaslt_ms"] / 1e9, 2)
print(json.dumps(res))
