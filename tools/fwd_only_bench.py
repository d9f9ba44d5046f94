"""Times only the two-set forward scan launch (C3 layer shape) -- used for same-box A/B runs of variant libraries."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops  # noqa: E402
E, SB, L, N = 512, 2, 131072, 16
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *sh: torch.randn(*sh, generator=g).to(dev).to(torch.bfloat16)
u, d, z, B, C = r(E, SB, L), r(E, SB, L), r(E, SB, L), r(N, SB, L), r(N, SB, L)
u2, d2, B2, C2 = r(E, SB, L), r(E, SB, L), r(N, SB, L), r(N, SB, L)
A = -(torch.arange(1, N + 1).float().repeat(E, 1)).to(dev)
D, bias = torch.ones(E, device=dev), (torch.randn(E, generator=g) - 4).to(dev)
sets = [(u, d, A, B, C, D, bias), (u2, d2, A, B2, C2, D, bias)]
fn = lambda: ops.selective_scan_multi(sets, z, 1, [(0, 1), (1, 0)])
for _ in range(3):
    fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    fn()
b.record()
torch.cuda.synchronize()
print(json.dumps({"lib": os.environ.get("CADUCEUS_AMD_LIB", "default"), "scan_fwd2_ms": round(a.elapsed_time(b) / 10, 4)}))
