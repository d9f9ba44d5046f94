"""Times the two-set scan launches of ONE training step of a layer (C3 layer shape): the forward WITH its saved states
and the backward -- kernel time from the library's event profiler; used for same-box A/B runs of variant libraries."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops, _lib  # noqa: E402
E, SB, L, N = 512, 2, 131072, 16
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *sh: torch.randn(*sh, generator=g).to(dev).to(torch.bfloat16)
u, d, z, B, C = r(E, SB, L), r(E, SB, L), r(E, SB, L), r(N, SB, L), r(N, SB, L)
u2, d2, B2, C2 = r(E, SB, L), r(E, SB, L), r(N, SB, L), r(N, SB, L)
A = -(torch.arange(1, N + 1).float().repeat(E, 1)).to(dev)
D, bias = torch.ones(E, device=dev), (torch.randn(E, generator=g) - 4).to(dev)
sets = [(u, d, A, B, C, D, bias), (u2, d2, A, B2, C2, D, bias)]
gsets = [tuple(x.clone().requires_grad_(True) for x in st) for st in sets]
zg = z.clone().requires_grad_(True)
g1, g2 = torch.randn_like(u), torch.randn_like(u)


def step():
    o1, o2 = ops.selective_scan_multi(gsets, zg, 1, [(0, 1), (1, 0)])
    torch.autograd.backward([o1, o2], [g1, g2])


for _ in range(2):
    step()
torch.cuda.synchronize()
_lib.prof_reset(); _lib.prof_enable(True)
for _ in range(6):
    step()
torch.cuda.synchronize()
p = _lib.prof_read()
_lib.prof_enable(False)
f, b = p[_lib.PROF_KINDS[0]], p[_lib.PROF_KINDS[1]]
print(json.dumps({"lib": os.environ.get("CADUCEUS_AMD_LIB", "default"), "train_fwd2_ms": round(f[0] / max(f[1], 1), 4),
                  "train_bwd2_ms": round(b[0] / max(b[1], 1), 4)}))
