"""GPU debug: where does dz of the lean backward differ from the oracle?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from caduceus_amd import ops
from oracle import oracle_model as om
from test_kernels import _scan_inputs, _rows_oracle, leaf

dev = torch.device("cuda:0")
captured = []
real = ops.gate_fix_buffers
def cap(lib, u, N):
    r = real(lib, u, N)
    captured.append(r)
    return r
ops.gate_fix_buffers = cap

def run(case, zero_gates=True):
    E, SB, L, split, rl, rh = case
    N, dtype = 16, torch.bfloat16
    order = ("u", "delta", "A", "B", "C", "D", "bias")
    act = {"u", "delta", "B", "C"}
    t = _scan_inputs(E, SB, L, N, 31, dev, dtype)
    raw = (0.5 * t["delta"] - 1.0)
    t["delta"] = torch.nn.functional.softplus(raw + t["bias"][:, None, None]).to(dtype).float()
    z = t["z"].clone()
    if zero_gates:
        z[1, 0, 3] = 0.0
        z[E - 1, SB - 1, L - 1] = 0.0
    zd = leaf(z, dev, dtype)
    ds = tuple(leaf(t[k], dev, dtype if k in act else torch.float32) for k in order)
    captured.clear()
    out = ops.selective_scan_multi([ds], zd, split, [(rl, rh)], delta_is_dt=True)[0]
    (out.float() * t["w"].to(dev)).sum().backward()
    zr = leaf(z, "cpu")
    tt = dict(t)
    dtv = t["delta"].double()
    tt["delta"] = torch.where(dtv > 0, dtv + torch.log(-torch.expm1(-dtv)), torch.full_like(dtv, -200.0)).float()
    tt["bias"] = torch.zeros_like(t["bias"])
    rs = tuple(leaf(tt[k], "cpu") for k in order)
    u, d, A, B, C, D, b = rs
    ref = _rows_oracle(lambda u_, d_, B_, C_, z_: om.selective_scan(u_, d_, A, B_, C_, D, z_, b), [u, d, B, C, zr], split, rl, rh)
    (ref * t["w"]).sum().backward()
    got, want = zd.grad.float().cpu(), zr.grad
    bad = (got - want).abs() > 0.05 * max(1.0, float(want.abs().max())) * 0.2 + 0.03 * want.abs()
    print("case", case, "zero_gates", zero_gates, "bad", int(bad.sum()), "fix count", [int(c[1].item()) for c in captured])
    if captured and int(captured[0][1].item()) > 0:
        n = int(captured[0][1].item())
        ent = captured[0][0][:n].cpu().tolist()
        print("  fix entries (e, sb, chunk):", [(x & 0xFFFFF, (x >> 20) & 0xFFFFF, x >> 40) for x in ent])
    idx = bad.nonzero()
    rows = {}
    for e, sb, l in idx.tolist():
        rows.setdefault((e, sb), []).append(l)
    for k, v in rows.items():
        print("  row", k, "n", len(v), "positions", v[:12], "...", v[-6:])
    for e, sb, l in idx.tolist()[:8]:
        print("   ", (e, sb, l), "got", float(got[e, sb, l]), "want", float(want[e, sb, l]), "z", float(z[e, sb, l]), "out", float(out[e, sb, l]))

for c in [(9, 3, 1104, 2, 1, 0), (8, 2, 512, 1, 0, 1), (16, 1, 8, 1, 1, 1), (9, 3, 1104, 2, 0, 1)]:
    run(c)
run((9, 3, 1104, 2, 1, 0), zero_gates=False)
