"""Phase-level cycle attribution of the two scan kernels (diagnostic -DSC_TIMING build, see scan_common.h).

GPU box:  python -c "from caduceus_amd import _build; _build.build_hip(defines=('SC_TIMING',), out='caduceus_amd/libcaduceus_hip_timing.so')"
          CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_timing.so python tools/phase_timing.py
Prints, per kernel, the shader-clock cycles per pair-step that each wave of workgroup (0,0,0) spent in every phase
(stalls included), C3 layer shape, both parameter sets per launch."""
import ctypes as C
import json
import os
import sys

import torch

os.environ.setdefault("CADUCEUS_AMD_ALLOW_TIMING_BUILD", "1")  # the loader refuses -DSC_TIMING libraries as the product (caduceus_amd/_lib.py)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops, _lib  # noqa: E402

E, SB, L, N = 512, 2, 131072, 16
FWD_PHASES = ["chunk epilogue (gate, store)", "chunk prologue (loads, softplus)", "stage issue + exp + serial scan",
              "wave scan + carry", "output phase", "staging store", "barrier"]
BWD_PHASES = ["loop overhead", "chunk prologue (unpack, gate, softplus)", "stage issue + B/C tile reads",
              "exp + serial scan", "forward wave scan", "true h + local reverse scan", "reverse wave scan + carry",
              "gradient loop + slab writes", "dA sum (+ chunk epilogue)", "staging store", "barrier wait",
              "next-chunk loads + flush"]


def read(lib, name, reset):
    buf = (C.c_ulonglong * (8 * 16))()
    rc = getattr(lib, name)(buf, int(reset))
    assert rc == 0, rc
    return [[buf[w * 16 + p] for p in range(16)] for w in range(8)]


def main():
    dev = torch.device("cuda:0")
    raw = C.CDLL(_lib.LIB_PATH)
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *sh: torch.randn(*sh, generator=g).to(dev).to(torch.bfloat16)
    u, d, z, B, Cm = r(E, SB, L), r(E, SB, L), r(E, SB, L), r(N, SB, L), r(N, SB, L)
    u2, d2, B2, C2 = r(E, SB, L), r(E, SB, L), r(N, SB, L), r(N, SB, L)
    # the production path hands the scans dt itself (softplus in the dt_proj epilogue): the lean backward instantiation
    d, d2 = (torch.nn.functional.softplus(t.float() - 3.0).to(torch.bfloat16) for t in (d, d2))
    A = -(torch.arange(1, N + 1).float().repeat(E, 1)).to(dev)
    D, bias = torch.ones(E, device=dev), (torch.randn(E, generator=g) - 4).to(dev)
    sets = [(u, d, A, B, Cm, D, bias), (u2, d2, A, B2, C2, D, bias)]
    gsets = [tuple(x.clone().requires_grad_(True) for x in st) for st in sets]
    zg = z.clone().requires_grad_(True)
    o1, o2 = ops.selective_scan_multi(gsets, zg, 1, [(0, 1), (1, 0)], delta_is_dt=True)
    g1, g2 = torch.randn_like(o1), torch.randn_like(o2)
    torch.cuda.synchronize()
    read(raw, "cad_debug_timing_fwd", True)
    reps = 3
    for _ in range(reps):
        ops.selective_scan_multi(sets, z, 1, [(0, 1), (1, 0)], delta_is_dt=True)
    torch.cuda.synchronize()
    fwd = read(raw, "cad_debug_timing_fwd", True)
    read(raw, "cad_debug_timing_bwd", True)
    for _ in range(reps):
        torch.autograd.backward([o1, o2], [g1, g2], retain_graph=True)
    torch.cuda.synchronize()
    bwd = read(raw, "cad_debug_timing_bwd", True)
    out = {}
    for name, tab, phases, chunk in (("scan_fwd", fwd, FWD_PHASES, 1024), ("scan_bwd", bwd, BWD_PHASES, 512)):
        steps = reps * (L // chunk) * (N // 2)  # pair-steps per wave
        print(f"== {name}: cycles per pair-step per wave (workgroup 0), {steps} pair-steps")
        tot = [sum(tab[w]) / steps for w in range(8)]
        for p, ph in enumerate(phases):
            row = [tab[w][p] / steps for w in range(8)]
            print(f"  {ph:42s} " + " ".join(f"{x:7.0f}" for x in row) + f"   mean {sum(row) / 8:7.0f} ({100 * sum(row) / sum(tot):4.1f} %)")
        print(f"  {'total':42s} " + " ".join(f"{x:7.0f}" for x in tot))
        out[name] = {ph: sum(tab[w][p] for w in range(8)) / 8 / steps for p, ph in enumerate(phases)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
