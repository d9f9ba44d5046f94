"""How the bf16 backward's dB / dC error grows with the number of channels E (VERDICT r4, weak spot: contributions are rounded to bf16
before the 8-channel sum and again per partial slot; E / 8 slots deep).  One scan op per E at fixed L, bf16 on the device against the fp32
C oracle (tests-only code, used here as the checker) on the same bf16-rounded inputs; prints a table for DESIGN.md section 4.
GPU box:  python tools/dbdc_error_vs_E.py [--L 4096]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from caduceus_amd import ops  # noqa: E402
from oracle import oracle_ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--L", type=int, default=4096)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, L = 16, a.L
    rows = []
    for E in (64, 128, 256, 512, 1024, 2048):
        g = torch.Generator().manual_seed(E)
        r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
        bf = lambda t: t.to(torch.bfloat16).float()  # noqa: E731
        u, z, Bm, Cm = bf(r(E, 1, L)), bf(r(E, 1, L)), bf(r(N, 1, L)), bf(r(N, 1, L))
        raw = bf(0.5 * r(E, 1, L) - 1.0)
        A = -(0.5 + 15.5 * torch.rand(E, N, generator=g))
        D, bias = r(E), r(E) - 3.0
        w = r(E, 1, L)
        order = (u, raw, A, Bm, Cm, D, z, bias)
        act = (True, True, False, True, True, False, True, False)
        dins = [t.clone().to(dev).to(torch.bfloat16 if is_act else torch.float32).requires_grad_(True) for t, is_act in zip(order, act)]
        out = ops.selective_scan(*dins, 1, 0, 0)
        (out.float() * w.to(dev)).sum().backward()
        # oracle: batch-major (1, E, L) / (1, N, L)
        rins = [t.clone().requires_grad_(True) for t in order]
        ru, rd, rA, rB, rC, rD, rz, rb = rins
        ref = oracle_ops.selective_scan_c(ru.permute(1, 0, 2), rd.permute(1, 0, 2), rA, rB.permute(1, 0, 2), rC.permute(1, 0, 2), rD,
                                          rz.permute(1, 0, 2), rb)
        (ref * w.permute(1, 0, 2)).sum().backward()
        rel = lambda x, y: float((x.float().cpu() - y).norm() / y.norm())  # noqa: E731
        rows.append({"E": E, "slots": (E + 7) // 8, "dB": rel(dins[3].grad, rB.grad), "dC": rel(dins[4].grad, rC.grad),
                     "du": rel(dins[0].grad, ru.grad), "ddelta": rel(dins[1].grad, rd.grad), "out": rel(out.detach(), ref.detach().permute(1, 0, 2))})
        print(json.dumps(rows[-1]), flush=True)
    print("| E | partial slots | dB | dC | du | d(delta) | out |")
    print("|---|---|---|---|---|---|---|")
    for r_ in rows:
        print(f"| {r_['E']} | {r_['slots']} | {r_['dB']:.2e} | {r_['dC']:.2e} | {r_['du']:.2e} | {r_['ddelta']:.2e} | {r_['out']:.2e} |")


if __name__ == "__main__":
    main()
