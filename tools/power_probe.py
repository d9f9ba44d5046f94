"""GPU box: socket power and shader clock while one kernel family runs back to back (rocm-smi polled from a side thread) -- the
evidence behind "the scans run at the chip's power limit" (DESIGN.md section 3).  Workloads: the two-set scan backward and forward
at the C3 layer shape, a streaming copy, idle.   python tools/power_probe.py"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops  # noqa: E402

E, SB, L, N = 512, 2, 131072, 16
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *sh: torch.randn(*sh, generator=g).to(dev).to(torch.bfloat16)
sp = lambda: torch.nn.functional.softplus(torch.randn(E, SB, L, generator=g) - 3.0).to(dev).to(torch.bfloat16)
A = -(torch.arange(1, N + 1).float().repeat(E, 1)).to(dev)
D, bias = torch.ones(E, device=dev), torch.zeros(E, device=dev)
sets = [(r(E, SB, L), sp(), A, r(N, SB, L), r(N, SB, L), D, bias) for _ in range(2)]
z = r(E, SB, L)
gsets = [tuple(x.clone().requires_grad_(True) for x in st) for st in sets]
zg = z.clone().requires_grad_(True)
o1, o2 = ops.selective_scan_multi(gsets, zg, 1, [(0, 1), (1, 0)], delta_is_dt=True)
g1, g2 = torch.randn_like(o1), torch.randn_like(o2)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
big2 = torch.empty_like(big)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        pw = next((float(v) for k, v in card.items() if "ower" in k and "W" in k and _num(v)), None)
        ck = next((v for k, v in card.items() if "sclk" in k.lower()), None)
        return pw, ck
    except Exception as ex:  # noqa: BLE001
        return None, repr(ex)[:80]


def _num(v):
    try:
        float(v)
        return True
    except (TypeError, ValueError):
        return False


def probe(name, fn, seconds=4.0):
    stop, samples = threading.Event(), []

    def poll():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.15)

    th = threading.Thread(target=poll)
    th.start()
    t0, n = time.time(), 0
    while time.time() - t0 < seconds:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
        n += 8
    stop.set()
    th.join()
    el = time.time() - t0
    pw = [p for p, _ in samples[2:] if p is not None]
    cks = [c for _, c in samples[2:] if c is not None]
    print(json.dumps({"workload": name, "ms_per_call": round(el / max(n, 1) * 1e3, 3), "power_W_avg": round(sum(pw) / len(pw), 1) if pw else None,
                      "power_W_max": max(pw) if pw else None, "sclk_samples": cks[:3] + cks[-2:], "n_samples": len(samples)}))


probe("idle", lambda: time.sleep(0.01), 2.0)
probe("stream copy 1 GiB", lambda: big2.copy_(big))
probe("scan_fwd two-set", lambda: ops.selective_scan_multi(sets, z, 1, [(0, 1), (1, 0)], delta_is_dt=True))
probe("scan_bwd two-set", lambda: torch.autograd.backward([o1, o2], [g1, g2], retain_graph=True))
