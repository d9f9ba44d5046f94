"""Per-kernel-family times (the library's HIP-event profiler) of whole training steps of a short Caduceus-PS stack at the BASELINE
configs[2] layer shape -- the same-box A/B unit for kernels outside the mixer (add + norm, embedding, LM head):

    CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_<variant>.so python tools/step_families.py [--n-layer 4] [--seqlen 131072]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import CaduceusForMaskedLM, _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d-model", type=int, default=256)
    ap.add_argument("--n-layer", type=int, default=4)
    ap.add_argument("--seqlen", type=int, default=131072)
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    import bench
    cfg = bench.make_config(a.d_model, a.n_layer)
    model = CaduceusForMaskedLM(cfg).to(dev).train()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(7, 11, (1, a.seqlen), generator=g).to(dev)
    labels = ids.clone()
    labels[torch.rand(1, a.seqlen, generator=g).to(dev) > 0.15] = 4

    def step():
        model.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(ids, labels=labels)
        out.loss.backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    _lib.prof_reset()
    _lib.prof_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        step()
    e1.record()
    torch.cuda.synchronize()
    pr = _lib.prof_read()
    _lib.prof_enable(False)
    res = {"lib": os.environ.get("CADUCEUS_AMD_LIB", "default"), "step_ms": round(e0.elapsed_time(e1) / a.reps, 3)}
    for k, (ms, n) in pr.items():
        if n:
            res[k + "_ms"] = round(ms / n, 4)
            res[k + "_n"] = n // a.reps
    print(json.dumps(res))


if __name__ == "__main__":
    main()
