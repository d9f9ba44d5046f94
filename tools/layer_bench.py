"""One BiMamba mixer layer of the production path (tied in/out projections, "add", RCPS strands as rows) at the BASELINE
configs[2] layer shape, forward + backward, exactly as the training step runs it (mixer.BiMambaMixerFn: dt from the dt_proj
epilogue, shared gate, partial-slot reduction, own MFMA projections).  Prints the per-kernel-family times of the library's
HIP-event profiler and the wall time of the layer -- the same-box A/B unit for variant libraries:

    CADUCEUS_AMD_LIB=caduceus_amd/libcaduceus_hip_<variant>.so python tools/layer_bench.py [--d-model 256] [--seqlen 131072]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import _lib, mixer  # noqa: E402
from caduceus_amd.mamba import Mamba  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d-model", type=int, default=256)
    ap.add_argument("--seqlen", type=int, default=131072)
    ap.add_argument("--strands", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--ab", default=None, help="mixer module switch (e.g. _OWN_DWX) to toggle IN THIS PROCESS: blocks of --reps "
                    "layers alternate on / off for --rounds rounds, so both arms see the same clocks and power state")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--tag", default=None, help="label of this run in the output (tools/ab_layer.sh: the variant name)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    mf, mr = Mamba(a.d_model, device=dev), Mamba(a.d_model, device=dev)
    mr.in_proj.weight = mf.in_proj.weight    # weight tying of BiMambaWrapper (modeling_caduceus.py:114-118)
    mr.out_proj.weight = mf.out_proj.weight
    S, B, L, D = a.strands, a.batch, a.seqlen, a.d_model
    hn = torch.randn(S, B, L, D, device=dev).to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(S, B, L, D, device=dev).to(torch.bfloat16)
    split = B if S == 2 else S * B

    def step():
        for p in list(mf.parameters()) + list(mr.parameters()):
            p.grad = None
        hn.grad = None
        mixer.prepare_step_cache([(mf, mr)], torch.bfloat16)
        out = mixer.bimamba_mixer(hn, mf, mr, split)
        out.backward(g)

    from caduceus_amd import ops
    ops.FOLD_GIVE_UPS = []  # diagnostics: slices the concurrent dB / dC fold left to its cleanup pass (0 when co-scheduled as designed)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    print("fold give-up records in the two warm-up layers:", [int(x) for x in ops.FOLD_GIVE_UPS], file=sys.stderr)
    ops.FOLD_GIVE_UPS = None
    if a.ab:
        ms = {True: [], False: []}
        for _ in range(a.rounds):
            for on in (True, False):
                setattr(mixer, a.ab, on)
                step()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                ms[on].append(round(e0.elapsed_time(e1) / a.reps, 3))
        print(json.dumps({"ab": a.ab, "on_ms": ms[True], "off_ms": ms[False],
                          "on_mean": round(sum(ms[True]) / a.rounds, 3), "off_mean": round(sum(ms[False]) / a.rounds, 3)}))
        return
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
    res = {"lib": a.tag or os.environ.get("CADUCEUS_AMD_LIB", "default"), "layer_ms": round(e0.elapsed_time(e1) / a.reps, 3)}
    for k, (ms, n) in pr.items():
        if n:
            res[k + "_ms"] = round(ms / n, 4)
            res[k + "_n"] = n // a.reps
    print(json.dumps(res))


if __name__ == "__main__":
    main()
