"""The four library GEMMs that are left on the training path, timed with hipBLASLt's default choice and
with PyTorch's TunableOp (run-time search over the rocBLAS / hipBLASLt solutions; results in a CSV that can be shipped and replayed with
tuning off).  GPU box:  python tools/tunable_gemm.py [out.csv]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import mixer  # noqa: E402

D, E, T = 256, 512, 262144
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev).to(torch.bfloat16)  # noqa: E731
ycat, dxz, dout2d, x2d = r(2 * E, T), r(2 * E, T), r(T, D), r(T, D)
w_out, w_in = r(D, E), r(2 * E, D)
wcat = torch.cat([w_out, w_out], 1)

ops = {
    "out_proj fwd  (T,2E)x(2E,D)": lambda: torch.mm(ycat.t(), wcat.t()),
    "d(x2d)        (T,2E)x(2E,D)": lambda: torch.mm(dxz.t(), w_in),
    "dW_out        (2E,T)x(T,D) K-split": lambda: mixer._wgrad_cm_tm(ycat, dout2d),
    "dW_in         (2E,T)x(T,D) K-split": lambda: mixer._wgrad_cm_tm(dxz, x2d),
}


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


base = {k: bench(f) for k, f in ops.items()}
ref = {k: f().float() for k, f in ops.items()}
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/tunableop_results.csv"
import torch.cuda.tunable as tun  # noqa: E402
tun.set_filename(out)
tun.enable(True)
tun.tuning_enable(True)
tun.set_max_tuning_duration(200)   # ms per solution
tun.set_max_tuning_iterations(20)
t0 = time.time()
for k, f in ops.items():
    f()
torch.cuda.synchronize()
print(f"tuning took {time.time() - t0:.1f} s")
tun.tuning_enable(False)
tuned = {k: bench(f) for k, f in ops.items()}
for k in ops:
    err = float((ops[k]().float() - ref[k]).abs().max() / ref[k].abs().max())
    print(f"{k:40s} default {base[k]:7.1f} us   tuned {tuned[k]:7.1f} us   ({tuned[k] / base[k] - 1:+.1%})   max rel diff {err:.2e}")
# (TunableOp writes the results file itself when the process exits)
