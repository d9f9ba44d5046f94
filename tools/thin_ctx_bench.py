"""x_proj (cad_proj_wx thin, M = 48, K = 512) timed alone and right behind a kernel that has just WRITTEN its (K, T) operand, as in
the training step (conv1d_fwd writes xc, x_proj reads it): 56 us alone, 88 us in the step trace."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from caduceus_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
T, E = 262144, 512
r = lambda *s: torch.randn(*s, device=dev, dtype=torch.bfloat16)
W, X, X2 = r(48, E) * 0.06, r(E, T), r(E, T)
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
out = torch.empty(48, T, device=dev, dtype=torch.bfloat16)


def timed(pre, reps=20):
    tot = 0.0
    for i in range(reps + 3):
        pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.proj_wx(W, X, out=out)
        b.record()
        torch.cuda.synchronize()
        if i >= 3:
            tot += a.elapsed_time(b)
    return round(tot / reps * 1e3, 1)


res = {"alone_us": timed(lambda: None),
       "after_writing_X_us": timed(lambda: X.copy_(X2)),
       "after_writing_X_twice_us": timed(lambda: (X2.copy_(X), X.copy_(X2))),
       "after_1GiB_memset_us": timed(lambda: big.zero_()),
       "after_reading_X_us": timed(lambda: X.sum())}
print(json.dumps(res))
