"""Which operators launch the FillFunctor / small reduce / copy kernels of a training step (torch.profiler: CPU op -> its kernels).
4-layer Caduceus-PS at L = 131072, one bench-style step (forward, backward, gradient gather, clip, AdamW)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from caduceus_amd import CaduceusForMaskedLM  # noqa: E402
from caduceus_amd.dp import BucketedGradReducer  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = CaduceusForMaskedLM(bench.make_config(256, 4)).to(dev).train()
reducer = BucketedGradReducer(model.parameters())
opt = torch.optim.AdamW(model.parameters(), lr=1e-3, fused=True)
gen = torch.Generator().manual_seed(1)
ids, labels = bench.synthetic_batch(gen, 1, 131072, dev)


def step():
    reducer.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(ids, labels=labels)
    out.loss.backward()
    reducer.finish()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    for k in getattr(e, "kernels", []) or []:
        if any(s in k.name for s in ("FillFunctor", "reduce_kernel", "copy_kernel", "copyBuffer", "fillBuffer")):
            st = [f for f in (e.stack or []) if "caduceus_amd" in f or "bench" in f or "tools/" in f]
            cnt[(k.name[:60], e.name, str(e.input_shapes)[:80], st[0][-90:] if st else "")] += 1
for (kn, op, shp, st), n in cnt.most_common(60):
    print(n, "|", kn, "|", op, "|", shp, "|", st)
