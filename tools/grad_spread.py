"""Run-to-run spread of the gradients that are accumulated with fp32 atomics (scan: dA, dD, d(delta_bias); conv1d: dw, dbias; add + norm:
dweight): the same one-layer training step at the configs[2] layer shape twice from identical state, per-parameter relative difference.
Everything else (dx, every activation gradient, dW_x / dW_dt / dW_in / dW_out, the loss) is summed in a fixed order and must be bit-equal."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_config, synthetic_batch  # noqa: E402
from caduceus_amd import CaduceusForMaskedLM  # noqa: E402
dev = "cuda:0"
torch.manual_seed(5)
model = CaduceusForMaskedLM(make_config(256, 1)).to(dev).train()
ids, labels = synthetic_batch(torch.Generator().manual_seed(3), 1, 131072, dev)
runs = []
for _ in range(3):
    model.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(ids, labels=labels)
    out.loss.backward()
    torch.cuda.synchronize()
    runs.append(({k: p.grad.detach().float().clone() for k, p in model.named_parameters()}, float(out.loss)))
res = {}
for k in runs[0][0]:
    ref = runs[0][0][k]
    d = max(float((r[0][k] - ref).abs().max()) for r in runs[1:])
    res[k.split("layers.0.")[-1]] = {"bit_equal": d == 0.0, "max_abs_diff": d, "rel_to_max": d / max(float(ref.abs().max()), 1e-30)}
print(json.dumps({"loss_equal": all(r[1] == runs[0][1] for r in runs), "params": res}, indent=1))
