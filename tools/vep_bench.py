"""Forward-only (inference) throughput of the backbone at seqlen 131072 through the variant-effect embedding step
(caduceus_amd/vep.py; reference vep_embeddings.py:352-392): Caduceus-PS d_model 256, n_layer 16, bf16 autocast, ref + alt
sequences of `--batch` variants per step.  Usage on the GPU box: python tools/vep_bench.py"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import COMP, make_config  # noqa: E402
from caduceus_amd import Caduceus, vep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--seqlen", type=int, default=131072)
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = Caduceus(make_config(256, 16)).to(dev).eval()
    comp = torch.tensor([COMP.get(i, i) for i in range(16)], device=dev)
    ref = torch.randint(7, 11, (a.batch, a.seqlen), device=dev)
    alt = ref.clone()
    alt[:, a.seqlen // 2] = comp[ref[:, a.seqlen // 2]]
    batch = {"ref_input_ids": ref, "alt_input_ids": alt, "variant_idx": vep.find_variant_idx(ref, alt)}
    f = lambda ids: model(ids, return_dict=False)
    out = vep.embed_variants(f, batch, rcps=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = vep.embed_variants(f, batch, rcps=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"metric": "forward-only tokens/s (ref + alt sequences), Caduceus-PS d256 n16 bf16", "seqlen": a.seqlen,
                      "variants_per_step": a.batch, "ms_per_step": round(dt * 1e3, 2),
                      "tokens_per_s": round(2 * a.batch * a.seqlen / dt), "embedding_shape": list(out["concat_avg_ws"].shape),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}))


if __name__ == "__main__":
    main()
