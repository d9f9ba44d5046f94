"""Data-parallel path on CPU: 2 processes, gloo backend, kernels from the host emulator.  Each rank trains on its own
sequence; after the bucketed all-reduce both ranks must hold the mean of the per-rank gradients, and gradient
accumulation under no_sync() must defer the collective."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden_model


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_model():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    from build_emu import build_emu
    from caduceus_amd import CaduceusConfig, CaduceusForMaskedLM, _lib
    _lib.use_library_for_testing(build_emu())
    cfg, sd, rec = load_golden_model("ps_fused")
    model = CaduceusForMaskedLM(CaduceusConfig(**cfg, pad_token_id=4))
    model.load_state_dict(sd)
    return model.train(), rec


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from caduceus_amd.dp import BucketedGradReducer
    model, rec = _setup_model()
    reducer = BucketedGradReducer(model.parameters(), bucket_bytes=16 << 10)  # several small buckets
    assert len(reducer.buckets) >= 3
    ids, labels = rec["input_ids"][rank:rank + 1], rec["labels"][rank:rank + 1]
    # micro-step 1 without sync, micro-step 2 with sync: all-reduced result = mean over ranks of (g1 + g2)
    reducer.zero_grad()
    with reducer.no_sync():
        model(ids, labels=labels).loss.backward()
    local_after_one = {n: p.grad.clone() for n, p in model.named_parameters()}
    model(ids.flip(-1), labels=labels.flip(-1)).loss.backward()
    reducer.finish()
    torch.save({"one": local_after_one, "final": {n: p.grad.clone() for n, p in model.named_parameters()}},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt")
    r1 = torch.load(tmp_path / "rank1.pt")
    # reference: single process, same computation per "rank", averaged by hand
    model, rec = _setup_model()
    expect = None
    per_rank_one = []
    for rank in range(2):
        model.zero_grad(set_to_none=True)
        ids, labels = rec["input_ids"][rank:rank + 1], rec["labels"][rank:rank + 1]
        model(ids, labels=labels).loss.backward()
        per_rank_one.append({n: p.grad.clone() for n, p in model.named_parameters()})
        model(ids.flip(-1), labels=labels.flip(-1)).loss.backward()
        g = {n: p.grad.clone() for n, p in model.named_parameters()}
        expect = g if expect is None else {n: expect[n] + g[n] for n in g}
    expect = {n: v / 2 for n, v in expect.items()}
    for n in expect:
        torch.testing.assert_close(r0["final"][n], expect[n], rtol=1e-5, atol=1e-6, msg=lambda m, n=n: f"{n}: {m}")
        assert torch.equal(r0["final"][n], r1["final"][n]), n  # both ranks hold identical reduced gradients
        # no collective happened during the no_sync micro-step
        torch.testing.assert_close(r0["one"][n], per_rank_one[0][n], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r1["one"][n], per_rank_one[1][n], rtol=1e-5, atol=1e-6)
    from caduceus_amd import _lib
    _lib.use_library_for_testing(None)
