"""variant-effect embedding extraction (forward-only path)."""
import torch

from caduceus_amd import CaduceusConfig, Caduceus, vep

COMP = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5, 6: 6, 7: 10, 8: 9, 9: 8, 10: 7, 11: 11}


def test_find_variant_idx_and_window_mean():
    ref = torch.randint(7, 11, (4, 20))
    alt = ref.clone()
    alt[0, 10] = (ref[0, 10] - 7 + 1) % 4 + 7          # centre
    alt[1, 3] = (ref[1, 3] - 7 + 1) % 4 + 7            # off-centre
    alt[2, 5] = (ref[2, 5] - 7 + 1) % 4 + 7
    alt[2, 15] = (ref[2, 15] - 7 + 1) % 4 + 7          # two differences: the last one wins (reference loop)
    assert vep.find_variant_idx(ref, alt).tolist() == [10, 3, 15, -1]
    assert vep.find_variant_idx(ref, alt, rc=True).tolist()[1:] == [3, 15, -1]
    h = torch.randn(2, 30, 5)
    got = vep.window_mean(h, torch.tensor([15, 1]), 6)
    want0 = h[0, 12:19].mean(0)
    idx1 = torch.tensor([0, 0, 0, 1, 2, 3, 4])      # clamped: the edge token repeats
    torch.testing.assert_close(got[0], want0)
    torch.testing.assert_close(got[1], h[1, idx1].mean(0))


def test_shard_batches_matches_distributed_sampler():
    from torch.utils.data import DataLoader, DistributedSampler
    ds = list(range(53))
    for world in (1, 2, 4):
        for rank in range(world):
            sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=False, drop_last=True)
            want = [b.tolist() for b in DataLoader(ds, batch_size=3, sampler=sampler, drop_last=True)]
            assert vep.shard_batches(len(ds), rank, world, 3) == want


def test_embed_variants_rcps_and_plain(backend):
    """RCPS model: the RC-strand embedding of a sequence equals the forward-strand embedding of its reverse complement
    (what makes the reference's single forward sufficient); plain model: four forwards batched into one."""
    _, dev = backend
    torch.manual_seed(0)
    L = 96
    cfg = dict(d_model=32, n_layer=2, vocab_size=12, bidirectional=True, complement_map=dict(COMP),
               ssm_cfg=dict(d_state=8, d_conv=4, expand=2), fused_add_norm=True, rms_norm=True)
    comp = torch.tensor([COMP.get(i, i) for i in range(16)], device=dev)
    ref = torch.randint(7, 11, (3, L), device=dev)
    alt = ref.clone()
    alt[:, L // 2] = comp[ref[:, L // 2]]
    rc = lambda x: comp[x.flip(-1)]
    batch = {"ref_input_ids": ref, "alt_input_ids": alt, "ref_rc_input_ids": rc(ref), "alt_rc_input_ids": rc(alt),
             "variant_idx": vep.find_variant_idx(ref, alt)}
    ps = Caduceus(CaduceusConfig(rcps=True, **cfg)).to(dev).eval()
    f_ps = lambda ids: ps(ids, return_dict=False)
    a = vep.embed_variants(f_ps, batch, rcps=True, bp_per_token=64, autocast_dtype=None)   # window of 24 tokens
    assert a["concat_avg_ws"].shape == (3, 64) and a["rc_concat_avg_ws"].shape == (3, 64)
    # the flipped second channel half IS the first half of the RC input's output (RC equivariance), position by position;
    # like the reference, the same variant_idx is used on it
    rc_batch = {"ref_input_ids": rc(ref), "alt_input_ids": rc(alt), "variant_idx": batch["variant_idx"]}
    b = vep.embed_variants(f_ps, rc_batch, rcps=True, bp_per_token=64, autocast_dtype=None)
    torch.testing.assert_close(a["rc_concat_avg_ws"], b["concat_avg_ws"], rtol=1e-5, atol=1e-6)
    ph = Caduceus(CaduceusConfig(rcps=False, **cfg)).to(dev).eval()
    f_ph = lambda ids: ph(ids, return_dict=False)
    c = vep.embed_variants(f_ph, batch, rcps=False, bp_per_token=64, autocast_dtype=None)
    one = window = vep.window_mean(f_ph(ref), batch["variant_idx"], 24)
    torch.testing.assert_close(c["concat_avg_ws"][:, :32], one, rtol=1e-5, atol=1e-6)
    out = vep.dump_embeddings(f_ph, {**batch, "labels": torch.arange(3)}, rcps=False, batch_size=2, bp_per_token=64,
                              autocast_dtype=None)
    assert out["concat_avg_ws"].shape == (2, 64) and out["labels"].tolist() == [0, 1]
