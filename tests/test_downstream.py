"""the backbone's downstream consumers (embedding model + sequence decoder)."""
import pytest
import torch

from caduceus_amd import CaduceusConfig, DNAEmbeddingModelCaduceus, SequenceDecoder

COMP = {0: 0, 1: 1, 2: 2, 3: 3, 4: 4, 5: 5, 6: 6, 7: 10, 8: 9, 9: 8, 10: 7, 11: 11}


def _cfg(rcps):
    return CaduceusConfig(d_model=32, n_layer=2, vocab_size=12, rcps=rcps, bidirectional=True, complement_map=dict(COMP),
                          ssm_cfg=dict(d_state=8, d_conv=4, expand=2), fused_add_norm=True, rms_norm=True)


def test_embedding_model_rcps_is_rc_invariant_after_strand_average(backend):
    """dna_embedding.py:180-187 + decoders.py:146-151: for RCPS the two stacked strands swap under reverse complement,
    so the conjoined (strand-averaged) prediction is RC-invariant."""
    _, dev = backend
    torch.manual_seed(0)
    emb = DNAEmbeddingModelCaduceus(_cfg(True)).to(dev).eval()
    dec = SequenceDecoder(32, d_output=3, l_output=0, mode="pool", conjoin_test=True).to(dev).eval()
    ids = torch.randint(7, 11, (2, 50), device=dev)
    comp = torch.tensor([COMP.get(i, i) for i in range(16)], device=dev)
    rc_ids = comp[ids.flip(-1)]
    with torch.no_grad():
        h, none = emb(ids)
        h_rc, _ = emb(rc_ids)
        y, y_rc = dec(h), dec(h_rc)
    assert none is None and h.shape == (2, 50, 32, 2)
    # the stacked strands are both in the forward frame, so reverse-complementing the input just swaps them
    assert torch.equal(h[..., 0], h_rc[..., 1]) and torch.equal(h[..., 1], h_rc[..., 0])
    torch.testing.assert_close(y, y_rc, rtol=1e-5, atol=1e-6)


def test_embedding_model_conjoin_two_passes(backend):
    _, dev = backend
    torch.manual_seed(0)
    emb = DNAEmbeddingModelCaduceus(_cfg(False), conjoin_train=True).to(dev)
    ids = torch.randint(7, 11, (2, 40, 2), device=dev)
    h, _ = emb(ids)
    assert h.shape == (2, 40, 32, 2)
    torch.testing.assert_close(h[..., 1], emb.caduceus(ids[..., 1], return_dict=False))
    with pytest.raises(AssertionError):
        emb(ids[..., 0])
    emb.conjoin_train = False
    emb.train()
    assert emb(ids[..., 0])[0].shape == (2, 40, 32)


@pytest.mark.parametrize("mode", ["last", "first", "pool", "sum"])
def test_sequence_decoder_modes(mode):
    torch.manual_seed(1)
    x = torch.randn(3, 10, 4)
    dec = SequenceDecoder(4, d_output=None, l_output=3, mode=mode)
    y = dec(x)
    want = {"last": x[:, -3:], "first": x[:, :3],
            "pool": torch.stack([x[:, :8].mean(1), x[:, :9].mean(1), x[:, :10].mean(1)], 1),
            "sum": torch.stack([x[:, :8].sum(1), x[:, :9].sum(1), x[:, :10].sum(1)], 1)}[mode]
    torch.testing.assert_close(y, want, rtol=1e-5, atol=1e-6)
    assert SequenceDecoder(4, l_output=0, mode=mode)(x).shape == (3, 4)
    lens = [10, 7, 5]
    yl = SequenceDecoder(4, l_output=0, mode=mode, use_lengths=True)(x, lengths=lens)
    ref1 = SequenceDecoder(4, l_output=0, mode=mode)(x[1:2, :7])
    torch.testing.assert_close(yl[1:2], ref1)


def test_sequence_decoder_errors_and_step():
    x = torch.randn(2, 6, 4, 2)
    with pytest.raises(NotImplementedError):
        SequenceDecoder(4, mode="median")(x[..., 0])
    with pytest.raises(AssertionError):
        SequenceDecoder(4, mode="ragged", use_lengths=True)
    assert SequenceDecoder(4, mode="ragged")(x[..., 0], lengths=[4, 3]).shape == (2, 4, 4)
    dec = SequenceDecoder(4, d_output=2, l_output=0, mode="pool", conjoin_train=True)
    y = dec(x)
    lin = dec.output_transform
    want = (lin(x[..., 0].mean(1)) + lin(x[..., 1].mean(1))) / 2
    torch.testing.assert_close(y, want, rtol=1e-5, atol=1e-6)
    assert dec.step(torch.randn(2, 6, 4)).shape == (2, 2)
