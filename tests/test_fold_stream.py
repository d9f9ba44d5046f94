"""The dB / dC fold that runs BEHIND a running scan backward (cad_fold_partials_stream, include/caduceus_hip.h): same sums, bit for bit,
as the fold kernel it takes off the critical path (cad_reduce_partials_multi), whichever of the two -- or which mixture of the two,
after a give-up -- produced a gradient.  The reference reduces dB / dC over the channels inside selective_scan_cuda.bwd
(mamba_inner_fn reached from /root/reference/caduceus/modeling_caduceus.py:11,128,130)."""
import ctypes as C

import pytest
import torch

from caduceus_amd import _lib as CL
from caduceus_amd import mixer, ops
from caduceus_amd.mamba import Mamba

CHUNK = 512


def _slots(dev, G, N, SB, L, seed):
    g = torch.Generator().manual_seed(seed)
    # magnitudes spread over a few binades and both signs: the order of the fp32 additions matters for such data
    x = torch.randn((2, G, N, SB, L), generator=g) * torch.exp2(torch.randint(-6, 3, (2, G, N, 1, 1), generator=g).float())
    return x.to(torch.bfloat16).to(dev)


def _reference_fold(lib, slots, G, N, SB, L, stream):
    out = torch.empty((2, N, SB, L), dtype=torch.bfloat16, device=slots.device)
    jobs = (CL.ReduceJob * 4)()
    for t in range(2):
        jobs[t] = CL.ReduceJob(CL.ptr(slots[t]), CL.ptr(out[t]))
    CL.check(lib.cad_reduce_partials_multi(jobs, 2, G, N * SB * L, CL.dtype_code(torch.bfloat16), stream), "reduce")
    return out


def _fold_args(slots, out, counters, aborts, G, N, SB, L, split, rev_lo, rev_hi):
    a = (CL.FoldArgs * 1)()
    a[0] = CL.FoldArgs(CL.ptr(slots[0]), CL.ptr(slots[1]), CL.ptr(out[0]), CL.ptr(out[1]), CL.ptr(counters), CL.ptr(aborts), SB, L,
                       split, N, G, rev_lo, rev_hi, CL.dtype_code(torch.bfloat16))
    return a


@pytest.mark.parametrize("G,N,SB,L,split", [(64, 16, 2, 2 * CHUNK, 1), (128, 16, 2, CHUNK, 1), (32, 16, 3, 3 * CHUNK, 2),
                                            (8, 16, 1, CHUNK, 0), (16, 8, 2, 2 * CHUNK, 2), (64, 4, 2, CHUNK, 1),
                                            (128, 16, 5, 2 * CHUNK, 2)])  # 640 items: more than one per workgroup (256 CUs)
def test_fold_stream_bit_identical_to_reduce_partials(backend, G, N, SB, L, split):
    name, dev = backend
    lib = CL.get_lib()
    assert lib.cad_fold_stream_supported(N, G, L, CL.dtype_code(torch.bfloat16)) == 1
    slots = _slots(dev, G, N, SB, L, seed=G + N + SB)
    stream = CL.stream_and_check(slots)
    want = _reference_fold(lib, slots, G, N, SB, L, stream)
    nch = L // CHUNK
    # (a) everything, no polling
    out = torch.full_like(want, float("nan"))
    # chunk arrivals [row][chunk] + the "workgroups started" count + the per-CU marks
    cflat = torch.zeros((int(lib.cad_scan_bwd_fold_counter_ints(SB, L)),), dtype=torch.int32, device=dev)
    counters = cflat[:SB * nch].view(SB, nch)
    cflat[SB * nch] = G * SB  # "every scan workgroup has been placed": no fold workgroup is in a scan workgroup's way
    aborts = torch.zeros((SB, G), dtype=torch.int32, device=dev)
    a = _fold_args(slots, out, counters, aborts, G, N, SB, L, split, 0, 1)
    CL.check(lib.cad_fold_partials_stream(a, 1, 2, stream), "fold all")
    assert torch.equal(out, want)
    # (b) concurrent mode with every chunk already published: no wait, the same sums; the cleanup pass finds nothing
    out.fill_(float("nan"))
    counters.fill_(G)
    CL.check(lib.cad_fold_partials_stream(a, 1, 0, stream), "fold concurrent")
    assert torch.equal(out, want) and int(aborts.abs().sum()) == 0
    CL.check(lib.cad_fold_partials_stream(a, 1, 1, stream), "fold cleanup")
    assert torch.equal(out, want)
    # (c) a chunk that never completes (one arrival missing in row 0): every slice of that row gives up AT that chunk after the bounded
    # wait, the cleanup pass folds the rest -- the union is the same gradient
    out.fill_(float("nan"))
    stuck = nch - 1 if nch == 1 else nch - 2  # chunks are taken from the last logical one down
    counters[0, stuck] = G - 1
    CL.check(lib.cad_fold_partials_stream(a, 1, 0, stream), "fold concurrent (stuck)")
    assert torch.equal(aborts[0], torch.full((G,), stuck + 1, dtype=torch.int32, device=dev))
    assert int(aborts[1:].abs().sum()) == 0
    assert not torch.equal(torch.nan_to_num(out), torch.nan_to_num(want))  # row 0 is incomplete
    if SB > 1:
        assert torch.equal(out[:, :, 1:], want[:, :, 1:])
    CL.check(lib.cad_fold_partials_stream(a, 1, 1, stream), "fold cleanup (stuck)")
    assert torch.equal(out, want) and int(aborts.abs().sum()) == 0


def test_fold_stream_shapes_refused():
    from caduceus_amd import _lib
    lib = _lib.get_lib()
    bf, f32 = CL.dtype_code(torch.bfloat16), CL.dtype_code(torch.float32)
    assert lib.cad_fold_stream_supported(16, 64, 131072, bf) == 1      # configs[2]
    assert lib.cad_fold_stream_supported(16, 128, 262144, bf) == 1     # configs[4]
    assert lib.cad_fold_stream_supported(16, 32, 1024, bf) == 1        # d_model 128
    assert lib.cad_fold_stream_supported(16, 64, 131072, f32) == 0     # fp32 slots: the fold kernel behind the scan
    assert lib.cad_fold_stream_supported(16, 64, 1000, bf) == 0        # ragged last chunk
    assert lib.cad_fold_stream_supported(16, 48, 1024, bf) == 0        # depth not a power of two
    assert lib.cad_fold_stream_supported(16, 4, 1024, bf) == 0


def _layer(dev, d_model, L, seed=0):
    torch.manual_seed(seed)
    mf, mr = Mamba(d_model, device=dev), Mamba(d_model, device=dev)
    mr.in_proj.weight = mf.in_proj.weight
    mr.out_proj.weight = mf.out_proj.weight
    hn = (torch.randn(2, 1, L, d_model, device=dev) * 0.5).to(torch.bfloat16).requires_grad_(True)
    g = torch.randn(2, 1, L, d_model, device=dev).to(torch.bfloat16)
    return mf, mr, hn, g


def _layer_grads(mf, mr, hn, g):
    for p in list(mf.parameters()) + list(mr.parameters()):
        p.grad = None
    hn.grad = None
    mixer.prepare_step_cache([(mf, mr)], torch.bfloat16)
    out = mixer.bimamba_mixer(hn, mf, mr, 1)
    out.backward(g)
    gr = {"hn": hn.grad.clone()}
    for tag, m in (("f", mf), ("r", mr)):
        for n, p in m.named_parameters():
            gr[f"{tag}.{n}"] = p.grad.clone()
    return gr


def test_mixer_layer_gradients_identical_with_and_without_the_stream_fold(backend, monkeypatch):
    """The production mixer layer, forward + backward: the fold on the second stream (default) against the fold kernel behind the scan."""
    name, dev = backend
    d_model, L = (32, 2 * CHUNK) if name == "emu" else (256, 16 * CHUNK)
    mf, mr, hn, g = _layer(dev, d_model, L)
    monkeypatch.setattr(mixer, "_STREAM_FOLD", False)
    ref = _layer_grads(mf, mr, hn, g)
    monkeypatch.setattr(mixer, "_STREAM_FOLD", True)
    monkeypatch.setattr(mixer, "_STREAM_FOLD_MIN_CHUNKS", 1)  # (the emulator's layer has two chunks per row)
    npart = CL.get_lib().cad_scan_bwd_partials(2 * d_model)
    assert ops.fold_stream_supported(16, npart, L, torch.bfloat16), "the layer must take the stream fold"
    got = _layer_grads(mf, mr, hn, g)
    for k in ref:
        if "conv1d" in k and name == "hip":  # sums of fp32 atomics: the order of the additions is not fixed on the device
            torch.testing.assert_close(got[k], ref[k], rtol=1e-5, atol=1e-5 * float(ref[k].abs().max()))
        else:
            assert torch.equal(ref[k], got[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("d_model,L,launches", [(256, 131072, 200), (512, 262144, 100)])
def test_stream_fold_stress_back_to_back_launches(d_model, L, launches):
    """Device stress test of the cross-kernel hand-off (write-through slot stores, drained waves, agent-scope counters, sc1 loads in a
    kernel on another stream): `launches` production layers back to back at the configs[2] / configs[4] layer shapes, every gradient
    compared bit for bit with the fold kernel's.  A stale slot line, a counter overtaking a store or a fold that never saw a chunk
    shows up as a differing element (or as give-up records, which are reported)."""
    from caduceus_amd import _lib
    _lib.use_library_for_testing(None)
    dev = torch.device("cuda:0")
    mf, mr, hn, g = _layer(dev, d_model, L)
    old = mixer._STREAM_FOLD
    try:
        mixer._STREAM_FOLD = False
        ref = _layer_grads(mf, mr, hn, g)
        mixer._STREAM_FOLD = True
        ops.FOLD_GIVE_UPS = []
        # (the conv weight / bias gradients are sums of fp32 atomics over the workgroups of a channel: run-to-run differences in the last
        # bits with or without the fold -- compared to a tolerance; everything else, dB / dC included through dW_x and d(hn), bit for bit)
        keys = sorted(k for k in ref if "conv1d" not in k)
        conv = sorted(k for k in ref if "conv1d" in k)
        wrong = torch.zeros((launches, len(keys)), dtype=torch.int64, device=dev)  # counted on the device: no host sync between launches
        conv_err = torch.zeros((launches, len(conv)), dtype=torch.float32, device=dev)
        for it in range(launches):
            got = _layer_grads(mf, mr, hn, g)
            for j, k in enumerate(keys):
                wrong[it, j] = (ref[k] != got[k]).sum()
            for j, k in enumerate(conv):
                conv_err[it, j] = (ref[k] - got[k]).abs().max() / ref[k].abs().max()
        torch.cuda.synchronize()
        bad = {keys[j]: [(int(i), int(wrong[i, j])) for i in torch.nonzero(wrong[:, j]).flatten()[:8]]
               for j in range(len(keys)) if int(wrong[:, j].sum())}
        gave_up = [int(x) for x in ops.FOLD_GIVE_UPS]
        assert not bad, (bad, "give-up records per launch:", gave_up)
        assert float(conv_err.max()) < 1e-5, conv_err.max(0)
        # co-scheduled as designed, the concurrent pass leaves nothing to the cleanup
        assert sum(gave_up) == 0, gave_up
    finally:
        mixer._STREAM_FOLD = old
        ops.FOLD_GIVE_UPS = None


def test_fold_f32_multi_sums_every_job_in_one_launch(backend):
    """cad_fold_f32_multi: plain folds of different depths and lengths and a two-level fold (the two halves of the tied out_proj's
    weight gradient added while its K slices are summed) in ONE launch, against fp64 sums of the same data."""
    name, dev = backend
    g = torch.Generator().manual_seed(11)
    a = torch.randn((7, 96, 20), generator=g).to(dev)        # 7 parts of 1920
    b = torch.randn((33, 8, 16), generator=g).to(dev)        # 33 parts of 128
    c = torch.randn((5, 2, 12, 16), generator=g).to(dev)     # 5 slices x 2 halves of 192
    d = torch.randn((1, 64), generator=g).to(dev)            # a single part: a copy
    outs = [torch.full((96, 20), float("nan"), device=dev), torch.full((8, 16), float("nan"), device=dev),
            torch.full((12, 16), float("nan"), device=dev), torch.full((64,), float("nan"), device=dev)]
    ops.fold_f32([(a, outs[0], 1920, 7, 1920, 1, 0), (b, outs[1], 128, 33, 128, 1, 0), (c, outs[2], 192, 5, 384, 2, 192),
                  (d, outs[3], 64, 1, 64, 1, 0)])
    want = [a.double().sum(0), b.double().sum(0), c.double().sum((0, 1)), d.double()[0]]
    for o, w in zip(outs, want):
        torch.testing.assert_close(o.double(), w, rtol=1e-6, atol=1e-6)
    # deterministic: the same launch again gives the same bits
    again = [torch.empty_like(o) for o in outs]
    ops.fold_f32([(a, again[0], 1920, 7, 1920, 1, 0), (b, again[1], 128, 33, 128, 1, 0), (c, again[2], 192, 5, 384, 2, 192),
                  (d, again[3], 64, 1, 64, 1, 0)])
    assert all(torch.equal(x, y) for x, y in zip(outs, again))


def test_mixer_layer_glue_fold_matches_the_torch_sums(backend, monkeypatch):
    """The weight gradients of the production layer with their partial tiles summed by the one own launch (default) and by torch.sum:
    the same fp32 sums up to the order of the additions."""
    name, dev = backend
    d_model, L = (32, 2 * CHUNK) if name == "emu" else (256, 16 * CHUNK)
    mf, mr, hn, g = _layer(dev, d_model, L)
    monkeypatch.setattr(mixer, "_GLUE_FOLD", False)
    ref = _layer_grads(mf, mr, hn, g)
    monkeypatch.setattr(mixer, "_GLUE_FOLD", True)
    got = _layer_grads(mf, mr, hn, g)
    for k in ref:
        if k == "hn" or "conv1d" in k or k.endswith((".A_log", ".D", ".dt_proj.bias")):
            tol = dict(rtol=1e-5, atol=1e-5 * float(ref[k].float().abs().max())) if "conv1d" in k and name == "hip" else dict(rtol=0, atol=0)
            torch.testing.assert_close(got[k].float(), ref[k].float(), **tol)  # not touched by the fold
        else:
            torch.testing.assert_close(got[k], ref[k], rtol=2e-5, atol=2e-6 * float(ref[k].abs().max()), msg=lambda m, k=k: f"{k}: {m}")
