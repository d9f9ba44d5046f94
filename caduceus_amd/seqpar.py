"""Sequence-parallel Caduceus (not in the reference): ONE sequence is split along L over the
ranks of a process group, so global batches smaller than the number of GPUs still use all of them and L >= 262144 fits.

Everything in the t-frame is token-local (embedding with complement ids, add+norm with the strand swap, every
projection, the LM head) except two steps per parameter set, which need a real exchange over RCCL / xGMI:
  * causal conv1d : a 3-position halo from each neighbour (`_HaloExchange`, two tiny all-gathers per layer);
  * selective scan: the state entering a rank's segment.  A row's segments form a chain in the row's direction
    (rank 0 -> W-1 for left-to-right rows, W-1 -> 0 for right-to-left rows; both occur in every layer, so a pipelined
    hand-off would serialise the ranks).  Instead every rank first scans its segment from a zero state
    (`hT`, `sum_dt` outputs of cad_scan_fwd), the per-segment affine maps  h -> exp(A * sum_dt) * h + hT  are
    all-gathered ((E, rows, N) floats) and composed locally, and the segment is scanned again from its true entry state
    (`h0`).  The backward does the same for the state gradient with `dhT` / `dh0` of cad_scan_bwd.  Cost: two scan
    passes per direction instead of one; exchange volume per layer ~ 4 * E * N * rows floats per rank.
Weight gradients are partial sums over a rank's tokens: reduce them with SUM over the group (BucketedGradReducer with
`average=False`), the loss with `masked_lm_loss` below.

Usage:
    with seqpar.sequence_parallel(group):           # ranks hold consecutive L/W slices of input_ids / labels
        logits = model(ids_local).logits
        loss = seqpar.masked_lm_loss(logits, labels_local, ignore_index=4)
    loss.backward()
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib as L
from . import ops

_GROUP = None
HALO = 3  # d_conv - 1 for the Caduceus configuration (d_conv = 4); checked at run time


@contextlib.contextmanager
def sequence_parallel(group=None):
    """Activates the sequence-parallel conv / scan inside engine.bimamba_tframe for the enclosed forward (and the
    backward of tensors produced in it)."""
    global _GROUP
    prev = _GROUP
    _GROUP = group if group is not None else dist.group.WORLD
    try:
        yield
    finally:
        _GROUP = prev


def active() -> bool:
    return _GROUP is not None


def _all_gather(t: torch.Tensor, group) -> List[torch.Tensor]:
    if t.is_cuda and dist.get_backend(group) == "gloo":
        # bring-up path (several ranks sharing one GPU, where RCCL refuses duplicate devices): gloo has no device all_gather,
        # the (tiny) maps / halos travel through host memory
        host = [torch.empty(t.shape, dtype=t.dtype) for _ in range(dist.get_world_size(group))]
        dist.all_gather(host, t.detach().cpu().contiguous(), group=group)
        return [h.to(t.device) for h in host]
    out = [torch.empty_like(t) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, t.contiguous(), group=group)
    return out


# ---- causal conv1d with neighbour halos ----------------------------------------------------------------------------------
class _HaloExchange(torch.autograd.Function):
    """x (E, SB, Lloc) -> (E, SB, HALO + Lloc + HALO): neighbours' edge positions attached (zeros at the global ends)."""

    @staticmethod
    def forward(ctx, x, group):
        r, W = dist.get_rank(group), dist.get_world_size(group)
        edges = _all_gather(torch.cat([x[..., :HALO], x[..., -HALO:]], -1), group)
        zeros = torch.zeros_like(edges[0][..., :HALO])
        left = edges[r - 1][..., HALO:] if r > 0 else zeros
        right = edges[r + 1][..., :HALO] if r + 1 < W else zeros
        ctx.group = group
        return torch.cat([left, x, right], -1)

    @staticmethod
    def backward(ctx, gx):
        group = ctx.group
        r, W = dist.get_rank(group), dist.get_world_size(group)
        halos = _all_gather(torch.cat([gx[..., :HALO], gx[..., -HALO:]], -1), group)
        g = gx[..., HALO:-HALO].clone()
        if r + 1 < W:  # the right neighbour's left halo is my last positions
            g[..., -HALO:] += halos[r + 1][..., :HALO]
        if r > 0:
            g[..., :HALO] += halos[r - 1][..., HALO:]
        return g, None


def causal_conv1d(x, w, bias, split, rev_lo, rev_hi):
    if w.shape[-1] - 1 > HALO:
        raise NotImplementedError("sequence-parallel conv supports d_conv <= 4")
    if x.shape[-1] < HALO:
        raise ValueError("sequence-parallel segments must hold at least 3 positions")
    ext = _HaloExchange.apply(x, _GROUP)
    return ops.causal_conv1d(ext, w, bias, split, rev_lo, rev_hi)[..., HALO:-HALO].contiguous()


# ---- selective scan with state hand-off --------------------------------------------------------------------------------
def _compose(maps_P, maps_S, rank: int, split: int, rev_lo: int, rev_hi: int, towards_end: bool):
    """State entering rank `rank` along each row's direction (towards_end=False), or the state gradient entering it from
    the far side (towards_end=True).  maps_*[q]: (E, SB, N) of rank q; a segment maps  v -> P * v + S."""
    W = len(maps_P)
    SB = maps_P[0].shape[1]
    out = torch.zeros_like(maps_S[0])
    for rows, rev in ((slice(0, split), rev_lo), (slice(split, SB), rev_hi)):
        if rows.start == rows.stop:
            continue
        forward_chain = (rev == 0) != towards_end   # ranks visited in ascending order before reaching `rank`?
        order = range(0, rank) if forward_chain else range(W - 1, rank, -1)
        v = torch.zeros_like(maps_S[0][:, rows])
        for q in order:
            v = maps_P[q][:, rows] * v + maps_S[q][:, rows]
        out[:, rows] = v
    return out


class _ScanSeqPar(torch.autograd.Function):
    """1 or 2 parameter sets (shared gate z), like ops._ScanMulti, over a row segment.  Tensor args per set:
    u, delta, A, Bm, Cm, D, delta_bias."""

    @staticmethod
    def forward(ctx, group, z, split, dirs, *tensors):
        lib = L.get_lib()
        nsets = len(tensors) // 7
        rank = dist.get_rank(group)
        z = None if z is None else z.contiguous()
        prepared = []
        for i in range(nsets):
            u, delta, A, Bm, Cm, D, bias = tensors[7 * i:7 * i + 7]
            prepared.append((u.contiguous(), delta.contiguous(), A.float().contiguous(), Bm.contiguous(), Cm.contiguous(),
                             D.float().contiguous(), bias.float().contiguous()))
        E, SB, Lq = prepared[0][0].shape
        N = prepared[0][2].shape[1]
        dev, act = prepared[0][0].device, prepared[0][0].dtype

        def launch(h0s, want_state):
            args = (L.ScanArgs * nsets)()
            outs, hTs, sdts, states = [], [], [], []
            for i, (u, delta, A, Bm, Cm, D, bias) in enumerate(prepared):
                out = torch.empty_like(u)
                hT = torch.empty((E, SB, N), dtype=torch.float32, device=dev)
                sdt = torch.empty((E, SB), dtype=torch.float32, device=dev)
                state = torch.empty((lib.cad_scan_state_floats(E, SB, Lq, N),), dtype=torch.float32, device=dev)
                stream = L.stream_and_check(u, delta, A, Bm, Cm, D, z, bias, out, state, hT, sdt, h0s[i] if h0s else None)
                args[i] = L.ScanArgs(L.ptr(u), L.ptr(delta), L.ptr(A), L.ptr(Bm), L.ptr(Cm), L.ptr(D), L.ptr(z),
                                     L.ptr(bias), L.ptr(out), L.ptr(state), SB, Lq, split, E, N, dirs[i][0], dirs[i][1],
                                     L.dtype_code(act), L.ptr(h0s[i]) if h0s else None, L.ptr(hT), L.ptr(sdt))
                outs.append(out), hTs.append(hT), sdts.append(sdt), states.append(state)
            L.check(lib.cad_scan_fwd_multi(args, nsets, stream), "cad_scan_fwd_multi")
            return outs, hTs, sdts, states

        _, hT_loc, sdt_loc, _ = launch(None, False)  # pass 1: segment from a zero state -> its affine map
        h0s, Ps = [], []
        for i in range(nsets):
            S_all = _all_gather(hT_loc[i], group)
            sdt_all = _all_gather(sdt_loc[i], group)
            A = prepared[i][2]
            P_all = [torch.exp(A.unsqueeze(1) * sd.unsqueeze(-1)) for sd in sdt_all]  # (E, SB, N) each
            h0s.append(_compose(P_all, S_all, rank, split, dirs[i][0], dirs[i][1], towards_end=False).contiguous())
            Ps.append(torch.stack(P_all))
        outs, _, _, states = launch(h0s, True)       # pass 2: from the true entry state
        ctx.group, ctx.meta = group, (split, dirs, nsets, [(t[2].dtype, t[5].dtype, t[6].dtype) for t in
                                                      [tensors[7 * i:7 * i + 7] for i in range(nsets)]])
        flat = []
        for i in range(nsets):
            flat += [*prepared[i], states[i], outs[i], Ps[i]]
        ctx.save_for_backward(z, *flat)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        lib = L.get_lib()
        z, *flat = ctx.saved_tensors
        group = ctx.group
        split, dirs, nsets, pdt = ctx.meta
        rank = dist.get_rank(group)
        sets = [flat[10 * i:10 * i + 10] for i in range(nsets)]
        E, SB, Lq = sets[0][0].shape
        N = sets[0][2].shape[1]
        dev, act = sets[0][0].device, sets[0][0].dtype
        npart = lib.cad_scan_bwd_partials(E)
        douts = [d.contiguous() for d in douts]

        def launch(dhTs):
            args = (L.ScanBwdArgs * nsets)()
            res = []
            for i, (u, delta, A, Bm, Cm, D, bias, state, fout, _P) in enumerate(sets):
                du, ddelta = torch.empty_like(u), torch.empty_like(u)
                dz = None if z is None else torch.empty_like(u)
                dA, dD, dbias = torch.zeros_like(A), torch.zeros_like(D), torch.zeros_like(bias)
                dBC = torch.empty((2, npart, N, SB, Lq), dtype=act, device=dev)
                dh0 = torch.empty((E, SB, N), dtype=torch.float32, device=dev)
                stream = L.stream_and_check(u, delta, A, Bm, Cm, D, z, bias, douts[i], state, du, ddelta, dz, dA, dBC, dD,
                                            dbias, dh0, dhTs[i] if dhTs else None)
                args[i] = L.ScanBwdArgs(L.ptr(u), L.ptr(delta), L.ptr(A), L.ptr(Bm), L.ptr(Cm), L.ptr(D), L.ptr(z),
                                        L.ptr(bias), L.ptr(douts[i]), L.ptr(fout), L.ptr(state), L.ptr(du), L.ptr(ddelta),
                                        L.ptr(dz), L.ptr(dA), L.ptr(dBC[0]), L.ptr(dBC[1]), L.ptr(dD), L.ptr(dbias), SB, Lq,
                                        split, E, N, dirs[i][0], dirs[i][1], L.dtype_code(act), npart,
                                        L.ptr(dhTs[i]) if dhTs else None, L.ptr(dh0))
                res.append([du, ddelta, dA, dBC, dD, dbias, dz, dh0])
            L.check(lib.cad_scan_bwd_multi(args, nsets, stream), "cad_scan_bwd_multi")
            return res, stream

        first, _ = launch(None)  # pass 1: state-gradient map of the segment (zero gradient entering from the far side)
        dhTs = []
        for i in range(nsets):
            G_all = _all_gather(first[i][7], group)
            P_all = list(sets[i][9].unbind(0))
            dhTs.append(_compose(P_all, G_all, rank, split, dirs[i][0], dirs[i][1], towards_end=True).contiguous())
        res, stream = launch(dhTs)
        grads, dz_tot = [], None
        for i in range(nsets):
            du, ddelta, dA, dBC, dD, dbias, dz, _ = res[i]
            n = dBC[0, 0].numel()
            dB, dC = torch.empty(dBC.shape[2:], dtype=act, device=dev), torch.empty(dBC.shape[2:], dtype=act, device=dev)
            for src, dst in ((dBC[0], dB), (dBC[1], dC)):
                L.check(lib.cad_reduce_partials(L.ptr(src), npart, n, L.ptr(dst), L.dtype_code(act), stream),
                        "cad_reduce_partials")
            Adt, Ddt, bdt = pdt[i]
            grads += [du, ddelta, dA.to(Adt), dB, dC, dD.to(Ddt), dbias.to(bdt)]
            if dz is not None:
                dz_tot = dz if dz_tot is None else dz_tot + dz
        return (None, dz_tot, None, None, *grads)


def selective_scan_multi(sets, z, split: int, dirs):
    flat = [t for s in sets for t in s]
    return _ScanSeqPar.apply(_GROUP, z, int(split), tuple(tuple(d) for d in dirs), *flat)


# ---- loss ------------------------------------------------------------------------------------------------------------------
def masked_lm_loss(logits: torch.Tensor, labels: torch.Tensor, ignore_index: int, group=None) -> torch.Tensor:
    """Mean cross-entropy over the targets of the WHOLE sequence.  Returns this rank's share (local sum / global count):
    backward gives the exact local gradients, and summing the returned values over the ranks gives the global loss."""
    group = group if group is not None else (_GROUP if _GROUP is not None else dist.group.WORLD)
    s = F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float(), labels.reshape(-1), ignore_index=ignore_index,
                        reduction="sum")
    count = (labels != ignore_index).sum().to(torch.float32)
    dist.all_reduce(count, group=group)
    return s / count.clamp_min(1.0)
