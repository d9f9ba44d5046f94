"""Flip-free "t-frame" composition of the Caduceus layer on top of the HIP kernels (caduceus_amd.ops).

Reference formulation (per layer, Caduceus-PS):   RCPSMambaBlock -> RCPSWrapper -> 2x BiMambaWrapper -> 4x Mamba
(/root/reference/caduceus/modeling_rcps.py:160-199, :85-99; modeling_caduceus.py:122-140), i.e. ~10 flips, 3 cats,
duplicated in/out projections.  Here:

  * the two D-wide strands of the RCPS stream are rows of ONE batch: hidden is (S, B, L, D) with strand 1 stored
    channel-reversed, so BOTH strands use every weight in natural order (one GEMM over S*B*L tokens);
  * `in_proj` runs once per token and is shared by both scan directions (weight tying, modeling_caduceus.py:114-118);
  * a direction is an index map inside the conv / scan kernels: parameter set `mamba_fwd` runs strand 0 left-to-right
    and strand 1 right-to-left, `mamba_rev` the opposite -- no `flip`, no `cat`;
  * activations between GEMMs and scans are channel-major (E, S*B, L): the in_proj GEMM writes that layout directly
    (W @ X^T) and every scan/conv access is a contiguous run along L.

On THIS (generic, per-op autograd) path the dense projections go through ops.mm (fp32: the own cad_gemm_f32; bf16: torch.mm / hipBLASLt); the production configuration (tied, "add") runs
mixer.BiMambaMixerFn instead, whose projections are the library's own MFMA kernels (csrc/gemm.hip) -- no library GEMM on that step.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import mixer, ops, seqpar


def _w(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    return t if t.dtype == dtype else t.to(dtype)


def _scan_inputs(xz: torch.Tensor, m, split: int, rev_lo: int, rev_hi: int, act: torch.dtype):
    """conv -> x_proj -> dt_proj for one Mamba parameter set `m` on channel-major xz (2E, SB, L): the operands of the
    selective scan.  Math: SURVEY.md section 7.2 (upstream mamba_inner_fn)."""
    E2, SB, L = xz.shape
    E = E2 // 2
    T = SB * L
    N, R = m.d_state, m.dt_rank
    conv = seqpar.causal_conv1d if seqpar.active() else ops.causal_conv1d
    xc = conv(xz[:E], m.conv1d.weight, m.conv1d.bias, split, rev_lo, rev_hi)
    dbc = ops.mm(_w(m.x_proj.weight, act), xc.view(E, T)).view(R + 2 * N, SB, L)
    delta = ops.mm(_w(m.dt_proj.weight, act), dbc[:R].reshape(R, T)).view(E, SB, L)
    A = -torch.exp(m.A_log.float())
    return (xc, delta, A, dbc[R:R + N], dbc[R + N:], m.D.float(), m.dt_proj.bias.float())


def _in_proj(m, x2d: torch.Tensor, SB: int, L: int, act: torch.dtype) -> torch.Tensor:
    xz = ops.mm(_w(m.in_proj.weight, act), x2d.t())
    if m.in_proj.bias is not None:
        xz = xz + _w(m.in_proj.bias, act).unsqueeze(1)
    return xz.view(-1, SB, L)


def _out_proj(m, y: torch.Tensor, acc: Optional[torch.Tensor], act: torch.dtype) -> torch.Tensor:
    E = y.shape[0]
    yt = y.view(E, -1).t()
    wt = _w(m.out_proj.weight, act).t()
    out = ops.mm(yt, wt) if acc is None else ops.addmm(acc, yt, wt)
    if m.out_proj.bias is not None:
        out = out + _w(m.out_proj.bias, act)
    return out


def bimamba_tframe(hn: torch.Tensor, mamba_fwd, mamba_rev, strategy: Optional[str], strand_swap: bool) -> torch.Tensor:
    """BiMambaWrapper (+ RCPSWrapper when hn has two strands) on normed t-frame input hn (S, B, L, D).

    strand_swap=True is the RCPS case: rows of strand 1 run every parameter set in the opposite direction."""
    dev = hn.device.type
    if torch.is_autocast_enabled(dev) and torch.get_autocast_dtype(dev) != hn.dtype:
        # a float16 request is computed in fp32 (mamba.act_dtype_of): keep autocast from re-casting the GEMMs in here
        with torch.autocast(dev, enabled=False):
            return _bimamba_tframe(hn, mamba_fwd, mamba_rev, strategy, strand_swap)
    return _bimamba_tframe(hn, mamba_fwd, mamba_rev, strategy, strand_swap)


def _bimamba_tframe(hn: torch.Tensor, mamba_fwd, mamba_rev, strategy: Optional[str], strand_swap: bool) -> torch.Tensor:
    S, B, L, D = hn.shape
    SB, T = S * B, S * B * L
    act = hn.dtype
    x2d = hn.reshape(T, D)
    split = B if (S == 2 and strand_swap) else SB
    scan_multi = seqpar.selective_scan_multi if seqpar.active() else ops.selective_scan_multi
    if not seqpar.active() and mixer.can_use(mamba_fwd, mamba_rev, strategy):  # released-model configuration
        return mixer.bimamba_mixer(hn, mamba_fwd, mamba_rev, split)
    xz_f = _in_proj(mamba_fwd, x2d, SB, L, act)
    E = xz_f.shape[0] // 2
    set_f = _scan_inputs(xz_f, mamba_fwd, split, 0, 1, act)
    if mamba_rev is None:
        y_f = scan_multi([set_f], xz_f[E:], split, [(0, 1)])[0]
        return _out_proj(mamba_fwd, y_f, None, act).view(S, B, L, D)
    tied_in = mamba_rev.in_proj.weight is mamba_fwd.in_proj.weight and mamba_rev.in_proj.bias is mamba_fwd.in_proj.bias
    xz_r = xz_f if tied_in else _in_proj(mamba_rev, x2d, SB, L, act)
    set_r = _scan_inputs(xz_r, mamba_rev, split, 1, 0, act)
    if tied_in:  # both parameter sets share the gate z: ONE launch runs the forward- and reverse-direction scans
        y_f, y_r = scan_multi([set_f, set_r], xz_f[E:], split, [(0, 1), (1, 0)])
    else:
        y_f = scan_multi([set_f], xz_f[E:], split, [(0, 1)])[0]
        y_r = scan_multi([set_r], xz_r[E:], split, [(1, 0)])[0]
    strategy = strategy or "add"
    if strategy == "add":
        tied_out = (mamba_rev.out_proj.weight is mamba_fwd.out_proj.weight and mamba_fwd.out_proj.bias is None
                    and mamba_rev.out_proj.bias is None)
        out_f = _out_proj(mamba_fwd, y_f, None, act)
        if tied_out:  # out_proj(y_f) + out_proj(y_r) accumulated inside the second GEMM
            out = _out_proj(mamba_rev, y_r, out_f, act)
        else:
            out = out_f + _out_proj(mamba_rev, y_r, None, act)
    elif strategy == "ew_multiply":
        out = _out_proj(mamba_fwd, y_f, None, act) * _out_proj(mamba_rev, y_r, None, act)
    else:
        raise NotImplementedError(f"`{strategy}` for bi-directionality not implemented!")
    return out.view(S, B, L, D)


# ---- frame conversions at the API boundary (reference frame <-> t-frame) ------------------------------------------

def to_tframe(x: torch.Tensor, rcps: bool) -> torch.Tensor:
    """(B, L, 2D) reference RCPS stream -> (2, B, L, D);  (B, L, D) -> (1, B, L, D)."""
    if not rcps:
        return x.unsqueeze(0)
    D = x.shape[-1] // 2
    return torch.stack([x[..., :D], x[..., D:].flip(-1)], dim=0)


def from_tframe(t: torch.Tensor) -> torch.Tensor:
    if t.shape[0] == 1:
        return t[0]
    return torch.cat([t[0], t[1].flip(-1)], dim=-1)
