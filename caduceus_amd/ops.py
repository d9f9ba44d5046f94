"""Autograd-aware Python entry points over the C-ABI kernels (include/caduceus_hip.h).

Each function mirrors one interface the reference reaches through its third-party native dependencies
(mamba_ssm `selective_scan_fn` / `causal_conv1d_fn` / `rms_norm_fn`; call sites cited in the header), but in the
flip-free "t-frame" / channel-major layouts described in DESIGN.md.  PyTorch is used here only as the owner of device
memory, streams and the autograd graph; all arithmetic of these ops happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib as L


def _zeros_like_f32(t, shape=None):
    return torch.zeros(t.shape if shape is None else shape, dtype=torch.float32, device=t.device)


# ------------------------------------------------------------------------------------------------------------------
# embedding
# ------------------------------------------------------------------------------------------------------------------
class _Embed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ids, weight, comp, n_strands, out_dtype):
        ids = ids.contiguous()
        w = weight.contiguous()
        B, Lq = ids.shape
        V, D = w.shape
        out = torch.empty((n_strands, B, Lq, D), dtype=out_dtype, device=w.device)
        stream = L.stream_and_check(ids, w, comp, out)
        a = L.EmbedArgs(L.ptr(ids), L.ptr(comp), L.ptr(w), L.ptr(out), B, Lq, D, V, n_strands, L.dtype_code(w.dtype),
                        L.dtype_code(out_dtype))
        L.check(L.get_lib().cad_embed_fwd(C.byref(a), stream), "cad_embed_fwd")
        ctx.save_for_backward(ids, comp)
        ctx.meta = (V, D, n_strands, w.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, comp = ctx.saved_tensors
        V, D, n_strands, wdt = ctx.meta
        dout = dout.contiguous()
        dw = torch.zeros((V, D), dtype=torch.float32, device=dout.device)
        if V * D * 4 > 64 * 1024:
            # the kernel keeps the whole (V, D) gradient in LDS (DNA vocabularies: 16 x D); larger vocabularies
            # (CaduceusConfig's default vocab_size is 50277) scatter-add through the allocator-owning framework instead
            flat = ids.reshape(-1)
            dw.index_add_(0, flat, dout[0].reshape(-1, D).float())
            if n_strands == 2:
                dw.index_add_(0, comp[flat], dout[1].reshape(-1, D).float())
            return None, dw.to(wdt), None, None, None
        stream = L.stream_and_check(ids, comp, dout, dw)
        B, Lq = ids.shape
        a = L.EmbedBwdArgs(L.ptr(ids), L.ptr(comp), L.ptr(dout), L.ptr(dw), B, Lq, D, V, n_strands,
                           L.dtype_code(dout.dtype))
        L.check(L.get_lib().cad_embed_bwd(C.byref(a), stream), "cad_embed_bwd")
        return None, dw.to(wdt), None, None, None


def embed(ids: torch.Tensor, weight: torch.Tensor, comp: Optional[torch.Tensor], n_strands: int,
          out_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """(B, L) int64 ids -> (n_strands, B, L, D) t-frame embedding.  RCPSEmbedding (modeling_rcps.py:46-67)."""
    if ids.dtype != torch.int64:
        ids = ids.long()
    if n_strands == 2 and comp is None:
        raise AssertionError("Complement map must be provided for RCPS.")  # modeling_caduceus.py:350
    return _Embed.apply(ids, weight, comp, n_strands, out_dtype)


# ------------------------------------------------------------------------------------------------------------------
# fused add + norm
# ------------------------------------------------------------------------------------------------------------------
# set by mixer.set_fp8_in_proj: add_norm also emits the e4m3 copy of the normed activations (cad_add_norm_args.y_fp8 / y_scale)
FP8_ACTIVATIONS = False


class _AddNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps, is_rms, swap_flip, y_dtype, want_fp8=False):
        x = x.contiguous()
        S, D = x.shape[0], x.shape[-1]
        rows = x.numel() // (S * D)
        w = weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        res = None if residual is None else residual.contiguous()
        if res is not None and res.dtype != torch.float32:
            raise TypeError("caduceus_amd keeps the residual stream in fp32")
        y = torch.empty(x.shape, dtype=y_dtype, device=x.device)
        res_out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        rstd = torch.empty((S * rows,), dtype=torch.float32, device=x.device)
        mean = None if is_rms else torch.empty((S * rows,), dtype=torch.float32, device=x.device)
        yq = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if want_fp8 else None
        ys = torch.empty((S * rows,), dtype=torch.float32, device=x.device) if want_fp8 else None
        stream = L.stream_and_check(x, res, w, b, y, res_out, rstd, mean, yq, ys)
        a = L.AddNormArgs(L.ptr(x), L.ptr(res), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(res_out), L.ptr(rstd), L.ptr(mean),
                          rows, S, D, float(eps), int(is_rms), int(swap_flip), L.dtype_code(x.dtype),
                          L.dtype_code(y_dtype), L.ptr(yq), L.ptr(ys))
        L.check(L.get_lib().cad_add_norm_fwd(C.byref(a), stream), "cad_add_norm_fwd")
        ctx.save_for_backward(res_out, rstd, mean, w)
        ctx.meta = (rows, S, D, is_rms, swap_flip, x.dtype, y_dtype, residual is not None, bias is not None,
                    weight.dtype)
        if want_fp8:
            ctx.mark_non_differentiable(yq, ys)
            return y, res_out, yq, ys
        return y, res_out

    @staticmethod
    def backward(ctx, dy, dres_out, *_unused):
        res_out, rstd, mean, w = ctx.saved_tensors
        rows, S, D, is_rms, swap_flip, x_dtype, y_dtype, has_res, has_bias, wdt = ctx.meta
        dy = dy.contiguous()
        dres_out = None if dres_out is None else dres_out.contiguous()
        dx = torch.empty(res_out.shape, dtype=x_dtype, device=dy.device)
        dres_in = torch.empty(res_out.shape, dtype=torch.float32, device=dy.device) if has_res else None
        dw = torch.zeros((D,), dtype=torch.float32, device=dy.device)
        db = torch.zeros((D,), dtype=torch.float32, device=dy.device) if has_bias else None
        stream = L.stream_and_check(dy, dres_out, res_out, rstd, mean, w, dx, dres_in, dw, db)
        a = L.AddNormBwdArgs(L.ptr(dy), L.ptr(dres_out), L.ptr(res_out), L.ptr(rstd), L.ptr(mean), L.ptr(w), L.ptr(dx),
                             L.ptr(dres_in), L.ptr(dw), L.ptr(db), rows, S, D, int(is_rms), int(swap_flip),
                             L.dtype_code(x_dtype), L.dtype_code(y_dtype))
        L.check(L.get_lib().cad_add_norm_bwd(C.byref(a), stream), "cad_add_norm_bwd")
        return dx, dres_in, dw.to(wdt), (None if db is None else db.to(wdt)), None, None, None, None, None


def add_norm(x: torch.Tensor, residual: Optional[torch.Tensor], weight: torch.Tensor, bias: Optional[torch.Tensor],
             eps: float, is_rms: bool, swap_flip: bool, y_dtype: torch.dtype, want_fp8: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """x: (S, ..., D) t-frame.  Returns (normed in y_dtype, fp32 residual stream) -- rms_norm_fn(prenorm=True) for both
    strands at once, with the reference's fused-path strand swap as an index map when `swap_flip`.
    want_fp8: the caller feeds `normed` to mixer.bimamba_mixer, whose in_proj runs on the fp8 matrix cores (configs[4]) -- only those
    call sites ask for the e4m3 epilogue (the final norm_f, the un-fused wrapper and the sequence-parallel path never consume it)."""
    if swap_flip and x.shape[0] != 2:
        raise ValueError("swap_flip needs two strands")
    D = int(x.shape[-1])
    if want_fp8 and FP8_ACTIVATIONS and y_dtype == torch.bfloat16 and D % 4 == 0 and D <= 512 and _fp8_epilogue_aligned(x, residual) and \
            bool(L.get_lib().cad_proj_fp8_supported(D)):
        # its e4m3 operand (+ per-token scales) is written by this kernel's epilogue and travels with the tensor, together with what
        # identifies the contents it was made from (fp8_operand_of checks version, storage and shape before the mixer trusts it)
        y, res, yq, ys = _AddNorm.apply(x, residual, weight, bias, eps, is_rms, swap_flip, y_dtype, True)
        y._cad_fp8 = (yq, ys, y._version, y.data_ptr(), tuple(y.shape))
        return y, res
    return _AddNorm.apply(x, residual, weight, bias, eps, is_rms, swap_flip, y_dtype)


def _fp8_epilogue_aligned(x, residual) -> bool:
    """cad_add_norm_fwd writes the e4m3 copy from its 16-byte vector instantiation only (addnorm.hip: CAD_ERR_UNSUPPORTED otherwise):
    mirror that here, so that an input the scalar kernel serves keeps working with the fp8 projection switched on."""
    ok = x.is_contiguous() and x.data_ptr() % 16 == 0 and (x.shape[-1] * x.element_size()) % 16 == 0
    if residual is not None:
        ok = ok and residual.data_ptr() % 16 == 0
    return ok


def fp8_operand_of(y: torch.Tensor):
    """(e4m3 copy, per-token scales) attached by add_norm(want_fp8=True), or None unless `y` still is the tensor they were made from:
    same storage, same shape, and not written since (an in-place op between the norm and the mixer bumps the version counter)."""
    rec = getattr(y, "_cad_fp8", None)
    if rec is None or not y.is_contiguous():
        return None
    yq, ys, ver, ptr, shape = rec
    if y._version != ver or y.data_ptr() != ptr or tuple(y.shape) != shape:
        return None
    return yq, ys


# ------------------------------------------------------------------------------------------------------------------
# causal conv1d
# ------------------------------------------------------------------------------------------------------------------
class _Conv1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, split, rev_lo, rev_hi):
        x = x.contiguous()
        E, SB, Lq = x.shape
        wf = w.float().reshape(E, -1).contiguous()
        bf = None if bias is None else bias.float().contiguous()
        out = torch.empty_like(x)
        stream = L.stream_and_check(x, wf, bf, out)
        a = L.Conv1dArgs(L.ptr(x), L.ptr(wf), L.ptr(bf), L.ptr(out), SB, Lq, split, E, wf.shape[1], rev_lo, rev_hi,
                         L.dtype_code(x.dtype))
        L.check(L.get_lib().cad_conv1d_fwd(C.byref(a), stream), "cad_conv1d_fwd")
        ctx.save_for_backward(x, wf, bf)
        ctx.meta = (split, rev_lo, rev_hi, w.shape, w.dtype, bias is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wf, bf = ctx.saved_tensors
        split, rev_lo, rev_hi, wshape, wdt, has_bias = ctx.meta
        E, SB, Lq = x.shape
        dout = dout.contiguous()
        dx = torch.empty_like(x)
        dw = torch.zeros_like(wf)
        db = torch.zeros((E,), dtype=torch.float32, device=x.device) if has_bias else None
        stream = L.stream_and_check(x, wf, bf, dout, dx, dw, db)
        a = L.Conv1dBwdArgs(L.ptr(x), L.ptr(wf), L.ptr(bf), L.ptr(dout), L.ptr(dx), L.ptr(dw), L.ptr(db), SB, Lq, split,
                            E, wf.shape[1], rev_lo, rev_hi, L.dtype_code(x.dtype), 0)
        L.check(L.get_lib().cad_conv1d_bwd(C.byref(a), stream), "cad_conv1d_bwd")
        return dx, dw.reshape(wshape).to(wdt), (None if db is None else db.to(wdt)), None, None, None


def causal_conv1d(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], split: int, rev_lo: int,
                  rev_hi: int) -> torch.Tensor:
    """x: (E, SB, L) channel-major.  Depthwise causal conv + SiLU in each row's own direction."""
    return _Conv1d.apply(x, w, bias, int(split), int(rev_lo), int(rev_hi))


# ------------------------------------------------------------------------------------------------------------------
# selective scan
# ------------------------------------------------------------------------------------------------------------------
# ---- L-split (two-pass) scans on one GPU --------------------------------------------------------------------------------
# A scan launch has ceil(E / 8) * rows * sets workgroups -- 256 for Caduceus-PS at batch 1, i.e. one per CU, but only 128 for
# Caduceus-Ph or a uni-directional model at batch 1.  When fewer than ~one workgroup per CU exist,
# every row is cut into k segments along L, presented to the SAME kernels as rows (E, SB * k, L / k) of the same buffers:
#   pass 1 (cheap kernel modes `map_only` / `carry_only`): each segment's affine state map from a zero entry state,
#   composition of the k maps per row in the row's own direction (a few element-wise launches on (E, SB * k, N) tensors),
#   pass 2 (the full kernels) from the true entry states (h0 / dhT of the scan C-ABI).
# The same carries and composition chain segments ACROSS ranks in caduceus_amd/seqpar.py.
_CU_COUNT = {}


def _cu_count() -> int:
    """Compute units of the CURRENT device (256 on MI355X), queried once per device; 256 without a device (host emulator).  The
    library's launchers size their grids with the same number (csrc/api.hip: cad_cu_count)."""
    if not torch.cuda.is_available():
        return 256
    dev = torch.cuda.current_device()
    if dev not in _CU_COUNT:
        _CU_COUNT[dev] = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    return _CU_COUNT[dev]


def lsplit_factor(E: int, SB: int, Lq: int, nsets: int) -> int:
    env = os.environ.get("CADUCEUS_AMD_LSPLIT", "")
    chunk = int(L.get_lib().cad_scan_chunk_len())
    if env:  # forced (tests, A/B runs): segments only have to be whole forward chunks
        k = max(1, int(env))
        return k if Lq % (k * chunk) == 0 else 1
    ok = lambda k: Lq % (k * chunk) == 0 and Lq // k >= 8 * chunk
    wgs = ((E + 7) // 8) * SB * nsets
    k = 1
    while wgs * 2 * k <= _cu_count() and ok(2 * k):
        k *= 2
    return k


def compose_segments(P, S, k: int, split: int, rev_lo: int, rev_hi: int, towards_end: bool):
    """P, S: (E, SB * k, N) maps  v -> P * v + S  of the k segments of every row (segment q of row r is row r * k + q).
    Returns the value ENTERING every segment when the chain starts from zero at the row's logical start
    (towards_end=False: forward states) or at its logical end (towards_end=True: state gradients)."""
    E, SBk, N = S.shape
    SB = SBk // k
    P4, S4 = P.view(E, SB, k, N), S.view(E, SB, k, N)
    out = torch.zeros_like(S4)
    for rows, rev in ((slice(0, split), rev_lo), (slice(split, SB), rev_hi)):
        if rows.start >= rows.stop:
            continue
        ascending = (rev == 0) != towards_end
        order = list(range(k)) if ascending else list(range(k - 1, -1, -1))
        v = None
        for j, q in enumerate(order):
            if v is not None:
                out[:, rows, q] = v
            if j + 1 < k:
                v = S4[:, rows, q] if v is None else P4[:, rows, q] * v + S4[:, rows, q]
    return out.view(E, SBk, N)


def scan_fwd_launch(lib, args, nsets: int, stream, k: int, As, dirs, split: int):
    """cad_scan_fwd_multi on argument structs that already describe the k-way reshaped problem (SB * k rows of L / k).
    k > 1: runs pass 1 (map_only), composes the entry states and points args[i].h0 at them.  Returns (keep-alive tensors,
    [P_i]) -- P_i = exp(A * sum_dt) per segment, which the backward needs again."""
    keep, Ps = [], []
    if k > 1:
        E, SBk, N = args[0].E, args[0].SB, args[0].N
        dev = As[0].device
        p1 = (L.ScanArgs * nsets)()
        hTs, sdts = [], []
        for i in range(nsets):
            C.memmove(C.byref(p1[i]), C.byref(args[i]), C.sizeof(L.ScanArgs))
            hT = torch.empty((E, SBk, N), dtype=torch.float32, device=dev)
            sdt = torch.empty((E, SBk), dtype=torch.float32, device=dev)
            p1[i].map_only, p1[i].out, p1[i].chunk_state, p1[i].z = 1, None, None, None
            p1[i].h0, p1[i].hT, p1[i].sum_dt = None, L.ptr(hT), L.ptr(sdt)
            hTs.append(hT), sdts.append(sdt)
        L.check(lib.cad_scan_fwd_multi(p1, nsets, stream), "cad_scan_fwd_multi (map pass)")
        for i in range(nsets):
            P = torch.exp(As[i].unsqueeze(1) * sdts[i].unsqueeze(-1))
            h0 = compose_segments(P, hTs[i], k, split, dirs[i][0], dirs[i][1], towards_end=False).contiguous()
            args[i].h0 = L.ptr(h0)
            keep.append(h0)
            Ps.append(P)
    L.check(lib.cad_scan_fwd_multi(args, nsets, stream), "cad_scan_fwd_multi")
    return keep, Ps


def scan_bwd_launch(lib, args, nsets: int, stream, k: int, Ps, dirs, split: int):
    """cad_scan_bwd_multi on k-way reshaped argument structs; k > 1: pass 1 (carry_only) + composition of the state
    gradients entering every segment (args[i].dhT)."""
    keep = []
    if k > 1:
        E, SBk, N = args[0].E, args[0].SB, args[0].N
        dev = Ps[0].device
        p1 = (L.ScanBwdArgs * nsets)()
        gs = []
        for i in range(nsets):
            C.memmove(C.byref(p1[i]), C.byref(args[i]), C.sizeof(L.ScanBwdArgs))
            g = torch.empty((E, SBk, N), dtype=torch.float32, device=dev)
            p1[i].carry_only, p1[i].dhT, p1[i].dh0 = 1, None, L.ptr(g)
            p1[i].dz, p1[i].out2, p1[i].gate_fix_list, p1[i].gate_fix_count = None, None, None, None
            p1[i].fold_counters = None  # (the carry pass writes no slots)
            gs.append(g)
        L.check(lib.cad_scan_bwd_multi(p1, nsets, stream), "cad_scan_bwd_multi (carry pass)")
        for i in range(nsets):
            dhT = compose_segments(Ps[i], gs[i], k, split, dirs[i][0], dirs[i][1], towards_end=True).contiguous()
            args[i].dhT = L.ptr(dhT)
            keep.append(dhT)
    L.check(lib.cad_scan_bwd_multi(args, nsets, stream), "cad_scan_bwd_multi")
    return keep


# ---- the dB / dC fold behind the running scan backward (cad_fold_partials_stream) ----------------------------------------------------
_FOLD_SIDE = {}  # device index -> (side stream, "inputs ready" event, "fold done" event)


def fold_stream_supported(N: int, npart: int, Lq: int, dtype) -> bool:
    return bool(L.get_lib().cad_fold_stream_supported(int(N), int(npart), int(Lq), L.dtype_code(dtype)))


def _fold_side(device):
    """(side stream, event, event) whose kernels really run NEXT TO the current stream's, or None.  HIP serves all streams from a few
    hardware queues: a second stream that shares the current stream's queue runs its kernels behind it (measured under torchrun + RCCL:
    the process group's stream shifted the assignment, the fold ran after every scan, +8 ms per step).  So candidates are probed once
    per (device, current stream) with cad_stream_probe -- one wave waiting up to 2 ms on one stream for a flag set from the other --
    and the first stream that overlaps is kept; with none, the caller keeps the fold kernel behind the scan."""
    main = torch.cuda.current_stream(device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), main.cuda_stream)
    if key not in _FOLD_SIDE:
        lib = L.get_lib()
        found = None
        buf = torch.zeros((16, 2), dtype=torch.int32, device=device)
        for i in range(8):
            cand = torch.cuda.Stream(device)
            cand.wait_stream(main)  # (the zero fill above)
            L.check(lib.cad_stream_probe(C.c_void_p(main.cuda_stream), C.c_void_p(cand.cuda_stream), L.ptr(buf[i, 0:1]),
                                         L.ptr(buf[i, 1:2]), 2000), "cad_stream_probe")
            main.wait_stream(cand)
            if int(buf[i, 1].item()) == 1:  # (one host synchronisation per candidate, once per process)
                found = cand
                break
        _FOLD_SIDE[key] = None if found is None else (found, torch.cuda.Event(), torch.cuda.Event())
        if os.environ.get("CADUCEUS_AMD_VERBOSE"):
            print(f"caduceus_amd: concurrent fold stream for {key}: {'candidate %d' % i if found is not None else 'none (fold kernel behind the scan)'}")
    return _FOLD_SIDE[key]


def fold_side_available(device) -> bool:
    return (not L.is_device_build()) or _fold_side(device) is not None


FOLD_GIVE_UPS = None  # a list: every device launch appends the number of give-up records (a device scalar) of its concurrent pass


def fold_behind_scan(lib, fold_args, nsets: int, device, launch_scan, give_ups=None):
    """launch_scan() enqueues cad_scan_bwd_multi (with fold_counters set) on the current stream; the fold is enqueued on a second stream
    right behind it so that the two kernels run side by side (the fold's workgroups -- 48 VGPRs, 8 KB of LDS -- fit on the CUs next to
    the scan's), the current stream then waits for the fold and runs the cleanup pass (a no-op unless the fold gave up waiting, i.e.
    the two kernels were NOT co-scheduled).  The host emulator has no streams: the same three launches in order."""
    if not L.is_device_build():
        out = launch_scan()
        L.check(lib.cad_fold_partials_stream(fold_args, nsets, 0, None), "cad_fold_partials_stream (concurrent)")
        L.check(lib.cad_fold_partials_stream(fold_args, nsets, 1, None), "cad_fold_partials_stream (cleanup)")
        return out
    side, ev_ready, ev_done = _fold_side(device)
    main = torch.cuda.current_stream(device)
    ev_ready.record(main)  # counters zeroed, slots and destination rows allocated (and their previous users finished) before the fold starts
    out = launch_scan()
    side.wait_event(ev_ready)
    L.check(lib.cad_fold_partials_stream(fold_args, nsets, 0, C.c_void_p(side.cuda_stream)), "cad_fold_partials_stream (concurrent)")
    ev_done.record(side)
    main.wait_event(ev_done)
    if FOLD_GIVE_UPS is not None and give_ups is not None:  # diagnostics (tests, tools): slices the concurrent pass left to the cleanup
        FOLD_GIVE_UPS.append(torch.count_nonzero(give_ups))
    L.check(lib.cad_fold_partials_stream(fold_args, nsets, 1, C.c_void_p(main.cuda_stream)), "cad_fold_partials_stream (cleanup)")
    return out


def gate_fix_buffers(lib, u, N):
    """Worklist (int64 slots) + zeroed counter (int32) for the exact gate gradient at z == 0 (cad_scan_bwd_gate_fix)."""
    E, SB, Lq = u.shape
    lst = torch.empty((lib.cad_scan_gate_fix_entries(E, SB, Lq),), dtype=torch.int64, device=u.device)
    cnt = torch.zeros((1,), dtype=torch.int32, device=u.device)
    return lst, cnt


class _ScanMulti(torch.autograd.Function):
    """1 or 2 parameter sets (same shapes, shared gate z) in one launch.  Tensor args per set:
    u, delta, A, Bm, Cm, D, delta_bias."""

    @staticmethod
    def forward(ctx, z, split, dirs, delta_is_dt, *tensors):
        nsets = len(tensors) // 7
        lib = L.get_lib()
        z = None if z is None else z.contiguous()
        sets, args = [], (L.ScanArgs * nsets)()
        need_grad = any(ctx.needs_input_grad)
        outs = []
        for i in range(nsets):
            u, delta, A, Bm, Cm, D, bias = tensors[7 * i:7 * i + 7]
            u, delta, Bm, Cm = u.contiguous(), delta.contiguous(), Bm.contiguous(), Cm.contiguous()
            if delta.dtype != u.dtype or Bm.dtype != u.dtype or Cm.dtype != u.dtype or \
                    (z is not None and z.dtype != u.dtype):
                raise TypeError("selective_scan: u, delta, B, C, z must share one dtype")
            E, SB, Lq = u.shape
            N = A.shape[1]
            k = lsplit_factor(E, SB, Lq, nsets)  # same shapes in every set -> same k
            Af, Df, bf = A.float().contiguous(), D.float().contiguous(), bias.float().contiguous()
            out = torch.empty_like(u)
            state = (torch.empty((lib.cad_scan_state_floats(E, SB * k, Lq // k, N),), dtype=torch.float32, device=u.device)
                     if need_grad else None)
            stream = L.stream_and_check(u, delta, Af, Bm, Cm, Df, z, bf, out, state)
            rl, rh = dirs[i]
            args[i] = L.ScanArgs(L.ptr(u), L.ptr(delta), L.ptr(Af), L.ptr(Bm), L.ptr(Cm), L.ptr(Df), L.ptr(z), L.ptr(bf),
                                 L.ptr(out), L.ptr(state), SB * k, Lq // k, split * k, E, N, rl, rh, L.dtype_code(u.dtype))
            args[i].delta_is_dt = int(bool(delta_is_dt))
            sets.append((u, delta, Af, Bm, Cm, Df, bf, state, out))
            outs.append(out)
        _keep, Ps = scan_fwd_launch(lib, args, nsets, stream, k, [s_[2] for s_ in sets], dirs, split)
        flat = [t for s_ in sets for t in s_]
        ctx.save_for_backward(z, *flat, *Ps)
        ctx.meta = (split, dirs, nsets, [(t[2].dtype, t[5].dtype, t[6].dtype) for t in
                                         [tensors[7 * i:7 * i + 7] for i in range(nsets)]], bool(delta_is_dt), k)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *douts):
        z, *flat = ctx.saved_tensors
        split, dirs, nsets, pdt, delta_is_dt, k = ctx.meta
        Ps, flat = (flat[9 * nsets:], flat[:9 * nsets]) if k > 1 else ([], flat)
        lib = L.get_lib()
        args = (L.ScanBwdArgs * nsets)()
        keep, res = [], []
        for i in range(nsets):
            u, delta, Af, Bm, Cm, Df, bf, state, fout = flat[9 * i:9 * i + 9]
            E, SB, Lq = u.shape
            N = Af.shape[1]
            dout = douts[i].contiguous()
            du, ddelta = torch.empty_like(u), torch.empty_like(u)
            dz = None if z is None else torch.empty_like(u)
            dA, dD, dbias = torch.zeros_like(Af), torch.zeros_like(Df), torch.zeros_like(bf)
            npart = lib.cad_scan_bwd_partials(E)  # one fp32 partial-sum slot per workgroup (written, not accumulated)
            dBC = torch.empty((2, npart, N, SB, Lq), dtype=u.dtype, device=u.device)
            stream = L.stream_and_check(u, delta, Af, Bm, Cm, Df, z, bf, dout, state, du, ddelta, dz, dA, dBC, dD, dbias)
            rl, rh = dirs[i]
            fix_list, fix_cnt = gate_fix_buffers(lib, u, N) if z is not None else (None, None)
            args[i] = L.ScanBwdArgs(L.ptr(u), L.ptr(delta), L.ptr(Af), L.ptr(Bm), L.ptr(Cm), L.ptr(Df), L.ptr(z),
                                    L.ptr(bf), L.ptr(dout), L.ptr(fout), L.ptr(state), L.ptr(du), L.ptr(ddelta), L.ptr(dz),
                                    L.ptr(dA), L.ptr(dBC[0]), L.ptr(dBC[1]), L.ptr(dD), L.ptr(dbias), SB * k, Lq // k,
                                    split * k, E, N, rl, rh, L.dtype_code(u.dtype), npart, None, None, None, L.ptr(fix_list),
                                    L.ptr(fix_cnt), L.ptr(dz))
            args[i].delta_is_dt = int(delta_is_dt)
            keep.append((dout, dBC, fix_list, fix_cnt))
            res.append([du, ddelta, dA, dBC, dD, dbias, dz])
        keep.append(scan_bwd_launch(lib, args, nsets, stream, k, Ps, dirs, split))
        if z is not None:  # exact gate gradient where z == 0 (rare; the launch is a no-op otherwise)
            L.check(lib.cad_scan_bwd_gate_fix(args, nsets, stream), "cad_scan_bwd_gate_fix")
        grads = []
        dz_tot = None
        for i in range(nsets):
            du, ddelta, dA, dBC, dD, dbias, dz = res[i]
            u = flat[9 * i]
            n = dBC[0, 0].numel()
            npart = dBC.shape[1]
            dB, dC = torch.empty(dBC.shape[2:], dtype=u.dtype, device=u.device), \
                torch.empty(dBC.shape[2:], dtype=u.dtype, device=u.device)
            for src, dst in ((dBC[0], dB), (dBC[1], dC)):
                L.check(lib.cad_reduce_partials(L.ptr(src), npart, n, L.ptr(dst), L.dtype_code(u.dtype), stream),
                        "cad_reduce_partials")
            Adt, Ddt, bdt = pdt[i]
            grads += [du, ddelta, dA.to(Adt), dB, dC, dD.to(Ddt), dbias.to(bdt)]
            if dz is not None:
                dz_tot = dz if dz_tot is None else dz_tot.add_(dz)
        return (dz_tot, None, None, None, *grads)


def selective_scan_multi(sets, z, split: int, dirs, delta_is_dt: bool = False):
    """sets: list of (u, delta, A, Bm, Cm, D, delta_bias) with u, delta: (E, SB, L); A: (E, N) (= -exp(A_log));
    Bm, Cm: (N, SB, L); D, delta_bias: (E).  z: shared gate (E, SB, L) or None.  dirs: [(rev_lo, rev_hi)] per set.
    mamba_ssm `selective_scan_fn(..., delta_softplus=True)` for each set, each row in its own direction.
    delta_is_dt: `delta` already holds dt = softplus(delta_raw + delta_bias) (proj_wx(..., softplus_bias=)); its gradient is
    still the one w.r.t. delta_raw, and delta_bias still receives its gradient."""
    flat = [t for s_ in sets for t in s_]
    return _ScanMulti.apply(z, int(split), tuple((int(a), int(b)) for a, b in dirs), bool(delta_is_dt), *flat)


def selective_scan(u, delta, A, Bm, Cm, D, z, delta_bias, split: int, rev_lo: int, rev_hi: int) -> torch.Tensor:
    """Single parameter set (see selective_scan_multi)."""
    return selective_scan_multi([(u, delta, A, Bm, Cm, D, delta_bias)], z, split, [(rev_lo, rev_hi)])[0]


class _ScanStateful(torch.autograd.Function):
    """One parameter set over a row SEGMENT: (u, delta, A, Bm, Cm, D, z, delta_bias, h0) -> (out, hT).  h0 / hT are the
    (E, SB, N) fp32 states entering / leaving the segment along each row's direction; both are differentiable, so
    consecutive segments chain through plain autograd (chunk-pipelined scans over sequences that do not fit at once, and
    the building block of caduceus_amd/seqpar.py)."""

    @staticmethod
    def forward(ctx, u, delta, A, Bm, Cm, D, z, bias, h0, split, rev_lo, rev_hi):
        lib = L.get_lib()
        u, delta, Bm, Cm = u.contiguous(), delta.contiguous(), Bm.contiguous(), Cm.contiguous()
        z = None if z is None else z.contiguous()
        E, SB, Lq = u.shape
        N = A.shape[1]
        Af, Df, bf = A.float().contiguous(), D.float().contiguous(), bias.float().contiguous()
        h0f = None if h0 is None else h0.float().contiguous()
        out = torch.empty_like(u)
        hT = torch.empty((E, SB, N), dtype=torch.float32, device=u.device)
        state = torch.empty((lib.cad_scan_state_floats(E, SB, Lq, N),), dtype=torch.float32, device=u.device)
        stream = L.stream_and_check(u, delta, Af, Bm, Cm, Df, z, bf, out, state, h0f, hT)
        a = L.ScanArgs(L.ptr(u), L.ptr(delta), L.ptr(Af), L.ptr(Bm), L.ptr(Cm), L.ptr(Df), L.ptr(z), L.ptr(bf), L.ptr(out),
                       L.ptr(state), SB, Lq, split, E, N, rev_lo, rev_hi, L.dtype_code(u.dtype), L.ptr(h0f), L.ptr(hT), None)
        L.check(lib.cad_scan_fwd(C.byref(a), stream), "cad_scan_fwd")
        ctx.save_for_backward(u, delta, Af, Bm, Cm, Df, z, bf, state, out)
        ctx.meta = (split, rev_lo, rev_hi, A.dtype, D.dtype, bias.dtype, h0 is not None)
        ctx.mark_non_differentiable()
        return out, hT

    @staticmethod
    def backward(ctx, dout, dhT):
        lib = L.get_lib()
        u, delta, Af, Bm, Cm, Df, z, bf, state, fout = ctx.saved_tensors
        split, rev_lo, rev_hi, Adt, Ddt, bdt, has_h0 = ctx.meta
        E, SB, Lq = u.shape
        N = Af.shape[1]
        dout = torch.zeros_like(u) if dout is None else dout.contiguous()
        dhT = None if dhT is None else dhT.float().contiguous()
        du, ddelta = torch.empty_like(u), torch.empty_like(u)
        dz = None if z is None else torch.empty_like(u)
        dA, dD, dbias = torch.zeros_like(Af), torch.zeros_like(Df), torch.zeros_like(bf)
        npart = lib.cad_scan_bwd_partials(E)
        dBC = torch.empty((2, npart, N, SB, Lq), dtype=u.dtype, device=u.device)
        dh0 = torch.empty((E, SB, N), dtype=torch.float32, device=u.device)
        stream = L.stream_and_check(u, delta, Af, Bm, Cm, Df, z, bf, dout, state, du, ddelta, dz, dA, dBC, dD, dbias, dhT, dh0)
        fix_list, fix_cnt = gate_fix_buffers(lib, u, N) if z is not None else (None, None)
        a = L.ScanBwdArgs(L.ptr(u), L.ptr(delta), L.ptr(Af), L.ptr(Bm), L.ptr(Cm), L.ptr(Df), L.ptr(z), L.ptr(bf),
                          L.ptr(dout), L.ptr(fout), L.ptr(state), L.ptr(du), L.ptr(ddelta), L.ptr(dz), L.ptr(dA),
                          L.ptr(dBC[0]), L.ptr(dBC[1]), L.ptr(dD), L.ptr(dbias), SB, Lq, split, E, N, rev_lo, rev_hi,
                          L.dtype_code(u.dtype), npart, L.ptr(dhT), L.ptr(dh0), None, L.ptr(fix_list), L.ptr(fix_cnt),
                          L.ptr(dz))
        L.check(lib.cad_scan_bwd(C.byref(a), stream), "cad_scan_bwd")
        if z is not None:
            L.check(lib.cad_scan_bwd_gate_fix(C.byref(a), 1, stream), "cad_scan_bwd_gate_fix")
        n = dBC[0, 0].numel()
        dB, dC = torch.empty(dBC.shape[2:], dtype=u.dtype, device=u.device), \
            torch.empty(dBC.shape[2:], dtype=u.dtype, device=u.device)
        for src, dst in ((dBC[0], dB), (dBC[1], dC)):
            L.check(lib.cad_reduce_partials(L.ptr(src), npart, n, L.ptr(dst), L.dtype_code(u.dtype), stream),
                    "cad_reduce_partials")
        return (du, ddelta, dA.to(Adt), dB, dC, dD.to(Ddt), dz, dbias.to(bdt), dh0 if has_h0 else None, None, None, None)


def selective_scan_stateful(u, delta, A, Bm, Cm, D, z, delta_bias, h0, split: int, rev_lo: int, rev_hi: int):
    """Segment scan with state carries: returns (out, hT).  See _ScanStateful."""
    return _ScanStateful.apply(u, delta, A, Bm, Cm, D, z, delta_bias, h0, int(split), int(rev_lo), int(rev_hi))


# ------------------------------------------------------------------------------------------------------------------
# LM head (+ masked cross entropy)
# ------------------------------------------------------------------------------------------------------------------
class _LmHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hidden, weight, comp, labels, ignore_index):
        hidden = hidden.contiguous()
        S, D = hidden.shape[0], hidden.shape[-1]
        rows = hidden.numel() // (S * D)
        w = weight.float().contiguous()
        V = w.shape[0]
        logits = torch.empty(hidden.shape[1:-1] + (V,), dtype=torch.float32, device=hidden.device)
        lab = None if labels is None else labels.reshape(-1).long().contiguous()
        acc = torch.zeros((2,), dtype=torch.float32, device=hidden.device)
        # deterministic loss: per-workgroup partial sums folded in a fixed order (no fp32 atomics)
        parts = (torch.empty((L.get_lib().cad_lm_head_partials(rows),), dtype=torch.float32, device=hidden.device)
                 if labels is not None else None)
        stream = L.stream_and_check(hidden, w, comp, lab, logits, acc, parts)
        a = L.LmHeadArgs(L.ptr(hidden), L.ptr(w), L.ptr(comp), L.ptr(lab), L.ptr(logits), C.c_void_p(acc.data_ptr()),
                         C.c_void_p(acc.data_ptr() + 4), rows, D, V, S, int(ignore_index), L.dtype_code(hidden.dtype),
                         L.ptr(parts))
        L.check(L.get_lib().cad_lm_head_fwd(C.byref(a), stream), "cad_lm_head_fwd")
        loss = acc[0] / acc[1] if labels is not None else acc[0]
        ctx.save_for_backward(hidden, w, comp, lab, logits, acc)
        ctx.meta = (ignore_index, weight.dtype, labels is not None)
        # an output nobody differentiated reaches backward as None instead of a zero-filled (rows, V) fp32 tensor the kernel would
        # allocate and read (training uses the loss alone)
        ctx.set_materialize_grads(False)
        return logits, loss

    @staticmethod
    def backward(ctx, dlogits, dloss):
        hidden, w, comp, lab, logits, acc = ctx.saved_tensors
        ignore_index, wdt, has_labels = ctx.meta
        S, D = hidden.shape[0], hidden.shape[-1]
        V = w.shape[0]
        lib = L.get_lib()
        use_loss = has_labels and dloss is not None
        # channel block of the kernel: the whole row (d_model 128 / 256) or 256 channels of a wider head (d_model 512: two launches,
        # the channels are independent in both products)
        DB = D if lib.cad_lm_head_bwd_supported(int(D), int(V)) else (256 if D % 256 == 0 and lib.cad_lm_head_bwd_supported(256, int(V)) else 0)
        if (use_loss or dlogits is not None) and hidden.dtype in (torch.float32, torch.bfloat16) and DB and \
                hidden.data_ptr() % 16 == 0:  # (kernel: 16-byte vector accesses)
            # on the matrix cores: softmax gradient, d hidden of both strands, dW partial slots (cad_lm_head_bwd)
            rows = hidden.numel() // (S * D)
            dh = torch.empty_like(hidden)
            parts = torch.empty((lib.cad_lm_head_bwd_partials(rows), V, D), dtype=torch.float32, device=hidden.device)
            coef = (dloss.float() / acc[1]).reshape(1) if use_loss else None
            dlg = None if dlogits is None else dlogits.reshape(-1, V).float().contiguous()
            stream = L.stream_and_check(hidden, w, comp, lab if use_loss else None, logits, dlg, coef, dh, parts)
            for c0 in range(0, D, DB):
                es = hidden.element_size()
                a = L.LmHeadBwdArgs(hidden.data_ptr() + c0 * es, w.data_ptr() + c0 * 4, L.ptr(comp), L.ptr(lab) if use_loss else None,
                                    L.ptr(logits), L.ptr(dlg), L.ptr(coef), dh.data_ptr() + c0 * es, parts.data_ptr() + c0 * 4, rows, DB, V, S,
                                    int(ignore_index), L.dtype_code(hidden.dtype), 0 if DB == D else D)
                L.check(lib.cad_lm_head_bwd(C.byref(a), stream), "cad_lm_head_bwd")
            return dh, parts.sum(dim=0).to(wdt), None, None, None
        # (other shapes: torch ops)
        # d loss / d logits = (softmax - onehot) * valid / count, assembled WITHOUT boolean-mask indexing: `sm[rows[valid], lab[valid]]`
        # goes through nonzero(), i.e. a device-to-host copy in the middle of the backward -- the launch queue ran dry behind it
        # (~0.4 ms of idle gaps per step in the kernel trace)
        g = None if dlogits is None else dlogits.reshape(-1, V).float()
        if has_labels and dloss is not None:
            valid = ((lab != ignore_index) & (lab >= 0) & (lab < V)).to(torch.float32).unsqueeze(1)  # as the forward counts
            sm = torch.softmax(logits.reshape(-1, V), dim=-1)
            sm.scatter_add_(1, lab.clamp(0, V - 1).unsqueeze(1), -valid)  # rows that do not count add -0 somewhere
            sm.mul_(valid * (dloss / acc[1]))
            g = sm if g is None else g + sm
        if g is None:
            g = torch.zeros((logits.numel() // V, V), dtype=torch.float32, device=hidden.device)
        gt = g.to(hidden.dtype)
        h = hidden.reshape(S, -1, D)
        dh = torch.empty_like(h)
        wt = w.to(hidden.dtype)
        dh[0] = mm(gt, wt)
        dw = mm(gt.t(), h[0]).float()
        if S == 2:
            dh[1] = mm(gt, wt[comp])
            d2 = mm(gt.t(), h[1]).float()
            dw.index_add_(0, comp, d2)
        return dh.reshape(hidden.shape), dw.to(wdt), None, None, None


def lm_head(hidden: torch.Tensor, weight: torch.Tensor, comp: Optional[torch.Tensor],
            labels: Optional[torch.Tensor] = None, ignore_index: int = -100):
    """hidden: (S, B, L, D) t-frame -> (fp32 logits (B, L, V), loss or None).  RCPSLMHead + cross_entropy."""
    logits, loss = _LmHead.apply(hidden, weight, comp, labels, ignore_index)
    return logits, (loss if labels is not None else None)


# ------------------------------------------------------------------------------------------------------------------
# dense projections on the matrix cores (raw ops, no autograd: caduceus_amd/mixer.py schedules their gradients by hand)
# ------------------------------------------------------------------------------------------------------------------
def mm_f32(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, addend: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (M, N) fp32 = [addend +] a (M, K) @ b (K, N) on the fp32 matrix core (cad_gemm_f32) -- torch.mm / torch.addmm for the fp32 path
    without a library GEMM.  a / b: 2-D fp32 views with ONE unit stride each (plain or transposed, any row pitch: the channel-major and
    token-major views of the mixer are taken as they are); out: any 2-D fp32 view (default: a new contiguous tensor); addend: out's
    strides (or out itself).  A product with a small result and a long reduction (a weight gradient over all tokens: 1024 x 256 from
    K = 262144) is cut into K slices -- one launch, fp32 partial tiles, one fp32 sum -- for the same two reasons as on the bf16 path:
    16 output tiles cannot fill 256 CUs, and 64 partial sums of 4096 terms round better than one chain of 262144.  The batched form is
    bmm_f32."""
    if a.dim() != 2 or b.dim() != 2:
        raise ValueError("mm_f32: 2-D operands")
    M, K = a.shape
    N = b.shape[1]
    n = _f32_kslices(M, N, K) if b.shape[0] == K else 1  # (a shape mismatch is reported by _gemm_f32)
    if n > 1:
        Kc = K // n
        part = _gemm_f32(a.unflatten(1, (n, Kc)).permute(1, 0, 2), b.unflatten(0, (n, Kc)), None, None)  # (n, M, N)
        res = torch.empty((M, N), dtype=torch.float32, device=a.device)
        if (M * N) % 4 == 0:
            fold_f32([(part, res, M * N, n, M * N, 1, 0)])
        else:
            torch.sum(part, dim=0, out=res)
        if addend is not None:
            res = res + addend
        if out is None:
            return res
        out.copy_(res)
        return out
    return _gemm_f32(a.unsqueeze(0), b.unsqueeze(0), None if out is None else out.unsqueeze(0),
                     None if addend is None else addend.unsqueeze(0))[0]


def _f32_kslices(M: int, N: int, K: int) -> int:
    """K slices of an fp32 product (1: none), chosen with the launcher's two tile configurations in mind (csrc/gemm_f32.hip): a result of
    at least eight 128 x 128 tiles is cut until two such tiles per CU exist (slices of >= 1024 terms) and runs on the large
    configuration; a smaller one is cut until its 64 x 64 tiles fill the chip (slices of >= 256 terms)."""
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    if M > 64 and N > 64 and t128 >= 8:
        tiles, want, least = t128, 2 * _cu_count(), 1024
    else:
        tiles, want, least = ((M + 63) // 64) * ((N + 63) // 64), _cu_count(), 256
    n = 1
    while tiles * n < want and n < 64 and K % (2 * n) == 0 and K // (2 * n) >= least:
        n *= 2
    return n


def bmm_f32(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (n, M, N) fp32 = a (n, M, K) @ b (n, K, N), one launch (cad_gemm_f32, blockIdx.z = batch); a / b: 3-D fp32 views, one unit
    stride among the two inner dimensions of each."""
    return _gemm_f32(a, b, out, None)


def _gemm_f32(a, b, out, addend):
    if a.dtype != torch.float32 or b.dtype != torch.float32 or a.dim() != 3 or b.dim() != 3:
        raise ValueError("gemm_f32: fp32 operands, (n, M, K) @ (n, K, N)")
    n, M, K = a.shape
    N = b.shape[2]
    if b.shape[0] != n or b.shape[1] != K:
        raise ValueError(f"gemm_f32: shapes {tuple(a.shape)} @ {tuple(b.shape)}")

    def strides(t):
        """(row, column) element strides the kernel needs: one of them 1.  The stride of a size-1 dimension is never used: report it as
        the unit one when the other is not; a matrix with two non-unit strides is copied."""
        rs, cs = t.stride(1), t.stride(2)
        if t.shape[1] == 1 and cs != 1:
            rs = 1
        elif t.shape[2] == 1 and rs != 1:
            cs = 1
        if rs != 1 and cs != 1:
            t = t.contiguous()
            rs, cs = t.stride(1), t.stride(2)
            if cs != 1:  # (n, M, 1) contiguous: strides (M, 1, 1) -- cannot happen; kept as a guard
                raise ValueError("gemm_f32: operand without a unit stride")
        return t, rs, cs

    a, a_rs, a_cs = strides(a)
    b, b_rs, b_cs = strides(b)
    if out is None:
        out = torch.empty((n, M, N), dtype=torch.float32, device=a.device)
    if out.dtype != torch.float32 or tuple(out.shape) != (n, M, N):
        raise ValueError("gemm_f32: out must be fp32 (n, M, N)")
    if addend is not None and (addend.dtype != torch.float32 or addend.shape != out.shape or addend.stride() != out.stride()):
        raise ValueError("gemm_f32: addend must have out's dtype, shape and strides")
    if M == 0 or N == 0:
        return out
    if K == 0:
        if addend is None:
            out.zero_()
        elif addend.data_ptr() != out.data_ptr():
            out.copy_(addend)
        return out
    stream = L.stream_and_check(a, b, out, contiguous=False)
    bs = lambda t: t.stride(0) if n > 1 else 0
    args = L.GemmF32Args(L.ptr(a), L.ptr(b), L.ptr(out), None if addend is None else L.ptr(addend), M, N, K,
                         a_rs, a_cs, b_rs, b_cs, max(out.stride(1), 1), max(out.stride(2), 1), n, bs(a), bs(b), bs(out))
    L.check(L.get_lib().cad_gemm_f32(C.byref(args), stream), "cad_gemm_f32")
    return out


class _MmF32(torch.autograd.Function):
    """a @ b (+ addend) in fp32 on cad_gemm_f32, differentiable: da = g @ b^T, db = a^T @ g through the same kernel (transposed views)."""

    @staticmethod
    def forward(ctx, a, b, addend):
        ctx.save_for_backward(a, b)
        return mm_f32(a, b, addend=addend)

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = mm_f32(g, b.t()) if ctx.needs_input_grad[0] else None
        db = mm_f32(a.t(), g) if ctx.needs_input_grad[1] else None
        return da, db, (g if ctx.needs_input_grad[2] else None)


def _own_f32(*ts) -> bool:
    return all(t.dtype == torch.float32 for t in ts)


def mm(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """torch.mm for the call sites that have no dedicated kernel: fp32 operands take the own fp32 matrix-core kernel (cad_gemm_f32;
    differentiable), anything else (bf16 shapes the MFMA projection kernels do not serve) the library."""
    if _own_f32(a, b):
        if out is not None:
            return mm_f32(a, b, out=out)
        return _MmF32.apply(a, b, None)
    return torch.mm(a, b) if out is None else torch.mm(a, b, out=out)


def addmm(acc: torch.Tensor, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """acc + a @ b (torch.addmm), fp32 on the own kernel."""
    if _own_f32(acc, a, b):
        return _MmF32.apply(a, b, acc)
    return torch.addmm(acc, a, b)


def bmm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.bmm (no autograd use in this package), fp32 on the own kernel."""
    if _own_f32(a, b):
        return bmm_f32(a, b)
    return torch.bmm(a, b)


def proj_supported(t: torch.Tensor, K: int) -> bool:
    """The MFMA projection kernels take bf16 operands with a supported reduction length.  (K = 512, d_model 512: stand-alone and cold the
    library GEMM is 18-25 % faster than the W-stationary kernel, inside the training step it is not -- 558.9 vs 555.9 ms per step;
    profiles/r04_proj_d512.txt.  The own kernel stays.)"""
    return t.dtype == torch.bfloat16 and bool(L.get_lib().cad_proj_supported(int(K)))


def proj_wxT(W: torch.Tensor, X: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (M, T) channel-major = W (M, K) @ X (T, K)^T, bf16 in / fp32 accumulate / bf16 out (cad_proj_wxT)."""
    M, K = W.shape
    T = X.shape[0]
    if X.shape[1] != K or W.stride(1) != 1 or X.stride(1) != 1:
        raise ValueError("proj_wxT: W (M, K) and X (T, K) with unit inner stride")
    if out is None:
        out = torch.empty((M, T), dtype=torch.bfloat16, device=X.device)
    stream = L.stream_and_check(W, X, out, contiguous=False)
    a = L.ProjArgs(L.ptr(W), L.ptr(X), L.ptr(out), T, M, K, W.stride(0), X.stride(0), out.stride(0), None, 0)
    L.check(L.get_lib().cad_proj_wxT(C.byref(a), stream), "cad_proj_wxT")
    return out


def proj_wx_supported(t: torch.Tensor, K: int, T: int, M: Optional[int] = None) -> bool:
    """thin K (K <= 64, any M, optional addend), or -- when M is given -- thin M / deep K (M <= 64, K % 64 == 0, no addend)"""
    if t.dtype != torch.bfloat16:
        return False
    lib = L.get_lib()
    if lib.cad_proj_wx_supported(int(K), int(T)):
        return True
    return M is not None and bool(lib.cad_proj_wx_thin_supported(int(M), int(K), int(T)))


def proj_wx(W: torch.Tensor, X: torch.Tensor, out: Optional[torch.Tensor] = None,
            acc: Optional[torch.Tensor] = None, softplus_bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (M, T) = W (M, K) @ X (K, T) [+ acc (M, T)], all channel-major bf16, thin K (cad_proj_wx).  `acc` may be `out`
    itself (in-place accumulate).  softplus_bias (M) fp32: out = softplus(W @ X + bias) evaluated in fp32 (thin K only)."""
    M, K = W.shape
    T = X.shape[1]
    if X.shape[0] != K or W.stride(1) != 1 or X.stride(1) != 1:
        raise ValueError("proj_wx: W (M, K) and X (K, T) with unit inner stride")
    if out is None:
        out = torch.empty((M, T), dtype=torch.bfloat16, device=X.device)
    stream = L.stream_and_check(W, X, out, acc, softplus_bias, contiguous=False)
    if softplus_bias is not None and (softplus_bias.dtype != torch.float32 or softplus_bias.numel() != M or
                                      not softplus_bias.is_contiguous()):
        raise ValueError("proj_wx: softplus_bias must be a contiguous fp32 vector of M elements")
    a = L.ProjArgs(L.ptr(W), L.ptr(X), L.ptr(out), T, M, K, W.stride(0), X.stride(0), out.stride(0), L.ptr(acc),
                   0 if acc is None else acc.stride(0), L.ptr(softplus_bias), 0 if softplus_bias is None else 1)
    L.check(L.get_lib().cad_proj_wx(C.byref(a), stream), "cad_proj_wx")
    return out


# ------------------------------------------------------------------------------------------------------------------
# fp8 (OCP e4m3) in_proj: BASELINE configs[4] "fp8 MFMA projections"
# ------------------------------------------------------------------------------------------------------------------
FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


def proj_xTw_supported(X: torch.Tensor, M: int, K: int, T: int) -> bool:
    return X.dtype == torch.bfloat16 and bool(L.get_lib().cad_proj_xTw_supported(int(M), int(K), int(T)))


def proj_xTw(W: torch.Tensor, X: torch.Tensor, X2: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (T, M) token-major = X (K, T)^T @ W (M, K)^T [+ X2 (K, T)^T @ W^T], X / X2 channel-major bf16 (cad_proj_xTw): out_proj on the
    sum of the two directions' scan outputs through ONE set of resident weight fragments, fp32 accumulation over both panels."""
    M, K = W.shape
    T = X.shape[1]
    if out is None:
        out = torch.empty((T, M), dtype=X.dtype, device=X.device)
    stream = L.stream_and_check(W, X, X2, out, contiguous=False)
    assert W.stride(1) == 1 and X.stride(1) == 1 and out.stride(1) == 1 and (X2 is None or (X2.stride() == X.stride() and X2.shape == X.shape))
    a = L.ProjTmArgs(L.ptr(W), L.ptr(X), L.ptr(X2), L.ptr(out), T, M, K, W.stride(0), X.stride(0), out.stride(0))
    L.check(L.get_lib().cad_proj_xTw(C.byref(a), stream), "cad_proj_xTw")
    return out


GEMM_PARTIALS, GEMM_OUT_T_BF16 = 0, 1


def gemm_stream_slices(R: int, Cc: int, K: int) -> int:
    """K slices of a weight gradient: as many as fill the GPU with one 256 x 256 tile per workgroup (cad_gemm_stream), 0 = unsupported."""
    if R % 256 or Cc % 256 or R < 256 or Cc < 256:
        return 0
    n = max(1, _cu_count() // ((R // 256) * (Cc // 256)))
    while n > 1 and (K % (32 * n) != 0):
        n -= 1
    return n if K % (32 * n) == 0 else 0


def fold_f32(jobs) -> None:
    """cad_fold_f32_multi: jobs = [(src, dst, n, nparts, stride, nparts2, stride2)], src / dst fp32 tensors (src addressed from its data
    pointer: dst[i] = sum_{j < nparts2, k < nparts} src[j * stride2 + k * stride + i]) -- every sum in ONE launch."""
    arr = (L.FoldF32Job * len(jobs))()
    tensors = []
    for q, (src, dst, n, nparts, stride, nparts2, stride2) in enumerate(jobs):
        if src.dtype != torch.float32 or dst.dtype != torch.float32 or not dst.is_contiguous() or dst.numel() != n:
            raise ValueError("fold_f32: fp32 tensors, dst contiguous with n elements")
        arr[q] = L.FoldF32Job(L.ptr(src), L.ptr(dst), n, stride, stride2, nparts, nparts2)
        tensors += [src, dst]
    stream = L.stream_and_check(*tensors, contiguous=False)
    L.check(L.get_lib().cad_fold_f32_multi(arr, len(jobs), stream), "cad_fold_f32_multi")


def wgrad_cm_tm(a_cm: torch.Tensor, b_tm: torch.Tensor, return_partials: bool = False) -> Optional[torch.Tensor]:
    """(M, N) fp32 = a (M, T) channel-major @ b (T, N) token-major, bf16 operands, fp32 accumulation over ALL tokens (cad_gemm_stream,
    CAD_GEMM_PARTIALS: one fp32 tile per K slice, summed here): dW_in and dW_out of the mixer.  None if the shape is not served.
    return_partials: the (slices, M, N) partial tiles as they are (the caller folds them, e.g. with fold_f32 next to other sums)."""
    M, T = a_cm.shape
    N = b_tm.shape[1]
    if a_cm.dtype != torch.bfloat16 or b_tm.dtype != torch.bfloat16 or a_cm.stride(1) != 1 or b_tm.stride(1) != 1:
        return None
    if a_cm.stride(0) % 8 or b_tm.stride(0) % 8 or a_cm.data_ptr() % 16 or b_tm.data_ptr() % 16:
        return None
    n = gemm_stream_slices(M, N, T)
    if n == 0 or not L.get_lib().cad_gemm_stream_supported(M, N, T, n):
        return None
    part = torch.empty((n, M, N), dtype=torch.float32, device=a_cm.device)
    stream = L.stream_and_check(a_cm, b_tm, part, contiguous=False)
    a = L.GemmStreamArgs(L.ptr(a_cm), L.ptr(b_tm), L.ptr(part), M, N, T, a_cm.stride(0), b_tm.stride(0), 0, n, GEMM_PARTIALS)
    L.check(L.get_lib().cad_gemm_stream(C.byref(a), stream), "cad_gemm_stream")
    if return_partials:
        return part
    return part[0] if n == 1 else part.sum(0)


def gemm_out_t(A: torch.Tensor, B: torch.Tensor, col_fastest: bool = True) -> Optional[torch.Tensor]:
    """out (C, R) bf16 = (A (R, K) @ B (K, C))^T, both operands row-major bf16 and streamed, fp32 accumulation (cad_gemm_stream,
    CAD_GEMM_OUT_T_BF16).  With A = token-major activations (T, d_model) and B = W^T (d_model, M) this is a projection with channel-major
    output (M, T): the in_proj and d(y) of the mixer at d_model 512 (configs[4]), where it replaces the W-stationary cad_proj_wxT
    (which streams X once per 128-row block of W).  col_fastest: neighbouring workgroups share an A tile.  None if the shape is not served."""
    return proj_xTw_stream(A, B, col_fastest=col_fastest)


def proj_xTw_stream(Wt: torch.Tensor, X: torch.Tensor, col_fastest: bool = False) -> Optional[torch.Tensor]:
    """out (T, M) token-major bf16 = X (K, T)^T @ Wt (M, K)^T with X channel-major and BOTH operands streamed (cad_gemm_stream,
    CAD_GEMM_OUT_T_BF16): d(x2d) = dxz^T W_in with Wt = W_in^T (D, 2E), K = 2E too deep for resident weight fragments.  None if the
    shape is not served."""
    M, K = Wt.shape
    T = X.shape[1]
    if Wt.dtype != torch.bfloat16 or X.dtype != torch.bfloat16 or Wt.stride(1) != 1 or X.stride(1) != 1:
        return None
    if Wt.stride(0) % 8 or X.stride(0) % 8 or Wt.data_ptr() % 16 or X.data_ptr() % 16:
        return None
    if not L.get_lib().cad_gemm_stream_supported(M, T, K, 1):
        return None
    out = torch.empty((T, M), dtype=torch.bfloat16, device=X.device)
    stream = L.stream_and_check(Wt, X, out, contiguous=False)
    a = L.GemmStreamArgs(L.ptr(Wt), L.ptr(X), L.ptr(out), M, T, K, Wt.stride(0), X.stride(0), out.stride(0), 1, GEMM_OUT_T_BF16,
                         int(bool(col_fastest)))
    L.check(L.get_lib().cad_gemm_stream(C.byref(a), stream), "cad_gemm_stream")
    return out


def fp8_proj_supported(t: torch.Tensor, K: int) -> bool:
    return t.dtype in (torch.bfloat16, torch.float32) and bool(L.get_lib().cad_proj_fp8_supported(int(K)))


def quant_rows_fp8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """x (T, K) fp32 / bf16 -> (q (T, K) e4m3 bytes as uint8, scale (T) fp32) with x ~ q * scale[:, None]: one scale per token
    (cad_quant_rows_fp8), so a token's quantisation does not depend on its position."""
    T, K = x.shape
    if x.stride(1) != 1:
        raise ValueError("quant_rows_fp8: unit inner stride")
    q = torch.empty((T, K), dtype=torch.uint8, device=x.device)
    scale = torch.empty((T,), dtype=torch.float32, device=x.device)
    stream = L.stream_and_check(x, q, scale, contiguous=False)
    a = L.QuantFp8Args(L.ptr(x), L.ptr(q), L.ptr(scale), T, K, x.stride(0), q.stride(0), L.dtype_code(x.dtype))
    L.check(L.get_lib().cad_quant_rows_fp8(C.byref(a), stream), "cad_quant_rows_fp8")
    return q, scale


def quant_weight_fp8(W: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """W (M, K) -> (e4m3 bytes as uint8, per-row scale (M) fp32).  Weights change once per optimizer step: plain torch ops."""
    w = W.detach().float()
    s = (w.abs().amax(dim=1) / FP8_MAX).clamp_min(1e-30)
    q = (w / s[:, None]).clamp_(-FP8_MAX, FP8_MAX).to(FP8).view(torch.uint8)
    return q.contiguous(), s.contiguous()


def proj_wxT_fp8(Wq: torch.Tensor, sw: torch.Tensor, Xq: torch.Tensor, sx: torch.Tensor,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out (M, T) channel-major bf16 = (Wq (M, K) @ Xq (T, K)^T) * sw[:, None] * sx[None, :] on the fp8 matrix cores
    (cad_proj_wxT_fp8); Wq / Xq: e4m3 bytes (uint8 tensors), sw / sx: fp32 scales."""
    M, K = Wq.shape
    T = Xq.shape[0]
    if Xq.shape[1] != K or Wq.dtype != torch.uint8 or Xq.dtype != torch.uint8 or Wq.stride(1) != 1 or Xq.stride(1) != 1:
        raise ValueError("proj_wxT_fp8: Wq (M, K), Xq (T, K) uint8 (e4m3 bytes) with unit inner stride")
    if sw.dtype != torch.float32 or sx.dtype != torch.float32 or sw.numel() != M or sx.numel() != T:
        raise ValueError("proj_wxT_fp8: sw (M), sx (T) fp32")
    if out is None:
        out = torch.empty((M, T), dtype=torch.bfloat16, device=Xq.device)
    stream = L.stream_and_check(Wq, Xq, sw, sx, out, contiguous=False)
    a = L.ProjFp8Args(L.ptr(Wq), L.ptr(Xq), L.ptr(sw), L.ptr(sx), L.ptr(out), T, M, K, Wq.stride(0), Xq.stride(0),
                      out.stride(0))
    L.check(L.get_lib().cad_proj_wxT_fp8(C.byref(a), stream), "cad_proj_wxT_fp8")
    return out


def proj_wx_wgrad_supported(X: torch.Tensor, M: int, K: int, T: int) -> bool:
    return X.dtype == torch.bfloat16 and bool(L.get_lib().cad_proj_wx_wgrad_supported(int(M), int(K), int(T)))


def wgrad_partials(T: int, K: int, M: int, device, nsets: int = 1) -> torch.Tensor:
    """(nsets, P, K, M) fp32 partial slots of cad_proj_wx_wgrad, one (K, M) slot per workgroup: a caller with several parameter sets
    hands each launch its own [i] and folds all of them with ONE sum over dim 1 afterwards."""
    return torch.empty((nsets, L.get_lib().cad_proj_wx_wgrad_partials(T), K, M), dtype=torch.float32, device=device)


def proj_wx_wgrad(W: torch.Tensor, X: torch.Tensor, Y: torch.Tensor, out: Optional[torch.Tensor] = None,
                  part: Optional[torch.Tensor] = None):
    """(out (M, T) = W (M, K) @ X (K, T),  dW (K, M) fp32 = X (K, T) @ Y (M, T)^T) from ONE pass over X (cad_proj_wx_wgrad):
    d(dt_lr) = W_dt^T d(delta) together with dW_dt = d(delta) dt_lr^T.  All operands channel-major bf16.
    part: a (P, K, M) slice of wgrad_partials() -- then the partial slots are left un-summed (returns (out, None))."""
    M, K = W.shape
    T = X.shape[1]
    if X.shape[0] != K or Y.shape != (M, T) or W.stride(1) != 1 or X.stride(1) != 1 or Y.stride(1) != 1:
        raise ValueError("proj_wx_wgrad: W (M, K), X (K, T), Y (M, T) with unit inner stride")
    lib = L.get_lib()
    if out is None:
        out = torch.empty((M, T), dtype=torch.bfloat16, device=X.device)
    own = part is None
    if own:
        part = wgrad_partials(T, K, M, X.device)[0]
    stream = L.stream_and_check(W, X, Y, out, part, contiguous=False)
    a = L.ProjArgs(L.ptr(W), L.ptr(X), L.ptr(out), T, M, K, W.stride(0), X.stride(0), out.stride(0), None, 0, None, 0,
                   L.ptr(Y), Y.stride(0), L.ptr(part))
    L.check(lib.cad_proj_wx_wgrad(C.byref(a), stream), "cad_proj_wx_wgrad")
    return out, (part.sum(dim=0) if own else None)


def proj_wgrad_only_supported(X: torch.Tensor, M: int, K: int, T: int) -> bool:
    return X.dtype == torch.bfloat16 and bool(L.get_lib().cad_proj_wgrad_only_supported(int(M), int(K), int(T)))


def proj_wgrad_only(X: torch.Tensor, Y: torch.Tensor, part: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
    """dW (M, K) fp32 = Y (M, T) @ X (K, T)^T, both channel-major bf16, M <= 64: the weight-gradient stage of cad_proj_wx_wgrad
    alone (W == NULL) -- dW_x = d(dbc) . xc^T of the x_proj backward.  part: as for proj_wx_wgrad (slots hold the (K, M) transpose)."""
    K, T = X.shape
    M = Y.shape[0]
    if Y.shape[1] != T or X.stride(1) != 1 or Y.stride(1) != 1:
        raise ValueError("proj_wgrad_only: X (K, T), Y (M, T) with unit inner stride")
    lib = L.get_lib()
    own = part is None
    if own:
        part = wgrad_partials(T, K, M, X.device)[0]
    stream = L.stream_and_check(X, Y, part, contiguous=False)
    a = L.ProjArgs(None, L.ptr(X), None, T, M, K, 0, X.stride(0), 0, None, 0, None, 0, L.ptr(Y), Y.stride(0), L.ptr(part))
    L.check(lib.cad_proj_wx_wgrad(C.byref(a), stream), "cad_proj_wx_wgrad")
    return part.sum(dim=0).t().contiguous() if own else None
