/* caduceus_hip.h  --  C-ABI of libcaduceus_hip.so: the MI355X (gfx950) kernels behind the Caduceus hot path.
 *
 * Boundary rules:
 *   - plain C, raw DEVICE pointers + sizes + a hipStream_t passed as void*; no torch / C++ types;
 *   - the caller (PyTorch's allocator, or any other runtime) owns every buffer; kernels never allocate;
 *   - every entry point returns an int status: 0 = ok, non-zero = cad_status (caller raises);
 *   - launches are asynchronous on the given stream, re-entrant, no global mutable state except the opt-in
 *     profiler (cad_prof_*).
 *
 * Each entry point states which interface of the reference it replaces.  The reference itself is pure Python
 * (the .py files under /root/reference/caduceus/) and reaches native code only through the third-party, un-vendored packages
 * mamba-ssm==1.2.0.post1 / causal-conv1d==1.2.0.post2 (/root/reference/caduceus_env.yml:46-50); the cited
 * file:line are the reference's call sites of those.
 *
 * Layout conventions
 *   "t-frame"  : the RCPS stream (B, L, 2*D) of the reference is held as n_strands=2 separate D-wide strands,
 *                tensor (S, B, L, D) row-major, strand 1 stored with its CHANNELS REVERSED
 *                (t2[c] = ref[..., 2D-1-c]).  In this frame both strands use every weight in natural order and
 *                differ only in scan direction, so all reference flips/cats become index maps.
 *                Caduceus-Ph has S=1.
 *   rows "SB"  : S*B independent sequences ("rows").  Rows [0, split) run in direction rev_lo, rows
 *                [split, SB) in direction rev_hi (0 = left-to-right, 1 = right-to-left along L).
 *   channel-major activations: (E, SB, L) row-major, i.e. L contiguous -- the layout the in_proj GEMM writes
 *                and the scan reads with coalesced loads along L.
 */
#ifndef CADUCEUS_HIP_H
#define CADUCEUS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { CAD_OK = 0, CAD_ERR_BAD_ARG = 1, CAD_ERR_UNSUPPORTED = 2, CAD_ERR_LAUNCH = 3 } cad_status;
typedef enum { CAD_F32 = 0, CAD_BF16 = 1 } cad_dtype;

/* Library identification / error text (static strings). */
const char* cad_version(void);
const char* cad_status_string(int status);
/* 1 when built for the GPU (hipcc, gfx950); 0 for the host-emulator build used only by the test-suite. */
int cad_is_device_build(void);

/* ---------------------------------------------------------------------------------------------------------
 * RCPS / plain embedding gather.   Replaces RCPSEmbedding.forward/.rc (modeling_rcps.py:46-67) and
 * nn.Embedding (modeling_caduceus.py:157).
 *   out[0][b][l][:] = W[ids[b][l]][:]                       strand 0
 *   out[1][b][l][:] = W[comp[ids[b][l]]][:]                 strand 1 (t-frame: the reference's two flipL
 *                                                            cancel and its flipC is the storage convention)
 * Integer index path is exact.  ids: int64 (B, L).  comp: int64 (V) device, NULL when n_strands == 1.
 * W: (V, D) in w_dtype.  out: (n_strands, B, L, D) in out_dtype.
 * Returns CAD_ERR_BAD_ARG if an id is outside [0, V) (checked on device, reported via *err_flag if given). */
typedef struct {
    const int64_t* ids;
    const int64_t* comp;
    const void* weight;
    void* out;
    int64_t B, L;
    int D, V, n_strands;
    int w_dtype, out_dtype;
} cad_embed_args;
int cad_embed_fwd(const cad_embed_args* a, void* stream);
/* dW (V, D) fp32 is ACCUMULATED into (caller zeroes):  dW[ids] += dout[0], dW[comp[ids]] += dout[1]. */
typedef struct {
    const int64_t* ids;
    const int64_t* comp;
    const void* dout;
    float* dweight;
    int64_t B, L;
    int D, V, n_strands;
    int dout_dtype;
} cad_embed_bwd_args;
int cad_embed_bwd(const cad_embed_bwd_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Fused residual-add + RMSNorm / LayerNorm over both strands.   Replaces mamba_ssm rms_norm_fn / layer_norm_fn
 * as called by RCPSMambaBlock.forward (modeling_rcps.py:170-197), RCPSAddNormWrapper (:107-130),
 * mamba_ssm Block (modeling_caduceus.py:64) and the final norm (modeling_caduceus.py:233-273).
 *   sum = x (+ residual_in)        fp32
 *   y   = norm(sum) * w (+ b)      fp32 statistics
 * x, y: (S, R, D) with R = rows_per_strand = B*L tokens.   swap_flip = 1 reproduces the reference's FUSED-path
 * quirk (modeling_rcps.py:177-197: the "fwd" norm is fed the second half, the "rc" norm the first): in the
 * t-frame the outputs go to the OTHER strand with channels reversed:
 *   y[1-s][r][D-1-c], residual_out[1-s][r][D-1-c]  <-  token (s, r), channel c.
 * swap_flip = 0 is the un-fused / final-norm / Caduceus-Ph behaviour (identity map).
 * residual_in may be NULL (first layer).  residual_out (fp32) and rstd (S*R floats) [+ mean for LayerNorm] are
 * always written; they are what the backward needs.  weight/bias fp32 (bias NULL for RMSNorm). */
typedef struct {
    const void* x;
    const float* residual_in;
    const float* weight;
    const float* bias;
    void* y;
    float* residual_out;
    float* rstd;
    float* mean;
    int64_t rows_per_strand;
    int n_strands, D;
    float eps;
    int is_rms, swap_flip;
    int x_dtype, y_dtype;
    /* optional (NULL = absent): an e4m3 copy of y with one scale per output row, y ~ y_fp8 * y_scale[row] -- the operand of the fp8
     * in_proj (cad_proj_wxT_fp8) written by the kernel that produces the normed activations instead of a separate
     * cad_quant_rows_fp8 pass over them (to which it is bit-identical).  y_fp8: (S, R, D) bytes, y_scale: (S * R) fp32, both in the
     * OUTPUT index space.  D % 4 == 0, D <= 512, 16-byte aligned tensors. */
    void* y_fp8;
    float* y_scale;
} cad_add_norm_args;
int cad_add_norm_fwd(const cad_add_norm_args* a, void* stream);
/* Backward.  dy (y_dtype) and dres_out (fp32, may be NULL) are in the OUTPUT index space, sum_saved/rstd/mean
 * as written by the forward (sum_saved = residual_out, output index space).  Writes dx (x_dtype) and, if not
 * NULL, dres_in (fp32) in the INPUT index space; ACCUMULATES dweight/dbias (fp32, caller zeroes). */
typedef struct {
    const void* dy;
    const float* dres_out;
    const float* sum_saved;
    const float* rstd;
    const float* mean;
    const float* weight;
    void* dx;
    float* dres_in;
    float* dweight;
    float* dbias;
    int64_t rows_per_strand;
    int n_strands, D;
    int is_rms, swap_flip;
    int x_dtype, y_dtype;
} cad_add_norm_bwd_args;
int cad_add_norm_bwd(const cad_add_norm_bwd_args* a, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Depthwise causal conv1d (+bias, +SiLU) along L, channel-major, with per-row direction.
 * Replaces causal_conv1d_cuda.causal_conv1d_fwd/bwd reached through mamba_ssm.Mamba.forward
 * (modeling_caduceus.py:128,130).  In logical (direction-mapped) coordinates:
 *   out[p] = silu(bias + sum_k w[k] * x[p - (K-1) + k]),  zero padding before the logical start.
 * A right-to-left row is therefore the reference's flipL -> conv -> flipL without moving data.
 * x, out: (E, SB, L) in dtype.  w: (E, K) fp32, K in [1, 4].  bias: (E) fp32 or NULL. */
typedef struct {
    const void* x;
    const float* w;
    const float* bias;
    void* out;
    int64_t SB, L, split;
    int E, K;
    int rev_lo, rev_hi;
    int dtype;
} cad_conv1d_args;
int cad_conv1d_fwd(const cad_conv1d_args* a, void* stream);
/* nsets (1 or 2) parameter sets reading the SAME x (mamba_fwd / mamba_rev of a BiMamba layer see one in_proj output in
 * opposite directions, modeling_caduceus.py:128-130): x is read once. */
int cad_conv1d_fwd_multi(const cad_conv1d_args* sets, int nsets, void* stream);
/* dx is WRITTEN (dtype), or ADDED to when accumulate != 0 (the second parameter set of a BiMamba layer adds its input
 * gradient onto the first one's); dw (E,K) and dbias (E) fp32 are ACCUMULATED (caller zeroes). */
typedef struct {
    const void* x;
    const float* w;
    const float* bias;
    const void* dout;
    void* dx;
    float* dw;
    float* dbias;
    int64_t SB, L, split;
    int E, K;
    int rev_lo, rev_hi;
    int dtype;
    int accumulate;
} cad_conv1d_bwd_args;
int cad_conv1d_bwd(const cad_conv1d_bwd_args* a, void* stream);
/* nsets (1 or 2) parameter sets with the same x, dx and accumulate flag: dx = (accumulate ? dx : 0) + sum over the sets. */
int cad_conv1d_bwd_multi(const cad_conv1d_bwd_args* sets, int nsets, void* stream);

/* ---------------------------------------------------------------------------------------------------------
 * Selective SSM scan, one direction per row.   Replaces selective_scan_cuda.fwd / .bwd reached through
 * mamba_ssm.Mamba.forward -> mamba_inner_fn (modeling_caduceus.py:128,130):
 *   dt   = softplus(delta + delta_bias)                           fp32
 *   h_p  = exp(dt_p * A) * h_{p-1} + dt_p * B_p * u_p             fp32 state (E x N), h_{-1} = 0
 *   y_p  = <C_p, h_p> + D * u_p ;   out_p = y_p * silu(z_p)       (z NULL -> no gate)
 * p is the logical position; physical l = p (left-to-right rows) or L-1-p (right-to-left rows).  The
 * right-to-left variant is an exact mirror (same chunk boundaries counted from the logical start, same
 * floating-point operation order), which is what keeps RC-equivariance bit-exact.
 * u, delta, z, out: (E, SB, L) dtype.  A: (E, N) fp32 (= -exp(A_log)).  Bm, Cm: (N, SB, L) dtype.
 * D, delta_bias: (E) fp32.  chunk_state: fp32 buffer of cad_scan_state_floats() elements (the running state at
 * every chunk start, needed by the backward), or NULL for inference.
 * Optional state carries, (E, SB, N) fp32, NULL = absent: h0 = state entering the row's first logical position (default 0),
 * hT = state after its last position (written); sum_dt (E, SB) = sum of dt over the row (written).  They make a row
 * a segment of a longer sequence (sequence-parallel / chunk-pipelined scans: caduceus_amd/seqpar.py). */
typedef struct {
    const void* u;
    const void* delta;
    const float* A;
    const void* Bm;
    const void* Cm;
    const float* D;
    const void* z;
    const float* delta_bias;
    void* out;
    float* chunk_state;
    int64_t SB, L, split;
    int E, N;
    int rev_lo, rev_hi;
    int dtype;
    const float* h0;
    float* hT;
    float* sum_dt;
    int delta_is_dt;   /* 1: `delta` already holds dt = softplus(delta_raw + delta_bias) (written by cad_proj_wx with
                        * act = CAD_ACT_SOFTPLUS_BIAS); delta_bias is ignored */
    int map_only;      /* 1: only the row's affine state map is wanted -- hT (from h0, default 0) and sum_dt are written, `out`
                        * and chunk_state are not touched (may be NULL), C / D / z are not read.  Pass 1 of an L-split scan:
                        * the segments of a row are presented as rows (E, SB * k, L / k) of the same buffers. */
} cad_scan_args;
int cad_scan_fwd(const cad_scan_args* a, void* stream);
/* Same, for nsets (1 or 2) independent parameter sets of identical shape in ONE launch -- the mamba_fwd and mamba_rev
 * scans of a BiMamba layer (modeling_caduceus.py:128-130) -- so that a CU holds two waves per SIMD even at batch 1. */
int cad_scan_fwd_multi(const cad_scan_args* sets, int nsets, void* stream);
int64_t cad_scan_chunk_len(void);
int64_t cad_scan_state_floats(int E, int64_t SB, int64_t L, int N);
/* Backward.  du, ddelta, dz are WRITTEN (dtype).  dA (E,N), dD (E), ddelta_bias (E) are ACCUMULATED with a few fp32
 * atomics per channel (caller zeroes).  dB, dC: n_partials = cad_scan_bwd_partials(E) slots of (N,SB,L) each, in the
 * activation dtype (fp32 mode: fp32 slots; bf16 mode: bf16 slots, each holding an fp32-accumulated 8-channel sum);
 * slot k is WRITTEN (plain coalesced stores, no atomics, no zeroing needed) with the sum over the channels of workgroup
 * k; cad_reduce_partials folds the slots (fp32 accumulation) into the final (N,SB,L) gradient.
 * chunk_state: as written by the forward.  out: the forward's (gated) output, required when z != NULL: the gate gradient
 * uses y = out / silu(z) instead of re-accumulating y.  Where z == 0 EXACTLY that quotient is 0/0: the kernel writes 0
 * there and, when gate_fix_list / gate_fix_count are given, records the (channel, row, chunk) in the list;
 * cad_scan_bwd_gate_fix (same argument structs, called after cad_scan_bwd[_multi] on the same stream) recomputes y for
 * the recorded chunks and adds the exact dout * y / 2 into gate_fix_dz (= dz, or the dz buffer of the set that shares the
 * gate).  gate_fix_list: cad_scan_gate_fix_entries(E, SB, L) int64 slots; gate_fix_count: one int32 the caller zeroes.
 * Without them dz stays 0 where z == 0.
 * Optional carries (E, SB, N) fp32: dhT = gradient w.r.t. the forward's hT (default 0), dh0 = gradient w.r.t. h0 (written).
 * Shared gate (BiMamba: the forward and the reverse scan are gated by the same z and receive the same dout): pass the
 * other scan's gated output as out2 and dz receives the gate gradient of BOTH scans (one fp32 evaluation, one rounding);
 * the other parameter set of the launch then passes dz = NULL (z is still required there: it gates dout). */
typedef struct {
    const void* u;
    const void* delta;
    const float* A;
    const void* Bm;
    const void* Cm;
    const float* D;
    const void* z;
    const float* delta_bias;
    const void* dout;
    const void* out;
    const float* chunk_state;
    void* du;
    void* ddelta;
    void* dz;
    float* dA;
    void* dB;
    void* dC;
    float* dD;
    float* ddelta_bias;
    int64_t SB, L, split;
    int E, N;
    int rev_lo, rev_hi;
    int dtype;
    int n_partials;
    const float* dhT;
    float* dh0;
    const void* out2;
    int64_t* gate_fix_list;
    int* gate_fix_count;
    void* gate_fix_dz;
    int delta_is_dt;   /* as in cad_scan_args; ddelta / ddelta_bias are still the gradients w.r.t. delta_raw / the bias:
                        * d(dt) * sigmoid(delta_raw + bias) = d(dt) * (1 - exp(-dt)) */
    int carry_only;    /* 1: only dh0 (from dhT, default 0) is wanted -- the reverse recurrence of the state gradient alone
                        * (exp, C * dy, one chain per item and state); no other output is written, B / chunk_state / out are
                        * not read (du, ddelta, dA, dB, dC, dD, ddelta_bias, chunk_state may be NULL).  Pass 1 of an L-split
                        * backward (see map_only). */
    int* fold_counters; /* optional (NULL = absent): cad_scan_bwd_fold_counter_ints(SB, L) int32, zeroed by the caller (chunk arrivals
                        * [row][chunk], ONE count of the workgroups that have started, one mark per CU hosting one).  When given, the
                        * dB / dC slot stores go out write-through and every workgroup adds 1 to counter [row][chunk] once ALL its
                        * slot stores of that 512-position chunk have left the CU, so that cad_fold_partials_stream -- launched on ANOTHER
                        * stream while this kernel runs -- can fold chunk by chunk behind it (a chunk is complete at
                        * n_partials arrivals).  Ignored by a carry_only pass. */
} cad_scan_bwd_args;
int cad_scan_bwd(const cad_scan_bwd_args* a, void* stream);
int cad_scan_bwd_multi(const cad_scan_bwd_args* sets, int nsets, void* stream);
int cad_scan_bwd_gate_fix(const cad_scan_bwd_args* sets, int nsets, void* stream);
int64_t cad_scan_gate_fix_entries(int E, int64_t SB, int64_t L);
int cad_scan_bwd_partials(int E);
/* dst[i] = sum_k src[k*n + i], k < n_partials (fp32 accumulation); src and dst in dtype (fp32 or bf16). */
int cad_reduce_partials(const void* src, int n_partials, int64_t n, void* dst, int dtype, void* stream);
/* The same fold for up to CAD_REDUCE_MAX_JOBS (src, dst) pairs of equal depth, length and dtype in ONE launch -- the dB and dC slots of
 * both parameter sets of a BiMamba layer (four folds per layer; same summation order as the single fold: bit-identical results). */
#define CAD_REDUCE_MAX_JOBS 4
typedef struct {
    const void* src;
    void* dst;
} cad_reduce_job;
int cad_reduce_partials_multi(const cad_reduce_job* jobs, int njobs, int n_partials, int64_t n, int dtype, void* stream);
/* Summation order of every fold of this library (cad_reduce_partials, _multi, cad_fold_partials_stream): the slots are summed in groups of
 * CAD_FOLD_GROUP consecutive slots (left to right inside a group, from 0), the group sums are added left to right -- one fixed order, so a
 * gradient does not depend on WHICH of the three folded it.
 *
 * cad_fold_partials_stream: the same fold of the dB / dC slots of a scan backward, chunk by chunk WHILE cad_scan_bwd_multi still runs (the
 * reference reduces dB / dC over the channels inside selective_scan_cuda.bwd itself, modeling_caduceus.py:11,128,130).  The scan
 * backward is bound by VALU issue and leaves ~85 % of the memory bandwidth idle; its partial slots (2.15 GB per configs[2] layer) used
 * to be re-read by a fold kernel AFTER it (0.39 ms per layer).  Launched on a second stream right after the scan, this kernel's
 * workgroups (one per slice of a chunk: 256 threads, < 64 VGPRs, 8 KB of LDS -- they fit on the CUs next to the scan's) poll the
 * scan's fold_counters and fold a chunk as soon as all n_partials producers have published it: the fold ends a few microseconds after
 * the scan.  Visibility across the non-coherent per-XCD L2s follows the write-through recipe: sc1 slot stores, every storing wave
 * drains (s_waitcnt vmcnt(0)), workgroup barrier, one relaxed agent-scope counter increment; the consumer polls with one lane
 * (relaxed agent-scope load), workgroup barrier, sc1 loads.
 * mode CAD_FOLD_CONCURRENT: poll, with a bounded wait -- a workgroup whose chunk does not complete within the budget records (chunk + 1) in
 * abort_from[row][slice] and returns (never a hang: if the two kernels are not co-scheduled the fold simply runs after the scan, when
 * every counter is complete).  mode CAD_FOLD_CLEANUP (same stream as the consumers, after both kernels): folds what a concurrent
 * launch left (abort_from > 0), without polling; returns at once otherwise.  mode CAD_FOLD_ALL: no polling, everything (test / fallback).
 * bf16, L % 512 == 0, n_partials a power of two in 8 .. 2048 with 2 N 512 / n_partials >= 8 (cad_fold_stream_supported). */
#define CAD_FOLD_GROUP 8
#define CAD_FOLD_CONCURRENT 0
#define CAD_FOLD_CLEANUP 1
#define CAD_FOLD_ALL 2
typedef struct {
    const void* dB_slots;   /* (n_partials, N, SB, L) as written by cad_scan_bwd_multi */
    const void* dC_slots;
    void* dB;               /* (N, SB, L) */
    void* dC;
    const int* counters;    /* the scan's fold_counters (CAD_FOLD_CONCURRENT): cad_scan_bwd_fold_counter_ints(SB, L) ints */
    int* abort_from;        /* (SB, n_partials) int32, zeroed by the caller: CAD_FOLD_CONCURRENT stores (chunk + 1) where a slice gave up,
                             * CAD_FOLD_CLEANUP folds chunks chunk .. 0 of such a slice and clears the entry; 0 = nothing left */
    int64_t SB, L, split;
    int N, n_partials;
    int rev_lo, rev_hi;
    int dtype;
} cad_fold_args;
int cad_fold_partials_stream(const cad_fold_args* sets, int nsets, int mode, void* stream);
int cad_fold_stream_supported(int N, int n_partials, int64_t L, int dtype);
int64_t cad_scan_bwd_chunk_len(void);
/* Do kernels of `stream_b` run WHILE a kernel of `stream_a` runs?  HIP maps streams onto a few hardware queues; two streams on one queue
 * are served in order, and a fold launched on such a stream simply runs after the scan (correct, but exposed and slower than the
 * streaming fold kernel).  One wave on stream_a waits (bounded: budget_us) for a flag that one wave on stream_b sets; result[0] = 1 if it
 * saw the flag, 0 if it ran out of time.  flag, result: device int32, zeroed by the caller; the caller synchronises and reads result. */
int cad_stream_probe(void* stream_a, void* stream_b, int* flag, int* result, int64_t budget_us);
int64_t cad_scan_bwd_fold_counter_ints(int64_t SB, int64_t L);

/* ---------------------------------------------------------------------------------------------------------
 * Dense projections of the mixer on the matrix cores (bf16 MFMA, fp32 accumulation).   Replace the `in_proj` /
 * `out_proj` nn.Linear calls inside mamba_ssm.Mamba.forward -> mamba_inner_fn reached from modeling_caduceus.py:128,130
 * (weights tied across directions at :114-118), in the channel-major / token-major layouts of this library.
 * cad_proj_wxT:  out (M, T) channel-major = W (M, K) . X (T, K)^T     -- in_proj (W = in_proj.weight, X = normed tokens)
 *                                                                      and d(y) = W_out^T . dout^T of the backward.
 * All operands bf16, row strides ld* in ELEMENTS (multiples of 8); K must satisfy cad_proj_supported(K); any M, T.
 * The per-token result does not depend on the token's position (fixed reduction order): t-frame strands / directions
 * get bit-identical projections. */
typedef struct {
    const void* W;
    const void* X;
    void* out;
    int64_t T;
    int M, K;
    int64_t ldw, ldx, ldo;
    const void* acc;   /* cad_proj_wx only: optional (M, T) addend, may alias out; NULL = none */
    int64_t ldacc;
    const float* bias; /* cad_proj_wx, K <= 64, no addend: per-row bias (M) of the activation below, or NULL */
    int act;           /* 0 = none; CAD_ACT_SOFTPLUS_BIAS: out = softplus(W . X + bias) evaluated in fp32 -- dt_proj + delta_bias +
                        * softplus of mamba_inner_fn / selective_scan_fn(delta_softplus=True) in one pass (the scans then take
                        * delta_is_dt = 1) */
    /* cad_proj_wx_wgrad only: the second operand Y (M, T) of the weight gradient and the partial-sum buffer (see below) */
    const void* wg_y;
    int64_t ld_wg_y;
    float* wg_partials;
} cad_proj_args;
#define CAD_ACT_SOFTPLUS_BIAS 1
int cad_proj_wxT(const cad_proj_args* a, void* stream);
int cad_proj_supported(int K);
/* cad_proj_wx:  out (M, T) = W (M, K) . X (K, T) [+ acc],  all channel-major, thin K (cad_proj_wx_supported: K <= 64,
 * K % 8 == 0, T % 8 == 0) -- dt_proj (K = dt_rank; `dt_proj` inside mamba_inner_fn) and the x_proj input gradient
 * d(xc) = du + W_x^T . d(dbc) of its backward.  X is token-contiguous: the MFMA fragments are transposed on the way out
 * of LDS (ds_read_b64_tr_b16). */
int cad_proj_wx(const cad_proj_args* a, void* stream);
int cad_proj_wx_supported(int K, int64_t T);
/* cad_proj_wx also takes thin M / deep K products (M <= 64, K a multiple of 64 up to 1024, T % 8 == 0; ldo % 4; W (M x K) and the
 * ring of X tiles must fit the 160 KB of LDS): x_proj (M = dt_rank + 2 d_state, K = d_inner; `x_proj` inside mamba_inner_fn) and
 * d(dt_lr) = W_dt^T . d(delta).  Here `acc` (bf16, ldacc % 4, may alias out) is added to the fp32 sums before the one rounding: a product too
 * deep for one W copy in LDS (x_proj at d_inner 1024: 64 x 1024) is run as two K halves, the second with the first as its addend. */
int cad_proj_wx_thin_supported(int M, int K, int64_t T);
/* cad_proj_wx_wgrad: the thin-M / deep-K product  out (M, T) = W (M, K) . X (K, T)  AND, from the same single pass over X, the weight
 * gradient  dW (K, M) = X (K, T) . Y (M, T)^T  -- d(dt_lr) = W_dt^T . d(delta) together with dW_dt = d(delta) . dt_lr^T of the dt_proj
 * backward (both stream the (d_inner, T) tensor d(delta); mamba_inner_fn's backward reached from modeling_caduceus.py:128,130).
 * M in {16, 32}, K in {256, 512} (cad_proj_wx_wgrad_supported), T >= 128 and T % 128 == 0 (whole 128-token blocks: the kernel has no
 * tail block), bf16.  wg_partials: cad_proj_wx_wgrad_partials(T) slots of
 * (K, M) fp32, one per workgroup, WRITTEN (no zeroing needed); dW = the sum over the slots (fixed order: deterministic). */
int cad_proj_wx_wgrad(const cad_proj_args* a, void* stream);
int cad_proj_wx_wgrad_supported(int M, int K, int64_t T);
/* With W == NULL and out == NULL the same entry point computes the weight gradient alone, for any M <= 64
 * (cad_proj_wgrad_only_supported; the same T >= 128, T % 128 == 0): dW_x = xc . d(dbc)^T of the x_proj backward
 * (M = dt_rank + 2 d_state). */
int cad_proj_wgrad_only_supported(int M, int K, int64_t T);
int cad_proj_wx_wgrad_partials(int64_t T);

/* cad_proj_xTw:  out (T, M) TOKEN-major = X (K, T)^T . W (M, K)^T  [+ X2 (K, T)^T . W (M, K)^T],  X / X2 channel-major, all bf16 --
 * the `out_proj` nn.Linear of mamba_inner_fn on the sum of the two directions' scan outputs (modeling_caduceus.py:128-138: with
 * bidirectional_strategy "add" and tied weights, out_proj(y_f) + out_proj(y_r) = W_out (y_f + y_r)): both panels go through the same
 * resident weight fragments, fp32 accumulation over both, one rounding.  X2 == NULL: a single panel.
 * M in {128, 256} (= d_model), K in {256, 512} (= d_inner), T % 8 == 0 (cad_proj_xTw_supported); ldw / ldx multiples of 8, ldo of 4. */
typedef struct {
    const void* W;
    const void* X;
    const void* X2;
    void* out;
    int64_t T;
    int M, K;
    int64_t ldw, ldx, ldo;
} cad_proj_tm_args;
int cad_proj_xTw(const cad_proj_tm_args* a, void* stream);
int cad_proj_xTw_supported(int M, int K, int64_t T);

/* ---------------------------------------------------------------------------------------------------------
 * cad_gemm_stream -- the dense products of the mixer's backward whose two operands BOTH stream (no operand fits a CU's registers):
 * the in_proj input gradient d(x2d) and the two weight gradients dW_in / dW_out of mamba_ssm.Mamba as run by
 * modeling_caduceus.py:128,130 (the backward of `xz = in_proj(hidden)` / `out = out_proj(y)`; torch.mm / K-split bmm -> hipBLASLt
 * until round 4).  One kernel, bf16 operands, fp32 accumulation on v_mfma_f32_16x16x32_bf16:
 *     D (R x C) = A (R x K) . B (K x C),   A row-major [r][k] (lda elements between rows, k contiguous),
 *                                          B row-major [k][c] (ldb elements between rows, c contiguous)
 * mode CAD_GEMM_PARTIALS: k in [0, K) is cut into `nslices` equal slices; slice s of tile (r, c) is one workgroup and writes its fp32
 *   tile to partials[s][R][C] (the caller sums the slices in fp32: a weight gradient over all tokens, K = T);
 * mode CAD_GEMM_OUT_T_BF16: out (C x R) bf16 = D^T, rows ldo elements apart (token-major d(x2d): A = W_in^T (D x 2E), B = dxz
 *   (2E x T) channel-major, C = T; nslices must be 1).
 *   Round 5: with K = d_model 512 the same mode is the in_proj itself and d(y) of configs[4] -- A = the token-major activations
 *   (T x d_model), B = W_in^T (d_model x 2E) resp. W_out (d_model x E), out = the channel-major (2E | E) x T result -- where the
 *   W-stationary cad_proj_wxT streams X once per 128-row block of W (sixteen times at M = 2048).
 * R, C multiples of 256, K / nslices a multiple of 32 (cad_gemm_stream_supported); 16-byte aligned operands, lda / ldb % 8 == 0.
 * col_fastest: order of the 256 x 256 tiles over the workgroups -- 0: row tiles fastest (neighbouring workgroups share the B rows of a
 *   slice: the weight gradients, d(x2d)), 1: column tiles fastest (neighbouring workgroups share an A tile: A is the big streamed
 *   operand and all of B stays cache-resident, the in_proj / d(y) use).  Placement only; the results are the same. */
#define CAD_GEMM_PARTIALS 0
#define CAD_GEMM_OUT_T_BF16 1
typedef struct {
    const void* A;
    const void* B;
    void* out;      /* fp32 partials (nslices, R, C) or bf16 (C, ldo) */
    int64_t R, C, K;
    int64_t lda, ldb, ldo;
    int nslices;
    int mode;
    int col_fastest;
} cad_gemm_stream_args;
int cad_gemm_stream(const cad_gemm_stream_args* a, void* stream);

/* cad_fold_f32_multi: up to CAD_FOLD_F32_MAX_JOBS sums of fp32 partial tiles in ONE launch -- the per-workgroup / per-K-slice partials of the
 * weight-gradient kernels of a mixer layer's backward (cad_gemm_stream's CAD_GEMM_PARTIALS slices, cad_proj_wx_wgrad's slots), which were
 * one torch reduction each (the autograd glue around mamba_inner_fn's weight gradients, modeling_caduceus.py:11,128,130 has no such step:
 * cuBLAS sums inside its GEMMs).  dst[i] = sum over j < nparts2, k < nparts of src[j * stride2 + k * stride + i], i < n, summed in that
 * fixed order (k fastest): deterministic.  nparts2 = 1 for a plain fold; the second level adds, e.g., the two halves of the tied
 * out_proj's weight gradient.  n % 4 == 0, strides % 4 == 0, src / dst 16-byte aligned. */
#define CAD_FOLD_F32_MAX_JOBS 8
typedef struct {
    const float* src;
    float* dst;
    int64_t n;
    int64_t stride, stride2;
    int nparts, nparts2;
} cad_fold_f32_job;
int cad_fold_f32_multi(const cad_fold_f32_job* jobs, int njobs, void* stream);

/* cad_gemm_f32: D (M x N) = [addend +] A (M x K) . B (K x N), fp32 operands, fp32 accumulation, fp32 result, on the fp32 matrix core
 * (v_mfma_f32_16x16x4_f32): the dense projections of the fp32 path -- the F.linear / matmul calls of mamba_inner_fn
 * (/root/reference/caduceus/modeling_caduceus.py:11,128,130: in_proj, x_proj, dt_proj, out_proj and their gradients) when the model runs in
 * fp32 (BASELINE configs[0]; the reference's fp16-AMP activations are computed by the same fp32 kernels here), which went through torch.mm
 * (hipBLASLt) until round 6.  Element (i, j) of an operand X is X[i * x_rs + j * x_cs] (elements; for A and B one of the two strides must
 * be 1), so transposed and channel-major views need no copy.  `batch` independent products, operands / results *_bs elements apart (the
 * K slices of a weight gradient over all tokens: partial tiles, summed by the caller in fp32).  addend (optional, may be D itself) has D's
 * layout.  Any M, N, K >= 1. */
typedef struct {
    const float* A;
    const float* B;
    float* D;
    const float* addend;
    int64_t M, N, K;
    int64_t a_rs, a_cs, b_rs, b_cs, d_rs, d_cs;
    int64_t batch, a_bs, b_bs, d_bs;
} cad_gemm_f32_args;
int cad_gemm_f32(const cad_gemm_f32_args* a, void* stream);
int cad_gemm_stream_supported(int64_t R, int64_t C, int64_t K, int nslices);

/* ---------------------------------------------------------------------------------------------------------
 * fp8 (OCP e4m3) projections -- BASELINE configs[4] "fp8 MFMA projections": the same `in_proj` nn.Linear call of
 * mamba_ssm.Mamba.forward (modeling_caduceus.py:128,130) on v_mfma_f32_16x16x32_fp8_fp8 with fp32 accumulation.
 * cad_quant_rows_fp8:  q (T, K) e4m3 = x (T, K) / scale[t],  scale[t] = max|x[t, :]| / 448 (1 for an all-zero row), one
 *   scale per TOKEN, so a token's quantisation -- hence its projection -- does not depend on its position or its batch
 *   (RC-equivariance stays exact).  x in `dtype` (fp32 / bf16), rows contiguous (ldx elements apart).
 * cad_proj_wxT_fp8:    out (M, T) channel-major bf16 = (Wq (M, K) . Xq (T, K)^T) * sw[m] * sx[t]
 *   Wq / Xq e4m3 (rows ldw / ldx BYTES apart, multiples of 16), sw (M) / sx (T) fp32 de-quantisation scales.
 *   K in {256, 512} (cad_proj_fp8_supported); any M, T. */
typedef struct {
    const void* x;
    void* q;
    float* scale;
    int64_t T;
    int K;
    int64_t ldx, ldq;
    int dtype;
} cad_quant_fp8_args;
int cad_quant_rows_fp8(const cad_quant_fp8_args* a, void* stream);
typedef struct {
    const void* Wq;
    const void* Xq;
    const float* sw;
    const float* sx;
    void* out;
    int64_t T;
    int M, K;
    int64_t ldw, ldx, ldo;
} cad_proj_fp8_args;
int cad_proj_wxT_fp8(const cad_proj_fp8_args* a, void* stream);
int cad_proj_fp8_supported(int K);

/* ---------------------------------------------------------------------------------------------------------
 * RCPS LM head + cross-entropy.   Replaces RCPSLMHead.forward (modeling_rcps.py:233-246), logits.float()
 * (modeling_caduceus.py:475) and cross_entropy(ignore_index) (modeling_caduceus.py:279-283,
 * src/tasks/metrics.py:181-184).  t-frame:  logits[b,l,v] = <W[v], t1[b,l]> + <W[comp[v]], t2[b,l]>.
 * hidden: (S, B*L, D) dtype.  W: (V, D) fp32.  logits: (B*L, V) fp32 (always written).
 * If labels != NULL: loss_sum[0] += sum over tokens with label != ignore_index of -log softmax[label],
 * count[0] += number of such tokens (both ACCUMULATED; caller zeroes; loss = loss_sum / count).
 * block_partials: scratch of cad_lm_head_partials(rows) floats, or NULL.  With it the sum is DETERMINISTIC (one slot per
 * workgroup, folded by a second launch in a fixed order: run-to-run identical bits); NULL falls back to two fp32
 * atomics per workgroup (order-dependent rounding in the last bits). */
typedef struct {
    const void* hidden;
    const float* weight;
    const int64_t* comp;
    const int64_t* labels;
    float* logits;
    float* loss_sum;
    float* count;
    int64_t rows;
    int D, V, n_strands;
    int64_t ignore_index;
    int dtype;
    float* block_partials;
} cad_lm_head_args;
int cad_lm_head_fwd(const cad_lm_head_args* a, void* stream);
int64_t cad_lm_head_partials(int64_t rows);

/* Backward of cad_lm_head_fwd in one launch (replaces the autograd of RCPSLMHead.forward + cross_entropy):
 *   g[t][v]      = (softmax(logits[t])[v] - [v == labels[t]]) * [labels[t] counts] * loss_scale[0]  (+ dlogits[t][v])
 *   dhidden[0,t] = sum_v g[t][v] W[v],   dhidden[1,t] = sum_v g[t][v] W[comp[v]]
 *   dW[v]        = sum_t g[t][v] hidden[0,t] + sum_t g[t][comp[v]] hidden[1,t]      (comp is an involution)
 * logits as written by the forward; loss_scale = d loss / count on the device (NULL iff labels is NULL); dlogits (rows, V) fp32 or
 * NULL.  dw_partials: cad_lm_head_bwd_partials(rows) slots of (V, D) fp32, one per workgroup, WRITTEN (the caller sums them in
 * order: deterministic).  Shapes: cad_lm_head_bwd_supported (d_model 128 / 256, V <= 16); others return CAD_ERR_UNSUPPORTED.
 * ld (elements; 0 = D): row stride of hidden, dhidden, weight and the (V, .) rows of a dW slot -- the channels are independent in both
 * products, so a wider head (d_model 512, configs[4]) is run as one call per block of 256 channels: D = 256, ld = 512, every pointer
 * advanced by the block's first channel (slots then hold (V, ld) floats). */
typedef struct {
    const void* hidden;
    const float* weight;
    const int64_t* comp;
    const int64_t* labels;
    const float* logits;
    const float* dlogits;
    const float* loss_scale;
    void* dhidden;
    float* dw_partials;
    int64_t rows;
    int D, V, n_strands;
    int64_t ignore_index;
    int dtype;
    int64_t ld;
} cad_lm_head_bwd_args;
int cad_lm_head_bwd(const cad_lm_head_bwd_args* a, void* stream);
int cad_lm_head_bwd_supported(int D, int V);
int64_t cad_lm_head_bwd_partials(int64_t rows);

/* ---------------------------------------------------------------------------------------------------------
 * hg38 data path -- the step in front of the model.
 *
 * cad_tokenize_mlm: ASCII bases -> (input_ids, labels) on the GPU in one pass.  Replaces, per batch, the per-sample
 * Python of HG38Dataset.__getitem__ (src/dataloaders/datasets/hg38_dataset.py:160-227): the CaduceusTokenizer character
 * path (upper-casing, unknown characters -> [UNK]; caduceus/tokenization_caduceus.py:104-110), string_reverse_complement
 * (src/dataloaders/utils/rc.py:17-26) for rows whose rc flag is set, replace_value(N -> [PAD]) (:212), left padding to L,
 * and mlm_getitem (src/dataloaders/utils/mlm.py:4-32): Bernoulli(p) targets, of which 80 % become [MASK], 10 % a random
 * id in [0, vocab), 10 % stay; labels = [PAD] everywhere else.  labels == NULL: tokenisation only.
 * Random numbers are Philox4x32-10 with key = seed and counter = (position, stream id, offset), where the stream id of
 * row b is row_ids[b] (e.g. the global sample index, so a sample's mask does not depend on where in which batch -- or
 * in which loader worker -- it lands) or b when row_ids is NULL: reproducible, independent of the launch geometry, restated bit for bit by oracle/data_oracle.py (they are NOT torch's CPU generator stream, so batches
 * are equal to the reference's in distribution, not sample by sample).
 *   bases: (B, ld_bases) bytes, row b holds lengths[b] bases (lengths NULL: L each); rc_flags: (B) bytes or NULL.
 *   thr_mask = cad_mlm_threshold(mlm_probability); base_ids = ids of A, C, G, T. */
typedef struct {
    const uint8_t* bases;
    const uint8_t* rc_flags;
    const int64_t* lengths;
    int64_t* input_ids;
    int64_t* labels;
    int64_t B, L, ld_bases;
    uint64_t seed, offset;
    uint32_t thr_mask;
    int pad_id, mask_id, unk_id, n_id, vocab;
    int base_ids[4];
    const int64_t* row_ids;
} cad_mlm_args;
int cad_tokenize_mlm(const cad_mlm_args* a, void* stream);
uint32_t cad_mlm_threshold(double probability);

/* Host side.  cad_hg38_interval = the interval arithmetic of FastaInterval.__call__ (hg38_dataset.py:41-89): the
 * i_shift-th max_length window of [start, start + 2^20), shifted back inside [0, chrom_len).  Returns
 * CAD_ERR_UNSUPPORTED where the reference raises ValueError (max_length > 2^20).
 * cad_fasta_*: memory-mapped FASTA (replaces pyfaidx.Fasta as used at hg38_dataset.py:30-39,84): index built on open,
 * cad_fasta_fetch copies bases [start, end) of sequence `seq` without line breaks into `out` (caller-owned, end - start
 * bytes, e.g. a pinned staging buffer). */
int cad_hg38_interval(int64_t start, int64_t end, int64_t max_length, int64_t i_shift, int64_t chrom_len,
                      int64_t* out_start, int64_t* out_end);
int cad_fasta_open(const char* path, void** handle);
int cad_fasta_close(void* handle);
int64_t cad_fasta_num_seqs(void* handle);
const char* cad_fasta_seq_name(void* handle, int64_t i);
int64_t cad_fasta_seq_len(void* handle, int64_t i);
int64_t cad_fasta_find(void* handle, const char* name);
int cad_fasta_fetch(void* handle, int64_t seq, int64_t start, int64_t end, uint8_t* out);

/* ---------------------------------------------------------------------------------------------------------
 * Opt-in kernel timer (HIP events on the launch stream) used by bench.py for the roofline line.
 * kind: 0 scan_fwd, 1 scan_bwd, 2 conv_fwd, 3 conv_bwd, 4 add_norm_fwd, 5 add_norm_bwd, 6 embed, 7 lm_head, 8 proj. */
#define CAD_PROF_KINDS 9
int cad_prof_enable(int on);                  /* all kinds */
/* Only the kinds whose bit is set (bit k = kind k).  Every timed launch costs two event records on its stream (~10 us of idle queue
 * per launch in the kernel trace: 5 % of a training step with all kinds on); bench.py times the two scan kinds in its timed region. */
int cad_prof_enable_kinds(unsigned mask);
int cad_prof_reset(void);
/* Synchronises the recorded events.  Outputs total milliseconds and number of launches of that kind. */
int cad_prof_read(int kind, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* CADUCEUS_HIP_H */
