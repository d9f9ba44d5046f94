// Shared pieces of the selective-scan forward / backward kernels.
//
// Work decomposition (wave64-native, see DESIGN.md "scan"):
//   * one wave owns one channel e of one row (sequence) sb and walks the row in chunks of 64 lanes x SC_S items
//     = 1024 logical positions; lane j owns SC_S consecutive positions, so every HBM access of u/delta/z/out is
//     a contiguous, fully coalesced 2 KB (bf16) segment along L;
//   * the recurrence over L is split into (i) an in-register serial scan over a lane's SC_S items, (ii) a
//     Kogge-Stone scan of the affine maps (a, b) across the 64 lanes -- DPP row shifts inside 16-lane rows,
//     row_bcast15/31 across rows, no LDS -- and (iii) a carry to the next chunk; the N states are processed two at a
//     time (float2 -> v_pk_mul/fma_f32);
//   * the SC_W waves (channels) of a workgroup share the B/C tiles of the current state pair through LDS
//     (double buffered, prefetched one pair ahead through registers, padded rows -> conflict-free ds_read_b64/b128);
//   * up to two parameter sets (mamba_fwd / mamba_rev of a BiMamba layer) run in ONE launch (grid.z) so that a CU
//     holds two independent waves per SIMD even at batch 1.
// A right-to-left row uses the same code with physical index L-1-p: an exact mirror of the left-to-right order.
#pragma once
#include "cad_common.h"

// Tuning constants (overridable with -D for A/B builds).  The forward walks 1024-position chunks (16 items per lane);
// the backward keeps ~2x more per-item state in registers and walks 512-position chunks (8 items per lane).  Both
// index the saved running states by 512-position "half chunks" (SC_STATE_STEP).
#ifndef SC_S_FWD
#define SC_S_FWD 16
#endif
#ifndef SC_S_BWD
#define SC_S_BWD 8
#endif
#ifndef SC_W_FWD
#define SC_W_FWD 8              // waves (= channels) per forward workgroup (measured: 8 beats 4 by 13 %)
#endif
#ifndef SC_W_BWD
#define SC_W_BWD 8              // waves (= channels) per backward workgroup: dB/dC are summed over them in LDS (4 = two workgroups per CU:
                                // measured 15 % slower, profiles/r06_ab_w4_workgroups.txt)
#endif
#ifndef SC_OCC
#define SC_OCC 2                // register budget: waves per SIMD the kernels are compiled for
#endif
#define SC_STATE_STEP (64 * SC_S_BWD)   // positions between saved running states = the backward's chunk
// timing experiments only (WRONG results): what does a mechanism cost?  (tools/ab_train_scan.sh over -DSC_WHATIF=... builds)
// SC_WHATIF_ARITH_ONLY = the backward's "stripped timing build" (VERDICT r3 item 1): no global stores, no dB/dC slab / flush, no B/C
// tile loads / conversion / LDS stores / reads, no workgroup barrier -- what is left is the input stream of the item vectors and the
// arithmetic of the recurrence and its gradients (prologue, exp, serial chains, both wave scans, gradient loop, dA sums, epilogue)
// at the same launch shape, register count and power state.  Kernel time / that time is quoted in bench.py's `roofline`.
#define SC_WHATIF_ARITH_ONLY (2 | 32 | 64 | 2048 | 4096 | 8192)
#ifndef SC_WHATIF
#define SC_WHATIF 0
#endif
#define SC_NMAX 64              // max d_state
#define SC_MAXSETS 2
#define SC_ROW(S) (2 * (S) + 4)        // floats per lane row of a B/C tile (S x float2 + 16 B pad -> conflict-free b128)
#define SC_TILE(S) (64 * SC_ROW(S))    // floats per tile
#define SC_SV(S) ((64 * (S)) / 128)    // tokens per staging thread (2 tensors x 128 threads x SC_SV tokens = one tile)

// ---- phase timing (diagnostic builds only: -DSC_TIMING; tools/phase_timing.py) -----------------------------------------
// Every wave of workgroup (0,0,0) adds the shader-clock cycles it spent between two marks to sc_timing[wave][phase];
// the marks are scheduling barriers, so the build is for ATTRIBUTING time (incl. stalls) to phases, not for speed.
#if defined(SC_TIMING) && !defined(CAD_EMU)
#define SC_TIME_PHASES 16
static __device__ unsigned long long sc_timing[8][SC_TIME_PHASES];
#define SC_TIME_DECL unsigned long long sc_t_last_ = __builtin_readcyclecounter()
#define SC_TIME(ph)                                                                                     \
    do {                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        const unsigned long long t_ = __builtin_readcyclecounter();                                     \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (threadIdx.x & 63) == 0)           \
            atomicAdd(&sc_timing[(threadIdx.x >> 6) & 7][(ph)], t_ - sc_t_last_);                       \
        sc_t_last_ = __builtin_readcyclecounter();                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                              \
    } while (0)
#define SC_TIME_EXPORT(name)                                                                            \
    extern "C" int name(unsigned long long* out, int reset) {                                           \
        if (hipMemcpyFromSymbol(out, HIP_SYMBOL(sc_timing), sizeof(sc_timing)) != hipSuccess) return 3;   \
        if (reset) {                                                                                    \
            static unsigned long long z[8][SC_TIME_PHASES];                                             \
            if (hipMemcpyToSymbol(HIP_SYMBOL(sc_timing), z, sizeof(z)) != hipSuccess) return 3;          \
        }                                                                                               \
        return 0;                                                                                       \
    }
#else
#define SC_TIME_DECL
#define SC_TIME(ph)
#define SC_TIME_EXPORT(name)
#endif

__device__ __forceinline__ f32x2 f2(float a) {
    f32x2 r = {a, a};
    return r;
}
__device__ __forceinline__ f32x2 f2(float a, float b) {
    f32x2 r = {a, b};
    return r;
}
__device__ __forceinline__ f32x2 ld2(const float* p) {
    f32x2 r = {p[0], p[1]};
    return r;
}
__device__ __forceinline__ f32x2 exp2_2(f32x2 v) { return f2(cad_exp2(v[0]), cad_exp2(v[1])); }
__device__ __forceinline__ float dot2(f32x2 a, f32x2 b) { return a[0] * b[0] + a[1] * b[1]; }
__device__ __forceinline__ f32x2 readlane2(f32x2 v, int l) { return f2(cad_readlane(v[0], l), cad_readlane(v[1], l)); }
__device__ __forceinline__ float wave_sum1(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// wave-wide sum, result uniform (SGPR): DPP row reduction (4 steps) + 2 row broadcasts, then one v_readlane of lane 63
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_row_shr<1>(0.f, v);
    v += dpp_row_shr<2>(0.f, v);
    v += dpp_row_shr<4>(0.f, v);
    v += dpp_row_shr<8>(0.f, v);  // lane 15 of each row now holds its row total
    v += dpp_row_bcast15(0.f, v);  // lanes 31 / 63: rows 0+1 / rows 2+3
    v += dpp_row_bcast31(0.f, v);  // lane 63: everything
    return cad_readlane(v, 63);
}

// One Kogge-Stone step of the affine-map scan: (A, H) <- (A, H) o (ua, uh) where (ua, uh) is the partner's map
// (identity where the DPP pattern has no source):  x -> A * (ua * x + uh) + H.
#define SC_COMBINE(A, H, UA0, UA1, UH0, UH1) \
    do {                                     \
        const f32x2 ua_ = f2(UA0, UA1);      \
        const f32x2 uh_ = f2(UH0, UH1);      \
        H = A * uh_ + H;                     \
        A = A * ua_;                         \
    } while (0)

// ---- the scans' own primitives: register-level tricks written in inline asm (packed / DPP-folded Kogge-Stone steps, selects on SGPR lane
// masks, loads the compiler must not see, counted waits) in scan_prims_gfx950.h; tests/emu/scan_prims_emu.h restates them for the host
// emulator (test infrastructure)
#ifdef CAD_EMU
#include "scan_prims_emu.h"
#else
#include "scan_prims_gfx950.h"
#endif

// ---- per-lane item vectors -------------------------------------------------------------------------------------------
template <typename T, int S>
struct __attribute__((aligned(sizeof(T) * S >= 16 ? 16 : sizeof(T) * S))) ScVec {
    T v[S];
};

// raw (un-converted) load of SC_S logical positions [p0, p0+S); tail / unaligned rows fall back to scalar loads.
// VEC = true is the production instantiation: L % SC_S == 0 and 16-byte aligned rows, so a lane's segment is either
// fully inside or fully outside [0, L) and is moved with one 16/32/64-byte access.  VEC = false handles any L / any
// alignment element by element (small and ragged inputs).
template <typename T, int S, bool VEC>
__device__ __forceinline__ void sc_load_raw(const T* row, int64_t p0, int64_t L, int rev, ScVec<T, S>& out) {
    if constexpr (VEC) {
        if (p0 < L) {
            const int64_t l0 = rev ? (L - p0 - S) : p0;
            out = *(const ScVec<T, S>*)(row + l0);
        } else {
#pragma unroll
            for (int j = 0; j < S; ++j) out.v[j] = from_f32<T>(0.f);
        }
    } else {
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const int64_t p = p0 + j;
            const int k = rev ? (S - 1 - j) : j;  // keep the physical (memory-order) register layout of the fast path
            if (p < L)
                out.v[k] = row[cad_phys(p, L, rev)];
            else
                out.v[k] = from_f32<T>(0.f);
        }
    }
}

template <typename T, int S>
__device__ __forceinline__ void sc_unpack(const ScVec<T, S>& raw, int rev, float* out) {
    if constexpr (sizeof(T) == 2 && S % 2 == 0) {
        // bf16: work on the raw dwords -- logical item j is physical element S-1-j on a right-to-left row, i.e. the
        // dword order is reversed and the halves of every dword swap: one select + one rotate per dword
        constexpr int NW = S / 2;
#ifdef CAD_EMU
        struct W { uint32_t w[NW]; };
        const W ww = __builtin_bit_cast(W, raw);
        const uint32_t* w = ww.w;
#else
        typedef uint32_t uw __attribute__((ext_vector_type(NW)));  // stays in registers (an array may go to scratch)
        const uw w = __builtin_bit_cast(uw, raw);
#endif
        const uint64_t rmask = sc_rev_mask(rev);
        const uint32_t rot = rev ? 16u : 0u;
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const uint32_t x = sc_rot(sc_sel(w[q], w[NW - 1 - q], rmask), rot);
            out[2 * q] = cad_bits2f(x << 16);
            out[2 * q + 1] = cad_bits2f(x & 0xFFFF0000u);
        }
    } else {
#pragma unroll
        for (int j = 0; j < S; ++j) out[j] = to_f32(raw.v[rev ? (S - 1 - j) : j]);
    }
}
template <typename T, int S, bool VEC>
__device__ __forceinline__ void sc_load(const T* row, int64_t p0, int64_t L, int rev, float* out) {
    ScVec<T, S> raw;
    sc_load_raw<T, S, VEC>(row, p0, L, rev, raw);
    sc_unpack<T, S>(raw, rev, out);
}
template <typename T, int S, bool VEC>
__device__ __forceinline__ void sc_store(T* row, int64_t p0, int64_t L, int rev, const float* v) {
    if constexpr (VEC && sizeof(T) == 2 && (S == 8 || S == 16)) {
        // bf16: convert in logical order, then reverse on the packed dwords (see sc_unpack)
        if (p0 < L) {
            const int64_t l0 = rev ? (L - p0 - S) : p0;
            constexpr int NW = S / 2;
            uint32_t pk[NW];
#pragma unroll
            for (int q = 0; q < NW; ++q) pk[q] = cad_pack_bf16x2(v[2 * q], v[2 * q + 1]);
            const uint64_t rmask = sc_rev_mask(rev);
            const uint32_t rot = rev ? 16u : 0u;
#pragma unroll
            for (int q = 0; q < NW; q += 4) {
                u32x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) o[t] = sc_rot(sc_sel(pk[q + t], pk[NW - 1 - q - t], rmask), rot);
                *(u32x4*)(row + l0 + 2 * q) = o;
            }
        }
    } else if constexpr (VEC) {
        if (p0 < L) {
            const int64_t l0 = rev ? (L - p0 - S) : p0;
            float m[S];  // memory order
#pragma unroll
            for (int k = 0; k < S; ++k) m[k] = v[rev ? (S - 1 - k) : k];
#pragma unroll
            for (int k = 0; k < S; k += 4) cad_cvt_store<T, 4>(row + l0 + k, m + k);
        }
    } else {
#pragma unroll
        for (int j = 0; j < S; ++j) {
            const int64_t p = p0 + j;
            if (p < L) row[cad_phys(p, L, rev)] = from_f32<T>(v[j]);
        }
    }
}

template <int V>
struct DirTagC {
    static constexpr int value = V;
};
template <typename T, int S, bool REV>
__device__ __forceinline__ void sc_unpack_d(const ScVec<T, S>& raw, float* out) {
    if constexpr (sizeof(T) == 2 && S % 2 == 0) {
        constexpr int NW = S / 2;
#ifdef CAD_EMU
        struct W { uint32_t w[NW]; };
        const W ww = __builtin_bit_cast(W, raw);
        const uint32_t* w = ww.w;
#else
        typedef uint32_t uw __attribute__((ext_vector_type(NW)));
        const uw w = __builtin_bit_cast(uw, raw);
#endif
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            const uint32_t x = w[REV ? NW - 1 - q : q];
            out[2 * q] = cad_bits2f(REV ? (x & 0xFFFF0000u) : (x << 16));
            out[2 * q + 1] = cad_bits2f(REV ? (x << 16) : (x & 0xFFFF0000u));
        }
    } else {
        sc_unpack<T, S>(raw, REV ? 1 : 0, out);
    }
}
template <typename T, int S, bool VEC, bool REV>
__device__ __forceinline__ void sc_store_d(T* row, int64_t p0, int64_t L, const float* v) {
    if constexpr (VEC && sizeof(T) == 2 && (S == 8 || S == 16)) {
        if (p0 < L) {
            const int64_t l0 = REV ? (L - p0 - S) : p0;
            constexpr int NW = S / 2;
#pragma unroll
            for (int q = 0; q < NW; q += 4) {
                u32x4 o;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int k = q + t;
                    o[t] = REV ? cad_pack_bf16x2(v[S - 1 - 2 * k], v[S - 2 - 2 * k]) : cad_pack_bf16x2(v[2 * k], v[2 * k + 1]);
                }
                *(u32x4*)(row + l0 + 2 * q) = o;
            }
        }
    } else {
        sc_store<T, S, VEC>(row, p0, L, REV ? 1 : 0, v);
    }
}
// Direction select + bf16 -> fp32 widening of an item in ONE v_perm_b32 (cad_perm): logical item j of a right-to-left row is physical
// element S-1-j, i.e. the OTHER half of dword NW-1-q -- which dword and which half are a byte selector, wave-uniform (SGPR), so an
// 8-item vector unpacks in 8 VALU instructions for either direction without a branch (select + rotate + widen: 16; branched: 8).
struct ScDirSel {
    uint32_t even, odd;  // selectors of logical items 2q / 2q + 1 from {s0 = w[NW-1-q], s1 = w[q]}
};
__device__ __forceinline__ ScDirSel sc_dir_sel(int rev) {
    ScDirSel r;
    r.even = (uint32_t)cad_uniform((int)(rev ? 0x07060c0cu : 0x01000c0cu));  // rev: high half of s0;  fwd: low half of s1
    r.odd = (uint32_t)cad_uniform((int)(rev ? 0x05040c0cu : 0x03020c0cu));   // rev: low half of s0;   fwd: high half of s1
    return r;
}
template <typename T, int S>
__device__ __forceinline__ void sc_unpack_p(const ScVec<T, S>& raw, int rev, ScDirSel sel, float* out) {
    if constexpr (sizeof(T) == 2 && S % 2 == 0) {
        constexpr int NW = S / 2;
#ifdef CAD_EMU
        struct W { uint32_t w[NW]; };
        const W ww = __builtin_bit_cast(W, raw);
        const uint32_t* w = ww.w;
#else
        typedef uint32_t uw __attribute__((ext_vector_type(NW)));
        const uw w = __builtin_bit_cast(uw, raw);
#endif
#pragma unroll
        for (int q = 0; q < NW; ++q) {
            out[2 * q] = cad_bits2f(cad_perm(w[NW - 1 - q], w[q], sel.even));
            out[2 * q + 1] = cad_bits2f(cad_perm(w[NW - 1 - q], w[q], sel.odd));
        }
    } else {
        sc_unpack<T, S>(raw, rev, out);
    }
}
template <typename F>
__device__ __forceinline__ void sc_by_dir(int rev, F&& f) {
    if (cad_uniform(rev)) {
        f(DirTagC<1>{});
        
    } else {
        f(DirTagC<0>{});
        
    }
}
// ---- B/C tile staging: global -> registers (prefetch) -> LDS ----------------------------------------------------------
// Threads 0..255 of the workgroup stage; thread t owns tensor (t >> 7) (0 = B, 1 = C) and the SV = S/2 logical
// positions base + SV * (t & 127) .. of BOTH states of the pair.  Tile layout in LDS: [lane j][item i][state 0/1] fp32
// with row stride SC_ROW(S), i.e. the 2*SV floats a staging thread writes are contiguous.
template <typename T, int SV>
struct __attribute__((aligned(sizeof(T) * SV >= 16 ? 16 : 8))) StVec {
    T v[SV];
};
template <typename T, int SV>
struct StageRegs {
    StVec<T, SV> s0, s1;
    bool ok0, ok1;  // VEC path: the vector was read from a clamped (always valid) address; zero it at conversion time
};

// counted form of the wait above: returns as soon as at most `keep` vector-memory operations are outstanding.  Used in
// the pair-step in which the LDS-DMA prefetch of the next chunk's item vectors (sc_glds16, SC_NDMA operations) was issued
// BEHIND the tile loads: the tile loads are waited for, the prefetch stays in flight.
#define SC_NDMA 6
static_assert(SC_NDMA == 6, "the immediate of s_waitcnt vmcnt(6) above");

// ---- LDS-DMA prefetch of the per-chunk item vectors -------------------------------------------------------------------
// The u / delta / z / dout / out vectors of a chunk are needed at its very start; loaded there they expose the full HBM
// latency once per chunk on every wave (measured: 19-32 % of the scan kernels, profiles/r02_phase_timing_mfma_flush.txt),
// and the register file has no room for a second set.  global_load_lds_dwordx4 moves 16 bytes per lane from a per-lane
// global address straight into LDS at (wave-uniform base) + 16 * lane without touching a VGPR, so the NEXT chunk's
// vectors are fetched one whole chunk ahead into wave-private LDS slots and read back with ds_read_b128 when needed.
// Completion is tracked by vmcnt like any load (the issuing wave waits vmcnt(0) before its ds_read).
__device__ __forceinline__ uint32_t sc_lds_off(const void* p) { return cad_lds_off(p); }
__device__ __forceinline__ void sc_glds16(const void* gsrc, uint32_t lds_base) { cad_glds16(gsrc, lds_base); }
// Static priority for the second half of a workgroup's waves (MI355X: of the two waves a 512-thread workgroup places on
// each SIMD, the later-dispatched one loses VALU arbitration; measured here: waves 4-7 spend 1.2-2.3x the cycles of
// waves 0-3 in the DPP / packed-FMA phases, profiles/r02_phase_timing_dma_prefetch.txt, while waves 0-3 -- which also do
// the tile staging -- idle at the barrier).
#ifndef SC_PRIO
#define SC_PRIO 0   // (3 = every scan wave above the co-resident fold kernel's priority 0: -2.3 % per layer while that kernel polled every
                    // ~0.5 us, nothing once it sleeps through the predicted gap between arrivals -- profiles/r06_ab_stream_fold.txt)
#endif
__device__ __forceinline__ void sc_static_priority(int wave, int nwaves) {
#if SC_PRIO == 1 && !defined(CAD_EMU)
    if (wave >= nwaves / 2) __builtin_amdgcn_s_setprio(1);  // wave is wave-uniform (readfirstlane): a scalar branch
#elif SC_PRIO == 2 && !defined(CAD_EMU)
    if (wave < nwaves / 2) __builtin_amdgcn_s_setprio(1);   // A/B: the staging half instead
#elif SC_PRIO == 3 && !defined(CAD_EMU)
    __builtin_amdgcn_s_setprio(1);  // every scan wave above the (priority 0) waves of a kernel that shares the CU: the concurrent dB / dC fold
#endif
}

// Per-thread constants of the staging path, computed once per kernel: the thread's tensor base (+ its row sb) and its
// first token inside a chunk.  Keeps the per-pair address arithmetic to one uniform multiply and two adds.
template <typename T>
struct StageCtx {
    const T* src;        // Bm or Cm, advanced to row (state 0, sb)
    int64_t row_stride;  // SB * L: distance between consecutive states
    int tok;             // first token of this thread inside a chunk
    bool on;             // threads 0..255 stage
    const T* cur;        // sc_stage_seek: this thread's vector of state 0 in the chunk being staged (clamped address)
    bool in;             // ... and whether it lies inside the sequence
};
template <typename T, int S>
__device__ __forceinline__ StageCtx<T> sc_stage_ctx(const T* Bm, const T* Cm, int64_t SB, int64_t sb, int64_t L) {
    const int t = threadIdx.x;
    StageCtx<T> c;
    c.on = t < 256;
    c.src = ((t >> 7) & 1 ? Cm : Bm) + sb * L;
    c.row_stride = SB * L;
    c.tok = (t & 127) * SC_SV(S);
    c.cur = c.src;
    c.in = false;
    return c;
}
// Points the staging thread at the chunk starting at logical position `base` (VEC path).  Done once per chunk, so that
// the per-pair sc_stage_issue is two address adds (the full address arithmetic costs ~40 instructions).
template <typename T, int S>
__device__ __forceinline__ void sc_stage_seek(StageCtx<T>& c, int64_t base, int64_t L, int rev) {
    constexpr int SV = SC_SV(S);
    const int64_t p0 = base + c.tok;
    c.in = p0 < L;
    c.cur = c.src + (c.in ? (rev ? (L - p0 - SV) : p0) : 0);
}
template <typename T, int S>
__device__ __forceinline__ void sc_stage_issue(StageRegs<T, SC_SV(S)>& r, const StageCtx<T>& c, int n0, int N) {
    if (!c.on) return;
    if (SC_WHATIF & 8192) return;  // (timing experiment: no B / C tile loads at all)
    const bool two = n0 + 1 < N;
    const T* p = c.cur + n0 * c.row_stride;  // n0 < N always
    sc_async_load(r.s0, p);
    sc_async_load(r.s1, two ? p + c.row_stride : p);
    r.ok0 = c.in;
    r.ok1 = c.in && two;
}

// Element-wise staging load (ragged L or unaligned rows; the vector path is sc_stage_seek + sc_stage_issue).
template <typename T, int S, bool VEC>
__device__ __forceinline__ void sc_stage_load(StageRegs<T, SC_SV(S)>& r, const StageCtx<T>& c, int n0, int N, int64_t base,
                                              int64_t L, int rev) {
    static_assert(!VEC, "the vector path stages through sc_stage_seek / sc_stage_issue");
    constexpr int SV = SC_SV(S);
    if (!c.on) return;
    const int64_t p0 = base + c.tok;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        StVec<T, SV>& dst = s ? r.s1 : r.s0;
        const T* row = c.src + (int64_t)(n0 + s) * c.row_stride;
        (s ? r.ok1 : r.ok0) = true;
#pragma unroll
        for (int j = 0; j < SV; ++j) {
            const int64_t p = p0 + j;
            const int k = rev ? (SV - 1 - j) : j;
            if (n0 + s < N && p < L)
                dst.v[k] = row[cad_phys(p, L, rev)];
            else
                dst.v[k] = from_f32<T>(0.f);
        }
    }
}

template <typename T, int S, bool VEC>
__device__ __forceinline__ void sc_stage_store(StageRegs<T, SC_SV(S)>& r, float* tiles /* B tile, C tile follows */,
                                               int rev, bool keep_dma = false) {
    constexpr int SV = SC_SV(S);
    const int t = threadIdx.x;
    if (t >= 256) return;
    if constexpr (VEC) {
        if constexpr (sizeof(StVec<T, SV>) <= 16)
            sc_async_wait_keep(r.s0, r.s1, keep_dma);
        else
            sc_async_wait(r.s0, r.s1);
    }
    if (SC_WHATIF & 4096) return;  // (timing experiment: no conversion / LDS stores of the staged tile)
    float* tile = tiles + (t >> 7) * SC_TILE(S);
    const int tok = (t & 127) * SV;  // position inside the chunk
    float* dst = tile + (tok / S) * SC_ROW(S) + (tok % S) * 2;
#ifndef CAD_EMU
    if constexpr (VEC && sizeof(T) == 2 && SV % 4 == 0) {
        // bf16 fast path on the raw dwords (the generic loop below costs ~7 instructions per stored float: per-element
        // selects for `rev` and `ok`).  The direction and "some vector came from a clamped address" are wave-uniform, so they
        // pick one of four straight-line variants (real branches: each variant ends in its own LDS stores): reversing the
        // token order is then a matter of WHICH dword feeds WHICH store and in which order its halves are widened -- no
        // select, no rotate -- and the zero mask exists only in the tail variants: 4 VALU instructions per stored 16 bytes
        // (2 shifts, 2 ANDs) instead of 10.
        constexpr int NW = SV / 2;
        typedef uint32_t uw __attribute__((ext_vector_type(NW)));
        const uw a = __builtin_bit_cast(uw, r.s0), b = __builtin_bit_cast(uw, r.s1);
        const uint32_t m0 = r.ok0 ? 0xFFFFFFFFu : 0u, m1 = r.ok1 ? 0xFFFFFFFFu : 0u;
        const bool rv = __builtin_amdgcn_readfirstlane(rev) != 0;
        const bool tail = cad_wave_any(!(r.ok0 && r.ok1));
        auto emit = [&](auto rtag, auto mtag) {
            constexpr bool REV = decltype(rtag)::value != 0, MASK = decltype(mtag)::value != 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) {  // logical tokens 2i, 2i + 1 of this thread; states 0 / 1 interleaved
                typedef uint32_t u4 __attribute__((ext_vector_type(4)));
                uint32_t x = a[REV ? NW - 1 - i : i], y = b[REV ? NW - 1 - i : i];
                if constexpr (MASK) x &= m0, y &= m1;
                const uint32_t lo0 = x << 16, lo1 = y << 16, hi0 = x & 0xFFFF0000u, hi1 = y & 0xFFFF0000u;
                u4 q;
                if constexpr (REV)
                    q[0] = hi0, q[1] = hi1, q[2] = lo0, q[3] = lo1;  // the later physical token comes first
                else
                    q[0] = lo0, q[1] = lo1, q[2] = hi0, q[3] = hi1;
                *(u4*)(dst + 4 * i) = q;
            }
        };
        if (rv) {
            if (tail) emit(DirTagC<1>{}, DirTagC<1>{}); else emit(DirTagC<1>{}, DirTagC<0>{});
        } else {
            if (tail) emit(DirTagC<0>{}, DirTagC<1>{}); else emit(DirTagC<0>{}, DirTagC<0>{});
        }
        return;
    }
#endif
#pragma unroll
    for (int j = 0; j < SV; ++j) {
        const int k = rev ? (SV - 1 - j) : j;
        dst[2 * j] = r.ok0 ? to_f32(r.s0.v[k]) : 0.f;
        dst[2 * j + 1] = r.ok1 ? to_f32(r.s1.v[k]) : 0.f;
    }
}

#define SC_BIG_LDS(kern, bytes) CAD_BIG_LDS(kern, bytes)  // (cad_prims_gfx950.h)

