// Shared pieces of the selective-scan forward / backward kernels.
//
// Work decomposition (wave64-native, see DESIGN.md "scan"):
//   * one wave owns one channel e of one row (sequence) sb and walks the row in chunks of 64 lanes x SC_S items
//     = 1024 logical positions; lane j owns SC_S consecutive positions, so every HBM access of u/delta/z/out is
//     a contiguous, fully coalesced 2 KB (bf16) segment along L;
//   * the recurrence over L is split into (i) an in-register serial scan over a lane's SC_S items, (ii) a
//     Kogge-Stone scan of the affine maps (a, b) across the 64 lanes (6 shuffle steps) and (iii) a carry to the
//     next chunk; the N states are processed two at a time (float2 -> v_pk_mul/fma_f32);
//   * the SC_W waves (channels) of a workgroup share the B/C tiles of the current state pair through LDS
//     (double buffered, padded rows -> conflict-free ds_read_b64/b128).
// A right-to-left row uses the same code with physical index L-1-p: an exact mirror of the left-to-right order.
#pragma once
#include "cad_common.h"

#define SC_S 16                 // items per lane
#define SC_W 4                  // waves (= channels) per workgroup
#define SC_CHUNK (64 * SC_S)    // logical positions per chunk step
#define SC_ROW (2 * SC_S + 4)   // floats per lane row of a B/C tile (16 x float2 + 16 B pad: stride 144 B)
#define SC_TILE (64 * SC_ROW)   // floats per tile
#define SC_NMAX 64              // max d_state (pairs are indexed by lane: N/2 <= 64 would allow 128; keep 64)

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
__device__ __forceinline__ f32x2 shfl_up2(f32x2 v, int d) { return f2(__shfl_up(v[0], d), __shfl_up(v[1], d)); }
__device__ __forceinline__ f32x2 shfl_down2(f32x2 v, int d) { return f2(__shfl_down(v[0], d), __shfl_down(v[1], d)); }
__device__ __forceinline__ f32x2 shfl2(f32x2 v, int src) { return f2(__shfl(v[0], src), __shfl(v[1], src)); }
__device__ __forceinline__ f32x2 exp2_2(f32x2 v) { return f2(cad_exp2(v[0]), cad_exp2(v[1])); }
__device__ __forceinline__ float dot2(f32x2 a, f32x2 b) { return a[0] * b[0] + a[1] * b[1]; }
__device__ __forceinline__ float wave_sum1(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// SC_S logical positions [p0, p0+S) of a row -> out[] (zeros outside [0, L))
template <typename T>
__device__ __forceinline__ void sc_load(const T* row, int64_t p0, int64_t L, int rev, bool vec_ok, float* out) {
    if (vec_ok && p0 + SC_S <= L) {
        const int64_t l0 = rev ? (L - p0 - SC_S) : p0;
        typedef struct __attribute__((aligned(16))) {
            T v[SC_S];
        } vec_t;
        const vec_t tmp = *(const vec_t*)(row + l0);
#pragma unroll
        for (int j = 0; j < SC_S; ++j) out[j] = to_f32(tmp.v[rev ? (SC_S - 1 - j) : j]);
    } else {
#pragma unroll
        for (int j = 0; j < SC_S; ++j) {
            const int64_t p = p0 + j;
            out[j] = (p < L) ? to_f32(row[cad_phys(p, L, rev)]) : 0.f;
        }
    }
}

template <typename T>
__device__ __forceinline__ void sc_store(T* row, int64_t p0, int64_t L, int rev, bool vec_ok, const float* v) {
    if (vec_ok && p0 + SC_S <= L) {
        const int64_t l0 = rev ? (L - p0 - SC_S) : p0;
        typedef struct __attribute__((aligned(16))) {
            T v[SC_S];
        } vec_t;
        vec_t tmp;
#pragma unroll
        for (int j = 0; j < SC_S; ++j) tmp.v[rev ? (SC_S - 1 - j) : j] = from_f32<T>(v[j]);
        *(vec_t*)(row + l0) = tmp;
    } else {
#pragma unroll
        for (int j = 0; j < SC_S; ++j) {
            const int64_t p = p0 + j;
            if (p < L) row[cad_phys(p, L, rev)] = from_f32<T>(v[j]);
        }
    }
}

// Stage the B and C values of state pair (n0, n0+1) for logical positions [base, base + SC_CHUNK) into LDS tiles
// laid out [lane j][item i][state 0/1] (fp32, row stride SC_ROW).  Whole workgroup cooperates.
template <typename T>
__device__ __forceinline__ void sc_stage_bc(float* tB, float* tC, const T* Bm, const T* Cm, int n0, int N, int64_t SB,
                                            int64_t sb, int64_t base, int64_t L, int rev) {
    for (int idx = threadIdx.x; idx < 2 * SC_CHUNK; idx += blockDim.x) {
        const int s = idx / SC_CHUNK;
        const int tok = idx - s * SC_CHUNK;
        const int64_t p = base + tok;
        float bv = 0.f, cv = 0.f;
        if (p < L && n0 + s < N) {
            const int64_t off = ((int64_t)(n0 + s) * SB + sb) * L + cad_phys(p, L, rev);
            bv = to_f32(Bm[off]);
            cv = to_f32(Cm[off]);
        }
        const int o = (tok / SC_S) * SC_ROW + (tok % SC_S) * 2 + s;
        tB[o] = bv;
        tC[o] = cv;
    }
}
