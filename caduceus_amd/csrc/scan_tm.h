// Token-major selective scan ("tm" kernels): shared definitions of the forward and backward.
//
// Decomposition (measured motivation in DESIGN.md section 3, profiles/r01_ubench_gfx950.log): one LANE owns one (or
// two adjacent) channels and walks the positions of its chunk sequentially with all N states of the channel in
// registers -- the recurrence needs no cross-lane scan, no LDS and no barrier; 16..32 independent FMA chains per lane
// keep the VALU issuing, and ~100 VGPRs allow 4 waves per SIMD.  Activations are token-major (rows, L, E): a wave
// reads one 128..512-byte contiguous segment per position and tensor.  B_t / C_t are the same for every lane of a wave
// and are fetched with SCALAR loads (s_load_dwordx16 from the fp32 x_proj output) and used as SGPR operands.
// Parallelism over L comes from TM_TC-position chunks: a chunk pass from a zero state (aggregate), a tiny sequential
// combine over the chunks, and the final pass from the true chunk-start state.
// Directions are index maps: logical position p <-> physical row L-1-p for right-to-left rows; chunk and block
// boundaries are counted from the logical start, so the right-to-left pass is an exact mirror of the left-to-right one.
#pragma once
#include "cad_common.h"

#define TM_TC 512       // positions per chunk (the unit of parallelism along L)
#define TM_BLK 32       // positions between saved forward states (consumed by the backward)
#define TM_PB 4         // positions per register block (software prefetch distance)
#define TM_MAXSETS 2
#define TM_THREADS 256

static_assert(TM_TC % TM_BLK == 0 && TM_BLK % TM_PB == 0, "chunk / block / register-block nesting");

// uniform (wave-invariant) fp32 loads: the constant address space makes the compiler select s_load_dword*
#ifdef CAD_EMU
typedef const float* tm_cptr;
#define TM_CPTR(p) (p)
#else
typedef const float __attribute__((address_space(4))) * tm_cptr;
#define TM_CPTR(p) ((tm_cptr)(uintptr_t)(p))
#endif

// CPL consecutive channels of one token row, raw (un-converted)
template <typename T, int CPL>
struct __attribute__((aligned(sizeof(T) * CPL))) TmRaw {
    T v[CPL];
};
template <typename T, int CPL>
__device__ __forceinline__ void tm_unpack(const TmRaw<T, CPL>& r, float* o) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) o[c] = to_f32(r.v[c]);
}
template <typename T, int CPL>
__device__ __forceinline__ TmRaw<T, CPL> tm_pack(const float* v);
template <>
__device__ __forceinline__ TmRaw<float, 1> tm_pack<float, 1>(const float* v) {
    TmRaw<float, 1> r;
    r.v[0] = v[0];
    return r;
}
template <>
__device__ __forceinline__ TmRaw<float, 2> tm_pack<float, 2>(const float* v) {
    TmRaw<float, 2> r;
    r.v[0] = v[0], r.v[1] = v[1];
    return r;
}
template <>
__device__ __forceinline__ TmRaw<bf16_t, 1> tm_pack<bf16_t, 1>(const float* v) {
    TmRaw<bf16_t, 1> r;
    r.v[0] = from_f32<bf16_t>(v[0]);
    return r;
}
template <>
__device__ __forceinline__ TmRaw<bf16_t, 2> tm_pack<bf16_t, 2>(const float* v) {
    union {
        uint32_t w;
        TmRaw<bf16_t, 2> r;
    } c;
    c.w = cad_pack_bf16x2(v[0], v[1]);
    return c.r;
}

__host__ __device__ __forceinline__ int64_t tm_div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
