// Shared device/host helpers for the Caduceus gfx950 kernels.
//
// The kernels are written for CDNA4 (wave64, LDS, HBM-coalesced channel-major streams).  The only portability
// seam is CAD_EMU: the test-suite compiles these same sources with g++ against tests/emu/emu_runtime.h so that
// kernel logic can be parity-checked against the oracle on a machine without a GPU.  There is no CUDA path.
#pragma once
#include <stdint.h>

#include "../../include/caduceus_hip.h"

#ifdef CAD_EMU
#include "emu_runtime.h"
#define CAD_LAUNCH(kern, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [=]() { kern(__VA_ARGS__); })
#define CAD_DEVICE_BUILD 0
#else
#include <hip/hip_runtime.h>
#define CAD_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), (shmem), (hipStream_t)(stream), __VA_ARGS__)
#define CAD_DEVICE_BUILD 1
#endif

#define CAD_WAVE 64

// dynamic LDS (16-byte aligned base; keep ALL of a kernel's LDS in this one region - guide G17)
#ifdef CAD_EMU
#define CAD_DYN_SMEM(T, name) T* name = (T*)emu::dyn_smem()
#else
#define CAD_DYN_SMEM(T, name)                                              \
    extern __shared__ __attribute__((aligned(16))) char cad_smem_raw[];   \
    T* name = (T*)cad_smem_raw
#endif

// ---- small numeric helpers ---------------------------------------------------------------------------------
typedef float f32x2 __attribute__((vector_size(8)));  // maps to v_pk_{mul,fma,add}_f32 on gfx950
typedef float f32x4 __attribute__((vector_size(16)));
typedef uint32_t u32x2 __attribute__((vector_size(8)));

struct bf16_t {
    uint16_t v;
};

__device__ __forceinline__ float cad_bits2f(uint32_t u) {
    union {
        uint32_t u;
        float f;
    } c;
    c.u = u;
    return c.f;
}
__device__ __forceinline__ uint32_t cad_f2bits(float f) {
    union {
        uint32_t u;
        float f;
    } c;
    c.f = f;
    return c.u;
}
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return cad_bits2f((uint32_t)x.v << 16); }
template <typename T>
__device__ __forceinline__ T from_f32(float f);
template <>
__device__ __forceinline__ float from_f32<float>(float f) {
    return f;
}
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) {  // round-to-nearest-even, NaN preserved
    uint32_t u = cad_f2bits(f);
    bf16_t r;
    if ((u & 0x7fffffffu) > 0x7f800000u) {
        r.v = (uint16_t)((u >> 16) | 0x40);
    } else {
        u += 0x7fffu + ((u >> 16) & 1u);
        r.v = (uint16_t)(u >> 16);
    }
    return r;
}

// two fp32 -> packed bf16x2 (lo in bits [15:0]); v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even)
__device__ __forceinline__ uint32_t cad_pack_bf16x2(float lo, float hi) {
#ifdef CAD_EMU
    return (uint32_t)from_f32<bf16_t>(lo).v | ((uint32_t)from_f32<bf16_t>(hi).v << 16);
#else
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
#endif
}
// N fp32 values -> N contiguous elements of T at dst (N even, dst suitably aligned by the caller's vector type)
template <typename T, int N>
__device__ __forceinline__ void cad_cvt_store(T* dst, const float* v);
template <>
__device__ __forceinline__ void cad_cvt_store<float, 4>(float* dst, const float* v) {
    struct __attribute__((aligned(16))) V { float f[4]; } t = {{v[0], v[1], v[2], v[3]}};
    *(V*)dst = t;
}
template <>
__device__ __forceinline__ void cad_cvt_store<bf16_t, 4>(bf16_t* dst, const float* v) {
    struct __attribute__((aligned(8))) V { uint32_t w[2]; } t = {{cad_pack_bf16x2(v[0], v[1]), cad_pack_bf16x2(v[2], v[3])}};
    *(V*)dst = t;
}

template <>
__device__ __forceinline__ void cad_cvt_store<float, 2>(float* dst, const float* v) {
    struct __attribute__((aligned(8))) V { float f[2]; } t = {{v[0], v[1]}};
    *(V*)dst = t;
}
template <>
__device__ __forceinline__ void cad_cvt_store<bf16_t, 2>(bf16_t* dst, const float* v) {
    *(uint32_t*)dst = cad_pack_bf16x2(v[0], v[1]);
}

#define CAD_LOG2E 1.4426950408889634f

__device__ __forceinline__ float cad_exp2(float x) {
#ifdef CAD_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);  // v_exp_f32
#endif
}
__device__ __forceinline__ float cad_exp(float x) { return cad_exp2(x * CAD_LOG2E); }
__device__ __forceinline__ float cad_log(float x) {
#ifdef CAD_EMU
    return logf(x);
#else
    return __builtin_amdgcn_logf(x) * 0.6931471805599453f;  // v_log_f32 (log2) * ln2
#endif
}
__device__ __forceinline__ float cad_rcp(float x) {
#ifdef CAD_EMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ float cad_rsqrt(float x) {
#ifdef CAD_EMU
    return 1.0f / sqrtf(x);
#else
    return __builtin_amdgcn_rsqf(x);
#endif
}
// softplus with the upstream threshold (x > 20 -> x); log1p(e) evaluated as log(w) * e / (w - 1), w = 1 + e,
// which is accurate to ~1 ulp also for tiny e (plain log(1+e) loses all digits there).
// Branch-free (selects only): every lane evaluates both sides; NaN/inf of the unselected side is discarded.
__device__ __forceinline__ float cad_softplus(float x) {
    const float e = cad_exp(x);
    const float w = 1.0f + e;
    const float d = w - 1.0f;
    const float lp = cad_log(w) * (e * cad_rcp(d));
    const float sp = (d == 0.0f) ? e : lp;
    return (x > 20.0f) ? x : sp;
}
__device__ __forceinline__ float cad_sigmoid(float x) { return cad_rcp(1.0f + cad_exp(-x)); }

// ---- cross-lane primitives (DPP on gfx950; emulated through the fiber exchange in the test build) -----------------
// Each returns, per lane, the value of `v` in the source lane selected by the pattern, or `old` where the pattern has
// no source for this lane -- exactly v_mov_b32_dpp with bound_ctrl:0.  Passing the identity element as `old` lets a
// scan step run unconditionally on all lanes.
#ifdef CAD_EMU
template <int N>
__device__ __forceinline__ float dpp_row_shr(float old, float v) {
    const int lane = emu::lane_id();
    const bool ok = (lane & 15) >= N;
    const float r = emu_exchange(v, ok ? lane - N : lane);
    return ok ? r : old;
}
template <int N>
__device__ __forceinline__ float dpp_row_shl(float old, float v) {
    const int lane = emu::lane_id();
    const bool ok = (lane & 15) + N < 16;
    const float r = emu_exchange(v, ok ? lane + N : lane);
    return ok ? r : old;
}
__device__ __forceinline__ float dpp_row_bcast15(float old, float v) {  // rows 1 and 3 <- lane 15 of the previous row
    const int lane = emu::lane_id();
    const bool ok = ((lane >> 4) & 1) == 1;
    const float r = emu_exchange(v, ok ? (lane & ~15) - 1 : lane);
    return ok ? r : old;
}
__device__ __forceinline__ float dpp_row_bcast31(float old, float v) {  // rows 2 and 3 <- lane 31
    const int lane = emu::lane_id();
    const bool ok = lane >= 32;
    const float r = emu_exchange(v, ok ? 31 : lane);
    return ok ? r : old;
}
__device__ __forceinline__ float dpp_wave_shr1(float old, float v) {
    const int lane = emu::lane_id();
    const float r = emu_exchange(v, lane >= 1 ? lane - 1 : lane);
    return lane >= 1 ? r : old;
}
__device__ __forceinline__ float dpp_wave_shl1(float old, float v) {
    const int lane = emu::lane_id();
    const float r = emu_exchange(v, lane < 63 ? lane + 1 : lane);
    return lane < 63 ? r : old;
}
__device__ __forceinline__ float cad_readlane(float v, int l) { return emu_exchange(v, l); }
__device__ __forceinline__ int cad_uniform(int v) { return v; }
#else
#define CAD_DPP(old, v, ctrl, rmask)                                                                          \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)),              \
                                                          __builtin_bit_cast(int, (float)(v)), (ctrl), (rmask), 0xf, false))
template <int N>
__device__ __forceinline__ float dpp_row_shr(float old, float v) {
    return CAD_DPP(old, v, 0x110 + N, 0xf);
}
template <int N>
__device__ __forceinline__ float dpp_row_shl(float old, float v) {
    return CAD_DPP(old, v, 0x100 + N, 0xf);
}
__device__ __forceinline__ float dpp_row_bcast15(float old, float v) { return CAD_DPP(old, v, 0x142, 0xa); }
__device__ __forceinline__ float dpp_row_bcast31(float old, float v) { return CAD_DPP(old, v, 0x143, 0xc); }
__device__ __forceinline__ float dpp_wave_shr1(float old, float v) { return CAD_DPP(old, v, 0x138, 0xf); }
__device__ __forceinline__ float dpp_wave_shl1(float old, float v) { return CAD_DPP(old, v, 0x130, 0xf); }
__device__ __forceinline__ float cad_readlane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// tell the compiler a value is wave-uniform (e.g. the wave index threadIdx.x >> 6): everything derived from it -- row
// base pointers, channel parameters -- then lives in SGPRs instead of VGPRs
__device__ __forceinline__ int cad_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// compiler-only fence: keeps the scheduler from hoisting (LDS) loads across this point, which bounds live ranges
__device__ __forceinline__ void cad_sched_fence() {
#ifndef CAD_EMU
    asm volatile("" ::: "memory");
#endif
}

// ---- direction / index maps ----------------------------------------------------------------------------------
// logical position p in [0, L) of a row <-> physical index along L
__device__ __forceinline__ int64_t cad_phys(int64_t p, int64_t L, int rev) { return rev ? (L - 1 - p) : p; }

// ---- host side -------------------------------------------------------------------------------------------------
#define CAD_CHECK_ARG(cond)              \
    do {                                 \
        if (!(cond)) return CAD_ERR_BAD_ARG; \
    } while (0)

int cad_after_launch();  // hipGetLastError -> cad_status

// RAII kernel timer (no-op unless cad_prof_enable(1)); see api.hip
struct CadProfScope {
    int kind;
    void* stream;
    int slot;
    CadProfScope(int kind, void* stream);
    ~CadProfScope();
};
