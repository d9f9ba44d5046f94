// Shared device/host helpers for the Caduceus gfx950 kernels.
//
// The kernels are written for CDNA4 (wave64, LDS, HBM-coalesced channel-major streams).  The only portability
// seam is CAD_EMU (one #ifdef below): the test-suite compiles these same sources with g++ against tests/emu/ so that
// kernel logic can be parity-checked against the oracle on a machine without a GPU.  There is no CUDA path.
#pragma once
#include <stdint.h>

#include "../../include/caduceus_hip.h"

// ---- the ONE seam between the product and the host emulator --------------------------------------------------------------------------
// Every primitive the kernels use that is not plain C++ -- launch / dynamic-LDS macros, hardware transcendentals, v_cvt_pk_bf16_f32, DPP
// cross-lane moves, v_readlane, MFMA, the transposing LDS read, LDS-DMA, wave votes -- is defined in cad_prims_gfx950.h (builtins and
// inline asm, the product) and restated for g++ in tests/emu/cad_prims_emu.h (test infrastructure: lanes as fibers, tests/emu/emu_runtime.h).
#ifdef CAD_EMU
#include "cad_prims_emu.h"
#else
#include "cad_prims_gfx950.h"
#endif

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

__device__ __forceinline__ float cad_exp(float x) { return cad_exp2(x * CAD_LOG2E); }
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
// softplus for results that are rounded to bf16 anyway: max(x, 0) + log(1 + exp(-|x|)) -- two transcendentals instead of
// three; below t = exp(-|x|) ~ 1e-4 the plain log(1 + t) keeps only 3-4 digits of a term that is < 1e-4 in absolute value
// (and is replaced by t itself below 2^-12), far inside bf16's 8 bits
__device__ __forceinline__ float cad_softplus_lowp(float x) {
    const float t = cad_exp(-fabsf(x));
    const float l = t < 0.000244140625f ? t : cad_log(1.0f + t);
    return fmaxf(x, 0.0f) + l;
}
__device__ __forceinline__ float cad_sigmoid(float x) { return cad_rcp(1.0f + cad_exp(-x)); }
// sigmoid(x) recovered from sp = softplus(x):  1 - exp(-sp)  (series below 1/16: no cancellation for tiny sp)
__device__ __forceinline__ float cad_sigmoid_from_softplus(float sp) {
    const float poly = sp * (1.0f - sp * (0.5f - sp * (0.16666667f - sp * 0.041666668f)));
    const float big = 1.0f - cad_exp(-sp);
    return sp < 0.0625f ? poly : big;
}

// ---- matrix core (MFMA) ------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16:  D (16 x 16 fp32) = A (16 x 32 bf16) * B (32 x 16 bf16) + C, one tile per wave.
// Operand layouts (lane l, g = l >> 4):  A: row i = l & 15, elements k = 8g .. 8g+7 (4 dwords, element t in dword t >> 1,
// half t & 1);  B: column j = l & 15, elements k = 8g .. 8g+7;  C / D: column j = l & 15, rows 4g + r (r = 0..3).

// ---- direction / index maps ----------------------------------------------------------------------------------
// logical position p in [0, L) of a row <-> physical index along L
__device__ __forceinline__ int64_t cad_phys(int64_t p, int64_t L, int rev) { return rev ? (L - 1 - p) : p; }

// ---- host side -------------------------------------------------------------------------------------------------
#define CAD_CHECK_ARG(cond)              \
    do {                                 \
        if (!(cond)) return CAD_ERR_BAD_ARG; \
    } while (0)

int cad_after_launch();  // hipGetLastError -> cad_status
int cad_cu_count();      // compute units of the current device (queried once per device; 256 on the host emulator / on failure): the
                         // "one workgroup per CU" launchers size their grids with it, as the python side sizes the work (ops._cu_count)

// RAII kernel timer (no-op unless cad_prof_enable(1)); see api.hip
struct CadProfScope {
    int kind;
    void* stream;
    int slot;
    CadProfScope(int kind, void* stream);
    ~CadProfScope();
};
