// Plain-C++ types and conversions shared by the product primitives (cad_prims_gfx950.h) and their host restatement (tests/emu/cad_prims_emu.h).
// Included by those two headers once `__device__` / `__forceinline__` mean something (HIP runtime header or the emulator's).
#pragma once
#include <stdint.h>

#define CAD_WAVE 64
#define CAD_MAX_DEVICES 64   // per-device launch-attribute caches (GP_BIG_LDS / SC_BIG_LDS)

// ---- small numeric helpers ---------------------------------------------------------------------------------
typedef float f32x2 __attribute__((vector_size(8)));  // maps to v_pk_{mul,fma,add}_f32 on gfx950
typedef float f32x4 __attribute__((vector_size(16)));
typedef uint32_t u32x2 __attribute__((vector_size(8)));
typedef uint32_t u32x4 __attribute__((vector_size(16)));

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
