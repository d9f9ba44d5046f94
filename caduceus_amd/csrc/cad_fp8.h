// e4m3 helpers shared by the fp8 projection (gemm_fp8.hip) and the producers that write e4m3 activations (addnorm.hip).
#pragma once
#include "cad_common.h"

// ---- e4m3 (OCP "fn": bias 7, no infinities, max 448, S.1111.111 = NaN) ------------------------------------------------
#ifdef CAD_EMU
static inline float cad_e4m3_to_f32(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7)
        r = NAN;
    else if (e == 0)
        r = ldexpf((float)m, -9);  // subnormal: m * 2^-3 * 2^-6
    else
        r = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -r : r;
}
static inline uint8_t cad_f32_to_e4m3(float f) {  // round-to-nearest-even, saturating
    if (f != f) return 0x7f;
    const uint8_t s = f < 0 ? 0x80 : 0;
    float a = fabsf(f);
    if (a >= 464.0f) return s | 0x7e;  // beyond the midpoint to the (non-existent) next value: saturate to 448
    if (a < ldexpf(1.0f, -10)) return s;  // below half the smallest subnormal
    int e;
    (void)frexpf(a, &e);  // a = m * 2^e, m in [0.5, 1)
    int E = e - 1;        // a = 1.x * 2^E
    if (E < -6) E = -6;   // subnormal range: fixed exponent
    const float q = ldexpf(1.0f, E - 3);  // spacing
    float n = nearbyintf(a / q);          // (default rounding mode: to nearest even)
    float r = n * q;
    if (r > 448.0f) r = 448.0f;
    // encode r
    if (r < ldexpf(1.0f, -6)) return s | (uint8_t)nearbyintf(r / ldexpf(1.0f, -9));
    int e2;
    const float m2 = frexpf(r, &e2);  // r = m2 * 2^e2
    const int be = e2 - 1 + 7;
    const int mant = (int)nearbyintf((m2 * 2.0f - 1.0f) * 8.0f);
    return s | (uint8_t)(be << 3) | (uint8_t)mant;
}
#endif

// four fp32 -> four e4m3 bytes (element j in byte j)
__device__ __forceinline__ uint32_t cad_pack_fp8x4(float a, float b, float c, float d) {
#ifdef CAD_EMU
    return (uint32_t)cad_f32_to_e4m3(a) | ((uint32_t)cad_f32_to_e4m3(b) << 8) | ((uint32_t)cad_f32_to_e4m3(c) << 16) |
           ((uint32_t)cad_f32_to_e4m3(d) << 24);
#else
    int p = 0;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (uint32_t)p;
#endif
}


// per-token scale of an e4m3 row: max|x| / 448, clamped (a denormal-small row must not turn 1 / scale into inf); 1 for an all-zero row
__device__ __forceinline__ float cad_fp8_row_scale(float maxabs) { return maxabs > 0.f ? fmaxf(maxabs * (1.0f / 448.0f), 1e-30f) : 1.0f; }
