// e4m3 helpers shared by the fp8 projection (gemm_fp8.hip) and the producers that write e4m3 activations (addnorm.hip).
#pragma once
#include "cad_common.h"

// (cad_pack_fp8x4 / cad_mfma_16x16x32_fp8: cad_prims_gfx950.h; e4m3 = OCP "fn": bias 7, no infinities, max 448, S.1111.111 = NaN)

// per-token scale of an e4m3 row: max|x| / 448, clamped (a denormal-small row must not turn 1 / scale into inf); 1 for an all-zero row
__device__ __forceinline__ float cad_fp8_row_scale(float maxabs) { return maxabs > 0.f ? fmaxf(maxabs * (1.0f / 448.0f), 1e-30f) : 1.0f; }
