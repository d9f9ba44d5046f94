// cad_gemm_f32 (include/caduceus_hip.h): D (M x N) = [addend +] A (M x K) . B (K x N) with fp32 operands, fp32 accumulation and an fp32
// result on the fp32 matrix core (v_mfma_f32_16x16x4_f32) -- the dense projections of the fp32 parity path (north_star's "stated fp32
// tolerance", BASELINE configs[0]; the reference's F.linear calls inside mamba_inner_fn, modeling_caduceus.py:11,128,130), which ran
// through torch.mm / hipBLASLt until round 6.  Every operand is addressed through (row, column) element strides, one of which is 1, so
// the transposed and channel-major views of the mixer (W . X^T, X^T . W^T, Y . X^T over all tokens) need no copy; blockIdx.z walks a
// batch (the K slices of a weight gradient: partial tiles summed by the caller in fp32, as the bf16 path does).
//
// A workgroup (4 waves as 2 x 2) owns a 128 x 128 tile of D, a wave 64 x 64 of it (4 x 4 MFMA tiles = 64 accumulator registers) -- or,
// where that would leave CUs idle or multiply padding (few tiles, thin M / N: x_proj's 48 rows, BASELINE configs[0]'s 2048 tokens), a
// 64 x 64 tile with 2 x 2 MFMA tiles per wave; k is walked in chunks of 16 through ONE LDS stage per operand, the next chunk's global
// loads in flight (registers) while the current one is multiplied.  LDS layouts follow the operand's contiguous direction so that both
// the global loads (consecutive lanes = consecutive addresses) and the fragment reads (lane (g, jl): row jl, k = g) are bank-conflict-
// free: [row][k] with 17-float rows for a k-contiguous operand, [k][row] with (tile + 16)-float rows for a row-contiguous one.
#include "cad_common.h"

namespace {

constexpr int GF_KC = 16, GF_T = 256;
constexpr int GF_KSTR = GF_KC + 1;    // floats per row of the [row][k] layout
// WT = MFMA tiles per wave and dimension (4: 128 x 128 workgroup tile, 2: 64 x 64); BT = rows (= columns) of the workgroup tile
template <int WT>
struct GfCfg {
    static constexpr int BT = 32 * WT;
    static constexpr int RSTR = BT + 16;              // floats per k of the [k][row] layout (16 mod 32 banks: the two k of a half-wave do not collide)
    static constexpr int PER = BT * GF_KC / GF_T;     // elements per thread, operand and chunk
};

// element i (of BT x GF_KC) of an operand tile -> (row r inside the tile, k inside the chunk); consecutive lanes walk the contiguous direction
template <bool KFAST, int WT>
__device__ __forceinline__ void gf_elem(int i, int& r, int& k) {
    if constexpr (KFAST) {
        k = i & (GF_KC - 1), r = i >> 4;
    } else {
        r = i & (GfCfg<WT>::BT - 1), k = i / GfCfg<WT>::BT;
    }
}
template <bool KFAST, int WT>
__device__ __forceinline__ int gf_lds(int r, int k) {
    return KFAST ? r * GF_KSTR + k : k * GfCfg<WT>::RSTR + r;
}

// KA / KB: the operand's k direction is the contiguous one (A row-major; B "column-major" = the transposed view of a row-major matrix)
template <bool KA, bool KB, int WT>
__global__ __launch_bounds__(GF_T) void gemm_f32_kernel(cad_gemm_f32_args a) {
    typedef GfCfg<WT> C;
    constexpr int GF_BM = C::BT, GF_BN = C::BT, GF_PER = C::PER, WS = 16 * WT;  // WS: rows (= columns) of a wave's share
    __shared__ float As[KA ? C::BT * GF_KSTR : GF_KC * C::RSTR];
    __shared__ float Bs[KB ? C::BT * GF_KSTR : GF_KC * C::RSTR];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int g = lane >> 4, jl = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * GF_BM, n0 = (int64_t)blockIdx.x * GF_BN;
    const float* A = a.A + (int64_t)blockIdx.z * a.a_bs;
    const float* B = a.B + (int64_t)blockIdx.z * a.b_bs;
    float ra[GF_PER], rb[GF_PER];
    auto fetch = [&](int64_t k0) {
#pragma unroll
        for (int j = 0; j < GF_PER; ++j) {
            int r, k;
            gf_elem<KA, WT>(t + GF_T * j, r, k);
            const int64_t m = m0 + r, kk = k0 + k;
            ra[j] = (m < a.M && kk < a.K) ? A[m * a.a_rs + kk * a.a_cs] : 0.f;
            gf_elem<KB, WT>(t + GF_T * j, r, k);
            const int64_t n = n0 + r, kb = k0 + k;
            rb[j] = (n < a.N && kb < a.K) ? B[kb * a.b_rs + n * a.b_cs] : 0.f;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < GF_PER; ++j) {
            int r, k;
            gf_elem<KA, WT>(t + GF_T * j, r, k);
            As[gf_lds<KA, WT>(r, k)] = ra[j];
            gf_elem<KB, WT>(t + GF_T * j, r, k);
            Bs[gf_lds<KB, WT>(r, k)] = rb[j];
        }
    };
    f32x4 acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    fetch(0);
    stash();
    __syncthreads();
    for (int64_t k0 = 0; k0 < a.K; k0 += GF_KC) {
        const bool more = k0 + GF_KC < a.K;
        if (more) fetch(k0 + GF_KC);
#pragma unroll
        for (int ks = 0; ks < GF_KC; ks += 4) {
            float af[WT], bf[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) af[i] = As[gf_lds<KA, WT>(wm * WS + 16 * i + jl, ks + g)];
#pragma unroll
            for (int j = 0; j < WT; ++j) bf[j] = Bs[gf_lds<KB, WT>(wn * WS + 16 * j + jl, ks + g)];
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j) acc[i][j] = cad_mfma_16x16x4_f32(af[i], bf[j], acc[i][j]);
        }
        __syncthreads();  // every wave has read the stage
        if (more) {
            stash();
            __syncthreads();
        }
    }
    // lane (g, jl) of tile (i, j): column n0 + wn WS + 16 j + jl, rows m0 + wm WS + 16 i + 4 g + r
    float* D = a.D + (int64_t)blockIdx.z * a.d_bs;
    const float* add = a.addend ? a.addend + (int64_t)blockIdx.z * a.d_bs : nullptr;
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            const int64_t n = n0 + wn * WS + 16 * j + jl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + wm * WS + 16 * i + 4 * g + r;
                if (m < a.M && n < a.N) {
                    const int64_t o = m * a.d_rs + n * a.d_cs;
                    D[o] = add ? add[o] + acc[i][j][r] : acc[i][j][r];
                }
            }
        }
}

}  // namespace

extern "C" int cad_gemm_f32(const cad_gemm_f32_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->A && a->B && a->D && a->M > 0 && a->N > 0 && a->K > 0 && a->batch >= 1);
    CAD_CHECK_ARG((a->a_rs == 1 || a->a_cs == 1) && (a->b_rs == 1 || a->b_cs == 1));  // one contiguous direction per operand
    CAD_CHECK_ARG(a->a_rs >= 0 && a->a_cs >= 0 && a->b_rs >= 0 && a->b_cs >= 0 && a->d_rs >= 1 && a->d_cs >= 1);
    // 128 x 128 tiles where they fill the chip without multiplying padding; 64 x 64 for few tiles and thin results
    const int64_t t128 = ((a->N + 127) / 128) * ((a->M + 127) / 128) * a->batch;
    const bool small = a->M <= 64 || a->N <= 64 || t128 < 2 * (int64_t)cad_cu_count();
    const int bt = small ? 64 : 128;
    const int64_t gx = (a->N + bt - 1) / bt, gy = (a->M + bt - 1) / bt;
    CAD_CHECK_ARG(gy <= 65535 && a->batch <= 65535);
    CadProfScope prof(8, stream);
    dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)a->batch), block(GF_T);
    const bool ka = a->a_cs == 1, kb = a->b_rs == 1;  // (a 1 x 1 stride pair counts as k-contiguous)
#define GF_LAUNCH(WT)                                                                            \
    do {                                                                                         \
        if (ka && kb)                                                                            \
            CAD_LAUNCH((gemm_f32_kernel<true, true, WT>), grid, block, 0, stream, *a);           \
        else if (ka)                                                                             \
            CAD_LAUNCH((gemm_f32_kernel<true, false, WT>), grid, block, 0, stream, *a);          \
        else if (kb)                                                                             \
            CAD_LAUNCH((gemm_f32_kernel<false, true, WT>), grid, block, 0, stream, *a);          \
        else                                                                                     \
            CAD_LAUNCH((gemm_f32_kernel<false, false, WT>), grid, block, 0, stream, *a);         \
    } while (0)
    if (small)
        GF_LAUNCH(2);
    else
        GF_LAUNCH(4);
#undef GF_LAUNCH
    return cad_after_launch();
}
