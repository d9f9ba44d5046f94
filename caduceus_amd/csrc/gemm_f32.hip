// cad_gemm_f32 (include/caduceus_hip.h): D (M x N) = [addend +] A (M x K) . B (K x N) with fp32 operands, fp32 accumulation and an fp32
// result on the fp32 matrix core (v_mfma_f32_16x16x4_f32) -- the dense projections of the fp32 parity path (north_star's "stated fp32
// tolerance", BASELINE configs[0]; the reference's F.linear calls inside mamba_inner_fn, modeling_caduceus.py:11,128,130), which ran
// through torch.mm / hipBLASLt until round 6.  Every operand is addressed through (row, column) element strides, one of which is 1, so
// the transposed and channel-major views of the mixer (W . X^T, X^T . W^T, Y . X^T over all tokens) need no copy; blockIdx.z walks a
// batch (the K slices of a weight gradient: partial tiles summed by the caller in fp32, as the bf16 path does).
//
// A workgroup (4 waves as 2 x 2) owns a 128 x 128 tile of D, a wave 64 x 64 of it (4 x 4 MFMA tiles = 64 accumulator registers) -- or,
// where that would leave CUs idle or multiply padding (few tiles, thin M / N: x_proj's 48 rows, BASELINE configs[0]'s 2048 tokens), a
// 64 x 64 tile with 2 x 2 MFMA tiles per wave; k is walked in chunks of 32 through ONE LDS stage per operand, the next chunk's global
// loads in flight (registers) while the current one is multiplied.  A thread's elements of a chunk lie at ONE constant stride from a
// per-thread base pointer (consecutive lanes walk the operand's contiguous direction), so the address arithmetic is a pointer and an
// increment, not an index product per element.  LDS layouts follow the contiguous direction: [row][k] with 36-float rows for a
// k-contiguous operand -- an MFMA k slot g takes the 8 consecutive k = 8 g .. 8 g + 7 of the chunk over its eight k-steps, so a fragment
// is two ds_read_b128 -- and [k][row] with (tile + 16)-float rows for a row-contiguous one (eight ds_read_b32).
#include "cad_common.h"

namespace {

constexpr int GF_KC = 32, GF_T = 256;
constexpr int GF_KSTR = GF_KC + 4;    // floats per row of the [row][k] layout (16-byte aligned rows)
// WT = MFMA tiles per wave and dimension (4: 128 x 128 workgroup tile, 2: 64 x 64); BT = rows (= columns) of the workgroup tile
template <int WT>
struct GfCfg {
    static constexpr int BT = 32 * WT;
    static constexpr int RSTR = BT + 16;              // floats per k of the [k][row] layout (16 mod 32 banks: the two k of a half-wave do not collide)
    static constexpr int PER = BT * GF_KC / GF_T;     // elements per thread, operand and chunk
    static constexpr int KPT = GF_T / BT;             // row-contiguous staging: k rows covered by the 256 threads at once
};

// Staging map of one operand tile (BT rows x GF_KC k): thread t owns PER elements (row r0 + j dr, k k0 + j dk).
//   k-contiguous operand:   k = t & 31 (fixed), rows (t >> 5) + 8 j       -- a wave reads two 128-byte row pieces per instruction
//   row-contiguous operand: row = t % BT (fixed), k = t / BT + KPT j      -- a wave reads 256 contiguous bytes per instruction
template <bool KFAST, int WT>
struct GfMap {
    static constexpr int dr = KFAST ? GF_T / GF_KC : 0, dk = KFAST ? 0 : GfCfg<WT>::KPT;
    static __device__ __forceinline__ int r0(int t) { return KFAST ? (t >> 5) : (t & (GfCfg<WT>::BT - 1)); }
    static __device__ __forceinline__ int k0(int t) { return KFAST ? (t & (GF_KC - 1)) : (t / GfCfg<WT>::BT); }
    static __device__ __forceinline__ int lds(int r, int k) { return KFAST ? r * GF_KSTR + k : k * GfCfg<WT>::RSTR + r; }
};

// the eight k values of MFMA slot g (k = 8 g + s, s = k-step) of tile row `row`
template <bool KFAST, int WT>
__device__ __forceinline__ void gf_frag(const float* tile, int row, int g, float* f) {
    if constexpr (KFAST) {
        const f32x4 lo = *(const f32x4*)(tile + row * GF_KSTR + 8 * g), hi = *(const f32x4*)(tile + row * GF_KSTR + 8 * g + 4);
        f[0] = lo[0], f[1] = lo[1], f[2] = lo[2], f[3] = lo[3], f[4] = hi[0], f[5] = hi[1], f[6] = hi[2], f[7] = hi[3];
    } else {
#pragma unroll
        for (int s = 0; s < 8; ++s) f[s] = tile[(8 * g + s) * GfCfg<WT>::RSTR + row];
    }
}

// KA / KB: the operand's k direction is the contiguous one (A row-major; B "column-major" = the transposed view of a row-major matrix)
template <bool KA, bool KB, int WT>
__global__ __launch_bounds__(GF_T, 2) void gemm_f32_kernel(cad_gemm_f32_args a) {
    typedef GfCfg<WT> C;
    typedef GfMap<KA, WT> MA;
    typedef GfMap<KB, WT> MB;
    constexpr int PER = C::PER, WS = 16 * WT;  // WS: rows (= columns) of a wave's share
    __shared__ __attribute__((aligned(16))) float As[KA ? C::BT * GF_KSTR : GF_KC * C::RSTR];
    __shared__ __attribute__((aligned(16))) float Bs[KB ? C::BT * GF_KSTR : GF_KC * C::RSTR];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int g = lane >> 4, jl = lane & 15;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * C::BT, n0 = (int64_t)blockIdx.x * C::BT;
    // per-thread staging bases and increments (elements)
    const int ar0 = MA::r0(t), ak0 = MA::k0(t), br0 = MB::r0(t), bk0 = MB::k0(t);
    const float* pa = a.A + (int64_t)blockIdx.z * a.a_bs + (m0 + ar0) * a.a_rs + ak0 * a.a_cs;
    const float* pb = a.B + (int64_t)blockIdx.z * a.b_bs + (n0 + br0) * a.b_cs + bk0 * a.b_rs;
    const int64_t astep = MA::dr * a.a_rs + MA::dk * a.a_cs, bstep = MB::dr * a.b_cs + MB::dk * a.b_rs;
    const int64_t arows = a.M - m0 - ar0, brows = a.N - n0 - br0;  // element j is inside the matrix while j dr < rows ...
    float ra[PER], rb[PER];
    auto fetch = [&](int64_t k0) {
        const float* qa = pa + k0 * a.a_cs;
        const float* qb = pb + k0 * a.b_rs;
        const int64_t aks = a.K - k0 - ak0, bks = a.K - k0 - bk0;  // ... and j dk < ks
#pragma unroll
        for (int j = 0; j < PER; ++j) {  // (running pointers: one 64-bit add per element instead of PER live addresses)
            ra[j] = (j * MA::dr < arows && j * MA::dk < aks) ? *qa : 0.f;
            rb[j] = (j * MB::dr < brows && j * MB::dk < bks) ? *qb : 0.f;
            qa += astep, qb += bstep;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            As[MA::lds(ar0 + j * MA::dr, ak0 + j * MA::dk)] = ra[j];
            Bs[MB::lds(br0 + j * MB::dr, bk0 + j * MB::dk)] = rb[j];
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
        // the B fragments of the wave's WT column tiles stay in registers for the chunk, the A fragments come one row tile at a time
        float bf[WT][8];
#pragma unroll
        for (int j = 0; j < WT; ++j) gf_frag<KB, WT>(Bs, wn * WS + 16 * j + jl, g, bf[j]);
#pragma unroll
        for (int i = 0; i < WT; ++i) {
            float af[8];
            gf_frag<KA, WT>(As, wm * WS + 16 * i + jl, g, af);
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int j = 0; j < WT; ++j) acc[i][j] = cad_mfma_16x16x4_f32(af[s], bf[j][s], acc[i][j]);
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
