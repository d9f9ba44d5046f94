// fp8 (OCP e4m3) in_proj on the matrix cores (include/caduceus_hip.h: cad_quant_rows_fp8, cad_proj_wxT_fp8) -- BASELINE
// configs[4] "fp8 MFMA projections".  Same skeleton as proj_wxT_kernel (gemm.hip): W-stationary B fragments in registers
// for the whole launch, X by LDS-DMA in 64-token blocks (double buffered, XOR-swizzled 16-byte pieces), a D lane's four
// consecutive tokens through a per-wave staging tile into 16-byte channel-major stores -- with
//   * v_mfma_f32_16x16x32_fp8_fp8: the same 32-deep K step per instruction as the bf16 form at half the operand bytes (an
//     A / B fragment is 8 e4m3 values = one 64-bit register pair) and twice the matrix-core rate;
//   * per-TOKEN activation scales (written by cad_quant_rows_fp8) and per-ROW weight scales applied to the fp32 accumulators
//     in the epilogue, so a token's result depends on nothing but that token (position independence: the t-frame strands /
//     directions stay bit-identical, RC-equivariance exact).
#include "cad_common.h"
#include "cad_fp8.h"
#include "cad_stream.h"

namespace {

#define GF_WAVES 8


// ---- per-token quantisation: one wave per row --------------------------------------------------------------------------
template <typename T, int EPL /* elements per lane: K = 64 * EPL */>
__global__ __launch_bounds__(256) void quant_rows_fp8_kernel(cad_quant_fp8_args a) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.T) return;
    const T* x = (const T*)a.x + row * a.ldx + lane * EPL;
    float v[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) v[i] = to_f32(x[i]);
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) m = fmaxf(m, fabsf(v[i]));
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    const float scale = cad_fp8_row_scale(m);  // clamped like the weight-side quantiser (ops.quant_weight_fp8)
    const float inv = 1.0f / scale;
    uint32_t* q = (uint32_t*)((uint8_t*)a.q + row * a.ldq + lane * EPL);
#pragma unroll
    for (int i = 0; i < EPL; i += 4) q[i / 4] = cad_pack_fp8x4(v[i] * inv, v[i + 1] * inv, v[i + 2] * inv, v[i + 3] * inv);
    if (lane == 0) a.scale[row] = scale;
}

// ---- out (M, T) bf16 = (Wq . Xq^T) * sw[m] * sx[t] -----------------------------------------------------------------------
template <int KS>
struct GfCfg {
    static constexpr int MB = 4;                      // 16-row blocks of W per wave: MB * KS * 2 <= 128 VGPRs
    static constexpr int MW = 16 * MB, MWG = MW * GF_WAVES;
    static constexpr int NT = 64;                     // tokens per block
    static constexpr int ROWB = KS * 32;              // bytes per token row (K e4m3)
    static constexpr int PPR = KS * 2;                // 16-byte pieces per token row
    static constexpr int SW = (PPR < 16 ? PPR : 16) - 1;
    static constexpr int XBUF = NT * ROWB;
    static constexpr int SSTR = NT * 2 + 16;          // bytes per output-channel row of the (bf16) staging tile
    static constexpr int STAGE = MW * SSTR;
    static constexpr size_t LDS = 2 * (size_t)XBUF + (size_t)GF_WAVES * STAGE;
};

template <int KS>
__device__ __forceinline__ void gf_issue_block(const uint8_t* X, int64_t ldx, int64_t t0, int64_t T, char* xbuf, int wave, int lane) {
    typedef GfCfg<KS> C;
    constexpr int PIECES = C::NT * C::PPR, INSTR = PIECES / 64;
#pragma unroll
    for (int it = 0; it < (INSTR + GF_WAVES - 1) / GF_WAVES; ++it) {
        const int ins = it * GF_WAVES + wave;  // wave-uniform
        if (ins < INSTR) {
            const int p = ins * 64 + lane;
            const int t = p / C::PPR, ps = p % C::PPR;           // token row, PHYSICAL piece slot
            const int s = (ps & ~C::SW) | ((ps ^ t) & C::SW);    // logical piece
            int64_t tok = t0 + t;
            tok = tok < T ? tok : T - 1;
            cad_glds16(X + tok * ldx + s * 16, cad_uniform((int)(cad_lds_off(xbuf) + ins * 1024)));
        }
    }
}

__device__ __forceinline__ void gf_wait_dma() { cad_wait_vmcnt<0>(); }

template <int KS>
__global__ __launch_bounds__(64 * GF_WAVES, 2) void proj_wxT_fp8_kernel(cad_proj_fp8_args a) {
    typedef GfCfg<KS> C;
    CAD_DYN_SMEM(char, smem);
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const int g = lane >> 4, jl = lane & 15;
    const uint8_t* W = (const uint8_t*)a.Wq;
    const uint8_t* X = (const uint8_t*)a.Xq;
    bf16_t* out = (bf16_t*)a.out;
    const int64_t T = a.T;
    const int M = a.M;
    const int m_wave = blockIdx.y * C::MWG + wave * C::MW;
    const int64_t nblk = (T + C::NT - 1) / C::NT;
    const int64_t b0 = blockIdx.x, bstep = gridDim.x;  // interleaved token blocks (see proj_wxT_kernel)
    if (b0 >= nblk) return;
    char* xb[2] = {smem, smem + C::XBUF};
    char* stage = smem + 2 * C::XBUF + wave * C::STAGE;

    gf_issue_block<KS>(X, a.ldx, b0 * C::NT, T, xb[0], wave, lane);
    u32x2 wf[C::MB][KS];
    float swr[C::MB];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
        const int m = m_wave + mb * 16 + jl;
        swr[mb] = m < M ? a.sw[m] : 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x2 v = {0u, 0u};
            if (m < M) v = *(const u32x2*)(W + (int64_t)m * a.ldw + ks * 32 + g * 8);
            wf[mb][ks] = v;
        }
    }
    gf_wait_dma();
    __syncthreads();

    int cur = 0;
    for (int64_t b = b0; b < nblk; b += bstep, cur ^= 1) {
        if (b + bstep < nblk) gf_issue_block<KS>(X, a.ldx, (b + bstep) * C::NT, T, xb[cur ^ 1], wave, lane);
        const char* xt = xb[cur];
        const int64_t t0 = b * C::NT;
        // A fragments of sub-block q: token t = 16 q + jl, k = 32 ks + 8 g .. + 7: byte offset 32 ks + 8 g = piece 2 ks + (g >> 1),
        // half g & 1; double buffered in registers (the reads of sub-block q + 1 run under the MFMAs of sub-block q)
        u32x2 xfb[2][KS];
        auto load_frags = [&](int q, u32x2* dst) {
            const int t = q * 16 + jl;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int s = ks * 2 + (g >> 1);
                const int ps = (s & ~C::SW) | ((s ^ t) & C::SW);
                dst[ks] = *(const u32x2*)(xt + t * C::ROWB + ps * 16 + (g & 1) * 8);
            }
        };
        load_frags(0, xfb[0]);
#pragma unroll
        for (int q = 0; q < C::NT / 16; ++q) {
            const u32x2* xf = xfb[q & 1];
            if (q + 1 < C::NT / 16) load_frags(q + 1, xfb[(q + 1) & 1]);
            f32x4 d[C::MB];
#pragma unroll
            for (int mb = 0; mb < C::MB; ++mb) d[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int mb = 0; mb < C::MB; ++mb) d[mb] = cad_mfma_16x16x32_fp8(xf[ks], wf[mb][ks], d[mb]);
            }
            // de-quantise: this lane's four consecutive tokens 16 q + 4 g .. + 3 (their scales; rows beyond T are never stored)
            float sxv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t tok = t0 + q * 16 + g * 4 + r;
                sxv[r] = a.sx[tok < T ? tok : T - 1];
            }
#pragma unroll
            for (int mb = 0; mb < C::MB; ++mb) {
                u32x2 pk;
                pk[0] = cad_pack_bf16x2_safe(d[mb][0] * (swr[mb] * sxv[0]), d[mb][1] * (swr[mb] * sxv[1]));
                pk[1] = cad_pack_bf16x2_safe(d[mb][2] * (swr[mb] * sxv[2]), d[mb][3] * (swr[mb] * sxv[3]));
                *(u32x2*)(stage + (mb * 16 + jl) * C::SSTR + (q * 16 + g * 4) * 2) = pk;
            }
        }
        asm volatile("" ::: "memory");
        cad_wave_sync();
        gf_wait_dma();
        constexpr int LPR = C::NT / 8, RPI = 64 / LPR;  // lanes per row, rows per instruction
        u32x4 sv[C::MW / RPI];  // all rows read from the tile first, stored afterwards (see proj_wxT_kernel)
#pragma unroll
        for (int r0 = 0; r0 < C::MW; r0 += RPI)
            sv[r0 / RPI] = *(const u32x4*)(stage + (r0 + lane / LPR) * C::SSTR + (lane % LPR) * 16);
        const bool fast = t0 + C::NT <= T && m_wave + C::MW <= M && (a.ldo % 8) == 0 && (((uintptr_t)out) & 15) == 0;  // wave-uniform
        if (fast) {
#pragma unroll
            for (int r0 = 0; r0 < C::MW; r0 += RPI)
                cad_store_stream<CAD_STREAM_PROJ>((u32x4*)(out + (int64_t)(m_wave + r0 + lane / LPR) * a.ldo + t0 + (lane % LPR) * 8),
                                                  sv[r0 / RPI]);
        } else {
#pragma unroll
            for (int r0 = 0; r0 < C::MW; r0 += RPI) {
                const int r = r0 + lane / LPR, c8 = lane % LPR;
                const u32x4 v = sv[r0 / RPI];
                const int m = m_wave + r;
                const int64_t t = t0 + c8 * 8;
                if (m < M) {
                    bf16_t* dst = out + (int64_t)m * a.ldo + t;
                    if (t + 8 <= T && (((uintptr_t)dst) & 15) == 0) {
                        *(u32x4*)dst = v;
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (t + e < T) dst[e].v = (uint16_t)(v[e >> 1] >> (16 * (e & 1)));
                    }
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace

#define GF_BIG_LDS(kern, bytes) CAD_BIG_LDS(kern, bytes)

extern "C" int cad_proj_fp8_supported(int K) { return K == 256 || K == 512; }

template <typename T, int EPL>
static int launch_quant(const cad_quant_fp8_args* a, void* stream) {
    dim3 grid((unsigned)((a->T + 3) / 4)), block(256);
    CAD_LAUNCH((quant_rows_fp8_kernel<T, EPL>), grid, block, 0, stream, *a);
    return cad_after_launch();
}

extern "C" int cad_quant_rows_fp8(const cad_quant_fp8_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->x && a->q && a->scale && a->T > 0 && a->T <= 0x7fffffffLL * 4);
    CAD_CHECK_ARG(cad_proj_fp8_supported(a->K) && a->ldx >= a->K && a->ldq >= a->K && (a->ldq % 4) == 0 &&
                  ((uintptr_t)a->q % 4) == 0);
    CadProfScope prof(8, stream);
#define GF_Q(T) return a->K == 256 ? launch_quant<T, 4>(a, stream) : launch_quant<T, 8>(a, stream);
    if (a->dtype == CAD_BF16) {
        GF_Q(bf16_t)
    } else if (a->dtype == CAD_F32) {
        GF_Q(float)
    }
#undef GF_Q
    return CAD_ERR_UNSUPPORTED;
}

template <int KS>
static int launch_fp8(const cad_proj_fp8_args* a, void* stream) {
    typedef GfCfg<KS> C;
    const int64_t nblk = (a->T + C::NT - 1) / C::NT;
    const int my = (a->M + C::MWG - 1) / C::MWG;
    int64_t gx = cad_cu_count() / my;
    if (gx < 1) gx = 1;
    if (gx > nblk) gx = nblk;
    dim3 grid((unsigned)gx, (unsigned)my), block(64 * GF_WAVES);
    GF_BIG_LDS((proj_wxT_fp8_kernel<KS>), C::LDS);
    CAD_LAUNCH((proj_wxT_fp8_kernel<KS>), grid, block, C::LDS, stream, *a);
    return cad_after_launch();
}

extern "C" int cad_proj_wxT_fp8(const cad_proj_fp8_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->Wq && a->Xq && a->sw && a->sx && a->out && a->T > 0 && a->M > 0);
    CAD_CHECK_ARG(cad_proj_fp8_supported(a->K) && a->ldw >= a->K && a->ldx >= a->K && a->ldo >= a->T);
    CAD_CHECK_ARG((a->ldw % 16) == 0 && (a->ldx % 16) == 0 && (((uintptr_t)a->Wq | (uintptr_t)a->Xq) % 16) == 0);
    CadProfScope prof(8, stream);
    switch (a->K) {
        case 256: return launch_fp8<8>(a, stream);
        case 512: return launch_fp8<16>(a, stream);
        default: return CAD_ERR_UNSUPPORTED;
    }
}
