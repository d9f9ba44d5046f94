// Dense projections of the Mamba mixer on the matrix cores (include/caduceus_hip.h, cad_proj_*): hand-written bf16 MFMA
// kernels for the skinny GEMMs around the scan -- in_proj / out_proj and their gradients -- whose small dimension
// (K = d_model or 2 d_model, a few hundred) makes them HBM-bound streaming problems: every activation byte is touched
// once, the weights stay on chip for the whole launch.
//
// cad_proj_wxT:   out (M, T) channel-major  =  W (M, K)  .  X (T, K)^T          (in_proj;  d(y) = W_out^T . dout^T)
//   * "W-stationary": a workgroup owns 128 * MB rows of W; every wave keeps its 16 * MB rows as MFMA B-fragments in
//     registers for the whole launch (MB * K / 32 * 4 VGPRs) and the workgroup walks a contiguous range of tokens;
//   * X travels global -> LDS by LDS-DMA (cad_glds16) in blocks of 64 tokens, double buffered, one barrier per block; the
//     16-byte pieces of a token row are XOR-swizzled through the per-lane SOURCE address (piece ^ (token & 15)), which
//     makes the ds_read_b128 of the A-fragments (16 tokens x 4 k-groups per instruction) bank-conflict-free;
//   * v_mfma_f32_16x16x32_bf16 with A = X fragment (rows = tokens), B = W fragment (columns = output channels): a D lane
//     then holds FOUR CONSECUTIVE TOKENS of one output channel -- exactly what the channel-major layout wants -- and
//     writes them (packed bf16) into a wave-private LDS tile, from which full 128-byte token runs go to HBM with
//     16-byte stores.
// The per-token arithmetic is independent of the token's position (fixed k order inside the MFMA), so both strands and
// both directions of the t-frame get bit-identical projections: RC-equivariance stays exact.
#include "cad_common.h"

namespace {

#define GP_WAVES 8

template <int KS>
struct GpCfg {
    static constexpr int MB = KS >= 16 ? 2 : 4;      // 16-row blocks of W per wave: MB * KS * 4 <= 128 VGPRs
    static constexpr int MW = 16 * MB;               // output channels per wave
    static constexpr int MWG = MW * GP_WAVES;        // ... per workgroup
    static constexpr int NT = KS >= 16 ? 32 : 64;    // tokens per block (two blocks + the staging tiles fit in 160 KB)
    static constexpr int ROWB = KS * 64;             // bytes per token row (K bf16)
    static constexpr int PPR = KS * 4;               // 16-byte pieces per token row
    static constexpr int SW = (PPR < 16 ? PPR : 16) - 1;  // swizzle mask: piece ^= token & SW
    static constexpr int XBUF = NT * ROWB;           // bytes per X block
    static constexpr int SSTR = NT * 2 + 16;         // bytes per output-channel row of the staging tile (16-byte aligned, skewed)
    static constexpr int STAGE = MW * SSTR;
    static constexpr size_t LDS = 2 * (size_t)XBUF + (size_t)GP_WAVES * STAGE;
};

// issue the LDS-DMA of one NT-token block of X (tokens t0 .. t0 + NT - 1; rows beyond T re-read row T - 1, never stored)
template <int KS>
__device__ __forceinline__ void gp_issue_block(const bf16_t* X, int64_t ldx, int64_t t0, int64_t T, char* xbuf, int wave, int lane) {
    typedef GpCfg<KS> C;
    constexpr int PIECES = C::NT * C::PPR;           // per block
    constexpr int INSTR = PIECES / 64;               // DMA instructions per block (64 pieces each)
#pragma unroll
    for (int it = 0; it < (INSTR + GP_WAVES - 1) / GP_WAVES; ++it) {
        const int ins = it * GP_WAVES + wave;        // wave-uniform
        if (ins < INSTR) {
            const int p0 = ins * 64;
            const int p = p0 + lane;
            const int t = p / C::PPR, ps = p % C::PPR;           // token row, PHYSICAL piece slot
            const int s = (ps & ~C::SW) | ((ps ^ t) & C::SW);    // logical piece
            int64_t tok = t0 + t;
            tok = tok < T ? tok : T - 1;
            cad_glds16((const char*)(X + tok * ldx) + s * 16, cad_uniform((int)(cad_lds_off(xbuf) + p0 * 16)));
        }
    }
}

__device__ __forceinline__ void gp_wait_dma() {
#ifndef CAD_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

template <int KS>
__global__ __launch_bounds__(64 * GP_WAVES, 2) void proj_wxT_kernel(cad_proj_args a) {
    typedef GpCfg<KS> C;
    CAD_DYN_SMEM(char, smem);
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const int g = lane >> 4, jl = lane & 15;
    const bf16_t* W = (const bf16_t*)a.W;
    const bf16_t* X = (const bf16_t*)a.X;
    bf16_t* out = (bf16_t*)a.out;
    const int64_t T = a.T;
    const int M = a.M;
    const int m_wave = blockIdx.y * C::MWG + wave * C::MW;  // first output channel of this wave
    // token blocks of this workgroup: b = blockIdx.x, + gridDim.x, ...  Interleaved on purpose: the workgroups that run
    // at the same time then read / write NEIGHBOURING 64-token runs of every row (HBM page locality: a channel-major
    // output row receives one contiguous multi-KB region from the whole grid instead of 128-byte pieces 4 KB apart)
    const int64_t nblk = (T + C::NT - 1) / C::NT;
    const int64_t b0 = blockIdx.x, bstep = gridDim.x;
    if (b0 >= nblk) return;
    char* xb[2] = {smem, smem + C::XBUF};
    char* stage = smem + 2 * C::XBUF + wave * C::STAGE;

    gp_issue_block<KS>(X, a.ldx, b0 * C::NT, T, xb[0], wave, lane);
    // B fragments of this wave's rows of W, resident for the whole launch (rows >= M read as zero)
    u32x4 wf[C::MB][KS];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
        const int m = m_wave + mb * 16 + jl;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (m < M) v = *(const u32x4*)(W + (int64_t)m * a.ldw + ks * 32 + g * 8);
            wf[mb][ks] = v;
        }
    }
    gp_wait_dma();
    __syncthreads();

    int cur = 0;
    for (int64_t b = b0; b < nblk; b += bstep, cur ^= 1) {
        if (b + bstep < nblk) gp_issue_block<KS>(X, a.ldx, (b + bstep) * C::NT, T, xb[cur ^ 1], wave, lane);
        const char* xt = xb[cur];
#pragma unroll
        for (int q = 0; q < C::NT / 16; ++q) {  // 16-token sub-blocks
            // A fragments: token t = 16 q + jl, k = 32 ks + 8 g .. + 7  ->  logical piece 4 ks + g, swizzled with the token
            u32x4 xf[KS];
            const int t = q * 16 + jl;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int s = ks * 4 + g;
                const int ps = (s & ~C::SW) | ((s ^ t) & C::SW);
                xf[ks] = *(const u32x4*)(xt + t * C::ROWB + ps * 16);
            }
            f32x4 d[C::MB];
#pragma unroll
            for (int mb = 0; mb < C::MB; ++mb) d[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int mb = 0; mb < C::MB; ++mb) d[mb] = cad_mfma_16x16x32_bf16(xf[ks], wf[mb][ks], d[mb]);
            }
            // lane (m = mb * 16 + jl, g) holds tokens 16 q + 4 g .. + 3 of channel m: 8 bytes into the staging tile
#pragma unroll
            for (int mb = 0; mb < C::MB; ++mb) {
                u32x2 pk;
                pk[0] = cad_pack_bf16x2_safe(d[mb][0], d[mb][1]);
                pk[1] = cad_pack_bf16x2_safe(d[mb][2], d[mb][3]);
                *(u32x2*)(stage + (mb * 16 + jl) * C::SSTR + (q * 16 + g * 4) * 2) = pk;
            }
        }
        // (the staging tile is written as 8-byte and read as 16-byte vectors: distinct types for the compiler's alias
        // analysis, which may otherwise move the reads above the writes -- on the host emulator build as well)
        asm volatile("" ::: "memory");
        cad_wave_sync();
        // the next block has landed (this wave's share; the barrier below publishes everybody's).  Waited for BEFORE this
        // block's stores are issued, so that the wait never sits behind fresh write acknowledgements.
        gp_wait_dma();
        // staging tile -> HBM: NT / 8 lanes cover the tokens of one channel row (16 bytes each)
        const int64_t t0 = b * C::NT;
        constexpr int LPR = C::NT / 8, RPI = 64 / LPR;  // lanes per row, rows per instruction
#pragma unroll
        for (int r0 = 0; r0 < C::MW; r0 += RPI) {
            const int r = r0 + lane / LPR, c8 = lane % LPR;
            const u32x4 v = *(const u32x4*)(stage + r * C::SSTR + c8 * 16);
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
        __syncthreads();  // publishes the next block; orders this block's LDS reads (X tile, staging tile) before their reuse
    }
}

}  // namespace

// more than 64 KB of dynamic LDS has to be requested per kernel (once)
#if defined(CAD_EMU)
#define GP_BIG_LDS(kern, bytes) (void)0
#else
#define GP_BIG_LDS(kern, bytes)                                                                                      \
    do {                                                                                                             \
        static bool done = false;                                                                                    \
        if ((bytes) > 65536 && !done) {                                                                              \
            if (hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != \
                hipSuccess)                                                                                          \
                return CAD_ERR_LAUNCH;                                                                               \
            done = true;                                                                                             \
        }                                                                                                            \
    } while (0)
#endif

extern "C" int cad_proj_supported(int K) { return K == 32 || K == 64 || K == 128 || K == 256 || K == 512; }

template <int KS>
static int launch_wxT(const cad_proj_args* a, void* stream) {
    typedef GpCfg<KS> C;
    const int64_t nblk = (a->T + C::NT - 1) / C::NT;
    const int my = (a->M + C::MWG - 1) / C::MWG;
    int64_t gx = 256 / my;  // ~ one workgroup per CU
    if (gx < 1) gx = 1;
    if (gx > nblk) gx = nblk;
    dim3 grid((unsigned)gx, (unsigned)my), block(64 * GP_WAVES);
    GP_BIG_LDS((proj_wxT_kernel<KS>), C::LDS);
    CAD_LAUNCH((proj_wxT_kernel<KS>), grid, block, C::LDS, stream, *a);
    return cad_after_launch();
}

extern "C" int cad_proj_wxT(const cad_proj_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->W && a->X && a->out && a->T > 0 && a->M > 0 && a->K > 0);
    CAD_CHECK_ARG(a->ldw >= a->K && a->ldx >= a->K && a->ldo >= a->T);
    CAD_CHECK_ARG((a->ldw % 8) == 0 && (a->ldx % 8) == 0 && (((uintptr_t)a->W | (uintptr_t)a->X) % 16) == 0);
    CadProfScope prof(8, stream);
    switch (a->K) {
        case 32: return launch_wxT<1>(a, stream);
        case 64: return launch_wxT<2>(a, stream);
        case 128: return launch_wxT<4>(a, stream);
        case 256: return launch_wxT<8>(a, stream);
        case 512: return launch_wxT<16>(a, stream);
        default: return CAD_ERR_UNSUPPORTED;
    }
}
