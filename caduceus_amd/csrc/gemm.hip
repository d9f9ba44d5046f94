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
#include "cad_stream.h"

namespace {

#define GP_WAVES 8

// K = 512 (d_model 512, configs[4]) tuning knobs: W rows per wave / tokens per block / register double-buffering of the A fragments
#ifndef GP_MB16
#define GP_MB16 2
#endif
#ifndef GP_NT16
#define GP_NT16 32
#endif
#ifndef GP_XFB16
#define GP_XFB16 1
#endif
template <int KS>
struct GpCfg {
    static constexpr int MB = KS >= 16 ? GP_MB16 : 4;  // 16-row blocks of W per wave: MB * KS * 4 <= 128 VGPRs
    static constexpr int MW = 16 * MB;               // output channels per wave
    static constexpr int MWG = MW * GP_WAVES;        // ... per workgroup
    static constexpr int NT = KS >= 16 ? GP_NT16 : 64;  // tokens per block (two blocks + the staging tiles fit in 160 KB)
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

__device__ __forceinline__ void gp_wait_dma() { cad_wait_vmcnt<0>(); }

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
        // A fragments of sub-block q: token t = 16 q + jl, k = 32 ks + 8 g .. + 7  ->  logical piece 4 ks + g, swizzled with the token.
        // Double buffered in registers where they fit (KS <= 8): the reads of sub-block q + 1 are issued before the MFMAs of
        // sub-block q, so that the matrix cores do not idle for an LDS round trip four times per block.
        constexpr int XFB = KS <= 8 ? 2 : GP_XFB16;
        u32x4 xfb[XFB][KS];
        auto load_frags = [&](int q, u32x4* dst) {
            const int t = q * 16 + jl;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int s = ks * 4 + g;
                const int ps = (s & ~C::SW) | ((s ^ t) & C::SW);
                dst[ks] = *(const u32x4*)(xt + t * C::ROWB + ps * 16);
            }
        };
        if constexpr (XFB == 2) load_frags(0, xfb[0]);
#pragma unroll
        for (int q = 0; q < C::NT / 16; ++q) {  // 16-token sub-blocks
            u32x4* xf = xfb[XFB == 2 ? (q & 1) : 0];
            if constexpr (XFB == 2) {
                if (q + 1 < C::NT / 16) load_frags(q + 1, xfb[(q + 1) & 1]);
            } else {
                load_frags(q, xf);
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
        // staging tile -> HBM: NT / 8 lanes cover the tokens of one channel row (16 bytes each).  All rows are read from the tile
        // first and stored afterwards (one LDS round trip per block instead of one per row group); whole blocks of a fully
        // populated, 16-byte aligned output take the straight-line path.
        const int64_t t0 = b * C::NT;
        constexpr int LPR = C::NT / 8, RPI = 64 / LPR;  // lanes per row, rows per instruction
        u32x4 sv[C::MW / RPI];
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
        __syncthreads();  // publishes the next block; orders this block's LDS reads (X tile, staging tile) before their reuse
    }
}

// ---- cad_proj_wx:  out (M, T) = W (M, K) . X (K, T) [+ acc (M, T)],  all channel-major, small K (<= 64) -------------------
// dt_proj (K = dt_rank) and the x_proj input gradient d(xc) = du + W_x^T . d(dbc) (K = dt_rank + 2 d_state): the
// streaming operands are the (M, T) output (and addend); X is a thin (K, T) panel.  Same skeleton as proj_wxT_kernel --
// W-stationary B-fragments, 64-token blocks by LDS-DMA, the D lanes' four consecutive tokens through the per-wave staging
// tile -- except that X is token-contiguous, so the A-fragments (consecutive k per lane) are TRANSPOSED on the way out
// of LDS: ds_read_b64_tr_b16 on the [k][token] tile (cad_lds_read_tr16), two reads per 8 k.  Rows k >= K of the tile are
// zero (written once), so K is padded to 32 / 64 for free.  The 32-byte token segments of a tile row are XOR-swizzled
// with the row (through the DMA source address) so that the eight rows a transposing read touches hit distinct banks.
template <int KS>
struct GxCfg {
    static constexpr int MB = 4, MW = 64, MWG = MW * GP_WAVES, NT = 64;
    static constexpr int KP = 32 * KS;               // tile rows (K padded)
    static constexpr int XROW = NT * 2;              // bytes per tile row
    static constexpr int XBUF = KP * XROW;
    static constexpr int SSTR = NT * 2 + 16;
    static constexpr int STAGE = MW * SSTR;
    static constexpr size_t LDS = 2 * (size_t)XBUF + (size_t)GP_WAVES * STAGE;
};
__device__ __forceinline__ int gx_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

template <int KS>
__device__ __forceinline__ void gx_issue_block(const bf16_t* X, int64_t ldx, int K, int64_t t0, int64_t T, char* xbuf,
                                               int wave, int lane) {
    const int ninstr = K / 8;  // K rows x 8 pieces of 16 bytes, 64 pieces per DMA instruction
    for (int ins = wave; ins < ninstr; ins += GP_WAVES) {  // wave-uniform
        const int p = ins * 64 + lane;
        const int row = p >> 3, pp = p & 7;                        // tile row, PHYSICAL piece
        const int lp = (((pp >> 1) ^ gx_swz(row)) << 1) | (pp & 1);  // logical piece = 8 tokens
        int64_t tok = t0 + lp * 8;
        tok = tok + 8 <= T ? tok : T - 8;                          // tail block: valid data, never stored
        cad_glds16(X + (int64_t)row * ldx + tok, cad_uniform((int)(cad_lds_off(xbuf) + ins * 1024)));
    }
}

template <int KS, bool ACC>
__global__ __launch_bounds__(64 * GP_WAVES, 2) void proj_wx_kernel(cad_proj_args a) {
    typedef GxCfg<KS> C;
    CAD_DYN_SMEM(char, smem);
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const int g = lane >> 4, jl = lane & 15;
    const bf16_t* W = (const bf16_t*)a.W;
    const bf16_t* X = (const bf16_t*)a.X;
    const bf16_t* acc = (const bf16_t*)a.acc;
    bf16_t* out = (bf16_t*)a.out;
    const int64_t T = a.T;
    const int M = a.M, K = a.K;
    const int m_wave = blockIdx.y * C::MWG + wave * C::MW;
    const int64_t nblk = (T + C::NT - 1) / C::NT;
    const int64_t b0 = blockIdx.x, bstep = gridDim.x;
    if (b0 >= nblk) return;
    char* xb[2] = {smem, smem + C::XBUF};
    char* stage = smem + 2 * C::XBUF + wave * C::STAGE;
    // rows K .. KP-1 of both tiles stay zero for the whole launch
    for (int i = threadIdx.x * 16; i < 2 * C::XBUF; i += 64 * GP_WAVES * 16) {
        const int row = (i % C::XBUF) / C::XROW;
        if (row >= K) *(u32x4*)(smem + i) = u32x4{0u, 0u, 0u, 0u};
    }
    gx_issue_block<KS>(X, a.ldx, K, b0 * C::NT, T, xb[0], wave, lane);
    u32x4 wf[C::MB][KS];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
        const int m = m_wave + mb * 16 + jl;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            const int k0 = ks * 32 + g * 8;
            if (m < M && k0 + 8 <= K) v = *(const u32x4*)(W + (int64_t)m * a.ldw + k0);
            wf[mb][ks] = v;
        }
    }
    // optional epilogue: softplus(. + bias[m]) in fp32 (dt_proj + delta_bias + softplus in one pass); the D lane's channel is
    // m_wave + 16 mb + jl
    const bool act_sp = a.act == CAD_ACT_SOFTPLUS_BIAS;  // wave-uniform
    float brow[C::MB];
#pragma unroll
    for (int mb = 0; mb < C::MB; ++mb) {
        const int m = m_wave + mb * 16 + jl;
        brow[mb] = (act_sp && a.bias && m < M) ? a.bias[m] : 0.f;
    }
    gp_wait_dma();
    __syncthreads();

    int cur = 0;
    for (int64_t b = b0; b < nblk; b += bstep, cur ^= 1) {
        if (b + bstep < nblk) gx_issue_block<KS>(X, a.ldx, K, (b + bstep) * C::NT, T, xb[cur ^ 1], wave, lane);
        const char* xt = xb[cur];
        const int64_t t0 = b * C::NT;
        constexpr int LPR = C::NT / 8, RPI = 64 / LPR;
        // the addend rows of this block (plain loads, issued ahead of the MFMA phase)
        u32x4 old[C::MW / RPI];
        if constexpr (ACC) {
#pragma unroll
            for (int r0 = 0; r0 < C::MW; r0 += RPI) {
                const int m = m_wave + r0 + lane / LPR;
                const int64_t t = t0 + (lane % LPR) * 8;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (m < M && t + 8 <= T) v = *(const u32x4*)(acc + (int64_t)m * a.ldacc + t);
                old[r0 / RPI] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < C::NT / 16; ++q) {
            // A fragments by transposing reads: lane (token 16 q + jl, group g) gets k = 32 ks + 8 g .. + 7
            u32x4 xf[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int r0 = ks * 32 + g * 8 + (jl >> 2);  // this lane's SOURCE row of the first 4 x 16 block
                const char* p0 = xt + r0 * C::XROW + ((q ^ gx_swz(r0)) * 32) + (jl & 3) * 8;
                const char* p1 = xt + (r0 + 4) * C::XROW + ((q ^ gx_swz(r0 + 4)) * 32) + (jl & 3) * 8;
                const u32x2 lo = cad_lds_read_tr16(p0), hi = cad_lds_read_tr16(p1);
                xf[ks] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
            f32x4 d[C::MB];
#pragma unroll
            for (int mb = 0; mb < C::MB; ++mb) d[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int mb = 0; mb < C::MB; ++mb) d[mb] = cad_mfma_16x16x32_bf16(xf[ks], wf[mb][ks], d[mb]);
            }
            if (act_sp) {
#pragma unroll
                for (int mb = 0; mb < C::MB; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) d[mb][r] = cad_softplus_lowp(d[mb][r] + brow[mb]);
            }
#pragma unroll
            for (int mb = 0; mb < C::MB; ++mb) {
                u32x2 pk;
                pk[0] = cad_pack_bf16x2_safe(d[mb][0], d[mb][1]);
                pk[1] = cad_pack_bf16x2_safe(d[mb][2], d[mb][3]);
                *(u32x2*)(stage + (mb * 16 + jl) * C::SSTR + (q * 16 + g * 4) * 2) = pk;
            }
        }
        asm volatile("" ::: "memory");
        cad_wave_sync();
        gp_wait_dma();
#pragma unroll
        for (int r0 = 0; r0 < C::MW; r0 += RPI) {
            const int r = r0 + lane / LPR, c8 = lane % LPR;
            u32x4 v = *(const u32x4*)(stage + r * C::SSTR + c8 * 16);
            const int m = m_wave + r;
            const int64_t t = t0 + c8 * 8;
            if constexpr (ACC) {  // out = bf16(acc + bf16 product): element-wise on the packed pairs
                const u32x4 o = old[r0 / RPI];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = cad_bits2f(v[e] << 16) + cad_bits2f(o[e] << 16);
                    const float hi = cad_bits2f(v[e] & 0xFFFF0000u) + cad_bits2f(o[e] & 0xFFFF0000u);
                    v[e] = cad_pack_bf16x2(lo, hi);
                }
            }
            if (m < M && t + 8 <= T) cad_store_stream<CAD_STREAM_PROJ>((u32x4*)(out + (int64_t)m * a.ldo + t), v);
        }
        __syncthreads();
    }
}


// ---- cad_proj_wx, thin M / deep K:  out (M <= 64, T) = W (M, K) . X (K, T),  K a multiple of 64 -----------------------------
// x_proj (M = dt_rank + 2 d_state, K = d_inner) and d(dt_lr) = W_dt^T . d(delta) (M = dt_rank): the streaming operand is the
// (K, T) channel-major activation, read exactly once; W (<= 64 x K) lives in LDS for the whole launch.  A workgroup owns blocks of
// 128 tokens (one 16-token column per wave) and walks K in chunks of 64 rows through a RING of 4 LDS tiles filled by LDS-DMA
// three chunks ahead (counted s_waitcnt: vmcnt retires in order, two DMA instructions per wave and chunk); A fragments by
// transposing reads as above, B fragments (W) by ds_read_b128 from the padded LDS copy, fp32 accumulation over the whole K in
// the MFMA accumulators, 8-byte stores of four consecutive tokens per lane (the output is 1/10 of the traffic).
struct GtCfg {
    static constexpr int NT = 128, KC = 64, RING = 4;
    static constexpr int XROW = NT * 2;               // bytes per tile row
    static constexpr int XBUF = KC * XROW;            // 16 KB per chunk
    static constexpr int DPW = (KC * NT * 2 / 16 / 64) / GP_WAVES;  // DMA instructions per wave and chunk (= 2)
    static constexpr size_t lds(int M, int K) { return (size_t)RING * XBUF + (size_t)((M + 15) / 16 * 16) * (K * 2 + 16); }
};
static_assert(GtCfg::DPW == 2, "the counted waits below assume two DMA instructions per wave and chunk");
static_assert((GtCfg::RING & (GtCfg::RING - 1)) == 0, "ring slots are advanced with a mask");

__device__ __forceinline__ void gt_issue_chunk(const bf16_t* X, int64_t ldx, int k0, int64_t t0, int64_t T, char* xbuf, int wave,
                                               int lane) {
#pragma unroll
    for (int i = 0; i < GtCfg::DPW; ++i) {
        const int ins = wave * GtCfg::DPW + i;       // 16 instructions of 64 pieces: 4 tile rows each
        const int p = ins * 64 + lane;
        const int row = p >> 4, pp = p & 15;                         // tile row, PHYSICAL 16-byte piece
        const int lp = (((pp >> 1) ^ gx_swz(row)) << 1) | (pp & 1);  // logical piece = 8 tokens
        int64_t tok = t0 + lp * 8;
        tok = tok + 8 <= T ? tok : T - 8;                            // tail block: valid data, never stored
        cad_glds16(X + (int64_t)(k0 + row) * ldx + tok, cad_uniform((int)(cad_lds_off(xbuf) + ins * 1024)));
    }
}

template <int MB>
__global__ __launch_bounds__(64 * GP_WAVES, 1) void proj_wx_thin_kernel(cad_proj_args a) {
    typedef GtCfg C;
    CAD_DYN_SMEM(char, smem);
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const int g = lane >> 4, jl = lane & 15;
    const bf16_t* W = (const bf16_t*)a.W;
    const bf16_t* X = (const bf16_t*)a.X;
    bf16_t* out = (bf16_t*)a.out;
    const int64_t T = a.T;
    const int M = a.M, K = a.K;
    const int NCH = K / C::KC;
    const int64_t nblk = (T + C::NT - 1) / C::NT;
    const int64_t b0 = blockIdx.x, bstep = gridDim.x;
    if (b0 >= nblk) return;
    const int64_t nmine = (nblk - b0 + bstep - 1) / bstep;
    const int64_t total = nmine * NCH;               // (block, chunk) iterations of this workgroup
    char* wl = smem + C::RING * C::XBUF;             // W copy: row stride WSTR (padded: the 16 rows of a B fragment read spread over the banks)
    const int WSTR = K * 2 + 16;
    // the (block, chunk) position of the next chunk to issue and of the chunk being consumed are carried as counters (no 64-bit
    // division by the run-time NCH between the barrier and the next DMA issue; time-neutral when measured: this kernel's 88 us in the
    // step against 56 us in a loop is the memory-side cache, profiles/r03_ab_conv_addnorm.txt (3))
    int ich = 0, islot = 0;
    int64_t iblk = b0;
    auto issue_next = [&]() {
        gt_issue_chunk(X, a.ldx, ich * C::KC, iblk * C::NT, T, smem + islot * C::XBUF, wave, lane);
        islot = (islot + 1) & (C::RING - 1);
        if (++ich == NCH) {
            ich = 0;
            iblk += bstep;
        }
    };
    for (int64_t it = 0; it < C::RING - 1; ++it)
        if (it < total) issue_next();
    // W -> LDS (rows >= M zero)
    for (int i = threadIdx.x; i < MB * 16 * (K / 8); i += 64 * GP_WAVES) {
        const int m = i / (K / 8), c8 = i % (K / 8);
        u32x4 v = {0u, 0u, 0u, 0u};
        if (m < M) v = *(const u32x4*)(W + (int64_t)m * a.ldw + c8 * 8);
        *(u32x4*)(wl + m * WSTR + c8 * 16) = v;
    }
    f32x4 d[MB];
    int ch = 0, slot = 0;
    int64_t blk = b0;
    for (int64_t it = 0; it < total; ++it) {
        // chunk `it` has landed: of this wave's DMA, at most the two later chunks (4 instructions) may still be in flight --
        // fewer near the end of the stream, where everything is waited for
        if (it + C::RING - 2 < total)
            cad_wait_vmcnt<4>();
        else
            cad_wait_vmcnt<0>();
        __syncthreads();  // every wave's share of the chunk is visible; the tile consumed LAST iteration is free again
        if (it + C::RING - 1 < total) issue_next();
        if (ch == 0) {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) d[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const char* xt = smem + slot * C::XBUF;
#pragma unroll
        for (int ks = 0; ks < C::KC / 32; ++ks) {
            const int r0 = ks * 32 + g * 8 + (jl >> 2);
            const char* p0 = xt + r0 * C::XROW + ((wave ^ gx_swz(r0)) * 32) + (jl & 3) * 8;
            const char* p1 = xt + (r0 + 4) * C::XROW + ((wave ^ gx_swz(r0 + 4)) * 32) + (jl & 3) * 8;
            const u32x2 lo = cad_lds_read_tr16(p0), hi = cad_lds_read_tr16(p1);
            const u32x4 xf = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const u32x4 wf = *(const u32x4*)(wl + (mb * 16 + jl) * WSTR + (ch * C::KC + ks * 32 + g * 8) * 2);
                d[mb] = cad_mfma_16x16x32_bf16(xf, wf, d[mb]);
            }
        }
        slot = (slot + 1) & (C::RING - 1);
        if (++ch == NCH) {
            const int64_t t = blk * C::NT + wave * 16 + g * 4;  // this lane's four consecutive tokens
            const bf16_t* acc = (const bf16_t*)a.acc;  // wave-uniform: the other K half of a product too deep for one W copy in LDS
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = mb * 16 + jl;
                f32x4 v = d[mb];
                if (acc && m < M && t + 4 <= T) {  // out = bf16(acc + the fp32 sums): the addend is widened, the sum rounded once
                    const u32x2 o = *(const u32x2*)(acc + (int64_t)m * a.ldacc + t);
                    v[0] += cad_bits2f(o[0] << 16), v[1] += cad_bits2f(o[0] & 0xFFFF0000u);
                    v[2] += cad_bits2f(o[1] << 16), v[3] += cad_bits2f(o[1] & 0xFFFF0000u);
                }
                u32x2 pk;
                pk[0] = cad_pack_bf16x2_safe(v[0], v[1]);
                pk[1] = cad_pack_bf16x2_safe(v[2], v[3]);
                if (m < M && t + 4 <= T) *(u32x2*)(out + (int64_t)m * a.ldo + t) = pk;
            }
            ch = 0;
            blk += bstep;
        }
    }
}


// ---- cad_proj_wx_wgrad: the thin-M / deep-K product AND the weight gradient of the same operand from ONE pass over X -----------
// d(dt_lr) (M = dt_rank rows) = W_dt^T . d(delta)  and  dW_dt (K = d_inner, M) = d(delta) . dt_lr^T  both stream the (K, T) tensor
// d(delta); the second used to be a K-split hipBLASLt bmm + sum that read the 268 MB again.  Same skeleton as proj_wx_thin_kernel
// (ring of four [64 rows][128 tokens] LDS tiles filled by LDS-DMA, counted waits); on top of it, per (block, chunk) every wave
// runs two more MFMAs with the ROLES of the tile swapped: A = 16 channel rows of the tile x 32 tokens (token-contiguous rows are
// A fragments as they lie: plain ds_read_b128), B = the matching 32 tokens of the M rows of Y (a [M][128] tile per block, double
// buffered, XOR-swizzled through the DMA source addresses), accumulated in registers over ALL blocks of the workgroup -- NCH x MB
// accumulator tiles per wave (the chunk loop is unrolled, so the tiles are addressed statically).  Wave w owns channel block w & 3
// of every chunk and the token half w >> 2; the halves are summed through LDS at the end and the workgroup writes its (K, M) fp32
// partial slot; the caller sums the <= 256 slots (fixed order).
// PROD = false: the weight gradient alone (a.W / a.out unused) -- dW_x = xc . d(dbc)^T of the x_proj backward (M = dt_rank + 2 d_state).
template <int MB, int NCH, bool PROD>
__global__ __launch_bounds__(64 * GP_WAVES, 1) void proj_wx_thin_wgrad_kernel(cad_proj_args a) {
    typedef GtCfg C;
    CAD_DYN_SMEM(char, smem);
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const int g = lane >> 4, jl = lane & 15;
    const bf16_t* W = (const bf16_t*)a.W;
    const bf16_t* X = (const bf16_t*)a.X;
    const bf16_t* Y = (const bf16_t*)a.wg_y;
    bf16_t* out = (bf16_t*)a.out;
    const int64_t T = a.T;
    const int M = a.M;
    constexpr int K = NCH * C::KC;
    constexpr int YBUF = MB * 16 * C::XROW;           // bytes per Y tile: [MB * 16 rows][128 tokens]
    const int64_t nblk = T / C::NT;                   // (T % 128 == 0: no tail block, every staged token is a real token)
    const int64_t b0 = blockIdx.x, bstep = gridDim.x;
    const int64_t nmine = (nblk - b0 + bstep - 1) / bstep;  // >= 1: the grid never exceeds the number of blocks
    const int64_t total = nmine * NCH;
    char* wl = smem + C::RING * C::XBUF;
    constexpr int WSTR = K * 2 + 16;
    char* yt = wl + (PROD ? MB * 16 * WSTR : 0);  // (the weight-gradient-only form keeps no W copy: M = 64 at K = 512 would not fit otherwise)
    auto issue = [&](int64_t it) {
        const int64_t blk = b0 + (it / NCH) * bstep;
        gt_issue_chunk(X, a.ldx, (int)(it % NCH) * C::KC, blk * C::NT, T, smem + (int)(it % C::RING) * C::XBUF, wave, lane);
    };
    // Y tile of block number bi (of this workgroup): MB * 4 DMA instructions of four rows each, dealt over the waves;
    // physical 16-byte piece pp of row r holds logical piece pp ^ (r & 15): the 16 rows a B-fragment read touches hit 16 distinct
    // pieces = all 64 banks
    auto issue_y = [&](int64_t bi) {
#pragma unroll
        for (int q0 = 0; q0 < MB * 4; q0 += GP_WAVES) {
            const int q = q0 + wave;
            if (q < MB * 4) {
                const int row = q * 4 + (lane >> 4), pp = lane & 15;
                const int lp = pp ^ (row & 15);
                const int64_t blk = b0 + bi * bstep;
                const int rr = row < M ? row : M - 1;  // rows >= M: valid data, their columns of the result are never stored
                cad_glds16(Y + (int64_t)rr * a.ld_wg_y + blk * C::NT + lp * 8,
                           cad_uniform((int)(cad_lds_off(yt) + (int)(bi & 1) * YBUF + q * 1024)));
            }
        }
    };
    issue_y(0);  // BEFORE the chunks: vmcnt retires in order, so the first counted wait below also covers it
    for (int64_t it = 0; it < C::RING - 1; ++it)
        if (it < total) issue(it);
    if constexpr (PROD) {
        for (int i = threadIdx.x; i < MB * 16 * (K / 8); i += 64 * GP_WAVES) {
            const int m = i / (K / 8), c8 = i % (K / 8);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (m < M) v = *(const u32x4*)(W + (int64_t)m * a.ldw + c8 * 8);
            *(u32x4*)(wl + m * WSTR + c8 * 16) = v;
        }
    }
    f32x4 d[MB];
    f32x4 dw[NCH][MB];
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) dw[ch][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rb = wave & 3, kh = wave >> 2;
    for (int64_t bi = 0; bi < nmine; ++bi) {
        const char* ytile = yt + (int)(bi & 1) * YBUF;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int64_t it = bi * NCH + ch;
            if (it + C::RING - 2 < total)
                cad_wait_vmcnt<4>();  // (a Y tile issued behind the chunks only makes this wait stricter)
            else
                cad_wait_vmcnt<0>();
            __syncthreads();
            if (it + C::RING - 1 < total) issue(it + C::RING - 1);
            if (ch == 0 && bi + 1 < nmine) issue_y(bi + 1);  // a whole block ahead of its use
            if (PROD && ch == 0) {
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) d[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            const char* xt = smem + (int)(it % C::RING) * C::XBUF;
            // (1) out += W . X: transposing reads, as proj_wx_thin_kernel
            if constexpr (PROD) {
#pragma unroll
            for (int ks = 0; ks < C::KC / 32; ++ks) {
                const int r0 = ks * 32 + g * 8 + (jl >> 2);
                const char* p0 = xt + r0 * C::XROW + ((wave ^ gx_swz(r0)) * 32) + (jl & 3) * 8;
                const char* p1 = xt + (r0 + 4) * C::XROW + ((wave ^ gx_swz(r0 + 4)) * 32) + (jl & 3) * 8;
                const u32x2 lo = cad_lds_read_tr16(p0), hi = cad_lds_read_tr16(p1);
                const u32x4 xf = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const u32x4 wf = *(const u32x4*)(wl + (mb * 16 + jl) * WSTR + (ch * C::KC + ks * 32 + g * 8) * 2);
                    d[mb] = cad_mfma_16x16x32_bf16(xf, wf, d[mb]);
                }
            }
            }
            // (2) dW[chunk rows, :] += X_tile . Y_tile^T over this wave's 64 tokens: A = channel rows rb * 16 .. + 15 (token-
            // contiguous: eight tokens of a row are one 16-byte piece), B = rows of Y
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int lp = kh * 8 + ks * 4 + g;  // logical 16-byte piece = tokens 8 lp .. 8 lp + 7 of the block
                const int r = rb * 16 + jl;
                const int pp = (((lp >> 1) ^ gx_swz(r)) << 1) | (lp & 1);
                const u32x4 af = *(const u32x4*)(xt + r * C::XROW + pp * 16);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int yr = mb * 16 + jl;
                    const u32x4 bf = *(const u32x4*)(ytile + yr * C::XROW + ((lp ^ (yr & 15)) * 16));
                    dw[ch][mb] = cad_mfma_16x16x32_bf16(af, bf, dw[ch][mb]);
                }
            }
            if (PROD && ch == NCH - 1) {
                const int64_t blk = b0 + bi * bstep;
                const int64_t t = blk * C::NT + wave * 16 + g * 4;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int m = mb * 16 + jl;
                    u32x2 pk;
                    pk[0] = cad_pack_bf16x2_safe(d[mb][0], d[mb][1]);
                    pk[1] = cad_pack_bf16x2_safe(d[mb][2], d[mb][3]);
                    if (m < M) *(u32x2*)(out + (int64_t)m * a.ldo + t) = pk;
                }
            }
        }
    }
    // the two token halves of every accumulator tile meet in LDS (the ring is free now), then the (K, M) partial slot is written
    __syncthreads();
    float* ex = (float*)smem;  // [4 channel blocks][NCH][MB][64 lanes][4]
    if (kh == 1) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) *(f32x4*)(ex + (((rb * NCH + ch) * MB + mb) * 64 + lane) * 4) = dw[ch][mb];
    }
    __syncthreads();
    if (kh == 0) {
        float* slot = a.wg_partials + (int64_t)blockIdx.x * K * M;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const f32x4 o = *(const f32x4*)(ex + (((rb * NCH + ch) * MB + mb) * 64 + lane) * 4);
                const int m = mb * 16 + jl;
                if (m < M) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) slot[(int64_t)(ch * C::KC + rb * 16 + g * 4 + r) * M + m] = dw[ch][mb][r] + o[r];
                }
            }
    }
}


// ---- cad_proj_xTw:  out (T, M) TOKEN-major  =  sum over panels p of  X_p (K, T)^T . W (M, K)^T ----------------------------------------
// out_proj of the tied BiMamba mixer: out = W_out (y_f + y_r) with y_f, y_r the two scans' channel-major outputs -- two panels that
// share ONE weight, so the kernel walks K = d_inner per panel with the SAME resident W fragments (the library GEMM it replaces
// multiplied the concatenation [y_f ; y_r] by a doubled weight [W_out, W_out] that had to be built per layer and step).
// "W-stationary" as cad_proj_wxT: wave w owns output features 16 MB w .. and keeps their MB x K / 32 A-fragments in registers for the
// whole launch; the workgroup walks blocks of 128 tokens, X travels by LDS-DMA in [64 rows][128 tokens] chunks through the ring of
// cad_proj_wx's thin kernel (same swizzle, same counted waits).  MFMA roles: A = W fragment (rows = output features), B = X fragment
// by transposing reads (columns = tokens), so a D lane holds FOUR CONSECUTIVE FEATURES of one token = 8 contiguous bytes of the
// token-major output; fp32 accumulation over both panels and all of K, one rounding to bf16.
#ifndef GP_XTW_PLAIN_STORES
#define GP_XTW_PLAIN_STORES 1   // ordinary stores: the 8-byte pieces of a token row meet in L2 (streaming stores of partial lines: 0.277 vs 0.212 ms)
#endif
#ifndef GP_XTW_RING
#define GP_XTW_RING 4           // LDS tiles of the X ring: RING - 1 chunks (16 KB each) in flight per CU (8 slots measured SLOWER:
                               // 0.222 vs 0.206 ms, profiles/r04_out_proj.txt)
#endif
#ifndef GP_XTW_KS_OUTER
#define GP_XTW_KS_OUTER 1
#endif
template <int MB, int KS>
__global__ __launch_bounds__(64 * GP_WAVES, 2) void proj_xTw_kernel(cad_proj_tm_args a) {
    typedef GtCfg C;
    CAD_DYN_SMEM(char, smem);
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const int g = lane >> 4, jl = lane & 15;
    const bf16_t* W = (const bf16_t*)a.W;
    bf16_t* out = (bf16_t*)a.out;
    const int64_t T = a.T;
    const int M = a.M;
    constexpr int NCH = KS / 2;                       // 64-row chunks per panel
    constexpr int XR = GP_XTW_RING;                   // ring slots
    constexpr int INFL = (XR - 2) * C::DPW;           // DMA instructions of the later chunks that may stay in flight at a wait
    static_assert((XR & (XR - 1)) == 0 && XR >= 4, "ring slots are advanced with a mask");
    const int NP = a.X2 ? 2 : 1;
    const int per_blk = NP * NCH;                     // chunks per token block
    const int64_t nblk = (T + C::NT - 1) / C::NT;
    const int64_t b0 = blockIdx.x, bstep = gridDim.x;
    if (b0 >= nblk) return;
    const int64_t nmine = (nblk - b0 + bstep - 1) / bstep;
    const int64_t total = nmine * per_blk;
    const int m_wave = wave * 16 * MB;
    // position of the next chunk to issue, carried as counters (no division on the issue path)
    int ich = 0, ipan = 0, islot = 0;
    int64_t iblk = b0;
    auto issue_next = [&]() {
        gt_issue_chunk((const bf16_t*)(ipan ? a.X2 : a.X), a.ldx, ich * C::KC, iblk * C::NT, T, smem + islot * C::XBUF, wave, lane);
        islot = (islot + 1) & (XR - 1);
        if (++ich == NCH) {
            ich = 0;
            if (++ipan == NP) {
                ipan = 0;
                iblk += bstep;
            }
        }
    };
    for (int64_t it = 0; it < XR - 1; ++it)
        if (it < total) issue_next();
    // A fragments of this wave's rows of W, resident for the whole launch (rows >= M read as zero)
    u32x4 wf[MB][KS];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const int m = m_wave + mb * 16 + jl;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (m < M) v = *(const u32x4*)(W + (int64_t)m * a.ldw + ks * 32 + g * 8);
            wf[mb][ks] = v;
        }
    }
    f32x4 d[C::NT / 16][MB];
    int slot = 0;
    int64_t blk = b0;
    constexpr int NST = (C::NT / 16) * MB;           // output stores per wave and block
    static_assert(INFL + NST <= 63, "vmcnt is a 6-bit counter");
    int since_store = XR;                             // iterations since the last store burst (wave-uniform)
    for (int64_t bi = 0; bi < nmine; ++bi, blk += bstep) {
#pragma unroll
        for (int q = 0; q < C::NT / 16; ++q)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) d[q][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int pan = 0; pan < NP; ++pan) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int64_t it = (bi * NP + pan) * NCH + ch;
                // chunk `it` has landed when at most the vector-memory operations issued AFTER its DMA are outstanding (vmcnt retires in
                // issue order on gfx9-class hardware, loads, LDS-DMA and stores alike): the two later chunks (4 instructions) and -- for the
                // three waits that follow a block's output stores -- those NST stores.  A plain vmcnt(4) there drained the stores AND the
                // freshly issued prefetch once per block (200 instead of ~140 us per launch, profiles/r04_out_proj.txt).
                if (it + XR - 2 >= total || since_store < 0)
                    cad_wait_vmcnt<0>();
                else if (since_store < XR - 1)
                    cad_wait_vmcnt<INFL + NST>();
                else
                    cad_wait_vmcnt<INFL>();
                ++since_store;
                __syncthreads();  // every wave's share of the chunk is visible; the tile consumed LAST iteration is free again
                if (it + XR - 1 < total) issue_next();
                const char* xt = smem + slot * C::XBUF;
                slot = (slot + 1) & (XR - 1);
                // (k step outer, token sub-block inner: consecutive MFMAs go to sixteen different accumulator tiles)
#pragma unroll
                for (int o1 = 0; o1 < (GP_XTW_KS_OUTER ? C::KC / 32 : C::NT / 16); ++o1) {
#pragma unroll
                    for (int o2 = 0; o2 < (GP_XTW_KS_OUTER ? C::NT / 16 : C::KC / 32); ++o2) {
                        const int ks = GP_XTW_KS_OUTER ? o1 : o2, q = GP_XTW_KS_OUTER ? o2 : o1;
                        const int r0 = ks * 32 + g * 8 + (jl >> 2);
                        const char* p0 = xt + r0 * C::XROW + ((q ^ gx_swz(r0)) * 32) + (jl & 3) * 8;
                        const char* p1 = xt + (r0 + 4) * C::XROW + ((q ^ gx_swz(r0 + 4)) * 32) + (jl & 3) * 8;
                        const u32x2 lo = cad_lds_read_tr16(p0), hi = cad_lds_read_tr16(p1);
                        const u32x4 xf = {lo[0], lo[1], hi[0], hi[1]};
#pragma unroll
                        for (int mb = 0; mb < MB; ++mb) d[q][mb] = cad_mfma_16x16x32_bf16(wf[mb][ch * 2 + ks], xf, d[q][mb]);
                    }
                }
            }
        }
        // lane (token 16 q + jl, features m_wave + 16 mb + 4 g .. + 3): 8 bytes of the token's output row
#pragma unroll
        for (int q = 0; q < C::NT / 16; ++q) {
            const int64_t t = blk * C::NT + q * 16 + jl;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int m = m_wave + mb * 16 + g * 4;
                u32x2 pk;
                pk[0] = cad_pack_bf16x2_safe(d[q][mb][0], d[q][mb][1]);
                pk[1] = cad_pack_bf16x2_safe(d[q][mb][2], d[q][mb][3]);
                // (every lane issues the store: the counted waits above rely on exactly NST store instructions per block; lanes of a tail
                // block / beyond M are masked off by the exec mask, the instruction still counts)
                if (t < T && m + 4 <= M) {
#if GP_XTW_PLAIN_STORES
                    *(u32x2*)(out + t * a.ldo + m) = pk;
#else
                    cad_store_stream<CAD_STREAM_PROJ>((u32x2*)(out + t * a.ldo + m), pk);
#endif
                }
            }
        }
        // (a block with masked-off store instructions -- the tail block, or a wave beyond M -- issued an unknown number of them: the
        // next three waits drain everything instead)
        // ... and so does a configuration whose blocks are shorter than the ring is deep (two store bursts inside one wait's window)
        since_store = (blk * C::NT + C::NT <= T && m_wave + 16 * MB <= M && per_blk >= XR - 1) ? 0 : -XR;
    }
}

// ---- cad_gemm_stream: D (R x C) = A (R x K) . B (K x C), both operands streamed ------------------------------------------------------
// A workgroup owns one 256 x 256 tile of D (8 waves as 2 x 4: a wave accumulates 128 rows x 64 columns = 32 MFMA tiles = 128 fp32
// registers) and walks its k range in chunks of 32 through a ring of 4 LDS stages (16 KB of A rows + 16 KB of B rows per stage) filled
// by LDS-DMA three chunks ahead; per chunk a wave reads 8 A fragments directly (row-major rows, 8 consecutive k = one 16-byte piece)
// and 4 B fragments through the transposing read (B rows are k, columns contiguous) and issues 32 MFMAs.  Both operands cross HBM
// once per tile row / column they belong to: the kernel is the classic tiled GEMM, written for the three shapes of the mixer
// backward where R or C is only one or four tiles wide and everything else is the stream.
#ifndef GS_SLICE_INTERLEAVE
#define GS_SLICE_INTERLEAVE 0   // measured: 0.303 vs 0.249 ms (neighbouring workgroups on neighbouring 64-byte pieces is WORSE)
#endif
#ifndef GS_SLICE_ROTATE
#define GS_SLICE_ROTATE 1
#endif
// start chunk of (slice sl, row tile rt) = (2 sl + rt) mod chunks-per-slice: the 64 x 4 workgroups of a configs[2] weight gradient then
// sit on all 128 chunk phases of the 8 KB every slice owns in a strided row at once.  tools/gemm_stream_bench.py, same box, ms per
// product (library K-split bmm + sum: 0.232): no rotation 0.249 | (37, 11) 0.232 | (53, 29) 0.243 | (19, 5) 0.232 | (45, 77) 0.243 |
// (27, 32) 0.250 | (64, 16) 0.255 | (1, 32) 0.255 | (3, 1) 0.226 | (3, 64) 0.228 | (1, 0) 0.225 | (4, 1) 0.231 | (5, 2) 0.233 |
// (7, 3) 0.222 | (2, 1) 0.215 -- and neighbouring workgroups on neighbouring 64-byte pieces (GS_SLICE_INTERLEAVE) 0.303.
#ifndef GS_XCD_REMAP
#define GS_XCD_REMAP 1
#endif
// timing experiments only (WRONG results; the library says TIMING-BUILD): 1 = no fragment reads / MFMAs (what do the two operand streams,
// the barriers and the stores cost alone), 2 = no LDS-DMA (what does the multiplication cost alone), 4 = MFMAs on constant fragments (no
// LDS fragment reads).  tools/gemm_stream_bench.py over -DGS_WHATIF=... builds; profiles/r06_gemm_stream_whatif.txt
#ifndef GS_WHATIF
#define GS_WHATIF 0
#endif
#ifndef GS_ROT_SL
#define GS_ROT_SL 2
#endif
#ifndef GS_ROT_RT
#define GS_ROT_RT 1
#endif
struct GsCfg {
    static constexpr int RT = 256, CT = 256, KC = 32, RING = 4;
    static constexpr int AROW = KC * 2;               // 64 bytes per A tile row: four 16-byte pieces, piece index ^ gs_aswz(row)
    static constexpr int ABUF = RT * AROW;            // 16 KB
    static constexpr int BROW = CT * 2;               // 512 bytes per B tile row (one k), 32-byte column blocks ^ gx_swz(row)
    static constexpr int BBUF = KC * BROW;            // 16 KB
    static constexpr int STAGE = ABUF + BBUF;
    static constexpr int DPW = 4;                     // DMA instructions per wave and chunk (2 of A, 2 of B)
    static constexpr size_t LDS = (size_t)RING * STAGE;
};
static_assert(GP_WAVES == 8, "cad_gemm_stream: 2 x 4 waves per tile, two DMA instructions per wave, operand and chunk");
// eight consecutive rows x one 16-byte piece (what eight lanes of a ds_read_b128 touch) fall into eight different 16-byte bank groups
__device__ __forceinline__ int gs_aswz(int row) { return (row >> 1) & 3; }

__device__ __forceinline__ void gs_issue_chunk(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, int64_t k0, char* stage,
                                               int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // A: 16 instructions of 16 rows x 4 pieces
        const int ins = wave * 2 + i;
        const int row = ins * 16 + (lane >> 2), pp = lane & 3;
        const int lp = pp ^ gs_aswz(row);  // logical piece = 8 consecutive k
        cad_glds16(A + (int64_t)row * lda + k0 + lp * 8, cad_uniform((int)(cad_lds_off(stage) + ins * 1024)));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {  // B: 16 instructions of 2 rows x 32 pieces
        const int ins = wave * 2 + i;
        const int p = ins * 64 + lane;
        const int row = p >> 5, pp = p & 31;
        const int lp = (((pp >> 1) ^ gx_swz(row)) << 1) | (pp & 1);  // logical piece = 8 consecutive columns
        cad_glds16(B + (k0 + row) * ldb + lp * 8, cad_uniform((int)(cad_lds_off(stage + GsCfg::ABUF) + ins * 1024)));
    }
}

template <int MODE>
__global__ __launch_bounds__(64 * GP_WAVES, 1) void gemm_stream_kernel(cad_gemm_stream_args a) {
    typedef GsCfg C;
    CAD_DYN_SMEM(char, smem);
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const int g = lane >> 4, jl = lane & 15;
    const int wm = wave >> 2, wn = wave & 3;
    const int64_t nrt = a.R / C::RT, nct = a.C / C::CT;
    const int64_t kper = a.K / a.nslices;            // k range of one slice
    const int nk = (int)(kper / C::KC);              // chunks per work item
    const int64_t nitems = nrt * nct * a.nslices;
    // XCD-aware start item: the hardware deals workgroup b to XCD b % 8, and the items that share an operand (the row tiles of one weight-
    // gradient slice read the same B rows; col_fastest: the column tiles of one A tile) are CONSECUTIVE items -- dealt round-robin they
    // sit on different XCDs and the shared operand crosses HBM once per XCD (profiles/r06_step_pmc_summary.txt: 1079 MB fetched for 671 MB
    // of operands at the configs[2] weight gradient = B four times).  Workgroup b therefore takes item (b % 8) (grid / 8) + b / 8: every XCD
    // owns a contiguous run of items, the sharers meet in one L2.
    const int64_t istep = gridDim.x;
    const int64_t i0 = (GS_XCD_REMAP && (gridDim.x % 8) == 0) ? (int64_t)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : (int64_t)blockIdx.x;
    if (i0 >= nitems) return;
    const int64_t nmine = (nitems - i0 + istep - 1) / istep;
    const int64_t total = nmine * nk;
    // work item -> (slice, column tile, row tile).  Row tiles fastest: neighbouring workgroups share the B rows of a slice; col_fastest:
    // neighbouring workgroups share an A tile (A = the streamed activations of in_proj / d(y), all of B cache-resident)
    auto decode = [&](int64_t item, int64_t& rt, int64_t& ct, int64_t& sl) {
        if (a.col_fastest) {
            ct = item % nct;
            const int64_t q = item / nct;
            rt = q % nrt;
            sl = q / nrt;
        } else {
            rt = item % nrt;
            const int64_t q = item / nrt;
            ct = q % nct;
            sl = q / nct;
        }
    };
    // issue cursor
    int ich = 0, islot = 0;
    int64_t iitem = i0;
    const bf16_t *iA = nullptr, *iB = nullptr;
    // GS_SLICE_INTERLEAVE (weight-gradient mode, nslices > 1): slice s takes the 32-k chunks s, s + nslices, s + 2 nslices, ... of the
    // reduction range instead of one contiguous k range, so that the workgroups running side by side read NEIGHBOURING 64-byte pieces of
    // the same strided rows (one DRAM page serves them all) instead of 64 bytes each from pages 8 KB apart
    const int64_t kstep = (GS_SLICE_INTERLEAVE && a.nslices > 1) ? (int64_t)a.nslices * C::KC : (int64_t)C::KC;
    // GS_SLICE_ROTATE (weight-gradient mode): every (slice, row tile) starts its walk at a different chunk of its k range and wraps
    // around.  The slices are 8 KB apart in every strided row and all workgroups advance in step, so without the rotation the whole
    // chip reads the same 4 KB phase of every 8 KB at any moment (the order of a slice's chunks only changes fp32 rounding).
    int irot = 0;
    auto seek = [&]() {
        int64_t rt, ct, sl;
        decode(iitem, rt, ct, sl);
        const int64_t kbase = (GS_SLICE_INTERLEAVE && a.nslices > 1) ? sl * C::KC : sl * kper;
        iA = (const bf16_t*)a.A + rt * C::RT * a.lda + kbase;
        iB = (const bf16_t*)a.B + kbase * a.ldb + ct * C::CT;
        irot = (GS_SLICE_ROTATE && a.nslices > 1) ? (int)((sl * GS_ROT_SL + rt * GS_ROT_RT) % nk) : 0;
    };
    seek();
    auto issue_next = [&]() {
        int kc = ich + irot;
        if (kc >= nk) kc -= nk;
        if (!(GS_WHATIF & 2)) gs_issue_chunk(iA, a.lda, iB, a.ldb, (int64_t)kc * kstep, smem + islot * C::STAGE, wave, lane);
        islot = (islot + 1) & (C::RING - 1);
        if (++ich == nk) {
            ich = 0;
            iitem += istep;
            if (iitem < nitems) seek();
        }
    };
    for (int64_t it = 0; it < C::RING - 1; ++it)
        if (it < total) issue_next();
    constexpr int INFL = (C::RING - 2) * C::DPW;     // DMA instructions of the two later chunks that may stay in flight at a wait
    constexpr int NST = MODE == CAD_GEMM_OUT_T_BF16 ? 32 : 128;  // output stores per wave and item
    static_assert(INFL + 32 <= 63, "vmcnt is a 6-bit counter");
    // lane-constant fragment addresses inside a stage
    const int a_off = (wm * 128 + jl) * C::AROW + ((g ^ gs_aswz(jl)) * 16);       // + i * 16 * AROW  (gs_aswz(row) = gs_aswz(jl))
    const int br0 = g * 8 + (jl >> 2);                                             // source k row of the first 4 x 16 block
    const int b_off0 = C::ABUF + br0 * C::BROW + wn * 128 + (jl & 3) * 8;          // + ((j ^ swz) * 32)
    const int b_off1 = b_off0 + 4 * C::BROW;
    const int bs0 = gx_swz(br0), bs1 = gx_swz(br0 + 4);
    f32x4 acc[8][4];
    int slot = 0;
    int64_t item = i0;
    int since_store = C::RING;                       // iterations since the last store burst (see proj_xTw_kernel)
    for (int64_t bi = 0; bi < nmine; ++bi, item += istep) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ch = 0; ch < nk; ++ch) {
            const int64_t it = bi * nk + ch;
            if (it + C::RING - 2 >= total || since_store < 0 || (MODE == CAD_GEMM_PARTIALS && since_store < C::RING - 1))
                cad_wait_vmcnt<0>();
            else if (MODE == CAD_GEMM_OUT_T_BF16 && since_store < C::RING - 1)
                cad_wait_vmcnt<INFL + 32>();
            else
                cad_wait_vmcnt<INFL>();
            ++since_store;
            __syncthreads();  // every wave's share of chunk `it` is visible; the stage consumed LAST iteration is free again
            if (it + C::RING - 1 < total) issue_next();
            const char* st = smem + slot * C::STAGE;
            slot = (slot + 1) & (C::RING - 1);
            if (GS_WHATIF & 1) continue;
            u32x4 bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (GS_WHATIF & 4) {
                    bfr[j] = u32x4{(uint32_t)(lane + j), (uint32_t)it, 0x3f803f80u, (uint32_t)slot};
                    continue;
                }
                const u32x2 lo = cad_lds_read_tr16(st + b_off0 + ((j ^ bs0) * 32));
                const u32x2 hi = cad_lds_read_tr16(st + b_off1 + ((j ^ bs1) * 32));
                bfr[j] = u32x4{lo[0], lo[1], hi[0], hi[1]};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const u32x4 af = (GS_WHATIF & 4) ? u32x4{(uint32_t)(lane ^ i), 0x3f803f80u, (uint32_t)it, 0u}
                                                  : *(const u32x4*)(st + a_off + i * 16 * C::AROW);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = cad_mfma_16x16x32_bf16(af, bfr[j], acc[i][j]);
            }
        }
        int64_t rt, ct, sl;
        decode(item, rt, ct, sl);
        // lane (g, jl) of tile (i, j): column ct * 256 + wn * 64 + 16 j + jl, rows rt * 256 + wm * 128 + 16 i + 4 g .. + 3
        const int64_t col = ct * C::CT + wn * 64 + jl;
        const int64_t row = rt * C::RT + wm * 128 + 4 * g;
        if constexpr (MODE == CAD_GEMM_PARTIALS) {
            float* dst = (float*)a.out + ((int64_t)sl * a.R + row) * a.C + col;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dst[(int64_t)(16 * i + r) * a.C + 16 * j] = acc[i][j][r];
        } else {
            bf16_t* dst = (bf16_t*)a.out + col * a.ldo + row;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x2 pk;
                    pk[0] = cad_pack_bf16x2_safe(acc[i][j][0], acc[i][j][1]);
                    pk[1] = cad_pack_bf16x2_safe(acc[i][j][2], acc[i][j][3]);
                    *(u32x2*)(dst + (int64_t)(16 * j) * a.ldo + 16 * i) = pk;
                }
        }
        since_store = nk >= C::RING - 1 ? 0 : -C::RING;  // (items shorter than the ring: the next waits drain everything)
    }
    (void)NST;
}

}  // namespace

#define GP_BIG_LDS(kern, bytes) CAD_BIG_LDS(kern, bytes)  // (cad_prims_gfx950.h)

extern "C" int cad_proj_supported(int K) { return K == 32 || K == 64 || K == 128 || K == 256 || K == 512; }

template <int KS>
static int launch_wxT(const cad_proj_args* a, void* stream) {
    typedef GpCfg<KS> C;
    const int64_t nblk = (a->T + C::NT - 1) / C::NT;
    const int my = (a->M + C::MWG - 1) / C::MWG;
    int64_t gx = cad_cu_count() / my;  // ~ one workgroup per CU
    if (gx < 1) gx = 1;
    if (gx > nblk) gx = nblk;
    dim3 grid((unsigned)gx, (unsigned)my), block(64 * GP_WAVES);
    GP_BIG_LDS((proj_wxT_kernel<KS>), C::LDS);
    CAD_LAUNCH((proj_wxT_kernel<KS>), grid, block, C::LDS, stream, *a);
    return cad_after_launch();
}

extern "C" int cad_proj_wxT(const cad_proj_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->W && a->X && a->out && a->T > 0 && a->M > 0 && a->K > 0 && a->act == 0);
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


template <int KS>
static int launch_wx(const cad_proj_args* a, void* stream) {
    typedef GxCfg<KS> C;
    const int64_t nblk = (a->T + C::NT - 1) / C::NT;
    const int my = (a->M + C::MWG - 1) / C::MWG;
    int64_t gx = cad_cu_count() / my;
    if (gx < 1) gx = 1;
    if (gx > nblk) gx = nblk;
    dim3 grid((unsigned)gx, (unsigned)my), block(64 * GP_WAVES);
    if (a->acc) {
        GP_BIG_LDS((proj_wx_kernel<KS, true>), C::LDS);
        CAD_LAUNCH((proj_wx_kernel<KS, true>), grid, block, C::LDS, stream, *a);
    } else {
        GP_BIG_LDS((proj_wx_kernel<KS, false>), C::LDS);
        CAD_LAUNCH((proj_wx_kernel<KS, false>), grid, block, C::LDS, stream, *a);
    }
    return cad_after_launch();
}

extern "C" int cad_proj_wx_supported(int K, int64_t T) { return K >= 8 && K <= 64 && (K % 8) == 0 && T >= 8 && (T % 8) == 0; }
// thin M / deep K variant (x_proj, d(dt_lr)): M <= 64 output rows, K a multiple of 64 up to 1024; the addend (may alias out) carries the
// other K half of a product whose W does not fit LDS in one piece (x_proj at d_inner 1024: 64 rows x 1024)
extern "C" int cad_proj_wx_thin_supported(int M, int K, int64_t T) {
    return M >= 1 && M <= 64 && K > 64 && K <= 1024 && (K % 64) == 0 && T >= 8 && (T % 8) == 0 &&
           GtCfg::lds(M, K) <= 160 * 1024;
}

template <int MB>
static int launch_wx_thin(const cad_proj_args* a, void* stream) {
    const int64_t nblk = (a->T + GtCfg::NT - 1) / GtCfg::NT;
    int64_t gx = cad_cu_count();  // one workgroup per CU
    if (gx > nblk) gx = nblk;
    const size_t lds = GtCfg::lds(MB * 16, a->K);
    dim3 grid((unsigned)gx), block(64 * GP_WAVES);
    GP_BIG_LDS((proj_wx_thin_kernel<MB>), lds);
    CAD_LAUNCH((proj_wx_thin_kernel<MB>), grid, block, lds, stream, *a);
    return cad_after_launch();
}

extern "C" int cad_proj_wx(const cad_proj_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->W && a->X && a->out && a->T > 0 && a->M > 0 && a->K > 0);
    if (a->act == 0 && cad_proj_wx_thin_supported(a->M, a->K, a->T)) {
        CAD_CHECK_ARG(a->ldw >= a->K && a->ldx >= a->T && a->ldo >= a->T);
        CAD_CHECK_ARG((a->ldw % 8) == 0 && (a->ldx % 8) == 0 && (a->ldo % 4) == 0);
        CAD_CHECK_ARG((((uintptr_t)a->W | (uintptr_t)a->X) % 16) == 0 && ((uintptr_t)a->out % 8) == 0);
        CAD_CHECK_ARG(a->acc == nullptr || (a->ldacc >= a->T && (a->ldacc % 4) == 0 && ((uintptr_t)a->acc % 8) == 0));
        CadProfScope prof(8, stream);
        switch ((a->M + 15) / 16) {
            case 1: return launch_wx_thin<1>(a, stream);
            case 2: return launch_wx_thin<2>(a, stream);
            case 3: return launch_wx_thin<3>(a, stream);
            default: return launch_wx_thin<4>(a, stream);
        }
    }
    if (!cad_proj_wx_supported(a->K, a->T)) return CAD_ERR_UNSUPPORTED;
    CAD_CHECK_ARG(a->act == 0 || (a->act == CAD_ACT_SOFTPLUS_BIAS && a->acc == nullptr));
    CAD_CHECK_ARG(a->ldw >= a->K && a->ldx >= a->T && a->ldo >= a->T && (a->acc == nullptr || a->ldacc >= a->T));
    CAD_CHECK_ARG((a->ldw % 8) == 0 && (a->ldx % 8) == 0 && (a->ldo % 8) == 0 && (a->ldacc % 8) == 0);
    CAD_CHECK_ARG((((uintptr_t)a->W | (uintptr_t)a->X | (uintptr_t)a->out | (uintptr_t)a->acc) % 16) == 0);
    CadProfScope prof(8, stream);
    return a->K <= 32 ? launch_wx<1>(a, stream) : launch_wx<2>(a, stream);
}

static size_t gt_wgrad_lds(int M, int K, bool prod) {
    const int mb = (M + 15) / 16;
    size_t lds = (prod ? GtCfg::lds(mb * 16, K) : (size_t)GtCfg::RING * GtCfg::XBUF) + 2 * (size_t)mb * 16 * GtCfg::XROW;
    const size_t exch = (size_t)4 * (K / GtCfg::KC) * mb * 1024;
    return lds < exch ? exch : lds;
}
extern "C" int cad_proj_wx_wgrad_supported(int M, int K, int64_t T) {
    return (M == 16 || M == 32) && (K == 256 || K == 512) && T >= GtCfg::NT && (T % GtCfg::NT) == 0 &&
           gt_wgrad_lds(M, K, true) <= 160 * 1024;
}
// the weight gradient alone (cad_proj_args.W == NULL, out == NULL): any M <= 64
extern "C" int cad_proj_wgrad_only_supported(int M, int K, int64_t T) {
    return M >= 1 && M <= 64 && (K == 256 || K == 512) && T >= GtCfg::NT && (T % GtCfg::NT) == 0 &&
           gt_wgrad_lds(M, K, false) <= 160 * 1024;
}
extern "C" int cad_proj_wx_wgrad_partials(int64_t T) {
    const int64_t nblk = T / GtCfg::NT;
    return (int)(nblk < 256 ? (nblk < 1 ? 1 : nblk) : 256);
}

template <int MB, int NCH, bool PROD>
static int launch_wx_wgrad(const cad_proj_args* a, void* stream) {
    const int gx = cad_proj_wx_wgrad_partials(a->T);
    size_t lds = (PROD ? GtCfg::lds(MB * 16, NCH * GtCfg::KC) : (size_t)GtCfg::RING * GtCfg::XBUF) + 2 * (size_t)MB * 16 * GtCfg::XROW;
    const size_t exch = (size_t)4 * NCH * MB * 1024;  // the final exchange of the accumulator tiles reuses the front of the LDS
    if (lds < exch) lds = exch;
    if (lds > 160 * 1024) return CAD_ERR_UNSUPPORTED;  // (the *_supported predicates exclude these shapes)
    dim3 grid((unsigned)gx), block(64 * GP_WAVES);
    GP_BIG_LDS((proj_wx_thin_wgrad_kernel<MB, NCH, PROD>), lds);
    CAD_LAUNCH((proj_wx_thin_wgrad_kernel<MB, NCH, PROD>), grid, block, lds, stream, *a);
    return cad_after_launch();
}

extern "C" int cad_proj_wx_wgrad(const cad_proj_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->X && a->wg_y && a->wg_partials && a->acc == nullptr && a->act == 0);
    CAD_CHECK_ARG(a->ldx >= a->T && a->ld_wg_y >= a->T && (a->ldx % 8) == 0 && (a->ld_wg_y % 8) == 0);
    CAD_CHECK_ARG((((uintptr_t)a->X | (uintptr_t)a->wg_y) % 16) == 0);
    CadProfScope prof(8, stream);
    if (a->W == nullptr && a->out == nullptr) {  // weight gradient alone
        if (!cad_proj_wgrad_only_supported(a->M, a->K, a->T)) return CAD_ERR_UNSUPPORTED;
        const int mb = (a->M + 15) / 16;
#define GW_ONLY(MB_)                                                                                     \
    return a->K == 256 ? launch_wx_wgrad<MB_, 4, false>(a, stream) : launch_wx_wgrad<MB_, 8, false>(a, stream)
        switch (mb) {
            case 1: GW_ONLY(1);
            case 2: GW_ONLY(2);
            case 3: GW_ONLY(3);
            default: GW_ONLY(4);
        }
#undef GW_ONLY
    }
    CAD_CHECK_ARG(a->W && a->out);
    if (!cad_proj_wx_wgrad_supported(a->M, a->K, a->T)) return CAD_ERR_UNSUPPORTED;
    CAD_CHECK_ARG(a->ldw >= a->K && a->ldo >= a->T && (a->ldw % 8) == 0 && (a->ldo % 4) == 0);
    CAD_CHECK_ARG(((uintptr_t)a->W % 16) == 0 && ((uintptr_t)a->out % 8) == 0);
    if (a->M == 16) return a->K == 256 ? launch_wx_wgrad<1, 4, true>(a, stream) : launch_wx_wgrad<1, 8, true>(a, stream);
    return a->K == 256 ? launch_wx_wgrad<2, 4, true>(a, stream) : launch_wx_wgrad<2, 8, true>(a, stream);
}


// ---- cad_proj_xTw ------------------------------------------------------------------------------------------------------------------
extern "C" int cad_proj_xTw_supported(int M, int K, int64_t T) {
    return (M == 128 || M == 256) && (K == 256 || K == 512) && T >= 8 && (T % 8) == 0;
}

template <int MB, int KS>
static int launch_xTw(const cad_proj_tm_args* a, void* stream) {
    const int64_t nblk = (a->T + GtCfg::NT - 1) / GtCfg::NT;
    int64_t gx = cad_cu_count();  // one workgroup per CU
    if (gx > nblk) gx = nblk;
    const size_t lds = (size_t)GP_XTW_RING * GtCfg::XBUF;
    dim3 grid((unsigned)gx), block(64 * GP_WAVES);
    GP_BIG_LDS((proj_xTw_kernel<MB, KS>), lds);
    CAD_LAUNCH((proj_xTw_kernel<MB, KS>), grid, block, lds, stream, *a);
    return cad_after_launch();
}

extern "C" int cad_proj_xTw(const cad_proj_tm_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->W && a->X && a->out && a->T > 0 && a->M > 0 && a->K > 0);
    if (!cad_proj_xTw_supported(a->M, a->K, a->T)) return CAD_ERR_UNSUPPORTED;
    CAD_CHECK_ARG(a->ldw >= a->K && a->ldx >= a->T && a->ldo >= a->M);
    CAD_CHECK_ARG((a->ldw % 8) == 0 && (a->ldx % 8) == 0 && (a->ldo % 4) == 0);
    CAD_CHECK_ARG((((uintptr_t)a->W | (uintptr_t)a->X | (uintptr_t)a->X2) % 16) == 0 && ((uintptr_t)a->out % 8) == 0);
    CadProfScope prof(8, stream);
    if (a->M == 256) return a->K == 512 ? launch_xTw<2, 16>(a, stream) : launch_xTw<2, 8>(a, stream);
    return a->K == 512 ? launch_xTw<1, 16>(a, stream) : launch_xTw<1, 8>(a, stream);
}

// ---- cad_gemm_stream ----------------------------------------------------------------------------------------------------------------
extern "C" int cad_gemm_stream_supported(int64_t R, int64_t C, int64_t K, int nslices) {
    return R >= GsCfg::RT && (R % GsCfg::RT) == 0 && C >= GsCfg::CT && (C % GsCfg::CT) == 0 && nslices >= 1 && K > 0 &&
           (K % nslices) == 0 && ((K / nslices) % GsCfg::KC) == 0;
}

extern "C" int cad_gemm_stream(const cad_gemm_stream_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->A && a->B && a->out && (a->mode == CAD_GEMM_PARTIALS || a->mode == CAD_GEMM_OUT_T_BF16));
    if (!cad_gemm_stream_supported(a->R, a->C, a->K, a->nslices)) return CAD_ERR_UNSUPPORTED;
    CAD_CHECK_ARG(a->lda >= a->K && a->ldb >= a->C && (a->lda % 8) == 0 && (a->ldb % 8) == 0);
    CAD_CHECK_ARG((((uintptr_t)a->A | (uintptr_t)a->B) % 16) == 0);
    CAD_CHECK_ARG(a->mode == CAD_GEMM_PARTIALS || (a->nslices == 1 && a->ldo >= a->R && (a->ldo % 4) == 0 && ((uintptr_t)a->out % 8) == 0));
    CadProfScope prof(8, stream);
    const int64_t nitems = (a->R / GsCfg::RT) * (a->C / GsCfg::CT) * a->nslices;
    int64_t gx = cad_cu_count();  // one workgroup per CU
    if (gx > nitems) gx = nitems;
    dim3 grid((unsigned)gx), block(64 * GP_WAVES);
    if (a->mode == CAD_GEMM_PARTIALS) {
        GP_BIG_LDS((gemm_stream_kernel<CAD_GEMM_PARTIALS>), GsCfg::LDS);
        CAD_LAUNCH((gemm_stream_kernel<CAD_GEMM_PARTIALS>), grid, block, GsCfg::LDS, stream, *a);
    } else {
        GP_BIG_LDS((gemm_stream_kernel<CAD_GEMM_OUT_T_BF16>), GsCfg::LDS);
        CAD_LAUNCH((gemm_stream_kernel<CAD_GEMM_OUT_T_BF16>), grid, block, GsCfg::LDS, stream, *a);
    }
    return cad_after_launch();
}

// ---- cad_fold_f32_multi: the fp32 partial tiles of a layer's weight gradients, every sum in one launch --------------------------------
namespace {
struct FoldF32Jobs {
    cad_fold_f32_job j[CAD_FOLD_F32_MAX_JOBS];
    int first_block[CAD_FOLD_F32_MAX_JOBS + 1];  // blocks [first_block[i], first_block[i + 1]) work on job i
    int njobs;
};
#define FF_THREADS 256
__global__ __launch_bounds__(FF_THREADS) void fold_f32_multi_kernel(FoldF32Jobs jobs) {
    int ji = 0;
    while (ji + 1 < jobs.njobs && (int)blockIdx.x >= jobs.first_block[ji + 1]) ++ji;  // (wave-uniform: blockIdx)
    const cad_fold_f32_job& jb = jobs.j[ji];
    const int64_t i = ((int64_t)((int)blockIdx.x - jobs.first_block[ji]) * FF_THREADS + threadIdx.x) * 4;
    if (i >= jb.n) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int j2 = 0; j2 < jb.nparts2; ++j2) {
        const float* p = jb.src + (int64_t)j2 * jb.stride2 + i;
        int k = 0;
        for (; k + 4 <= jb.nparts; k += 4) {  // four loads in flight, added in order
            const f32x4 a = *(const f32x4*)(p + (int64_t)k * jb.stride);
            const f32x4 b = *(const f32x4*)(p + (int64_t)(k + 1) * jb.stride);
            const f32x4 c = *(const f32x4*)(p + (int64_t)(k + 2) * jb.stride);
            const f32x4 d = *(const f32x4*)(p + (int64_t)(k + 3) * jb.stride);
            acc = acc + a;
            acc = acc + b;
            acc = acc + c;
            acc = acc + d;
        }
        for (; k < jb.nparts; ++k) acc = acc + *(const f32x4*)(p + (int64_t)k * jb.stride);
    }
    *(f32x4*)(jb.dst + i) = acc;
}
}  // namespace

extern "C" int cad_fold_f32_multi(const cad_fold_f32_job* jobs, int njobs, void* stream) {
    CAD_CHECK_ARG(jobs && njobs >= 1 && njobs <= CAD_FOLD_F32_MAX_JOBS);
    FoldF32Jobs kj;
    kj.njobs = njobs;
    int64_t blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const cad_fold_f32_job& j = jobs[i];
        CAD_CHECK_ARG(j.src && j.dst && j.n > 0 && j.n % 4 == 0 && j.nparts >= 1 && j.nparts2 >= 1);
        CAD_CHECK_ARG(j.stride % 4 == 0 && j.stride2 % 4 == 0 && (((uintptr_t)j.src | (uintptr_t)j.dst) % 16) == 0);
        kj.j[i] = j;
        kj.first_block[i] = (int)blocks;
        blocks += (j.n / 4 + FF_THREADS - 1) / FF_THREADS;
        CAD_CHECK_ARG(blocks < (1 << 30));
    }
    for (int i = njobs; i < CAD_FOLD_F32_MAX_JOBS; ++i) kj.j[i] = jobs[0], kj.first_block[i] = (int)blocks;
    kj.first_block[njobs] = (int)blocks;
    for (int i = njobs + 1; i <= CAD_FOLD_F32_MAX_JOBS; ++i) kj.first_block[i] = (int)blocks;
    dim3 grid((unsigned)blocks), block(FF_THREADS);
    CAD_LAUNCH(fold_f32_multi_kernel, grid, block, 0, stream, kj);
    return cad_after_launch();
}
