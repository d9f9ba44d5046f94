// RCPS LM head (16-way vocabulary) fused with fp32 logits and the masked cross-entropy partial sums
// (include/caduceus_hip.h, cad_lm_head_fwd).  One wave64 per token; W (V x D fp32, <= 16 KB) is read through L1.
#include "cad_common.h"

namespace {

#define LM_VMAX 16
#define LM_WAVES 4

template <typename T>
__global__ __launch_bounds__(64 * LM_WAVES) void lm_head_fwd_kernel(cad_lm_head_args a) {
    __shared__ float red[LM_WAVES][2];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int D = a.D, V = a.V;
    const T* hid = (const T*)a.hidden;
    float loss_part = 0.f, cnt_part = 0.f;
    for (int64_t row = (int64_t)blockIdx.x * LM_WAVES + wave; row < a.rows; row += (int64_t)gridDim.x * LM_WAVES) {
        // Per-strand partial sums are kept apart and combined with ONE commutative add per lane, so that
        // logits(x)[v] and logits(RC x)[comp v] (strand roles exchanged) are bit-identical, like the reference's
        // `fwd_logits + rc_logits` (modeling_rcps.py:240-246).
        float acc[LM_VMAX], acc1[LM_VMAX];
#pragma unroll
        for (int v = 0; v < LM_VMAX; ++v) acc[v] = acc1[v] = 0.f;
        {
            const T* h = hid + row * D;
            for (int c = lane; c < D; c += 64) {
                const float hv = to_f32(h[c]);
#pragma unroll
                for (int v = 0; v < LM_VMAX; ++v)
                    if (v < V) acc[v] += hv * a.weight[(int64_t)v * D + c];
            }
        }
        if (a.n_strands == 2) {
            const T* h = hid + (a.rows + row) * D;
            for (int c = lane; c < D; c += 64) {
                const float hv = to_f32(h[c]);
#pragma unroll
                for (int v = 0; v < LM_VMAX; ++v)
                    if (v < V) acc1[v] += hv * a.weight[a.comp[v] * D + c];
            }
#pragma unroll
            for (int v = 0; v < LM_VMAX; ++v) acc[v] = acc[v] + acc1[v];
        }
#pragma unroll
        for (int v = 0; v < LM_VMAX; ++v) {
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) acc[v] += __shfl_xor(acc[v], m);
        }
        if (lane == 0) {
            float mx = -3.0e38f;
#pragma unroll
            for (int v = 0; v < LM_VMAX; ++v) {
                if (v < V) {
                    a.logits[row * V + v] = acc[v];
                    mx = acc[v] > mx ? acc[v] : mx;
                }
            }
            if (a.labels) {
                const int64_t lab = a.labels[row];
                if (lab != a.ignore_index && lab >= 0 && lab < V) {
                    float se = 0.f, pick = 0.f;
#pragma unroll
                    for (int v = 0; v < LM_VMAX; ++v) {
                        if (v < V) {
                            se += expf(acc[v] - mx);
                            if (v == lab) pick = acc[v];
                        }
                    }
                    loss_part += (logf(se) + mx) - pick;
                    cnt_part += 1.f;
                }
            }
        }
    }
    if (a.labels) {
        if (lane == 0) {
            red[wave][0] = loss_part;
            red[wave][1] = cnt_part;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float l = 0.f, n = 0.f;
            for (int w = 0; w < LM_WAVES; ++w) {
                l += red[w][0];
                n += red[w][1];
            }
            if (a.block_partials) {  // deterministic: one slot per workgroup, folded in a fixed order by lm_loss_fold_kernel
                a.block_partials[2 * blockIdx.x] = l;
                a.block_partials[2 * blockIdx.x + 1] = n;
            } else if (n > 0.f) {
                atomicAdd(a.loss_sum, l);
                atomicAdd(a.count, n);
            }
        }
    }
}

// Matrix-core variant (D = 32 NJ, NJ in {4, 8, 16}: d_model 128 / 256 / 512).  A wave owns 16-token tiles: logits (16 x 16) =
// H (16 x D) . W^T with v_mfma_f32_16x16x4_f32 -- fp32 operands, so W stays the fp32 master weight -- one accumulator tile per
// strand.  The k axis is walked in a permuted order that makes both operands 16-byte vector loads: MFMA number 8 j + e takes, from
// lane (token or vocabulary row, k-group g), element  k = 8 g + 32 j + e.  W (64 or fewer floats per lane) is loaded once per wave.
// RCPS: the second strand's tile is computed with the SAME W registers and its columns are permuted by comp afterwards
// (ds_bpermute), then ONE commutative add per logit: logits(x)[v] and logits(RC x)[comp v] are the same two numbers added.
// The cross entropy of a token is a reduction over the 16 lanes that hold its row.
template <typename T, int NJ>
__global__ __launch_bounds__(64 * LM_WAVES) void lm_head_fwd_mfma_kernel(cad_lm_head_args a) {
    __shared__ float red[LM_WAVES][2];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, jl = lane & 15;
    constexpr int D = 32 * NJ;
    constexpr int WPT = sizeof(T) * 8 / 4;  // 32-bit words per 8 elements
    struct __attribute__((aligned(sizeof(T) * 8))) Raw { uint32_t w[WPT]; };
    const int V = a.V;
    const T* hid = (const T*)a.hidden;
    const int64_t rows = a.rows;
    const int64_t ntiles = (rows + 15) / 16;
    float wreg[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) wreg[j][e] = jl < V ? a.weight[(int64_t)jl * D + 8 * g + 32 * j + e] : 0.f;
    const int cv = (a.n_strands == 2 && jl < V) ? (int)a.comp[jl] : jl;  // the column of strand 1's tile that feeds logit jl
    const int src = (lane & 48) | cv;
    auto load_tile = [&](int64_t tile, int s, Raw* dst) {
        int64_t row = tile * 16 + jl;
        row = row < rows ? row : rows - 1;  // tail tile: valid data, never stored
        const T* h = hid + ((int64_t)s * rows + row) * D + 8 * g;
#pragma unroll
        for (int j = 0; j < NJ; ++j) dst[j] = *(const Raw*)(h + 32 * j);
    };
    auto unpack = [&](const Raw& r, float* o) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[2 * q] = cad_bits2f(r.w[q] << 16);
                o[2 * q + 1] = cad_bits2f(r.w[q] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) o[q] = cad_bits2f(r.w[q]);
        }
    };
    float loss_part = 0.f, cnt_part = 0.f;
    const int64_t tstep = (int64_t)gridDim.x * LM_WAVES;
    int64_t tile = (int64_t)blockIdx.x * LM_WAVES + wave;
    Raw h0[NJ], h1[NJ];
    if (tile < ntiles) {
        load_tile(tile, 0, h0);
        if (a.n_strands == 2) load_tile(tile, 1, h1);
    }
    for (; tile < ntiles; tile += tstep) {
        f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float hv[8];
            unpack(h0[j], hv);
#pragma unroll
            for (int e = 0; e < 8; ++e) z0 = cad_mfma_16x16x4_f32(hv[e], wreg[j][e], z0);
        }
        if (tile + tstep < ntiles) load_tile(tile + tstep, 0, h0);  // the next tile's strand 0 in flight under strand 1's MFMAs
        if (a.n_strands == 2) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                float hv[8];
                unpack(h1[j], hv);
#pragma unroll
                for (int e = 0; e < 8; ++e) z1 = cad_mfma_16x16x4_f32(hv[e], wreg[j][e], z1);
            }
            if (tile + tstep < ntiles) load_tile(tile + tstep, 1, h1);
#pragma unroll
            for (int r = 0; r < 4; ++r) z0[r] = z0[r] + __shfl(z1[r], src);
        }
        // lane (column jl, rows 4 g + r): logits of tokens tile * 16 + 4 g + r
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = tile * 16 + 4 * g + r;
            const bool ok = row < rows;
            const float z = z0[r];
            if (ok && jl < V) a.logits[row * V + jl] = z;
            if (a.labels) {
                const int64_t lab = ok ? a.labels[row] : a.ignore_index;
                float mx = jl < V ? z : -3.0e38f;
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) {
                    const float o = __shfl_xor(mx, m);
                    mx = o > mx ? o : mx;
                }
                float se = jl < V ? expf(z - mx) : 0.f;
                float pick = (int64_t)jl == lab ? z : 0.f;
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) {
                    se += __shfl_xor(se, m);
                    pick += __shfl_xor(pick, m);
                }
                if (jl == 0 && lab != a.ignore_index && lab >= 0 && lab < V) {
                    loss_part += (logf(se) + mx) - pick;
                    cnt_part += 1.f;
                }
            }
        }
    }
    if (a.labels) {
        // the four row groups of a wave, then the waves of the workgroup: fixed order
#pragma unroll
        for (int m = 32; m >= 16; m >>= 1) {
            loss_part += __shfl_xor(loss_part, m);
            cnt_part += __shfl_xor(cnt_part, m);
        }
        if (lane == 0) {
            red[wave][0] = loss_part;
            red[wave][1] = cnt_part;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float l = 0.f, n = 0.f;
            for (int w = 0; w < LM_WAVES; ++w) {
                l += red[w][0];
                n += red[w][1];
            }
            if (a.block_partials) {
                a.block_partials[2 * blockIdx.x] = l;
                a.block_partials[2 * blockIdx.x + 1] = n;
            } else if (n > 0.f) {
                atomicAdd(a.loss_sum, l);
                atomicAdd(a.count, n);
            }
        }
    }
}

// Backward on the matrix cores (D = 16 NCB, NCB in {8, 16}), one 16-token tile per wave at a time, v_mfma_f32_16x16x4_f32 throughout.
// MFMA column / row index j of channel block cb stands for channel  j * NCB + cb : a lane then owns NCB CONTIGUOUS channels of a
// token (or of a weight row) across the NCB blocks -- hidden rows, weight rows, d hidden rows and the dW slot are all moved with
// 16-byte accesses.
//   G tile (the loss gradient w.r.t. the logits) in the layout of the forward's epilogue (lane = vocabulary column, 4 token rows),
//   softmax over the 16 lanes of a row; it goes through a wave-private LDS tile, from which both products take their fragments:
//   d hidden^T block cb = sum_m  W^T[channels of cb][v = 4m + k]  .  G^T[v = 4m + k][tokens]          (4 MFMAs per block and strand)
//   dW block cb        += sum_m  G^T[v][tokens 4m + k]            .  hidden[tokens 4m + k][channels of cb]
//   strand 1 reads the G tile with its columns permuted by comp (the same W registers and accumulators serve both strands).
// Each wave keeps dW (16 x D fp32) in NCB accumulator tiles; the four waves' tiles are added through LDS, one slot per workgroup.
template <typename T, int NCB>
__global__ __launch_bounds__(64 * LM_WAVES) void lm_head_bwd_mfma_kernel(cad_lm_head_bwd_args a) {
    constexpr int D = 16 * NCB;
    constexpr int GS = 17;  // row stride of the G tile (floats)
    __shared__ float gs_all[LM_WAVES][16 * GS];
    __shared__ float fold[LM_WAVES - 1][NCB][64 * 4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane >> 4, jl = lane & 15;
    const int V = a.V;
    const T* hid = (const T*)a.hidden;
    T* dh = (T*)a.dhidden;
    const int64_t rows = a.rows;
    const int64_t ntiles = (rows + 15) / 16;
    float* gs = gs_all[wave];
    const int64_t LD = a.ld ? a.ld : D;  // row stride of hidden / dhidden / weight / dW-slot rows: a channel block of a wider head
    const float coef = a.loss_scale ? a.loss_scale[0] : 0.f;
    // A fragments of d hidden^T: W[v = 4m + g][channel jl * NCB + cb]  (rows v >= V are zero)
    float wreg[4][NCB];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) wreg[m][cb] = (4 * m + g) < V ? a.weight[(int64_t)(4 * m + g) * LD + jl * NCB + cb] : 0.f;
    int cperm[4];  // strand 1: column comp[4m + g] of the G tile feeds k = 4m + g
#pragma unroll
    for (int m = 0; m < 4; ++m) cperm[m] = (a.n_strands == 2 && 4 * m + g < V) ? (int)a.comp[4 * m + g] : 4 * m + g;
    const int cjl = (a.n_strands == 2 && jl < V) ? (int)a.comp[jl] : jl;
    f32x4 dw[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) dw[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int EPV = 16 / sizeof(T);       // elements per 16-byte access
    constexpr int NV = NCB / EPV;             // 16-byte accesses per lane and token row segment
    struct __attribute__((aligned(16))) Raw { uint32_t w[4]; };
    for (int64_t tile = (int64_t)blockIdx.x * LM_WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * LM_WAVES) {
        // ---- G tile: lane (column jl, rows 4 g + r)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = tile * 16 + 4 * g + r;
            const bool ok = row < rows;
            float gv = 0.f;
            if (a.labels) {
                const float z = (ok && jl < V) ? a.logits[row * V + jl] : -3.0e38f;
                float mx = z;
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) {
                    const float o = __shfl_xor(mx, m);
                    mx = o > mx ? o : mx;
                }
                const float e = (ok && jl < V) ? expf(z - mx) : 0.f;
                float se = e;
#pragma unroll
                for (int m = 8; m >= 1; m >>= 1) se += __shfl_xor(se, m);
                const int64_t lab = ok ? a.labels[row] : a.ignore_index;
                const bool counts = lab != a.ignore_index && lab >= 0 && lab < V;
                if (counts && jl < V) gv = (e / se - ((int64_t)jl == lab ? 1.f : 0.f)) * coef;
            }
            if (a.dlogits && ok && jl < V) gv += a.dlogits[row * V + jl];
            gs[(4 * g + r) * GS + jl] = gv;
        }
        cad_wave_sync();
        for (int s = 0; s < a.n_strands; ++s) {
            int64_t trow = tile * 16 + jl;
            const bool tok = trow < rows;
            trow = tok ? trow : rows - 1;
            // ---- d hidden: B fragments G_s[token jl][v = 4m + g]
            float bg[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) bg[m] = gs[jl * GS + (s == 0 ? 4 * m + g : cperm[m])];
            f32x4 z[NCB];
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                z[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 4; ++m) z[cb] = cad_mfma_16x16x4_f32(wreg[m][cb], bg[m], z[cb]);
            }
            // lane (token jl, rows 4 g + r): channels (4 g + r) * NCB .. + NCB - 1
            if (tok) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    T* dst = dh + ((int64_t)s * rows + trow) * LD + (4 * g + r) * NCB;
#pragma unroll
                    for (int q = 0; q < NCB; q += 4) {
                        const float v4[4] = {z[q][r], z[q + 1][r], z[q + 2][r], z[q + 3][r]};
                        cad_cvt_store<T, 4>(dst + q, v4);
                    }
                }
            }
            // ---- dW: A fragments G_s[token 4m + g][v' = jl], B fragments hidden[token 4m + g][channels jl * NCB .. + NCB - 1]
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float ag = gs[(4 * m + g) * GS + (s == 0 ? jl : cjl)];
                int64_t hrow = tile * 16 + 4 * m + g;
                const bool hok = hrow < rows;
                hrow = hok ? hrow : rows - 1;
                const T* src = hid + ((int64_t)s * rows + hrow) * LD + jl * NCB;
                float hv[NCB];
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const Raw rw = *(const Raw*)(src + q * EPV);
                    if constexpr (sizeof(T) == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hv[q * 8 + 2 * e] = cad_bits2f(rw.w[e] << 16);
                            hv[q * 8 + 2 * e + 1] = cad_bits2f(rw.w[e] & 0xffff0000u);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[q * 4 + e] = cad_bits2f(rw.w[e]);
                    }
                }
                const float agm = hok ? ag : 0.f;  // rows beyond the end add nothing
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) dw[cb] = cad_mfma_16x16x4_f32(agm, hv[cb], dw[cb]);
            }
        }
        cad_wave_sync();  // the G tile is rewritten by the next tile
    }
    // the four waves' dW tiles -> one slot per workgroup (fixed order); lane (column jl, rows v = 4 g + r): channels jl * NCB + cb
    if (wave > 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) *(f32x4*)(&fold[wave - 1][cb][lane * 4]) = dw[cb];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
            for (int w = 0; w < LM_WAVES - 1; ++w) {
                const f32x4 o = *(const f32x4*)(&fold[w][cb][lane * 4]);
                dw[cb][0] += o[0], dw[cb][1] += o[1], dw[cb][2] += o[2], dw[cb][3] += o[3];
            }
        float* slot = a.dw_partials + (int64_t)blockIdx.x * V * LD;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int v = 4 * g + r;
            if (v < V) {
#pragma unroll
                for (int q = 0; q < NCB; q += 4) {
                    const float v4[4] = {dw[q][r], dw[q + 1][r], dw[q + 2][r], dw[q + 3][r]};
                    cad_cvt_store<float, 4>(slot + (int64_t)v * LD + jl * NCB + q, v4);
                }
            }
        }
    }
}

// second stage of the deterministic loss: one workgroup folds the per-workgroup (loss, count) pairs with a fixed-shape
// tree (the same association order on every run and for every grid of the same size)
#define LM_FOLD_THREADS 256
__global__ __launch_bounds__(LM_FOLD_THREADS) void lm_loss_fold_kernel(const float* parts, int nparts, float* loss_sum,
                                                                       float* count) {
    __shared__ float sl[LM_FOLD_THREADS], sn[LM_FOLD_THREADS];
    float l = 0.f, n = 0.f;
    for (int i = threadIdx.x; i < nparts; i += LM_FOLD_THREADS) {
        l += parts[2 * i];
        n += parts[2 * i + 1];
    }
    sl[threadIdx.x] = l;
    sn[threadIdx.x] = n;
    __syncthreads();
    for (int st = LM_FOLD_THREADS / 2; st >= 1; st >>= 1) {
        if ((int)threadIdx.x < st) {
            sl[threadIdx.x] += sl[threadIdx.x + st];
            sn[threadIdx.x] += sn[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        loss_sum[0] += sl[0];
        count[0] += sn[0];
    }
}

}  // namespace

// W resident in registers: 32 / 64 VGPRs; d_model 512 (configs[4]) with 128, in bf16 only (363 registers, one wave per SIMD: the fp32
// instantiation would spill)
static bool lm_head_mfma(int D, int dtype) { return D == 128 || D == 256 || (D == 512 && dtype == CAD_BF16); }
// (the number of per-workgroup loss slots depends on rows only, so that the caller can size them without knowing D: the matrix-core
// kernel uses at most as many workgroups as the general one)
static int64_t lm_head_blocks(int64_t rows) {
    int64_t nb = (rows + LM_WAVES - 1) / LM_WAVES;
    return nb > 4096 ? 4096 : nb;
}
static int64_t lm_head_blocks_mfma(int64_t rows) {
    int64_t nb = ((rows + 15) / 16 + LM_WAVES - 1) / LM_WAVES;
    return nb > 512 ? 512 : nb;  // two workgroups per CU: a wave walks several tiles with its W registers (2048: 0.101 ms at 131072 tokens)
}

extern "C" int64_t cad_lm_head_partials(int64_t rows) { return 2 * lm_head_blocks(rows); }

extern "C" int cad_lm_head_fwd(const cad_lm_head_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->hidden && a->weight && a->logits);
    CAD_CHECK_ARG(a->rows > 0 && a->D > 0 && a->V > 0);
    CAD_CHECK_ARG(a->n_strands == 1 || (a->n_strands == 2 && a->comp));
    CAD_CHECK_ARG(!a->labels || (a->loss_sum && a->count));
    if (a->V > LM_VMAX) return CAD_ERR_UNSUPPORTED;
    CadProfScope prof(7, stream);
    if (a->dtype != CAD_F32 && a->dtype != CAD_BF16) return CAD_ERR_UNSUPPORTED;
    const bool mfma = lm_head_mfma(a->D, a->dtype) && ((uintptr_t)a->hidden % 32) == 0;
    const int64_t nb = mfma ? lm_head_blocks_mfma(a->rows) : lm_head_blocks(a->rows);
    dim3 grid((unsigned)nb), block(64 * LM_WAVES);
#define LM_MFMA(NJ_)                                                                              \
    do {                                                                                          \
        if (a->dtype == CAD_F32)                                                                  \
            CAD_LAUNCH((lm_head_fwd_mfma_kernel<float, NJ_>), grid, block, 0, stream, *a);        \
        else                                                                                      \
            CAD_LAUNCH((lm_head_fwd_mfma_kernel<bf16_t, NJ_>), grid, block, 0, stream, *a);       \
    } while (0)
    if (mfma && a->D == 128)
        LM_MFMA(4);
    else if (mfma && a->D == 256)
        LM_MFMA(8);
    else if (mfma)
        CAD_LAUNCH((lm_head_fwd_mfma_kernel<bf16_t, 16>), grid, block, 0, stream, *a);
    else if (a->dtype == CAD_F32)
        CAD_LAUNCH((lm_head_fwd_kernel<float>), grid, block, 0, stream, *a);
    else
        CAD_LAUNCH((lm_head_fwd_kernel<bf16_t>), grid, block, 0, stream, *a);
#undef LM_MFMA
    if (a->labels && a->block_partials)
        CAD_LAUNCH(lm_loss_fold_kernel, dim3(1), dim3(LM_FOLD_THREADS), 0, stream, (const float*)a->block_partials, (int)nb,
                   a->loss_sum, a->count);
    return cad_after_launch();
}

extern "C" int cad_lm_head_bwd_supported(int D, int V) { return (D == 128 || D == 256) && V >= 1 && V <= LM_VMAX; }

static int64_t lm_head_bwd_blocks(int64_t rows) {
    int64_t nb = ((rows + 15) / 16 + LM_WAVES - 1) / LM_WAVES;
    return nb > 512 ? 512 : nb;
}
extern "C" int64_t cad_lm_head_bwd_partials(int64_t rows) { return lm_head_bwd_blocks(rows); }

extern "C" int cad_lm_head_bwd(const cad_lm_head_bwd_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->hidden && a->weight && a->dhidden && a->dw_partials);
    CAD_CHECK_ARG(a->rows > 0 && a->D > 0 && a->V > 0);
    CAD_CHECK_ARG(a->n_strands == 1 || (a->n_strands == 2 && a->comp));
    CAD_CHECK_ARG(!a->labels || (a->logits && a->loss_scale));
    CAD_CHECK_ARG(a->labels || a->dlogits);
    if (!cad_lm_head_bwd_supported(a->D, a->V) || (a->dtype != CAD_F32 && a->dtype != CAD_BF16)) return CAD_ERR_UNSUPPORTED;
    CAD_CHECK_ARG(a->ld == 0 || (a->ld >= a->D && (a->ld % 8) == 0));
    CAD_CHECK_ARG((((uintptr_t)a->hidden | (uintptr_t)a->dhidden | (uintptr_t)a->dw_partials) % 16) == 0);
    CadProfScope prof(7, stream);
    dim3 grid((unsigned)lm_head_bwd_blocks(a->rows)), block(64 * LM_WAVES);
#define LM_BWD(NCB_)                                                                              \
    do {                                                                                          \
        if (a->dtype == CAD_F32)                                                                  \
            CAD_LAUNCH((lm_head_bwd_mfma_kernel<float, NCB_>), grid, block, 0, stream, *a);       \
        else                                                                                      \
            CAD_LAUNCH((lm_head_bwd_mfma_kernel<bf16_t, NCB_>), grid, block, 0, stream, *a);      \
    } while (0)
    if (a->D == 128)
        LM_BWD(8);
    else
        LM_BWD(16);
#undef LM_BWD
    return cad_after_launch();
}
