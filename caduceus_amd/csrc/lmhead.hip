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

// Matrix-core variant (D = 32 NJ, NJ in {4, 8}: d_model 128 / 256).  A wave owns 16-token tiles: logits (16 x 16) =
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

static bool lm_head_mfma(int D) { return D == 128 || D == 256; }  // W resident in registers: 32 / 64 VGPRs
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
    const bool mfma = lm_head_mfma(a->D) && ((uintptr_t)a->hidden % 32) == 0;
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
    else if (mfma)
        LM_MFMA(8);
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
