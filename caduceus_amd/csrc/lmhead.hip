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

static int64_t lm_head_blocks(int64_t rows) {
    int64_t nb = (rows + LM_WAVES - 1) / LM_WAVES;
    return nb > 4096 ? 4096 : nb;
}

extern "C" int64_t cad_lm_head_partials(int64_t rows) { return 2 * lm_head_blocks(rows); }

extern "C" int cad_lm_head_fwd(const cad_lm_head_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->hidden && a->weight && a->logits);
    CAD_CHECK_ARG(a->rows > 0 && a->D > 0 && a->V > 0);
    CAD_CHECK_ARG(a->n_strands == 1 || (a->n_strands == 2 && a->comp));
    CAD_CHECK_ARG(!a->labels || (a->loss_sum && a->count));
    if (a->V > LM_VMAX) return CAD_ERR_UNSUPPORTED;
    CadProfScope prof(7, stream);
    const int64_t nb = lm_head_blocks(a->rows);
    dim3 grid((unsigned)nb), block(64 * LM_WAVES);
    if (a->dtype == CAD_F32)
        CAD_LAUNCH((lm_head_fwd_kernel<float>), grid, block, 0, stream, *a);
    else if (a->dtype == CAD_BF16)
        CAD_LAUNCH((lm_head_fwd_kernel<bf16_t>), grid, block, 0, stream, *a);
    else
        return CAD_ERR_UNSUPPORTED;
    if (a->labels && a->block_partials)
        CAD_LAUNCH(lm_loss_fold_kernel, dim3(1), dim3(LM_FOLD_THREADS), 0, stream, (const float*)a->block_partials, (int)nb,
                   a->loss_sum, a->count);
    return cad_after_launch();
}
