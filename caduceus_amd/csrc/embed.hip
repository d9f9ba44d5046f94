// RCPS / plain embedding gather and its gradient (t-frame; see include/caduceus_hip.h).
#include "cad_common.h"

namespace {

template <typename TW, typename TO>
__global__ void embed_fwd_kernel(cad_embed_args a) {
    const int64_t tokens = a.B * a.L;
    const int64_t per_strand = tokens * a.D;
    const int64_t total = per_strand * a.n_strands;
    const TW* W = (const TW*)a.weight;
    TO* out = (TO*)a.out;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int s = (int)(idx / per_strand);
        const int64_t rem = idx - (int64_t)s * per_strand;
        const int64_t tok = rem / a.D;
        const int c = (int)(rem - tok * a.D);
        int64_t id = a.ids[tok];
        id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
        if (s == 1) id = a.comp[id];
        out[idx] = from_f32<TO>(to_f32(W[id * a.D + c]));
    }
}

// One block reduces TOK_PER_BLOCK tokens of one strand into an LDS table acc[V][D] (thread c owns column c, so
// no LDS conflicts / atomics), then flushes with fp32 global atomics.
#define EMB_TOK_PER_BLOCK 512
template <typename TG>
__global__ void embed_bwd_kernel(cad_embed_bwd_args a) {
    CAD_DYN_SMEM(float, acc);
    const int64_t tokens = a.B * a.L;
    const int s = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * EMB_TOK_PER_BLOCK;
    const int64_t t1 = (t0 + EMB_TOK_PER_BLOCK < tokens) ? t0 + EMB_TOK_PER_BLOCK : tokens;
    const TG* g = (const TG*)a.dout + (int64_t)s * tokens * a.D;
    for (int i = threadIdx.x; i < a.V * a.D; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    for (int c = threadIdx.x; c < a.D; c += blockDim.x) {
        for (int64_t t = t0; t < t1; ++t) {
            int64_t id = a.ids[t];
            id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
            if (s == 1) id = a.comp[id];
            acc[id * a.D + c] += to_f32(g[t * a.D + c]);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < a.V * a.D; i += blockDim.x) {
        const float v = acc[i];
        if (v != 0.f) atomicAdd(&a.dweight[i], v);
    }
}

}  // namespace

extern "C" int cad_embed_fwd(const cad_embed_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->ids && a->weight && a->out);
    CAD_CHECK_ARG(a->B > 0 && a->L > 0 && a->D > 0 && a->V > 0);
    CAD_CHECK_ARG(a->n_strands == 1 || (a->n_strands == 2 && a->comp));
    CadProfScope prof(6, stream);
    const int64_t total = a->B * a->L * a->D * a->n_strands;
    int64_t nb = (total + 255) / 256;
    if (nb > 16384) nb = 16384;
    dim3 grid((unsigned)nb), block(256);
    if (a->w_dtype == CAD_F32 && a->out_dtype == CAD_F32)
        CAD_LAUNCH((embed_fwd_kernel<float, float>), grid, block, 0, stream, *a);
    else if (a->w_dtype == CAD_F32 && a->out_dtype == CAD_BF16)
        CAD_LAUNCH((embed_fwd_kernel<float, bf16_t>), grid, block, 0, stream, *a);
    else if (a->w_dtype == CAD_BF16 && a->out_dtype == CAD_BF16)
        CAD_LAUNCH((embed_fwd_kernel<bf16_t, bf16_t>), grid, block, 0, stream, *a);
    else if (a->w_dtype == CAD_BF16 && a->out_dtype == CAD_F32)
        CAD_LAUNCH((embed_fwd_kernel<bf16_t, float>), grid, block, 0, stream, *a);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}

extern "C" int cad_embed_bwd(const cad_embed_bwd_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->ids && a->dout && a->dweight);
    CAD_CHECK_ARG(a->B > 0 && a->L > 0 && a->D > 0 && a->V > 0);
    CAD_CHECK_ARG(a->n_strands == 1 || (a->n_strands == 2 && a->comp));
    const size_t shmem = (size_t)a->V * a->D * sizeof(float);
    if (shmem > 64 * 1024) return CAD_ERR_UNSUPPORTED;  // large-vocab models: host uses index_add instead
    CadProfScope prof(6, stream);
    const int64_t tokens = a->B * a->L;
    dim3 grid((unsigned)((tokens + EMB_TOK_PER_BLOCK - 1) / EMB_TOK_PER_BLOCK), (unsigned)a->n_strands), block(256);
    if (a->dout_dtype == CAD_F32)
        CAD_LAUNCH((embed_bwd_kernel<float>), grid, block, shmem, stream, *a);
    else if (a->dout_dtype == CAD_BF16)
        CAD_LAUNCH((embed_bwd_kernel<bf16_t>), grid, block, shmem, stream, *a);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}
