// RCPS / plain embedding gather and its gradient (t-frame; see include/caduceus_hip.h).
#include "cad_common.h"

namespace {

// 4 consecutive elements -> fp32 (one 16- / 8-byte load)
template <typename T>
__device__ __forceinline__ void ld4e(const T* p, float* o);
template <>
__device__ __forceinline__ void ld4e<float>(const float* p, float* o) {
    struct __attribute__((aligned(16))) V { float f[4]; };
    const V t = *(const V*)p;
    o[0] = t.f[0], o[1] = t.f[1], o[2] = t.f[2], o[3] = t.f[3];
}
template <>
__device__ __forceinline__ void ld4e<bf16_t>(const bf16_t* p, float* o) {
    struct __attribute__((aligned(8))) V { uint32_t w[2]; };
    const V t = *(const V*)p;
    o[0] = cad_bits2f(t.w[0] << 16), o[1] = cad_bits2f(t.w[0] & 0xffff0000u);
    o[2] = cad_bits2f(t.w[1] << 16), o[3] = cad_bits2f(t.w[1] & 0xffff0000u);
}

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

// Vector variant (fp32 table, D % 4 == 0, 16-byte aligned): one wave per token, a lane moves 4 channels (one 16-byte load, one 16- or
// 8-byte store) per strand -- no index arithmetic per element (the scalar kernel divides twice per element: 0.15 ms for the 268 MB of
// configs[2], 1.8 TB/s).
#define EMB_WAVES 4
template <typename TO>
__global__ __launch_bounds__(64 * EMB_WAVES) void embed_fwd_vec_kernel(cad_embed_args a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t tokens = a.B * a.L;
    const int D = a.D, D4 = D >> 2;
    const float* W = (const float*)a.weight;
    TO* out = (TO*)a.out;
    struct __attribute__((aligned(16))) F4 { float f[4]; };
    for (int64_t tok = (int64_t)blockIdx.x * EMB_WAVES + wave; tok < tokens; tok += (int64_t)gridDim.x * EMB_WAVES) {
        int64_t id = a.ids[tok];
        id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
        const int64_t id1 = a.n_strands == 2 ? a.comp[id] : id;
        for (int c4 = lane; c4 < D4; c4 += 64) {
            const F4 v0 = *(const F4*)(W + id * D + 4 * c4);
            cad_cvt_store<TO, 4>(out + tok * D + 4 * c4, v0.f);
            if (a.n_strands == 2) {
                const F4 v1 = *(const F4*)(W + id1 * D + 4 * c4);
                cad_cvt_store<TO, 4>(out + (tokens + tok) * D + 4 * c4, v1.f);
            }
        }
    }
}

// Vector variant of the backward (D % 4 == 0, V * D * 16 bytes of LDS): every wave owns a table acc[V][D] and walks its share of
// the block's tokens, a lane adds 4 channels per token (one 16-byte load of dout, one LDS read-modify-write), the next token's row is
// in flight under the current one; the four tables are added and flushed with fp32 atomics as below.
#define EMB_TOK_PER_BLOCK_VEC 256
template <typename TG>
__global__ __launch_bounds__(64 * EMB_WAVES) void embed_bwd_vec_kernel(cad_embed_bwd_args a) {
    CAD_DYN_SMEM(float, acc);  // [EMB_WAVES][V][D]
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t tokens = a.B * a.L;
    const int s = blockIdx.y;
    const int D = a.D, D4 = D >> 2, VD = a.V * D;
    const int64_t t0 = (int64_t)blockIdx.x * EMB_TOK_PER_BLOCK_VEC;
    const int64_t t1 = (t0 + EMB_TOK_PER_BLOCK_VEC < tokens) ? t0 + EMB_TOK_PER_BLOCK_VEC : tokens;
    const TG* g = (const TG*)a.dout + (int64_t)s * tokens * D;
    float* mine = acc + wave * VD;
    for (int i = threadIdx.x; i < EMB_WAVES * VD; i += blockDim.x) acc[i] = 0.f;
    __syncthreads();
    struct __attribute__((aligned(16))) F4 { float f[4]; };
    for (int c4 = lane; c4 < D4; c4 += 64) {
        float nx[4] = {0.f, 0.f, 0.f, 0.f};
        int64_t nid = 0;
        auto fetch = [&](int64_t t) {
            nid = a.ids[t];
            ld4e<TG>(g + t * D + 4 * c4, nx);
        };
        if (t0 + wave < t1) fetch(t0 + wave);
        for (int64_t t = t0 + wave; t < t1; t += EMB_WAVES) {
            float cur[4] = {nx[0], nx[1], nx[2], nx[3]};
            int64_t id = nid;
            if (t + EMB_WAVES < t1) fetch(t + EMB_WAVES);
            id = id < 0 ? 0 : (id >= a.V ? a.V - 1 : id);
            if (s == 1) id = a.comp[id];
            F4* p = (F4*)(mine + id * D + 4 * c4);
            F4 v = *p;
#pragma unroll
            for (int q = 0; q < 4; ++q) v.f[q] += cur[q];
            *p = v;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < VD; i += blockDim.x) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < EMB_WAVES; ++w) v += acc[w * VD + i];
        if (v != 0.f) atomicAdd(&a.dweight[i], v);
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
    if (a->w_dtype == CAD_F32 && (a->D % 4) == 0 && (((uintptr_t)a->weight | (uintptr_t)a->out) % 16) == 0 &&
        (a->out_dtype == CAD_F32 || a->out_dtype == CAD_BF16)) {
        int64_t nbv = (a->B * a->L + EMB_WAVES - 1) / EMB_WAVES;
        if (nbv > 8192) nbv = 8192;
        dim3 gridv((unsigned)nbv), blockv(64 * EMB_WAVES);
        if (a->out_dtype == CAD_F32)
            CAD_LAUNCH((embed_fwd_vec_kernel<float>), gridv, blockv, 0, stream, *a);
        else
            CAD_LAUNCH((embed_fwd_vec_kernel<bf16_t>), gridv, blockv, 0, stream, *a);
        return cad_after_launch();
    }
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
    const size_t shmem_vec = shmem * EMB_WAVES;
    if ((a->D % 4) == 0 && shmem_vec <= 64 * 1024 && ((uintptr_t)a->dout % 16) == 0 &&
        (a->dout_dtype == CAD_F32 || a->dout_dtype == CAD_BF16)) {
        dim3 gridv((unsigned)((tokens + EMB_TOK_PER_BLOCK_VEC - 1) / EMB_TOK_PER_BLOCK_VEC), (unsigned)a->n_strands), blockv(64 * EMB_WAVES);
        if (a->dout_dtype == CAD_F32)
            CAD_LAUNCH((embed_bwd_vec_kernel<float>), gridv, blockv, shmem_vec, stream, *a);
        else
            CAD_LAUNCH((embed_bwd_vec_kernel<bf16_t>), gridv, blockv, shmem_vec, stream, *a);
        return cad_after_launch();
    }
    dim3 grid((unsigned)((tokens + EMB_TOK_PER_BLOCK - 1) / EMB_TOK_PER_BLOCK), (unsigned)a->n_strands), block(256);
    if (a->dout_dtype == CAD_F32)
        CAD_LAUNCH((embed_bwd_kernel<float>), grid, block, shmem, stream, *a);
    else if (a->dout_dtype == CAD_BF16)
        CAD_LAUNCH((embed_bwd_kernel<bf16_t>), grid, block, shmem, stream, *a);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}
