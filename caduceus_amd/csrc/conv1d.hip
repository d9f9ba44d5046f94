// Depthwise causal conv1d (+bias +SiLU) along L on channel-major (E, SB, L) activations, with a per-row
// direction map (include/caduceus_hip.h, cad_conv1d_*).  HBM-bound: each thread produces 8 consecutive logical
// positions from one 16-byte (bf16) / 32-byte (fp32) vector plus its predecessor (halo through L1).
#include "cad_common.h"

namespace {

#define CV_VEC 8
#define CV_THREADS 256
#define CV_KMAX 4

// Load CV_VEC logical positions [p0, p0+8) of one row into out[0..7] (zeros outside [0, L)).
template <typename T>
__device__ __forceinline__ void load8(const T* row, int64_t p0, int64_t L, int rev, bool vec_ok, float* out) {
    if (vec_ok && p0 >= 0 && p0 + CV_VEC <= L) {
        const int64_t l0 = rev ? (L - p0 - CV_VEC) : p0;
        typedef struct __attribute__((aligned(sizeof(T) * CV_VEC))) {
            T v[CV_VEC];
        } vec_t;
        const vec_t tmp = *(const vec_t*)(row + l0);
#pragma unroll
        for (int j = 0; j < CV_VEC; ++j) out[j] = to_f32(tmp.v[rev ? (CV_VEC - 1 - j) : j]);
    } else {
#pragma unroll
        for (int j = 0; j < CV_VEC; ++j) {
            const int64_t p = p0 + j;
            out[j] = (p >= 0 && p < L) ? to_f32(row[cad_phys(p, L, rev)]) : 0.f;
        }
    }
}

template <typename T>
__device__ __forceinline__ void store8(T* row, int64_t p0, int64_t L, int rev, bool vec_ok, const float* v) {
    if (vec_ok && p0 + CV_VEC <= L) {
        const int64_t l0 = rev ? (L - p0 - CV_VEC) : p0;
        typedef struct __attribute__((aligned(sizeof(T) * CV_VEC))) {
            T v[CV_VEC];
        } vec_t;
        vec_t tmp;
#pragma unroll
        for (int j = 0; j < CV_VEC; ++j) tmp.v[rev ? (CV_VEC - 1 - j) : j] = from_f32<T>(v[j]);
        *(vec_t*)(row + l0) = tmp;
    } else {
#pragma unroll
        for (int j = 0; j < CV_VEC; ++j) {
            const int64_t p = p0 + j;
            if (p < L) row[cad_phys(p, L, rev)] = from_f32<T>(v[j]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(CV_THREADS) void conv1d_fwd_kernel(cad_conv1d_args a) {
    const int64_t rowid = blockIdx.x;  // e * SB + sb
    const int e = (int)(rowid / a.SB);
    const int64_t sb = rowid - (int64_t)e * a.SB;
    const int rev = sb < a.split ? a.rev_lo : a.rev_hi;
    const int64_t L = a.L;
    const T* x = (const T*)a.x + rowid * L;
    T* out = (T*)a.out + rowid * L;
    const bool vec_ok = (L % CV_VEC) == 0 && (((uintptr_t)a.x | (uintptr_t)a.out) % (sizeof(T) * CV_VEC)) == 0;
    float w[CV_KMAX];
#pragma unroll
    for (int k = 0; k < CV_KMAX; ++k) w[k] = k < a.K ? a.w[e * a.K + k] : 0.f;
    const float b = a.bias ? a.bias[e] : 0.f;
    const int64_t p0 = ((int64_t)blockIdx.y * CV_THREADS + threadIdx.x) * CV_VEC;
    if (p0 >= L) return;
    float xs[2 * CV_VEC];
    load8(x, p0 - CV_VEC, L, rev, vec_ok, xs);
    load8(x, p0, L, rev, vec_ok, xs + CV_VEC);
    float o[CV_VEC];
#pragma unroll
    for (int j = 0; j < CV_VEC; ++j) {
        float acc = b;
#pragma unroll
        for (int k = 0; k < CV_KMAX; ++k)
            if (k < a.K) acc += w[k] * xs[CV_VEC + j - (a.K - 1) + k];
        o[j] = acc * cad_sigmoid(acc);
    }
    store8(out, p0, L, rev, vec_ok, o);
}

// Backward.  dpre[p] = dout[p] * silu'(pre[p]);  dx[q] = sum_k w[k] * dpre[q + (K-1) - k];
// dw[k] = sum_p dpre[p] * x[p - (K-1) + k];  dbias = sum_p dpre[p].
template <typename T>
__global__ __launch_bounds__(CV_THREADS) void conv1d_bwd_kernel(cad_conv1d_bwd_args a) {
    __shared__ float red[CV_THREADS / 64][CV_KMAX + 1];
    const int64_t rowid = blockIdx.x;
    const int e = (int)(rowid / a.SB);
    const int64_t sb = rowid - (int64_t)e * a.SB;
    const int rev = sb < a.split ? a.rev_lo : a.rev_hi;
    const int64_t L = a.L;
    const int K = a.K;
    const T* x = (const T*)a.x + rowid * L;
    const T* dout = (const T*)a.dout + rowid * L;
    T* dx = (T*)a.dx + rowid * L;
    const bool vec_ok =
        (L % CV_VEC) == 0 && (((uintptr_t)a.x | (uintptr_t)a.dout | (uintptr_t)a.dx) % (sizeof(T) * CV_VEC)) == 0;
    float w[CV_KMAX];
#pragma unroll
    for (int k = 0; k < CV_KMAX; ++k) w[k] = k < K ? a.w[e * K + k] : 0.f;
    const float b = a.bias ? a.bias[e] : 0.f;
    const int64_t p0 = ((int64_t)blockIdx.y * CV_THREADS + threadIdx.x) * CV_VEC;
    float part[CV_KMAX + 1];
#pragma unroll
    for (int k = 0; k <= CV_KMAX; ++k) part[k] = 0.f;
    if (p0 < L) {
        float xs[3 * CV_VEC], gs[2 * CV_VEC];
        load8(x, p0 - CV_VEC, L, rev, vec_ok, xs);
        load8(x, p0, L, rev, vec_ok, xs + CV_VEC);
        load8(x, p0 + CV_VEC, L, rev, vec_ok, xs + 2 * CV_VEC);
        load8(dout, p0, L, rev, vec_ok, gs);
        load8(dout, p0 + CV_VEC, L, rev, vec_ok, gs + CV_VEC);
        // dpre at logical p0 .. p0 + 8 + (K-1) - 1
        float dpre[CV_VEC + CV_KMAX - 1];
#pragma unroll
        for (int j = 0; j < CV_VEC + CV_KMAX - 1; ++j) {
            float acc = b;
#pragma unroll
            for (int k = 0; k < CV_KMAX; ++k)
                if (k < K) acc += w[k] * xs[CV_VEC + j - (K - 1) + k];
            const float sg = cad_sigmoid(acc);
            dpre[j] = (j < CV_VEC + K - 1 && p0 + j < L) ? gs[j] * sg * (1.f + acc * (1.f - sg)) : 0.f;
        }
        float o[CV_VEC];
#pragma unroll
        for (int j = 0; j < CV_VEC; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < CV_KMAX; ++k)
                if (k < K) acc += w[k] * dpre[j + (K - 1) - k];
            o[j] = acc;
#pragma unroll
            for (int k = 0; k < CV_KMAX; ++k)
                if (k < K) part[k] += dpre[j] * xs[CV_VEC + j - (K - 1) + k];
            part[CV_KMAX] += dpre[j];
        }
        if (a.accumulate) {
            float prev[CV_VEC];
            load8(dx, p0, L, rev, vec_ok, prev);
#pragma unroll
            for (int j = 0; j < CV_VEC; ++j) o[j] += prev[j];
        }
        store8(dx, p0, L, rev, vec_ok, o);
    }
    // block reduction of dw / dbias partials
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k <= CV_KMAX; ++k) {
        float v = part[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x <= CV_KMAX) {
        float t = 0.f;
        for (int wv = 0; wv < CV_THREADS / 64; ++wv) t += red[wv][threadIdx.x];
        if (threadIdx.x < CV_KMAX) {
            if ((int)threadIdx.x < K && t != 0.f) atomicAdd(&a.dw[e * K + threadIdx.x], t);
        } else if (a.dbias && t != 0.f) {
            atomicAdd(&a.dbias[e], t);
        }
    }
}

}  // namespace

extern "C" int cad_conv1d_fwd(const cad_conv1d_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->x && a->w && a->out);
    CAD_CHECK_ARG(a->E > 0 && a->SB > 0 && a->L > 0 && a->K >= 1 && a->K <= CV_KMAX);
    CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB);
    CAD_CHECK_ARG((int64_t)a->E * a->SB < (1LL << 31));
    CadProfScope prof(2, stream);
    const int64_t per_block = (int64_t)CV_THREADS * CV_VEC;
    dim3 grid((unsigned)((int64_t)a->E * a->SB), (unsigned)((a->L + per_block - 1) / per_block)), block(CV_THREADS);
    if (grid.y > 65535u) return CAD_ERR_UNSUPPORTED;
    if (a->dtype == CAD_F32)
        CAD_LAUNCH((conv1d_fwd_kernel<float>), grid, block, 0, stream, *a);
    else if (a->dtype == CAD_BF16)
        CAD_LAUNCH((conv1d_fwd_kernel<bf16_t>), grid, block, 0, stream, *a);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}

extern "C" int cad_conv1d_bwd(const cad_conv1d_bwd_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->x && a->w && a->dout && a->dx && a->dw);
    CAD_CHECK_ARG(a->E > 0 && a->SB > 0 && a->L > 0 && a->K >= 1 && a->K <= CV_KMAX);
    CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB);
    CadProfScope prof(3, stream);
    const int64_t per_block = (int64_t)CV_THREADS * CV_VEC;
    dim3 grid((unsigned)((int64_t)a->E * a->SB), (unsigned)((a->L + per_block - 1) / per_block)), block(CV_THREADS);
    if (grid.y > 65535u) return CAD_ERR_UNSUPPORTED;
    if (a->dtype == CAD_F32)
        CAD_LAUNCH((conv1d_bwd_kernel<float>), grid, block, 0, stream, *a);
    else if (a->dtype == CAD_BF16)
        CAD_LAUNCH((conv1d_bwd_kernel<bf16_t>), grid, block, 0, stream, *a);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}
