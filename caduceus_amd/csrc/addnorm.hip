// Fused residual-add + RMSNorm/LayerNorm over both RCPS strands, with the reference's fused-path strand swap
// expressed as an output index map (include/caduceus_hip.h, cad_add_norm_*).
//
// One wave64 per token row; lane handles channels c = lane + 64*k.  HBM-bound elementwise kernel: every input
// byte is read once and every output byte written once; statistics stay in registers (wave xor-reduction).
#include "cad_common.h"
#include "cad_fp8.h"
#include "cad_stream.h"

namespace {

#define AN_KMAX 16  // D <= 64 * AN_KMAX = 1024
#define AN_WAVES 4  // rows per block-iteration

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}
// the value a bf16 / fp32 store of `o` leaves in memory (the e4m3 copy is taken from the ROUNDED normed activation, so that it equals
// cad_quant_rows_fp8 of the stored tensor bit for bit)
template <typename TY>
__device__ __forceinline__ float stored_value(float o) { return to_f32(from_f32<TY>(o)); }

template <typename TX, typename TY>
__global__ __launch_bounds__(64 * AN_WAVES) void add_norm_fwd_kernel(cad_add_norm_args a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t R = a.rows_per_strand;
    const int64_t nrows = R * a.n_strands;
    const int D = a.D;
    const int kiter = (D + 63) >> 6;
    const TX* x = (const TX*)a.x;
    TY* y = (TY*)a.y;
    const float invD = 1.0f / (float)D;
    for (int64_t row = (int64_t)blockIdx.x * AN_WAVES + wave; row < nrows; row += (int64_t)gridDim.x * AN_WAVES) {
        const int s = row >= R ? 1 : 0;  // n_strands <= 2 (a 64-bit division per row costs more than the row's arithmetic)
        const int64_t r = row - (int64_t)s * R;
        const int64_t orow = a.swap_flip ? ((int64_t)(a.n_strands - 1 - s) * R + r) : row;
        float v[AN_KMAX];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < AN_KMAX; ++k) {
            v[k] = 0.f;
            const int c = lane + 64 * k;
            if (k < kiter && c < D) {
                float t = to_f32(x[row * D + c]);
                if (a.residual_in) t += a.residual_in[row * D + c];
                v[k] = t;
                s1 += t;
                s2 += t * t;
            }
        }
        float mean = 0.f, rstd;
        if (a.is_rms) {
            s2 = wave_sum(s2);
            rstd = cad_rsqrt(s2 * invD + a.eps);
        } else {
            s1 = wave_sum(s1);
            mean = s1 * invD;
            float q = 0.f;
#pragma unroll
            for (int k = 0; k < AN_KMAX; ++k) {
                const int c = lane + 64 * k;
                if (k < kiter && c < D) {
                    const float d = v[k] - mean;
                    q += d * d;
                }
            }
            q = wave_sum(q);
            rstd = cad_rsqrt(q * invD + a.eps);
        }
        if (lane == 0) {
            a.rstd[row] = rstd;
            if (a.mean) a.mean[row] = mean;
        }
#pragma unroll
        for (int k = 0; k < AN_KMAX; ++k) {
            const int c = lane + 64 * k;
            if (k < kiter && c < D) {
                const int oc = a.swap_flip ? (D - 1 - c) : c;
                float o = (v[k] - mean) * rstd * a.weight[oc];
                if (a.bias) o += a.bias[oc];
                y[orow * D + oc] = from_f32<TY>(o);
                if (a.residual_out) a.residual_out[orow * D + oc] = v[k];
            }
        }
    }
}

// Backward: rows are indexed in the INPUT index space; dy / dres_out / sum_saved are read through the map.
// Rows one backward wave walks before the workgroup folds its dweight / dbias sums into the result (LDS + one atomic per
// channel): chosen per launch.  The fold is what a short walk pays for -- 0.262 ms at 8 rows, 0.249 at 32, 0.222 at 64 for the
// configs[2] layer (262144 rows of 256 channels; 128 rows leave too few workgroups: 0.327 ms) -- profiles/r03_ab_conv_addnorm.txt
#define ANB_ROWS_MIN 8
#define ANB_ROWS_MAX 64
#define ANB_TARGET_BLOCKS 1024
template <typename TX, typename TY>
__global__ __launch_bounds__(64 * AN_WAVES) void add_norm_bwd_kernel(cad_add_norm_bwd_args a, int rows_per_wave) {
    __shared__ float red[AN_WAVES][64 * AN_KMAX];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t R = a.rows_per_strand;
    const int64_t nrows = R * a.n_strands;
    const int D = a.D;
    const int kiter = (D + 63) >> 6;
    const TY* dy = (const TY*)a.dy;
    TX* dx = (TX*)a.dx;
    const float invD = 1.0f / (float)D;
    float dw[AN_KMAX], db[AN_KMAX];
#pragma unroll
    for (int k = 0; k < AN_KMAX; ++k) dw[k] = db[k] = 0.f;
    const int64_t row0 = ((int64_t)blockIdx.x * AN_WAVES + wave) * rows_per_wave;
    for (int i = 0; i < rows_per_wave; ++i) {
        const int64_t row = row0 + i;
        if (row >= nrows) break;
        const int s = row >= R ? 1 : 0;  // n_strands <= 2 (a 64-bit division per row costs more than the row's arithmetic)
        const int64_t r = row - (int64_t)s * R;
        const int64_t orow = a.swap_flip ? ((int64_t)(a.n_strands - 1 - s) * R + r) : row;
        const float rstd = a.rstd[row];
        const float mean = (a.is_rms || !a.mean) ? 0.f : a.mean[row];
        float g[AN_KMAX], xh[AN_KMAX];
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int k = 0; k < AN_KMAX; ++k) {
            g[k] = xh[k] = 0.f;
            const int c = lane + 64 * k;
            if (k < kiter && c < D) {
                const int oc = a.swap_flip ? (D - 1 - c) : c;
                const float dyv = to_f32(dy[orow * D + oc]);
                xh[k] = (a.sum_saved[orow * D + oc] - mean) * rstd;
                g[k] = dyv * a.weight[oc];
                sg += g[k];
                sgx += g[k] * xh[k];
                dw[k] += dyv * xh[k];
                db[k] += dyv;
            }
        }
        sgx = wave_sum(sgx) * invD;
        sg = a.is_rms ? 0.f : wave_sum(sg) * invD;
#pragma unroll
        for (int k = 0; k < AN_KMAX; ++k) {
            const int c = lane + 64 * k;
            if (k < kiter && c < D) {
                const int oc = a.swap_flip ? (D - 1 - c) : c;
                float d = rstd * (g[k] - sg - xh[k] * sgx);
                if (a.dres_out) d += a.dres_out[orow * D + oc];
                dx[row * D + c] = from_f32<TX>(d);
                if (a.dres_in) a.dres_in[row * D + c] = d;
            }
        }
    }
    // block-level reduction of the weight/bias gradients (indexed by OUTPUT channel), then one atomic per column
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !a.dbias) break;
#pragma unroll
        for (int k = 0; k < AN_KMAX; ++k) red[wave][lane + 64 * k] = pass == 0 ? dw[k] : db[k];
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += blockDim.x) {
            float t = 0.f;
            for (int w = 0; w < AN_WAVES; ++w) t += red[w][c];
            const int oc = a.swap_flip ? (D - 1 - c) : c;
            if (t != 0.f) atomicAdd(pass == 0 ? &a.dweight[oc] : &a.dbias[oc], t);
        }
        __syncthreads();
    }
}


// ---- vectorised variants (D % 4 == 0): a lane owns 4 consecutive channels per 256-channel step --------------------
// 8-byte (bf16) / 16-byte (fp32) accesses; with swap_flip the 4 outputs land, reversed, on 4 consecutive channels of
// the other strand, so they are still one vector store.
#define ANV_KMAX 4  // D <= 1024

template <typename T>
__device__ __forceinline__ void ld4(const T* p, float* o);
template <>
__device__ __forceinline__ void ld4<float>(const float* p, float* o) {
    struct __attribute__((aligned(16))) V { float f[4]; };
    const V t = *(const V*)p;
    o[0] = t.f[0], o[1] = t.f[1], o[2] = t.f[2], o[3] = t.f[3];
}
template <>
__device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float* o) {
    struct __attribute__((aligned(8))) V { uint32_t w[2]; };
    const V t = *(const V*)p;
    o[0] = cad_bits2f(t.w[0] << 16), o[1] = cad_bits2f(t.w[0] & 0xffff0000u);
    o[2] = cad_bits2f(t.w[1] << 16), o[3] = cad_bits2f(t.w[1] & 0xffff0000u);
}

// KMAX = 256-channel steps per lane: 1 for D <= 256 (a third of the registers of the general instantiation, twice the waves per
// SIMD), 2 for D <= 512 (configs[4])
template <typename TX, typename TY, int KMAX, bool FP8 = false>
__global__ __launch_bounds__(64 * AN_WAVES) void add_norm_fwd_vec_kernel(cad_add_norm_args a) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t R = a.rows_per_strand;
    const int64_t nrows = R * a.n_strands;
    const int D = a.D;
    const TX* x = (const TX*)a.x;
    TY* y = (TY*)a.y;
    const float invD = 1.0f / (float)D;
    for (int64_t row = (int64_t)blockIdx.x * AN_WAVES + wave; row < nrows; row += (int64_t)gridDim.x * AN_WAVES) {
        const int s = row >= R ? 1 : 0;  // n_strands <= 2 (a 64-bit division per row costs more than the row's arithmetic)
        const int64_t r = row - (int64_t)s * R;
        const int64_t orow = a.swap_flip ? ((int64_t)(a.n_strands - 1 - s) * R + r) : row;
        float v[KMAX][4];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c < D) {
                ld4<TX>(x + row * D + c, v[k]);
                if (a.residual_in) {
                    float t[4];
                    ld4<float>(a.residual_in + row * D + c, t);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[k][q] += t[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s1 += v[k][q];
                    s2 += v[k][q] * v[k][q];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[k][q] = 0.f;
            }
        }
        float mean = 0.f, rstd;
        if (a.is_rms) {
            s2 = wave_sum(s2);
            rstd = cad_rsqrt(s2 * invD + a.eps);
        } else {
            s1 = wave_sum(s1);
            mean = s1 * invD;
            float qq = 0.f;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int c = (lane + 64 * k) * 4;
                if (c < D) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float d = v[k][q] - mean;
                        qq += d * d;
                    }
                }
            }
            qq = wave_sum(qq);
            rstd = cad_rsqrt(qq * invD + a.eps);
        }
        if (lane == 0) {
            a.rstd[row] = rstd;
            if (a.mean) a.mean[row] = mean;
        }
        float ov[KMAX][4];   // the normed row as it is stored (only read again for the e4m3 copy)
        float amax = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c < D) {
                const int oc = a.swap_flip ? (D - 4 - c) : c;  // first of the 4 output channels
                float w[4], o[4], ro[4];
                ld4<float>(a.weight + oc, w);
                float b[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.bias) ld4<float>(a.bias + oc, b);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int qi = a.swap_flip ? (3 - q) : q;  // input element feeding output element q
                    o[q] = (v[k][qi] - mean) * rstd * w[q] + b[q];
                    ro[q] = v[k][qi];
                }
                cad_cvt_store_stream<CAD_STREAM_NORM, TY, 4>(y + orow * D + oc, o);
                if (a.residual_out) cad_cvt_store_stream<CAD_STREAM_NORM, float, 4>(a.residual_out + orow * D + oc, ro);
                if (FP8) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        ov[k][q] = stored_value<TY>(o[q]);
                        amax = fmaxf(amax, fabsf(ov[k][q]));
                    }
                }
            }
        }
        if (FP8) {
            // BASELINE configs[4] (fp8 projections): the e4m3 operand of the in_proj is written HERE, by the kernel that produces the
            // normed activations, with one scale per token (= output row) -- no separate quantisation pass over the tensor
            amax = wave_max(amax);
            const float scale = cad_fp8_row_scale(amax), inv = 1.0f / scale;
            uint8_t* qrow = (uint8_t*)a.y_fp8 + orow * D;
#pragma unroll
            for (int k = 0; k < KMAX; ++k) {
                const int c = (lane + 64 * k) * 4;
                if (c < D) {
                    const int oc = a.swap_flip ? (D - 4 - c) : c;
                    *(uint32_t*)(qrow + oc) = cad_pack_fp8x4(ov[k][0] * inv, ov[k][1] * inv, ov[k][2] * inv, ov[k][3] * inv);
                }
            }
            if (lane == 0) a.y_scale[orow] = scale;
        }
    }
}

template <typename TX, typename TY, int KMAX>
__global__ __launch_bounds__(64 * AN_WAVES) void add_norm_bwd_vec_kernel(cad_add_norm_bwd_args a, int rows_per_wave) {
    __shared__ float red[AN_WAVES][64 * KMAX * 4];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t R = a.rows_per_strand;
    const int64_t nrows = R * a.n_strands;
    const int D = a.D;
    const TY* dy = (const TY*)a.dy;
    TX* dx = (TX*)a.dx;
    const float invD = 1.0f / (float)D;
    float dw[KMAX][4], db[KMAX][4];  // indexed by OUTPUT channel oc + q
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) dw[k][q] = db[k][q] = 0.f;
    float wreg[KMAX][4];  // the norm weight of this lane's OUTPUT channels, loaded once
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int c = (lane + 64 * k) * 4;
        if (c < D) {
            ld4<float>(a.weight + (a.swap_flip ? (D - 4 - c) : c), wreg[k]);
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) wreg[k][q] = 0.f;
        }
    }
    const int64_t row0 = ((int64_t)blockIdx.x * AN_WAVES + wave) * rows_per_wave;
    for (int i = 0; i < rows_per_wave; ++i) {
        const int64_t row = row0 + i;
        if (row >= nrows) break;
        const int s = row >= R ? 1 : 0;  // n_strands <= 2 (a 64-bit division per row costs more than the row's arithmetic)
        const int64_t r = row - (int64_t)s * R;
        const int64_t orow = a.swap_flip ? ((int64_t)(a.n_strands - 1 - s) * R + r) : row;
        const float rstd = a.rstd[row];
        const float mean = (a.is_rms || !a.mean) ? 0.f : a.mean[row];
        float g[KMAX][4], xh[KMAX][4];  // output-channel order
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c < D) {
                const int oc = a.swap_flip ? (D - 4 - c) : c;
                float dyv[4], sm[4];
                ld4<TY>(dy + orow * D + oc, dyv);
                ld4<float>(a.sum_saved + orow * D + oc, sm);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    xh[k][q] = (sm[q] - mean) * rstd;
                    g[k][q] = dyv[q] * wreg[k][q];
                    sg += g[k][q];
                    sgx += g[k][q] * xh[k][q];
                    dw[k][q] += dyv[q] * xh[k][q];
                    db[k][q] += dyv[q];
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) g[k][q] = xh[k][q] = 0.f;
            }
        }
        sgx = wave_sum(sgx) * invD;
        sg = a.is_rms ? 0.f : wave_sum(sg) * invD;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = (lane + 64 * k) * 4;
            if (c < D) {
                const int oc = a.swap_flip ? (D - 4 - c) : c;
                float dro[4] = {0.f, 0.f, 0.f, 0.f}, d[4];
                if (a.dres_out) ld4<float>(a.dres_out + orow * D + oc, dro);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int qo = a.swap_flip ? (3 - q) : q;  // output element that input element q produced
                    d[q] = rstd * (g[k][qo] - sg - xh[k][qo] * sgx) + dro[qo];
                }
                cad_cvt_store_stream<CAD_STREAM_NORM, TX, 4>(dx + row * D + c, d);
                if (a.dres_in) cad_cvt_store_stream<CAD_STREAM_NORM, float, 4>(a.dres_in + row * D + c, d);
            }
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1 && !a.dbias) break;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = (lane + 64 * k) * 4;
            const int oc = a.swap_flip ? (D - 4 - c) : c;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c < D) red[wave][oc + q] = pass == 0 ? dw[k][q] : db[k][q];
        }
        __syncthreads();
        for (int c = threadIdx.x; c < D; c += blockDim.x) {
            float t = 0.f;
            for (int w = 0; w < AN_WAVES; ++w) t += red[w][c];
            if (t != 0.f) atomicAdd(pass == 0 ? &a.dweight[c] : &a.dbias[c], t);
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int cad_add_norm_fwd(const cad_add_norm_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->x && a->weight && a->y && a->rstd);
    CAD_CHECK_ARG(a->rows_per_strand > 0 && a->D > 0 && (a->n_strands == 1 || a->n_strands == 2));
    CAD_CHECK_ARG(a->is_rms || a->mean);
    if (a->D > 64 * AN_KMAX) return CAD_ERR_UNSUPPORTED;
    CadProfScope prof(4, stream);
    const int64_t nrows = a->rows_per_strand * a->n_strands;
    int64_t nb = (nrows + AN_WAVES - 1) / AN_WAVES;
    if (nb > 8192) nb = 8192;
    dim3 grid((unsigned)nb), block(64 * AN_WAVES);
    const bool vec = (a->D % 4) == 0 && (((uintptr_t)a->x | (uintptr_t)a->residual_in | (uintptr_t)a->weight |
                                          (uintptr_t)a->bias | (uintptr_t)a->y | (uintptr_t)a->residual_out) % 16) == 0;
    if (a->y_fp8) {  // the e4m3 copy is an epilogue of the vector kernels (every production shape)
        CAD_CHECK_ARG(a->y_scale != nullptr);
        if (!vec || a->D > 512 || ((uintptr_t)a->y_fp8 % 4) != 0) return CAD_ERR_UNSUPPORTED;
    }
#define AN_FWD(TX, TY)                                                                  \
    do {                                                                                \
        if (vec && a->y_fp8 && a->D <= 256)                                             \
            CAD_LAUNCH((add_norm_fwd_vec_kernel<TX, TY, 1, true>), grid, block, 0, stream, *a);  \
        else if (vec && a->y_fp8)                                                       \
            CAD_LAUNCH((add_norm_fwd_vec_kernel<TX, TY, 2, true>), grid, block, 0, stream, *a);  \
        else if (vec && a->D <= 256)                                                    \
            CAD_LAUNCH((add_norm_fwd_vec_kernel<TX, TY, 1>), grid, block, 0, stream, *a);  \
        else if (vec && a->D <= 512)                                                    \
            CAD_LAUNCH((add_norm_fwd_vec_kernel<TX, TY, 2>), grid, block, 0, stream, *a);  \
        else if (vec)                                                                   \
            CAD_LAUNCH((add_norm_fwd_vec_kernel<TX, TY, ANV_KMAX>), grid, block, 0, stream, *a);  \
        else                                                                            \
            CAD_LAUNCH((add_norm_fwd_kernel<TX, TY>), grid, block, 0, stream, *a);      \
    } while (0)
    if (a->x_dtype == CAD_F32 && a->y_dtype == CAD_F32)
        AN_FWD(float, float);
    else if (a->x_dtype == CAD_F32 && a->y_dtype == CAD_BF16)
        AN_FWD(float, bf16_t);
    else if (a->x_dtype == CAD_BF16 && a->y_dtype == CAD_BF16)
        AN_FWD(bf16_t, bf16_t);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}

extern "C" int cad_add_norm_bwd(const cad_add_norm_bwd_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->dy && a->sum_saved && a->rstd && a->weight && a->dx && a->dweight);
    CAD_CHECK_ARG(a->rows_per_strand > 0 && a->D > 0 && (a->n_strands == 1 || a->n_strands == 2));
    if (a->D > 64 * AN_KMAX) return CAD_ERR_UNSUPPORTED;
    CadProfScope prof(5, stream);
    const int64_t nrows = a->rows_per_strand * a->n_strands;
    int64_t rpw = nrows / ((int64_t)AN_WAVES * ANB_TARGET_BLOCKS);
    rpw = rpw < ANB_ROWS_MIN ? ANB_ROWS_MIN : (rpw > ANB_ROWS_MAX ? ANB_ROWS_MAX : rpw);
    const int rows_per_wave = (int)rpw;
    const int64_t per_block = (int64_t)AN_WAVES * rows_per_wave;
    dim3 grid((unsigned)((nrows + per_block - 1) / per_block)), block(64 * AN_WAVES);
    const bool vec = (a->D % 4) == 0 && (((uintptr_t)a->dy | (uintptr_t)a->dres_out | (uintptr_t)a->sum_saved |
                                          (uintptr_t)a->weight | (uintptr_t)a->dx | (uintptr_t)a->dres_in) % 16) == 0;
#define AN_BWD(TX, TY)                                                                  \
    do {                                                                                \
        if (vec && a->D <= 256)                                                         \
            CAD_LAUNCH((add_norm_bwd_vec_kernel<TX, TY, 1>), grid, block, 0, stream, *a, rows_per_wave);  \
        else if (vec && a->D <= 512)                                                    \
            CAD_LAUNCH((add_norm_bwd_vec_kernel<TX, TY, 2>), grid, block, 0, stream, *a, rows_per_wave);  \
        else if (vec)                                                                   \
            CAD_LAUNCH((add_norm_bwd_vec_kernel<TX, TY, ANV_KMAX>), grid, block, 0, stream, *a, rows_per_wave);  \
        else                                                                            \
            CAD_LAUNCH((add_norm_bwd_kernel<TX, TY>), grid, block, 0, stream, *a, rows_per_wave);      \
    } while (0)
    if (a->x_dtype == CAD_F32 && a->y_dtype == CAD_F32)
        AN_BWD(float, float);
    else if (a->x_dtype == CAD_F32 && a->y_dtype == CAD_BF16)
        AN_BWD(float, bf16_t);
    else if (a->x_dtype == CAD_BF16 && a->y_dtype == CAD_BF16)
        AN_BWD(bf16_t, bf16_t);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}
