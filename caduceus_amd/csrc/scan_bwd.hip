// Selective SSM scan, backward (include/caduceus_hip.h, cad_scan_bwd).  See scan_common.h for the decomposition.
//
// Per chunk (processed from the logical END of the row to its start, because the state gradient flows backwards)
// and per state pair:
//   1. recompute h over the chunk from the chunk-start state saved by the forward (serial + wave scan, as fwd);
//   2. reverse scan of  G_i = a_i * (c_i + G_{i+1}),  c_i = C_i * dy_i   (G_i = gradient flowing into h_{i-1});
//      g_i = c_i + G_{i+1} is dL/dh_i;
//   3. per item:  d(dt) += g*h_{i-1}*a*A + u*<g,B>,  dA += g*h_{i-1}*a*dt,  du += dt*<g,B>,
//                 dB_i = g*dt*u,  dC_i = dy*h_i  -- the last two are summed over the SC_W channels of the workgroup
//                 in LDS (ds_add_f32) and then added to the fp32 global buffers with one atomic per element.
#include "scan_common.h"

namespace {

__device__ __forceinline__ f32x2 wave_sum2(f32x2 v) { return f2(wave_sum1(v[0]), wave_sum1(v[1])); }

template <typename T>
__global__ __launch_bounds__(64 * SC_W) void scan_bwd_kernel(cad_scan_bwd_args a) {
    CAD_DYN_SMEM(float, smem);  // [2 buffers][B,C][SC_TILE] inputs, then [dB,dC][SC_TILE] accumulators
    float* accB = smem + 4 * SC_TILE;
    float* accC = accB + SC_TILE;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int64_t sb = blockIdx.y;
    const int e_raw = blockIdx.x * SC_W + wave;
    const bool act = e_raw < a.E;
    const int e = act ? e_raw : a.E - 1;
    const int rev = sb < a.split ? a.rev_lo : a.rev_hi;
    const int64_t L = a.L, SB = a.SB;
    const int N = a.N, NP = (N + 1) >> 1;
    const int64_t row_off = ((int64_t)e * SB + sb) * L;
    const T* u_row = (const T*)a.u + row_off;
    const T* d_row = (const T*)a.delta + row_off;
    const T* z_row = a.z ? (const T*)a.z + row_off : nullptr;
    const T* g_row = (const T*)a.dout + row_off;
    T* du_row = (T*)a.du + row_off;
    T* dd_row = (T*)a.ddelta + row_off;
    T* dz_row = a.dz ? (T*)a.dz + row_off : nullptr;
    const T* Bm = (const T*)a.Bm;
    const T* Cm = (const T*)a.Cm;
    const bool vec_ok =
        ((L * sizeof(T)) % 16) == 0 && (((uintptr_t)a.u | (uintptr_t)a.delta | (uintptr_t)a.z | (uintptr_t)a.dout |
                                         (uintptr_t)a.du | (uintptr_t)a.ddelta | (uintptr_t)a.dz) %
                                        16) == 0;
    const float Dv = a.D ? a.D[e] : 0.f;
    const float bias = a.delta_bias ? a.delta_bias[e] : 0.f;
    const int64_t nchunks = (L + SC_CHUNK - 1) / SC_CHUNK;
    const float keep = act ? 1.f : 0.f;  // padding waves (E % SC_W != 0) contribute nothing

    for (int i = threadIdx.x; i < 2 * SC_TILE; i += blockDim.x) accB[i] = 0.f;
    __syncthreads();

    f32x2 carryG = f2(0.f);  // lane np: G flowing out of the later chunk into this one, for pair np
    f32x2 dAacc = f2(0.f);   // lane np: dA of pair np
    float dDacc = 0.f, dbacc = 0.f;

    for (int64_t c = nchunks - 1; c >= 0; --c) {
        const int64_t base = c * SC_CHUNK;
        const int64_t p0 = base + (int64_t)lane * SC_S;
        float uu[SC_S], dt[SC_S], dy[SC_S], ddt[SC_S], ddu[SC_S], y[SC_S];
        sc_load(u_row, p0, L, rev, vec_ok, uu);
        sc_load(d_row, p0, L, rev, vec_ok, dt);
        sc_load(g_row, p0, L, rev, vec_ok, dy);
        if (z_row) {
            float zz[SC_S];
            sc_load(z_row, p0, L, rev, vec_ok, zz);
#pragma unroll
            for (int i = 0; i < SC_S; ++i) dy[i] *= zz[i] * cad_sigmoid(zz[i]);
        }
#pragma unroll
        for (int i = 0; i < SC_S; ++i) {
            const bool ok = p0 + i < L;
            dt[i] = ok ? cad_softplus(dt[i] + bias) : 0.f;
            dy[i] = ok ? dy[i] * keep : 0.f;
            y[i] = Dv * uu[i];
            ddt[i] = 0.f;
            ddu[i] = dy[i] * Dv;
            dDacc += dy[i] * uu[i];
        }
        sc_stage_bc(smem, smem + SC_TILE, Bm, Cm, 0, N, SB, sb, base, L, rev);
        __syncthreads();
        for (int np = 0; np < NP; ++np) {
            const int buf = np & 1;
            if (np + 1 < NP)
                sc_stage_bc(smem + (buf ^ 1) * 2 * SC_TILE, smem + (buf ^ 1) * 2 * SC_TILE + SC_TILE, Bm, Cm,
                            2 * (np + 1), N, SB, sb, base, L, rev);
            const float* tB = smem + buf * 2 * SC_TILE + lane * SC_ROW;
            const float* tC = tB + SC_TILE;
            const int n0 = 2 * np;
            const f32x2 Av = f2(a.A[e * N + n0], (n0 + 1 < N) ? a.A[e * N + n0 + 1] : 0.f);
            const f32x2 A2 = Av * f2(CAD_LOG2E);
            const float* st = a.chunk_state + ((((int64_t)e * SB + sb) * nchunks + c) * NP + np) * 2;
            const f32x2 hin = f2(st[0], st[1]);
            // 1. forward recompute: serial totals, wave scan, then the true h_i
            f32x2 av[SC_S], hs[SC_S];
            f32x2 acc_a = f2(1.f), acc_h = f2(0.f);
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
                av[i] = exp2_2(f2(dt[i]) * A2);
                hs[i] = f2(dt[i] * uu[i]) * ld2(tB + 2 * i);  // b_i
                acc_h = av[i] * acc_h + hs[i];
                acc_a = acc_a * av[i];
            }
            f32x2 PA = acc_a, PH = acc_h;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const f32x2 ua = shfl_up2(PA, d), uh = shfl_up2(PH, d);
                if (lane >= d) {
                    PH = PA * uh + PH;
                    PA = PA * ua;
                }
            }
            f32x2 ea = shfl_up2(PA, 1), eh = shfl_up2(PH, 1);
            if (lane == 0) {
                ea = f2(1.f);
                eh = f2(0.f);
            }
            const f32x2 h0 = ea * hin + eh;  // state entering this lane's segment
            {
                f32x2 h = h0;
#pragma unroll
                for (int i = 0; i < SC_S; ++i) {
                    h = av[i] * h + hs[i];
                    hs[i] = h;  // h_i
                    y[i] += dot2(ld2(tC + 2 * i), h);
                }
            }
            // 2. reverse scan of G
            f32x2 RG = f2(0.f);
#pragma unroll
            for (int i = SC_S - 1; i >= 0; --i) RG = av[i] * (ld2(tC + 2 * i) * f2(dy[i]) + RG);
            f32x2 QA = acc_a, QG = RG;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const f32x2 ua = shfl_down2(QA, d), ug = shfl_down2(QG, d);
                if (lane + d < 64) {
                    QG = QA * ug + QG;
                    QA = QA * ua;
                }
            }
            f32x2 fa = shfl_down2(QA, 1), fg = shfl_down2(QG, 1);
            if (lane == 63) {
                fa = f2(1.f);
                fg = f2(0.f);
            }
            const f32x2 gin = shfl2(carryG, np);
            f32x2 G = fa * gin + fg;  // G_{i+1} for this lane's last item
            const f32x2 newc = shfl2(QA * gin + QG, 0);
            if (lane == np) carryG = newc;
            // 3. gradients
            f32x2 dAp = f2(0.f);
            float* aB = accB + lane * SC_ROW;
            float* aC = accC + lane * SC_ROW;
#pragma unroll
            for (int i = SC_S - 1; i >= 0; --i) {
                const f32x2 Bv = ld2(tB + 2 * i);
                const f32x2 g = ld2(tC + 2 * i) * f2(dy[i]) + G;
                G = av[i] * g;
                const f32x2 hprev = (i > 0) ? hs[i > 0 ? i - 1 : 0] : h0;
                const f32x2 t = g * hprev * av[i];
                const float gB = dot2(g, Bv);
                ddt[i] += dot2(t, Av) + uu[i] * gB;
                ddu[i] += dt[i] * gB;
                dAp = dAp + t * f2(dt[i]);
                const f32x2 dBv = g * f2(dt[i] * uu[i]);
                const f32x2 dCv = hs[i] * f2(dy[i]);
                atomicAdd(aB + 2 * i, dBv[0]);
                atomicAdd(aB + 2 * i + 1, dBv[1]);
                atomicAdd(aC + 2 * i, dCv[0]);
                atomicAdd(aC + 2 * i + 1, dCv[1]);
            }
            dAp = wave_sum2(dAp);
            if (lane == np) dAacc = dAacc + dAp * f2(keep);
            __syncthreads();  // all channels of the workgroup have added their dB/dC; tile buf is free again
            // flush the channel-summed dB / dC tile of this pair to global (fp32 atomics), then clear it
            for (int idx = threadIdx.x; idx < 2 * SC_CHUNK; idx += blockDim.x) {
                const int s = idx / SC_CHUNK;
                const int tok = idx - s * SC_CHUNK;
                const int64_t p = base + tok;
                const int o = (tok / SC_S) * SC_ROW + (tok % SC_S) * 2 + s;
                if (p < L && n0 + s < N) {
                    const int64_t off = ((int64_t)(n0 + s) * SB + sb) * L + cad_phys(p, L, rev);
                    atomicAdd(a.dB + off, accB[o]);
                    atomicAdd(a.dC + off, accC[o]);
                }
                accB[o] = 0.f;
                accC[o] = 0.f;
            }
            __syncthreads();
        }
        // per-item outputs of this chunk
        float dl[SC_S];
        sc_load(d_row, p0, L, rev, vec_ok, dl);
#pragma unroll
        for (int i = 0; i < SC_S; ++i) {
            const float xraw = dl[i] + bias;
            const float sg = xraw > 20.f ? 1.f : cad_sigmoid(xraw);
            ddt[i] = (p0 + i < L) ? ddt[i] * sg : 0.f;
            dbacc += ddt[i];
        }
        if (act) {
            sc_store(du_row, p0, L, rev, vec_ok, ddu);
            sc_store(dd_row, p0, L, rev, vec_ok, ddt);
        }
        if (dz_row) {
            float zz[SC_S], go[SC_S];
            sc_load(z_row, p0, L, rev, vec_ok, zz);
            sc_load(g_row, p0, L, rev, vec_ok, go);
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
                const float sg = cad_sigmoid(zz[i]);
                go[i] = go[i] * y[i] * sg * (1.f + zz[i] * (1.f - sg));
            }
            if (act) sc_store(dz_row, p0, L, rev, vec_ok, go);
        }
    }
    // per-channel parameter gradients
    if (act) {
        if (lane < NP) {
            const int n0 = 2 * lane;
            atomicAdd(a.dA + e * N + n0, dAacc[0]);
            if (n0 + 1 < N) atomicAdd(a.dA + e * N + n0 + 1, dAacc[1]);
        }
    }
    dDacc = wave_sum1(dDacc);
    dbacc = wave_sum1(dbacc);
    if (act && lane == 0) {
        if (a.dD) atomicAdd(a.dD + e, dDacc);
        if (a.ddelta_bias) atomicAdd(a.ddelta_bias + e, dbacc);
    }
}

}  // namespace

extern "C" int cad_scan_bwd(const cad_scan_bwd_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->u && a->delta && a->A && a->Bm && a->Cm && a->dout && a->chunk_state);
    CAD_CHECK_ARG(a->du && a->ddelta && a->dA && a->dB && a->dC);
    CAD_CHECK_ARG((a->z == nullptr) == (a->dz == nullptr));
    CAD_CHECK_ARG(a->E > 0 && a->SB > 0 && a->L > 0 && a->N > 0 && a->N <= SC_NMAX);
    CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB && a->SB <= 65535);
    CadProfScope prof(1, stream);
    dim3 grid((unsigned)((a->E + SC_W - 1) / SC_W), (unsigned)a->SB), block(64 * SC_W);
    const size_t shmem = (size_t)6 * SC_TILE * sizeof(float);
    if (a->dtype == CAD_F32)
        CAD_LAUNCH((scan_bwd_kernel<float>), grid, block, shmem, stream, *a);
    else if (a->dtype == CAD_BF16)
        CAD_LAUNCH((scan_bwd_kernel<bf16_t>), grid, block, shmem, stream, *a);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}
