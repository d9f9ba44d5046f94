/* CPU restatement (plain C + OpenMP, fp32) of the selective scan and causal conv1d the reference reaches through
 * mamba_ssm.Mamba.forward (/root/reference/caduceus/modeling_caduceus.py:11,128,130; algorithm of the un-vendored
 * dependencies mamba-ssm==1.2.0.post1 `selective_scan_ref` and causal-conv1d==1.2.0.post2, restated from their
 * published definition -- SURVEY.md section 7.2).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: used by tests/ as a second, independent checker at sizes where the
 * torch loop in oracle_model.py is too slow, and by bench.py's `cpu_baseline` leg ("port", all host cores).
 * Pinned by tests/test_oracle_golden.py against the committed third-party vectors (tests/golden/scan_op_*.npz,
 * conv_op.npz) and against oracle_model.selective_scan.
 *
 * Layout: batch-major like the reference: u, delta, z, out (b, E, L); A (E, N); B, C (b, N, L); D, bias (E).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }
static inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

int cad_oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* out = (sum_n C_n h_n + D u) * silu(z),  h_l = exp(dt_l A) h_{l-1} + dt_l B_l u_l,  dt = softplus(delta + bias) */
void cad_oracle_scan_fwd(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                         const float* D, const float* z, const float* bias, float* out, int64_t nb, int64_t E,
                         int64_t L, int64_t N) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < nb; ++b)
        for (int64_t e = 0; e < E; ++e) {
            float h[64];
            for (int64_t n = 0; n < N; ++n) h[n] = 0.f;
            const float* ur = u + (b * E + e) * L;
            const float* dr = delta + (b * E + e) * L;
            const float* zr = z ? z + (b * E + e) * L : NULL;
            float* orow = out + (b * E + e) * L;
            for (int64_t l = 0; l < L; ++l) {
                const float dt = softplusf(dr[l] + (bias ? bias[e] : 0.f));
                float y = 0.f;
                for (int64_t n = 0; n < N; ++n) {
                    h[n] = expf(dt * A[e * N + n]) * h[n] + dt * Bm[(b * N + n) * L + l] * ur[l];
                    y += h[n] * Cm[(b * N + n) * L + l];
                }
                y += (D ? D[e] : 0.f) * ur[l];
                if (zr) y *= zr[l] * sigmoidf(zr[l]);
                orow[l] = y;
            }
        }
}

/* Gradients of the above.  dB, dC are reduced over channels through per-thread buffers (deterministic given the
 * thread count).  All gradient outputs are overwritten. */
void cad_oracle_scan_bwd(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                         const float* D, const float* z, const float* bias, const float* dout, float* du,
                         float* ddelta, float* dA, float* dB, float* dC, float* dD, float* dz, float* dbias,
                         int64_t nb, int64_t E, int64_t L, int64_t N) {
    const int nt = cad_oracle_num_threads();
    float* dBt = (float*)calloc((size_t)nt * nb * N * L, sizeof(float));
    float* dCt = (float*)calloc((size_t)nt * nb * N * L, sizeof(float));
    float* dAt = (float*)calloc((size_t)nt * E * N, sizeof(float));
    float* dDt = (float*)calloc((size_t)nt * E, sizeof(float));
    float* dbt = (float*)calloc((size_t)nt * E, sizeof(float));
#pragma omp parallel
    {
#ifdef _OPENMP
        const int tid = omp_get_thread_num();
#else
        const int tid = 0;
#endif
        float* hs = (float*)malloc((size_t)(L + 1) * N * sizeof(float)); /* h_{-1..L-1} */
        float* as = (float*)malloc((size_t)L * N * sizeof(float));
        float* dts = (float*)malloc((size_t)L * sizeof(float));
        float* dBp = dBt + (size_t)tid * nb * N * L;
        float* dCp = dCt + (size_t)tid * nb * N * L;
#pragma omp for collapse(2) schedule(static)
        for (int64_t b = 0; b < nb; ++b)
            for (int64_t e = 0; e < E; ++e) {
                const int64_t ro = (b * E + e) * L;
                const float bi = bias ? bias[e] : 0.f, Dv = D ? D[e] : 0.f;
                for (int64_t n = 0; n < N; ++n) hs[n] = 0.f;
                for (int64_t l = 0; l < L; ++l) {
                    const float dt = softplusf(delta[ro + l] + bi);
                    dts[l] = dt;
                    for (int64_t n = 0; n < N; ++n) {
                        const float a = expf(dt * A[e * N + n]);
                        as[l * N + n] = a;
                        hs[(l + 1) * N + n] = a * hs[l * N + n] + dt * Bm[(b * N + n) * L + l] * u[ro + l];
                    }
                }
                float G[64];
                for (int64_t n = 0; n < N; ++n) G[n] = 0.f;
                for (int64_t l = L - 1; l >= 0; --l) {
                    float y = Dv * u[ro + l];
                    for (int64_t n = 0; n < N; ++n) y += hs[(l + 1) * N + n] * Cm[(b * N + n) * L + l];
                    float dy = dout[ro + l];
                    if (z) {
                        const float zz = z[ro + l], sg = sigmoidf(zz);
                        dz[ro + l] = dy * y * sg * (1.f + zz * (1.f - sg));
                        dy *= zz * sg;
                    }
                    const float dt = dts[l], uu = u[ro + l];
                    float ddt = 0.f, ddu = dy * Dv;
                    dDt[(size_t)tid * E + e] += dy * uu;
                    for (int64_t n = 0; n < N; ++n) {
                        const float a = as[l * N + n], Bv = Bm[(b * N + n) * L + l], Cv = Cm[(b * N + n) * L + l];
                        const float g = Cv * dy + G[n];
                        G[n] = a * g;
                        const float t = g * hs[l * N + n] * a;
                        ddt += t * A[e * N + n] + uu * g * Bv;
                        ddu += dt * g * Bv;
                        dAt[((size_t)tid * E + e) * N + n] += t * dt;
                        dBp[(b * N + n) * L + l] += g * dt * uu;
                        dCp[(b * N + n) * L + l] += dy * hs[(l + 1) * N + n];
                    }
                    const float xr = delta[ro + l] + bi;
                    const float dd = ddt * (xr > 20.f ? 1.f : sigmoidf(xr));
                    ddelta[ro + l] = dd;
                    du[ro + l] = ddu;
                    dbt[(size_t)tid * E + e] += dd;
                }
            }
        free(hs);
        free(as);
        free(dts);
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nb * N * L; ++i) {
        float sb = 0.f, sc = 0.f;
        for (int t = 0; t < nt; ++t) {
            sb += dBt[(size_t)t * nb * N * L + i];
            sc += dCt[(size_t)t * nb * N * L + i];
        }
        dB[i] = sb;
        dC[i] = sc;
    }
    for (int64_t i = 0; i < E * N; ++i) {
        float s = 0.f;
        for (int t = 0; t < nt; ++t) s += dAt[(size_t)t * E * N + i];
        dA[i] = s;
    }
    for (int64_t i = 0; i < E; ++i) {
        float s1 = 0.f, s2 = 0.f;
        for (int t = 0; t < nt; ++t) {
            s1 += dDt[(size_t)t * E + i];
            s2 += dbt[(size_t)t * E + i];
        }
        if (dD) dD[i] = s1;
        if (dbias) dbias[i] = s2;
    }
    free(dBt);
    free(dCt);
    free(dAt);
    free(dDt);
    free(dbt);
}

/* out[l] = silu(bias + sum_k w[k] x[l-(K-1)+k]);  x, out: (b, E, L); w: (E, K) */
void cad_oracle_conv_fwd(const float* x, const float* w, const float* bias, float* out, int64_t nb, int64_t E,
                         int64_t L, int64_t K) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t b = 0; b < nb; ++b)
        for (int64_t e = 0; e < E; ++e) {
            const float* xr = x + (b * E + e) * L;
            float* orow = out + (b * E + e) * L;
            for (int64_t l = 0; l < L; ++l) {
                float acc = bias ? bias[e] : 0.f;
                for (int64_t k = 0; k < K; ++k) {
                    const int64_t j = l - (K - 1) + k;
                    if (j >= 0) acc += w[e * K + k] * xr[j];
                }
                orow[l] = acc * sigmoidf(acc);
            }
        }
}
