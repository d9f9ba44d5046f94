// Selective SSM scan, forward, token-major (include/caduceus_hip.h: cad_scan_tm_fwd / cad_scan_tm_fwd_multi).
// See scan_tm.h for the decomposition.  Three launches:
//   1. aggregate : every chunk but the last, from a zero state  ->  S_k (N x E) and sum_k(dt) (E)
//   2. combine   : sequential over chunks, in place             ->  state at the START of chunk k
//   3. final     : every chunk from its true start state        ->  out, and the state at every TM_BLK-th position
#include "scan_tm.h"

namespace {

struct TmFwdSets {
    cad_scan_tm_args s[TM_MAXSETS];
};

// compiler-level fence: memory operations are not moved across it (keeps the scalar B/C loads of position p+2 from
// being hoisted above position p, which would spill SGPRs)
#ifdef CAD_EMU
#define TM_FENCE() do {} while (0)
#else
#define TM_FENCE() asm volatile("" ::: "memory")
#endif

// MODE 0 = aggregate, 1 = final.  FULL: N == NS (no per-state bounds checks anywhere).
template <typename T, int NS, int CPL, int MODE, bool FULL>
__global__ __launch_bounds__(TM_THREADS) void scan_tm_fwd_kernel(TmFwdSets sets) {
    const cad_scan_tm_args& a = sets.s[blockIdx.z];
    const int E = a.E, N = FULL ? NS : a.N;
    const int64_t L = a.L, SB = a.SB;
    const int EB = (int)tm_div_up(E, TM_THREADS * CPL);
    const int64_t sb = blockIdx.y / EB;
    const int c0 = ((int)(blockIdx.y % EB) * TM_THREADS + (int)threadIdx.x) * CPL;  // first channel of this lane
    const bool act = c0 < E;  // E % CPL == 0 (host-checked): a lane is entirely inside or outside
    const int cc = act ? c0 : 0;
    const int rev = sb < a.split ? a.rev_lo : a.rev_hi;
    const int64_t k = blockIdx.x;  // chunk
    const int64_t nchunks = tm_div_up(L, TM_TC);
    const int64_t p_begin = k * TM_TC;
    const int64_t p_end = (p_begin + TM_TC < L) ? p_begin + TM_TC : L;
    const int64_t tstep = rev ? -1 : 1;
    const int64_t row0 = sb * L;

    float A2[CPL][NS], h[CPL][NS], Dv[CPL], bias[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        Dv[c] = a.D ? a.D[cc + c] : 0.f;
        bias[c] = a.delta_bias ? a.delta_bias[cc + c] : 0.f;
#pragma unroll
        for (int n = 0; n < NS; ++n) {
            A2[c][n] = n < N ? a.A[(int64_t)(cc + c) * N + n] * CAD_LOG2E : 0.f;
            h[c][n] = 0.f;
        }
    }
    float* agg = a.scratch + ((sb * nchunks + k) * N) * E + cc;                       // [sb][k][n][E]
    float* sdt = a.scratch + SB * nchunks * N * E + (sb * nchunks + k) * E + cc;       // [sb][k][E]
    if (MODE == 1) {
#pragma unroll
        for (int c = 0; c < CPL; ++c)
#pragma unroll
            for (int n = 0; n < NS; ++n)
                if (n < N) h[c][n] = agg[(int64_t)n * E + c];
    }
    float sum_dt[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) sum_dt[c] = 0.f;
    const int64_t nblk = tm_div_up(L, TM_BLK);
    const bool gate = MODE == 1 && a.z != nullptr;
    const int npos = (int)(p_end - p_begin);

    // running pointers: per-lane (VGPR) pointers of the activations advanced by wave-uniform steps, and a uniform
    // pointer of the B|C row -- no 64-bit multiplies and no kernel-argument reloads inside the loops
    const int64_t t0 = rev ? L - 1 - p_begin : p_begin;
    const int64_t su = tstep * a.ld_u, sd = tstep * a.ld_delta, sz = gate ? tstep * a.ld_z : 0, so = tstep * a.ld_out,
                  sbc = tstep * a.ld_bc;
    const T* up = (const T*)a.u + (row0 + t0) * a.ld_u + cc;       // next position to FETCH
    const T* dp = (const T*)a.delta + (row0 + t0) * a.ld_delta + cc;
    const T* zp = gate ? (const T*)a.z + (row0 + t0) * a.ld_z + cc : nullptr;
    T* op = MODE == 1 ? (T*)a.out + (row0 + t0) * a.ld_out + cc : nullptr;  // next position to STORE
    const float* bcp = a.BC + (row0 + t0) * a.ld_bc;                          // next position to COMPUTE
    float* stp = (MODE == 1 && a.state) ? a.state + ((sb * nblk + p_begin / TM_BLK) * N) * E + cc : nullptr;

    // register-block software pipeline: block j+1 is in flight while block j is computed
    TmRaw<T, CPL> ur[TM_PB], dr[TM_PB], zr[TM_PB];
    auto fetch = [&](int pb) {  // positions pb .. pb+TM_PB-1 of the chunk; the tail re-reads the last valid row
#pragma unroll
        for (int i = 0; i < TM_PB; ++i) {
            ur[i] = *(const TmRaw<T, CPL>*)up;
            dr[i] = *(const TmRaw<T, CPL>*)dp;
            if (gate) zr[i] = *(const TmRaw<T, CPL>*)zp;
            if (pb + i + 1 < npos) {
                up += su, dp += sd;
                if (gate) zp += sz;
            }
        }
    };
    fetch(0);
    for (int pb = 0; pb < npos; pb += TM_PB) {
        TmRaw<T, CPL> uc[TM_PB], dc[TM_PB], zc[TM_PB];
#pragma unroll
        for (int i = 0; i < TM_PB; ++i) {
            uc[i] = ur[i], dc[i] = dr[i];
            if (MODE == 1) zc[i] = zr[i];
        }
        if (pb + TM_PB < npos) fetch(pb + TM_PB);
        if (MODE == 1 && stp && (pb % TM_BLK) == 0) {  // state entering this block (p_begin is a multiple of TM_BLK)
            if (act) {
#pragma unroll
                for (int n = 0; n < NS; ++n)
                    if (n < N) {
#pragma unroll
                        for (int c = 0; c < CPL; ++c) stp[(int64_t)n * E + c] = h[c][n];
                    }
            }
            stp += (int64_t)N * E;
        }
        const bool whole = pb + TM_PB <= npos;  // uniform
#pragma unroll
        for (int i = 0; i < TM_PB; ++i) {
            if (!whole && pb + i >= npos) break;
            TM_FENCE();
            tm_cptr bc = TM_CPTR(bcp);
            float Bv[NS], Cv[NS];
#pragma unroll
            for (int n = 0; n < NS; ++n) {
                Bv[n] = (FULL || n < N) ? bc[n] : 0.f;
                if (MODE == 1) Cv[n] = (FULL || n < N) ? bc[N + n] : 0.f;
            }
            bcp += sbc;
            float uu[CPL], dd[CPL], y[CPL];
            tm_unpack<T, CPL>(uc[i], uu);
            tm_unpack<T, CPL>(dc[i], dd);
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const float dt = cad_softplus(dd[c] + bias[c]);
                const float dtu = dt * uu[c];
                sum_dt[c] += dt;
                float acc = Dv[c] * uu[c];
#pragma unroll
                for (int n = 0; n < NS; ++n) {
                    const float av = cad_exp2(dt * A2[c][n]);
                    h[c][n] = av * h[c][n] + dtu * Bv[n];
                    if (MODE == 1) acc += h[c][n] * Cv[n];
                }
                y[c] = acc;
            }
            if (MODE == 1) {
                if (gate) {
                    float zz[CPL];
                    tm_unpack<T, CPL>(zc[i], zz);
#pragma unroll
                    for (int c = 0; c < CPL; ++c) y[c] *= zz[c] * cad_sigmoid(zz[c]);
                }
                if (act) *(TmRaw<T, CPL>*)op = tm_pack<T, CPL>(y);
                op += so;
            }
        }
    }
    if (MODE == 0 && act) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            sdt[c] = sum_dt[c];
#pragma unroll
            for (int n = 0; n < NS; ++n)
                if (n < N) agg[(int64_t)n * E + c] = h[c][n];
        }
    }
}

// In place: agg[k] = S_k (end state of chunk k from zero)  ->  agg[k] = state at the start of chunk k.
// One thread per (row, state, channel); the loads do not depend on the recurrence, so they pipeline.
__global__ __launch_bounds__(TM_THREADS) void scan_tm_combine_kernel(TmFwdSets sets) {
    const cad_scan_tm_args& a = sets.s[blockIdx.z];
    const int E = a.E, N = a.N;
    const int64_t L = a.L, SB = a.SB;
    const int64_t nchunks = tm_div_up(L, TM_TC);
    const int64_t idx = (int64_t)blockIdx.x * TM_THREADS + threadIdx.x;  // over SB * N * E
    if (idx >= SB * N * E) return;
    const int c = (int)(idx % E);
    const int n = (int)((idx / E) % N);
    const int64_t sb = idx / ((int64_t)E * N);
    const float A2 = a.A[(int64_t)c * N + n] * CAD_LOG2E;
    float* agg = a.scratch + (sb * nchunks * N + n) * E + c;
    const float* sdt = a.scratch + SB * nchunks * N * E + sb * nchunks * E + c;
    float h = 0.f;
    for (int64_t k = 0; k < nchunks; ++k) {
        float* slot = agg + k * N * E;
        const bool more = k + 1 < nchunks;
        const float S = more ? *slot : 0.f;
        const float P = more ? cad_exp2(A2 * sdt[k * E]) : 0.f;
        *slot = h;
        h = P * h + S;
    }
}

template <typename T, int NS, int CPL, bool FULL>
int tm_fwd_launch(const TmFwdSets& ks, int nsets, void* stream) {
    const cad_scan_tm_args& a = ks.s[0];
    const int64_t nchunks = tm_div_up(a.L, TM_TC);
    const int EB = (int)tm_div_up(a.E, TM_THREADS * CPL);
    dim3 block(TM_THREADS);
    if (nchunks > 1) {
        dim3 g0((unsigned)(nchunks - 1), (unsigned)(a.SB * EB), (unsigned)nsets);
        CAD_LAUNCH((scan_tm_fwd_kernel<T, NS, CPL, 0, FULL>), g0, block, 0, stream, ks);
    }
    dim3 g1((unsigned)tm_div_up(a.SB * a.N * a.E, TM_THREADS), 1, (unsigned)nsets);
    CAD_LAUNCH(scan_tm_combine_kernel, g1, block, 0, stream, ks);
    dim3 g2((unsigned)nchunks, (unsigned)(a.SB * EB), (unsigned)nsets);
    CAD_LAUNCH((scan_tm_fwd_kernel<T, NS, CPL, 1, FULL>), g2, block, 0, stream, ks);
    return cad_after_launch();
}

}  // namespace

extern "C" int64_t cad_scan_tm_block_len(void) { return TM_BLK; }

extern "C" int64_t cad_scan_tm_state_floats(int E, int64_t SB, int64_t L, int N) {
    return SB * tm_div_up(L, TM_BLK) * N * E;
}

extern "C" int64_t cad_scan_tm_scratch_floats(int E, int64_t SB, int64_t L, int N) {
    return SB * tm_div_up(L, TM_TC) * (N + 1) * E;
}

extern "C" int cad_scan_tm_fwd_multi(const cad_scan_tm_args* sets, int nsets, void* stream) {
    CAD_CHECK_ARG(sets && nsets >= 1 && nsets <= TM_MAXSETS);
    TmFwdSets ks;
    bool two = true;  // two channels per lane: even E, even strides, 2-element aligned bases
    for (int i = 0; i < nsets; ++i) {
        const cad_scan_tm_args* a = &sets[i];
        CAD_CHECK_ARG(a->u && a->delta && a->A && a->BC && a->out && a->scratch);
        CAD_CHECK_ARG(a->E > 0 && a->SB > 0 && a->L > 0 && a->N > 0 && a->N <= 64);
        CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB);
        CAD_CHECK_ARG(a->ld_u >= a->E && a->ld_delta >= a->E && a->ld_out >= a->E && a->ld_bc >= 2 * a->N);
        CAD_CHECK_ARG(a->z == nullptr || a->ld_z >= a->E);
        CAD_CHECK_ARG(a->E == sets[0].E && a->SB == sets[0].SB && a->L == sets[0].L && a->N == sets[0].N &&
                      a->dtype == sets[0].dtype);
        CAD_CHECK_ARG(((uintptr_t)a->BC % 4) == 0);
        const size_t es = a->dtype == CAD_F32 ? 4 : 2;
        two = two && (a->E % 2 == 0) && (a->ld_u % 2 == 0) && (a->ld_delta % 2 == 0) && (a->ld_out % 2 == 0) &&
              (a->z == nullptr || a->ld_z % 2 == 0) &&
              (((uintptr_t)a->u | (uintptr_t)a->delta | (uintptr_t)a->z | (uintptr_t)a->out) % (2 * es)) == 0;
        ks.s[i] = *a;
    }
    for (int i = nsets; i < TM_MAXSETS; ++i) ks.s[i] = sets[0];
    const cad_scan_tm_args* a = &sets[0];
    CAD_CHECK_ARG(a->SB * tm_div_up(a->E, TM_THREADS) <= 65535);
    CadProfScope prof(0, stream);
    const bool f32 = a->dtype == CAD_F32;
    if (!f32 && a->dtype != CAD_BF16) return CAD_ERR_UNSUPPORTED;
    if (a->N == 16 && two)  // the production shape: d_state 16, even E
        return f32 ? tm_fwd_launch<float, 16, 2, true>(ks, nsets, stream) : tm_fwd_launch<bf16_t, 16, 2, true>(ks, nsets, stream);
    if (a->N <= 16)
        return f32 ? tm_fwd_launch<float, 16, 1, false>(ks, nsets, stream) : tm_fwd_launch<bf16_t, 16, 1, false>(ks, nsets, stream);
    return f32 ? tm_fwd_launch<float, 64, 1, false>(ks, nsets, stream) : tm_fwd_launch<bf16_t, 64, 1, false>(ks, nsets, stream);
}

extern "C" int cad_scan_tm_fwd(const cad_scan_tm_args* a, void* stream) { return cad_scan_tm_fwd_multi(a, 1, stream); }
