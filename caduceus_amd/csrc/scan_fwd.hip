// Selective SSM scan, forward (include/caduceus_hip.h, cad_scan_fwd / cad_scan_fwd_multi).
// See scan_common.h for the decomposition.
#include "scan_common.h"

#ifndef SC_PRE_WAIT
#define SC_PRE_WAIT 0   // 1: counted wait at the chunk start (leaves the most recent stores in flight)
#endif

namespace {

struct ScanFwdSets {
    cad_scan_args s[SC_MAXSETS];
};

#define SC_S SC_S_FWD
#define SC_W SC_W_FWD
#define SC_CHUNK (64 * SC_S)
// B/C tiles live in a ring of SC_RING_FWD LDS slots; a tile is staged SC_RING_FWD / 2 pairs ahead of its use and the
// workgroup meets at a barrier once per SC_RING_FWD / 2 pairs (2: double buffer, one barrier per pair).
#ifndef SC_OCC_FWD
#define SC_OCC_FWD SC_OCC
#endif
// LDS-DMA prefetch of the next chunk's u / delta / z vectors (bf16 production kernel; see sc_glds16): 32-byte vectors =
// two 16-byte planes per tensor; z is read at the END of a chunk, so it alternates between two slot pairs.
#ifndef SC_FWD_DMA
#define SC_FWD_DMA 1
#endif
#ifndef SC_RING_FWD
#define SC_RING_FWD (SC_FWD_DMA ? 4 : 8)   // the prefetch slots take 64 KB of the LDS the deeper ring used (-1.7 %)
#endif
#define PRE_SLOT (SC_W * 64 * 16)          // bytes per 16-byte plane (all waves)
#define PRE_BYTES (8 * PRE_SLOT)           // u0 u1 d0 d1 | z0 z1 (even chunks) | z0 z1 (odd chunks)

// SC_FWD_UNROLL_NP = 8: the production instantiation (bf16, vector path, d_state = 16) has its pair loop fully unrolled, as the backward
// (compile-time pair index: ring slot, barrier parity and staging cursor fold).  -DSC_FWD_UNROLL_NP=0: the run-time loop everywhere.
#ifndef SC_FWD_UNROLL_NP
#define SC_FWD_UNROLL_NP 8
#endif
// MO = map-only instantiation (cad_scan_args.map_only, pass 1 of an L-split scan): recurrence and wave scan only -- hT and
// sum_dt are the outputs; no C tile reads, no output phase, no gate, no stores of `out` / chunk states.
template <typename T, bool VEC, bool MO, int NPC = 0>
__global__ __launch_bounds__(64 * SC_W, SC_OCC_FWD) void scan_fwd_kernel(ScanFwdSets sets) {
    CAD_DYN_SMEM(float, smem);  // [SC_RING_FWD slots][B,C][SC_TILE]
    constexpr int TILE = SC_TILE(SC_S), ROW = SC_ROW(SC_S);
    constexpr int RING = SC_RING_FWD, AHEAD = RING / 2;
    static_assert(RING >= 2 && (RING & (RING - 1)) == 0, "ring of 2^k tiles");
    constexpr int SLOTS = SC_CHUNK / SC_STATE_STEP;  // saved-state slots per forward chunk (1 or 2)
    const cad_scan_args& a = sets.s[blockIdx.z];
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    sc_static_priority(wave, SC_W);
    const int64_t sb = blockIdx.y;
    const int e_raw = blockIdx.x * SC_W + wave;
    const bool act = e_raw < a.E;
    const int e = act ? e_raw : a.E - 1;
    const int rev = sb < a.split ? a.rev_lo : a.rev_hi;
    const int64_t L = a.L, SB = a.SB;
    const int N = NPC ? 2 * NPC : a.N, NP = NPC ? NPC : (a.N + 1) >> 1;  // NPC: compile-time pair count (the launcher checks N)
    static_assert(NPC == 0 || (NPC % RING == 0 && NPC > AHEAD), "unrolled pair loop: the ring position restarts with every chunk");
    const int64_t row_off = ((int64_t)e * SB + sb) * L;
    const T* u_row = (const T*)a.u + row_off;
    const T* d_row = (const T*)a.delta + row_off;
    const T* z_row = (a.z && !MO) ? (const T*)a.z + row_off : nullptr;
    T* o_row = (T*)a.out + row_off;
    const T* Bm = (const T*)a.Bm;
    const T* Cm = (const T*)a.Cm;
    const float Dv = a.D ? a.D[e] : 0.f;
    const bool is_dt = a.delta_is_dt != 0;  // wave-uniform: delta already holds softplus(delta_raw + bias)
    const float bias = (a.delta_bias && !is_dt) ? a.delta_bias[e] : 0.f;
    const int64_t nchunks = (L + SC_CHUNK - 1) / SC_CHUNK;
    const int64_t nslots = (L + SC_STATE_STEP - 1) / SC_STATE_STEP;
    constexpr bool PREF = SC_FWD_DMA && VEC && SC_S * sizeof(T) == 32;
    char* pre = (char*)(smem + RING * 2 * TILE);  // behind the tile ring
    const uint32_t pre_lds = cad_uniform((int)(sc_lds_off(pre) + wave * (64 * 16)));
    // SC_NDMA (= 6) DMA operations per chunk: u, delta, z (u again when there is no gate) x two 16-byte planes
    auto prefetch_vectors = [&](int64_t cq) {
        const int64_t pq = cq * SC_CHUNK + (int64_t)lane * SC_S;
        const int64_t l0 = pq < L ? (rev ? (L - pq - SC_S) : pq) : 0;  // clamped: out-of-range segments are zeroed at use
        const uint32_t zs = pre_lds + (4 + 2 * (uint32_t)(cq & 1)) * PRE_SLOT;
        const T* zsrc = z_row ? z_row : u_row;
        sc_glds16(u_row + l0, pre_lds);
        sc_glds16(u_row + l0 + 8, pre_lds + PRE_SLOT);
        sc_glds16(d_row + l0, pre_lds + 2 * PRE_SLOT);
        sc_glds16(d_row + l0 + 8, pre_lds + 3 * PRE_SLOT);
        sc_glds16(zsrc + l0, zs);
        sc_glds16(zsrc + l0 + 8, zs + PRE_SLOT);
    };
    auto read_vector = [&](int plane0, int64_t p0, ScVec<T, SC_S>& out) {  // two ds_read_b128
        struct __attribute__((aligned(16))) P { u32x4 a, b; };
        static_assert(sizeof(P) == sizeof(ScVec<T, SC_S>) || !PREF, "two 16-byte planes per vector");
        const char* slot = pre + wave * (64 * 16) + lane * 16 + plane0 * PRE_SLOT;
        P v;
        v.a = *(const u32x4*)slot, v.b = *(const u32x4*)(slot + PRE_SLOT);
        const uint32_t m = p0 < L ? 0xFFFFFFFFu : 0u;  // bf16 zero = zero bits
        const u32x4 mm = {m, m, m, m};
        v.a &= mm, v.b &= mm;
        if constexpr (sizeof(P) == sizeof(ScVec<T, SC_S>)) out = __builtin_bit_cast(ScVec<T, SC_S>, v);
    };

    // software pipeline: the B/C tile of the NEXT (chunk, pair) and the u/delta/z vectors of the NEXT chunk are in
    // flight (registers) while the current pair is computed.
    StageRegs<T, SC_SV(SC_S)> st;
    StageCtx<T> sctx = sc_stage_ctx<T, SC_S>(Bm, Cm, SB, sb, L);
    if constexpr (VEC) sc_stage_seek<T, SC_S>(sctx, 0, L, rev);
    ScVec<T, SC_S> u_raw, d_raw, z_raw;
    const int ntiles = (int)nchunks * NP;
    int s_np = 0;        // staging cursor: (chunk base, pair) of the next tile to stage
    int64_t s_base = 0;
    if constexpr (PREF) {
        prefetch_vectors(0);
    } else {
        sc_load_raw<T, SC_S, VEC>(u_row, (int64_t)lane * SC_S, L, rev, u_raw);
        sc_load_raw<T, SC_S, VEC>(d_row, (int64_t)lane * SC_S, L, rev, d_raw);
    }
    // stage the tile of the cursor / move the cursor on (wave-uniform)
#define SC_FWD_STAGE()                                                                 \
    do {                                                                               \
        if constexpr (VEC)                                                             \
            sc_stage_issue<T, SC_S>(st, sctx, 2 * s_np, N);                            \
        else                                                                           \
            sc_stage_load<T, SC_S, false>(st, sctx, 2 * s_np, N, s_base, L, rev);        \
    } while (0)
#define SC_FWD_ADVANCE()                                                               \
    do {                                                                               \
        if (++s_np == NP) {                                                            \
            s_np = 0, s_base += SC_CHUNK;                                              \
            if constexpr (VEC) sc_stage_seek<T, SC_S>(sctx, s_base, L, rev);           \
        }                                                                              \
    } while (0)
    for (int g = 0; g < AHEAD; ++g) {
        SC_FWD_STAGE();
        sc_stage_store<T, SC_S, VEC>(st, smem + g * 2 * TILE, rev);
        SC_FWD_ADVANCE();
    }
    __syncthreads();

    f32x2 carry = f2(0.f);  // lane np holds the running state of pair np at the current chunk start
    if (a.h0 && lane < NP) {
        const float* hp = a.h0 + ((int64_t)e * SB + sb) * N + 2 * lane;
        carry = f2(hp[0], (2 * lane + 1 < N) ? hp[1] : 0.f);
    }
    float sdt = 0.f;  // this lane's share of sum(dt) over the row
    // lane np holds (A[2np], A[2np+1]) * log2(e): read once, broadcast per pair with v_readlane (no memory access and
    // therefore no s_waitcnt vmcnt(0) inside the pair loop, which would drain the tile prefetch)
    f32x2 Areg = f2(0.f);
    if (lane < NP) {
        const int n0 = 2 * lane;
        Areg = f2(a.A[e * N + n0] * CAD_LOG2E, (n0 + 1 < N) ? a.A[e * N + n0 + 1] * CAD_LOG2E : 0.f);
    }
    int tix = 0;            // tiles consumed so far: tile tix lives in LDS buffer tix & 1
    SC_TIME_DECL;
    for (int64_t c = 0; c < nchunks; ++c) {
        const int64_t base = c * SC_CHUNK;
        const int64_t p0 = base + (int64_t)lane * SC_S;
        SC_TIME(0);  // chunk epilogue of the previous chunk (gate, store)
        float du[SC_S], dt[SC_S], y[SC_S];
        f32x2 y2[SC_S];  // per item: the output's even-state / odd-state partial sums (one v_pk_fma per item and pair)
        f32x2 dd[SC_S];  // (dt, dt * u)
#ifndef SC_FWD_PREFETCH
        if constexpr (PREF) {
            sc_wait_loads<SC_PRE_WAIT ? 2 : 0>();  // this chunk's vectors were fetched into LDS one chunk ago
            read_vector(0, p0, u_raw);
            read_vector(2, p0, d_raw);
        } else if (c > 0) {
            sc_load_raw<T, SC_S, VEC>(u_row, p0, L, rev, u_raw);
            sc_load_raw<T, SC_S, VEC>(d_row, p0, L, rev, d_raw);
        }
#endif
        sc_by_dir(rev, [&](auto rtag) {  // (one scalar branch per chunk instead of a select + rotate per dword)
            sc_unpack_d<T, SC_S, decltype(rtag)::value != 0>(u_raw, du);
            sc_unpack_d<T, SC_S, decltype(rtag)::value != 0>(d_raw, dt);
        });
#ifdef SC_FWD_PREFETCH
        if (z_row) sc_load_raw<T, SC_S, VEC>(z_row, p0, L, rev, z_raw);
        if (c + 1 < nchunks) {
            sc_load_raw<T, SC_S, VEC>(u_row, p0 + SC_CHUNK, L, rev, u_raw);
            sc_load_raw<T, SC_S, VEC>(d_row, p0 + SC_CHUNK, L, rev, d_raw);
        }
#endif
        // softplus: a wave-uniform BRANCH around the whole loop when delta already is dt (cad_proj_wx evaluated it); per
        // item it is evaluated for every lane and masked afterwards (a select instead of a branch around the transcendentals;
        // on the vector path a lane's items are in or out of range together)
        if (!is_dt && !(SC_WHATIF & 1024)) {
#pragma unroll
            for (int i = 0; i < SC_S; ++i) dt[i] = cad_softplus(dt[i] + bias);
        }
        float csum = 0.f;  // sum of dt over the lane's items: the product of the lane's a_i is exp2(A2 * csum)
#pragma unroll
        for (int i = 0; i < SC_S; ++i) {
            const float sp = dt[i];
            const float dti = (VEC ? (p0 < L) : (p0 + i < L)) ? sp : 0.f;
            y2[i] = f2(Dv * du[i], 0.f);
            dd[i] = f2(dti, dti * du[i]);
            csum += dti;
        }
        if (a.sum_dt) sdt += csum;  // wave-uniform; only the sequence-parallel / segmented callers ask for it
        // running state at this chunk's start (first slot of the chunk); a mid-chunk state (second slot) is written per pair
        float* st_base = (a.chunk_state && !MO) ? a.chunk_state + (((int64_t)e * SB + sb) * nslots + SLOTS * c) * NP * 2 : nullptr;
        if (st_base && act && lane < NP) {
            st_base[lane * 2] = carry[0];
            st_base[lane * 2 + 1] = carry[1];
        }
        SC_TIME(1);  // chunk prologue: loads, unpack, softplus
        // (unrolled instantiation: tix = NPC * c + np with NPC a multiple of the ring size, so the slot, the barrier parity and the
        // staging cursor -- which runs AHEAD pairs in front, still inside this chunk at np = 0 -- are functions of np alone)
        if constexpr (NPC != 0) s_np = AHEAD;
#if SC_FWD_UNROLL_NP
#pragma unroll
#endif
        for (int np = 0; np < NP; ++np, ++tix) {
            const int buf = NPC ? (np & (RING - 1)) : (tix & (RING - 1));
            // prefetch the tile AHEAD pairs from now (this chunk's or the next one's)
            const bool more = tix + AHEAD < ntiles;
            if (more) SC_FWD_STAGE();
            // the next chunk's u / delta / z: DMA into LDS behind this pair-step's tile loads, a whole chunk ahead of use
            const bool dma_now = PREF && np == 0 && c + 1 < nchunks;
            if constexpr (PREF) {
                if (dma_now) prefetch_vectors(c + 1);
            }
            const float* tB = smem + buf * 2 * TILE + lane * ROW;
            const float* tC = tB + TILE;
            const f32x2 A2 = readlane2(Areg, np);
            // (i) serial scan over the lane's items: the lane's map is (prod a_i, acc_h); the a_i / b_i are kept for the second pass
            // (round 5: the true states are then ONE dependent v_pk_fma per item from the state entering the lane -- the earlier form
            // kept the cumulative maps (ha, hh) instead, one more v_pk_mul per item for the running product of the a_i)
            f32x2 acc_h = f2(0.f);
            f32x2 ha[SC_S], hh[SC_S];  // a_i, b_i
            const f32x2 acc_a = exp2_2(f2(csum) * A2);
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
                const f32x2 av = (SC_WHATIF & 128) ? splat_lo(dd[i]) * A2 : exp2_2(splat_lo(dd[i]) * A2);
                const f32x2 bv = splat_hi(dd[i]) * ((SC_WHATIF & 64) ? f2(__builtin_bit_cast(float, lane + i)) : ld2(tB + 2 * i));
                acc_h = av * acc_h + bv;
                ha[i] = av;
                hh[i] = bv;
            }
            SC_TIME(2);  // staging issue + exp + serial scan (B tile reads)
            // (ii) inclusive scan of the affine maps across lanes (DPP), the state entering the chunk folded into lane 0's map: PH is
            // the TRUE state leaving every lane, the map products are dead afterwards
            const f32x2 hin = readlane2(carry, np);
            f32x2 PH = acc_h;
            if (!(SC_WHATIF & 512)) wave_scan_fwd_carry(acc_a, PH, hin, lane);
            // (iii) state entering this lane's segment / carry out
            const f32x2 h0 = f2(dpp_wave_shr1(hin[0], PH[0]), dpp_wave_shr1(hin[1], PH[1]));
            // state entering lane 32 = state at logical position base + 512: the backward's half-chunk start
            if (SLOTS == 2 && st_base && act && base + SC_STATE_STEP < L) {
                const f32x2 mid = readlane2(h0, SC_STATE_STEP / SC_S);
                if (lane == 0) {
                    st_base[(NP + np) * 2] = mid[0];
                    st_base[(NP + np) * 2 + 1] = mid[1];
                }
            }
            const f32x2 newc = readlane2(PH, 63);
            if (lane == np) carry = newc;
            SC_TIME(3);  // wave scan + carry
            cad_sched_fence();  // do not hoist the C-tile reads above the wave scan (register pressure)
            if constexpr (!MO) {
            f32x2 h = h0;
#pragma unroll
            for (int i = 0; i < SC_S; i += 2) {  // two items per step (one 16-byte C read); the output FMA of item i sits between the
                                                 // dependent state updates of items i and i + 1
                const f32x2 hA = ha[i] * h + hh[i];
                const f32x4 c4 = (SC_WHATIF & 64) ? f32x4{ha[i][0], ha[i][1], hh[i][0], hh[i][1]} : *(const f32x4*)(tC + 2 * i);
                h = ha[i + 1] * hA + hh[i + 1];
                pk_fma_acc(y2[i], f2(c4[0], c4[1]), hA);
                pk_fma_acc(y2[i + 1], f2(c4[2], c4[3]), h);
            }
            }
            SC_TIME(4);  // output phase (C tile reads)
            if (more) {
                sc_stage_store<T, SC_S, VEC>(st, smem + ((NPC ? np + AHEAD : tix + AHEAD) & (RING - 1)) * 2 * TILE, rev, dma_now);
                SC_FWD_ADVANCE();
            }
            SC_TIME(5);  // staging store
            if ((((NPC ? np : tix) + 1) & (AHEAD - 1)) == 0 && !(SC_WHATIF & 2)) __syncthreads();
            SC_TIME(6);  // barrier
        }
        if constexpr (MO) continue;  // no output of a map-only pass
#pragma unroll
        for (int i = 0; i < SC_S; ++i) y[i] = y2[i][0] + y2[i][1];
        if (z_row) {
            float zz[SC_S];
#ifndef SC_FWD_PREFETCH
            if constexpr (PREF)
                read_vector(4 + 2 * (int)(c & 1), p0, z_raw);  // landed long ago: the chunk-start wait covered it
            else
                sc_load_raw<T, SC_S, VEC>(z_row, p0, L, rev, z_raw);
#endif
            sc_by_dir(rev, [&](auto rtag) { sc_unpack_d<T, SC_S, decltype(rtag)::value != 0>(z_raw, zz); });
#pragma unroll
            for (int i = 0; i < SC_S; ++i) y[i] *= (SC_WHATIF & 1024) ? zz[i] : zz[i] * cad_sigmoid(zz[i]);
        }
        if (act && !(SC_WHATIF & 2048))
            sc_by_dir(rev, [&](auto rtag) { sc_store_d<T, SC_S, VEC, decltype(rtag)::value != 0>(o_row, p0, L, y); });
    }
    if (a.hT && act && lane < NP) {
        float* hp = a.hT + ((int64_t)e * SB + sb) * N + 2 * lane;
        hp[0] = carry[0];
        if (2 * lane + 1 < N) hp[1] = carry[1];
    }
    if (a.sum_dt) {  // wave-uniform
        sdt = wave_sum_dpp(sdt);
        if (act && lane == 0) a.sum_dt[(int64_t)e * SB + sb] = sdt;
    }
#undef SC_FWD_STAGE
#undef SC_FWD_ADVANCE
}

}  // namespace

SC_TIME_EXPORT(cad_debug_timing_fwd)

static_assert(SC_CHUNK == SC_STATE_STEP || SC_CHUNK == 2 * SC_STATE_STEP, "forward chunk = one or two state slots");

extern "C" int64_t cad_scan_chunk_len(void) { return SC_CHUNK; }

extern "C" int64_t cad_scan_state_floats(int E, int64_t SB, int64_t L, int N) {
    const int64_t nslots = (L + SC_STATE_STEP - 1) / SC_STATE_STEP;
    return (int64_t)E * SB * (nslots + 1) * ((N + 1) / 2) * 2;
}

extern "C" int cad_scan_fwd_multi(const cad_scan_args* sets, int nsets, void* stream) {
    CAD_CHECK_ARG(sets && nsets >= 1 && nsets <= SC_MAXSETS);
    ScanFwdSets ks;
    for (int i = 0; i < nsets; ++i) {
        const cad_scan_args* a = &sets[i];
        CAD_CHECK_ARG(a->u && a->delta && a->A && a->Bm && a->Cm && (a->out || a->map_only));
        CAD_CHECK_ARG(!a->map_only || (a->hT && a->sum_dt));
        CAD_CHECK_ARG(a->map_only == sets[0].map_only);
        CAD_CHECK_ARG(a->E > 0 && a->SB > 0 && a->L > 0 && a->N > 0 && a->N <= SC_NMAX);
        CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB && a->SB <= 65535);
        CAD_CHECK_ARG(a->E == sets[0].E && a->SB == sets[0].SB && a->L == sets[0].L && a->N == sets[0].N &&
                      a->dtype == sets[0].dtype);
        ks.s[i] = *a;
    }
    for (int i = nsets; i < SC_MAXSETS; ++i) ks.s[i] = sets[0];
    const cad_scan_args* a = &sets[0];
    // fast path: whole lane segments + 16-byte aligned rows (any production shape); else the element-wise kernel
    bool vec = (a->L % SC_S) == 0;
    for (int i = 0; i < nsets; ++i)
        vec = vec && (((uintptr_t)sets[i].u | (uintptr_t)sets[i].delta | (uintptr_t)sets[i].z | (uintptr_t)sets[i].out |
                       (uintptr_t)sets[i].Bm | (uintptr_t)sets[i].Cm) % 16) == 0;
    CadProfScope prof(0, stream);
    dim3 grid((unsigned)((a->E + SC_W - 1) / SC_W), (unsigned)a->SB, (unsigned)nsets), block(64 * SC_W);
    const size_t shmem = (size_t)SC_RING_FWD * 2 * SC_TILE(SC_S) * sizeof(float) +
                         ((SC_FWD_DMA && vec && a->dtype == CAD_BF16 && SC_S == 16) ? PRE_BYTES : 0);
#define SC_FWD_LAUNCH(T, V)                                                                  \
    do {                                                                                     \
        if (SC_FWD_UNROLL_NP && !a->map_only && V && sizeof(T) == 2 && a->N == 2 * SC_FWD_UNROLL_NP) { \
            SC_BIG_LDS((scan_fwd_kernel<T, V, false, SC_FWD_UNROLL_NP>), shmem);             \
            CAD_LAUNCH((scan_fwd_kernel<T, V, false, SC_FWD_UNROLL_NP>), grid, block, shmem, stream, ks); \
        } else if (a->map_only) {                                                            \
            SC_BIG_LDS((scan_fwd_kernel<T, V, true>), shmem);                                \
            CAD_LAUNCH((scan_fwd_kernel<T, V, true>), grid, block, shmem, stream, ks);       \
        } else {                                                                             \
            SC_BIG_LDS((scan_fwd_kernel<T, V, false>), shmem);                               \
            CAD_LAUNCH((scan_fwd_kernel<T, V, false>), grid, block, shmem, stream, ks);      \
        }                                                                                    \
    } while (0)
    if (a->dtype == CAD_F32) {
        if (vec)
            SC_FWD_LAUNCH(float, true);
        else
            SC_FWD_LAUNCH(float, false);
    } else if (a->dtype == CAD_BF16) {
        if (vec)
            SC_FWD_LAUNCH(bf16_t, true);
        else
            SC_FWD_LAUNCH(bf16_t, false);
    } else {
        return CAD_ERR_UNSUPPORTED;
    }
#undef SC_FWD_LAUNCH
    return cad_after_launch();
}

extern "C" int cad_scan_fwd(const cad_scan_args* a, void* stream) { return cad_scan_fwd_multi(a, 1, stream); }
