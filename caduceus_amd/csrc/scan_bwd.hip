// Selective SSM scan, backward (include/caduceus_hip.h, cad_scan_bwd / cad_scan_bwd_multi).
// See scan_common.h for the decomposition.
//
// Per 512-position chunk (processed from the logical END of the row to its start, because the state gradient flows
// backwards) and per state pair:
//   1. recompute h over the chunk from the running state saved by the forward (serial + DPP wave scan);
//   2. reverse scan of  G_i = a_i * (c_i + G_{i+1}),  c_i = C_i * dy_i   (G_i = gradient flowing into h_{i-1});
//      g_i = c_i + G_{i+1} is dL/dh_i;
//   3. per item:  d(dt) += g*h_{i-1}*a*A + u*<g,B>,  dA += g*h_{i-1}*a*dt,  du += dt*<g,B>,
//                 dB_i = g*dt*u,  dC_i = dy*h_i.
// dB / dC must be summed over all E channels.  Measured on MI355X (profiles/r01_scan_v2_pmc_summary.txt and the
// microbenchmarks next to it): a global fp32 atomic costs one un-coalesced 64-byte HBM write per LANE, and an LDS
// ds_add_f32 retires only ~1 lane every 3 cycles (~200 cycles per wave instruction) -- so NO atomics of either kind are
// used.  Each of the SC_W_BWD waves (channels) of a workgroup writes its dB/dC contributions with plain ds_write_b64
// into its own region of a double-buffered LDS slab; after the (single) barrier of the pair the 512 threads sum the
// SC_W_BWD regions and store the workgroup's partial sums with coalesced 16-byte stores to slot blockIdx.x of a
// (E / SC_W_BWD)-deep partial buffer; cad_reduce_partials folds the slots (and converts to the activation dtype) in a
// second, purely streaming pass.
#include "scan_common.h"

#ifndef SC_PRE_WAIT
#define SC_PRE_WAIT 0   // 1: counted wait at the chunk start (leaves the most recent stores in flight)
#endif

namespace {

struct ScanBwdSets {
    cad_scan_bwd_args s[SC_MAXSETS];
};

#define SC_S SC_S_BWD
#define SC_W SC_W_BWD
#define SC_CHUNK (64 * SC_S)
#define SC_DY(i) (((i) & 1) ? splat_hi(dy2[(i) >> 1]) : splat_lo(dy2[(i) >> 1]))  // dy of item i on both halves
#define ACC_ISTR (64 * 2 + 2)           // floats between items: 128 + 2 pad, so that the flush's two half-chunks
                                        // (items i and i + 4 of neighbouring lanes) fall into different LDS banks
#define ACC_TILE (SC_S * ACC_ISTR)      // floats per (wave, tensor) region, layout [item i][lane j][state s]
#define ACC_BUF (SC_W * 2 * ACC_TILE)  // floats per buffer: [wave][dB,dC][ACC_TILE]
// bf16 kernels (8-wave workgroups) keep the slab as packed bf16x2 dwords -- one dword = both states of one position.
// Region of one (channel = wave, tensor): [q = item-pair block 0/1][lane 64][4 dwords = items 4q .. 4q+3], i.e. a lane's
// four items of block q form one 16-byte PIECE.  The contributions are rounded to bf16 before the 8-channel sum; the
// partial slots they are summed into are bf16 anyway (same error order).
// The sum over the 8 channels runs on the MATRIX core: for a tile of 16 pieces, lane (g, j) feeds piece j of channel g
// (then g + 4) as the B operand of v_mfma_f32_16x16x32_bf16, the A operand is a constant 0/1 selection matrix
// A[i][8g + t] = (t == pi(i & 7)), so  D[i][j] = sum over the channels of element pi(i) of piece j  in fp32.  pi orders
// the rows as (state 0: items 0..3, state 1: items 0..3), which leaves every D lane with four consecutive positions of
// one state row.  One wave owns the two tiles (q = 0, 1) of (tensor, 16-lane block): 4 ds_read_b128 + 4 MFMA + 4
// v_cvt_pk + one 16-byte store per pair-step instead of 8 ds_read_b64 + ~64 VALU + 2 stores per thread.
#ifndef SC_SLAB_PACKED
#define SC_SLAB_PACKED 1
#endif
#define PK_Q (64 * 4)                   // dwords per item-pair block
#define PK_TILE (2 * PK_Q)              // dwords per (wave, tensor) region (2 KB: a multiple of the 256-byte bank row, so
                                        // the ds_read_b128 of the four channel groups are conflict-free)
#define PK_BUF (SC_W * 2 * PK_TILE)     // dwords per buffer
// LDS-DMA prefetch of the next chunk's item vectors (bf16 production kernel): 6 vectors x SC_W waves x 64 lanes x 16 bytes
#ifndef SC_BWD_SLAB_SWZ
#define SC_BWD_SLAB_SWZ 0  // 1: exchange the item pairs of a slab piece on lanes 8..15 of every 16 (no bank conflicts on the slab
                           // writes; measured 1.5 % SLOWER: the 8 selects after the MFMA cost more than the conflicts, which hide)
#endif
#ifndef SC_BWD_PREFETCH
#define SC_BWD_PREFETCH 1
#endif
#define PRE_SLOT (SC_W * 64 * 16)       // bytes per vector slot (all waves)
#define PRE_BYTES (SC_NDMA * PRE_SLOT)
#ifndef SC_SLAB_BUFS
#define SC_SLAB_BUFS 2                  // 2: one barrier per pair; 1: half the LDS (two workgroups per CU), two barriers
#endif

static_assert(SC_CHUNK == SC_STATE_STEP, "backward chunk = one saved-state slot");
// Only S = 8 items per lane is a supported backward.  S = 4 (a 128-VGPR build: four waves per SIMD) was asked for as an experiment
//  and is refused here instead of left as an instantiation that faults on the device: (i) at one sequence per GPU
// a BiMamba layer has E x 2 strands x 2 sets = 2048 channel-row waves = two per SIMD, so a four-wave build only fills half the CUs
// (or needs 16-channel workgroups: 128 workgroups on 256 CUs); (ii) the two wave scans, the staging and the flush are per (lane
// segment, pair), so halving the segment doubles ~40 % of the instructions per element (the forward built that way measured -15 %,
// profiles/r03_occupancy_experiments.txt); (iii) the packed slab / MFMA flush / 16-byte LDS-DMA vectors are laid out for 8 items.
static_assert(SC_S_BWD == 8, "the backward scan is written for 8 items per lane (see the comment above)");
// SC_BWD_UNROLL_NP = 8: the production instantiation (bf16, vector path, d_state = 16) has its pair loop fully unrolled -- the pair
// index is a compile-time constant, so lane selections become immediates and the tile-buffer parity and the `more` / staging
// conditions fold: 3.93 -> 3.80 ms per two-set launch in a same-box A/B (profiles/r04_scan_bwd_floor_and_unroll.txt; 244 VGPRs, no
// scratch).  Every other shape runs the generic instantiation (NPC = 0: run-time pair count).  -DSC_BWD_UNROLL_NP=0: round-3 kernel.
#ifndef SC_BWD_UNROLL_NP
#define SC_BWD_UNROLL_NP 8
#endif
#ifndef SC_BWD_FLUSH_HALF
#define SC_BWD_FLUSH_HALF 1   // which waves run the MFMA flush of a pair-step's dB / dC slab: 1 = the staging waves 0-3 (both tensors of their
                              // lane block), 0 = all eight (one tile pair each), 2 = waves 4-7
#endif
#ifndef SC_BWD_LEAN
#define SC_BWD_LEAN 1   // 0: the round-4 prologue / epilogue / per-pair-step dA wave sum for every launch (A/B switch)
#endif
static_assert(SC_W == 4 || SC_W == 8, "staging needs >= 256 threads; the flush mapping is written for 256 / 512");

// The dB / dC partial-slot stores of the production (packed, vector) kernel: write-through (sc1), which is what lets a fold kernel on
// another stream read them while this launch still runs -- and they are never read again by THIS kernel, so nothing is lost when the
// line leaves the XCD's L2.  -DSC_BWD_SLOT_WT=0: plain stores (A/B switch; the concurrent fold then must not be used).
#ifndef SC_BWD_SLOT_WT
#define SC_BWD_SLOT_WT 1
#endif
__device__ __forceinline__ void sc_slot_store16(void* p, u32x4 v) {
#if SC_BWD_SLOT_WT
    cad_store16_wt(p, v);
#else
    *(u32x4*)p = v;
#endif
}

__device__ __forceinline__ f32x2 wave_sum2(f32x2 v) { return f2(wave_sum_dpp(v[0]), wave_sum_dpp(v[1])); }

// sigmoid(raw delta) recovered from dt = softplus(raw delta): 1 - exp(-dt).  Results that are rounded to bf16 take the two-term series
// dt (1 - dt / 2) below 2^-9 (relative error dt^2 / 6 < 7e-7 there; above it 1 - exp(-dt) carries at most 2^-24 / 2^-9 = 3e-5) -- 5 VALU
// + 1 transcendental per item instead of the fp32 form's 10 + 1 (cad_sigmoid_from_softplus: four-term series below 1 / 16).
template <typename T>
__device__ __forceinline__ float sc_sigmoid_from_dt(float dt) {
    if constexpr (sizeof(T) == 2) {
        const float small = dt * (1.0f - 0.5f * dt);
        const float big = 1.0f - cad_exp(-dt);
        return dt < 0.001953125f ? small : big;
    } else {
        return cad_sigmoid_from_softplus(dt);
    }
}

// CO = carry-only instantiation (cad_scan_bwd_args.carry_only, pass 1 of an L-split backward): the reverse recurrence of the
// state gradient alone -- exp, C * dy, one chain per item and state, the reverse wave scan -- and dh0 as its only output.  A
// separate instantiation, so the full kernel carries none of its branches (measured: +5 % when they were run-time branches).
// ISDT = every set's delta already holds dt (cad_scan_bwd_args.delta_is_dt, the production path: softplus in the dt_proj epilogue) as a
// COMPILE-TIME fact of the unrolled production instantiation.  Together with the LDS-DMA prefetch it selects the LEAN chunk prologue /
// epilogue (round 5): a lane outside the row is silenced by dt = 0 and dy = 0 alone (every contribution vanishes by arithmetic, no
// per-item select), direction + widening of an item is one v_perm_b32, dt comes from the (dt, dt u) pairs (the raw delta vector is dead
// after the prologue), the accumulators start from the first pair's products instead of zeros, and dA is summed inside 8-lane groups per
// pair-step and across the groups once per kernel.  Static count of a 512-position chunk: profiles/r05_scan_isa.json (tools/isa_mix.py).
template <typename T, bool VEC, bool CO, int NPC = 0, bool ISDT = false>
__global__ __launch_bounds__(64 * SC_W, SC_OCC) void scan_bwd_kernel(ScanBwdSets sets) {
    CAD_DYN_SMEM(float, smem);  // [2 buffers][B,C][TILE] inputs, then [2 buffers][wave][dB,dC][ACC_TILE] contributions
    constexpr int TILE = SC_TILE(SC_S), ROW = SC_ROW(SC_S);
    const cad_scan_bwd_args& a = sets.s[blockIdx.z];
    float* acc = smem + 4 * TILE;
    constexpr bool PACKED = SC_SLAB_PACKED && sizeof(T) == 2 && (SC_W == 8 || SC_W == 4) && SC_SLAB_BUFS == 2;
    uint32_t* accp = (uint32_t*)acc;
    static_assert(!PACKED || SC_S == 8, "packed slab: two 4-item blocks per lane");
    // item vectors of the next chunk travel global -> LDS by DMA one chunk ahead (16-byte vectors: bf16, 8 items)
    constexpr bool PREF = SC_BWD_PREFETCH && PACKED && VEC && SC_S * sizeof(T) == 16;
    // [vector 0..5][wave][lane][16 bytes], behind the two slab buffers (a carry-only pass has no slab: 68 KB of LDS in all, so two
    // of its workgroups -- 92 VGPRs -- share a CU)
    char* pre = CO ? (char*)acc : (char*)(accp + 2 * PK_BUF);
    constexpr bool LEAN = PREF && ISDT && !CO && NPC != 0;
    static_assert(!ISDT || LEAN, "ISDT is the production instantiation's switch");
    const int lane = threadIdx.x & 63;
    // selection matrix of the MFMA flush (see PK_TILE): row i = lane & 15 picks element pi(i & 7) of every piece
    u32x4 selA = {0u, 0u, 0u, 0u};
    if constexpr (PACKED) {
        const uint32_t one = 0x3F80u << (16 * ((lane >> 2) & 1));  // bf16 1.0 in the low / high half
        selA[0] = (lane & 3) == 0 ? one : 0u, selA[1] = (lane & 3) == 1 ? one : 0u;
        selA[2] = (lane & 3) == 2 ? one : 0u, selA[3] = (lane & 3) == 3 ? one : 0u;
    }
    const int wave = cad_uniform(threadIdx.x >> 6);
    sc_static_priority(wave, SC_W);
    const int64_t sb = blockIdx.y;
    const int e_raw = blockIdx.x * SC_W + wave;
    const bool act = e_raw < a.E;
    const int e = act ? e_raw : a.E - 1;
    const int rev = sb < a.split ? a.rev_lo : a.rev_hi;
    const int64_t L = a.L, SB = a.SB;
    const int N = NPC ? 2 * NPC : a.N, NP = NPC ? NPC : (a.N + 1) >> 1;  // NPC: compile-time pair count (the launcher checks N)
    const int64_t row_off = ((int64_t)e * SB + sb) * L;
    const T* u_row = (const T*)a.u + row_off;
    const T* d_row = (const T*)a.delta + row_off;
    const T* z_row = a.z ? (const T*)a.z + row_off : nullptr;
    const T* g_row = (const T*)a.dout + row_off;
    T* dz_row = (a.dz && !CO) ? (T*)a.dz + row_off : nullptr;
    const T* o_row = (a.out && dz_row) ? (const T*)a.out + row_off : nullptr;      // only the gate gradient needs it
    const T* o2_row = (a.out2 && dz_row) ? (const T*)a.out2 + row_off : nullptr;  // the other scan under the same gate
    T* du_row = (T*)a.du + row_off;
    T* dd_row = (T*)a.ddelta + row_off;
    const T* Bm = (const T*)a.Bm;
    const T* Cm = (const T*)a.Cm;
    const float Dv = a.D ? a.D[e] : 0.f;
    const bool is_dt = ISDT || a.delta_is_dt != 0;  // wave-uniform: delta already holds dt = softplus(delta_raw + bias)
    const ScDirSel dsel = sc_dir_sel(rev);
    const float bias = (a.delta_bias && !is_dt) ? a.delta_bias[e] : 0.f;
    const int64_t nchunks = (L + SC_CHUNK - 1) / SC_CHUNK;
    const float keep = act ? 1.f : 0.f;  // padding waves (E % SC_W != 0) contribute nothing
    const int64_t part_stride = (int64_t)N * SB * L;
    T* dBg = (T*)a.dB + (int64_t)blockIdx.x * part_stride;  // this workgroup's partial-sum slot
    T* dCg = (T*)a.dC + (int64_t)blockIdx.x * part_stride;
    // concurrent fold (cad_fold_partials_stream on another stream, include/caduceus_hip.h): one arrival per (row, chunk) and workgroup,
    // published once every slot store of the chunk has left the CU
    int* pub = (!CO && a.fold_counters) ? a.fold_counters + sb * nchunks : nullptr;
    // ... and one "this workgroup is resident" arrival (the int behind the SB x nchunks chunk counters): cad_fold_partials_stream's
    // gate launch lets the fold kernel onto the CUs only once the scan's workgroups are placed (see fold_gate_kernel)
    // ... and marks the CU it runs on (CAD_CU_KEYS ints behind that): a fold workgroup that finds no scan workgroup on its CU while scan
    // workgroups are still waiting for a CU (another kernel -- an RCCL all-reduce -- held theirs when the launch started) leaves at once
    // instead of blocking that CU for the whole scan
    if (pub && threadIdx.x == 0) {
        cad_counter_add_agent(a.fold_counters + SB * nchunks, 1);
        cad_counter_add_agent(a.fold_counters + SB * nchunks + 1 + cad_cu_key(), 1);
    }

    StageRegs<T, SC_SV(SC_S)> st;
    StageCtx<T> sctx = sc_stage_ctx<T, SC_S>(Bm, Cm, SB, sb, L);
    ScVec<T, SC_S> u_raw, d_raw, g_raw, z_raw, o_raw, o2_raw;  // u_raw / d_raw stay in registers until the chunk's epilogue
    {
        const int64_t base = (nchunks - 1) * SC_CHUNK;
        if constexpr (VEC) {
            sc_stage_seek<T, SC_S>(sctx, base, L, rev);
            sc_stage_issue<T, SC_S>(st, sctx, 0, N);
        } else {
            sc_stage_load<T, SC_S, false>(st, sctx, 0, N, base, L, rev);
        }
        if constexpr (!PREF) {
            sc_load_raw<T, SC_S, VEC>(u_row, base + (int64_t)lane * SC_S, L, rev, u_raw);
            sc_load_raw<T, SC_S, VEC>(d_row, base + (int64_t)lane * SC_S, L, rev, d_raw);
            sc_load_raw<T, SC_S, VEC>(g_row, base + (int64_t)lane * SC_S, L, rev, g_raw);
            if (z_row) sc_load_raw<T, SC_S, VEC>(z_row, base + (int64_t)lane * SC_S, L, rev, z_raw);
            if (o_row) sc_load_raw<T, SC_S, VEC>(o_row, base + (int64_t)lane * SC_S, L, rev, o_raw);
            if (o2_row) sc_load_raw<T, SC_S, VEC>(o2_row, base + (int64_t)lane * SC_S, L, rev, o2_raw);
        }
        sc_stage_store<T, SC_S, VEC>(st, smem, rev);
    }
    __syncthreads();
    // LDS-DMA of the item vectors of the chunk whose lane segment starts at logical position pq.  Always SC_NDMA
    // operations (absent tensors re-fetch u into their slot) so that the counted wait of the staging path is exact.
    const uint32_t pre_lds = cad_uniform((int)(sc_lds_off(pre) + wave * (64 * 16)));
    auto prefetch_vectors = [&](int64_t pq) {
        const int64_t l0 = pq < L ? (rev ? (L - pq - SC_S) : pq) : 0;  // clamped: out-of-range segments are zeroed at use
        sc_glds16(u_row + l0, pre_lds);
        sc_glds16(d_row + l0, pre_lds + PRE_SLOT);
        sc_glds16(g_row + l0, pre_lds + 2 * PRE_SLOT);
        sc_glds16((z_row ? z_row : u_row) + l0, pre_lds + 3 * PRE_SLOT);
        sc_glds16((o_row ? o_row : u_row) + l0, pre_lds + 4 * PRE_SLOT);
        sc_glds16((o2_row ? o2_row : u_row) + l0, pre_lds + 5 * PRE_SLOT);
    };
    if constexpr (PREF) prefetch_vectors((nchunks - 1) * SC_CHUNK + (int64_t)lane * SC_S);

    // lane np holds (A[2np], A[2np+1]); broadcast per pair with v_readlane (no memory access in the pair loop)
    f32x2 Areg = f2(0.f);
    if (lane < NP) {
        const int n0 = 2 * lane;
        Areg = f2(a.A[e * N + n0], (n0 + 1 < N) ? a.A[e * N + n0 + 1] : 0.f);
    }
    f32x2 carryG = f2(0.f);  // lane np: G flowing out of the later chunk into this one, for pair np
    if (a.dhT && act && lane < NP) {  // padding waves (E % SC_W != 0) must not inject a state gradient
        const float* gp = a.dhT + ((int64_t)e * SB + sb) * N + 2 * lane;
        carryG = f2(gp[0], (2 * lane + 1 < N) ? gp[1] : 0.f);
    }
    f32x2 hin_next = f2(0.f);
    if (!CO && lane < NP) {
        const float* stp = a.chunk_state + ((((int64_t)e * SB + sb) * nchunks + nchunks - 1) * NP + lane) * 2;
        hin_next = f2(stp[0], stp[1]);
    }
    f32x2 dAacc = f2(0.f);   // lane np: dA of pair np
    float dDacc = 0.f, dbacc = 0.f;
    int tix = 0;

    SC_TIME_DECL;
    for (int64_t c = nchunks - 1; c >= 0; --c) {
        const int64_t base = c * SC_CHUNK;
        const int64_t p0 = base + (int64_t)lane * SC_S;
        SC_TIME(0);  // flush tail of the previous pair-step / loop overhead
        float ddt[SC_S], gBs[SC_S];   // sum over the states of g * h_{i-1} * a * A  and of  <g, B>
        f32x2 dd[SC_S];               // (dt, dt * u) per item
        f32x2 dy2[SC_S / 2];          // dy of items (2q, 2q + 1)
        float sum_dt = 0.f;    // sum of dt over the lane's items: prod_i a_i = exp2(A2 * sum_dt)
        // (publishing: every wave's slot stores of the previous chunk must have completed before the barrier of pair-step 0 -- the
        // prefetching instantiations wait for vmcnt(0) here anyway, for their item vectors)
        if constexpr (!PREF || SC_PRE_WAIT) {
            if (pub) cad_wait_vmcnt<0>();
        }
        if constexpr (LEAN) {
            // this chunk's vectors were fetched into LDS one chunk ago.  A lane's segment lies inside the row or outside it as a whole;
            // outside, dt = 0 and dy = 0 make every contribution of the lane vanish by arithmetic (a = 1, b = 0, g = G, dB = dC = 0,
            // dA term = t * 0, d(delta) = (..) * (1 - exp(-0))), so ONLY these two vectors are masked -- the clamped prefetch address
            // delivers finite data of the row start for the others -- and no per-item select is left.  dy = 0 also silences a padding
            // wave (E % SC_W != 0: its state-gradient carry starts from 0, so g stays 0).
            sc_wait_loads<SC_PRE_WAIT ? 3 : 0>();
            const char* slot = pre + wave * (64 * 16) + lane * 16;
            typedef ScVec<T, SC_S> V;
            const uint32_t inm = p0 < L ? 0xffffffffu : 0u, livem = (p0 < L && act) ? 0xffffffffu : 0u;
            auto rd = [&](int k, uint32_t m) {
                u32x4 v = *(const u32x4*)(slot + k * PRE_SLOT);
                v[0] &= m, v[1] &= m, v[2] &= m, v[3] &= m;
                return __builtin_bit_cast(V, v);
            };
            auto rd_raw = [&](int k) { return __builtin_bit_cast(V, *(const u32x4*)(slot + k * PRE_SLOT)); };
            u_raw = rd_raw(0);
            d_raw = rd(1, inm);
            g_raw = rd(2, livem);
            if (z_row) z_raw = rd_raw(3);
            if (o_row) o_raw = rd_raw(4);
            if (o2_row) o2_raw = rd_raw(5);
        } else if constexpr (PREF) {
            // this chunk's vectors were fetched into LDS one chunk ago
            sc_wait_loads<SC_PRE_WAIT ? 3 : 0>();
            const char* slot = pre + wave * (64 * 16) + lane * 16;
            const bool in = p0 < L;
            typedef ScVec<T, SC_S> V;
            // (unconditional LDS read, then a mask: `in ? *slot : zero` is compiled to a select between the slot's address
            // and a zero vector kept in scratch, read back through flat loads -- 0.4 GB of scratch stores per launch)
            const uint32_t keepm = in ? 0xffffffffu : 0u;
            auto rd = [&](int k) {
                u32x4 v = *(const u32x4*)(slot + k * PRE_SLOT);
                v[0] &= keepm, v[1] &= keepm, v[2] &= keepm, v[3] &= keepm;
                return __builtin_bit_cast(V, v);
            };
            u_raw = rd(0);
            d_raw = rd(1);
            g_raw = rd(2);
            if (z_row) z_raw = rd(3);
            if (o_row) o_raw = rd(4);
            if (o2_row) o2_raw = rd(5);
        }
        if constexpr (LEAN) {
            float uu[SC_S], dt[SC_S], dy[SC_S];
            sc_unpack_p<T, SC_S>(u_raw, rev, dsel, uu);
            sc_unpack_p<T, SC_S>(d_raw, rev, dsel, dt);
            sc_unpack_p<T, SC_S>(g_raw, rev, dsel, dy);
            if (z_row) {
                // gate: dy <- dout * silu(z), and the gate gradient right away (it needs nothing from the scan):
                // out = y z sigmoid(z)  =>  y sigmoid(z) = out / z ;  dz = dout * (out / z) * (1 + z - z sigmoid(z))
                float zz[SC_S];
                sc_unpack_p<T, SC_S>(z_raw, rev, dsel, zz);
                if (a.gate_fix_list) {
                    // out / z cannot recover y where the gate is exactly 0 (out == 0 there): remember the chunk, the fix-up launch
                    // (cad_scan_bwd_gate_fix) recomputes y for it and adds dout * y / 2 to dz.  min |z| over the lane's items == 0.
                    float zmin = __builtin_fabsf(zz[0]);
#pragma unroll
                    for (int i = 1; i < SC_S; ++i) zmin = __builtin_fminf(zmin, __builtin_fabsf(zz[i]));
                    if (cad_wave_any(zmin == 0.f && p0 < L && act) && lane == 0) {
                        const int slot = atomicAdd(a.gate_fix_count, 1);
                        a.gate_fix_list[slot] = (int64_t)e | ((int64_t)sb << 20) | ((int64_t)c << 40);
                    }
                }
                if (o_row) {  // wave-uniform: this set writes the gate gradient (of both scans sharing the gate when out2 is given)
                    float oo[SC_S], dzv[SC_S];
                    sc_unpack_p<T, SC_S>(o_raw, rev, dsel, oo);
                    if (o2_row) {
                        float o2[SC_S];
                        sc_unpack_p<T, SC_S>(o2_raw, rev, dsel, o2);
#pragma unroll
                        for (int i = 0; i < SC_S; ++i) oo[i] += o2[i];
                    }
#pragma unroll
                    for (int i = 0; i < SC_S; ++i) {
                        const float sg = cad_sigmoid(zz[i]);
                        // (legacy product: 0 where out == 0, i.e. also at z == 0 where 1 / z is inf -- the value the fix-up launch adds to)
                        const float ys = cad_mul_legacy(oo[i], cad_rcp(zz[i]));
                        const float silu = zz[i] * sg;
                        dzv[i] = dy[i] * ys * ((1.f + zz[i]) - silu);
                        dy[i] *= silu;
                    }
                    if (act && !(SC_WHATIF & 2048))
                        sc_by_dir(rev, [&](auto rtag) { sc_store_d<T, SC_S, VEC, decltype(rtag)::value != 0>(dz_row, p0, L, dzv); });
                } else {
#pragma unroll
                    for (int i = 0; i < SC_S; ++i) dy[i] *= zz[i] * cad_sigmoid(zz[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
                dDacc += dy[i] * uu[i];
                dd[i] = f2(dt[i], dt[i] * uu[i]);
                sum_dt += dt[i];
                dy2[i >> 1][i & 1] = dy[i];
            }
        } else {
            float uu[SC_S], dt[SC_S], dy[SC_S];
            sc_unpack<T, SC_S>(u_raw, rev, uu);
            sc_unpack<T, SC_S>(d_raw, rev, dt);
            sc_unpack<T, SC_S>(g_raw, rev, dy);
            if (z_row) {
                // gate: dy <- dout * silu(z), and the gate gradient right away (it needs nothing from the scan):
                // out = y * z * sigmoid(z)  =>  y * sigmoid(z) = out / z ;  dz = dout * y * sigmoid(z) * (1 + z (1 - sigmoid(z)))
                float zz[SC_S], oo[SC_S], dzv[SC_S];
                sc_unpack<T, SC_S>(z_raw, rev, zz);
                if (!CO && a.gate_fix_list) {
                    // out / z cannot recover y where the gate is exactly 0 (out == 0 there): remember the chunk, the
                    // fix-up launch (cad_scan_bwd_gate_fix) recomputes y for it and adds dout * y / 2 to dz
                    int z0 = 0;  // (bitwise, not short-circuit: no branch per item)
#pragma unroll
                    for (int i = 0; i < SC_S; ++i) z0 |= (int)(zz[i] == 0.f) & (int)(p0 + i < L);
                    if (cad_wave_any(z0 != 0 && act) && lane == 0) {
                        const int slot = atomicAdd(a.gate_fix_count, 1);
                        a.gate_fix_list[slot] = (int64_t)e | ((int64_t)sb << 20) | ((int64_t)c << 40);
                    }
                }
                if (o_row) {  // wave-uniform: this set writes the gate gradient
                    sc_unpack<T, SC_S>(o_raw, rev, oo);
                    if (o2_row) {  // ... of both scans sharing the gate
                        float o2[SC_S];
                        sc_unpack<T, SC_S>(o2_raw, rev, o2);
#pragma unroll
                        for (int i = 0; i < SC_S; ++i) oo[i] += o2[i];
                    }
                }
                // (the wave-uniform `o_row` test is hoisted out of the item loop: inside it the compiler keeps one
                // branch per item and cannot schedule across them)
                if (o_row) {
#pragma unroll
                    for (int i = 0; i < SC_S; ++i) {
                        const float sg = cad_sigmoid(zz[i]);
                        const float yq = oo[i] * cad_rcp(zz[i]);
                        const float ys = (zz[i] == 0.f) ? 0.f : yq;
                        dzv[i] = dy[i] * ys * (1.f + zz[i] * (1.f - sg));
                        dy[i] *= zz[i] * sg;
                    }
                    if (act && !(SC_WHATIF & 2048))
                        sc_by_dir(rev, [&](auto rtag) { sc_store_d<T, SC_S, VEC, decltype(rtag)::value != 0>(dz_row, p0, L, dzv); });
                } else {
#pragma unroll
                    for (int i = 0; i < SC_S; ++i) dy[i] *= zz[i] * cad_sigmoid(zz[i]);
                }
            }
            // softplus: a wave-uniform BRANCH around the whole loop when delta already is dt; per item it is evaluated for
            // every lane and masked afterwards (a select, not a branch around the transcendental sequence); on the vector
            // path all items of a lane are in or out of range together
            if (!is_dt && !(SC_WHATIF & 1024)) {
#pragma unroll
                for (int i = 0; i < SC_S; ++i) dt[i] = cad_softplus(dt[i] + bias);
            }
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
                const bool ok = VEC ? (p0 < L) : (p0 + i < L);
                const float sp = dt[i];
                const float dti = ok ? sp : 0.f;
                const float dyi = ok ? dy[i] * keep : 0.f;
                ddt[i] = 0.f;
                gBs[i] = 0.f;
                dDacc += dyi * uu[i];
                dd[i] = f2(dti, dti * uu[i]);
                sum_dt += dti;
                dy2[i >> 1][i & 1] = dyi;
            }
        }
        // running states of all pairs at this chunk's start (saved by the forward): lane np holds pair np; the next
        // (earlier) chunk's states are fetched now and land while this chunk computes
        const f32x2 hin_reg = hin_next;
        if (!CO && lane < NP && c > 0) {
            const float* stp = a.chunk_state + ((((int64_t)e * SB + sb) * nchunks + c - 1) * NP + lane) * 2;
            hin_next = f2(stp[0], stp[1]);
        }
        // per-item outputs of this chunk (all states folded in): u / delta from the raw vectors loaded at the chunk's start.
        // With the LDS-DMA prefetch nothing overwrites those registers, so it runs once BEHIND the pair loop (inside the
        // loop the compiler if-converts it and evaluates its transcendentals in every pair-step).
        auto chunk_epilogue = [&]() {
            if constexpr (LEAN) {
                // dt is the low half of the (dt, dt u) pairs; u is widened again from its raw vector (4 registers across the pair loop
                // instead of 8); sigmoid(raw delta) = 1 - exp(-dt).  A lane outside the row has dt = 0, hence d(delta) = (..) * 0 and
                // du = 0 * <g, B> + 0 * D: no select
                float uu[SC_S], du[SC_S];
                sc_unpack_p<T, SC_S>(u_raw, rev, dsel, uu);
#pragma unroll
                for (int i = 0; i < SC_S; ++i) {
                    const float dti = dd[i][0];
                    const float sg = sc_sigmoid_from_dt<T>(dti);
                    du[i] = dti * gBs[i] + dy2[i >> 1][i & 1] * Dv;
                    ddt[i] = (ddt[i] + uu[i] * gBs[i]) * sg;
                    dbacc += ddt[i];
                }
                if (act && !(SC_WHATIF & 2048)) {
                    sc_by_dir(rev, [&](auto rtag) {
                        sc_store_d<T, SC_S, VEC, decltype(rtag)::value != 0>(du_row, p0, L, du);
                        sc_store_d<T, SC_S, VEC, decltype(rtag)::value != 0>(dd_row, p0, L, ddt);
                    });
                }
                return;
            }
            float uu[SC_S], dl[SC_S], du[SC_S];
            sc_by_dir(rev, [&](auto rtag) {
                sc_unpack_d<T, SC_S, decltype(rtag)::value != 0>(u_raw, uu);
                sc_unpack_d<T, SC_S, decltype(rtag)::value != 0>(d_raw, dl);
            });
            float sgv[SC_S];
            if (is_dt) {  // wave-uniform
#pragma unroll
                for (int i = 0; i < SC_S; ++i) sgv[i] = cad_sigmoid_from_softplus(dl[i]);
            } else {
#pragma unroll
                for (int i = 0; i < SC_S; ++i) {
                    const float xraw = dl[i] + bias;
                    sgv[i] = xraw > 20.f ? 1.f : cad_sigmoid(xraw);
                }
            }
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
                const float sg = sgv[i];
                const float dyi = dy2[i >> 1][i & 1];
                const bool ok = VEC ? (p0 < L) : (p0 + i < L);
                du[i] = dd[i][0] * gBs[i] + dyi * Dv;
                ddt[i] = ok ? (ddt[i] + uu[i] * gBs[i]) * sg : 0.f;
                dbacc += ddt[i];
            }
            if (act && !(SC_WHATIF & 2048)) {
                sc_by_dir(rev, [&](auto rtag) {
                    sc_store_d<T, SC_S, VEC, decltype(rtag)::value != 0>(du_row, p0, L, du);
                    sc_store_d<T, SC_S, VEC, decltype(rtag)::value != 0>(dd_row, p0, L, ddt);
                });
            }
        };
        SC_TIME(1);  // chunk prologue: unpack, gate, softplus
#if SC_BWD_UNROLL_NP
#pragma unroll
#endif
        for (int np = 0; np < NP; ++np, ++tix) {
            const int buf = (NPC && (NPC % 2) == 0) ? (np & 1) : (tix & 1);  // an even pair count: the parity restarts with every chunk
            const bool more = (np + 1 < NP) || (c > 0);
            if (more) {
                const int nn = (np + 1 < NP) ? 2 * (np + 1) : 0;
                const int64_t nb = (np + 1 < NP) ? base : base - SC_CHUNK;
                if constexpr (VEC) {
                    if (nn == 0) sc_stage_seek<T, SC_S>(sctx, nb, L, rev);  // wave-uniform: once per chunk
                    sc_stage_issue<T, SC_S>(st, sctx, nn, N);
                } else {
                    sc_stage_load<T, SC_S, false>(st, sctx, nn, N, nb, L, rev);
                }
            }
            // the next (earlier) chunk's item vectors: by DMA into LDS, issued behind this pair-step's tile loads (the
            // counted wait at the staging store lets them fly) and a whole chunk ahead of their use
            const bool dma_now = PREF && np == 0 && c > 0;
            if constexpr (PREF) {
                if (dma_now) prefetch_vectors(p0 - SC_CHUNK);
            }
            const float* tB = smem + buf * 2 * TILE + lane * ROW;
            const float* tC = tB + TILE;
            float* aB = acc + (SC_SLAB_BUFS == 2 ? buf : 0) * ACC_BUF + wave * 2 * ACC_TILE + lane * 2;  // (i, s) at aB[i * ACC_ISTR + s]: 8-byte stride
            float* aC = aB + ACC_TILE;
            const int n0 = 2 * np;
            const f32x2 Av = readlane2(Areg, np);
            const f32x2 A2 = Av * f2(CAD_LOG2E);
            const f32x2 hin = readlane2(hin_reg, np);
            // B and C of this pair are each needed twice (recompute / gradient step, reverse scan / gradient step): read
            // them from the LDS tile once, up front
            f32x2 Cv[SC_S], Bw[SC_S];
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
                if (SC_WHATIF & 64)
                    Bw[i] = f2(__builtin_bit_cast(float, lane + i)), Cv[i] = f2(__builtin_bit_cast(float, lane - i));
                else
                    Bw[i] = ld2(tB + 2 * i), Cv[i] = ld2(tC + 2 * i);
            }
            SC_TIME(2);  // staging issue + B/C tile reads
            if constexpr (CO) {
                // reverse recurrence of the state gradient alone: G_i = a_i (C_i dy_i + G_{i+1}), lane-local, then across the wave
                const f32x2 acc_a = exp2_2(f2(sum_dt) * A2);
                f32x2 RGc = f2(0.f);
#pragma unroll
                for (int r = SC_S - 1; r >= 0; --r) RGc = exp2_2(splat_lo(dd[r]) * A2) * (Cv[r] * SC_DY(r) + RGc);
                const f32x2 ginc = readlane2(carryG, np);
                wave_scan_rev_carry(acc_a, RGc, ginc, lane);
                const f32x2 newcc = readlane2(RGc, 0);
                if (lane == np) carryG = newcc;
            } else {
            // 1. forward recompute: serial totals, wave scan, then the true h_i
            f32x2 av[SC_S], hs[SC_S];
            f32x2 acc_h = f2(0.f);
            const f32x2 acc_a = exp2_2(f2(sum_dt) * A2);  // product of the lane's a_i
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
                av[i] = (SC_WHATIF & 128) ? splat_lo(dd[i]) * A2 : exp2_2(splat_lo(dd[i]) * A2);
                hs[i] = splat_hi(dd[i]) * Bw[i];  // b_i
                acc_h = av[i] * acc_h + hs[i];
            }
            SC_TIME(3);  // exp + serial scan
            f32x2 PH = acc_h;
            if (!(SC_WHATIF & 512)) wave_scan_fwd_carry(acc_a, PH, hin, lane);  // PH: the true state leaving each lane
            const f32x2 h0 = f2(dpp_wave_shr1(hin[0], PH[0]), dpp_wave_shr1(hin[1], PH[1]));  // state entering this lane's segment
            // the true h_i (forward chain) and 2. the reverse scan of G (backward chain), interleaved: two independent
            // serial v_pk_fma chains, each step of one fills the wait state the other needs between dependent packed ops
            SC_TIME(4);  // forward wave scan
            f32x2 RG = f2(0.f);
            {
                f32x2 h = h0;
#pragma unroll
                for (int i = 0; i < SC_S; ++i) {
                    const int r = SC_S - 1 - i;
                    h = av[i] * h + hs[i];
                    RG = av[r] * (Cv[r] * SC_DY(r) + RG);
                    hs[i] = h;  // h_i
                }
            }
            SC_TIME(5);  // true h + lane-local reverse scan
            f32x2 QG = RG;
            const f32x2 gin = readlane2(carryG, np);
            if (!(SC_WHATIF & 256)) wave_scan_rev_carry(acc_a, QG, gin, lane);  // QG: the true G flowing out of each lane
            f32x2 G = f2(dpp_wave_shl1(gin[0], QG[0]), dpp_wave_shl1(gin[1], QG[1]));  // G_{i+1} for this lane's last item
            const f32x2 newc = readlane2(QG, 0);
            if (lane == np) carryG = newc;
            SC_TIME(6);  // reverse wave scan + carry
            // 3. gradients
            f32x2 dAp = f2(0.f);
            uint32_t pkB = 0, pkC = 0;
#pragma unroll
            for (int i = SC_S - 1; i >= 0; --i) {
                const f32x2 Bv = Bw[i];
                const f32x2 g = Cv[i] * SC_DY(i) + G;
                G = av[i] * g;
                const f32x2 hprev = (i > 0) ? hs[i > 0 ? i - 1 : 0] : h0;
                const f32x2 t = G * hprev;  // g * a_i * h_{i-1}
                if (LEAN && np == 0) {  // (compile-time after unrolling: the first pair's products start the chunk's sums)
                    ddt[i] = dot2_first(t, Av);
                    gBs[i] = dot2_first(g, Bv);
                } else {
                    ddt[i] = dot2_acc(ddt[i], t, Av);  // scalar accumulators: the packed form (one v_pk_fma each) needs 16 more
                    gBs[i] = dot2_acc(gBs[i], g, Bv);  // VGPRs, spills, and the spill traffic reaches HBM (+0.75 GB per launch)
                }
                dAp = dAp + t * splat_lo(dd[i]);
                const f32x2 dBv = g * splat_hi(dd[i]);
                const f32x2 dCv = hs[i] * SC_DY(i);
                if constexpr (PACKED) {
                    const uint32_t pB = cad_pack_bf16x2(dBv[0], dBv[1]), pC = cad_pack_bf16x2(dCv[0], dCv[1]);
                    if (i & 1) {
                        pkB = pB, pkC = pC;  // the odd item waits for its even partner: one 8-byte store per item pair
                    } else {
                        // (8-byte stores at a 16-byte lane stride: 2-way bank conflicts, 31 % of the LDS cycles -- see
                        // SC_BWD_SLAB_SWZ for the swizzle that removes them and why it is off)
                        uint32_t* qB = accp + buf * PK_BUF + wave * 2 * PK_TILE + (i >> 2) * PK_Q + lane * 4 +
                                       ((((i >> 1) ^ ((lane >> 3) & SC_BWD_SLAB_SWZ)) & 1) * 2);
                        if (!(SC_WHATIF & 32)) {
                            *(u32x2*)qB = u32x2{pB, pkB};
                            *(u32x2*)(qB + PK_TILE) = u32x2{pC, pkC};
                        } else {
                            asm volatile("" ::"v"(pB), "v"(pkB), "v"(pC), "v"(pkC));
                        }
                    }
                } else {
                    *(f32x2*)(aB + i * ACC_ISTR) = dBv;  // ds_write_b64, conflict-free
                    *(f32x2*)(aC + i * ACC_ISTR) = dCv;
                }
            }
            SC_TIME(7);  // gradient loop + slab writes
            if constexpr (LEAN) {
                // sum inside the 8-lane groups now (3 DPP steps), across the 8 groups once per kernel: lane 8 g + np collects pair np
                add_on_lanes_mod8(dAacc, group8_sum2_dpp(dAp), np);
            } else {
                dAp = wave_sum2_dpp(dAp);
                if (lane == np) dAacc = dAacc + dAp * f2(keep);
            }
            if constexpr (!PREF && !CO) {
                // (register prefetch: the next chunk's vectors overwrite u_raw / d_raw behind this pair's barrier)
                if (np == NP - 1) chunk_epilogue();
            }
            }  // !CO
            SC_TIME(8);  // dA wave sum (+ chunk epilogue on the last pair)
            if (more) sc_stage_store<T, SC_S, VEC>(st, smem + (buf ^ 1) * 2 * TILE, rev, dma_now);
            SC_TIME(9);  // staging store (waits for the tile loads)
            if (!(SC_WHATIF & 2)) __syncthreads();  // every channel has written its dB/dC; the prefetched B/C tile is visible
            SC_TIME(10);  // barrier wait
            // the chunk processed before this one is complete in memory: all waves drained their stores at this chunk's start and have
            // passed a barrier since (one lane of a NON-staging wave: its only counted wait is the one at the chunk start)
            if (np == 0 && pub && c + 1 < nchunks && threadIdx.x == 64 * (SC_W - 1)) cad_counter_add_agent(pub + c + 1, 1);
            if (!PREF && np == NP - 1 && c > 0) {
                // the item vectors of the next (earlier) chunk: issued now, they land behind this pair's flush and the
                // chunk epilogue instead of stalling the next chunk's start
                const int64_t pn = p0 - SC_CHUNK;
                sc_load_raw<T, SC_S, VEC>(u_row, pn, L, rev, u_raw);
                sc_load_raw<T, SC_S, VEC>(d_row, pn, L, rev, d_raw);
                sc_load_raw<T, SC_S, VEC>(g_row, pn, L, rev, g_raw);
                if (z_row) sc_load_raw<T, SC_S, VEC>(z_row, pn, L, rev, z_raw);
                if (o_row) sc_load_raw<T, SC_S, VEC>(o_row, pn, L, rev, o_raw);
                if (o2_row) sc_load_raw<T, SC_S, VEC>(o2_row, pn, L, rev, o2_raw);
            }
            // sum the SC_W regions and flush: thread t owns one tensor (dB / dC), one state of the pair and FT
            // consecutive positions, stored 4 at a time (8/16-byte stores).  With two slab buffers the next pair writes
            // the other buffer, so one barrier per pair suffices.
            if constexpr (CO) continue;  // no slab, no flush
            if constexpr (PACKED) {
                if (!(SC_WHATIF & 32)) {
                // one (tensor, 16-lane block) tile pair per call: tensor `ten`, lanes 16 jb .. + 15, summed over the 8 channels on the matrix core
                auto flush_tile = [&](const int ten, const int jb) {
                const int g = lane >> 4, jl = lane & 15;
                const uint32_t* src = accp + buf * PK_BUF + ten * PK_TILE + (jb * 16 + jl) * 4 + g * (2 * PK_TILE);
                f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int hf = 0; hf < SC_W / 4; ++hf) {  // (K = 32 = four channels x 8 elements per matrix product)
                    const u32x4 b0 = *(const u32x4*)(src + hf * (8 * PK_TILE));
                    const u32x4 b1 = *(const u32x4*)(src + hf * (8 * PK_TILE) + PK_Q);
                    d0 = cad_mfma_16x16x32_bf16(selA, b0, d0);
                    d1 = cad_mfma_16x16x32_bf16(selA, b1, d1);
                }
                // lanes 0..31: g = state of the pair; d0 = items 0..3, d1 = items 4..7 of lane (jb, jl); pieces of lanes with
                // bit 3 set were stored with their item pairs exchanged (bank swizzle of the slab writes)
                if (SC_BWD_SLAB_SWZ && (jl & 8)) {
                    d0 = f32x4{d0[2], d0[3], d0[0], d0[1]};
                    d1 = f32x4{d1[2], d1[3], d1[0], d1[1]};
                }
                if (!(SC_WHATIF & 1) && g < 2 && n0 + g < N) {
                    const int64_t p = base + (int64_t)(jb * 16 + jl) * SC_S;
                    // row (state n0 + g) of this workgroup's slot: scalar base + one per-lane select (g is 0 or 1 here)
                    T* grow = (ten ? dCg : dBg) + ((int64_t)n0 * SB + sb) * L + (g ? SB * L : (int64_t)0);
                    if (VEC) {
                        if (p < L) {
                            u32x4 o;
                            if (rev) {
                                o[0] = cad_pack_bf16x2_safe(d1[3], d1[2]), o[1] = cad_pack_bf16x2_safe(d1[1], d1[0]);
                                o[2] = cad_pack_bf16x2_safe(d0[3], d0[2]), o[3] = cad_pack_bf16x2_safe(d0[1], d0[0]);
                                sc_slot_store16(grow + (L - p - SC_S), o);
                            } else {
                                o[0] = cad_pack_bf16x2_safe(d0[0], d0[1]), o[1] = cad_pack_bf16x2_safe(d0[2], d0[3]);
                                o[2] = cad_pack_bf16x2_safe(d1[0], d1[1]), o[3] = cad_pack_bf16x2_safe(d1[2], d1[3]);
                                sc_slot_store16(grow + p, o);
                            }
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (p + q < L) grow[cad_phys(p + q, L, rev)] = from_f32<T>(d0[q]);
                            if (p + 4 + q < L) grow[cad_phys(p + 4 + q, L, rev)] = from_f32<T>(d1[q]);
                        }
                    }
                }
                };  // flush_tile
#if SC_BWD_FLUSH_HALF == 0
                flush_tile(wave >> 2, wave & 3);  // every wave one tile pair (rounds 2-4)
#else
                // the STAGING waves 0-3 flush both tensors of their lane block, waves 4-7 go straight on (SC_BWD_FLUSH_HALF = 2: the other
                // way round).  Waves 4-7 are the ones that reach the pair-step's barrier last (each shares its SIMD with a staging wave that
                // wins the VALU arbitration by age, then waits for its tile loads): the flush -- LDS reads, four matrix products, a store,
                // hardly any VALU work -- costs the staging waves waiting time they have and takes ~30 instructions + a latency chain out of
                // the critical waves' pair-step.  Same-box: 3.391 vs 3.458 ms (-1.9 %); the other half: +1.2 % (profiles/r05_ab_flush_placement.txt).
                // (NOT unrolled: with both tiles' address arithmetic hoisted out of the chunk loop the kernel spills -- 256 VGPRs + 156 bytes
                // of scratch, +24 % -- while this form needs 223.)
                // (SC_W = 4, two workgroups per CU: every wave stages and flushes -- its own lane block)
                if (SC_W == 4 || (wave < SC_W / 2) == (SC_BWD_FLUSH_HALF == 1)) {  // wave-uniform
#pragma unroll 1
                    for (int ften = 0; ften < 2; ++ften) flush_tile(ften, wave & 3);
                }
#endif
                }  // SC_WHATIF & 32
            } else {
                constexpr int QT = 64 * SC_W / 4;     // threads per (tensor, state)
                constexpr int FT = SC_CHUNK / QT;     // positions per thread (2, 4 or 8)
                constexpr int FV = FT < 4 ? FT : 4;   // ... stored FV at a time
                const int t = threadIdx.x;
                const int ten = t / (2 * QT), s = (t / QT) & 1, idx = t % QT;
                const float* tile = acc + (SC_SLAB_BUFS == 2 ? buf : 0) * ACC_BUF + ten * ACC_TILE;
                T* grow = (ten ? dCg : dBg) + ((int64_t)(n0 + s) * SB + sb) * L;
#pragma unroll
                for (int h4 = 0; h4 < FT; h4 += FV) {
                    const int tok = idx * FT + h4;
                    const int j = tok / SC_S, i0 = tok % SC_S;
                    float v[FV];
#pragma unroll
                    for (int q = 0; q < FV; ++q) {
                        const float* src = tile + (i0 + q) * ACC_ISTR + j * 2 + s;
                        float sum = 0.f;
#pragma unroll
                        for (int w = 0; w < SC_W; ++w) sum += src[w * 2 * ACC_TILE];
                        v[q] = sum;
                    }
                    if (n0 + s < N) {
                        const int64_t p = base + tok;
                        if (VEC) {
                            if (p < L) {
                                if (rev) {
                                    float o[FV];
#pragma unroll
                                    for (int q = 0; q < FV; ++q) o[q] = v[FV - 1 - q];
                                    cad_cvt_store<T, FV>(grow + (L - p - FV), o);
                                } else {
                                    cad_cvt_store<T, FV>(grow + p, v);
                                }
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < FV; ++q)
                                if (p + q < L) grow[cad_phys(p + q, L, rev)] = from_f32<T>(v[q]);
                        }
                    }
                }
            }
            if (SC_SLAB_BUFS == 1) __syncthreads();  // the slab is rewritten by the next pair
            SC_TIME(11);  // next chunk's loads issued + flush
        }
        if constexpr (PREF && !CO) chunk_epilogue();
    }
    if (pub) {  // the last chunk (c = 0)
        cad_wait_vmcnt<0>();
        __syncthreads();
        if (threadIdx.x == 64 * (SC_W - 1)) cad_counter_add_agent(pub, 1);
    }
    if (a.dh0 && act && lane < NP) {  // gradient w.r.t. the state entering the row
        float* gp = a.dh0 + ((int64_t)e * SB + sb) * N + 2 * lane;
        gp[0] = carryG[0];
        if (2 * lane + 1 < N) gp[1] = carryG[1];
    }
    if constexpr (CO) return;  // nothing else is an output of a carry-only pass
    if constexpr (LEAN) {  // lanes 8 g + np hold group g's share of pair np: fold the groups, lane np ends up with the wave's sum
#pragma unroll
        for (int m = 8; m <= 32; m <<= 1) dAacc = dAacc + f2(__shfl_xor(dAacc[0], m), __shfl_xor(dAacc[1], m));
    }
    // per-channel parameter gradients (E x N, E: a few device-scope atomics per wave, once per kernel)
    if (act && lane < NP) {
        const int n0 = 2 * lane;
        atomicAdd(a.dA + e * N + n0, dAacc[0]);
        if (n0 + 1 < N) atomicAdd(a.dA + e * N + n0 + 1, dAacc[1]);
    }
    dDacc = wave_sum1(dDacc);
    dbacc = wave_sum1(dbacc);
    if (act && lane == 0) {
        if (a.dD) atomicAdd(a.dD + e, dDacc);
        if (a.ddelta_bias) atomicAdd(a.ddelta_bias + e, dbacc);
    }
}

// Gate gradient where z == 0 exactly (see the worklist in scan_bwd_kernel): one WORKGROUP per recorded (channel, row, chunk)
// recomputes the UNGATED y of that chunk from the saved chunk state (serial scan + DPP wave scan, as the forward) and
// adds  dout * y * sigmoid(0) = dout * y / 2  to dz at the positions whose gate is 0.  Rare path: element-wise loads.
// (Rare per element, not per launch: 134 M gate values per configs[2] launch and P(an fp32 sum rounds to 0) ~ 4e-9 make about one
// entry per two launches.)  The four waves share the state pairs -- the path is a chain of dependent global round trips, one per
// state pair -- and their partial y are added in a fixed order through LDS.
#define GF_WAVES 4
template <typename T>
__global__ __launch_bounds__(64 * GF_WAVES) void scan_gate_fix_kernel(cad_scan_bwd_args a) {
    __shared__ float ypart[GF_WAVES - 1][SC_S][64];
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const int count = *a.gate_fix_count;
    const int64_t L = a.L, SB = a.SB;
    const int N = a.N, NP = (N + 1) >> 1;
    const int64_t nchunks = (L + SC_CHUNK - 1) / SC_CHUNK;
    for (int idx = blockIdx.x; idx < count; idx += gridDim.x) {  // (count is uniform: every wave makes the same trips)
        const int64_t ent = a.gate_fix_list[idx];
        const int e = (int)(ent & 0xFFFFF);
        const int64_t sb = (ent >> 20) & 0xFFFFF, c = ent >> 40;
        const int rev = sb < a.split ? a.rev_lo : a.rev_hi;
        const int64_t row_off = ((int64_t)e * SB + sb) * L;
        const T* u_row = (const T*)a.u + row_off;
        const T* d_row = (const T*)a.delta + row_off;
        const T* z_row = (const T*)a.z + row_off;
        const T* g_row = (const T*)a.dout + row_off;
        T* dz_row = (T*)a.gate_fix_dz + row_off;
        const float Dv = a.D ? a.D[e] : 0.f;
        const bool is_dt = a.delta_is_dt != 0;
        const float bias = (a.delta_bias && !is_dt) ? a.delta_bias[e] : 0.f;
        const int64_t p0 = c * SC_CHUNK + (int64_t)lane * SC_S;
        float y[SC_S];
        f32x2 dd[SC_S];
#pragma unroll
        for (int i = 0; i < SC_S; ++i) {
            const bool ok = p0 + i < L;
            const int64_t l = ok ? cad_phys(p0 + i, L, rev) : 0;
            const float ui = ok ? to_f32(u_row[l]) : 0.f;
            const float draw = to_f32(d_row[l]) + bias;
            const float dti = ok ? (is_dt ? draw : cad_softplus(draw)) : 0.f;
            y[i] = wave == 0 ? Dv * ui : 0.f;
            dd[i] = f2(dti, dti * ui);
        }
        for (int np = wave; np < NP; np += GF_WAVES) {
            const int n0 = 2 * np;
            const bool two = n0 + 1 < N;
            const f32x2 A2 = f2(a.A[e * N + n0], two ? a.A[e * N + n0 + 1] : 0.f) * f2(CAD_LOG2E);
            const float* stp = a.chunk_state + ((((int64_t)e * SB + sb) * nchunks + c) * NP + np) * 2;
            const f32x2 hin = f2(stp[0], stp[1]);
            const T* Brow = (const T*)a.Bm + ((int64_t)n0 * SB + sb) * L;
            const T* Crow = (const T*)a.Cm + ((int64_t)n0 * SB + sb) * L;
            f32x2 av[SC_S], bv[SC_S], Cv[SC_S];
            f32x2 acc_a = f2(1.f), acc_h = f2(0.f);
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
                const bool ok = p0 + i < L;
                const int64_t l = ok ? cad_phys(p0 + i, L, rev) : 0;
                const f32x2 Bv = ok ? f2(to_f32(Brow[l]), two ? to_f32(Brow[SB * L + l]) : 0.f) : f2(0.f);
                Cv[i] = ok ? f2(to_f32(Crow[l]), two ? to_f32(Crow[SB * L + l]) : 0.f) : f2(0.f);
                av[i] = exp2_2(splat_lo(dd[i]) * A2);
                bv[i] = splat_hi(dd[i]) * Bv;
                acc_h = av[i] * acc_h + bv[i];
                acc_a = acc_a * av[i];
            }
            f32x2 PA = acc_a, PH = acc_h;
            wave_scan_fwd(PA, PH);
            const f32x2 ea = f2(dpp_wave_shr1(1.f, PA[0]), dpp_wave_shr1(1.f, PA[1]));
            const f32x2 eh = f2(dpp_wave_shr1(0.f, PH[0]), dpp_wave_shr1(0.f, PH[1]));
            f32x2 h = ea * hin + eh;
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
                h = av[i] * h + bv[i];
                y[i] += dot2(Cv[i], h);
            }
        }
        if (wave > 0) {
#pragma unroll
            for (int i = 0; i < SC_S; ++i) ypart[wave - 1][i][lane] = y[i];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < SC_S; ++i) {
#pragma unroll
                for (int w = 0; w < GF_WAVES - 1; ++w) y[i] += ypart[w][i][lane];
                if (p0 + i < L) {
                    const int64_t l = cad_phys(p0 + i, L, rev);
                    if (to_f32(z_row[l]) == 0.f)
                        dz_row[l] = from_f32<T>(to_f32(dz_row[l]) + 0.5f * to_f32(g_row[l]) * y[i]);
                }
            }
        }
        __syncthreads();  // ypart is reused by the next entry
    }
}

// dst[i] = sum_k src[k * n + i]  (partial slots in T, fp32 accumulation); 4 elements per thread
template <typename T>
__device__ __forceinline__ void ld4p(const T* p, float* o);
template <>
__device__ __forceinline__ void ld4p<float>(const float* p, float* o) {
    struct __attribute__((aligned(16))) V { float f[4]; };
    const V t = *(const V*)p;
    o[0] = t.f[0], o[1] = t.f[1], o[2] = t.f[2], o[3] = t.f[3];
}
template <>
__device__ __forceinline__ void ld4p<bf16_t>(const bf16_t* p, float* o) {
    struct __attribute__((aligned(8))) V { uint32_t w[2]; };
    const V t = *(const V*)p;
    o[0] = cad_bits2f(t.w[0] << 16), o[1] = cad_bits2f(t.w[0] & 0xffff0000u);
    o[2] = cad_bits2f(t.w[1] << 16), o[3] = cad_bits2f(t.w[1] & 0xffff0000u);
}

template <typename T>
__global__ void reduce_partials_kernel(const T* src, int nparts, int64_t n, T* dst, int vec) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        // (the library's one summation order, include/caduceus_hip.h: groups of CAD_FOLD_GROUP consecutive slots, then the group sums)
        if (vec && i + 4 <= n) {
            float s[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k0 = 0; k0 < nparts; k0 += CAD_FOLD_GROUP) {
                float g[4] = {0.f, 0.f, 0.f, 0.f};
                for (int k = k0; k < nparts && k < k0 + CAD_FOLD_GROUP; ++k) {
                    float x[4];
                    ld4p<T>(src + (int64_t)k * n + i, x);
                    g[0] += x[0], g[1] += x[1], g[2] += x[2], g[3] += x[3];
                }
                s[0] += g[0], s[1] += g[1], s[2] += g[2], s[3] += g[3];
            }
            cad_cvt_store<T, 4>(dst + i, s);
        } else {
            for (int64_t q = i; q < n && q < i + 4; ++q) {
                float acc = 0.f;
                for (int k0 = 0; k0 < nparts; k0 += CAD_FOLD_GROUP) {
                    float g = 0.f;
                    for (int k = k0; k < nparts && k < k0 + CAD_FOLD_GROUP; ++k) g += to_f32(src[(int64_t)k * n + q]);
                    acc += g;
                }
                dst[q] = from_f32<T>(acc);
            }
        }
    }
}

// several folds of the same depth and length in one launch (blockIdx.y = job): the dB and dC slots of both parameter sets of a layer
struct ReduceJobs {
    const void* src[CAD_REDUCE_MAX_JOBS];
    void* dst[CAD_REDUCE_MAX_JOBS];
};
template <typename T>
__global__ void reduce_partials_multi_kernel(ReduceJobs jobs, int nparts, int64_t n) {
    const T* src = (const T*)jobs.src[blockIdx.y];
    T* dst = (T*)jobs.dst[blockIdx.y];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {  // (n % 4 == 0, 16-byte aligned: checked)
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < nparts; k0 += CAD_FOLD_GROUP) {
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = k0; k < nparts && k < k0 + CAD_FOLD_GROUP; ++k) {
                float x[4];
                ld4p<T>(src + (int64_t)k * n + i, x);
                g[0] += x[0], g[1] += x[1], g[2] += x[2], g[3] += x[3];
            }
            s[0] += g[0], s[1] += g[1], s[2] += g[2], s[3] += g[3];
        }
        cad_cvt_store<T, 4>(dst + i, s);
    }
}

// ---- the dB / dC fold behind a RUNNING scan backward (cad_fold_partials_stream, include/caduceus_hip.h) -----------------------------
// One workgroup per (slice x of a chunk, row, parameter set), 256 threads.  A 512-position chunk of a row's dB / dC is 2 N rows x 512
// bf16; slice x is elements [x EPW, (x + 1) EPW) of it, EPW = 2 N 512 / n_partials (256 at configs[2]: half a row).  Per slot the
// slice is VPS = EPW / 8 16-byte vectors; thread t owns vector t % VPS of the CAD_FOLD_GROUP = 8 slots of group t / VPS (8 loads in
// flight per thread, 32 KB per workgroup), sums them in slot order, and the first VPS threads add the group sums in group order
// through LDS: the library's one summation order, bit-identical to cad_reduce_partials_multi.  Chunks are taken in the order the scan
// produces them (last logical chunk first); a right-to-left row's logical chunk c lies at physical positions L - (c + 1) 512.
#define FOLD_T 256
#define FOLD_CHUNK 512
struct FoldSets {
    cad_fold_args s[SC_MAXSETS];
};
// Placement gate.  The fold kernel must reach a CU AFTER the scan workgroup it shares that CU with: a 48-VGPR / 10 KB allocation that
// lands first -- or next to the waves of a third kernel that then leave (the carry pass of an L-split backward, an RCCL all-reduce) --
// sits in the MIDDLE of the register file / LDS and leaves no contiguous 2 x 232 VGPRs / 132 KB for the scan workgroup: measured, the
// full pass of an L-split backward did not start until the fold gave up 20 ms later (profiles/r06_ab_stream_fold.txt).  So one wave runs
// AHEAD of the fold kernel on its stream and returns only when the scan's workgroups have been placed: every scan workgroup adds 1 to
// counters[SB x nchunks] as it starts; the gate waits for the first arrival, then until the count has stopped rising for ~20 us (a whole
// grid is dispatched within a microsecond; a launch with more workgroups than CUs stalls at the resident ones) or the budget is spent.
__global__ __launch_bounds__(64) void fold_gate_kernel(FoldSets sets, int nsets, int want, uint64_t budget_ticks) {
    if (threadIdx.x != 0) return;
    const uint64_t t0 = cad_wall_clock();
    int last = -1;
    uint64_t t_change = t0;
    for (;;) {
        int n = 0;
        for (int i = 0; i < nsets; ++i) {
            const cad_fold_args& a = sets.s[i];
            n += cad_counter_load_agent(a.counters + a.SB * (a.L / FOLD_CHUNK));
        }
        const uint64_t now = cad_wall_clock();
        if (n >= want) return;
        if (n != last) last = n, t_change = now;
        if (n > 0 && now - t_change >= 2000) return;   // 20 us without a new workgroup: everything that fits is resident
        if (now - t0 >= budget_ticks) return;          // the scan is not running next to us: the fold kernel deals with that itself
        cad_poll_sleep();
    }
}

#define FOLD_MAX_ITEMS 256  // items (slice, row, set) one workgroup may be given
#ifndef FOLD_WAKE_DIV
#define FOLD_WAKE_DIV 4
#endif
__global__ __launch_bounds__(FOLD_T) void fold_stream_kernel(FoldSets sets, int nsets, int mode, uint64_t budget_ticks) {
    // One launch has AT MOST one workgroup per CU (the host passes the CU count as the grid limit): a second resident fold workgroup would
    // take the registers the next scan workgroup needs on that CU (2 x 232 + 2 x 48 > 512 VGPRs per SIMD) and starve the scan of launches
    // with more workgroups than CUs (configs[4]: measured +19 % per layer with one fold workgroup per item).  Items beyond the grid are
    // taken by the same workgroups, item = blockIdx.x + j gridDim.x -- in the order the scan's workgroups are dispatched, and never
    // blocking on one item while another has a chunk ready.
    __shared__ float part[FOLD_T / 2 * 8];
    __shared__ int nextc[FOLD_MAX_ITEMS];  // next chunk of item j (chunks are taken from the last logical one down); < 0: done
    __shared__ int pick_s[2];              // {item to fold now or -1, give up}
    const int t = threadIdx.x;
    const cad_fold_args& a0 = sets.s[0];
    const int G = a0.n_partials, N = a0.N;
    const int64_t L = a0.L, SB = a0.SB;
    const int64_t nchunks = L / FOLD_CHUNK;
    const int total = G * (int)SB * nsets;
    const int nitems = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int EPW = 2 * N * FOLD_CHUNK / G, VPS = EPW / 8, NG = G / CAD_FOLD_GROUP;  // NG x VPS = 16 N threads work (all 256 at d_state 16)
    const bool active = t < NG * VPS;
    const int v = t % VPS, grp = active ? t / VPS : 0;
    const int64_t part_stride = (int64_t)N * SB * L;
    auto item_set = [&](int j) { return (blockIdx.x + j * gridDim.x) / (G * (int)SB); };
    auto item_row = [&](int j) { return ((blockIdx.x + j * gridDim.x) / G) % (int)SB; };
    auto item_slice = [&](int j) { return (blockIdx.x + j * gridDim.x) % G; };
    auto abort_slot = [&](int j) -> int* {
        const cad_fold_args& a = sets.s[item_set(j)];
        return a.abort_from ? a.abort_from + (int64_t)item_row(j) * G + item_slice(j) : nullptr;
    };
    if (mode == CAD_FOLD_CONCURRENT) {
        // co-location check (see scan_bwd_kernel): not next to a scan workgroup AND scan workgroups still unplaced -> this workgroup is
        // in their way (its registers / LDS sit where theirs must go): hand everything to the cleanup launch and leave
        if (t == 0) {
            int started = 0, here = 0;
            const int key = cad_cu_key();
            for (int i = 0; i < nsets; ++i) {
                const int* base = sets.s[i].counters + sets.s[i].SB * nchunks;
                started += cad_counter_load_agent(base);
                here += cad_counter_load_agent(base + 1 + key);
            }
            pick_s[0] = (here == 0 && started < total) ? 1 : 0;
        }
        __syncthreads();
        if (pick_s[0]) {
            for (int j = t; j < nitems; j += FOLD_T) {
                int* as = abort_slot(j);
                if (as) *as = (int)nchunks;
            }
            return;
        }
        __syncthreads();  // pick_s is rewritten below
    }
    for (int j = t; j < nitems; j += FOLD_T) {
        int c = (int)nchunks - 1;
        if (mode == CAD_FOLD_CLEANUP) {  // what a concurrent pass left (stored as chunk + 1: 0 = nothing)
            int* as = abort_slot(j);
            c = (as ? *as : 0) - 1;
            if (as) *as = 0;
        }
        nextc[j] = c;
    }
    __syncthreads();
    int first = 0;  // (thread 0 only) items before `first` are done
    uint64_t t_last = (mode == CAD_FOLD_CONCURRENT && t == 0) ? cad_wall_clock() : 0;
    // Polling costs the scan next door (every poll is a load through the CU's memory pipeline that its staging waves wait on: same-box
    // A/B, layer 7.60 -> 7.49 ms with 4x longer sleeps): thread 0 learns the cadence of the arrivals (a chunk every ~13 us) and sleeps
    // through most of the predicted gap after a fold -- one or two failed polls per chunk instead of five to ten.
    uint64_t period = 0;       // ticks between the last two picks that had to wait (0: unknown)
    uint64_t t_wake = 0;       // do not poll before this time
    for (;;) {
        // ---- choose: the first item (in dispatch order) whose next chunk is complete; at most 4 pending items are polled per round
        if (t == 0) {
            int pick = -1, give_up = 0, alive = 0;
            while (first < nitems && nextc[first] < 0) ++first;
            bool waited = false;
            for (;;) {
                if (mode == CAD_FOLD_CONCURRENT && t_wake) {
                    while (cad_wall_clock() < t_wake) cad_poll_sleep();
                    t_wake = 0;
                }
                int polled = 0;
                alive = 0;
                for (int j = first; j < nitems && pick < 0 && polled < 4; ++j) {
                    const int c = nextc[j];
                    if (c < 0) continue;
                    alive = 1;
                    if (mode != CAD_FOLD_CONCURRENT) {
                        pick = j;
                    } else {
                        const cad_fold_args& a = sets.s[item_set(j)];
                        ++polled;
                        if (cad_counter_load_agent(a.counters + (int64_t)item_row(j) * nchunks + c) >= G) pick = j;
                    }
                }
                if (pick >= 0 || !alive) break;
                if (cad_wall_clock() - t_last >= budget_ticks) {  // no arrival anywhere for the whole budget: not co-scheduled with a
                    give_up = 1;                                  // progressing scan -- leave the rest to the cleanup launch
                    break;
                }
                waited = true;
                cad_poll_sleep();
            }
            if (pick >= 0 && mode == CAD_FOLD_CONCURRENT) {
                const uint64_t now = cad_wall_clock();
                if (waited) {  // steady state: this chunk arrived while we were watching -- the next one is a period away
                    if (t_last) period = now - t_last;
                    if (period > 5000) period = 5000;           // (50 us: never sleep long on a stale estimate)
                    t_wake = now + period - period / FOLD_WAKE_DIV;  // wake a fraction of the period early
                }
                t_last = now;
            }
            pick_s[0] = pick, pick_s[1] = give_up;
        }
        __syncthreads();
        const int j = cad_uniform(pick_s[0]);  // (workgroup-uniform: everything derived from the item stays in scalar registers)
        if (j < 0) {
            if (pick_s[1]) {
                for (int q = t; q < nitems; q += FOLD_T) {
                    int* as = abort_slot(q);
                    if (nextc[q] >= 0 && as) *as = nextc[q] + 1;
                }
            }
            return;  // everything folded, or given up
        }
        const int64_t c = cad_uniform(nextc[j]);
        // ---- fold chunk c of item j
        const cad_fold_args& a = sets.s[item_set(j)];
        const int x = item_slice(j);
        const int64_t sb = item_row(j);
        const int e0 = x * EPW + v * 8, r = e0 / FOLD_CHUNK, p = e0 % FOLD_CHUNK;
        const int ten = cad_uniform((x * EPW) / (N * FOLD_CHUNK));  // a slice (EPW divides N 512) never straddles the two tensors: per WORKGROUP
        const int n = r % N;
        const int rev = sb < a.split ? a.rev_lo : a.rev_hi;
        const int64_t cphys = rev ? L - (c + 1) * FOLD_CHUNK : c * FOLD_CHUNK;
        // element offsets inside one tensor's slots fit 32 bits (the launcher checks n_partials N SB L 2 < 2^32): lane arithmetic in 32 bits
        const uint32_t row_off = ((uint32_t)n * (uint32_t)SB + (uint32_t)sb) * (uint32_t)L + (uint32_t)p + (uint32_t)cphys;
        // slot k of the lane's group at (workgroup-uniform base of the tensor + k part_stride, scalar registers) + a 32-bit lane offset
        const char* tbase = (const char*)(ten ? a.dC_slots : a.dB_slots);
        const uint32_t voff = ((uint32_t)(grp * CAD_FOLD_GROUP) * (uint32_t)part_stride + row_off) * 2u;
        char* dbase = (char*)(ten ? a.dC : a.dB);
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (active) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {  // two rounds of four loads: 16 data registers instead of 32
                const void* base[4];
                u32x4 w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) base[k] = tbase + (int64_t)(4 * h + k) * part_stride * 2;
                cad_load16x4_wt(base, voff, w);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s[2 * q] += cad_bits2f(w[k][q] << 16);
                        s[2 * q + 1] += cad_bits2f(w[k][q] & 0xffff0000u);
                    }
                }
                // the sums of this round exist before the next round's loads are issued (otherwise the scheduler hoists those loads
                // and both rounds' 32 data registers are live at once: 52 instead of 40 VGPRs)
#pragma unroll
                for (int q = 0; q < 8; ++q) cad_order_point(s[q]);
            }
        }
        // group sums through LDS in two rounds (groups 1 .. NG/2 - 1, then NG/2 .. NG - 1): half the staging area -- 4 KB, so that the
        // workgroup fits behind TWO resident 76 KB scan workgroups (SC_W_BWD = 4) as well as behind one of 132 KB -- same order of additions
        const int gh = NG / 2;
        if (active && grp > 0 && grp < gh) {
#pragma unroll
            for (int q = 0; q < 8; ++q) part[(grp * VPS + v) * 8 + q] = s[q];
        }
        __syncthreads();
        if (active && grp == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) s[q] = 0.f + s[q];  // (group 0 first: the accumulator of the group sums starts from 0)
#pragma unroll 1
            for (int g = 1; g < gh; ++g) {  // (not unrolled: the kernel must fit the 48 VGPRs two resident scan waves leave on a SIMD)
#pragma unroll
                for (int q = 0; q < 8; ++q) s[q] += part[(g * VPS + v) * 8 + q];
            }
        }
        __syncthreads();
        if (active && grp > 0 && grp >= gh) {
#pragma unroll
            for (int q = 0; q < 8; ++q) part[((grp - gh) * VPS + v) * 8 + q] = s[q];
        }
        __syncthreads();
        if (active && grp == 0) {
#pragma unroll 1
            for (int g = (gh > 1 ? gh : 1); g < NG; ++g) {
#pragma unroll
                for (int q = 0; q < 8; ++q) s[q] += part[((g - gh) * VPS + v) * 8 + q];
            }
            u32x4 ov;
            ov[0] = cad_pack_bf16x2(s[0], s[1]), ov[1] = cad_pack_bf16x2(s[2], s[3]);
            ov[2] = cad_pack_bf16x2(s[4], s[5]), ov[3] = cad_pack_bf16x2(s[6], s[7]);
            *(u32x4*)(dbase + row_off * 2u) = ov;
        }
        if (t == 0) nextc[j] = (int)c - 1;
        __syncthreads();  // `part`, nextc and pick_s are rewritten by the next round
    }
}

}  // namespace

SC_TIME_EXPORT(cad_debug_timing_bwd)

extern "C" int cad_scan_bwd_partials(int E) { return (E + SC_W - 1) / SC_W; }

extern "C" int cad_scan_bwd_multi(const cad_scan_bwd_args* sets, int nsets, void* stream) {
    CAD_CHECK_ARG(sets && nsets >= 1 && nsets <= SC_MAXSETS);
    ScanBwdSets ks;
    for (int i = 0; i < nsets; ++i) {
        const cad_scan_bwd_args* a = &sets[i];
        CAD_CHECK_ARG(a->u && a->delta && a->A && a->Bm && a->Cm && a->dout);
        if (a->carry_only) {
            CAD_CHECK_ARG(a->dh0 != nullptr);
        } else {
            CAD_CHECK_ARG(a->chunk_state && a->du && a->ddelta && a->dA && a->dB && a->dC);
        }
        CAD_CHECK_ARG(a->carry_only == sets[0].carry_only);
        // the concurrent fold reads write-through slots: bf16, 16-byte stores of the packed flush, whole chunks
        CAD_CHECK_ARG(a->carry_only || !a->fold_counters || cad_fold_stream_supported(a->N, a->n_partials, a->L, a->dtype));
        CAD_CHECK_ARG(a->z != nullptr || a->dz == nullptr);               // dz needs the gate; dz == NULL: not wanted here
        CAD_CHECK_ARG(a->dz == nullptr || a->out != nullptr);
        CAD_CHECK_ARG(a->out2 == nullptr || a->dz != nullptr);
        CAD_CHECK_ARG(a->E > 0 && a->SB > 0 && a->L > 0 && a->N > 0 && a->N <= SC_NMAX);
        CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB && a->SB <= 65535);
        CAD_CHECK_ARG(a->carry_only || a->n_partials == cad_scan_bwd_partials(a->E));
        CAD_CHECK_ARG(a->E == sets[0].E && a->SB == sets[0].SB && a->L == sets[0].L && a->N == sets[0].N &&
                      a->dtype == sets[0].dtype);
        ks.s[i] = *a;
    }
    for (int i = nsets; i < SC_MAXSETS; ++i) ks.s[i] = sets[0];
    const cad_scan_bwd_args* a = &sets[0];
    bool vec = (a->L % SC_S) == 0;
    for (int i = 0; i < nsets; ++i)
        vec = vec && (((uintptr_t)sets[i].u | (uintptr_t)sets[i].delta | (uintptr_t)sets[i].z | (uintptr_t)sets[i].dout | (uintptr_t)sets[i].out |
                       (uintptr_t)sets[i].du | (uintptr_t)sets[i].ddelta | (uintptr_t)sets[i].dz | (uintptr_t)sets[i].out2 |
                       (uintptr_t)sets[i].Bm | (uintptr_t)sets[i].Cm | (uintptr_t)sets[i].dB | (uintptr_t)sets[i].dC) %
                      16) == 0;
    for (int i = 0; i < nsets; ++i) CAD_CHECK_ARG(sets[i].carry_only || !sets[i].fold_counters || vec);  // (16-byte write-through slot stores)
    CadProfScope prof(1, stream);
    dim3 grid((unsigned)((a->E + SC_W - 1) / SC_W), (unsigned)a->SB, (unsigned)nsets), block(64 * SC_W);
    const bool packed = SC_SLAB_PACKED && a->dtype == CAD_BF16 && (SC_W == 8 || SC_W == 4) && SC_SLAB_BUFS == 2;
    const bool pref = SC_BWD_PREFETCH && packed && vec && SC_S * 2 == 16;
    bool all_dt = true;  // every set hands over dt itself: the lean production instantiation (ISDT)
    for (int i = 0; i < nsets; ++i) all_dt = all_dt && sets[i].delta_is_dt != 0;
    const size_t slab_floats = a->carry_only ? 0 : (packed ? 2 * PK_BUF : SC_SLAB_BUFS * ACC_BUF);
    const size_t shmem = (size_t)(4 * SC_TILE(SC_S) + slab_floats) * sizeof(float) +
                         (pref ? PRE_BYTES : 0);
    // The lean production instantiation (ISDT) exists only in builds whose tuning defines allow it (its static_assert: LDS-DMA prefetch,
    // packed slab, 8-wave workgroups, unrolled pair loop): in every other variant build (-DSC_BWD_PREFETCH=0, -DSC_SLAB_PACKED=0, -DSC_W_BWD=4,
    // -DSC_BWD_UNROLL_NP=0, -DSC_BWD_LEAN=0) the template argument below is `false` and the branch is the unrolled round-4 kernel again.
    constexpr bool kLeanBuild = SC_BWD_LEAN && (SC_BWD_UNROLL_NP != 0) && SC_BWD_PREFETCH && SC_SLAB_PACKED && (SC_W == 8 || SC_W == 4) && SC_SLAB_BUFS == 2;
#define SC_BWD_LAUNCH(T, V)                                                                  \
    do {                                                                                     \
        if (kLeanBuild && SC_BWD_UNROLL_NP && !a->carry_only && V && sizeof(T) == 2 && pref && all_dt &&   \
            a->N == 2 * SC_BWD_UNROLL_NP) {                                                  \
            SC_BIG_LDS((scan_bwd_kernel<T, V, false, SC_BWD_UNROLL_NP, kLeanBuild && V && sizeof(T) == 2>), shmem);             \
            CAD_LAUNCH((scan_bwd_kernel<T, V, false, SC_BWD_UNROLL_NP, kLeanBuild && V && sizeof(T) == 2>), grid, block, shmem, stream, ks); \
        } else if (SC_BWD_UNROLL_NP && !a->carry_only && V && sizeof(T) == 2 && a->N == 2 * SC_BWD_UNROLL_NP) { \
            SC_BIG_LDS((scan_bwd_kernel<T, V, false, SC_BWD_UNROLL_NP>), shmem);             \
            CAD_LAUNCH((scan_bwd_kernel<T, V, false, SC_BWD_UNROLL_NP>), grid, block, shmem, stream, ks); \
        } else if (a->carry_only) {                                                          \
            SC_BIG_LDS((scan_bwd_kernel<T, V, true>), shmem);                                \
            CAD_LAUNCH((scan_bwd_kernel<T, V, true>), grid, block, shmem, stream, ks);       \
        } else {                                                                             \
            SC_BIG_LDS((scan_bwd_kernel<T, V, false>), shmem);                               \
            CAD_LAUNCH((scan_bwd_kernel<T, V, false>), grid, block, shmem, stream, ks);      \
        }                                                                                    \
    } while (0)
    if (a->dtype == CAD_F32) {
        if (vec)
            SC_BWD_LAUNCH(float, true);
        else
            SC_BWD_LAUNCH(float, false);
    } else if (a->dtype == CAD_BF16) {
        if (vec)
            SC_BWD_LAUNCH(bf16_t, true);
        else
            SC_BWD_LAUNCH(bf16_t, false);
    } else {
        return CAD_ERR_UNSUPPORTED;
    }
#undef SC_BWD_LAUNCH
    return cad_after_launch();
}

extern "C" int cad_scan_bwd(const cad_scan_bwd_args* a, void* stream) { return cad_scan_bwd_multi(a, 1, stream); }

// diagnostic (tools/gpu_w4.sh): how many workgroups of the production instantiation the runtime places on one CU, and its LDS bytes
extern "C" int cad_debug_scan_bwd_occupancy(int* out) {
    const size_t shmem = (size_t)(4 * SC_TILE(SC_S) + 2 * PK_BUF) * sizeof(float) + PRE_BYTES;
    out[0] = CAD_OCCUPANCY((scan_bwd_kernel<bf16_t, true, false, SC_BWD_UNROLL_NP>), 64 * SC_W, shmem);
    out[1] = (int)shmem, out[2] = SC_W;
    return out[0] > 0 ? CAD_OK : CAD_ERR_UNSUPPORTED;
}

extern "C" int64_t cad_scan_gate_fix_entries(int E, int64_t SB, int64_t L) {
    return (int64_t)E * SB * ((L + SC_CHUNK - 1) / SC_CHUNK);
}

extern "C" int cad_scan_bwd_gate_fix(const cad_scan_bwd_args* sets, int nsets, void* stream) {
    CAD_CHECK_ARG(sets && nsets >= 1 && nsets <= SC_MAXSETS);
    for (int i = 0; i < nsets; ++i) {  // one launch per set, in order: sets sharing a dz buffer add to it one after the other
        const cad_scan_bwd_args* a = &sets[i];
        if (!a->gate_fix_list) continue;
        CAD_CHECK_ARG(a->gate_fix_count && a->gate_fix_dz && a->z && a->chunk_state);
        CAD_CHECK_ARG(a->E <= (1 << 20) && a->SB <= (1 << 20));
        // an empty or one-entry worklist as a rule (bf16 products: P(z == 0) ~ 4e-9) -- but thousands of entries per launch behind the fp8
        // in_proj, whose sums of few-bit products cancel EXACTLY far more often: 32 workgroups took 120 us per launch there (3.8 ms per
        // configs[4] step, profiles/r06_step_trace_c4_fp8.txt), one per CU takes the list in parallel; idle workgroups leave at once
        dim3 grid((unsigned)cad_cu_count()), block(64 * GF_WAVES);
        if (a->dtype == CAD_F32)
            CAD_LAUNCH((scan_gate_fix_kernel<float>), grid, block, 0, stream, *a);
        else if (a->dtype == CAD_BF16)
            CAD_LAUNCH((scan_gate_fix_kernel<bf16_t>), grid, block, 0, stream, *a);
        else
            return CAD_ERR_UNSUPPORTED;
    }
    return cad_after_launch();
}

namespace {
__global__ __launch_bounds__(64) void stream_probe_wait_kernel(const int* flag, int* result, uint64_t budget_ticks) {
    if (threadIdx.x != 0) return;
    const uint64_t t0 = cad_wall_clock();
    int seen = 0;
    do {
        seen = cad_counter_load_agent(flag) != 0;
        if (!seen) cad_poll_sleep();
    } while (!seen && cad_wall_clock() - t0 < budget_ticks);
    result[0] = seen;
}
__global__ __launch_bounds__(64) void stream_probe_set_kernel(int* flag) {
    if (threadIdx.x == 0) cad_counter_add_agent(flag, 1);
}
}  // namespace

extern "C" int cad_stream_probe(void* stream_a, void* stream_b, int* flag, int* result, int64_t budget_us) {
    CAD_CHECK_ARG(flag && result && budget_us > 0 && budget_us <= 1000000);
    CAD_LAUNCH(stream_probe_wait_kernel, dim3(1), dim3(64), 0, stream_a, (const int*)flag, result,
               (uint64_t)budget_us * CAD_WALL_CLOCK_TICKS_PER_US);
    CAD_LAUNCH(stream_probe_set_kernel, dim3(1), dim3(64), 0, stream_b, flag);
    return cad_after_launch();
}

extern "C" int64_t cad_scan_bwd_chunk_len(void) { return SC_CHUNK; }
extern "C" int64_t cad_scan_bwd_fold_counter_ints(int64_t SB, int64_t L) {
    return SB * ((L + SC_CHUNK - 1) / SC_CHUNK) + 1 + CAD_CU_KEYS;
}

extern "C" int cad_fold_stream_supported(int N, int n_partials, int64_t L, int dtype) {
    if (dtype != CAD_BF16 || N < 1 || L < FOLD_CHUNK || L % FOLD_CHUNK != 0 || SC_CHUNK != FOLD_CHUNK) return 0;
    if (n_partials < CAD_FOLD_GROUP || n_partials > 2048 || (n_partials & (n_partials - 1)) != 0) return 0;
    const int elems = 2 * N * FOLD_CHUNK;
    if (elems % n_partials != 0) return 0;
    const int epw = elems / n_partials;
    if (epw < 8 || epw % 8 != 0) return 0;
    if (n_partials % CAD_FOLD_GROUP != 0) return 0;  // whole groups of 8 slots
    const int vps = epw / 8;                         // vectors per slot and workgroup; (n_partials / 8) x vps = 16 N threads work
    if ((n_partials / CAD_FOLD_GROUP) * vps > FOLD_T) return 0;    // d_state <= 16
    if (FOLD_CHUNK % epw != 0 && epw % FOLD_CHUNK != 0) return 0;  // a slice lies inside one row, or covers whole rows
    if ((N * FOLD_CHUNK) % epw != 0) return 0;                     // ... and inside one tensor
    // (8-wave workgroups only: the 4-wave variant -- two workgroups per CU, profiles/r06_ab_w4_workgroups.txt -- fails one device test next to
    // the concurrent fold and is 15 % slower without it)
    return SC_BWD_SLOT_WT && SC_SLAB_PACKED && SC_W == 8 && SC_SLAB_BUFS == 2;  // the write-through slot stores of the packed flush
}

extern "C" int cad_fold_partials_stream(const cad_fold_args* sets, int nsets, int mode, void* stream) {
    CAD_CHECK_ARG(sets && nsets >= 1 && nsets <= SC_MAXSETS);
    CAD_CHECK_ARG(mode == CAD_FOLD_CONCURRENT || mode == CAD_FOLD_CLEANUP || mode == CAD_FOLD_ALL);
    FoldSets ks;
    for (int i = 0; i < nsets; ++i) {
        const cad_fold_args* a = &sets[i];
        CAD_CHECK_ARG(a->dB_slots && a->dC_slots && a->dB && a->dC && a->SB > 0 && a->SB <= 65535);
        CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB);
        if (!cad_fold_stream_supported(a->N, a->n_partials, a->L, a->dtype)) return CAD_ERR_UNSUPPORTED;
        CAD_CHECK_ARG((((uintptr_t)a->dB_slots | (uintptr_t)a->dC_slots | (uintptr_t)a->dB | (uintptr_t)a->dC) % 16) == 0);
        if ((int64_t)a->n_partials * a->N * a->SB * a->L * 2 >= ((int64_t)1 << 32)) return CAD_ERR_UNSUPPORTED;  // 32-bit lane offsets (per tensor)
        CAD_CHECK_ARG(mode != CAD_FOLD_CONCURRENT || (a->counters && a->abort_from));
        CAD_CHECK_ARG(mode != CAD_FOLD_CLEANUP || a->abort_from);
        CAD_CHECK_ARG(a->N == sets[0].N && a->n_partials == sets[0].n_partials && a->L == sets[0].L && a->SB == sets[0].SB);
        ks.s[i] = *a;
    }
    for (int i = nsets; i < SC_MAXSETS; ++i) ks.s[i] = sets[0];
    // a poll that sees no arrival for this long gives the chunk (and the rest of the row slice) to the cleanup launch: the scan produces a
    // chunk every ~13 us, so 20 ms means "the scan is not running next to us" (serialised queues, a profiler, a debugger)
    const uint64_t budget = 2000000ull;  // ticks of the 100 MHz wall clock
    const int64_t items = (int64_t)sets[0].n_partials * sets[0].SB * nsets;
    const int cus = cad_cu_count();  // one workgroup per CU at most (see the kernel)
    if (items > (int64_t)cus * FOLD_MAX_ITEMS) return CAD_ERR_UNSUPPORTED;
    dim3 grid((unsigned)(items < cus ? items : cus)), block(FOLD_T);
    if (mode == CAD_FOLD_CONCURRENT)  // (same stream: the fold kernel is dispatched when the gate has returned)
        CAD_LAUNCH(fold_gate_kernel, dim3(1), dim3(64), 0, stream, ks, nsets, (int)items, budget);
    CAD_LAUNCH(fold_stream_kernel, grid, block, 0, stream, ks, nsets, mode, budget);
    return cad_after_launch();
}

extern "C" int cad_reduce_partials(const void* src, int n_partials, int64_t n, void* dst, int dst_dtype, void* stream) {
    CAD_CHECK_ARG(src && dst && n_partials >= 1 && n > 0);
    const int vec = (n % 4) == 0 && (((uintptr_t)src | (uintptr_t)dst) % 16) == 0;
    int64_t nb = (n / 4 + 255) / 256 + 1;
    if (nb > 16384) nb = 16384;
    dim3 grid((unsigned)nb), block(256);
    if (dst_dtype == CAD_F32)
        CAD_LAUNCH((reduce_partials_kernel<float>), grid, block, 0, stream, (const float*)src, n_partials, n, (float*)dst, vec);
    else if (dst_dtype == CAD_BF16)
        CAD_LAUNCH((reduce_partials_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)src, n_partials, n, (bf16_t*)dst, vec);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}

extern "C" int cad_reduce_partials_multi(const cad_reduce_job* jobs, int njobs, int n_partials, int64_t n, int dst_dtype, void* stream) {
    CAD_CHECK_ARG(jobs && njobs >= 1 && njobs <= CAD_REDUCE_MAX_JOBS && n_partials >= 1 && n > 0);
    ReduceJobs kj;
    bool vec = (n % 4) == 0;
    for (int i = 0; i < CAD_REDUCE_MAX_JOBS; ++i) {
        const cad_reduce_job& j = jobs[i < njobs ? i : 0];
        CAD_CHECK_ARG(j.src && j.dst);
        kj.src[i] = j.src, kj.dst[i] = j.dst;
        vec = vec && (((uintptr_t)j.src | (uintptr_t)j.dst) % 16) == 0;
    }
    if (!vec) {  // ragged / unaligned: one plain fold per job
        for (int i = 0; i < njobs; ++i) {
            const int rc = cad_reduce_partials(jobs[i].src, n_partials, n, jobs[i].dst, dst_dtype, stream);
            if (rc != CAD_OK) return rc;
        }
        return CAD_OK;
    }
    int64_t nb = (n / 4 + 255) / 256 + 1;
    if (nb > 16384) nb = 16384;
    dim3 grid((unsigned)nb, (unsigned)njobs), block(256);
    if (dst_dtype == CAD_F32)
        CAD_LAUNCH((reduce_partials_multi_kernel<float>), grid, block, 0, stream, kj, n_partials, n);
    else if (dst_dtype == CAD_BF16)
        CAD_LAUNCH((reduce_partials_multi_kernel<bf16_t>), grid, block, 0, stream, kj, n_partials, n);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}
