// Depthwise causal conv1d (+bias +SiLU) along L on channel-major (E, SB, L) activations, with a per-row direction
// map (include/caduceus_hip.h, cad_conv1d_*).  HBM-bound streaming kernels.
//
// A lane owns 8 consecutive PHYSICAL positions (one 16-byte bf16 / 32-byte fp32 vector); the 3-position halos on both
// sides come from the neighbouring lanes through DPP wave shifts, so every element of x / dout is loaded exactly once.
// Lanes 0 and 63 of a wave are halo lanes (they load and pre-compute but do not store), i.e. a wave produces 62 * 8 =
// 496 positions.  Working in physical order makes the direction a choice of taps instead of an index map: a row that
// runs left-to-right uses the taps on its left, a right-to-left row the mirrored taps on its right (template parameter,
// selected by a wave-uniform branch).
// Up to two parameter sets that read the SAME x (mamba_fwd / mamba_rev of a BiMamba layer, which see the same in_proj
// output in opposite directions) run in one launch: x is read once, and in the backward dx = dx_0 + dx_1 is written once.
#include "cad_common.h"
#include "cad_stream.h"

namespace {

#define CV_VEC 8
#define CV_WAVES 4
#define CV_THREADS (64 * CV_WAVES)
#define CV_KMAX 4
#define CV_TAPS (2 * CV_KMAX - 1)                 // 7-tap window: offsets -3 .. +3
#define CV_HALO (CV_KMAX - 1)
#define CV_WAVE_POS (62 * CV_VEC)                 // useful positions per wave tile
#define CV_TILES_BWD 8                            // wave tiles a backward workgroup walks before reducing dw / dbias
#define CV_TILES_FWD 4                            // wave tiles a forward wave walks (the next tile's load in flight under the arithmetic)
#define CV_MAXSETS 2
#ifndef CV_BWD_WAVES
#define CV_BWD_WAVES 4                            // waves per SIMD the backward is compiled for (<= 128 VGPRs)
#endif

struct ConvFwdSets {
    cad_conv1d_args s[CV_MAXSETS];
};
struct ConvBwdSets {
    cad_conv1d_bwd_args s[CV_MAXSETS];
};

template <typename T>
struct __attribute__((aligned(sizeof(T) * CV_VEC))) CvVec {
    T v[CV_VEC];
};

// the same 8 positions as they lie in memory (zeros outside [0, L)), converted later: the load of the NEXT tile is issued before
// the arithmetic of the current one, so a wave keeps two tiles in flight
// Raw tiles are kept as opaque 32-bit words: a vector of 16-bit elements is taken apart by the compiler right behind its load (and
// the load is then waited for at once), words are only touched where cvt8 unpacks them.
template <typename T>
struct __attribute__((aligned(sizeof(T) * CV_VEC))) CvRaw {
    uint32_t w[sizeof(T) * CV_VEC / 4];
};
template <typename T, bool VEC>
__device__ __forceinline__ CvRaw<T> load8_raw(const T* row, int64_t l0, int64_t L) {
    CvRaw<T> r;
    if constexpr (VEC) {
        // L % 8 == 0 and l0 % 8 == 0: a vector lies either inside [0, L) or outside.  The load is UNCONDITIONAL (from a clamped
        // address; cvt8 zeroes a vector that lay outside): a load under a divergent branch is waited for at the end of the branch
        int64_t lc = l0 < 0 ? 0 : l0;
        lc = lc < L ? lc : L - CV_VEC;
        r = *(const CvRaw<T>*)(row + lc);
    } else {
        T e[CV_VEC];
#pragma unroll
        for (int j = 0; j < CV_VEC; ++j) {
            const int64_t l = l0 + j;
            e[j] = (l >= 0 && l < L) ? row[l] : from_f32<T>(0.f);
        }
        __builtin_memcpy(&r, e, sizeof(r));
    }
    return r;
}
template <typename T, bool VEC>
__device__ __forceinline__ void cvt8(const CvRaw<T>& r, int64_t l0, int64_t L, float* out) {
    const bool inside = !VEC || (l0 >= 0 && l0 < L);
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < CV_VEC / 2; ++j) {
            out[2 * j] = inside ? cad_bits2f(r.w[j] << 16) : 0.f;
            out[2 * j + 1] = inside ? cad_bits2f(r.w[j] & 0xffff0000u) : 0.f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < CV_VEC; ++j) out[j] = inside ? cad_bits2f(r.w[j]) : 0.f;
    }
}
// fp32 -> the raw words of 8 output elements (rounded like store8v), and their store: the vector kernels keep a tile's packed results
// in registers and store them at the START of the next tile -- loads and stores return out of order with respect to each other, so
// waiting for the next tile's load is a vmcnt(0) that also waits for every store in flight; stores issued a tile's arithmetic
// earlier have landed by then
template <typename T>
__device__ __forceinline__ CvRaw<T> pack8(const float* v) {
    CvRaw<T> r;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < CV_VEC / 2; ++j)
            r.w[j] = (uint32_t)from_f32<T>(v[2 * j]).v | ((uint32_t)from_f32<T>(v[2 * j + 1]).v << 16);
    } else {
#pragma unroll
        for (int j = 0; j < CV_VEC; ++j) r.w[j] = __builtin_bit_cast(uint32_t, v[j]);
    }
    return r;
}
template <typename T>
__device__ __forceinline__ void store_raw(T* row, int64_t l0, int64_t L, const CvRaw<T>& r) {
    if (l0 >= 0 && l0 < L) {  // written once, read by a later kernel: streaming stores, 16 bytes at a time
        typedef uint32_t cv4 __attribute__((vector_size(16)));
        struct P { cv4 q[sizeof(CvRaw<T>) / 16]; };
        const P p = __builtin_bit_cast(P, r);
#pragma unroll
        for (int i = 0; i < (int)(sizeof(CvRaw<T>) / 16); ++i) cad_store_stream<CAD_STREAM_CONV>((cv4*)(row + l0) + i, p.q[i]);
    }
}
// VEC: one vector store or nothing
template <typename T, bool VEC>
__device__ __forceinline__ void store8v(T* row, int64_t l0, int64_t L, const float* v) {
    if constexpr (VEC) {
        if (l0 >= 0 && l0 < L) {
            CvVec<T> tmp;
#pragma unroll
            for (int j = 0; j < CV_VEC; ++j) tmp.v[j] = from_f32<T>(v[j]);
            typedef uint32_t cv4 __attribute__((vector_size(16)));
            struct P { cv4 q[sizeof(CvVec<T>) / 16]; };
            const P p = __builtin_bit_cast(P, tmp);
#pragma unroll
            for (int i = 0; i < (int)(sizeof(CvVec<T>) / 16); ++i) cad_store_stream<CAD_STREAM_CONV>((cv4*)(row + l0) + i, p.q[i]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < CV_VEC; ++j) {
            const int64_t l = l0 + j;
            if (l >= 0 && l < L) row[l] = from_f32<T>(v[j]);
        }
    }
}
// own[8] -> window e[14] = 3 left-halo + 8 own + 3 right-halo values (neighbouring lanes; 0 at the wave edges)
__device__ __forceinline__ void halo_window(const float* own, float* e) {
#pragma unroll
    for (int i = 0; i < CV_HALO; ++i) {
        e[i] = dpp_wave_shr1(0.f, own[CV_VEC - CV_HALO + i]);
        e[CV_HALO + CV_VEC + i] = dpp_wave_shl1(0.f, own[i]);
    }
#pragma unroll
    for (int j = 0; j < CV_VEC; ++j) e[CV_HALO + j] = own[j];
}

// Taps.  Weights are held as w4[0..3] aligned to K = 4 (w4[k] = w[k - (4 - K)], zero for the missing leading taps), so
// that for any K <= 4 the conv of a left-to-right row is  b + sum_k w4[k] * win[k]  and of a right-to-left row
// b + sum_k w4[k] * win[6 - k]  over the 7-value window win[0..6] = x[l-3 .. l+3].  The direction is a template
// parameter (the row's direction is wave-uniform: one scalar branch selects the instantiation); taps are accumulated in
// the same order k = 0..3 for both directions, so a right-to-left row is the exact mirror of a left-to-right row.
template <int REV>
__device__ __forceinline__ float conv4(const float* w4, const float* win, float b) {
    float acc = b;
#pragma unroll
    for (int k = 0; k < CV_KMAX; ++k) acc = __builtin_fmaf(w4[k], win[REV ? (CV_TAPS - 1 - k) : k], acc);  // explicit
    return acc;  // fused multiply-adds: -ffp-contract may otherwise fuse the two direction instantiations differently
}
__device__ __forceinline__ void load_w4(const float* w, int e, int K, float* w4) {
#pragma unroll
    for (int k = 0; k < CV_KMAX; ++k) w4[k] = (k >= CV_KMAX - K) ? w[e * K + k - (CV_KMAX - K)] : 0.f;
}
template <int REV>
struct DirTag {
    static constexpr int value = REV;
};

template <typename T, int NSETS, bool VEC>
__global__ __launch_bounds__(CV_THREADS) void conv1d_fwd_kernel(ConvFwdSets sets) {
    const cad_conv1d_args& a0 = sets.s[0];
    // A workgroup belongs to ONE channel e and walks "virtual tiles" v = row * tiles_per_row + tile over that channel's rows: long rows
    // (one sequence per row) behave as before, short rows (L = 1024, 128 rows) no longer leave a wave idle and a workgroup with a
    // single tile of work per wave (the backward also folds its per-channel sums once per 32 tiles instead of once per row).
    const int e = blockIdx.x;
    const int64_t L = a0.L, SB = a0.SB;
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const uint32_t tpr = (uint32_t)((L + CV_WAVE_POS - 1) / CV_WAVE_POS);  // tiles per row
    const uint32_t nvt = (uint32_t)SB * tpr;                               // (the launcher checks that this fits)
    float W4[NSETS][CV_KMAX], bias[NSETS];
#pragma unroll
    for (int s = 0; s < NSETS; ++s) {
        const cad_conv1d_args& a = sets.s[s];
        load_w4(a.w, e, a.K, W4[s]);
        bias[s] = a.bias ? a.bias[e] : 0.f;
    }
    const uint32_t v0 = blockIdx.y * (CV_TILES_FWD * CV_WAVES) + wave;
    if (v0 >= nvt) return;  // wave-uniform
    const bool useful = lane >= 1 && lane <= 62;
    auto locate = [&](uint32_t v, int64_t& row, int64_t& sb, int64_t& l0) {
        const uint32_t r = v / tpr;
        sb = r;
        row = (int64_t)e * SB + r;
        l0 = (int64_t)(v - r * tpr) * CV_WAVE_POS + (int64_t)(lane - 1) * CV_VEC;
    };
    int64_t row, sb, l0;
    locate(v0, row, sb, l0);
    CvRaw<T> raw = load8_raw<T, VEC>((const T*)a0.x + row * L, l0, L);
    CvRaw<T> po[NSETS];      // the previous tile's packed results (VEC), stored one tile late
    int64_t l0_prev = 0, row_prev = 0;
    for (int it = 0; it < CV_TILES_FWD; ++it) {
        const uint32_t v = v0 + it * CV_WAVES;  // the workgroup's waves cover 4 neighbouring tiles at a time
        if (v >= nvt) break;                     // wave-uniform
        float own[CV_VEC], xe[CV_VEC + 2 * CV_HALO];
        cvt8<T, VEC>(raw, l0, L, own);
        if constexpr (VEC) {
            // the stores stay BEHIND the wait for this tile's load: the scheduler would hoist them in front of it (and the wait
            // for a load is a vmcnt(0) that then waits for the stores just issued).  The asm consumes one converted value, so the
            // wait is placed before it, and orders the memory operations around it.
#ifndef CAD_EMU
            asm volatile("" : "+v"(own[0]) : : "memory");
#endif
            if (it > 0 && useful) {
#pragma unroll
                for (int s = 0; s < NSETS; ++s) store_raw<T>((T*)sets.s[s].out + row_prev * L, l0_prev, L, po[s]);
            }
        }
        const int64_t row_c = row, sb_c = sb, l0_c = l0;
        if (it + 1 < CV_TILES_FWD && v + CV_WAVES < nvt) {
            locate(v + CV_WAVES, row, sb, l0);
            raw = load8_raw<T, VEC>((const T*)a0.x + row * L, l0, L);
        }
        halo_window(own, xe);
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            float o[CV_VEC];
            auto body = [&](auto dir) {
                constexpr int REV = decltype(dir)::value;
#pragma unroll
                for (int j = 0; j < CV_VEC; ++j) {
                    const float acc = conv4<REV>(W4[s], xe + j, bias[s]);
                    o[j] = acc * cad_sigmoid(acc);
                }
            };
            const int rev = sb_c < sets.s[s].split ? sets.s[s].rev_lo : sets.s[s].rev_hi;  // wave-uniform: the tile's row
            if (rev) body(DirTag<1>{}); else body(DirTag<0>{});
            if constexpr (VEC) {
                po[s] = pack8<T>(o);
            } else {
                if (useful) store8v<T, VEC>((T*)sets.s[s].out + row_c * L, l0_c, L, o);
            }
        }
        l0_prev = l0_c;
        row_prev = row_c;
    }
    if constexpr (VEC) {
        if (useful) {  // (at least one tile was computed: the wave returned above otherwise)
#pragma unroll
            for (int s = 0; s < NSETS; ++s) store_raw<T>((T*)sets.s[s].out + row_prev * L, l0_prev, L, po[s]);
        }
    }
}

// Backward.  dpre[l] = dout[l] * silu'(pre[l]);  dx = sum over the sets of the transposed conv of dpre (the taps of
// the opposite direction);  dw[k] = sum_l dpre[l] * x[l + offset_k];  dbias = sum_l dpre[l].
template <typename T, int NSETS, bool VEC, bool ACC>
__global__ __launch_bounds__(CV_THREADS, CV_BWD_WAVES) void conv1d_bwd_kernel(ConvBwdSets sets) {
    __shared__ float red[CV_WAVES][NSETS][CV_KMAX + 1];
    const cad_conv1d_bwd_args& a0 = sets.s[0];
    const int e = blockIdx.x;  // one channel per workgroup, virtual tiles over its rows (see the forward)
    const int64_t L = a0.L, SB = a0.SB;
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const uint32_t tpr = (uint32_t)((L + CV_WAVE_POS - 1) / CV_WAVE_POS);
    const uint32_t nvt = (uint32_t)SB * tpr;
    float W4[NSETS][CV_KMAX], bias[NSETS], part[NSETS][CV_KMAX + 1];  // part: dw4[0..3], dbias
#pragma unroll
    for (int s = 0; s < NSETS; ++s) {
        const cad_conv1d_bwd_args& a = sets.s[s];
        load_w4(a.w, e, a.K, W4[s]);
#pragma unroll
        for (int m = 0; m <= CV_KMAX; ++m) part[s][m] = 0.f;
        bias[s] = a.bias ? a.bias[e] : 0.f;
    }
    const float useful = (lane >= 1 && lane <= 62) ? 1.f : 0.f;
    auto locate = [&](uint32_t v, int64_t& row, int64_t& sb, int64_t& l0) {
        const uint32_t r = v / tpr;
        sb = r;
        row = (int64_t)e * SB + r;
        l0 = (int64_t)(v - r * tpr) * CV_WAVE_POS + (int64_t)(lane - 1) * CV_VEC;
    };
    // all loads of a tile (x, dout of every set) are issued one tile ahead, raw, and converted when the tile is computed
    // (the dx addend is loaded at the start of its own tile and added last: the arithmetic of the tile covers most of its latency)
    CvRaw<T> rx, rg[NSETS];
    int64_t row = 0, sb = 0, l0 = 0;  // of the tile in flight
    auto fetch = [&](uint32_t v) {
        locate(v, row, sb, l0);
        rx = load8_raw<T, VEC>((const T*)a0.x + row * L, l0, L);
#pragma unroll
        for (int s = 0; s < NSETS; ++s) rg[s] = load8_raw<T, VEC>((const T*)sets.s[s].dout + row * L, l0, L);
    };
    const uint32_t v0 = blockIdx.y * (CV_TILES_BWD * CV_WAVES) + wave;
    if (v0 < nvt) fetch(v0);
    // (the dx store is NOT delayed by a tile as the forward's stores are: measured equal, 0.260-0.265 against 0.261-0.266 ms, for four
    // more registers -- the backward's tile arithmetic is long enough to cover it)
    for (int it = 0; it < CV_TILES_BWD; ++it) {
        const uint32_t v = v0 + it * CV_WAVES;
        if (v >= nvt) break;  // wave-uniform
        const int64_t row_c = row, sb_c = sb, l0_c = l0;
        T* dx = (T*)a0.dx + row_c * L;
        float own[CV_VEC], xe[CV_VEC + 2 * CV_HALO], o[CV_VEC];
        cvt8<T, VEC>(rx, l0_c, L, own);
        CvRaw<T> rdx;  // (ACC is a template parameter: a run-time branch would merge the register behind it and wait for the load there)
        if constexpr (ACC) rdx = load8_raw<T, VEC>((const T*)dx, l0_c, L);
#pragma unroll
        for (int j = 0; j < CV_VEC; ++j) o[j] = 0.f;
        CvRaw<T> cg[NSETS];  // this tile's dout, still raw: converted set by set
#pragma unroll
        for (int s = 0; s < NSETS; ++s) cg[s] = rg[s];
        if (it + 1 < CV_TILES_BWD && v + CV_WAVES < nvt) fetch(v + CV_WAVES);
        halo_window(own, xe);
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            float g[CV_VEC], dpre[CV_VEC], dpe[CV_VEC + 2 * CV_HALO];
            cvt8<T, VEC>(cg[s], l0_c, L, g);
            auto body = [&](auto dir) {
                constexpr int REV = decltype(dir)::value;
#pragma unroll
                for (int j = 0; j < CV_VEC; ++j) {
                    const float acc = conv4<REV>(W4[s], xe + j, bias[s]);
                    const float sg = cad_sigmoid(acc);
                    dpre[j] = g[j] * sg * (1.f + acc * (1.f - sg));  // dout is 0 outside [0, L) -> so is dpre
                }
                halo_window(dpre, dpe);
#pragma unroll
                for (int j = 0; j < CV_VEC; ++j) {
                    o[j] += conv4<!REV>(W4[s], dpe + j, 0.f);  // transposed conv = the taps of the opposite direction
                    const float dm = dpre[j] * useful;        // halo lanes belong to the neighbouring tile
#pragma unroll
                    for (int k = 0; k < CV_KMAX; ++k) part[s][k] += dm * xe[j + (REV ? (CV_TAPS - 1 - k) : k)];
                    part[s][CV_KMAX] += dm;
                }
            };
            const int rev = sb_c < sets.s[s].split ? sets.s[s].rev_lo : sets.s[s].rev_hi;  // wave-uniform: the tile's row
            if (rev) body(DirTag<1>{}); else body(DirTag<0>{});
        }
        if constexpr (ACC) {
            float add[CV_VEC];
            cvt8<T, VEC>(rdx, l0_c, L, add);
#pragma unroll
            for (int j = 0; j < CV_VEC; ++j) o[j] += add[j];
        }
        if (useful != 0.f) store8v<T, VEC>(dx, l0_c, L, o);
    }
    // workgroup reduction of the dw / dbias partials, then a handful of atomics per workgroup
#pragma unroll
    for (int s = 0; s < NSETS; ++s)
#pragma unroll
        for (int m = 0; m <= CV_KMAX; ++m) {
            float v = part[s][m];
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) v += __shfl_xor(v, sh);
            if (lane == 0) red[wave][s][m] = v;
        }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < NSETS * (CV_KMAX + 1)) {
        const int s = t / (CV_KMAX + 1), k = t % (CV_KMAX + 1);
        const cad_conv1d_bwd_args& a = sets.s[s];
        if (k < CV_KMAX) {
            if (k < a.K) {  // w[k] lives in w4[k + 4 - K]
                float tot = 0.f;
                for (int wv = 0; wv < CV_WAVES; ++wv) tot += red[wv][s][k + CV_KMAX - a.K];
                if (tot != 0.f) atomicAdd(&a.dw[e * a.K + k], tot);
            }
        } else if (a.dbias) {
            float tot = 0.f;
            for (int wv = 0; wv < CV_WAVES; ++wv) tot += red[wv][s][CV_KMAX];
            if (tot != 0.f) atomicAdd(&a.dbias[e], tot);
        }
    }
}


__device__ __forceinline__ float wave_sum_conv(float v) {  // uniform result (every lane takes part): DPP row reduction + row broadcasts
    v += dpp_row_shr<1>(0.f, v);
    v += dpp_row_shr<2>(0.f, v);
    v += dpp_row_shr<4>(0.f, v);
    v += dpp_row_shr<8>(0.f, v);
    v += dpp_row_bcast15(0.f, v);
    v += dpp_row_bcast31(0.f, v);
    return cad_readlane(v, 63);
}
__device__ __forceinline__ void gp_wait_dma_conv() {
#ifndef CAD_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ======================================================================================================================
// Fused backward:  conv1d backward  +  the x_proj input gradient  +  the x_proj weight gradient.
//   d(xc) = du + W_x^T . d(dbc)      (the x_proj nn.Linear of mamba_inner_fn feeds on the conv output xc: its input gradient is
//                                     ADDED to the scan's du before the conv backward; used to be cad_proj_wx with an addend:
//                                     read du, write du -- 536 MB per parameter set that existed only to be read again here)
//   dW_x  = d(dbc) . xc^T             (used to be cad_proj_wx_wgrad's weight-gradient stage: re-read the 268 MB of xc per set;
//                                     here xc is RECOMPUTED from x -- the conv backward evaluates conv(x) and its sigmoid anyway)
//   dx, dw, dbias                     exactly as conv1d_bwd_kernel (same tap order, same rounding points)
// One workgroup (16 waves, one channel each: CX_CB = 16) walks 512-position tiles (496 useful, as above) of its rows.  Per tile and
// parameter set:  (1) the (M = dt_rank + 2 d_state, 512) tile of d(dbc) comes in by LDS-DMA (one 1 KB row per instruction, rows
// padded by 32 bytes: the transposing reads of eight consecutive rows hit distinct banks), a tile ahead of its use;  (2) P = W_x^T .
// d(dbc) for the 32 x 512 tile on the matrix cores (A = d(dbc) fragments by transposing reads, B = W_x^T fragments from global: a D
// lane holds four consecutive positions of one channel), rounded to bf16 into a [channel][position] staging tile -- the same product,
// instruction and rounding as cad_proj_wx;  (3) every wave runs the conv backward of its channel on
// d(xc) = bf16(du + P) and writes xc (bf16, zero on halo lanes) over P in the staging tile;  (4) dW_x[channel][m] += xc_tile .
// d(dbc)_tile^T on the matrix cores (both operands position-contiguous: plain 16-byte fragment reads), accumulated in registers over
// all tiles of the workgroup, one (E, M) fp32 partial slot per token group at the end.  dx of both sets is summed in registers.
#ifndef CX_WAVES
#define CX_WAVES 16                               // 1024 threads = four waves per SIMD: the conv arithmetic (two transcendentals and ~60
                                                  // VALU instructions per element and set) needs the occupancy -- with 8 waves that meet
                                                  // at barriers the same kernel ran at half the VALU rate (0.65 vs 0.28 ms of the plain
                                                  // conv backward, profiles/r04_ab_conv_xproj_fused.txt)
#endif
#ifndef CX_CPW
#define CX_CPW 1                                  // channels per wave (sequential): with 2 the second dx accumulator set pushes the kernel
                                                  // past 128 VGPRs -- spilled prefetch registers made every step wait for its own loads
#endif
#define CX_CB (CX_WAVES * CX_CPW)                 // channels per workgroup (a multiple of 16)
#define CX_NCBK (CX_CB / 16)                      // 16-channel MFMA blocks
static_assert(CX_CB % 16 == 0 && CX_CB <= 32 && 32 % CX_WAVES == 0, "channel blocks of the matrix-core stages");
#define CX_TW 512                                 // positions per tile (64 lanes x 8); CV_WAVE_POS = 496 of them useful
#define CX_ROWB (CX_TW * 2 + 32)                  // bytes per LDS row of the d(dbc) tile and of the staging tile
#define CX_MMAX 64

struct ConvXprojSets {
    cad_conv_xproj_bwd_args s[CV_MAXSETS];
};

template <int NSETS>
__global__ __launch_bounds__(64 * CX_WAVES, CX_WAVES / 4) void conv_xproj_bwd_kernel(ConvXprojSets sets) {
    typedef bf16_t T;
    CAD_DYN_SMEM(char, smem);
    const cad_conv_xproj_bwd_args& a0 = sets.s[0];
    const int lane = threadIdx.x & 63;
    const int wave = cad_uniform(threadIdx.x >> 6);
    const int g = lane >> 4, jl = lane & 15;
    const int64_t L = a0.L, SB = a0.SB;
    const int M = a0.M, MB = (M + 15) >> 4;
    const int cb0 = blockIdx.x * CX_CB;           // first channel of this workgroup
    const uint32_t tpr = (uint32_t)((L + CV_WAVE_POS - 1) / CV_WAVE_POS);
    const uint32_t nvt = (uint32_t)SB * tpr;
    const int mrows = MB * 16;                    // rows of the d(dbc) tile (rows >= M: never read as data)
    char* dtile[CV_MAXSETS];
    dtile[0] = smem;
    dtile[1] = smem + (size_t)mrows * CX_ROWB;
    char* stage = smem + (size_t)NSETS * mrows * CX_ROWB;  // [CX_CB channels][CX_ROWB]
    const float useful = (lane >= 1 && lane <= 62) ? 1.f : 0.f;

    // (positions as 32-bit integers: L < 2^31 is part of cad_conv_xproj_bwd_supported; 64-bit lane positions cost the registers that
    // decide between 128 VGPRs and spills)
    auto tile_pos = [&](uint32_t v, int& sb, int& l_first) {  // row and first LOADED position of virtual tile v
        const uint32_t r = v / tpr;
        sb = (int)r;
        l_first = (int)(v - r * tpr) * CV_WAVE_POS - CV_VEC;
    };
    // LDS-DMA of the d(dbc) tile of set s for virtual tile v: row m by wave (m % 8), lane = 16-byte piece (8 positions); pieces outside
    // the row re-read valid data from a clamped address (their products are masked in the conv stage, their xc is zero)
    auto issue_dtile = [&](int s, uint32_t v) {
        int sb, lf;
        tile_pos(v, sb, lf);
        int l = lf + lane * CV_VEC;
        l = l < 0 ? 0 : l;
        l = l + CV_VEC <= (int)L ? l : (int)L - CV_VEC;
        const T* base = (const T*)sets.s[s].ddbc + (int64_t)sb * L + l;
        for (int m = wave; m < M; m += CX_WAVES)  // wave-uniform
            cad_glds16(base + (int64_t)m * sets.s[s].ld_ddbc, cad_uniform((int)(cad_lds_off(dtile[s]) + m * CX_ROWB)));
    };
    const uint32_t v0 = blockIdx.y, vstep = gridDim.y;
    if (v0 >= nvt) return;  // (the launcher never starts such a workgroup)
#pragma unroll
    for (int s = 0; s < NSETS; ++s) issue_dtile(s, v0);

    // conv weight / bias gradients: every (channel, set) step sums its five per-lane partials (dw4[0..3], dbias) over the wave at once and
    // adds them to lane (5 (c * NSETS + s) + m) of ONE register -- 20 per-lane accumulators would not fit next to the rest at 128 VGPRs
    float pacc = 0.f;
    // dW_x accumulators: the (channel block, m block) tiles of the (32, M) result are dealt to the waves -- wave j owns tile
    // (cbk, mb) = (j / MB, j % MB) over ALL 512 positions of every tile (2 MB <= 8 jobs), so a wave carries ONE accumulator tile per
    // set and nothing has to be exchanged at the end
    f32x4 dwx[NSETS];
#pragma unroll
    for (int s = 0; s < NSETS; ++s) dwx[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool wx_job = wave < CX_NCBK * MB;      // wave-uniform
    const int wx_cbk = wx_job ? wave / MB : 0, wx_mb = wx_job ? wave % MB : 0;
    int wx_mr = wx_mb * 16 + jl;
    wx_mr = wx_mr < M ? wx_mr : M - 1;            // rows >= M: valid data, their columns are never stored

    // x and du of the NEXT (tile, set, channel) step are loaded one step ahead, raw (the conv arithmetic of a step covers the HBM
    // latency of the next one's 32 bytes per lane; without it every step opened with an exposed round trip -- two waves per SIMD
    // that meet at barriers cannot hide it)
    CvRaw<T> nrx, ndu;
    auto fetch = [&](uint32_t v, int s, int c) {
        int sb, lf;
        tile_pos(v, sb, lf);
        const int64_t row = (int64_t)(cb0 + wave * CX_CPW + c) * SB + sb;
        const int l0 = lf + lane * CV_VEC;
        nrx = load8_raw<T, true>((const T*)a0.x + row * L, l0, L);
        ndu = load8_raw<T, true>((const T*)sets.s[s].du + row * L, l0, L);
    };
    fetch(v0, 0, 0);
    for (uint32_t v = v0; v < nvt; v += vstep) {
        int sb, lf;
        tile_pos(v, sb, lf);
        const int l0 = lf + lane * CV_VEC;                 // this lane's 8 positions
        const bool inside = l0 >= 0 && l0 < (int)L;             // (L % 8 == 0: a vector lies inside or outside)
        float o[CX_CPW][CV_VEC];                           // dx of this wave's channels, summed over the sets
#pragma unroll
        for (int c = 0; c < CX_CPW; ++c)
#pragma unroll
            for (int j = 0; j < CV_VEC; ++j) o[c][j] = 0.f;
#pragma unroll
        for (int s = 0; s < NSETS; ++s) {
            const cad_conv_xproj_bwd_args& a = sets.s[s];
            const int rev = sb < a.split ? a.rev_lo : a.rev_hi;  // wave-uniform: the tile's row
            // (1) this set's d(dbc) tile has landed: this wave's share, then everybody's (the barrier also orders the previous
            // user of the staging tile -- the dW_x stage of the set before -- ahead of the P product below)
            gp_wait_dma_conv();
            __syncthreads();
            // (2) P = W_x^T . d(dbc): wave w takes 32 / CX_WAVES of the 16-position sub-blocks, both 16-channel blocks
            {
                u32x4 wfr[CX_NCBK][2];  // B fragments: channel cb0 + 16 cbk + jl, k = 32 ks + 8 g .. + 7 of W_x^T (zero beyond M)
                // (re-read from L2 per tile and set: the loads are loop-invariant, and hoisted out of the tile loop they would occupy 32
                // VGPRs for the whole kernel -- the pointer is made opaque so that they stay here)
                const T* wxp = (const T*)a.wxT + (int64_t)(cb0 + jl) * a.ldw;
#ifndef CAD_EMU
                asm volatile("" : "+v"(wxp));
#endif
#pragma unroll
                for (int cbk = 0; cbk < CX_NCBK; ++cbk)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int k0 = ks * 32 + g * 8;
                        u32x4 w4 = {0u, 0u, 0u, 0u};
                        if (k0 + 8 <= M) w4 = *(const u32x4*)(wxp + (int64_t)(cbk * 16) * a.ldw + k0);
                        wfr[cbk][ks] = w4;
                    }
#pragma unroll
                for (int qq = 0; qq < 32 / CX_WAVES; ++qq) {
                    const int q = wave * (32 / CX_WAVES) + qq;
                    u32x4 xf[2];
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int k0 = ks * 32 + g * 8;
                        // rows k0 .. k0 + 7 exist for k0 + 8 <= M; the other 16-lane groups read rows 0 .. 7 instead (every lane of the
                        // wave takes part in a transposing read) and supply zeros
                        const bool have = k0 + 8 <= M;
                        const int r0 = (have ? k0 : 0) + (jl >> 2);
                        const char* p0 = dtile[s] + r0 * CX_ROWB + q * 32 + (jl & 3) * 8;
                        const u32x2 lo = cad_lds_read_tr16(p0), hi = cad_lds_read_tr16(p0 + 4 * CX_ROWB);
                        const uint32_t km = have ? 0xFFFFFFFFu : 0u;
                        xf[ks] = u32x4{lo[0] & km, lo[1] & km, hi[0] & km, hi[1] & km};
                    }
#pragma unroll
                    for (int cbk = 0; cbk < CX_NCBK; ++cbk) {
                        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks) d = cad_mfma_16x16x32_bf16(xf[ks], wfr[cbk][ks], d);
                        u32x2 pk;
                        pk[0] = cad_pack_bf16x2_safe(d[0], d[1]);
                        pk[1] = cad_pack_bf16x2_safe(d[2], d[3]);
                        *(u32x2*)(stage + (cbk * 16 + jl) * CX_ROWB + (q * 16 + g * 4) * 2) = pk;
                    }
                }
            }
            __syncthreads();  // the whole P tile is in the staging tile
            // (3) conv backward of this wave's channels on d(xc) = bf16(du + P); xc goes back into the staging tile
#pragma unroll
            for (int c = 0; c < CX_CPW; ++c) {
                const int cl = wave * CX_CPW + c, e = cb0 + cl;
                const int64_t row = (int64_t)e * SB + sb;
                float W4[CV_KMAX];
                load_w4(a.w, e, a.K, W4);
                const float bias = a.bias ? a.bias[e] : 0.f;
                const CvRaw<T> rx = nrx, rdu = ndu;
                if (c + 1 < CX_CPW)
                    fetch(v, s, c + 1);
                else if (s + 1 < NSETS)
                    fetch(v, s + 1, 0);        // stays in flight across the barriers and the matrix-core stages in between
                else if (v + vstep < nvt)
                    fetch(v + vstep, 0, 0);
                char* srow = stage + cl * CX_ROWB + lane * 16;
                const u32x4 pv = *(const u32x4*)srow;
                float own[CV_VEC], xe[CV_VEC + 2 * CV_HALO], gg[CV_VEC], dpre[CV_VEC], dpe[CV_VEC + 2 * CV_HALO];
                u32x4 xo = {0u, 0u, 0u, 0u};  // xc of this lane's 8 positions, packed pair by pair as it is produced
                cvt8<T, true>(rx, l0, L, own);
#pragma unroll
                for (int j = 0; j < CV_VEC / 2; ++j) {  // bf16(du + bf16 product), element-wise on the packed pairs (as cad_proj_wx)
                    const float lo = cad_bits2f(pv[j] << 16) + cad_bits2f(rdu.w[j] << 16);
                    const float hi = cad_bits2f(pv[j] & 0xFFFF0000u) + cad_bits2f(rdu.w[j] & 0xFFFF0000u);
                    const uint32_t pk = cad_pack_bf16x2(lo, hi);
                    gg[2 * j] = inside ? cad_bits2f(pk << 16) : 0.f;
                    gg[2 * j + 1] = inside ? cad_bits2f(pk & 0xFFFF0000u) : 0.f;
                }
                halo_window(own, xe);
                float p5[CV_KMAX + 1] = {0.f, 0.f, 0.f, 0.f, 0.f};
                auto body = [&](auto dir) {
                    constexpr int REV = decltype(dir)::value;
#pragma unroll
                    for (int j = 0; j < CV_VEC; j += 2) {
                        float xcp[2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const float acc = conv4<REV>(W4, xe + j + h, bias);
                            const float sg = cad_sigmoid(acc);
                            xcp[h] = acc * sg;                          // = the forward's xc (same operations, same rounding below)
                            dpre[j + h] = gg[j + h] * sg * (1.f + acc * (1.f - sg));
                        }
                        xo[j >> 1] = cad_pack_bf16x2(xcp[0], xcp[1]);
                    }
                    halo_window(dpre, dpe);
#pragma unroll
                    for (int j = 0; j < CV_VEC; ++j) {
                        o[c][j] += conv4<!REV>(W4, dpe + j, 0.f);
                        const float dm = dpre[j] * useful;
#pragma unroll
                        for (int k = 0; k < CV_KMAX; ++k) p5[k] += dm * xe[j + (REV ? (CV_TAPS - 1 - k) : k)];
                        p5[CV_KMAX] += dm;
                    }
                };
                if (rev) body(DirTag<1>{}); else body(DirTag<0>{});
#pragma unroll
                for (int m = 0; m <= CV_KMAX; ++m) {
                    const float t = wave_sum_conv(p5[m]);
                    if (lane == 5 * (c * NSETS + s) + m) pacc += t;
                }
                const uint32_t keep = (inside && useful != 0.f) ? 0xFFFFFFFFu : 0u;  // halo lanes / positions outside the row: nothing to dW_x
                xo[0] &= keep, xo[1] &= keep, xo[2] &= keep, xo[3] &= keep;
                *(u32x4*)srow = xo;
                if (s == NSETS - 1 && useful != 0.f) store8v<T, true>((T*)a0.dx + row * L, l0, L, o[c]);
            }
            __syncthreads();  // xc of all 32 channels is in the staging tile
            // (4) dW_x[channel][m] += xc . d(dbc)^T: this wave's (channel block, m block) tile over the 512 positions (16 k steps)
            if (wx_job) {
#pragma unroll 4
                for (int ks = 0; ks < CX_TW / 32; ++ks) {
                    const int tb = (ks * 32 + g * 8) * 2;  // byte offset of this lane's 8 positions inside a tile row
                    const u32x4 af = *(const u32x4*)(stage + (wx_cbk * 16 + jl) * CX_ROWB + tb);
                    const u32x4 bfr = *(const u32x4*)(dtile[s] + wx_mr * CX_ROWB + tb);
                    dwx[s] = cad_mfma_16x16x32_bf16(af, bfr, dwx[s]);
                }
            }
            // this set's d(dbc) tile and the staging tile are free once every wave is past (4): the next tile's DMA goes out behind
            // a barrier, under the other set's / the next tile's arithmetic
            __syncthreads();
            if (v + vstep < nvt) issue_dtile(s, v + vstep);
        }
    }
    // ---- per-channel conv gradients: lane (5 (c * NSETS + s) + m) holds the sum over all tiles; one atomic each (as conv1d_bwd_kernel) ----
    if (lane < 5 * CX_CPW * NSETS && pacc != 0.f) {
        const int m = lane % 5, cs = lane / 5;
        const int c = cs / NSETS, s_ = cs % NSETS;
        const int e = cb0 + wave * CX_CPW + c;
        const cad_conv_xproj_bwd_args& a = sets.s[s_];
        if (m < CV_KMAX) {
            const int k = m - (CV_KMAX - a.K);  // w[k] lives in w4[k + 4 - K]
            if (k >= 0) atomicAdd(&a.dw[e * a.K + k], pacc);
        } else if (a.dbias) {
            atomicAdd(&a.dbias[e], pacc);
        }
    }
    // ---- dW_x: D layout = lane (column m = 16 mb + jl, rows = channels 16 cbk + 4 g + r); one (32, M) block of this token group's slot
    if (wx_job) {
        const int m = wx_mb * 16 + jl;
        if (m < M) {
#pragma unroll
            for (int s = 0; s < NSETS; ++s) {
                float* slot = sets.s[s].dwx_partials + ((int64_t)blockIdx.y * sets.s[s].E + cb0 + wx_cbk * 16 + g * 4) * M + m;
#pragma unroll
                for (int r = 0; r < 4; ++r) slot[(int64_t)r * M] = dwx[s][r];
            }
        }
    }
}

}  // namespace

extern "C" int cad_conv1d_fwd_multi(const cad_conv1d_args* sets, int nsets, void* stream) {
    CAD_CHECK_ARG(sets && nsets >= 1 && nsets <= CV_MAXSETS);
    ConvFwdSets ks;
    for (int i = 0; i < nsets; ++i) {
        const cad_conv1d_args* a = &sets[i];
        CAD_CHECK_ARG(a->x && a->w && a->out);
        CAD_CHECK_ARG(a->E > 0 && a->SB > 0 && a->L > 0 && a->K >= 1 && a->K <= CV_KMAX);
        CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB);
        CAD_CHECK_ARG((int64_t)a->E * a->SB < (1LL << 31));
        CAD_CHECK_ARG(a->x == sets[0].x && a->E == sets[0].E && a->SB == sets[0].SB && a->L == sets[0].L &&
                      a->dtype == sets[0].dtype);
        ks.s[i] = *a;
    }
    for (int i = nsets; i < CV_MAXSETS; ++i) ks.s[i] = sets[0];
    const cad_conv1d_args* a = &sets[0];
    CadProfScope prof(2, stream);
    const int64_t nvt = a->SB * ((a->L + CV_WAVE_POS - 1) / CV_WAVE_POS);  // virtual tiles of one channel
    if (nvt >= (1LL << 31)) return CAD_ERR_UNSUPPORTED;
    const int64_t per_block = (int64_t)CV_WAVES * CV_TILES_FWD;
    dim3 grid((unsigned)a->E, (unsigned)((nvt + per_block - 1) / per_block)), block(CV_THREADS);
    if (grid.y > 65535u) return CAD_ERR_UNSUPPORTED;
    const size_t vb = (a->dtype == CAD_F32 ? 4 : 2) * CV_VEC;
    uintptr_t ptrs = (uintptr_t)a->x;
    for (int i = 0; i < nsets; ++i) ptrs |= (uintptr_t)sets[i].out;
    const bool vec = (a->L % CV_VEC) == 0 && (ptrs % vb) == 0;
#define CV_FWD(T, NS)                                                                   \
    do {                                                                                \
        if (vec)                                                                        \
            CAD_LAUNCH((conv1d_fwd_kernel<T, NS, true>), grid, block, 0, stream, ks);   \
        else                                                                            \
            CAD_LAUNCH((conv1d_fwd_kernel<T, NS, false>), grid, block, 0, stream, ks);  \
    } while (0)
    if (a->dtype == CAD_F32) {
        if (nsets == 1) CV_FWD(float, 1); else CV_FWD(float, 2);
    } else if (a->dtype == CAD_BF16) {
        if (nsets == 1) CV_FWD(bf16_t, 1); else CV_FWD(bf16_t, 2);
    } else {
        return CAD_ERR_UNSUPPORTED;
    }
#undef CV_FWD
    return cad_after_launch();
}

extern "C" int cad_conv1d_fwd(const cad_conv1d_args* a, void* stream) { return cad_conv1d_fwd_multi(a, 1, stream); }

extern "C" int cad_conv1d_bwd_multi(const cad_conv1d_bwd_args* sets, int nsets, void* stream) {
    CAD_CHECK_ARG(sets && nsets >= 1 && nsets <= CV_MAXSETS);
    ConvBwdSets ks;
    for (int i = 0; i < nsets; ++i) {
        const cad_conv1d_bwd_args* a = &sets[i];
        CAD_CHECK_ARG(a->x && a->w && a->dout && a->dx && a->dw);
        CAD_CHECK_ARG(a->E > 0 && a->SB > 0 && a->L > 0 && a->K >= 1 && a->K <= CV_KMAX);
        CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB);
        CAD_CHECK_ARG(a->x == sets[0].x && a->dx == sets[0].dx && a->accumulate == sets[0].accumulate &&
                      a->E == sets[0].E && a->SB == sets[0].SB && a->L == sets[0].L && a->dtype == sets[0].dtype);
        ks.s[i] = *a;
    }
    for (int i = nsets; i < CV_MAXSETS; ++i) ks.s[i] = sets[0];
    const cad_conv1d_bwd_args* a = &sets[0];
    CadProfScope prof(3, stream);
    const int64_t nvt = a->SB * ((a->L + CV_WAVE_POS - 1) / CV_WAVE_POS);
    if (nvt >= (1LL << 31)) return CAD_ERR_UNSUPPORTED;
    const int64_t per_block = (int64_t)CV_WAVES * CV_TILES_BWD;
    dim3 grid((unsigned)a->E, (unsigned)((nvt + per_block - 1) / per_block)), block(CV_THREADS);
    if (grid.y > 65535u) return CAD_ERR_UNSUPPORTED;
    const size_t vb = (a->dtype == CAD_F32 ? 4 : 2) * CV_VEC;
    uintptr_t ptrs = (uintptr_t)a->x | (uintptr_t)a->dx;
    for (int i = 0; i < nsets; ++i) ptrs |= (uintptr_t)sets[i].dout;
    const bool vec = (a->L % CV_VEC) == 0 && (ptrs % vb) == 0;
#define CV_BWD(T, NS)                                                                   \
    do {                                                                                \
        if (vec && a->accumulate)                                                              \
            CAD_LAUNCH((conv1d_bwd_kernel<T, NS, true, true>), grid, block, 0, stream, ks);    \
        else if (vec)                                                                          \
            CAD_LAUNCH((conv1d_bwd_kernel<T, NS, true, false>), grid, block, 0, stream, ks);   \
        else if (a->accumulate)                                                                \
            CAD_LAUNCH((conv1d_bwd_kernel<T, NS, false, true>), grid, block, 0, stream, ks);   \
        else                                                                                   \
            CAD_LAUNCH((conv1d_bwd_kernel<T, NS, false, false>), grid, block, 0, stream, ks);  \
    } while (0)
    if (a->dtype == CAD_F32) {
        if (nsets == 1) CV_BWD(float, 1); else CV_BWD(float, 2);
    } else if (a->dtype == CAD_BF16) {
        if (nsets == 1) CV_BWD(bf16_t, 1); else CV_BWD(bf16_t, 2);
    } else {
        return CAD_ERR_UNSUPPORTED;
    }
#undef CV_BWD
    return cad_after_launch();
}

extern "C" int cad_conv1d_bwd(const cad_conv1d_bwd_args* a, void* stream) { return cad_conv1d_bwd_multi(a, 1, stream); }


// ---- fused backward: launcher -----------------------------------------------------------------------------------------------
extern "C" int cad_conv_xproj_bwd_supported(int E, int K, int M, int64_t SB, int64_t L) {
    // two d(dbc) tiles of ceil16(M) rows + the 32-channel staging tile must fit the 160 KB of LDS: M <= 48 (d_model <= 256 with
    // d_state 16); a d_model 512 layer (dt_rank 32: M = 64) takes the three-kernel path
    const int mrows = (M + 15) / 16 * 16;
    return E > 0 && (E % CX_CB) == 0 && K >= 1 && K <= CV_KMAX && M >= 8 && M <= CX_MMAX && (M % 8) == 0 &&
           (size_t)(CV_MAXSETS * mrows + CX_CB) * CX_ROWB <= 160 * 1024 && SB >= 1 && L >= CV_VEC &&
           (L % CV_VEC) == 0 && L < (1LL << 30) && SB * ((L + CV_WAVE_POS - 1) / CV_WAVE_POS) < (1LL << 31);
}
// token groups = partial slots of dW_x: enough workgroups for every CU, never more than virtual tiles
extern "C" int cad_conv_xproj_bwd_partials(int E, int64_t SB, int64_t L) {
    const int64_t nvt = SB * ((L + CV_WAVE_POS - 1) / CV_WAVE_POS);
    const int cbs = E / CX_CB > 0 ? E / CX_CB : 1;
    int64_t gy = (256 + cbs - 1) / cbs;
    if (gy > nvt) gy = nvt;
    if (gy < 1) gy = 1;
    return (int)gy;
}

extern "C" int cad_conv_xproj_bwd_multi(const cad_conv_xproj_bwd_args* sets, int nsets, void* stream) {
    CAD_CHECK_ARG(sets && nsets >= 1 && nsets <= CV_MAXSETS);
    ConvXprojSets ks;
    for (int i = 0; i < nsets; ++i) {
        const cad_conv_xproj_bwd_args* a = &sets[i];
        CAD_CHECK_ARG(a->x && a->w && a->du && a->ddbc && a->wxT && a->dx && a->dw && a->dwx_partials);
        CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB);
        CAD_CHECK_ARG(a->x == sets[0].x && a->dx == sets[0].dx && a->E == sets[0].E && a->SB == sets[0].SB && a->L == sets[0].L &&
                      a->M == sets[0].M && a->dtype == sets[0].dtype);
        CAD_CHECK_ARG(a->ld_ddbc >= a->SB * a->L && (a->ld_ddbc % 8) == 0 && a->ldw >= a->M && (a->ldw % 8) == 0);
        CAD_CHECK_ARG((((uintptr_t)a->x | (uintptr_t)a->du | (uintptr_t)a->ddbc | (uintptr_t)a->wxT | (uintptr_t)a->dx) % 16) == 0);
        ks.s[i] = *a;
    }
    for (int i = nsets; i < CV_MAXSETS; ++i) ks.s[i] = sets[0];
    const cad_conv_xproj_bwd_args* a = &sets[0];
    if (a->dtype != CAD_BF16 || !cad_conv_xproj_bwd_supported(a->E, a->K, a->M, a->SB, a->L)) return CAD_ERR_UNSUPPORTED;
    for (int i = 0; i < nsets; ++i)
        if (sets[i].K < 1 || sets[i].K > CV_KMAX) return CAD_ERR_UNSUPPORTED;
    CadProfScope prof(3, stream);
    const int mrows = (a->M + 15) / 16 * 16;
    const size_t lds = (size_t)nsets * mrows * CX_ROWB + (size_t)CX_CB * CX_ROWB;
    dim3 grid((unsigned)(a->E / CX_CB), (unsigned)cad_conv_xproj_bwd_partials(a->E, a->SB, a->L)), block(64 * CX_WAVES);
#if !defined(CAD_EMU)
#define CX_BIG_LDS(kern)                                                                                                     \
    do {                                                                                                                     \
        static size_t cur[CAD_MAX_DEVICES] = {0};                                                                            \
        int dev_ = 0;                                                                                                        \
        if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= CAD_MAX_DEVICES) return CAD_ERR_LAUNCH;                 \
        if (lds > 65536 && lds > cur[dev_]) {                                                                                \
            if (hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                return CAD_ERR_LAUNCH;                                                                                       \
            cur[dev_] = lds;                                                                                                 \
        }                                                                                                                    \
    } while (0)
#else
#define CX_BIG_LDS(kern) (void)0
#endif
    if (nsets == 1) {
        CX_BIG_LDS((conv_xproj_bwd_kernel<1>));
        CAD_LAUNCH((conv_xproj_bwd_kernel<1>), grid, block, lds, stream, ks);
    } else {
        CX_BIG_LDS((conv_xproj_bwd_kernel<2>));
        CAD_LAUNCH((conv_xproj_bwd_kernel<2>), grid, block, lds, stream, ks);
    }
#undef CX_BIG_LDS
    return cad_after_launch();
}
