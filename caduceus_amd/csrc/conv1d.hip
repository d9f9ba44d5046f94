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
            cad_order_point(own[0]);
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

