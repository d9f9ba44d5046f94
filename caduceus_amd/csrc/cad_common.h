// Shared device/host helpers for the Caduceus gfx950 kernels.
//
// The kernels are written for CDNA4 (wave64, LDS, HBM-coalesced channel-major streams).  The only portability
// seam is CAD_EMU: the test-suite compiles these same sources with g++ against tests/emu/emu_runtime.h so that
// kernel logic can be parity-checked against the oracle on a machine without a GPU.  There is no CUDA path.
#pragma once
#include <stdint.h>

#include "../../include/caduceus_hip.h"

#ifdef CAD_EMU
#include "emu_runtime.h"
#define CAD_LAUNCH(kern, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [=]() { kern(__VA_ARGS__); })
#define CAD_DEVICE_BUILD 0
#else
#include <hip/hip_runtime.h>
#define CAD_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), (shmem), (hipStream_t)(stream), __VA_ARGS__)
#define CAD_DEVICE_BUILD 1
#endif

#define CAD_WAVE 64
#define CAD_MAX_DEVICES 64   // per-device launch-attribute caches (GP_BIG_LDS / SC_BIG_LDS)

// dynamic LDS (16-byte aligned base; keep ALL of a kernel's LDS in this one region - guide G17)
#ifdef CAD_EMU
#define CAD_DYN_SMEM(T, name) T* name = (T*)emu::dyn_smem()
#else
#define CAD_DYN_SMEM(T, name)                                              \
    extern __shared__ __attribute__((aligned(16))) char cad_smem_raw[];   \
    T* name = (T*)cad_smem_raw
#endif

// ---- small numeric helpers ---------------------------------------------------------------------------------
typedef float f32x2 __attribute__((vector_size(8)));  // maps to v_pk_{mul,fma,add}_f32 on gfx950
typedef float f32x4 __attribute__((vector_size(16)));
typedef uint32_t u32x2 __attribute__((vector_size(8)));

struct bf16_t {
    uint16_t v;
};

__device__ __forceinline__ float cad_bits2f(uint32_t u) {
    union {
        uint32_t u;
        float f;
    } c;
    c.u = u;
    return c.f;
}
__device__ __forceinline__ uint32_t cad_f2bits(float f) {
    union {
        uint32_t u;
        float f;
    } c;
    c.f = f;
    return c.u;
}
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return cad_bits2f((uint32_t)x.v << 16); }
template <typename T>
__device__ __forceinline__ T from_f32(float f);
template <>
__device__ __forceinline__ float from_f32<float>(float f) {
    return f;
}
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) {  // round-to-nearest-even, NaN preserved
    uint32_t u = cad_f2bits(f);
    bf16_t r;
    if ((u & 0x7fffffffu) > 0x7f800000u) {
        r.v = (uint16_t)((u >> 16) | 0x40);
    } else {
        u += 0x7fffu + ((u >> 16) & 1u);
        r.v = (uint16_t)(u >> 16);
    }
    return r;
}

// two fp32 -> packed bf16x2 (lo in bits [15:0]); v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even)
__device__ __forceinline__ uint32_t cad_pack_bf16x2(float lo, float hi) {
#ifdef CAD_EMU
    return (uint32_t)from_f32<bf16_t>(lo).v | ((uint32_t)from_f32<bf16_t>(hi).v << 16);
#else
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
#endif
}
// N fp32 values -> N contiguous elements of T at dst (N even, dst suitably aligned by the caller's vector type)
template <typename T, int N>
__device__ __forceinline__ void cad_cvt_store(T* dst, const float* v);
template <>
__device__ __forceinline__ void cad_cvt_store<float, 4>(float* dst, const float* v) {
    struct __attribute__((aligned(16))) V { float f[4]; } t = {{v[0], v[1], v[2], v[3]}};
    *(V*)dst = t;
}
template <>
__device__ __forceinline__ void cad_cvt_store<bf16_t, 4>(bf16_t* dst, const float* v) {
    struct __attribute__((aligned(8))) V { uint32_t w[2]; } t = {{cad_pack_bf16x2(v[0], v[1]), cad_pack_bf16x2(v[2], v[3])}};
    *(V*)dst = t;
}

template <>
__device__ __forceinline__ void cad_cvt_store<float, 2>(float* dst, const float* v) {
    struct __attribute__((aligned(8))) V { float f[2]; } t = {{v[0], v[1]}};
    *(V*)dst = t;
}
template <>
__device__ __forceinline__ void cad_cvt_store<bf16_t, 2>(bf16_t* dst, const float* v) {
    *(uint32_t*)dst = cad_pack_bf16x2(v[0], v[1]);
}

#define CAD_LOG2E 1.4426950408889634f

__device__ __forceinline__ float cad_exp2(float x) {
#ifdef CAD_EMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);  // v_exp_f32
#endif
}
__device__ __forceinline__ float cad_exp(float x) { return cad_exp2(x * CAD_LOG2E); }
__device__ __forceinline__ float cad_log(float x) {
#ifdef CAD_EMU
    return logf(x);
#else
    return __builtin_amdgcn_logf(x) * 0.6931471805599453f;  // v_log_f32 (log2) * ln2
#endif
}
__device__ __forceinline__ float cad_rcp(float x) {
#ifdef CAD_EMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}
__device__ __forceinline__ float cad_rsqrt(float x) {
#ifdef CAD_EMU
    return 1.0f / sqrtf(x);
#else
    return __builtin_amdgcn_rsqf(x);
#endif
}
// softplus with the upstream threshold (x > 20 -> x); log1p(e) evaluated as log(w) * e / (w - 1), w = 1 + e,
// which is accurate to ~1 ulp also for tiny e (plain log(1+e) loses all digits there).
// Branch-free (selects only): every lane evaluates both sides; NaN/inf of the unselected side is discarded.
__device__ __forceinline__ float cad_softplus(float x) {
    const float e = cad_exp(x);
    const float w = 1.0f + e;
    const float d = w - 1.0f;
    const float lp = cad_log(w) * (e * cad_rcp(d));
    const float sp = (d == 0.0f) ? e : lp;
    return (x > 20.0f) ? x : sp;
}
// softplus for results that are rounded to bf16 anyway: max(x, 0) + log(1 + exp(-|x|)) -- two transcendentals instead of
// three; below t = exp(-|x|) ~ 1e-4 the plain log(1 + t) keeps only 3-4 digits of a term that is < 1e-4 in absolute value
// (and is replaced by t itself below 2^-12), far inside bf16's 8 bits
__device__ __forceinline__ float cad_softplus_lowp(float x) {
    const float t = cad_exp(-fabsf(x));
    const float l = t < 0.000244140625f ? t : cad_log(1.0f + t);
    return fmaxf(x, 0.0f) + l;
}
__device__ __forceinline__ float cad_sigmoid(float x) { return cad_rcp(1.0f + cad_exp(-x)); }
// sigmoid(x) recovered from sp = softplus(x):  1 - exp(-sp)  (series below 1/16: no cancellation for tiny sp)
__device__ __forceinline__ float cad_sigmoid_from_softplus(float sp) {
    const float poly = sp * (1.0f - sp * (0.5f - sp * (0.16666667f - sp * 0.041666668f)));
    const float big = 1.0f - cad_exp(-sp);
    return sp < 0.0625f ? poly : big;
}

// ---- cross-lane primitives (DPP on gfx950; emulated through the fiber exchange in the test build) -----------------
// Each returns, per lane, the value of `v` in the source lane selected by the pattern, or `old` where the pattern has
// no source for this lane -- exactly v_mov_b32_dpp with bound_ctrl:0.  Passing the identity element as `old` lets a
// scan step run unconditionally on all lanes.
#ifdef CAD_EMU
template <int N>
__device__ __forceinline__ float dpp_row_shr(float old, float v) {
    const int lane = emu::lane_id();
    const bool ok = (lane & 15) >= N;
    const float r = emu_exchange(v, ok ? lane - N : lane);
    return ok ? r : old;
}
template <int N>
__device__ __forceinline__ float dpp_row_shl(float old, float v) {
    const int lane = emu::lane_id();
    const bool ok = (lane & 15) + N < 16;
    const float r = emu_exchange(v, ok ? lane + N : lane);
    return ok ? r : old;
}
__device__ __forceinline__ float dpp_row_bcast15(float old, float v) {  // rows 1 and 3 <- lane 15 of the previous row
    const int lane = emu::lane_id();
    const bool ok = ((lane >> 4) & 1) == 1;
    const float r = emu_exchange(v, ok ? (lane & ~15) - 1 : lane);
    return ok ? r : old;
}
__device__ __forceinline__ float dpp_row_bcast31(float old, float v) {  // rows 2 and 3 <- lane 31
    const int lane = emu::lane_id();
    const bool ok = lane >= 32;
    const float r = emu_exchange(v, ok ? 31 : lane);
    return ok ? r : old;
}
__device__ __forceinline__ float dpp_wave_shr1(float old, float v) {
    const int lane = emu::lane_id();
    const float r = emu_exchange(v, lane >= 1 ? lane - 1 : lane);
    return lane >= 1 ? r : old;
}
__device__ __forceinline__ float dpp_wave_shl1(float old, float v) {
    const int lane = emu::lane_id();
    const float r = emu_exchange(v, lane < 63 ? lane + 1 : lane);
    return lane < 63 ? r : old;
}
__device__ __forceinline__ float cad_readlane(float v, int l) { return emu_exchange(v, l); }
__device__ __forceinline__ int cad_uniform(int v) { return v; }
#else
#define CAD_DPP(old, v, ctrl, rmask)                                                                          \
    __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)),              \
                                                          __builtin_bit_cast(int, (float)(v)), (ctrl), (rmask), 0xf, false))
template <int N>
__device__ __forceinline__ float dpp_row_shr(float old, float v) {
    return CAD_DPP(old, v, 0x110 + N, 0xf);
}
template <int N>
__device__ __forceinline__ float dpp_row_shl(float old, float v) {
    return CAD_DPP(old, v, 0x100 + N, 0xf);
}
__device__ __forceinline__ float dpp_row_bcast15(float old, float v) { return CAD_DPP(old, v, 0x142, 0xa); }
__device__ __forceinline__ float dpp_row_bcast31(float old, float v) { return CAD_DPP(old, v, 0x143, 0xc); }
__device__ __forceinline__ float dpp_wave_shr1(float old, float v) { return CAD_DPP(old, v, 0x138, 0xf); }
__device__ __forceinline__ float dpp_wave_shl1(float old, float v) { return CAD_DPP(old, v, 0x130, 0xf); }
__device__ __forceinline__ float cad_readlane(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// tell the compiler a value is wave-uniform (e.g. the wave index threadIdx.x >> 6): everything derived from it -- row
// base pointers, channel parameters -- then lives in SGPRs instead of VGPRs
__device__ __forceinline__ int cad_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// ---- matrix core (MFMA) ------------------------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16:  D (16 x 16 fp32) = A (16 x 32 bf16) * B (32 x 16 bf16) + C, one tile per wave.
// Operand layouts (lane l, g = l >> 4):  A: row i = l & 15, elements k = 8g .. 8g+7 (4 dwords, element t in dword t >> 1,
// half t & 1);  B: column j = l & 15, elements k = 8g .. 8g+7;  C / D: column j = l & 15, rows 4g + r (r = 0..3).
typedef uint32_t u32x4 __attribute__((vector_size(16)));
#ifdef CAD_EMU
__device__ __forceinline__ f32x4 cad_mfma_16x16x32_bf16(u32x4 a, u32x4 b, f32x4 c) {
    const int lane = emu::lane_id();
    const int col = lane & 15, rg = lane >> 4;
    float bk[32];      // B[k][col]
    float ak[4][32];   // A[4 rg + r][k]
    for (int g = 0; g < 4; ++g) {
        for (int h = 0; h < 2; ++h) {
            const uint64_t mine_b = (uint64_t)b[2 * h] | ((uint64_t)b[2 * h + 1] << 32);
            const uint64_t vb = emu_exchange(mine_b, g * 16 + col);
            for (int t = 0; t < 4; ++t)
                bk[8 * g + 4 * h + t] = cad_bits2f((uint32_t)((vb >> (16 * t)) & 0xffffu) << 16);
            const uint64_t mine_a = (uint64_t)a[2 * h] | ((uint64_t)a[2 * h + 1] << 32);
            for (int r = 0; r < 4; ++r) {
                const uint64_t va = emu_exchange(mine_a, g * 16 + 4 * rg + r);
                for (int t = 0; t < 4; ++t)
                    ak[r][8 * g + 4 * h + t] = cad_bits2f((uint32_t)((va >> (16 * t)) & 0xffffu) << 16);
            }
        }
    }
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        float s = c[r];
        for (int k = 0; k < 32; ++k) s += ak[r][k] * bk[k];
        d[r] = s;
    }
    return d;
}
#else
__device__ __forceinline__ f32x4 cad_mfma_16x16x32_bf16(u32x4 a, u32x4 b, f32x4 c) {
    typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
    typedef float f32x4_hw __attribute__((ext_vector_type(4)));
    const f32x4_hw r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b),
                                                               __builtin_bit_cast(f32x4_hw, c), 0, 0, 0);
    return __builtin_bit_cast(f32x4, r);
}
#endif
// v_mfma_f32_16x16x4_f32:  D (16 x 16 fp32) = A (16 x 4 fp32) * B (4 x 16 fp32) + C.  Lane l, g = l >> 4:  A: row l & 15, k = g;
// B: k = g, column l & 15;  C / D as above (column l & 15, rows 4g + r).  Full fp32 operands: the LM head keeps its fp32 weight.
#ifdef CAD_EMU
__device__ __forceinline__ f32x4 cad_mfma_16x16x4_f32(float a, float b, f32x4 c) {
    const int lane = emu::lane_id();
    const int col = lane & 15, rg = lane >> 4;
    f32x4 d = c;
    for (int k = 0; k < 4; ++k) {
        const float bk = emu_exchange(b, k * 16 + col);
        for (int r = 0; r < 4; ++r) d[r] += emu_exchange(a, k * 16 + 4 * rg + r) * bk;
    }
    return d;
}
#else
__device__ __forceinline__ f32x4 cad_mfma_16x16x4_f32(float a, float b, f32x4 c) {
    typedef float f32x4_hw __attribute__((ext_vector_type(4)));
    const f32x4_hw r = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, __builtin_bit_cast(f32x4_hw, c), 0, 0, 0);
    return __builtin_bit_cast(f32x4, r);
}
#endif
// ds_read_b64_tr_b16: transposing LDS read for 16-bit elements.  Within every 16-lane group, lane 4 r + c (r, c in 0..3)
// supplies the address of 4 contiguous elements S[r][4c .. 4c+3] of a 4 x 16 block S; lane l of the group receives the
// COLUMN  (S[0][l], S[1][l], S[2][l], S[3][l]).  This is how a [k][token] tile (token-contiguous, as the channel-major
// activations are) yields MFMA operand fragments, which want consecutive k per lane.  (Semantics as used by ck_tile's
// transpose loads, /opt/rocm/include/ck_tile/core/tensor/load_tile_transpose.hpp: Quad16 input / output encodings.)
#ifdef CAD_EMU
__device__ __forceinline__ u32x2 cad_lds_read_tr16(const void* p) {
    uint64_t mine;
    std::memcpy(&mine, p, 8);
    const int lane = emu::lane_id();
    const int base = lane & ~15, l = lane & 15;
    uint32_t e[4];
    for (int r = 0; r < 4; ++r) {
        const uint64_t v = emu_exchange(mine, base + 4 * r + (l >> 2));
        e[r] = (uint32_t)((v >> (16 * (l & 3))) & 0xffffu);
    }
    u32x2 out;
    out[0] = e[0] | (e[1] << 16);
    out[1] = e[2] | (e[3] << 16);
    return out;
}
#else
__device__ __forceinline__ u32x2 cad_lds_read_tr16(const void* p) {
    typedef __bf16 bf16x4_hw __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) bf16x4_hw lds_vec_t;
    const bf16x4_hw v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_vec_t*)p);
    return __builtin_bit_cast(u32x2, v);
}
#endif

// two fp32 -> packed bf16x2 through a conversion the COMPILER sees (it emits v_cvt_pk_bf16_f32 and pads the MFMA / DOT
// result hazards itself; the inline-asm cad_pack_bf16x2 is invisible to its hazard recognizer)
__device__ __forceinline__ uint32_t cad_pack_bf16x2_safe(float lo, float hi) {
#ifdef CAD_EMU
    return cad_pack_bf16x2(lo, hi);
#else
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    typedef float f32x2_hw __attribute__((ext_vector_type(2)));
    const f32x2_hw v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
#endif
}

// ---- LDS-DMA: asynchronous 16-byte-per-lane copy global -> LDS (no VGPR involved) ----------------------------------
// global_load_lds_dwordx4: lane l's 16 bytes, read from its own global address, land at (wave-uniform LDS base in M0)
// + 16 l.  Tracked by vmcnt like any load: the issuing wave waits vmcnt before its ds_read (and a barrier before another
// wave's).  A swizzled LDS image is obtained by permuting the per-lane SOURCE addresses; the destination is always linear.
__device__ __forceinline__ uint32_t cad_lds_off(const void* p) {  // byte offset of an LDS object inside the LDS aperture
#ifdef CAD_EMU
    return (uint32_t)((const char*)p - emu::dyn_smem());
#else
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
#endif
}
__device__ __forceinline__ void cad_glds16(const void* gsrc /* per lane */, uint32_t lds_base /* wave-uniform, SGPR */) {
#ifdef CAD_EMU
    std::memcpy(emu::dyn_smem() + lds_base + 16 * emu::lane_id(), gsrc, 16);
#else
    uint32_t keep;  // M0 holds the LDS base of the DMA; it is compiler-reserved, so save / restore it in the same statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
#endif
}

// Point at which every lane of the wave has executed what precedes it.  The hardware runs a wave in lock-step and its LDS
// queue is in order, so this is only a compiler fence on the device; the host emulator runs lanes as separate fibers and
// needs a real rendezvous wherever a lane reads LDS written by ANOTHER lane of its wave without a workgroup barrier.
__device__ __forceinline__ void cad_wave_sync() {
#ifdef CAD_EMU
    emu::wave_sync();
#else
    __builtin_amdgcn_wave_barrier();
#endif
}

// wave-uniform "any lane" vote
__device__ __forceinline__ bool cad_wave_any(bool p) {
#ifdef CAD_EMU
    bool r = false;
    for (int l = 0; l < emu::wave_lanes(); ++l) r = emu_exchange((int)p, l) != 0 || r;
    return r;
#else
    return __builtin_amdgcn_ballot_w64(p) != 0;
#endif
}

// compiler-only fence: keeps the scheduler from hoisting (LDS) loads across this point, which bounds live ranges
__device__ __forceinline__ void cad_sched_fence() {
#ifndef CAD_EMU
    asm volatile("" ::: "memory");
#endif
}

// ---- direction / index maps ----------------------------------------------------------------------------------
// logical position p in [0, L) of a row <-> physical index along L
__device__ __forceinline__ int64_t cad_phys(int64_t p, int64_t L, int rev) { return rev ? (L - 1 - p) : p; }

// ---- host side -------------------------------------------------------------------------------------------------
#define CAD_CHECK_ARG(cond)              \
    do {                                 \
        if (!(cond)) return CAD_ERR_BAD_ARG; \
    } while (0)

int cad_after_launch();  // hipGetLastError -> cad_status

// RAII kernel timer (no-op unless cad_prof_enable(1)); see api.hip
struct CadProfScope {
    int kind;
    void* stream;
    int slot;
    CadProfScope(int kind, void* stream);
    ~CadProfScope();
};
