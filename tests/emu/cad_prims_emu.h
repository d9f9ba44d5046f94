// Host-emulator restatement of the gfx950 primitives in caduceus_amd/csrc/cad_prims_gfx950.h (same names, same per-lane semantics; lanes
// are fibers that rendezvous in emu_exchange / emu::wave_sync).  TEST INFRASTRUCTURE: compiled only with -DCAD_EMU by tests/emu/build_emu.py.
#pragma once
#include "emu_runtime.h"
#define CAD_LAUNCH(kern, grid, block, shmem, stream, ...) \
    emu::launch((grid), (block), (shmem), [=]() { kern(__VA_ARGS__); })
#define CAD_DEVICE_BUILD 0

#define CAD_DYN_SMEM(T, name) T* name = (T*)emu::dyn_smem()

#include "../../caduceus_amd/csrc/cad_types.h"

__device__ __forceinline__ uint32_t cad_pack_bf16x2(float lo, float hi) {
    return (uint32_t)from_f32<bf16_t>(lo).v | ((uint32_t)from_f32<bf16_t>(hi).v << 16);
}

__device__ __forceinline__ float cad_exp2(float x) {
    return exp2f(x);
}

__device__ __forceinline__ float cad_log(float x) {
    return logf(x);
}

__device__ __forceinline__ float cad_rcp(float x) {
    return 1.0f / x;
}

__device__ __forceinline__ float cad_rsqrt(float x) {
    return 1.0f / sqrtf(x);
}

__device__ __forceinline__ uint32_t cad_perm(uint32_t s0, uint32_t s1, uint32_t sel) {  // v_perm_b32 (selectors 0..7 and 0x0c)
    uint32_t r = 0;
    for (int b = 0; b < 4; ++b) {
        const uint32_t k = (sel >> (8 * b)) & 0xffu;
        uint32_t byte = 0;
        if (k < 4)
            byte = (s1 >> (8 * k)) & 0xffu;
        else if (k < 8)
            byte = (s0 >> (8 * (k - 4))) & 0xffu;
        else if (k >= 13)
            byte = 0xffu;
        r |= byte << (8 * b);
    }
    return r;
}

__device__ __forceinline__ float cad_mul_legacy(float a, float b) {  // v_mul_legacy_f32
    return (a == 0.0f || b == 0.0f) ? 0.0f : a * b;
}

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

__device__ __forceinline__ f32x4 cad_mfma_16x16x32_bf16(u32x4 a, u32x4 b, f32x4 c) {
    const int lane = emu::lane_id();
    const int col = lane & 15, rg = lane >> 4;
    float bk[32];      // B[k][col]
    float ak[4][32];   // A[4 rg + r][k]
    for (int h = 0; h < 2; ++h) {
        const uint64_t* pb = emu_publish((uint64_t)b[2 * h] | ((uint64_t)b[2 * h + 1] << 32));
        for (int g = 0; g < 4; ++g) {
            const uint64_t vb = pb[g * 16 + col];
            for (int t = 0; t < 4; ++t) bk[8 * g + 4 * h + t] = cad_bits2f((uint32_t)((vb >> (16 * t)) & 0xffffu) << 16);
        }
        const uint64_t* pa = emu_publish((uint64_t)a[2 * h] | ((uint64_t)a[2 * h + 1] << 32));
        for (int g = 0; g < 4; ++g)
            for (int r = 0; r < 4; ++r) {
                const uint64_t va = pa[g * 16 + 4 * rg + r];
                for (int t = 0; t < 4; ++t) ak[r][8 * g + 4 * h + t] = cad_bits2f((uint32_t)((va >> (16 * t)) & 0xffffu) << 16);
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

__device__ __forceinline__ f32x4 cad_mfma_16x16x4_f32(float a, float b, f32x4 c) {
    const int lane = emu::lane_id();
    const int col = lane & 15, rg = lane >> 4;
    uint32_t ab[2];
    std::memcpy(&ab[0], &a, 4);
    std::memcpy(&ab[1], &b, 4);
    const uint64_t* p = emu_publish((uint64_t)ab[0] | ((uint64_t)ab[1] << 32));  // every lane's (a, b) in one rendezvous
    f32x4 d = c;
    for (int k = 0; k < 4; ++k) {
        const float bk = cad_bits2f((uint32_t)(p[k * 16 + col] >> 32));
        for (int r = 0; r < 4; ++r) d[r] += cad_bits2f((uint32_t)p[k * 16 + 4 * rg + r]) * bk;
    }
    return d;
}

__device__ __forceinline__ u32x2 cad_lds_read_tr16(const void* p) {
    uint64_t mine;
    std::memcpy(&mine, p, 8);
    const int lane = emu::lane_id();
    const int base = lane & ~15, l = lane & 15;
    uint32_t e[4];
    const uint64_t* p4 = emu_publish(mine);
    for (int r = 0; r < 4; ++r) {
        const uint64_t v = p4[base + 4 * r + (l >> 2)];
        e[r] = (uint32_t)((v >> (16 * (l & 3))) & 0xffffu);
    }
    u32x2 out;
    out[0] = e[0] | (e[1] << 16);
    out[1] = e[2] | (e[3] << 16);
    return out;
}

__device__ __forceinline__ uint32_t cad_pack_bf16x2_safe(float lo, float hi) {
    return cad_pack_bf16x2(lo, hi);
}

__device__ __forceinline__ uint32_t cad_lds_off(const void* p) {  // byte offset of an LDS object inside the LDS aperture
    return (uint32_t)((const char*)p - emu::dyn_smem());
}

__device__ __forceinline__ void cad_glds16(const void* gsrc /* per lane */, uint32_t lds_base /* wave-uniform, SGPR */) {
    std::memcpy(emu::dyn_smem() + lds_base + 16 * emu::lane_id(), gsrc, 16);
}

__device__ __forceinline__ void cad_wave_sync() {
    emu::wave_sync();
}

__device__ __forceinline__ bool cad_wave_any(bool p) {
    bool r = false;
    for (int l = 0; l < emu::wave_lanes(); ++l) r = emu_exchange((int)p, l) != 0 || r;
    return r;
}

__device__ __forceinline__ void cad_sched_fence() {
}

template <int N>
__device__ __forceinline__ void cad_wait_vmcnt() {}
__device__ __forceinline__ void cad_order_point(float&) {}
template <typename V>
__device__ __forceinline__ void cad_nt_store(V* p, V v) {
    *p = v;
}
// hand-off primitives of the concurrent dB / dC fold: launches are synchronous here (a consumer kernel only ever runs after its
// producer has finished), so write-through stores / loads are plain ones and the poll sees the final counts at once
__device__ __forceinline__ void cad_store16_wt(void* p, u32x4 v) { *(u32x4*)p = v; }
__device__ __forceinline__ void cad_load16x4_wt(const void* const (&base)[4], uint32_t voff, u32x4 (&v)[4]) {
    for (int i = 0; i < 4; ++i) v[i] = *(const u32x4*)((const char*)base[i] + voff);
}
__device__ __forceinline__ void cad_counter_add_agent(int* p, int v) { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
__device__ __forceinline__ int cad_counter_load_agent(const int* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
#define CAD_CU_KEYS 4096
__device__ __forceinline__ int cad_cu_key() { return 0; }  // one "CU": every kernel of a test is co-located with every other
__device__ __forceinline__ void cad_poll_sleep() {}
__device__ __forceinline__ uint64_t cad_wall_clock() {
    static uint64_t t = 0;  // every call "takes" 1 ms: a poll that does not succeed runs out of its budget after a few calls
    return __atomic_add_fetch(&t, 100000, __ATOMIC_RELAXED);
}
#define CAD_WALL_CLOCK_TICKS_PER_US 100ull  /* the wall clock of the device side runs at 100 MHz */
__device__ __forceinline__ uint64_t cad_wall_clock_hz() { return 100000000ull; }
#define CAD_BIG_LDS(kern, bytes) (void)0
#define CAD_OCCUPANCY(kern, threads, bytes) 0  /* no CUs on the host */

static inline float cad_e4m3_to_f32(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7)
        r = NAN;
    else if (e == 0)
        r = ldexpf((float)m, -9);  // subnormal: m * 2^-3 * 2^-6
    else
        r = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -r : r;
}
static inline uint8_t cad_f32_to_e4m3(float f) {  // round-to-nearest-even, saturating
    if (f != f) return 0x7f;
    const uint8_t s = f < 0 ? 0x80 : 0;
    float a = fabsf(f);
    if (a >= 464.0f) return s | 0x7e;  // beyond the midpoint to the (non-existent) next value: saturate to 448
    if (a < ldexpf(1.0f, -10)) return s;  // below half the smallest subnormal
    int e;
    (void)frexpf(a, &e);  // a = m * 2^e, m in [0.5, 1)
    int E = e - 1;        // a = 1.x * 2^E
    if (E < -6) E = -6;   // subnormal range: fixed exponent
    const float q = ldexpf(1.0f, E - 3);  // spacing
    float n = nearbyintf(a / q);          // (default rounding mode: to nearest even)
    float r = n * q;
    if (r > 448.0f) r = 448.0f;
    // encode r
    if (r < ldexpf(1.0f, -6)) return s | (uint8_t)nearbyintf(r / ldexpf(1.0f, -9));
    int e2;
    const float m2 = frexpf(r, &e2);  // r = m2 * 2^e2
    const int be = e2 - 1 + 7;
    const int mant = (int)nearbyintf((m2 * 2.0f - 1.0f) * 8.0f);
    return s | (uint8_t)(be << 3) | (uint8_t)mant;
}
__device__ __forceinline__ uint32_t cad_pack_fp8x4(float a, float b, float c, float d) {
    return (uint32_t)cad_f32_to_e4m3(a) | ((uint32_t)cad_f32_to_e4m3(b) << 8) | ((uint32_t)cad_f32_to_e4m3(c) << 16) |
           ((uint32_t)cad_f32_to_e4m3(d) << 24);
}
__device__ __forceinline__ f32x4 cad_mfma_16x16x32_fp8(u32x2 a, u32x2 b, f32x4 c) {
    const int lane = emu::lane_id();
    const int col = lane & 15, rg = lane >> 4;
    float bk[32], ak[4][32];
    const uint64_t mine_a = (uint64_t)a[0] | ((uint64_t)a[1] << 32), mine_b = (uint64_t)b[0] | ((uint64_t)b[1] << 32);
    const uint64_t* pb = emu_publish(mine_b);
    for (int g = 0; g < 4; ++g) {
        const uint64_t vb = pb[g * 16 + col];
        for (int t = 0; t < 8; ++t) bk[8 * g + t] = cad_e4m3_to_f32((uint8_t)(vb >> (8 * t)));
    }
    const uint64_t* pa = emu_publish(mine_a);
    for (int g = 0; g < 4; ++g)
        for (int r = 0; r < 4; ++r) {
            const uint64_t va = pa[g * 16 + 4 * rg + r];
            for (int t = 0; t < 8; ++t) ak[r][8 * g + t] = cad_e4m3_to_f32((uint8_t)(va >> (8 * t)));
        }
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        float s = c[r];
        for (int k = 0; k < 32; ++k) s += ak[r][k] * bk[k];
        d[r] = s;
    }
    return d;
}
