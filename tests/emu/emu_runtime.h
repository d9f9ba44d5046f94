// Host emulator for the HIP kernels in caduceus_amd/csrc  --  TEST INFRASTRUCTURE ONLY.
//
// The build container has no GPU.  To exercise the *actual kernel sources* (index maps, wave scans, LDS
// staging, barriers) against the oracle before spending GPU minutes, tests/emu/build_emu.py compiles the very
// same .hip files with g++ and -DCAD_EMU; this header then supplies the tiny subset of the HIP device/runtime
// API the kernels use.  Each GPU thread of a workgroup runs as a fiber on one OS thread (workgroups are spread over the host's cores), so
// __syncthreads(), wave64 shuffles and LDS behave with real workgroup semantics (round-robin interleaving).
//
// The resulting libcaduceus_emu.so is loaded ONLY by the test-suite through the explicit hook
// caduceus_amd._lib.use_library_for_testing(); the product loader never falls back to it.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace emu {
extern thread_local dim3 g_threadIdx, g_blockIdx;  // workgroups of a launch run on several OS threads
extern dim3 g_blockDim, g_gridDim;
void launch(dim3 grid, dim3 block, size_t dyn_smem, const std::function<void()>& body);
void syncthreads();
void wave_sync();
uint64_t* wave_buf();  // 2 x 64 exchange slots of the calling fiber's wave (double buffered by parity)
int& lane_parity();   // per-fiber shuffle parity
int lane_id();
int wave_lanes();
char* dyn_smem();
}  // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static thread_local  // one workgroup per OS thread at a time
#define __launch_bounds__(...)
#ifndef __restrict__
#define __restrict__
#endif

inline void __syncthreads() { emu::syncthreads(); }

template <class T>
inline T emu_exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    // one rendezvous per shuffle: consecutive shuffles alternate between two buffers, and a lane can never be two
    // shuffles ahead of another lane of its wave because each shuffle contains a wave-wide rendezvous.
    int& par = emu::lane_parity();
    uint64_t* buf = emu::wave_buf() + 64 * par;
    par ^= 1;
    buf[emu::lane_id()] = raw;
    emu::wave_sync();
    uint64_t r = buf[src_lane];
    T out;
    std::memcpy(&out, &r, sizeof(T));
    return out;
}
// One rendezvous, many reads: every lane publishes 8 bytes and may then read ANY lane's slot until its own next publish / exchange (the
// slots are double buffered by parity and no lane can be two rendezvous ahead of another lane of its wave).  The matrix-core primitives
// gather their 16 x 4 / 16 x 32 operand panels with 1 / 4 of these instead of 20 / 40 single-slot exchanges.
inline const uint64_t* emu_publish(uint64_t raw) {
    int& par = emu::lane_parity();
    uint64_t* buf = emu::wave_buf() + 64 * par;
    par ^= 1;
    buf[emu::lane_id()] = raw;
    emu::wave_sync();
    return buf;
}
template <class T>
inline T __shfl(T v, int src, int width = 64) {
    int lane = emu::lane_id();
    int base = lane & ~(width - 1);
    return emu_exchange(v, base + (src & (width - 1)));
}
template <class T>
inline T __shfl_up(T v, unsigned delta, int width = 64) {
    int lane = emu::lane_id();
    int src = ((lane & (width - 1)) < (int)delta) ? lane : lane - (int)delta;
    return emu_exchange(v, src);
}
template <class T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
    int lane = emu::lane_id();
    int src = ((lane & (width - 1)) + (int)delta >= width) ? lane : lane + (int)delta;
    if (src >= emu::wave_lanes()) src = lane;
    return emu_exchange(v, src);
}
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    int lane = emu::lane_id();
    int src = lane ^ mask;
    if (src >= emu::wave_lanes()) src = lane;
    return emu_exchange(v, src);
}

// (workgroups run concurrently on several OS threads: global-memory atomics are real ones)
inline float atomicAdd(float* p, float v) {
    uint32_t* q = (uint32_t*)p;
    uint32_t old = __atomic_load_n(q, __ATOMIC_RELAXED);
    for (;;) {
        float o;
        std::memcpy(&o, &old, 4);
        const float n = o + v;
        uint32_t nb;
        std::memcpy(&nb, &n, 4);
        if (__atomic_compare_exchange_n(q, &old, nb, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) return o;
    }
}
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

// ---- host runtime subset --------------------------------------------------------------------------
typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    std::memset(p, v, n);
    return 0;
}
