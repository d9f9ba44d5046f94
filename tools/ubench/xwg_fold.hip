// Cross-workgroup fold microbenchmark (gfx950): what would it cost scan_bwd to sum the dB/dC partial tiles of the workgroups
// of one (row, parameter set) INSIDE the launch instead of writing 64-deep partial slots for cad_reduce_partials?
//
// Shape of the real kernel: 256 workgroups x 512 threads, one per CU; a (row, set) is 64 workgroups
// (8 channels each); per 512-position chunk a workgroup emits 8 pair tiles x (dB, dC) x 2 states x 512 positions bf16 = 32 KB;
// 256 chunks per launch; ~15 us of arithmetic per chunk.
//
// Modes
//   0  today:   every (workgroup, chunk) tile goes to its own slot (2.1 GB footprint), plain stores, no synchronisation; the fold is
//               a second, streaming kernel (timed separately).
//   1  ring:    tiles go write-through (sc1) into a ring of R chunks per group; per chunk ONE lane bumps an agent-scope counter after
//               every storing wave drained its stores; D chunks later each workgroup folds ITS 1/G share of that chunk (G tiles x
//               32 KB / G, fixed order -> deterministic sums) and writes the final rows.  Visibility follows the placement-
//               independent recipe (cdna_hip_programming.md, Guideline 16 R1): sc1 payload, relaxed agent counter, relaxed poll by one
//               lane, one agent-scope acquire, then loads.  A second counter per ring slot is the back-pressure (a slot is rewritten
//               only after all G members folded its previous occupant).  Needs the G workgroups of a group co-resident: true for
//               this launch (256 workgroups, 256 CUs) -- the spins are bounded and counted, a timeout is reported, never a hang.
//   2  ring, fold reads with sc1 loads instead of acquire + plain loads.
//   4  ring, publish only (sc1 stores, drain, counter; no waits, no fold): isolates the cost of draining the stores every chunk.
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o xwg_fold xwg_fold.hip && ./xwg_fold [iters_compute] [G] [D]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NWG = 256, NT = 512, NCHUNK = 256, R = 4;
constexpr int TILE_BYTES = 32768;                 // per workgroup and chunk
constexpr int TILE_VECS = TILE_BYTES / 16;        // 2048 16-byte vectors: 4 per thread
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct Args {
    u32x4* slots;        // mode 0: [NWG][NCHUNK][TILE_VECS];  ring: [ngroups][R][G][TILE_VECS]
    u32x4* out;          // folded rows: [ngroups][NCHUNK][TILE_VECS]
    unsigned* arrive;    // [ngroups][NCHUNK] tiles published
    unsigned* folded;    // [ngroups][NCHUNK] members that finished folding the chunk
    unsigned* timeouts;  // [1]
    float* sink;
    long long* cycles;   // [NWG] kernel cycles per workgroup
    int iters, G, D, mode;
};

__device__ __forceinline__ u32x4 make_tile_vec(int wg, int chunk, int v) {
    // small integers: the fold of G tiles is exact in uint32, checked on the host
    const unsigned x = (unsigned)(wg * 7 + chunk * 3 + v);
    return u32x4{x & 1023u, (x >> 1) & 1023u, (x >> 2) & 1023u, (x >> 3) & 1023u};
}

__device__ __forceinline__ void store_sc1(u32x4* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ bool wait_ge(unsigned* w, unsigned want, unsigned* timeouts) {
    // bounded: a timeout anywhere is sticky for the whole launch (every later wait returns at once), so a protocol error ends in
    // a report after ~0.5 s, never in a hang
    unsigned spins = 0;
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if ((++spins & 1023u) == 0u && __hip_atomic_load(timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        if (spins > SPIN_LIMIT) { atomicAdd(timeouts, 1u); return false; }
    }
    return true;
}

__global__ __launch_bounds__(NT) void producer(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 96 KB requested: one workgroup per CU, as the real kernel
    unsigned* flag = (unsigned*)smem;
    const int wg = blockIdx.x, t = threadIdx.x;
    const int G = a.G, grp = wg / G, mem = wg % G;
    float acc = (float)t * 1e-3f;
    const long long t0 = __builtin_readcyclecounter();
    bool ok = true;
    // backward order, as the kernel walks L
    for (int c = NCHUNK - 1; c >= -a.D; --c) {
        if (c >= 0) {
            // "arithmetic" of the chunk
            for (int i = 0; i < a.iters; ++i) acc = acc * 1.0001f + 0.5f;
            if (a.mode == 3) continue;  // calibration: the arithmetic alone
            if (a.mode == 0) {
                u32x4* dst = a.slots + ((size_t)wg * NCHUNK + c) * TILE_VECS;
#pragma unroll
                for (int k = 0; k < 4; ++k) dst[k * NT + t] = make_tile_vec(wg, c, k * NT + t);
            } else {
                // back-pressure: the ring slot's previous occupant (chunk c + R) must have been folded by every member
                if (c + R < NCHUNK && a.mode != 4) {
                    if (t == 0) flag[0] = wait_ge(a.folded + grp * NCHUNK + c + R, (unsigned)G, a.timeouts) ? 1u : 0u;
                    __syncthreads();
                    ok = ok && flag[0] != 0u;
                    __syncthreads();
                }
                u32x4* dst = a.slots + (((size_t)grp * R + (c % R)) * G + mem) * TILE_VECS;
#pragma unroll
                for (int k = 0; k < 4; ++k) store_sc1(dst + k * NT + t, make_tile_vec(wg, c, k * NT + t));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains
                __syncthreads();
                if (t == 0) __hip_atomic_fetch_add(a.arrive + grp * NCHUNK + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const int f = c + a.D;  // the chunk this workgroup helps to fold now
        if ((a.mode == 1 || a.mode == 2) && f < NCHUNK && ok) {
            if (t == 0) flag[0] = wait_ge(a.arrive + grp * NCHUNK + f, (unsigned)G, a.timeouts) ? 1u : 0u;
            __syncthreads();
            const bool seen = flag[0] != 0u;
            __syncthreads();
            if (!seen) { ok = false; continue; }
            if (a.mode == 1) {
                if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
            }
            // my share: TILE_VECS / G vectors of every member's tile; thread t sums `per` members of vector t % share (always four
            // 16-byte loads per thread), the parts meet in LDS and are added in a fixed order
            const int share = TILE_VECS / G, parts = NT / share, per = G / parts;
            const int v = t % share, part = t / share;
            const u32x4* src = a.slots + ((size_t)grp * R + (f % R)) * G * TILE_VECS + mem * share + v;
            u32x4 x[4];
            if (a.mode == 2) {
#pragma unroll
                for (int m = 0; m < 4; ++m)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(x[m]) : "v"(src + (size_t)(part * per + (m < per ? m : 0)) * TILE_VECS) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3])::"memory");
            } else {
#pragma unroll
                for (int m = 0; m < 4; ++m) x[m] = src[(size_t)(part * per + (m < per ? m : 0)) * TILE_VECS];
            }
            u32x4 s = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m < per) s += x[m];
            u32x4* red = (u32x4*)(smem + 64);
            red[part * share + v] = s;
            __syncthreads();
            if (t < share) {
                u32x4 tot = red[t];
                for (int q = 1; q < parts; ++q) tot += red[q * share + t];
                a.out[((size_t)grp * NCHUNK + f) * TILE_VECS + mem * share + t] = tot;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(a.folded + grp * NCHUNK + f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (t == 0) a.cycles[wg] = __builtin_readcyclecounter() - t0;
    if (acc == 12345.678f) a.sink[0] = acc;
}

// mode 0's second kernel: fold the G slots of a group (streaming, as cad_reduce_partials)
__global__ __launch_bounds__(256) void fold_slots(const u32x4* slots, u32x4* out, int G, size_t n_per_wg) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;  // vector index inside a group's (chunk, vec) space
    const int grp = blockIdx.y;
    if (i >= n_per_wg) return;
    u32x4 s = {0u, 0u, 0u, 0u};
    for (int m = 0; m < G; ++m) s += slots[((size_t)(grp * G + m)) * n_per_wg + i];
    out[(size_t)grp * n_per_wg + i] = s;
}

static int check(const std::vector<u32x4>& out, int G) {
    int bad = 0;
    const int ngroups = NWG / G;
    for (int grp = 0; grp < ngroups; ++grp)
        for (int c = 0; c < NCHUNK; c += 37)
            for (int v = 0; v < TILE_VECS; v += 101) {
                unsigned e[4] = {0, 0, 0, 0};
                for (int m = 0; m < G; ++m) {
                    const unsigned x = (unsigned)((grp * G + m) * 7 + c * 3 + v);
                    e[0] += x & 1023u, e[1] += (x >> 1) & 1023u, e[2] += (x >> 2) & 1023u, e[3] += (x >> 3) & 1023u;
                }
                const u32x4 g = out[((size_t)grp * NCHUNK + c) * TILE_VECS + v];
                for (int k = 0; k < 4; ++k) bad += g[k] != e[k];
            }
    return bad;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 6000;
    const int G = argc > 2 ? atoi(argv[2]) : 64;
    const int D = argc > 3 ? atoi(argv[3]) : 2;
    if (NWG % G || G < 4 || G > 64 || (G & (G - 1)) || D >= R || D < 1) { printf("bad G / D (G: power of two in 4..64, 1 <= D < R)\n"); return 1; }
    const int ngroups = NWG / G;
    Args a{};
    const size_t slot_vecs = (size_t)NWG * NCHUNK * TILE_VECS;  // mode 0 footprint (2.1 GB)
    CHECK(hipMalloc(&a.slots, slot_vecs * 16));
    CHECK(hipMalloc(&a.out, (size_t)ngroups * NCHUNK * TILE_VECS * 16));
    CHECK(hipMalloc(&a.arrive, ngroups * NCHUNK * 4));
    CHECK(hipMalloc(&a.folded, ngroups * NCHUNK * 4));
    CHECK(hipMalloc(&a.timeouts, 4));
    CHECK(hipMalloc(&a.sink, 4));
    CHECK(hipMalloc(&a.cycles, NWG * 8));
    a.iters = iters, a.G = G, a.D = D;
    CHECK(hipFuncSetAttribute((const void*)producer, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    std::vector<u32x4> host((size_t)ngroups * NCHUNK * TILE_VECS);
    printf("xwg_fold: %d workgroups x %d threads, %d chunks, 32 KB per (workgroup, chunk), G = %d, D = %d, ring R = %d, compute iters = %d\n",
           NWG, NT, NCHUNK, G, D, R, iters);
    const int modes[] = {-1, 0, 4, 1, 2};
    for (int mode : modes) {
        for (int rep = 0; rep < 3; ++rep) {
            a.mode = mode < 0 ? 0 : mode;
            const int save_iters = a.iters;
            CHECK(hipMemsetAsync(a.arrive, 0, ngroups * NCHUNK * 4));
            CHECK(hipMemsetAsync(a.folded, 0, ngroups * NCHUNK * 4));
            CHECK(hipMemsetAsync(a.timeouts, 0, 4));
            CHECK(hipMemsetAsync(a.out, 0xff, (size_t)ngroups * NCHUNK * TILE_VECS * 16));
            CHECK(hipEventRecord(e0));
            if (mode < 0) {  // compute only: no stores at all (calibration of the arithmetic phase)
                Args b = a; b.mode = 3;  // mode 3 falls through both branches: only the loop
                hipLaunchKernelGGL(producer, dim3(NWG), dim3(NT), 96 * 1024, 0, b);
            } else {
                hipLaunchKernelGGL(producer, dim3(NWG), dim3(NT), 96 * 1024, 0, a);
            }
            CHECK(hipEventRecord(e1));
            if (mode == 0) {
                const size_t n_per_wg = (size_t)NCHUNK * TILE_VECS;
                hipLaunchKernelGGL(fold_slots, dim3((unsigned)((n_per_wg + 255) / 256), ngroups), dim3(256), 0, 0, a.slots, a.out, G, n_per_wg);
            }
            CHECK(hipEventRecord(e2));
            CHECK(hipDeviceSynchronize());
            a.iters = save_iters;
            float ms_k = 0, ms_f = 0;
            CHECK(hipEventElapsedTime(&ms_k, e0, e1));
            CHECK(hipEventElapsedTime(&ms_f, e1, e2));
            unsigned tmo = 0;
            CHECK(hipMemcpy(&tmo, a.timeouts, 4, hipMemcpyDeviceToHost));
            int bad = -1;
            if (mode >= 0 && mode != 4) {
                CHECK(hipMemcpy(host.data(), a.out, host.size() * 16, hipMemcpyDeviceToHost));
                bad = check(host, G);
            }
            const char* names[] = {"compute only (no stores)", "0 slots + fold kernel", "1 ring, sc1 stores, acquire + plain loads",
                                   "2 ring, sc1 stores, sc1 loads", "", "4 ring, publish only (no fold)"};
            printf("mode %-44s rep %d: kernel %.3f ms (%.2f us per chunk), fold kernel %.3f ms, wrong %d, spin timeouts %u\n",
                   names[mode + 1], rep, ms_k, ms_k * 1e3 / NCHUNK, mode == 0 ? ms_f : 0.f, bad, tmo);
        }
    }
    return 0;
}
