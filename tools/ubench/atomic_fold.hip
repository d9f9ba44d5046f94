// dB / dC cross-workgroup sum by global atomics (gfx950): what does it cost to let the 64 workgroups of one (row, parameter set) ADD
// their flush results into ONE shared buffer (the reference's selective_scan_cuda.bwd does exactly that with fp32 atomics) instead of
// writing 64 bf16 partial slots that cad_reduce_partials_multi folds afterwards?
//
// Launch shape of the real scan_bwd at configs[2]: 256 workgroups x 512 threads (one per CU), 256 chunks of 512 positions, 8 pair-steps per
// chunk; per pair-step a workgroup emits (dB, dC) x 2 states x 512 positions = 2048 sums.  Today: bf16, 16 bytes from each of 256
// lanes into the workgroup's own slot (mode 1).  The atomic forms give every lane CONSECUTIVE dwords (a wave instruction covers 256
// contiguous bytes) -- the slot layout is free, only the fold / the consumer has to know it.
//
// Modes (template parameter, so that rocprofv3 --pmc rows are told apart by kernel name)
//   0  compute only
//   1  today: bf16 slots, plain 16-byte stores by waves 0-3
//   2  fp32 atomics, no return, into the (row, set)'s shared fp32 buffer [chunk][pair-step][2048]; waves 0-3, 8 per lane and pair-step
//   3  the same from all 8 waves (4 per lane)
//   4  packed bf16 atomics (global_atomic_pk_add_bf16) into a shared bf16 buffer; waves 0-3, 4 per lane
//   5  fp32 atomics into one slot per XCD (HW_REG_XCC_ID): 8 slots instead of 64 -- does the line stay in that XCD's L2?
// Every mode checks its sums on the host (small integers: exact in fp32 and in bf16).
// Build / run:  hipcc --offload-arch=gfx950 -O3 -o atomic_fold atomic_fold.hip && ./atomic_fold [iters] [mode | -1 = all] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int NWG = 256, NT = 512, NCHUNK = 256, NPS = 8, G = 64, NGRP = NWG / G;
constexpr int VALS = 2048;  // sums per workgroup and pair-step

struct Args {
    void* buf;       // mode 1: bf16 [NWG][NCHUNK][NPS][VALS]; 2/3: fp32 [NGRP][NCHUNK][NPS][VALS]; 4: bf16 (same shape); 5: fp32 [8][NGRP][...]
    float* sink;
    unsigned* xcd_of_wg;
    int iters;
};

__device__ __forceinline__ unsigned val_of(int wg, int c, int ps, int idx) { return (unsigned)((wg * 5 + c * 3 + ps + idx) & 3); }
__device__ __forceinline__ unsigned bf16_bits(unsigned v) { return __builtin_bit_cast(unsigned, (float)v) >> 16; }

__device__ __forceinline__ void atomic_add_f32(float* p, float v) { asm volatile("global_atomic_add_f32 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void atomic_pk_add_bf16(unsigned* p, unsigned v) { asm volatile("global_atomic_pk_add_bf16 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

template <int MODE>
__global__ __launch_bounds__(NT) void flush_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];  // ~96 KB requested: one workgroup per CU, as the real kernel
    const int wg = blockIdx.x, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int grp = wg / G;
    unsigned xcd;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcd));
    if (t == 0) a.xcd_of_wg[wg] = xcd;
    float a0 = t * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    for (int c = NCHUNK - 1; c >= 0; --c) {
        for (int ps = 0; ps < NPS; ++ps) {
            for (int i = 0; i < a.iters; ++i) {
                a0 = a0 * 1.0001f + 0.5f, a1 = a1 * 1.0001f + 0.25f, a2 = a2 * 0.9999f + 0.5f, a3 = a3 * 0.9999f + 0.25f;
            }
            const size_t tile = ((size_t)c * NPS + ps) * VALS;
            if (MODE == 1) {
                if (wave < 4) {  // lane stores 8 consecutive bf16 sums
                    const int i0 = (wave * 64 + lane) * 8;
                    u32x4 o;
                    for (int q = 0; q < 4; ++q)
                        o[q] = bf16_bits(val_of(wg, c, ps, i0 + 2 * q)) | (bf16_bits(val_of(wg, c, ps, i0 + 2 * q + 1)) << 16);
                    *(u32x4*)((uint16_t*)a.buf + ((size_t)wg * NCHUNK * NPS * VALS + tile + i0)) = o;
                }
            } else if (MODE == 2) {
                if (wave < 4) {
                    float* dst = (float*)a.buf + (size_t)grp * NCHUNK * NPS * VALS + tile;
                    for (int q = 0; q < 8; ++q) {
                        const int i = (wave * 8 + q) * 64 + lane;
                        atomic_add_f32(dst + i, (float)val_of(wg, c, ps, i));
                    }
                }
            } else if (MODE == 3) {
                float* dst = (float*)a.buf + (size_t)grp * NCHUNK * NPS * VALS + tile;
                for (int q = 0; q < 4; ++q) {
                    const int i = (wave * 4 + q) * 64 + lane;
                    atomic_add_f32(dst + i, (float)val_of(wg, c, ps, i));
                }
            } else if (MODE == 4) {
                if (wave < 4) {
                    unsigned* dst = (unsigned*)a.buf + ((size_t)grp * NCHUNK * NPS * VALS + tile) / 2;
                    for (int q = 0; q < 4; ++q) {
                        const int d = (wave * 4 + q) * 64 + lane;  // dword = sums 2d, 2d + 1
                        atomic_pk_add_bf16(dst + d, bf16_bits(val_of(wg, c, ps, 2 * d)) | (bf16_bits(val_of(wg, c, ps, 2 * d + 1)) << 16));
                    }
                }
            } else if (MODE == 5) {
                if (wave < 4) {
                    float* dst = (float*)a.buf + ((size_t)xcd * NGRP + grp) * NCHUNK * NPS * VALS + tile;
                    for (int q = 0; q < 8; ++q) {
                        const int i = (wave * 8 + q) * 64 + lane;
                        atomic_add_f32(dst + i, (float)val_of(wg, c, ps, i));
                    }
                }
            }
            __syncthreads();  // the real kernel has one barrier per pair-step
        }
    }
    if (a0 + a1 + a2 + a3 == 12345.678f) a.sink[t] = a0;
}

static unsigned h_val(int wg, int c, int ps, int idx) { return (unsigned)((wg * 5 + c * 3 + ps + idx) & 3); }
static float bf16_to_f(uint16_t b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

template <int MODE>
static void run(int iters, int reps, const char* name) {
    size_t bytes = 0;
    if (MODE == 1) bytes = (size_t)NWG * NCHUNK * NPS * VALS * 2;
    if (MODE == 2 || MODE == 3) bytes = (size_t)NGRP * NCHUNK * NPS * VALS * 4;
    if (MODE == 4) bytes = (size_t)NGRP * NCHUNK * NPS * VALS * 2;
    if (MODE == 5) bytes = (size_t)8 * NGRP * NCHUNK * NPS * VALS * 4;
    Args a{};
    if (bytes) CHECK(hipMalloc(&a.buf, bytes));
    CHECK(hipMalloc(&a.sink, NT * 4));
    CHECK(hipMalloc(&a.xcd_of_wg, NWG * 4));
    a.iters = iters;
    CHECK(hipFuncSetAttribute((const void*)flush_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    hipEvent_t e0, e1, e2;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&e2));
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(e0));
        if (bytes && MODE != 1) CHECK(hipMemsetAsync(a.buf, 0, bytes));  // the atomic forms start from zero: part of their price
        CHECK(hipEventRecord(e1));
        flush_kernel<MODE><<<NWG, NT, 96 * 1024>>>(a);
        CHECK(hipEventRecord(e2));
        CHECK(hipDeviceSynchronize());
        float tz = 0, tk = 0;
        CHECK(hipEventElapsedTime(&tz, e0, e1)); CHECK(hipEventElapsedTime(&tk, e1, e2));
        long wrong = -1;
        if (r == reps - 1 && bytes) {  // check the last repetition (sampled chunks: the copy is up to 2 GB)
            wrong = 0;
            std::vector<unsigned> xcd(NWG);
            CHECK(hipMemcpy(xcd.data(), a.xcd_of_wg, NWG * 4, hipMemcpyDeviceToHost));
            const int sample_c[3] = {0, 97, NCHUNK - 1};
            for (int grp = 0; grp < NGRP; ++grp)
                for (int sc = 0; sc < 3; ++sc) {
                    const int c = sample_c[sc];
                    for (int ps = 0; ps < NPS; ++ps) {
                        const size_t tile = ((size_t)c * NPS + ps) * VALS;
                        std::vector<float> want(VALS, 0.f), got(VALS, 0.f);
                        for (int m = 0; m < G; ++m)
                            for (int i = 0; i < VALS; ++i) want[i] += (float)h_val(grp * G + m, c, ps, i);
                        if (MODE == 1) {
                            std::vector<uint16_t> tmp(VALS);
                            for (int m = 0; m < G; ++m) {
                                CHECK(hipMemcpy(tmp.data(), (uint16_t*)a.buf + (size_t)(grp * G + m) * NCHUNK * NPS * VALS + tile, VALS * 2, hipMemcpyDeviceToHost));
                                for (int i = 0; i < VALS; ++i) got[i] += bf16_to_f(tmp[i]);
                            }
                        } else if (MODE == 2 || MODE == 3) {
                            CHECK(hipMemcpy(got.data(), (float*)a.buf + (size_t)grp * NCHUNK * NPS * VALS + tile, VALS * 4, hipMemcpyDeviceToHost));
                        } else if (MODE == 4) {
                            std::vector<uint16_t> tmp(VALS);
                            CHECK(hipMemcpy(tmp.data(), (uint16_t*)a.buf + (size_t)grp * NCHUNK * NPS * VALS + tile, VALS * 2, hipMemcpyDeviceToHost));
                            for (int i = 0; i < VALS; ++i) got[i] = bf16_to_f(tmp[i]);
                        } else if (MODE == 5) {
                            std::vector<float> tmp(VALS);
                            for (int x = 0; x < 8; ++x) {
                                CHECK(hipMemcpy(tmp.data(), (float*)a.buf + ((size_t)x * NGRP + grp) * NCHUNK * NPS * VALS + tile, VALS * 4, hipMemcpyDeviceToHost));
                                for (int i = 0; i < VALS; ++i) got[i] += tmp[i];
                            }
                        }
                        for (int i = 0; i < VALS; ++i) wrong += got[i] != want[i];
                    }
                }
            int per_xcd[8] = {0};
            bool rr = true;
            for (int w = 0; w < NWG; ++w) per_xcd[xcd[w] & 7]++, rr = rr && (xcd[w] == (unsigned)(w % 8));
            printf("    workgroups per XCD: %d %d %d %d %d %d %d %d; blockIdx %% 8 == XCC_ID for every workgroup: %s\n", per_xcd[0], per_xcd[1],
                   per_xcd[2], per_xcd[3], per_xcd[4], per_xcd[5], per_xcd[6], per_xcd[7], rr ? "yes" : "no");
        }
        printf("mode %d %-58s rep %d: kernel %8.3f ms (%6.3f us per pair-step), zero-fill %6.3f ms, wrong %ld\n", MODE, name, r, tk,
               tk * 1e3 / (NCHUNK * NPS), tz, wrong);
    }
    if (a.buf) CHECK(hipFree(a.buf));
    CHECK(hipFree(a.sink)); CHECK(hipFree(a.xcd_of_wg));
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 0;
    const int mode = argc > 2 ? atoi(argv[2]) : -1;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;
    printf("atomic_fold: %d workgroups x %d threads, %d chunks x %d pair-steps, %d sums per workgroup and pair-step, groups of %d, compute iters = %d\n",
           NWG, NT, NCHUNK, NPS, VALS, G, iters);
    if (mode < 0 || mode == 0) run<0>(iters, reps, "compute only");
    if (mode < 0 || mode == 1) run<1>(iters, reps, "bf16 slots, 16-byte plain stores (today)");
    if (mode < 0 || mode == 2) run<2>(iters, reps, "fp32 atomics -> shared buffer, waves 0-3 x 8");
    if (mode < 0 || mode == 3) run<3>(iters, reps, "fp32 atomics -> shared buffer, 8 waves x 4");
    if (mode < 0 || mode == 4) run<4>(iters, reps, "pk bf16 atomics -> shared buffer, waves 0-3 x 4");
    if (mode < 0 || mode == 5) run<5>(iters, reps, "fp32 atomics -> one slot per XCD, waves 0-3 x 8");
    return 0;
}
