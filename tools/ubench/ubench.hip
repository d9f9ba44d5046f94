// Instruction-throughput microbenchmark for gfx950 (cycles per wave64 instruction on one SIMD), used to price the scan
// kernels' instruction mix.  Standalone: hipcc --offload-arch=gfx950 -O3 -o ubench ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X X X X X X X X
#define BODY(INS0, INS1, INS2, INS3, INS4, INS5, INS6, INS7) \
    asm volatile(REP8(INS0 "\n\t" INS1 "\n\t" INS2 "\n\t" INS3 "\n\t" INS4 "\n\t" INS5 "\n\t" INS6 "\n\t" INS7 "\n\t") \
                 : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "s"(sc))

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void ub(float* out, long long* cyc, int iters, float seed) {
    float r0 = seed + threadIdx.x * 1e-3f, r1 = r0 + 1.f, r2 = r0 + 2.f, r3 = r0 + 3.f, r4 = r0 + .4f, r5 = r0 + .5f, r6 = r0 + .6f, r7 = r0 + .7f;
    f32x2 p0 = {r0, r1}, p1 = {r2, r3}, p2 = {r4, r5}, p3 = {r6, r7};
    float sc = seed * 0.5f;
    __shared__ float lds[4096];
    lds[threadIdx.x] = r0; lds[threadIdx.x + 256] = r1;
    __syncthreads();
    unsigned la = (threadIdx.x & 63) * 8;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) BODY("v_fma_f32 %0, %0, %0, %0", "v_fma_f32 %1, %1, %1, %1", "v_fma_f32 %2, %2, %2, %2", "v_fma_f32 %3, %3, %3, %3", "v_fma_f32 %4, %4, %4, %4", "v_fma_f32 %5, %5, %5, %5", "v_fma_f32 %6, %6, %6, %6", "v_fma_f32 %7, %7, %7, %7");
        if (KIND == 1) BODY("v_exp_f32 %0, %0", "v_exp_f32 %1, %1", "v_exp_f32 %2, %2", "v_exp_f32 %3, %3", "v_exp_f32 %4, %4", "v_exp_f32 %5, %5", "v_exp_f32 %6, %6", "v_exp_f32 %7, %7");
        if (KIND == 2) BODY("v_pk_fma_f32 %8, %8, %8, %8", "v_pk_fma_f32 %9, %9, %9, %9", "v_pk_fma_f32 %10, %10, %10, %10", "v_pk_fma_f32 %11, %11, %11, %11", "v_pk_fma_f32 %8, %8, %8, %8", "v_pk_fma_f32 %9, %9, %9, %9", "v_pk_fma_f32 %10, %10, %10, %10", "v_pk_fma_f32 %11, %11, %11, %11");
        if (KIND == 3) BODY("v_pk_mul_f32 %8, %8, %8", "v_pk_mul_f32 %9, %9, %9", "v_pk_mul_f32 %10, %10, %10", "v_pk_mul_f32 %11, %11, %11", "v_pk_mul_f32 %8, %8, %8", "v_pk_mul_f32 %9, %9, %9", "v_pk_mul_f32 %10, %10, %10", "v_pk_mul_f32 %11, %11, %11");
        if (KIND == 4) BODY("v_mul_f32 %0, %0, %0", "v_mul_f32 %1, %1, %1", "v_mul_f32 %2, %2, %2", "v_mul_f32 %3, %3, %3", "v_mul_f32 %4, %4, %4", "v_mul_f32 %5, %5, %5", "v_mul_f32 %6, %6, %6", "v_mul_f32 %7, %7, %7");
        if (KIND == 5) BODY("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf");
        if (KIND == 6) BODY("v_mov_b32 %0, %1", "v_mov_b32 %1, %2", "v_mov_b32 %2, %3", "v_mov_b32 %3, %4", "v_mov_b32 %4, %5", "v_mov_b32 %5, %6", "v_mov_b32 %6, %7", "v_mov_b32 %7, %0");
        if (KIND == 7) BODY("v_rcp_f32 %0, %0", "v_rcp_f32 %1, %1", "v_rcp_f32 %2, %2", "v_rcp_f32 %3, %3", "v_rcp_f32 %4, %4", "v_rcp_f32 %5, %5", "v_rcp_f32 %6, %6", "v_rcp_f32 %7, %7");
        if (KIND == 8) BODY("v_fma_f32 %0, %0, %0, %0", "v_exp_f32 %1, %1", "v_fma_f32 %2, %2, %2, %2", "v_fma_f32 %3, %3, %3, %3", "v_fma_f32 %4, %4, %4, %4", "v_exp_f32 %5, %5", "v_fma_f32 %6, %6, %6, %6", "v_fma_f32 %7, %7, %7, %7");  // 2 exp : 6 fma
        if (KIND == 9) BODY("v_fma_f32 %0, %0, %12, %0", "v_fma_f32 %1, %1, %12, %1", "v_fma_f32 %2, %2, %12, %2", "v_fma_f32 %3, %3, %12, %3", "v_fma_f32 %4, %4, %12, %4", "v_fma_f32 %5, %5, %12, %5", "v_fma_f32 %6, %6, %12, %6", "v_fma_f32 %7, %7, %12, %7");  // sgpr operand
        if (KIND == 10) BODY("v_fma_f32 %0, %0, %0, %0", "v_fma_f32 %0, %0, %0, %0", "v_fma_f32 %0, %0, %0, %0", "v_fma_f32 %0, %0, %0, %0", "v_fma_f32 %0, %0, %0, %0", "v_fma_f32 %0, %0, %0, %0", "v_fma_f32 %0, %0, %0, %0", "v_fma_f32 %0, %0, %0, %0");  // dependent chain
        if (KIND == 11) BODY("v_pk_fma_f32 %8, %8, %8, %8", "v_pk_fma_f32 %8, %8, %8, %8", "v_pk_fma_f32 %8, %8, %8, %8", "v_pk_fma_f32 %8, %8, %8, %8", "v_pk_fma_f32 %8, %8, %8, %8", "v_pk_fma_f32 %8, %8, %8, %8", "v_pk_fma_f32 %8, %8, %8, %8", "v_pk_fma_f32 %8, %8, %8, %8");  // dependent pk chain
        if (KIND == 12) BODY("v_exp_f32 %0, %0", "v_exp_f32 %0, %0", "v_exp_f32 %0, %0", "v_exp_f32 %0, %0", "v_exp_f32 %0, %0", "v_exp_f32 %0, %0", "v_exp_f32 %0, %0", "v_exp_f32 %0, %0");  // dependent exp chain
        if (KIND == 13) BODY("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "s_nop 1", "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "s_nop 1", "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "s_nop 1", "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "s_nop 1");  // dependent dpp chain (4 per BODY)
        if (KIND == 14) BODY("v_log_f32 %0, %0", "v_log_f32 %1, %1", "v_log_f32 %2, %2", "v_log_f32 %3, %3", "v_log_f32 %4, %4", "v_log_f32 %5, %5", "v_log_f32 %6, %6", "v_log_f32 %7, %7");
        if (KIND == 15) BODY("v_cvt_pk_bf16_f32 %0, %0, %1", "v_cvt_pk_bf16_f32 %1, %1, %2", "v_cvt_pk_bf16_f32 %2, %2, %3", "v_cvt_pk_bf16_f32 %3, %3, %4", "v_cvt_pk_bf16_f32 %4, %4, %5", "v_cvt_pk_bf16_f32 %5, %5, %6", "v_cvt_pk_bf16_f32 %6, %6, %7", "v_cvt_pk_bf16_f32 %7, %7, %0");
        if (KIND == 16) BODY("v_exp_f16 %0, %0", "v_exp_f16 %1, %1", "v_exp_f16 %2, %2", "v_exp_f16 %3, %3", "v_exp_f16 %4, %4", "v_exp_f16 %5, %5", "v_exp_f16 %6, %6", "v_exp_f16 %7, %7");
        if (KIND == 17) BODY("v_ldexp_f32 %0, %0, %1", "v_ldexp_f32 %1, %1, %2", "v_ldexp_f32 %2, %2, %3", "v_ldexp_f32 %3, %3, %4", "v_ldexp_f32 %4, %4, %5", "v_ldexp_f32 %5, %5, %6", "v_ldexp_f32 %6, %6, %7", "v_ldexp_f32 %7, %7, %0");
        if (KIND == 20)  // fma, four distinct registers per instruction (register-file port / bank pressure as in real code)
            asm volatile(REP8("v_fma_f32 v40, v41, v42, v43\n\tv_fma_f32 v44, v45, v46, v47\n\tv_fma_f32 v48, v49, v50, v51\n\t"
                              "v_fma_f32 v52, v53, v54, v55\n\tv_fma_f32 v41, v44, v49, v54\n\tv_fma_f32 v45, v48, v53, v42\n\t"
                              "v_fma_f32 v49, v52, v43, v46\n\tv_fma_f32 v53, v40, v47, v50\n\t")
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (KIND == 21)  // fma with all operands in the same bank (register index mod 4 equal)
            asm volatile(REP8("v_fma_f32 v40, v44, v48, v52\n\tv_fma_f32 v56, v60, v64, v68\n\tv_fma_f32 v44, v48, v52, v56\n\t"
                              "v_fma_f32 v60, v64, v68, v40\n\tv_fma_f32 v48, v52, v56, v60\n\tv_fma_f32 v64, v68, v40, v44\n\t"
                              "v_fma_f32 v52, v56, v60, v64\n\tv_fma_f32 v68, v40, v44, v48\n\t")
                         ::: "v40", "v44", "v48", "v52", "v56", "v60", "v64", "v68");
        if (KIND == 22)  // pk_fma, distinct register pairs
            asm volatile(REP8("v_pk_fma_f32 v[40:41], v[42:43], v[44:45], v[46:47]\n\tv_pk_fma_f32 v[48:49], v[50:51], v[52:53], v[54:55]\n\t"
                              "v_pk_fma_f32 v[56:57], v[58:59], v[60:61], v[62:63]\n\tv_pk_fma_f32 v[64:65], v[66:67], v[68:69], v[70:71]\n\t"
                              "v_pk_fma_f32 v[42:43], v[48:49], v[58:59], v[68:69]\n\tv_pk_fma_f32 v[50:51], v[56:57], v[66:67], v[44:45]\n\t"
                              "v_pk_fma_f32 v[58:59], v[64:65], v[46:47], v[52:53]\n\tv_pk_fma_f32 v[66:67], v[40:41], v[54:55], v[60:61]\n\t")
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55",
                             "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71");
        if (KIND == 23)  // mul, distinct registers
            asm volatile(REP8("v_mul_f32 v40, v41, v42\n\tv_mul_f32 v43, v44, v45\n\tv_mul_f32 v46, v47, v48\n\tv_mul_f32 v49, v50, v51\n\t"
                              "v_mul_f32 v41, v43, v47\n\tv_mul_f32 v44, v46, v50\n\tv_mul_f32 v47, v49, v42\n\tv_mul_f32 v50, v40, v45\n\t")
                         ::: "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51");
        if (KIND == 24)  // fma with an SGPR operand + 2 distinct VGPRs
            asm volatile(REP8("v_fma_f32 v40, v41, %0, v43\n\tv_fma_f32 v44, v45, %0, v47\n\tv_fma_f32 v48, v49, %0, v51\n\t"
                              "v_fma_f32 v52, v53, %0, v55\n\tv_fma_f32 v41, v44, %0, v54\n\tv_fma_f32 v45, v48, %0, v42\n\t"
                              "v_fma_f32 v49, v52, %0, v46\n\tv_fma_f32 v53, v40, %0, v50\n\t")
                         :: "s"(sc) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] + lds[la & 1023];
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_body, int waves_per_simd) {
    const int iters = 2000;
    const int threads = 256 * waves_per_simd, blocks = 256;
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, sizeof(long long) * threads * blocks / 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    ub<KIND><<<blocks, threads>>>(out, cyc, 10, 0.5f);
    hipEventRecord(e0);
    ub<KIND><<<blocks, threads>>>(out, cyc, iters, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(threads * blocks / 64);
    hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double n = (double)iters * 8 * per_body;
    printf("%-28s waves/SIMD=%d  wall %.3f ms  ->  %.2f ns/instr/wave  readcyclecounter ticks/instr/wave %.3f  (x waves/SIMD = SIMD ticks per instr: %.3f)\n",
           name, waves_per_simd, ms, ms * 1e6 / n, avg / n, avg / n / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_fma_f32", 8, w);
        run<20>("v_fma_f32 4 distinct regs", 8, w);
        run<21>("v_fma_f32 same-bank regs", 8, w);
        run<24>("v_fma_f32 sgpr + distinct", 8, w);
        run<22>("v_pk_fma_f32 distinct pairs", 8, w);
        run<23>("v_mul_f32 distinct regs", 8, w);
        run<4>("v_mul_f32", 8, w);
        run<9>("v_fma_f32 (sgpr src)", 8, w);
        run<6>("v_mov_b32", 8, w);
        run<2>("v_pk_fma_f32", 8, w);
        run<3>("v_pk_mul_f32", 8, w);
        run<1>("v_exp_f32", 8, w);
        run<14>("v_log_f32", 8, w);
        run<7>("v_rcp_f32", 8, w);
        run<16>("v_exp_f16", 8, w);
        run<17>("v_ldexp_f32", 8, w);
        run<15>("v_cvt_pk_bf16_f32", 8, w);
        run<5>("v_mul_f32_dpp row_shr:1", 8, w);
        run<8>("mix 2 exp : 6 fma", 8, w);
        run<10>("v_fma_f32 dependent", 8, w);
        run<11>("v_pk_fma_f32 dependent", 8, w);
        run<12>("v_exp_f32 dependent", 8, w);
        run<13>("v_mul_f32_dpp dependent+nop", 4, w);
    }
    return 0;
}
