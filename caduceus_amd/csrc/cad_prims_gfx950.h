// gfx950 primitives of the Caduceus kernels: compiler builtins and inline asm (included by cad_common.h; the host emulator's restatement
// of exactly these names lives in tests/emu/cad_prims_emu.h).
#pragma once
#include <hip/hip_runtime.h>
#define CAD_LAUNCH(kern, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kern, (grid), (block), (shmem), (hipStream_t)(stream), __VA_ARGS__)
#define CAD_DEVICE_BUILD 1

// dynamic LDS (16-byte aligned base; keep ALL of a kernel's LDS in this one region - guide G17)
#define CAD_DYN_SMEM(T, name)                                              \
    extern __shared__ __attribute__((aligned(16))) char cad_smem_raw[];   \
    T* name = (T*)cad_smem_raw

#include "cad_types.h"

// two fp32 -> packed bf16x2 (lo in bits [15:0]); v_cvt_pk_bf16_f32 on gfx950 (round-to-nearest-even)
__device__ __forceinline__ uint32_t cad_pack_bf16x2(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ float cad_exp2(float x) {
    return __builtin_amdgcn_exp2f(x);  // v_exp_f32
}

__device__ __forceinline__ float cad_log(float x) {
    return __builtin_amdgcn_logf(x) * 0.6931471805599453f;  // v_log_f32 (log2) * ln2
}

__device__ __forceinline__ float cad_rcp(float x) {
    return __builtin_amdgcn_rcpf(x);
}

__device__ __forceinline__ float cad_rsqrt(float x) {
    return __builtin_amdgcn_rsqf(x);
}

// v_perm_b32: every byte of the result is picked by the matching selector byte from the 8 bytes {s0[3..0], s1[3..0]} -- selector values
// 0..3 = bytes 0..3 of s1, 4..7 = bytes 0..3 of s0, 0x0c = the constant 0x00.  One instruction selects a dword, swaps its halves and / or
// widens one bf16 half to fp32 (selector 0x..0c0c: the half lands in bits [31:16] over zero bits).
__device__ __forceinline__ uint32_t cad_perm(uint32_t s0, uint32_t s1, uint32_t sel) {
    return __builtin_amdgcn_perm(s0, s1, sel);
}

// v_mul_legacy_f32: a product in which 0 * anything (inf, NaN) is 0.  Through the LLVM intrinsic (this clang has no builtin for it), NOT
// inline asm: the operand is usually the result of a transcendental (v_rcp_f32), and a VALU read of a transcendental's result needs a
// wait state that the compiler's hazard recognizer only inserts for instructions it can see -- the asm form read stale registers on
// the device (round 5: dz of the first item of ~10 % of the lanes was 0; the host emulator cannot show this class of error).
extern "C" __device__ float cad_llvm_fmul_legacy(float, float) __asm("llvm.amdgcn.fmul.legacy");
__device__ __forceinline__ float cad_mul_legacy(float a, float b) {
    return cad_llvm_fmul_legacy(a, b);
}

// ---- cross-lane primitives (DPP on gfx950; emulated through the fiber exchange in the test build) -----------------
// Each returns, per lane, the value of `v` in the source lane selected by the pattern, or `old` where the pattern has
// no source for this lane -- exactly v_mov_b32_dpp with bound_ctrl:0.  Passing the identity element as `old` lets a
// scan step run unconditionally on all lanes.
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

__device__ __forceinline__ f32x4 cad_mfma_16x16x32_bf16(u32x4 a, u32x4 b, f32x4 c) {
    typedef __bf16 bf16x8_hw __attribute__((ext_vector_type(8)));
    typedef float f32x4_hw __attribute__((ext_vector_type(4)));
    const f32x4_hw r = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_hw, a), __builtin_bit_cast(bf16x8_hw, b),
                                                               __builtin_bit_cast(f32x4_hw, c), 0, 0, 0);
    return __builtin_bit_cast(f32x4, r);
}

// v_mfma_f32_16x16x4_f32:  D (16 x 16 fp32) = A (16 x 4 fp32) * B (4 x 16 fp32) + C.  Lane l, g = l >> 4:  A: row l & 15, k = g;
// B: k = g, column l & 15;  C / D as above (column l & 15, rows 4g + r).  Full fp32 operands: the LM head keeps its fp32 weight.
__device__ __forceinline__ f32x4 cad_mfma_16x16x4_f32(float a, float b, f32x4 c) {
    typedef float f32x4_hw __attribute__((ext_vector_type(4)));
    const f32x4_hw r = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, __builtin_bit_cast(f32x4_hw, c), 0, 0, 0);
    return __builtin_bit_cast(f32x4, r);
}

// ds_read_b64_tr_b16: transposing LDS read for 16-bit elements.  Within every 16-lane group, lane 4 r + c (r, c in 0..3)
// supplies the address of 4 contiguous elements S[r][4c .. 4c+3] of a 4 x 16 block S; lane l of the group receives the
// COLUMN  (S[0][l], S[1][l], S[2][l], S[3][l]).  This is how a [k][token] tile (token-contiguous, as the channel-major
// activations are) yields MFMA operand fragments, which want consecutive k per lane.  (Semantics as used by ck_tile's
// transpose loads, /opt/rocm/include/ck_tile/core/tensor/load_tile_transpose.hpp: Quad16 input / output encodings.)
__device__ __forceinline__ u32x2 cad_lds_read_tr16(const void* p) {
    typedef __bf16 bf16x4_hw __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) bf16x4_hw lds_vec_t;
    const bf16x4_hw v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_vec_t*)p);
    return __builtin_bit_cast(u32x2, v);
}

// two fp32 -> packed bf16x2 through a conversion the COMPILER sees (it emits v_cvt_pk_bf16_f32 and pads the MFMA / DOT
// result hazards itself; the inline-asm cad_pack_bf16x2 is invisible to its hazard recognizer)
__device__ __forceinline__ uint32_t cad_pack_bf16x2_safe(float lo, float hi) {
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    typedef float f32x2_hw __attribute__((ext_vector_type(2)));
    const f32x2_hw v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
}

// ---- LDS-DMA: asynchronous 16-byte-per-lane copy global -> LDS (no VGPR involved) ----------------------------------
// global_load_lds_dwordx4: lane l's 16 bytes, read from its own global address, land at (wave-uniform LDS base in M0)
// + 16 l.  Tracked by vmcnt like any load: the issuing wave waits vmcnt before its ds_read (and a barrier before another
// wave's).  A swizzled LDS image is obtained by permuting the per-lane SOURCE addresses; the destination is always linear.
__device__ __forceinline__ uint32_t cad_lds_off(const void* p) {  // byte offset of an LDS object inside the LDS aperture
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

__device__ __forceinline__ void cad_glds16(const void* gsrc /* per lane */, uint32_t lds_base /* wave-uniform, SGPR */) {
    uint32_t keep;  // M0 holds the LDS base of the DMA; it is compiler-reserved, so save / restore it in the same statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_base)
                 : "memory");
}

// Point at which every lane of the wave has executed what precedes it.  The hardware runs a wave in lock-step and its LDS
// queue is in order, so this is only a compiler fence on the device; the host emulator runs lanes as separate fibers and
// needs a real rendezvous wherever a lane reads LDS written by ANOTHER lane of its wave without a workgroup barrier.
__device__ __forceinline__ void cad_wave_sync() {
    __builtin_amdgcn_wave_barrier();
}

// wave-uniform "any lane" vote
__device__ __forceinline__ bool cad_wave_any(bool p) {
    return __builtin_amdgcn_ballot_w64(p) != 0;
}

// compiler-only fence: keeps the scheduler from hoisting (LDS) loads across this point, which bounds live ranges
__device__ __forceinline__ void cad_sched_fence() {
    asm volatile("" ::: "memory");
}

// s_waitcnt vmcnt(N): at most the N most recently issued vector-memory operations (loads, LDS-DMA, stores: they retire in issue order)
// may still be outstanding
template <int N>
__device__ __forceinline__ void cad_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// a point the scheduler may not move memory operations across, that also consumes `x` (so it sits behind the producer of `x`)
__device__ __forceinline__ void cad_order_point(float& x) { asm volatile("" : "+v"(x) : : "memory"); }
// streaming (nontemporal) store
template <typename V>
__device__ __forceinline__ void cad_nt_store(V* p, V v) {
    __builtin_nontemporal_store(v, p);
}

// ---- hand-off between workgroups of DIFFERENT, concurrently running kernels (the dB / dC fold behind the scan backward) --------------
// The per-XCD L2s are not coherent with each other and a CU's L1 is never refreshed by another CU's stores
// (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup visibility").  The valid write-through form used here:
// producer = sc1 (write-through) 16-byte stores, every storing wave drains them (s_waitcnt vmcnt(0)), workgroup barrier, ONE relaxed
// agent-scope atomic on the counter; consumer = ONE lane polls the counter with relaxed agent-scope loads, workgroup barrier, sc1 loads
// of the payload (never plain loads: an L1 / L2 line of the slot may predate the producer's store).
__device__ __forceinline__ void cad_store16_wt(void* p, u32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
// four write-through-coherent 16-byte loads, all in flight together, then the wait (the compiler does not see asm loads: the wait is
// part of the statement).  Scalar base + 32-bit per-lane byte offset: 1 address VGPR for the four loads -- the fold kernel that uses
// them has to fit into the 48 VGPRs two resident scan waves leave on a SIMD.
__device__ __forceinline__ void cad_load16x4_wt(const void* const (&base)[4] /* wave-uniform */, uint32_t voff, u32x4 (&v)[4]) {
    asm volatile(
        "global_load_dwordx4 %0, %4, %5 sc1\n\t"
        "global_load_dwordx4 %1, %4, %6 sc1\n\t"
        "global_load_dwordx4 %2, %4, %7 sc1\n\t"
        "global_load_dwordx4 %3, %4, %8 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
        : "v"(voff), "s"(base[0]), "s"(base[1]), "s"(base[2]), "s"(base[3])
        : "memory");
}
__device__ __forceinline__ void cad_counter_add_agent(int* p, int v) {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int cad_counter_load_agent(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// which CU this wave runs on, as a small table index: XCC_ID (0..15) x the {se_id, sh_id, cu_id} field of HW_REG_HW_ID (bits 15:8)
#define CAD_CU_KEYS 4096
__device__ __forceinline__ int cad_cu_key() {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return (int)(((xcc & 15u) << 8) | ((hw >> 8) & 255u));
}
#ifndef CAD_POLL_SLEEP
#define CAD_POLL_SLEEP 32
#endif
__device__ __forceinline__ void cad_poll_sleep() { __builtin_amdgcn_s_sleep(CAD_POLL_SLEEP); }  // 64 cycles per unit
__device__ __forceinline__ uint64_t cad_wall_clock() { return wall_clock64(); }             // constant-rate counter
#define CAD_WALL_CLOCK_TICKS_PER_US 100ull  /* the wall clock of the device side runs at 100 MHz */
__device__ __forceinline__ uint64_t cad_wall_clock_hz() { return 100000000ull; }            // 100 MHz on gfx9 (s_memrealtime)

// more than 64 KB of dynamic LDS has to be requested per kernel: the largest size asked for so far is remembered per call site (= per
// kernel instantiation) AND per device, so a later launch with a larger size (or on another GPU of the same process) raises it again
#define CAD_BIG_LDS(kern, bytes)                                                                                     \
    do {                                                                                                             \
        static size_t cur[CAD_MAX_DEVICES] = {0};                                                                    \
        int dev_ = 0;                                                                                                \
        if (hipGetDevice(&dev_) != hipSuccess || dev_ < 0 || dev_ >= CAD_MAX_DEVICES) return CAD_ERR_LAUNCH;         \
        if ((size_t)(bytes) > 65536 && (size_t)(bytes) > cur[dev_]) {                                                \
            if (hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != \
                hipSuccess)                                                                                          \
                return CAD_ERR_LAUNCH;                                                                               \
            cur[dev_] = (size_t)(bytes);                                                                             \
        }                                                                                                            \
    } while (0)

// workgroups of `kern` the runtime places on one CU (0 on failure): a diagnostic for the launch-shape experiments
#define CAD_OCCUPANCY(kern, threads, bytes)                                                                          \
    ([&]() {                                                                                                         \
        int nb_ = 0;                                                                                                 \
        return hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, (kern), (threads), (bytes)) == hipSuccess ? nb_ : 0; \
    }())

// ---- fp8 (OCP e4m3) ------------------------------------------------------------------------------------------------------------------
// four fp32 -> four e4m3 bytes (element j in byte j)
__device__ __forceinline__ uint32_t cad_pack_fp8x4(float a, float b, float c, float d) {
    int p = 0;
    p = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, p, false);
    p = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, p, true);
    return (uint32_t)p;
}
// v_mfma_f32_16x16x32_fp8_fp8: D (16 x 16 fp32) = A (16 x 32 e4m3) . B (32 x 16 e4m3) + C.  Lane l, g = l >> 4:
// A: row l & 15, elements k = 8g .. 8g+7 (byte t of the 64-bit operand = element 8g + t); B: column l & 15, same k; C / D as
// the bf16 form (column l & 15, rows 4g + r).
__device__ __forceinline__ f32x4 cad_mfma_16x16x32_fp8(u32x2 a, u32x2 b, f32x4 c) {
    typedef float f32x4_hw __attribute__((ext_vector_type(4)));
    const f32x4_hw r = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(__builtin_bit_cast(long, a), __builtin_bit_cast(long, b),
                                                                  __builtin_bit_cast(f32x4_hw, c), 0, 0, 0);
    return __builtin_bit_cast(f32x4, r);
}
