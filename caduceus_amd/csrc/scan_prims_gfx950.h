// gfx950 forms of the scan kernels' register-level primitives (included by scan_common.h behind its helpers f2 / readlane2 / SC_COMBINE;
// the host emulator's restatement of exactly these names: tests/emu/scan_prims_emu.h).
#pragma once

// broadcast one half of a float2 to both halves: folds into op_sel / op_sel_hi of the consuming v_pk_* instruction,
// so a per-item scalar pair such as (dt, dt*u) costs 2 VGPRs instead of 4 and no v_mov
__device__ __forceinline__ f32x2 splat_lo(f32x2 v) {
    return __builtin_shufflevector(v, v, 0, 0);
}

__device__ __forceinline__ f32x2 splat_hi(f32x2 v) {
    return __builtin_shufflevector(v, v, 1, 1);
}

// acc + a . b as two scalar v_fmac.  Written in asm because the SLP vectoriser otherwise packs the dot products of two
// neighbouring items into v_pk_* and pays for the transposition with 6 v_mov + 1 v_pk_mov per item pair (12
// instructions per two items instead of 6; measured in the forward scan's output phase).  -fno-slp-vectorize gives the
// same instruction count but lets the scheduler hoist the C-tile reads (256 VGPRs + spills instead of 215).
// acc += a * b as ONE v_pk_fma_f32, pinned by asm (keeps the consumer order of the LDS reads, hence the register
// pressure, under the programmer's control).  A packed-FMA result needs one wait state before a dependent VALU read:
// callers produce b one item ahead so that no s_nop is needed.
__device__ __forceinline__ void pk_fma_acc(f32x2& acc, f32x2 a, f32x2 b) {
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

__device__ __forceinline__ float dot2_acc(float acc, f32x2 a, f32x2 b) {
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a[0]), "v"(b[0]));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a[1]), "v"(b[1]));
    return acc;
}

// The first dot product of an accumulation (v_mul + v_fmac): no zero-initialised accumulator register, no v_mov
__device__ __forceinline__ float dot2_first(f32x2 a, f32x2 b) {
    float acc;
    asm("v_mul_f32 %0, %1, %2" : "=v"(acc) : "v"(a[0]), "v"(b[0]));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(a[1]), "v"(b[1]));
    return acc;
}

// Sums inside every group of 8 consecutive lanes (each of the 8 lanes receives its group's sum), two values at once: a butterfly of
// three DPP steps -- lane ^ 1, lane ^ 2 (quad_perm), then the mirrored lane of the 8-lane half row, which sits in the other quad and
// therefore holds the other quad's sum.  Half of a wave-wide sum; the caller accumulates the group sums per group and folds the
// eight groups ONCE per kernel (the backward scan's dA).
__device__ __forceinline__ f32x2 group8_sum2_dpp(f32x2 v) {
    float x = v[0], y = v[1];
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        : "+v"(x), "+v"(y));
    return f2(x, y);
}

// acc += x on the lanes with (lane & 7) == k, k = 0..7: exec = 0x01010101 << k in both halves (every lane of the wave is active on
// entry: wave-uniform trip counts)
__device__ __forceinline__ void add_on_lanes_mod8(f32x2& acc, f32x2 x, int k) {
    const uint32_t m = 0x01010101u << k;
    asm volatile(
        "s_mov_b32 exec_lo, %2\n\t"
        "s_mov_b32 exec_hi, %2\n\t"
        "v_pk_add_f32 %0, %0, %1\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(acc)
        : "v"(x), "s"(m));
}

// Two wave-wide sums at once, the two DPP chains interleaved by hand: a DPP read needs two wait states after the VALU
// write of its source, so one chain alone is padded with an s_nop before every step (and the compiler emits the two
// chains one after the other); interleaved, each chain's step fills the other's wait states.
__device__ __forceinline__ f32x2 wave_sum2_dpp(f32x2 v) {
    float x = v[0], y = v[1];
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        : "+v"(x), "+v"(y));
    return f2(cad_readlane(x, 63), cad_readlane(y, 63));
}

// The same step with the DPP shift folded into the arithmetic (VOP2 + DPP, 4 instructions for a float2 map instead of
// 4 v_mov + 4 v_mov_dpp + 2 v_pk): a lane whose DPP source does not exist (or whose row is masked off) is simply not
// written, which IS the identity.  fmac before mul: H needs the old A.  Four instructions separate every DPP read from
// the write of the same register (>= the 2 wait states the hardware requires); the leading s_nop covers the
// compiler-generated producer of the inputs.
#define SC_KS_ASM(CTRL)                          \
    "v_fmac_f32_dpp %0, %0, %2 " CTRL "\n\t"     \
    "v_fmac_f32_dpp %1, %1, %3 " CTRL "\n\t"     \
    "v_mul_f32_dpp %2, %2, %2 " CTRL "\n\t"      \
    "v_mul_f32_dpp %3, %3, %3 " CTRL "\n\t"

// Inclusive scan in lane order (lane 0 first).  On return (A, H) of lane j is the composition of lanes 0..j.
__device__ __forceinline__ void wave_scan_fwd(f32x2& A, f32x2& H) {
    float h0 = H[0], h1 = H[1], a0 = A[0], a1 = A[1];
    asm("s_nop 1\n\t"
        SC_KS_ASM("row_shr:1 row_mask:0xf bank_mask:0xf")
        SC_KS_ASM("row_shr:2 row_mask:0xf bank_mask:0xf")
        SC_KS_ASM("row_shr:4 row_mask:0xf bank_mask:0xf")
        SC_KS_ASM("row_shr:8 row_mask:0xf bank_mask:0xf")
        SC_KS_ASM("row_bcast:15 row_mask:0xa bank_mask:0xf")
        SC_KS_ASM("row_bcast:31 row_mask:0xc bank_mask:0xf")
        : "+v"(h0), "+v"(h1), "+v"(a0), "+v"(a1));
    H = f2(h0, h1);
    A = f2(a0, a1);
}

// The forward scan with the state entering lane 0 (`hin`, wave-uniform) folded into lane 0's map before the scan: on return H of
// lane j is the TRUE state leaving lane j; the map products are dead afterwards (no exclusive shift of A, no `A * hin + H` per
// lane, no A update in the last step).
__device__ __forceinline__ void wave_scan_fwd_carry(f32x2 A, f32x2& H, f32x2 hin, int lane) {
    // every lane of the wave is active here (wave-uniform trip counts): exec is restored to all ones
    asm volatile(
        "s_mov_b32 exec_lo, 1\n\t"
        "s_mov_b32 exec_hi, 0\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(H)
        : "v"(A), "s"(hin));
    float h0 = H[0], h1 = H[1], a0 = A[0], a1 = A[1];
    asm("s_nop 1\n\t"
        SC_KS_ASM("row_shr:1 row_mask:0xf bank_mask:0xf")
        SC_KS_ASM("row_shr:2 row_mask:0xf bank_mask:0xf")
        SC_KS_ASM("row_shr:4 row_mask:0xf bank_mask:0xf")
        SC_KS_ASM("row_shr:8 row_mask:0xf bank_mask:0xf")
        SC_KS_ASM("row_bcast:15 row_mask:0xa bank_mask:0xf")
        "v_fmac_f32_dpp %0, %0, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %1, %1, %3 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        : "+v"(h0), "+v"(h1), "+v"(a0), "+v"(a1));
    H = f2(h0, h1);
}

// Inclusive scan in REVERSE lane order (lane 63 first): (A, G) of lane j is the composition of lanes 63..j.
// Row-local steps use DPP row_shl; the two cross-row steps have no DPP broadcast in this direction and go through
// v_readlane + masked updates.
__device__ __forceinline__ void wave_scan_rev(f32x2& A, f32x2& G, int lane) {
    {
        float g0 = G[0], g1 = G[1], a0 = A[0], a1 = A[1];
        asm("s_nop 1\n\t"
            SC_KS_ASM("row_shl:1 row_mask:0xf bank_mask:0xf")
            SC_KS_ASM("row_shl:2 row_mask:0xf bank_mask:0xf")
            SC_KS_ASM("row_shl:4 row_mask:0xf bank_mask:0xf")
            SC_KS_ASM("row_shl:8 row_mask:0xf bank_mask:0xf")
            : "+v"(g0), "+v"(g1), "+v"(a0), "+v"(a1));
        G = f2(g0, g1);
        A = f2(a0, a1);
    }
    // after the row-local steps the FIRST lane of every row holds its whole row; fold the later rows in.  The totals
    // travel through SGPRs (v_readlane) and are applied under a lane mask: no LDS round trip (ds_bpermute costs
    // ~150 cycles of exposed latency twice per scan).
    {   // rows 0 and 2 <- total of the next row (lanes 16 / 48)
        const f32x2 a16 = readlane2(A, 16), g16 = readlane2(G, 16), a48 = readlane2(A, 48), g48 = readlane2(G, 48);
        const int row = lane >> 4;
        if (row == 0) {
            G = A * g16 + G;
            A = A * a16;
        }
        if (row == 2) {
            G = A * g48 + G;
            A = A * a48;
        }
    }
    {   // rows 0 and 1 <- total of rows 2..3 (now at lane 32)
        const f32x2 a32 = readlane2(A, 32), g32 = readlane2(G, 32);
        if (lane < 32) {
            G = A * g32 + G;
            A = A * a32;
        }
    }
}

// The same reverse scan with the carry `gin` (G flowing in behind lane 63, wave-uniform) folded into lane 63's map before
// the scan: on return G of lane j is the TRUE value flowing out of lane j (towards lane j - 1), the map products A are not
// needed afterwards (no exclusive shift of A, no `A * gin + G` per lane, no A update in the last step), and the cross-row
// steps run under an exec mask instead of compute-then-select (2 packed ops + 2 s_mov per step instead of 2 + 4 v_cndmask).
__device__ __forceinline__ void wave_scan_rev_carry(f32x2 A, f32x2& G, f32x2 gin, int lane) {
    // every lane of the wave is active here (the pair loop has a wave-uniform trip count): exec is restored to all ones
    asm volatile(
        "s_mov_b32 exec_lo, 0\n\t"
        "s_mov_b32 exec_hi, 0x80000000\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0\n\t"
        "s_mov_b64 exec, -1"
        : "+v"(G)
        : "v"(A), "s"(gin));
    {
        float g0 = G[0], g1 = G[1], a0 = A[0], a1 = A[1];
        asm("s_nop 1\n\t"
            SC_KS_ASM("row_shl:1 row_mask:0xf bank_mask:0xf")
            SC_KS_ASM("row_shl:2 row_mask:0xf bank_mask:0xf")
            SC_KS_ASM("row_shl:4 row_mask:0xf bank_mask:0xf")
            SC_KS_ASM("row_shl:8 row_mask:0xf bank_mask:0xf")
            : "+v"(g0), "+v"(g1), "+v"(a0), "+v"(a1));
        G = f2(g0, g1);
        A = f2(a0, a1);
    }
    {   // rows 0 and 2 <- total of the next row (lanes 16 / 48)
        const f32x2 a16 = readlane2(A, 16), g16 = readlane2(G, 16), a48 = readlane2(A, 48), g48 = readlane2(G, 48);
        asm volatile(
            "s_mov_b32 exec_lo, 0xffff\n\t"
            "s_mov_b32 exec_hi, 0\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0\n\t"
            "v_pk_mul_f32 %1, %1, %3\n\t"
            "s_mov_b32 exec_lo, 0\n\t"
            "s_mov_b32 exec_hi, 0xffff\n\t"
            "v_pk_fma_f32 %0, %1, %4, %0\n\t"
            "v_pk_mul_f32 %1, %1, %5\n\t"
            "s_mov_b64 exec, -1"
            : "+v"(G), "+v"(A)
            : "s"(g16), "s"(a16), "s"(g48), "s"(a48));
    }
    {   // rows 0 and 1 <- total of rows 2..3 (now at lane 32)
        const f32x2 g32 = readlane2(G, 32);
        asm volatile(
            "s_mov_b32 exec_hi, 0\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0\n\t"
            "s_mov_b64 exec, -1"
            : "+v"(G)
            : "v"(A), "s"(g32));
    }
}

// v_cndmask with a wave-uniform 64-bit lane mask in SGPRs.  Written in asm: a C++ select between two ELEMENTS of a vector
// (`rev ? v[S-1-j] : v[j]`) is canonicalised into a dynamic index = a chain of S-1 compares + selects per element
// (measured in the chunk epilogue of the backward: 119 v_cndmask + 56 s_cmp/s_cselect for two 8-item stores).
__device__ __forceinline__ uint32_t sc_sel(uint32_t if0, uint32_t if1, uint64_t mask) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(if0), "v"(if1), "s"(mask));
    return r;
}

__device__ __forceinline__ uint64_t sc_rev_mask(int rev) {  // all lanes set <=> right-to-left row (wave-uniform)
    const uint32_t r = __builtin_amdgcn_readfirstlane(rev ? ~0u : 0u);
    return ((uint64_t)r << 32) | r;
}

__device__ __forceinline__ uint32_t sc_rot(uint32_t x, uint32_t rot) {  // rotate right by rot bits (0 or 16 here)
    return __builtin_amdgcn_alignbit(x, x, rot);
}

// Asynchronous vector load into registers: issued through inline asm, so the compiler neither knows it is a load nor
// waits for it; sc_stage_wait() below is the matching s_waitcnt, placed by hand right before the data is consumed one
// pair-step later.  (A compiler-visible load is unpacked -- bf16 high halves -- right where it is issued, i.e. it is
// waited for immediately, which exposes the full L2 latency on the staging waves every pair-step.)  Loads the compiler
// issues itself stay correct: an unknown extra load in flight can only make its vmcnt waits longer, never too short.
template <typename V>
__device__ __forceinline__ void sc_async_load(V& dst, const void* p) {
    static_assert(sizeof(V) == 8 || sizeof(V) == 16 || sizeof(V) == 32, "vector sizes of the staging path");
    if constexpr (sizeof(V) == 8) {
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        u2 v;
        asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
        dst = __builtin_bit_cast(V, v);
    } else if constexpr (sizeof(V) == 16) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        u4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
        dst = __builtin_bit_cast(V, v);
    } else {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        struct P { u4 a, b; } v;
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16"
                     : "=&v"(v.a), "=&v"(v.b) : "v"(p) : "memory");
        dst = __builtin_bit_cast(V, v);
    }
}

template <typename V>
__device__ __forceinline__ void sc_async_wait(V& a, V& b) {
    if constexpr (sizeof(V) == 8) {
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        u2 x = __builtin_bit_cast(u2, a), y = __builtin_bit_cast(u2, b);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x), "+v"(y)::"memory");
        a = __builtin_bit_cast(V, x), b = __builtin_bit_cast(V, y);
    } else if constexpr (sizeof(V) == 16) {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        u4 x = __builtin_bit_cast(u4, a), y = __builtin_bit_cast(u4, b);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x), "+v"(y)::"memory");
        a = __builtin_bit_cast(V, x), b = __builtin_bit_cast(V, y);
    } else {
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        struct P { u4 a, b; };
        P x = __builtin_bit_cast(P, a), y = __builtin_bit_cast(P, b);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(x.a), "+v"(x.b), "+v"(y.a), "+v"(y.b)::"memory");
        a = __builtin_bit_cast(V, x), b = __builtin_bit_cast(V, y);
    }
}

template <typename V>
__device__ __forceinline__ void sc_async_wait_keep(V& a, V& b, bool keep_dma) {
    static_assert(sizeof(V) == 8 || sizeof(V) == 16, "vector sizes of the prefetching kernels");
    typedef uint32_t uw __attribute__((ext_vector_type(sizeof(V) / 4)));
    uw x = __builtin_bit_cast(uw, a), y = __builtin_bit_cast(uw, b);
    // ONE statement (the scalar branch on the wave-uniform flag is inside it): with the choice made in C++ the compiler
    // materialises the "+v" operands in one arm BEFORE the wait, i.e. copies registers that are still in flight
    // (caught by tests/test_isa_async.py)
    const uint32_t k = __builtin_amdgcn_readfirstlane(keep_dma ? 1u : 0u);
    asm volatile(
        "s_cmp_eq_u32 %2, 0\n\t"
        "s_cbranch_scc1 .Lsc_wait0_%=\n\t"
        "s_waitcnt vmcnt(6)\n\t"
        "s_branch .Lsc_waitd_%=\n"
        ".Lsc_wait0_%=:\n\t"
        "s_waitcnt vmcnt(0)\n"
        ".Lsc_waitd_%=:"
        : "+v"(x), "+v"(y)
        : "s"(k)
        : "memory", "scc");
    a = __builtin_bit_cast(V, x), b = __builtin_bit_cast(V, y);
}

// Wait until at most KEEP vector-memory operations are outstanding.  vmcnt retires in issue order on gfx9-class hardware
// (loads, LDS-DMA and stores alike -- the compiler's own counted waits rely on it), so with KEEP = the number of stores
// issued after the prefetch that may still be waiting for their write acknowledgement, the prefetch is known to have
// landed while those (recent) stores stay in flight.
template <int KEEP>
__device__ __forceinline__ void sc_wait_loads() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KEEP) : "memory");
}
