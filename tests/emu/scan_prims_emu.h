// Host-emulator restatement of caduceus_amd/csrc/scan_prims_gfx950.h (same names, plain C++ on the fiber lanes).  TEST INFRASTRUCTURE.
#pragma once

__device__ __forceinline__ f32x2 splat_lo(f32x2 v) {
    return f2(v[0]);
}

__device__ __forceinline__ f32x2 splat_hi(f32x2 v) {
    return f2(v[1]);
}

__device__ __forceinline__ void pk_fma_acc(f32x2& acc, f32x2 a, f32x2 b) {
    acc = a * b + acc;
}

__device__ __forceinline__ float dot2_acc(float acc, f32x2 a, f32x2 b) {
    return __builtin_fmaf(a[1], b[1], __builtin_fmaf(a[0], b[0], acc));
}

__device__ __forceinline__ float dot2_first(f32x2 a, f32x2 b) {
    return __builtin_fmaf(a[1], b[1], a[0] * b[0]);
}

__device__ __forceinline__ f32x2 group8_sum2_dpp(f32x2 v) {  // butterfly over lane ^ 1, lane ^ 2, then the mirrored lane of the 8-lane half row
    const int lane = emu::lane_id();
    for (int c = 0; c < 2; ++c) {
        float x = v[c];
        x = emu_exchange(x, lane ^ 1) + x;
        x = emu_exchange(x, lane ^ 2) + x;
        x = emu_exchange(x, (lane & ~7) | (7 - (lane & 7))) + x;
        v[c] = x;
    }
    return v;
}

__device__ __forceinline__ void add_on_lanes_mod8(f32x2& acc, f32x2 x, int k) {
    if ((emu::lane_id() & 7) == k) acc = acc + x;
}

__device__ __forceinline__ f32x2 wave_sum2_dpp(f32x2 v) {
    return f2(wave_sum_dpp(v[0]), wave_sum_dpp(v[1]));
}

__device__ __forceinline__ void wave_scan_fwd(f32x2& A, f32x2& H) {
    SC_COMBINE(A, H, dpp_row_shr<1>(1.f, A[0]), dpp_row_shr<1>(1.f, A[1]), dpp_row_shr<1>(0.f, H[0]), dpp_row_shr<1>(0.f, H[1]));
    SC_COMBINE(A, H, dpp_row_shr<2>(1.f, A[0]), dpp_row_shr<2>(1.f, A[1]), dpp_row_shr<2>(0.f, H[0]), dpp_row_shr<2>(0.f, H[1]));
    SC_COMBINE(A, H, dpp_row_shr<4>(1.f, A[0]), dpp_row_shr<4>(1.f, A[1]), dpp_row_shr<4>(0.f, H[0]), dpp_row_shr<4>(0.f, H[1]));
    SC_COMBINE(A, H, dpp_row_shr<8>(1.f, A[0]), dpp_row_shr<8>(1.f, A[1]), dpp_row_shr<8>(0.f, H[0]), dpp_row_shr<8>(0.f, H[1]));
    SC_COMBINE(A, H, dpp_row_bcast15(1.f, A[0]), dpp_row_bcast15(1.f, A[1]), dpp_row_bcast15(0.f, H[0]), dpp_row_bcast15(0.f, H[1]));
    SC_COMBINE(A, H, dpp_row_bcast31(1.f, A[0]), dpp_row_bcast31(1.f, A[1]), dpp_row_bcast31(0.f, H[0]), dpp_row_bcast31(0.f, H[1]));
}

__device__ __forceinline__ void wave_scan_fwd_carry(f32x2 A, f32x2& H, f32x2 hin, int lane) {
    if (lane == 0) H = A * hin + H;
    wave_scan_fwd(A, H);
}

__device__ __forceinline__ void wave_scan_rev(f32x2& A, f32x2& G, int lane) {
    SC_COMBINE(A, G, dpp_row_shl<1>(1.f, A[0]), dpp_row_shl<1>(1.f, A[1]), dpp_row_shl<1>(0.f, G[0]), dpp_row_shl<1>(0.f, G[1]));
    SC_COMBINE(A, G, dpp_row_shl<2>(1.f, A[0]), dpp_row_shl<2>(1.f, A[1]), dpp_row_shl<2>(0.f, G[0]), dpp_row_shl<2>(0.f, G[1]));
    SC_COMBINE(A, G, dpp_row_shl<4>(1.f, A[0]), dpp_row_shl<4>(1.f, A[1]), dpp_row_shl<4>(0.f, G[0]), dpp_row_shl<4>(0.f, G[1]));
    SC_COMBINE(A, G, dpp_row_shl<8>(1.f, A[0]), dpp_row_shl<8>(1.f, A[1]), dpp_row_shl<8>(0.f, G[0]), dpp_row_shl<8>(0.f, G[1]));
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

__device__ __forceinline__ void wave_scan_rev_carry(f32x2 A, f32x2& G, f32x2 gin, int lane) {
    if (lane == 63) G = A * gin + G;
    wave_scan_rev(A, G, lane);
}

__device__ __forceinline__ uint32_t sc_sel(uint32_t if0, uint32_t if1, uint64_t mask) {
    return mask ? if1 : if0;
}

__device__ __forceinline__ uint64_t sc_rev_mask(int rev) {  // all lanes set <=> right-to-left row (wave-uniform)
    return rev ? ~0ull : 0ull;
}

__device__ __forceinline__ uint32_t sc_rot(uint32_t x, uint32_t rot) {  // rotate right by rot bits (0 or 16 here)
    return rot ? ((x >> rot) | (x << (32 - rot))) : x;
}

template <typename V>
__device__ __forceinline__ void sc_async_load(V& dst, const void* p) {
    dst = *(const V*)p;
}

template <typename V>
__device__ __forceinline__ void sc_async_wait(V& a, V& b) {
}

template <typename V>
__device__ __forceinline__ void sc_async_wait_keep(V& a, V& b, bool keep_dma) {
}

template <int KEEP>
__device__ __forceinline__ void sc_wait_loads() {
}
