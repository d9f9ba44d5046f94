// Streaming stores for the HBM-bound kernels around the scans (projections, conv1d, add + norm).
// Their outputs -- hundreds of MB per launch -- are written once and read by a LATER kernel; written with ordinary stores they
// displace the lines the kernels are about to read.  Nontemporal stores, measured on the MI355X (profiles/r03_ab_nt_stores.txt):
// in_proj 0.191 -> 0.151 ms (3.5 -> 4.5 TB/s), d(y) 0.100 -> 0.084 ms stand-alone; a whole training step 131.4 -> 129.8 ms with
// all three families on (same box).  The scans keep ordinary stores (no gain there, same file) and do not include this header.
// CAD_NT_MASK (tuning builds only): bit 0 projections, bit 1 conv1d, bit 2 add + norm.
#pragma once
#include "cad_common.h"

#ifndef CAD_NT_MASK
#define CAD_NT_MASK 7
#endif
#define CAD_STREAM_PROJ 0
#define CAD_STREAM_CONV 1
#define CAD_STREAM_NORM 2

template <int FAMILY, typename V>
__device__ __forceinline__ void cad_store_stream(V* p, V v) {
    if constexpr ((CAD_NT_MASK >> FAMILY) & 1)
        cad_nt_store(p, v);
    else
        *p = v;
}
// N fp32 values -> N contiguous elements of T at dst (as cad_cvt_store), streamed
template <int FAMILY, typename T, int N>
__device__ __forceinline__ void cad_cvt_store_stream(T* dst, const float* v);
template <>
__device__ __forceinline__ void cad_cvt_store_stream<CAD_STREAM_NORM, float, 4>(float* dst, const float* v) {
    cad_store_stream<CAD_STREAM_NORM>((f32x4*)dst, f32x4{v[0], v[1], v[2], v[3]});
}
template <>
__device__ __forceinline__ void cad_cvt_store_stream<CAD_STREAM_NORM, bf16_t, 4>(bf16_t* dst, const float* v) {
    cad_store_stream<CAD_STREAM_NORM>((u32x2*)dst, u32x2{cad_pack_bf16x2(v[0], v[1]), cad_pack_bf16x2(v[2], v[3])});
}
