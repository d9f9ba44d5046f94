// fp32 tile image of the scans' B / C operands (include/caduceus_hip.h: cad_scan_bc_tiles; layout in scan_common.h).
// Written once per layer from the x_proj output and read by both scan kernels through LDS-DMA: the bf16 -> fp32 conversion, the
// direction map and the (lane, item, state) interleave are done here, ONCE per element, instead of once per workgroup and pair-step
// on the staging waves of every scan workgroup (64 workgroups per row re-did them).
#include "scan_common.h"

namespace {

#define IMG_CHUNK 512   // positions per image chunk (= the backward's chunk in the production build)
#define IMG_ITEMS 8     // positions per image lane

template <typename T>
__global__ __launch_bounds__(256) void bc_tiles_kernel(cad_bc_tiles_args a) {
    const int64_t NC = (a.L + IMG_CHUNK - 1) / IMG_CHUNK;
    const int NP = (a.N + 1) >> 1;
    const int64_t total = a.SB * NC * NP * 2 * 256;  // 16-byte elements
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx & 63), piece = (int)((idx >> 6) & 3), ten = (int)((idx >> 8) & 1);
        int64_t r = idx >> 9;
        const int np = (int)(r % NP);
        r /= NP;
        const int64_t c = r % NC, sb = r / NC;
        const int rev = sb < a.split ? a.rev_lo : a.rev_hi;
        const T* src = (const T*)(ten ? a.Cm : a.Bm);
        const int n0 = 2 * np;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // e = 2 * (item within the piece) + state
            const int64_t p = c * IMG_CHUNK + (int64_t)lane * IMG_ITEMS + 2 * piece + (e >> 1);
            const int n = n0 + (e & 1);
            v[e] = (p < a.L && n < a.N) ? to_f32(src[((int64_t)n * a.SB + sb) * a.L + cad_phys(p, a.L, rev)]) : 0.f;
        }
        struct __attribute__((aligned(16))) V { float f[4]; } o = {{v[0], v[1], v[2], v[3]}};
        *(V*)(a.tiles + idx * 4) = o;
    }
}

}  // namespace

extern "C" int64_t cad_scan_bc_tiles_floats(int64_t SB, int64_t L, int N) {
    const int64_t NC = (L + IMG_CHUNK - 1) / IMG_CHUNK;
    return SB * NC * ((N + 1) / 2) * 2 * SC_IMG_TILE;
}

extern "C" int cad_scan_bc_tiles(const cad_bc_tiles_args* a, void* stream) {
    CAD_CHECK_ARG(a && a->Bm && a->Cm && a->tiles && a->SB > 0 && a->L > 0 && a->N > 0 && a->N <= SC_NMAX);
    CAD_CHECK_ARG(a->split >= 0 && a->split <= a->SB && ((uintptr_t)a->tiles % 16) == 0);
    const int64_t total = cad_scan_bc_tiles_floats(a->SB, a->L, a->N) / 4;
    int64_t nb = (total + 255) / 256;
    if (nb > 65536) nb = 65536;
    dim3 grid((unsigned)nb), block(256);
    CadProfScope prof(0, stream);
    if (a->dtype == CAD_BF16)
        CAD_LAUNCH((bc_tiles_kernel<bf16_t>), grid, block, 0, stream, *a);
    else if (a->dtype == CAD_F32)
        CAD_LAUNCH((bc_tiles_kernel<float>), grid, block, 0, stream, *a);
    else
        return CAD_ERR_UNSUPPORTED;
    return cad_after_launch();
}
