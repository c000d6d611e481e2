// layout.hip — operand re-layout kernels.
//
// The C-ABI hands over row-major packed rows ([T][W] words).  The all-pairs kernels (commutation,
// product phase) want the *word-major* layout out[w][t]: for a fixed word index w the 64 lanes of a
// wavefront read 64 consecutive terms with one coalesced load, and 8 consecutive terms of the other
// operand arrive in SGPRs with a single s_load_dwordx16.
#include "common.h"

namespace symgpu {

constexpr int TT = 64;  // terms per tile
constexpr int TW = 32;  // words per tile

__global__ __launch_bounds__(256) void k_to_wordmajor(const u64 *__restrict__ rows, i64 T, int W, u64 *__restrict__ out, i64 Tpad) {
    __shared__ u64 tile[TW][TT + 1];
    const i64 t0 = (i64)blockIdx.x * TT;
    const int w0 = blockIdx.y * TW;
    for (int idx = threadIdx.x; idx < TT * TW; idx += 256) {
        int tl = idx / TW, wl = idx % TW;
        i64 t = t0 + tl;
        int w = w0 + wl;
        tile[wl][tl] = (t < T && w < W) ? rows[t * W + w] : 0ULL;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < TT * TW; idx += 256) {
        int wl = idx / TT, tl = idx % TT;
        i64 t = t0 + tl;
        int w = w0 + wl;
        if (w < W && t < Tpad) out[(i64)w * Tpad + t] = tile[wl][tl];
    }
}

int to_wordmajor(const u64 *rows, i64 T, int W, u64 *out, i64 Tpad) {
    if (Tpad <= 0) return SYMGPU_OK;
    dim3 grid((unsigned)((Tpad + TT - 1) / TT), (unsigned)((W + TW - 1) / TW));
    hipLaunchKernelGGL(k_to_wordmajor, grid, dim3(256), 0, ctx().stream, rows, T, W, out, Tpad);
    KERNEL_CHECK();
    return SYMGPU_OK;
}

}  // namespace symgpu
