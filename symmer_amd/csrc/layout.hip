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

int to_wordmajor(const u64 *rows, i64 T, int W, u64 *out, i64 Tpad, hipStream_t st) {
    if (Tpad <= 0) return SYMGPU_OK;
    dim3 grid((unsigned)((Tpad + TT - 1) / TT), (unsigned)((W + TW - 1) / TW));
    hipLaunchKernelGGL(k_to_wordmajor, grid, dim3(256), 0, st ? st : ctx().stream, rows, T, W, out, Tpad);
    KERNEL_CHECK();
    return SYMGPU_OK;
}

static void drop_wordmajor(symgpu_op_s *op) {
    if (op && op->wm) { dev_free(op->wm); op->wm = nullptr; }
    if (op) { op->wm_pad = 0; op->wm_T = -1; }
}

void op_invalidate(symgpu_op_s *op) {
    drop_wordmajor(op);
    if (!op) return;
    op->dup_free = 0;
    if (op->hash) { dev_free(op->hash); op->hash = nullptr; }
    op->hash_seed = 0;
    if (op->bt) { dev_free(op->bt); op->bt = nullptr; }
    op->bt_pad = 0; op->bt_T = -1;
    if (op->yc) { dev_free(op->yc); op->yc = nullptr; }
    op->yc_T = -1;
    if (op->first) { dev_free(op->first); op->first = nullptr; }
}

int op_ycount(symgpu_op_s *op, const int **out) {
    if (!op->yc || op->yc_T != op->T) {
        if (op->yc) { dev_free(op->yc); op->yc = nullptr; }
        op->yc_T = -1;
        SG_TRY(dev_alloc((size_t)(op->T > 0 ? op->T : 1) * sizeof(int), (void **)&op->yc));
        SG_TRY(ycount_dev(op->rows, op->T, op->Wq, op->yc));
        op->yc_T = op->T;
    }
    *out = op->yc;
    return SYMGPU_OK;
}

int op_wordmajor(symgpu_op_s *op, i64 mult, const u64 **out, i64 *pad) {
    const i64 need = (op->T + mult - 1) / mult * mult;
    if (!op->wm || op->wm_T != op->T || op->wm_pad % mult != 0 || op->wm_pad < need) {
        drop_wordmajor(op);
        // pad to a multiple of 256 as well so that every all-pairs kernel can share the copy
        i64 m = mult;
        while (m % 256) m *= 2;
        const i64 p = (op->T + m - 1) / m * m;
        SG_TRY(dev_alloc((size_t)(p > 0 ? p : m) * 2 * op->Wq * sizeof(u64), (void **)&op->wm));
        SG_TRY(to_wordmajor(op->rows, op->T, 2 * op->Wq, op->wm, p));
        op->wm_pad = p;
        op->wm_T = op->T;
    }
    *out = op->wm;
    *pad = op->wm_pad;
    return SYMGPU_OK;
}

}  // namespace symgpu
