// genrec.hip — SURVEY §8f row f2 on packed rows: PauliwordOp.generators (symmer/operators/base.py:1436-1456), check_independent
// (utils.py:504-519) and PauliwordOp.generator_reconstruction (base.py:523-560 = cref_binary of vstack([G, M]), utils.py:317-359).
//
// The reference works on one-byte-per-bit matrices: `_rref_binary(symp_matrix)` for the generators and, for the reconstruction, the
// transposes `rref_binary(vstack([G, M]).T).T`.  Here nothing of that shape exists:
//  * generators / rank: the packed rows ARE the GF(2) matrix (X words then Z words; the zero padding bits can never become pivots and
//    the column order x_0 .. x_{n-1}, z_0 .. z_{n-1} is kept), reduced in place by the blocked elimination of gf2.hip;
//  * reconstruction: the transposed stack is built bit-packed on the device (one wavefront transposes a 64-term x 64-qubit tile with 64
//    ballots, as k_build_symmat does for the symmetry generators), reduced, and read out in pivot order straight into the reference's
//    return values — R as int64 [T][g] (`reduced[dim:, :dim].astype(int)`), the mask as one byte per term.
#include "common.h"
#include <vector>
#include <algorithm>

namespace symgpu {

// mat is (2n) x Wc, Wc = ceil((g + T) / 64): row c < n = [ x_c of G's terms | x_c of M's terms ], row n + c the same for z_c — the
// transpose of vstack([G.symp_matrix, M.symp_matrix]).
__global__ __launch_bounds__(256) void k_build_stack_t(const u64 *__restrict__ G, i64 g, const u64 *__restrict__ M, i64 T, int n, int Wq,
                                                       u64 *__restrict__ mat, i64 Wc) {
    const int lane = threadIdx.x & 63;
    const i64 tile = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);   // 64-column tile of the stack
    const int sw = blockIdx.y;                                   // source word 0 .. 2Wq-1
    if (tile >= Wc) return;
    const i64 j = tile * 64 + lane;
    u64 word = 0;
    if (j < g) word = G[j * 2 * Wq + sw];
    else if (j - g < T) word = M[(j - g) * 2 * Wq + sw];
    u64 mine = 0;
    for (int b = 0; b < 64; ++b) {
        const u64 m = __ballot((word >> b) & 1ULL);
        if (lane == b) mine = m;
    }
    const int q = 64 * (sw % Wq) + lane;
    if (q < n) {
        const i64 c = (sw < Wq) ? q : (i64)n + q;                // X words feed rows 0 .. n-1, Z words rows n .. 2n-1
        mat[c * Wc + tile] = mine;
    }
}

// orw[w] = OR of the rows whose position in rref_binary's row order is >= g (tail[r] != 0)
__global__ __launch_bounds__(256) void k_or_tail_rows(const u64 *__restrict__ mat, i64 R, i64 Wc, const uint8_t *__restrict__ tail, u64 *__restrict__ orw) {
    const i64 w = (i64)blockIdx.x * 256 + threadIdx.x;
    if (w >= Wc) return;
    u64 acc = 0;
    for (i64 r = 0; r < R; ++r)
        if (tail[r]) acc |= mat[r * Wc + w];
    orw[w] = acc;
}

__global__ __launch_bounds__(256) void k_recon_mask(const u64 *__restrict__ orw, i64 g, i64 T, uint8_t *__restrict__ mask) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const i64 j = g + t;
    mask[t] = (uint8_t)(((orw[j >> 6] >> (j & 63)) & 1ULL) ^ 1ULL);          // reconstructed <=> no row beyond the first g touches column g + t
}

// out[t][k] = bit (g + t) of the k-th row in pivot order, k < g: one wavefront per (64 terms) x (64 generators) tile, lanes over k
__global__ __launch_bounds__(256) void k_recon_out(const u64 *__restrict__ mat, i64 Wc, const int *__restrict__ ord, i64 g, i64 T, i64 *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const i64 tt = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    const i64 k = (i64)blockIdx.y * 64 + lane;
    if (tt * 64 >= T) return;
    u64 word = 0;
    if (k < g) {
        const i64 r = ord[k];
        const i64 j0 = g + tt * 64, w0 = j0 >> 6;
        const int sh = (int)(j0 & 63);
        const u64 lo = mat[r * Wc + w0], hi = (w0 + 1 < Wc) ? mat[r * Wc + w0 + 1] : 0ULL;
        word = sh ? ((lo >> sh) | (hi << (64 - sh))) : lo;
    }
    const i64 t_end = (T - tt * 64 < 64) ? T - tt * 64 : 64;
    if (k < g)
        for (int b = 0; b < (int)t_end; ++b) out[(tt * 64 + b) * g + k] = (i64)((word >> b) & 1ULL);
}

__global__ __launch_bounds__(256) void k_gather_ones(const u64 *__restrict__ rows, int W, const i64 *__restrict__ idx, i64 k, u64 *__restrict__ out_rows,
                                                     double *__restrict__ out_coeff) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i < k * W) out_rows[i] = rows[idx[i / W] * W + (i % W)];
    if (i < k) { out_coeff[2 * i] = 1.0; out_coeff[2 * i + 1] = 0.0; }
}

// reduce a copy of the operator's packed rows; pivots (host, [T]) tell which rows stay non-zero
static int reduce_rows_copy(symgpu_op_s *op, Scratch &work, std::vector<i64> &piv) {
    const i64 T = op->T;
    const int W = 2 * op->Wq;
    SG_TRY(work.alloc((size_t)T * W * 8));
    HIP_TRY(hipMemcpyAsync(work.p, op->rows, (size_t)T * W * 8, hipMemcpyDeviceToDevice, ctx().stream));
    piv.assign((size_t)T, -1);
    return rref_dev(work.as<u64>(), T, W, nullptr, piv.data());
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_op_gf2_rank(symgpu_op_t op, int64_t *rank) {
    SG_ENTER(op);
    SG_REQUIRE(op && rank, "op_gf2_rank: null argument");
    *rank = 0;
    if (op->T == 0) return SYMGPU_OK;
    Scratch work;
    std::vector<i64> piv;
    SG_TRY(reduce_rows_copy(op, work, piv));
    i64 r = 0;
    for (i64 p : piv) r += p >= 0;
    *rank = r;
    return SYMGPU_OK;
}

int symgpu_generators_dev(symgpu_op_t op, symgpu_op_t *out) {
    SG_ENTER(op);
    SG_REQUIRE(op && out, "generators_dev: null argument");
    const int W = 2 * op->Wq;
    hipStream_t st = ctx().stream;
    std::vector<i64> piv, keep;
    Scratch work;
    if (op->T > 0) SG_TRY(reduce_rows_copy(op, work, piv));
    for (i64 r = 0; r < (i64)piv.size(); ++r)
        if (piv[r] >= 0) keep.push_back(r);                       // a row that is non-zero after the reduction owns a pivot, and vice versa
    const i64 k = (i64)keep.size();
    symgpu_op_t res = nullptr;
    SG_TRY(symgpu_op_alloc(k > 0 ? k : 1, op->Wq, 1, &res));
    res->T = k;
    if (k > 0) {
        Scratch idx;
        int rc = idx.alloc((size_t)k * 8);
        if (rc != SYMGPU_OK) { symgpu_op_free(res); return rc; }
        hipError_t e = hipMemcpyAsync(idx.p, keep.data(), (size_t)k * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_gather_ones, dim3((unsigned)((k * W + 255) / 256)), dim3(256), 0, st, work.as<u64>(), W, idx.as<i64>(), k, res->rows, res->coeff);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);          // `keep` and `idx` end here
        if (e != hipSuccess) { symgpu_op_free(res); return hip_fail(e, "generators_dev", __FILE__, __LINE__); }
    }
    res->dup_free = 1;                                              // reduced rows with distinct pivots are distinct
    *out = res;
    return SYMGPU_OK;
}

int symgpu_generator_reconstruction_dev(symgpu_op_t G, symgpu_op_t M, int n_qubits, int64_t *recon_host, uint8_t *mask_host) {
    SG_ENTER(G, M);
    SG_REQUIRE(G && M && n_qubits >= 1, "generator_reconstruction_dev: null argument");
    SG_REQUIRE(G->Wq == M->Wq && (n_qubits + 63) / 64 == G->Wq, "generator_reconstruction_dev: operands must share the qubit count");
    const i64 g = G->T, T = M->T;
    SG_REQUIRE(g >= 1 && T >= 1 && recon_host && mask_host, "generator_reconstruction_dev: needs at least one generator and one term");
    // with g > 2n the reference's reduced[dim:, :dim] has 2n columns only — not a case a set of generators can be in
    SG_REQUIRE(g <= 2 * (i64)n_qubits, "generator_reconstruction_dev: more generators than symplectic columns");
    hipStream_t st = ctx().stream;
    const int n = n_qubits, Wq = G->Wq;
    const i64 R = 2 * (i64)n, Wc = (g + T + 63) / 64;
    Scratch mat;
    SG_TRY(mat.alloc((size_t)R * Wc * 8));
    HIP_TRY(hipMemsetAsync(mat.p, 0, (size_t)R * Wc * 8, st));
    hipLaunchKernelGGL(k_build_stack_t, dim3((unsigned)((Wc + 3) / 4), (unsigned)(2 * Wq)), dim3(256), 0, st, G->rows, g, M->rows, T, n, Wq, mat.as<u64>(), Wc);
    KERNEL_CHECK();
    std::vector<i64> piv((size_t)R, -1);
    SG_TRY(rref_dev(mat.as<u64>(), R, Wc, nullptr, piv.data()));
    // rref_binary's row order (utils.py:328-335): rows that own a pivot by pivot column, then the zero rows (all equal: their order is immaterial)
    std::vector<int> ord;
    for (i64 r = 0; r < R; ++r) if (piv[r] >= 0) ord.push_back((int)r);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return piv[a] < piv[b]; });
    for (i64 r = 0; r < R; ++r) if (piv[r] < 0) ord.push_back((int)r);
    std::vector<uint8_t> tail((size_t)R, 0);
    for (i64 k = g; k < R; ++k) tail[ord[k]] = 1;
    Scratch d_ord, d_tail, orw, d_mask, d_out;
    SG_TRY(d_ord.alloc((size_t)R * 4));
    SG_TRY(d_tail.alloc((size_t)R));
    SG_TRY(orw.alloc((size_t)Wc * 8));
    SG_TRY(d_mask.alloc((size_t)T));
    SG_TRY(d_out.alloc((size_t)T * g * 8));
    HIP_TRY(hipMemcpyAsync(d_ord.p, ord.data(), (size_t)R * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_tail.p, tail.data(), (size_t)R, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_or_tail_rows, dim3((unsigned)((Wc + 255) / 256)), dim3(256), 0, st, mat.as<u64>(), R, Wc, d_tail.as<uint8_t>(), orw.as<u64>());
    hipLaunchKernelGGL(k_recon_mask, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, orw.as<u64>(), g, T, d_mask.as<uint8_t>());
    hipLaunchKernelGGL(k_recon_out, dim3((unsigned)(((T + 63) / 64 + 3) / 4), (unsigned)((g + 63) / 64)), dim3(256), 0, st, mat.as<u64>(), Wc, d_ord.as<int>(), g, T,
                       d_out.as<i64>());
    KERNEL_CHECK();
    prefault_host(recon_host, (size_t)T * g * 8);
    HIP_TRY(hipMemcpyAsync(recon_host, d_out.p, (size_t)T * g * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(mask_host, d_mask.p, (size_t)T, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));                              // `ord` / `tail` end here
    count_d2h((size_t)T * g * 8 + (size_t)T);
    return SYMGPU_OK;
}

}  // extern "C"
