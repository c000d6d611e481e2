// partition.hip — device side of the hash-partitioned multi-GPU cleanup (symmer_amd/parallel.py, SURVEY.md §8e "cleanup across GPUs").
//
// A rank owns the pairs whose product row falls in its GF(2)-linear class: a union of full sub-products (inner rows of class a) x (outer
// rows whose class maps to the rank).  Per sub-product: symgpu_op_gather picks the sub-operands out of the complete operands,
// symgpu_mul_cleanup_indexed_dev (cleanup.hip) multiplies and cleans them and keeps the first pair (o << 32 | i, sub-operand numbering) of
// every output row, symgpu_part_global_index turns that into the reference's pair index o_global * Ni + i_global (base.py:783-792).
// symgpu_merge_indexed_dev then concatenates the parts, orders them by pair index (radix sort of the 8-byte indices, rows gathered once),
// merges the rows that several parts share (first-occurrence cleanup: duplicates across the sub-products of one rank) and applies the
// threshold — the result carries the pair index of each term's first occurrence again, so the same call (without cleanup) puts the
// all-gathered shares of all ranks into the reference's order (utils.py:271).  Nothing leaves the device.
#include "common.h"
#include <stdlib.h>
#include <vector>

namespace symgpu {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

static int grid_of(i64 n, int block = 256) {
    i64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > (1 << 30)) g = 1 << 30;
    return (int)g;
}

// out row r = in row idx[r]: one 16-byte chunk per thread
__global__ __launch_bounds__(256) void k_gather_rows(const u32x4 *__restrict__ rows, const double *__restrict__ coeff, const i64 *__restrict__ idx, i64 n, int Wq,
                                                      u32x4 *__restrict__ out_rows, double *__restrict__ out_coeff) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * Wq) return;
    const i64 r = t / Wq;
    const int c = (int)(t - r * Wq);
    const i64 s = idx[r];
    __builtin_nontemporal_store(rows[s * Wq + c], out_rows + t);
    if (c == 0 && coeff) reinterpret_cast<f64x2 *>(out_coeff)[r] = reinterpret_cast<const f64x2 *>(coeff)[s];
}
__global__ __launch_bounds__(256) void k_gather_rows_u32(const u32x4 *__restrict__ rows, const double *__restrict__ coeff, const u32 *__restrict__ pos, i64 n, int Wq,
                                                          u32x4 *__restrict__ out_rows, double *__restrict__ out_coeff) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n * Wq) return;
    const i64 r = t / Wq;
    const int c = (int)(t - r * Wq);
    const i64 s = pos[r];
    __builtin_nontemporal_store(rows[s * Wq + c], out_rows + t);
    if (c == 0) reinterpret_cast<f64x2 *>(out_coeff)[r] = reinterpret_cast<const f64x2 *>(coeff)[s];
}
// first[t] = (o << 32 | i) in sub-operand numbering  ->  outer_idx[o] * Ni + inner_idx[i]
__global__ __launch_bounds__(256) void k_global_index(u64 *__restrict__ first, i64 n, const i64 *__restrict__ inner_idx, const i64 *__restrict__ outer_idx, i64 Ni) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const u64 f = first[t];
    first[t] = (u64)(outer_idx[f >> 32] * Ni + inner_idx[f & 0xFFFFFFFFULL]);
}
__global__ __launch_bounds__(256) void k_iota_u32(u32 *__restrict__ v, i64 n) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t < n) v[t] = (u32)t;
}
// out[t] = keys[pos[t]]
__global__ __launch_bounds__(256) void k_pick_u64(const u64 *__restrict__ keys, const u64 *__restrict__ pos, i64 n, u64 *__restrict__ out) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t < n) out[t] = keys[pos[t]];
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_op_gather(symgpu_op_t op, const int64_t *idx_host, int64_t n, symgpu_op_t *out) {
    SG_ENTER(op);
    SG_REQUIRE(op && out && n >= 0 && (idx_host || n == 0), "op_gather: arguments");
    for (i64 r = 0; r < n; ++r) SG_REQUIRE(idx_host[r] >= 0 && idx_host[r] < op->T, "op_gather: index out of range");
    hipStream_t st = ctx().stream;
    symgpu_op_t res = nullptr;
    SG_TRY(symgpu_op_alloc(n > 0 ? n : 1, op->Wq, op->coeff != nullptr, &res));
    res->T = n;
    if (n > 0) {
        Scratch idx;
        int rc = idx.alloc((size_t)n * 8);
        if (rc != SYMGPU_OK) { symgpu_op_free(res); return rc; }
        hipError_t e = hipMemcpyAsync(idx.p, idx_host, (size_t)n * 8, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_gather_rows, dim3(grid_of(n * op->Wq)), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(op->rows), op->coeff, idx.as<i64>(), n, op->Wq,
                               reinterpret_cast<u32x4 *>(res->rows), res->coeff);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);          // idx_host is the caller's; idx goes back to the allocator
        if (e != hipSuccess) { symgpu_op_free(res); return hip_fail(e, "op_gather", __FILE__, __LINE__); }
    }
    *out = res;
    return SYMGPU_OK;
}

int symgpu_part_global_index(symgpu_op_t part, const int64_t *inner_idx_host, int64_t n_inner, const int64_t *outer_idx_host, int64_t n_outer, int64_t Ni_global) {
    SG_ENTER(part);
    SG_REQUIRE(part && inner_idx_host && outer_idx_host && n_inner >= 1 && n_outer >= 1 && Ni_global >= 1, "part_global_index: arguments");
    SG_REQUIRE(part->first || part->T == 0, "part_global_index: the operator does not come from symgpu_mul_cleanup_indexed_dev");
    if (part->T == 0) return SYMGPU_OK;
    hipStream_t st = ctx().stream;
    Scratch ii, oi;
    SG_TRY(ii.alloc((size_t)n_inner * 8));
    SG_TRY(oi.alloc((size_t)n_outer * 8));
    HIP_TRY(hipMemcpyAsync(ii.p, inner_idx_host, (size_t)n_inner * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(oi.p, outer_idx_host, (size_t)n_outer * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_global_index, dim3(grid_of(part->T)), dim3(256), 0, st, part->first, part->T, ii.as<i64>(), oi.as<i64>(), Ni_global);
    KERNEL_CHECK();
    HIP_TRY(hipStreamSynchronize(st));
    return SYMGPU_OK;
}

int symgpu_op_set_first_index(symgpu_op_t op, const uint64_t *first_host) {
    SG_ENTER(op);
    SG_REQUIRE(op && (first_host || op->T == 0), "op_set_first_index: arguments");
    if (!op->first) SG_TRY(dev_alloc((size_t)(op->capacity > 0 ? op->capacity : 1) * 8, (void **)&op->first));
    if (op->T > 0) {
        HIP_TRY(hipMemcpyAsync(op->first, first_host, (size_t)op->T * 8, hipMemcpyHostToDevice, ctx().stream));
        count_h2d((size_t)op->T * 8);
        HIP_TRY(hipStreamSynchronize(ctx().stream));
    }
    return SYMGPU_OK;
}

int symgpu_merge_indexed_dev(const symgpu_op_t *parts, int n_parts, int key_bits, int do_cleanup, double thr, int use_thr, symgpu_op_t *out) {
    SG_ENTER((parts && n_parts > 0) ? parts[0] : nullptr);
    SG_REQUIRE(parts && out && n_parts >= 1 && key_bits >= 0 && key_bits <= 64, "merge_indexed_dev: arguments");
    const int sort_bits = key_bits == 0 ? 64 : (key_bits + 7) / 8 * 8;
    hipStream_t st = ctx().stream;
    const int Wq = parts[0]->Wq, W = 2 * Wq;
    i64 total = 0;
    for (int p = 0; p < n_parts; ++p) {
        SG_REQUIRE(parts[p] && parts[p]->Wq == Wq && (parts[p]->T == 0 || (parts[p]->first && parts[p]->coeff)), "merge_indexed_dev: parts must be indexed operators of one width");
        total += parts[p]->T;
    }
    SG_REQUIRE(total < ((i64)1 << 32) - 1, "merge_indexed_dev: too many rows");
    *out = nullptr;
    symgpu_op_t sorted = nullptr;
    SG_TRY(symgpu_op_alloc(total > 0 ? total : 1, Wq, 1, &sorted));
    sorted->T = total;
    if (total == 0) { sorted->dup_free = 1; *out = sorted; return SYMGPU_OK; }
    // concatenation (rows, coefficients, keys) and its order by key
    Scratch cat_rows, cat_coeff, keys, keys2, pos, pos2;
    int rc = cat_rows.alloc((size_t)total * W * 8);
    if (rc == SYMGPU_OK) rc = cat_coeff.alloc((size_t)total * 16);
    if (rc == SYMGPU_OK) rc = keys.alloc((size_t)total * 8);
    if (rc == SYMGPU_OK) rc = keys2.alloc((size_t)total * 8);
    if (rc == SYMGPU_OK) rc = pos.alloc((size_t)total * 4);
    if (rc == SYMGPU_OK) rc = pos2.alloc((size_t)total * 4);
    if (rc != SYMGPU_OK) { symgpu_op_free(sorted); return rc; }
    hipError_t e = hipSuccess;
    i64 at = 0;
    for (int p = 0; p < n_parts && e == hipSuccess; ++p) {
        const i64 T = parts[p]->T;
        if (T == 0) continue;
        e = hipMemcpyAsync(cat_rows.as<u64>() + at * W, parts[p]->rows, (size_t)T * W * 8, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(cat_coeff.as<double>() + 2 * at, parts[p]->coeff, (size_t)T * 16, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(keys.as<u64>() + at, parts[p]->first, (size_t)T * 8, hipMemcpyDeviceToDevice, st);
        at += T;
    }
    if (e != hipSuccess) { symgpu_op_free(sorted); return hip_fail(e, "merge_indexed_dev: concatenation", __FILE__, __LINE__); }
    hipLaunchKernelGGL(k_iota_u32, dim3(grid_of(total)), dim3(256), 0, st, pos.as<u32>(), total);
    bool in_tmp = false;
    rc = radix_sort_pairs_u64_u32(keys.as<u64>(), pos.as<u32>(), keys2.as<u64>(), pos2.as<u32>(), total, 0, sort_bits, &in_tmp);
    if (rc != SYMGPU_OK) { symgpu_op_free(sorted); return rc; }
    const u64 *ks = in_tmp ? keys2.as<u64>() : keys.as<u64>();
    const u32 *ps = in_tmp ? pos2.as<u32>() : pos.as<u32>();
    hipLaunchKernelGGL(k_gather_rows_u32, dim3(grid_of(total * Wq)), dim3(256), 0, st, reinterpret_cast<const u32x4 *>(cat_rows.p), cat_coeff.as<double>(), ps, total, Wq,
                       reinterpret_cast<u32x4 *>(sorted->rows), sorted->coeff);
    e = hipGetLastError();
    if (e != hipSuccess) { symgpu_op_free(sorted); return hip_fail(e, "merge_indexed_dev: gather", __FILE__, __LINE__); }
    if (!do_cleanup) {
        // parts without common rows (the shares of different ranks): the order is all there is to do
        rc = dev_alloc((size_t)total * 8, (void **)&sorted->first);
        if (rc != SYMGPU_OK) { symgpu_op_free(sorted); return rc; }
        e = hipMemcpyAsync(sorted->first, ks, (size_t)total * 8, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { symgpu_op_free(sorted); return hip_fail(e, "merge_indexed_dev", __FILE__, __LINE__); }
        sorted->dup_free = 1;
        *out = sorted;
        return SYMGPU_OK;
    }
    // rows that several parts share merge at their first occurrence = smallest pair index; sums in pair order of the parts' partial sums
    symgpu_op_t res = nullptr;
    rc = cleanup_core(sorted->rows, sorted->coeff, total, W, nullptr, 0, nullptr, 0, thr, use_thr, &res, Wq, nullptr, nullptr, 1, true);
    if (rc == SYMGPU_OK && res->T > 0) {
        // res->first[t] = position in the sorted concatenation -> its pair index
        Scratch g;
        rc = g.alloc((size_t)res->T * 8);
        if (rc == SYMGPU_OK) {
            hipLaunchKernelGGL(k_pick_u64, dim3(grid_of(res->T)), dim3(256), 0, st, ks, res->first, res->T, g.as<u64>());
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(res->first, g.p, (size_t)res->T * 8, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) rc = hip_fail(e, "merge_indexed_dev: indices", __FILE__, __LINE__);
        }
    }
    symgpu_op_free(sorted);
    if (rc != SYMGPU_OK) { if (res) symgpu_op_free(res); return rc; }
    *out = res;
    return SYMGPU_OK;
}

}  // extern "C"
