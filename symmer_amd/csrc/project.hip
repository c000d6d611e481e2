// project.hip — the callers on either side of the hot path that SURVEY.md §8f lists next (f3 / f4), device side:
//   symgpu_project_dev        S3Projection._perform_projection (reference symmer/projection/base.py:44-84): terms that anticommute
//                             with a fixed single-qubit stabiliser vanish, the others pick up the stabilisers' eigenvalues on the
//                             occupied positions, the stabilised qubits are deleted and equal terms merged — on packed rows: no
//                             one-byte-per-bit expansion, no host round trip of the operator.
//   symgpu_noncontextual_dev  PauliwordOp.is_noncontextual / check_adjmat_noncontextual (base.py:1074-1088, utils.py:567-589): the rows
//                             of the bit-packed adjacency matrix restricted to the terms that do not commute with everything must
//                             split into disjoint cliques — unique rows by the device cleanup, disjointness as a popcount identity.
//   symgpu_state_inner_dev    QuantumState bra * ket (base.py:1808-1815): a hash join of the packed basis rows of two cleaned states,
//                             products added in the left state's order (the reference's Python loop order).
#include "common.h"
#include <string.h>
#include "rotate_common.h"
#include <stdlib.h>
#include <vector>

namespace symgpu {

typedef double f64x2 __attribute__((ext_vector_type(2)));

static int grid_of(i64 n, int block = 256, int cap = 65535) {
    i64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---- projection ---------------------------------------------------------------------------------------------------------------
// survive[t] = term t commutes with every stabiliser row; sign[t] = parity of the term's bits under neg_mask (the symplectic positions
// of the stabilisers whose eigenvalue is -1): the reference's product of eigenvalues over the occupied stabilised positions
// (projection/base.py:68-71: an eigenvalue 0 counts as 1) is (-1)^that.
__global__ __launch_bounds__(256) void k_proj_flags(const u64 *__restrict__ rows, i64 T, int Wq, const u64 *__restrict__ stab, int k,
                                                     const u64 *__restrict__ neg_mask, u32 *__restrict__ survive, unsigned char *__restrict__ sign) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= T) return;
    const u64 *r = rows + t * 2 * Wq;
    bool ok = true;
    for (int s = 0; s < k && ok; ++s) {
        const u64 *q = stab + (i64)s * 2 * Wq;
        u64 par = 0;
        for (int w = 0; w < Wq; ++w) par ^= (r[w] & q[Wq + w]) ^ (r[Wq + w] & q[w]);
        ok = (__popcll(par) & 1) == 0;
    }
    u64 neg = 0;
    for (int w = 0; w < 2 * Wq; ++w) neg ^= r[w] & neg_mask[w];
    survive[t] = ok ? 1u : 0u;
    sign[t] = (unsigned char)(__popcll(neg) & 1);
}

// the surviving terms, compacted in order, with the stabilised qubits deleted: output word w of a row collects its 64 bits from the
// source positions src[64 w ..] (0xFFFFFFFF: padding); coefficient times the sign (exact)
__global__ __launch_bounds__(256) void k_proj_emit(const u64 *__restrict__ rows, const double *__restrict__ coeff, i64 T, int Wq, const u32 *__restrict__ survive,
                                                    const u32 *__restrict__ pos, const unsigned char *__restrict__ sign, const u32 *__restrict__ src, int W_out,
                                                    u64 *__restrict__ out_rows, double *__restrict__ out_coeff) {
    const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 t = idx / W_out;
    const int w = (int)(idx - t * W_out);
    if (t >= T || !survive[t]) return;
    const u64 *r = rows + t * 2 * Wq;
    u64 v = 0;
    for (int b = 0; b < 64; ++b) {
        const u32 sp = src[w * 64 + b];
        if (sp != 0xFFFFFFFFu) v |= ((r[sp >> 6] >> (sp & 63u)) & 1ULL) << b;
    }
    const i64 d = pos[t];
    out_rows[d * W_out + w] = v;
    if (w == 0) {
        const f64x2 c = reinterpret_cast<const f64x2 *>(coeff)[t];
        reinterpret_cast<f64x2 *>(out_coeff)[d] = sign[t] ? f64x2{-c.x, -c.y} : c;
    }
}

// ---- noncontextuality -------------------------------------------------------------------------------------------------------------
// one wavefront per adjacency row: number of terms it commutes with
__global__ __launch_bounds__(256) void k_row_popcount(const u64 *__restrict__ bits, i64 T, i64 Mw, u32 *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const i64 r = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= T) return;
    u32 c = 0;
    for (i64 w = lane; w < Mw; w += 64) c += (u32)__popcll(bits[r * Mw + w]);
    for (int off = 32; off > 0; off >>= 1) c += (u32)__shfl_xor((int)c, off);
    if (lane == 0) cnt[r] = c;
}
// word w of the mask of NON-universal terms, and the per-term flag for the compaction
__global__ __launch_bounds__(256) void k_nonuniversal(const u32 *__restrict__ cnt, i64 T, i64 Mw, u64 *__restrict__ mask, u32 *__restrict__ flag) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    const bool nu = t < T && cnt[t] != (u32)T;
    const u64 m = __ballot(nu);
    if (t < T) flag[t] = nu ? 1u : 0u;
    if ((threadIdx.x & 63) == 0 && t / 64 < Mw) mask[t / 64] = m;
}
// the characters: adjacency rows of the non-universal terms restricted to the non-universal columns, compacted, padded to W2 words
__global__ __launch_bounds__(256) void k_characters(const u64 *__restrict__ bits, i64 T, i64 Mw, const u64 *__restrict__ mask, const u32 *__restrict__ flag,
                                                     const u32 *__restrict__ pos, int W2, u64 *__restrict__ out_rows, double *__restrict__ out_coeff) {
    const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 t = idx / W2;
    const int w = (int)(idx - t * W2);
    if (t >= T || !flag[t]) return;
    const i64 d = pos[t];
    out_rows[d * W2 + w] = w < Mw ? (bits[t * Mw + w] & mask[w]) : 0ULL;
    if (w == 0) reinterpret_cast<f64x2 *>(out_coeff)[d] = f64x2{1.0, 0.0};
}

// ---- inner product of two states --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_join_insert(const u64 *__restrict__ h, i64 N, u64 *__restrict__ table, u32 mask) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= N) return;
    const u64 entry = (h[t] & 0xFFFFFFFF00000000ULL) | (u64)(t + 1);
    u32 p = (u32)mix64(h[t]) & mask;
    while (atomicCAS(reinterpret_cast<unsigned long long *>(&table[p]), 0ULL, (unsigned long long)entry) != 0ULL) p = (p + 1) & mask;
}
// prod[i] = c_a[i] * c_b[j] for the row j of b equal to row i of a (rows compared word by word: exactness does not rest on the hash), else 0
__global__ __launch_bounds__(256) void k_join_probe(const u64 *__restrict__ ra, const double *__restrict__ ca, const u64 *__restrict__ ha, i64 Na,
                                                     const u64 *__restrict__ rb, const double *__restrict__ cb, const u64 *__restrict__ hb, int W,
                                                     const u64 *__restrict__ table, u32 mask, double *__restrict__ prod) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= Na) return;
    const u64 h = ha[i];
    f64x2 out = {0.0, 0.0};
    for (u32 p = (u32)mix64(h) & mask;; p = (p + 1) & mask) {
        const u64 e = table[p];
        if (e == 0ULL) break;
        if ((e >> 32) != (h >> 32)) continue;
        const i64 j = (i64)(e & 0xFFFFFFFFULL) - 1;
        if (hb[j] != h) continue;
        bool same = true;
        for (int w = 0; w < W; ++w) same &= ra[i * W + w] == rb[j * W + w];
        if (!same) continue;
        const f64x2 a = reinterpret_cast<const f64x2 *>(ca)[i], b = reinterpret_cast<const f64x2 *>(cb)[j];
        double re, im;
        pair_coefficient(a.x, a.y, b.x, b.y, 0, re, im);
        out = f64x2{re, im};
        break;
    }
    reinterpret_cast<f64x2 *>(prod)[i] = out;
}
// 0 + v[0] + v[1] + ... in that order (the reference's `inner_product += ...` loop): one wavefront, 64 values per step through LDS
__global__ __launch_bounds__(64) void k_seq_sum(const double *__restrict__ v, i64 N, double *__restrict__ out) {
    __shared__ f64x2 s_p[2][64];
    const int lane = threadIdx.x;
    double re = 0.0, im = 0.0;
    int buf = 0;
    for (i64 base = 0; base < N; base += 64, buf ^= 1) {
        if (base + lane < N) s_p[buf][lane] = reinterpret_cast<const f64x2 *>(v)[base + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) {
            const int m = (int)(N - base < 64 ? N - base : 64);
            for (int k = 0; k < m; ++k) { const f64x2 p = s_p[buf][k]; re = __dadd_rn(re, p.x); im = __dadd_rn(im, p.y); }
        }
    }
    if (lane == 0) { out[0] = re; out[1] = im; }
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_project_dev(symgpu_op_t op, const uint64_t *stab_rows, int k, const uint64_t *neg_mask, const int *keep_qubits, int n_keep, int n_qubits,
                       double thr, int use_thr, symgpu_op_t *out, int64_t *n_survived) {
    SG_ENTER(op);
    if (n_survived) *n_survived = 0;
    SG_REQUIRE(op && out && k >= 0 && n_keep >= 1 && n_qubits >= 1 && neg_mask && keep_qubits && (k == 0 || stab_rows), "project_dev: arguments");
    SG_REQUIRE((n_qubits + 63) / 64 == op->Wq, "project_dev: n_qubits does not match the operator's Wq");
    SG_REQUIRE(op->coeff || op->T == 0, "project_dev: operator has no coefficients");
    hipStream_t st = ctx().stream;
    const int Wq = op->Wq, W = 2 * Wq, Wq_out = (n_keep + 63) / 64, W_out = 2 * Wq_out;
    const i64 T = op->T;
    *out = nullptr;
    for (int j = 0; j < n_keep; ++j) SG_REQUIRE(keep_qubits[j] >= 0 && keep_qubits[j] < n_qubits && (j == 0 || keep_qubits[j] > keep_qubits[j - 1]), "project_dev: keep_qubits must ascend inside [0, n)");
    if (T == 0) {
        SG_TRY(symgpu_op_alloc(1, Wq_out, 1, out));
        (*out)->T = 0;
        return SYMGPU_OK;
    }
    // source position of every output bit: X half then Z half of the kept qubits
    std::vector<u32> src((size_t)W_out * 64, 0xFFFFFFFFu);
    for (int j = 0; j < n_keep; ++j) {
        src[j] = (u32)keep_qubits[j];
        src[(size_t)Wq_out * 64 + j] = (u32)(Wq * 64 + keep_qubits[j]);
    }
    Scratch d_stab, d_neg, d_src, survive, pos, sign, total, t_rows, t_coeff;
    SG_TRY(d_stab.alloc((size_t)(k > 0 ? k : 1) * W * 8));
    SG_TRY(d_neg.alloc((size_t)W * 8));
    SG_TRY(d_src.alloc(src.size() * 4));
    SG_TRY(survive.alloc((size_t)T * 4));
    SG_TRY(pos.alloc((size_t)T * 4));
    SG_TRY(sign.alloc((size_t)T));
    SG_TRY(total.alloc(16));
    if (k > 0) HIP_TRY(hipMemcpyAsync(d_stab.p, stab_rows, (size_t)k * W * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_neg.p, neg_mask, (size_t)W * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_src.p, src.data(), src.size() * 4, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_proj_flags, dim3(grid_of(T, 256, 1 << 30)), dim3(256), 0, st, op->rows, T, Wq, d_stab.as<u64>(), k, d_neg.as<u64>(), survive.as<u32>(),
                       sign.as<unsigned char>());
    KERNEL_CHECK();
    SG_TRY(exclusive_scan_u32(survive.as<u32>(), pos.as<u32>(), T, total.as<u32>()));
    u32 n_s = 0;
    SG_TRY(read_back_words(total.as<u32>(), 1, nullptr, 0, &n_s));   // also: src / stab_rows / neg_mask are host temporaries of the caller (their copies are ahead in the stream)
    if (n_survived) *n_survived = n_s;
    if (n_s == 0) {
        SG_TRY(symgpu_op_alloc(1, Wq_out, 1, out));
        (*out)->T = 0;
        return SYMGPU_OK;
    }
    SG_TRY(t_rows.alloc((size_t)n_s * W_out * 8));
    SG_TRY(t_coeff.alloc((size_t)n_s * 16));
    hipLaunchKernelGGL(k_proj_emit, dim3(grid_of(T * W_out, 256, 1 << 30)), dim3(256), 0, st, op->rows, op->coeff, T, Wq, survive.as<u32>(), pos.as<u32>(),
                       sign.as<unsigned char>(), d_src.as<u32>(), W_out, t_rows.as<u64>(), t_coeff.as<double>());
    KERNEL_CHECK();
    // equal projected terms merge (the reference's .cleanup() at projection/base.py:82)
    return cleanup_core(t_rows.as<u64>(), t_coeff.as<double>(), n_s, W_out, nullptr, 0, nullptr, 0, thr, use_thr, out, Wq_out);
}

int symgpu_noncontextual_dev(symgpu_op_t op, int *is_noncontextual) {
    SG_ENTER(op);
    SG_REQUIRE(op && is_noncontextual, "noncontextual_dev: null argument");
    hipStream_t st = ctx().stream;
    const i64 T = op->T;
    *is_noncontextual = 1;
    if (T <= 1) return SYMGPU_OK;
    SG_REQUIRE(T < ((i64)1 << 31), "noncontextual_dev: too many terms");
    const i64 Mw = (T + 63) / 64;
    const int W2 = (int)(2 * ((Mw + 1) / 2));
    Scratch bits, cnt, mask, flag, pos, total, c_rows, c_coeff;
    SG_TRY(bits.alloc((size_t)T * Mw * 8));
    SG_TRY(cnt.alloc((size_t)T * 4));
    SG_TRY(mask.alloc((size_t)Mw * 8));
    SG_TRY(flag.alloc((size_t)T * 4));
    SG_TRY(pos.alloc((size_t)T * 4));
    SG_TRY(total.alloc(16));
    SG_TRY(symgpu_commutes_bits_dev(op, 0, T, op, bits.as<u64>()));
    hipLaunchKernelGGL(k_row_popcount, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, bits.as<u64>(), T, Mw, cnt.as<u32>());
    hipLaunchKernelGGL(k_nonuniversal, dim3((unsigned)(Mw * 64 / 256 + 1)), dim3(256), 0, st, cnt.as<u32>(), T, Mw, mask.as<u64>(), flag.as<u32>());
    KERNEL_CHECK();
    SG_TRY(exclusive_scan_u32(flag.as<u32>(), pos.as<u32>(), T, total.as<u32>()));
    u32 n_nu = 0;
    SG_TRY(read_back_words(total.as<u32>(), 1, nullptr, 0, &n_nu));
    if (n_nu == 0) return SYMGPU_OK;                                 // everything commutes with everything
    SG_TRY(c_rows.alloc((size_t)n_nu * W2 * 8));
    SG_TRY(c_coeff.alloc((size_t)n_nu * 16));
    hipLaunchKernelGGL(k_characters, dim3(grid_of(T * W2, 256, 1 << 30)), dim3(256), 0, st, bits.as<u64>(), T, Mw, mask.as<u64>(), flag.as<u32>(), pos.as<u32>(), W2,
                       c_rows.as<u64>(), c_coeff.as<double>());
    KERNEL_CHECK();
    // np.unique(axis=0) of utils.py:587 = the device cleanup of the bit-packed rows (no threshold); the unique characters partition the
    // non-universal terms into cliques iff no column sits in two of them: every column is in at least one (a term commutes with itself),
    // so iff the set bits of the unique rows add up to the number of columns
    symgpu_op_t uniq = nullptr;
    SG_TRY(cleanup_core(c_rows.as<u64>(), c_coeff.as<double>(), n_nu, W2, nullptr, 0, nullptr, 0, 0.0, 0, &uniq, W2 / 2));
    uint64_t ones = 0;
    const int rc = symgpu_op_popcount(uniq, &ones);
    symgpu_op_free(uniq);
    if (rc != SYMGPU_OK) return rc;
    *is_noncontextual = ones == (uint64_t)n_nu ? 1 : 0;
    return SYMGPU_OK;
}

int symgpu_state_inner_dev(symgpu_op_t a, symgpu_op_t b, double *out) {
    SG_ENTER(a, b);
    SG_REQUIRE(a && b && out && a->Wq == b->Wq, "state_inner_dev: arguments");
    SG_REQUIRE((a->coeff || a->T == 0) && (b->coeff || b->T == 0), "state_inner_dev: states have no coefficients");
    SG_REQUIRE(a->dup_free && b->dup_free, "state_inner_dev: both states must come from a cleanup (to_dictionary cleans them, base.py:2104)");
    hipStream_t st = ctx().stream;
    out[0] = out[1] = 0.0;
    const i64 Na = a->T, Nb = b->T;
    if (Na == 0 || Nb == 0) return SYMGPU_OK;
    SG_REQUIRE(Nb < ((i64)1 << 31), "state_inner_dev: too many terms");
    const int W = 2 * a->Wq;
    Scratch ha, hb, table, prod, res;
    SG_TRY(ha.alloc((size_t)Na * 8));
    SG_TRY(hb.alloc((size_t)Nb * 8));
    size_t cap = 1024;
    while ((i64)cap < 2 * Nb) cap <<= 1;
    SG_TRY(table.alloc(cap * 8));
    SG_TRY(prod.alloc((size_t)Na * 16));
    SG_TRY(res.alloc(16));
    SG_TRY(ensure_hash_tables(ctx().hash_tab ? ctx().hash_seed : 1));
    SG_TRY(hash_rows(a->rows, Na, W, ha.as<u64>()));
    SG_TRY(hash_rows(b->rows, Nb, W, hb.as<u64>()));
    HIP_TRY(hipMemsetAsync(table.p, 0, cap * 8, st));
    hipLaunchKernelGGL(k_join_insert, dim3(grid_of(Nb, 256, 1 << 30)), dim3(256), 0, st, hb.as<u64>(), Nb, table.as<u64>(), (u32)(cap - 1));
    hipLaunchKernelGGL(k_join_probe, dim3(grid_of(Na, 256, 1 << 30)), dim3(256), 0, st, a->rows, a->coeff, ha.as<u64>(), Na, b->rows, b->coeff, hb.as<u64>(), W,
                       table.as<u64>(), (u32)(cap - 1), prod.as<double>());
    hipLaunchKernelGGL(k_seq_sum, dim3(1), dim3(64), 0, st, prod.as<double>(), Na, res.as<double>());
    KERNEL_CHECK();
    u32 w[4] = {0, 0, 0, 0};
    SG_TRY(read_back_words(res.as<u32>(), 4, nullptr, 0, w));
    memcpy(out, w, 16);
    return SYMGPU_OK;
}

}  // extern "C"
