// wide.hip — few pairs of VERY long rows (the reference's README claim "multiply two 100,000,000-qubit Pauli terms",
// README.md:54; PauliwordOp._multiply_by_operator base.py:764-794, commutes_termwise base.py:938-971).
//
// The all-pairs kernels (product.hip k_mul_coeff, commute.hip k_commutes) give every lane a pair and let it walk the words of
// its two rows: right for 1e5 x 1e5 terms of 16-32 words, wrong for 1 x 1 terms of 1.5 million words — one wavefront, one
// dependent load per word, 1.6 us each: 2.5 s for the 1e8-qubit product.  Here the WORD axis is the parallel one: a block takes
// 1024 words of one pair (row-major rows: coalesced 8-byte loads), reduces its popcounts and adds them to the pair's two
// accumulators; a tiny second kernel turns the accumulators into commutation bytes / bits, coefficients or cleanup keys.
// Used while the pair count is small enough for re-reading both rows per pair to beat the register-tiled kernels
// (wide_pairs_worthwhile): 1e8 qubits, 1 x 1 terms: 2.5 s -> 3 ms.
#include "common.h"

namespace symgpu {

constexpr int WIDE_U = 4;      // words per thread

// acc[2 * pair]     += 3 (Y_i + Y_o) + Y_out + 2 |x_left & z_right|   over the block's words (the phase exponent mod 4)
// acc[2 * pair + 1] += |x_i & z_o| + |z_i & x_o|                        (its parity: 1 = the two terms anticommute)
// pair = o * Ni + i;  inner rows i in [0, Ni), outer rows o in [0, No)
__global__ __launch_bounds__(256) void k_wide_pair_counts(const u64 *__restrict__ inner, i64 Ni, const u64 *__restrict__ outer, int Wq,
                                                           int inner_is_left, i64 pair_base, u32 *__restrict__ acc) {
    __shared__ u32 red[2][4];
    const i64 pair = pair_base + blockIdx.y;
    const i64 o = pair / Ni, i = pair - o * Ni;
    const u64 *ri = inner + i * 2 * Wq, *ro = outer + o * 2 * Wq;
    u32 e = 0, par = 0;
#pragma unroll
    for (int u = 0; u < WIDE_U; ++u) {
        const i64 w = ((i64)blockIdx.x * WIDE_U + u) * 256 + threadIdx.x;
        if (w < Wq) {
            const u64 xi = ri[w], zi = ri[Wq + w], xo = ro[w], zo = ro[Wq + w];
            e += 3u * (u32)(__popcll(xi & zi) + __popcll(xo & zo)) + (u32)__popcll((xi ^ xo) & (zi ^ zo)) +
                 2u * (u32)__popcll(inner_is_left ? (xi & zo) : (xo & zi));
            par += (u32)__popcll((xi & zo) ^ (zi & xo));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        e += (u32)__shfl_xor((int)e, off);
        par += (u32)__shfl_xor((int)par, off);
    }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = e; red[1][threadIdx.x >> 6] = par; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&acc[2 * (pair - pair_base)], red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        atomicAdd(&acc[2 * (pair - pair_base) + 1], red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
}

// commutation table: A rows = "inner" index i (N of them), B rows = "outer" index j: out[i * M + j] = 1 if they commute
__global__ void k_wide_commutes_bytes(const u32 *__restrict__ acc, i64 N, i64 M, uint8_t *__restrict__ out) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * M) return;
    const i64 i = t / M, j = t - i * M;
    out[t] = (uint8_t)(1u ^ (acc[2 * (j * N + i) + 1] & 1u));
}
__global__ void k_wide_commutes_bits(const u32 *__restrict__ acc, i64 N, i64 M, u64 *__restrict__ out_bits) {
    const i64 Mw = (M + 63) / 64;
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * Mw) return;
    const i64 i = t / Mw, jw = t - i * Mw;
    u64 word = 0;
    for (int b = 0; b < 64; ++b) {
        const i64 j = jw * 64 + b;
        if (j < M && !(acc[2 * (j * N + i) + 1] & 1u)) word |= 1ULL << b;
    }
    out_bits[t] = word;
}
// coefficients (keys == null) or packed cleanup keys of the pairs [pair_base, pair_base + n_pairs)
__global__ void k_wide_coeff(const u32 *__restrict__ acc, i64 Ni, i64 pair_base, i64 n_pairs, const double *__restrict__ ci,
                             const double *__restrict__ co, double *__restrict__ out, PairKeyArgs ka, int keys) {
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    const i64 pair = pair_base + p;
    const i64 o = pair / Ni, i = pair - o * Ni;
    const u32 e = acc[2 * p] & 3u;
    if (!keys) {
        double re, im;
        pair_coefficient(ci[2 * i], ci[2 * i + 1], co[2 * o], co[2 * o + 1], (int)e, re, im);
        out[2 * pair] = re;
        out[2 * pair + 1] = im;
        return;
    }
    const int F = ka.bi + ka.bo + 2;
    const u64 hmask = ~((1ULL << F) - 1ULL);
    const u64 key = ((ka.hI[i] ^ ka.hO[o]) & hmask) | ((u64)e << (ka.bi + ka.bo)) | ((u64)o << ka.bi) | (u64)i;
    if (ka.squared) {
        if (i >= o) ka.keys[(u64)o * Ni - (u64)o * (o - 1) / 2 - o + i] = key;      // compacted slot of the pair (o, i), i >= o
    } else ka.keys[pair] = key;
}

// few pairs, long rows: re-reading both rows per pair (32 B per word and pair, L2 / Infinity Cache resident for all but the first
// reader) beats one wavefront per 8 x 256 pairs walking Wq words at one dependent load each
bool wide_pairs_worthwhile(i64 Ni, i64 No, int Wq) {
    if (const char *e = getenv("SYMGPU_WIDE")) {
        if (e[0] == '1') return Ni * No <= (1 << 22);
        if (e[0] == '0') return false;
    }
    return Wq >= 256 && Ni * No <= 65536;
}

static int wide_counts(const u64 *inner, i64 Ni, const u64 *outer, i64 No, int Wq, int inner_is_left, i64 pair_base, i64 n_pairs, u32 *acc) {
    hipStream_t st = ctx().stream;
    HIP_TRY(hipMemsetAsync(acc, 0, (size_t)n_pairs * 2 * sizeof(u32), st));
    const unsigned gx = (unsigned)((Wq + 256 * WIDE_U - 1) / (256 * WIDE_U));
    for (i64 p0 = 0; p0 < n_pairs; p0 += 65535) {
        const i64 np = n_pairs - p0 < 65535 ? n_pairs - p0 : 65535;
        hipLaunchKernelGGL(k_wide_pair_counts, dim3(gx, (unsigned)np), dim3(256), 0, st, inner, Ni, outer, Wq, inner_is_left, pair_base + p0, acc + 2 * p0);
        KERNEL_CHECK();
    }
    (void)No;
    return SYMGPU_OK;
}

// commutes_dev's contract (commute.hip): exactly one of out / out_bits is non-null
int wide_commutes_dev(const u64 *A, i64 N, const u64 *B, i64 M, int Wq, uint8_t *out, u64 *out_bits) {
    Scratch acc;
    SG_TRY(acc.alloc((size_t)N * M * 2 * sizeof(u32)));
    SG_TRY(wide_counts(A, N, B, M, Wq, 1, 0, N * M, acc.as<u32>()));
    hipStream_t st = ctx().stream;
    if (out) hipLaunchKernelGGL(k_wide_commutes_bytes, dim3((unsigned)((N * M + 255) / 256)), dim3(256), 0, st, acc.as<u32>(), N, M, out);
    else hipLaunchKernelGGL(k_wide_commutes_bits, dim3((unsigned)((N * ((M + 63) / 64) + 255) / 256)), dim3(256), 0, st, acc.as<u32>(), N, M, out_bits);
    KERNEL_CHECK();
    return SYMGPU_OK;
}

// mul_coeff_dev's contract (product.hip): out_coeff[(o - o_begin) * Ni + i]; keys != null: packed keys of ALL pairs (o_begin = 0)
int wide_mul_coeff_dev(const u64 *inner, const double *ci, i64 Ni, const u64 *outer, const double *co, i64 o_begin, i64 o_end, int Wq,
                       int inner_is_left, double *out_coeff, const PairKeyArgs *keys) {
    const i64 No = o_end - o_begin, n_pairs = Ni * No;
    Scratch acc;
    SG_TRY(acc.alloc((size_t)n_pairs * 2 * sizeof(u32)));
    SG_TRY(wide_counts(inner, Ni, outer + o_begin * 2 * Wq, No, Wq, inner_is_left, 0, n_pairs, acc.as<u32>()));
    hipLaunchKernelGGL(k_wide_coeff, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, ctx().stream, acc.as<u32>(), Ni, (i64)0, n_pairs, ci,
                       co ? co + 2 * o_begin : nullptr, out_coeff, keys ? *keys : PairKeyArgs(), keys ? 1 : 0);
    KERNEL_CHECK();
    return SYMGPU_OK;
}

}  // namespace symgpu
