// cleanup.hip — duplicate-term cleanup (reference: symplectic_cleanup, symmer/operators/utils.py:230-279;
// PauliwordOp.cleanup base.py:617-638) and the fused product+cleanup (base.py:764-794).
//
// The reference keys a hash map with the full row.  Here rows are grouped by sorting a 64-bit GF(2)-LINEAR
// row hash  h(r) = XOR over set bits of a random 64-bit vector per bit position, evaluated as 8 byte-table
// lookups per word (LDS), a per-word rotation and a per-64-word-block xorshift step.  Linearity gives
// h(a ^ b) = h(a) ^ h(b): the key of product row (i, o) is hI[i] ^ hO[o], so the fused path never
// materialises the N*M product rows — only the surviving unique rows are written.
// Exactness does not rest on the hash: after the stable sort every adjacent equal-key pair is compared
// word by word; any mismatch reseeds the tables and retries (SYMGPU_E_COLLISION if it survives 4 seeds).
//
// Pipeline: hash -> stable LSD radix sort (key, input index) -> head flags + verify -> scan (segment ids)
//           -> per-segment SEQUENTIAL coefficient sum in input order (== np.add.at, utils.py:273-274)
//           -> threshold |c| > thr (strict, utils.py:275-278) -> first-occurrence order via mark+scan over
//           input positions (qiskit `unordered_unique` order, utils.py:271) -> gather surviving rows.
#include "common.h"
#include <stdlib.h>
#include <stdio.h>
#include <vector>

namespace symgpu {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

static u64 host_splitmix64(u64 &s) {
    u64 z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

i64 g_hash_reseeds = 0;                // statistics (symgpu_debug_counter 0)
// k_hash_rows_long: columns of M^(2^j), M = the xorshift step of the per-lane Horner scheme (a linear map on GF(2)^64), j < 32
static u64 host_xorshift_step(u64 h) { h ^= h << 13; h ^= h >> 7; h ^= h << 17; return h; }
static int ensure_xs_pow() {
    if (ctx().xs_pow) return SYMGPU_OK;
    std::vector<u64> P(32 * 64);
    for (int i = 0; i < 64; ++i) P[i] = host_xorshift_step(1ULL << i);
    for (int j = 1; j < 32; ++j)
        for (int i = 0; i < 64; ++i) {                               // column i of M^(2^j) = M^(2^(j-1)) applied to column i of M^(2^(j-1))
            const u64 x = P[(j - 1) * 64 + i];
            u64 y = 0;
            for (int b = 0; b < 64; ++b)
                if ((x >> b) & 1) y ^= P[(j - 1) * 64 + b];
            P[j * 64 + i] = y;
        }
    HIP_TRY(hipMalloc((void **)&ctx().xs_pow, P.size() * sizeof(u64)));
    HIP_TRY(hipMemcpy(ctx().xs_pow, P.data(), P.size() * sizeof(u64), hipMemcpyHostToDevice));
    return SYMGPU_OK;
}

int ensure_hash_tables(u64 seed) {
    Context &c = ctx();
    if (c.hash_tab && c.hash_seed == seed) return SYMGPU_OK;
    if (!c.hash_tab) HIP_TRY(hipMalloc((void **)&c.hash_tab, 8 * 256 * 2 * sizeof(u64)));
    std::vector<u64> tab(8 * 256 * 2);
    u64 s = seed * 0x2545f4914f6cdd1dULL + 0x1234567ULL;
    u64 basis[2][64];
    for (int h = 0; h < 2; ++h)
        for (int b = 0; b < 64; ++b) basis[h][b] = host_splitmix64(s);
    // test hook: SYMGPU_HASH_WEAK_ODD=1 leaves an odd seed only the 4 top hash bits, so that different rows collide in bulk and the
    // exactness guard (row-against-row verification -> reseed -> retry; long mixed prefix runs -> full 64-bit sort) actually runs
    if (const char *e = getenv("SYMGPU_HASH_WEAK_ODD"))
        if (e[0] == '1' && (seed & 1))
            for (int h = 0; h < 2; ++h)
                for (int b = 0; b < 64; ++b) basis[h][b] &= 0xF000000000000000ULL;
    for (int k = 0; k < 8; ++k)
        for (int v = 0; v < 256; ++v)
            for (int h = 0; h < 2; ++h) {
                u64 x = 0;
                for (int b = 0; b < 8; ++b)
                    if ((v >> b) & 1) x ^= basis[h][8 * k + b];
                tab[((size_t)k * 256 + v) * 2 + h] = x;
            }
    HIP_TRY(hipMemcpyAsync(c.hash_tab, tab.data(), tab.size() * sizeof(u64), hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));   // tab is a host temporary
    c.hash_seed = seed;
    c.host_hash_tab = tab;
    return SYMGPU_OK;
}

// host evaluation of the same linear hash h1 as k_hash_rows (per-lane Horner over 64-word blocks, XOR over lanes)
u64 host_row_hash(const u64 *row, int W) {
    const std::vector<u64> &tab = ctx().host_hash_tab;
    const int n_blk = (W + 63) / 64;
    u64 h = 0;
    for (int g = 0; g < 64; ++g) {
        u64 hg = 0;
        for (int b = 0; b < n_blk; ++b) {
            const int w = b * 64 + g;
            u64 a1 = 0;
            if (w < W) {
                const u64 x = row[w];
                for (int k = 0; k < 8; ++k) a1 ^= tab[((size_t)k * 256 + ((x >> (8 * k)) & 255)) * 2];
            }
            hg ^= hg << 13; hg ^= hg >> 7; hg ^= hg << 17;
            const int r = g & 63;
            hg ^= r ? ((a1 << r) | (a1 >> (64 - r))) : a1;
        }
        h ^= hg;
    }
    return h;
}

__device__ __forceinline__ u64 rotl64(u64 x, int r) { r &= 63; return r ? ((x << r) | (x >> (64 - r))) : x; }
__device__ __forceinline__ u64 xorshift_step(u64 h) { h ^= h << 13; h ^= h >> 7; h ^= h << 17; return h; }

// Row hash on row-major rows.  G (power of two, <= 64) lanes cooperate on one row; lane g handles words
// g, g+64, g+128, ... (only G == 64 has more than one).  Only the first of the two table columns is used (16 KiB of LDS).
__global__ __launch_bounds__(256) void k_hash_rows(const u64 *__restrict__ rows, i64 T, int W, int G, const u64 *__restrict__ tab_g,
                                                    u64 *__restrict__ out1) {
    __shared__ u64 tab[8 * 256];
    for (int k = threadIdx.x; k < 8 * 256; k += 256) tab[k] = tab_g[2 * k];
    __syncthreads();
    const int rows_per_block = 256 / G;
    const int g = threadIdx.x % G, rsub = threadIdx.x / G;
    const int n_blk = (W + 63) / 64;
    constexpr int HU = 4;                                           // row groups in flight per step (the loop is latency bound)
    for (i64 t0 = (i64)blockIdx.x * rows_per_block * HU; t0 < T; t0 += (i64)gridDim.x * rows_per_block * HU) {
        u64 h1[HU];
#pragma unroll
        for (int u = 0; u < HU; ++u) h1[u] = 0;
        // BU 64-word blocks of every row group are loaded before the (sequential) Horner steps consume them: a 1e8-qubit row is
        // 48,828 blocks long, and one dependent load per step made its hash 80 ms
        constexpr int BU = 4;
        for (int b0 = 0; b0 < n_blk; b0 += BU) {
            u64 x[BU][HU];
#pragma unroll
            for (int bu = 0; bu < BU; ++bu) {
                const int w = (b0 + bu) * 64 + g;
#pragma unroll
                for (int u = 0; u < HU; ++u) {
                    const i64 t = t0 + (i64)u * rows_per_block + rsub;
                    x[bu][u] = (t < T && w < W && g < 64) ? rows[t * W + w] : 0ULL;
                }
            }
#pragma unroll
            for (int bu = 0; bu < BU; ++bu) {
                if (b0 + bu >= n_blk) break;                            // uniform
                const int w = (b0 + bu) * 64 + g;
#pragma unroll
                for (int u = 0; u < HU; ++u) {
                    u64 a1 = 0;
                    if (w < W && g < 64) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) a1 ^= tab[k * 256 + (int)((x[bu][u] >> (8 * k)) & 255)];
                    }
                    h1[u] = xorshift_step(h1[u]) ^ rotl64(a1, g);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            for (int off = G >> 1; off > 0; off >>= 1) h1[u] ^= __shfl_xor(h1[u], off);
            const i64 t = t0 + (i64)u * rows_per_block + rsub;
            if (g == 0 && t < T) out1[t] = h1[u];
        }
    }
}

// ---- a hash for rows that nothing XORs together afterwards (plain cleanup, joins of two operators) ---------------------------------------
// The tabulated hash above is GF(2)-linear because the fused product needs h(a ^ b) = h(a) ^ h(b); it costs eight LDS look-ups per 64-bit
// word and runs at 1.0-1.8 TB/s (a plain cleanup of 1e7 rows of 1,000 qubits spent a third of its time in it).  Where linearity is not
// needed every word goes through an injective 64-bit mix salted with its position and the seed (two rounds of the murmur3 32-bit finaliser
// with the halves crossed: four v_mul_lo_u32), and the words of a row are XORed: memory bound.  Exactness never rests on it (equal
// keys are verified row against row, a mismatch reseeds).
__device__ __forceinline__ u32 fmix32(u32 h) { h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16; return h; }
__device__ __forceinline__ u64 mix_word(u64 x, u32 w, u64 seed) {
    u32 a = (u32)x ^ (u32)seed ^ (w * 0x9E3779B1u);
    u32 b = (u32)(x >> 32) ^ (u32)(seed >> 32) ^ (w * 0x85EBCA77u + 0x165667B1u);
    a = fmix32(a + __builtin_amdgcn_alignbit(b, b, 17));             // (rotl 15)
    b = fmix32(b ^ a);
    return ((u64)b << 32) | a;
}
__global__ __launch_bounds__(256) void k_hash_rows_mix(const u64 *__restrict__ rows, i64 T, int W, int G, u64 seed, u64 keep_mask, u64 *__restrict__ out1,
                                                        u32 *__restrict__ iota /* null, or [T]: iota[t] = t (the index array the sort carries) */) {
    // G lanes per row, a lane takes 16-byte chunks g, g + G, ... (W = 2 Wq is even: a row is a whole number of chunks)
    const int rows_per_block = 256 / G;
    const int g = threadIdx.x % G, rsub = threadIdx.x / G;
    const int C = W / 2;                                             // chunks per row
    const u32x4 *rows16 = reinterpret_cast<const u32x4 *>(rows);
    constexpr int HU = 4;                                           // row groups in flight per step
    for (i64 t0 = (i64)blockIdx.x * rows_per_block * HU; t0 < T; t0 += (i64)gridDim.x * rows_per_block * HU) {
        u64 h[HU];
#pragma unroll
        for (int u = 0; u < HU; ++u) h[u] = 0;
        for (int c0 = 0; c0 < C; c0 += 2 * G) {                      // two chunks per lane and step (rows of up to 256 words: one step)
            u32x4 x[2][HU];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int u = 0; u < HU; ++u) {
                    const i64 t = t0 + (i64)u * rows_per_block + rsub;
                    const int c = c0 + k * G + g;
                    const u32x4 z = {0u, 0u, 0u, 0u};
                    x[k][u] = (t < T && c < C) ? rows16[t * C + c] : z;
                }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int u = 0; u < HU; ++u) {
                    const int c = c0 + k * G + g;
                    if (c < C) {
                        h[u] ^= mix_word(((u64)x[k][u].y << 32) | x[k][u].x, (u32)(2 * c), seed);
                        h[u] ^= mix_word(((u64)x[k][u].w << 32) | x[k][u].z, (u32)(2 * c + 1), seed);
                    }
                }
        }
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            for (int off = G >> 1; off > 0; off >>= 1) h[u] ^= __shfl_xor(h[u], off);
            const i64 t = t0 + (i64)u * rows_per_block + rsub;
            if (g == 0 && t < T) { out1[t] = h[u] & keep_mask; if (iota) iota[t] = (u32)t; }
        }
    }
}

// The same hash for VERY long rows (>= 8192 words: > 262,144 qubits; the reference's "two 100,000,000-qubit Pauli terms",
// README.md:54).  The Horner scheme of k_hash_rows is one dependent step per 64 words — 48,828 steps, 37 ms, for a 1e8-qubit row on
// ONE wavefront.  It is linear:  h_g = sum_b M^(n_blk-1-b) v_(b,g),  so a wavefront can run it over a SEGMENT of LSEG blocks and
// shift its partial result to the end of the row with M^(n_blk - segment end) (square-and-multiply on the precomputed columns
// of M^(2^j): <= 16 bit-matrix products), and the segments of a row combine with XOR (atomicXor; out1 zeroed by the caller).
constexpr int LSEG = 64;
__global__ __launch_bounds__(256) void k_hash_rows_long(const u64 *__restrict__ rows, i64 t_base, int W, const u64 *__restrict__ tab_g,
                                                         const u64 *__restrict__ xs_pow, u64 *__restrict__ out1) {
    __shared__ u64 tab[8 * 256];
    for (int k = threadIdx.x; k < 8 * 256; k += 256) tab[k] = tab_g[2 * k];
    __syncthreads();
    const int g = threadIdx.x & 63;
    const int n_blk = (W + 63) / 64;
    const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b_lo = seg * LSEG;
    if (b_lo >= n_blk) return;
    const int b_hi = b_lo + LSEG < n_blk ? b_lo + LSEG : n_blk;
    const i64 t = t_base + blockIdx.y;
    const u64 *row = rows + t * W;
    u64 h = 0;
    constexpr int BU = 8;
    for (int b0 = b_lo; b0 < b_hi; b0 += BU) {
        u64 x[BU];
#pragma unroll
        for (int bu = 0; bu < BU; ++bu) {
            const int w = (b0 + bu) * 64 + g;
            x[bu] = (b0 + bu < b_hi && w < W) ? row[w] : 0ULL;
        }
#pragma unroll
        for (int bu = 0; bu < BU; ++bu) {
            if (b0 + bu >= b_hi) break;                              // wave-uniform
            u64 a1 = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) a1 ^= tab[k * 256 + (int)((x[bu] >> (8 * k)) & 255)];   // padding words are zero: tab[..][0] = 0
            h = xorshift_step(h) ^ rotl64(a1, g);
        }
    }
    // h <- M^(n_blk - b_hi) h
    for (unsigned k = (unsigned)(n_blk - b_hi), j = 0; k; k >>= 1, ++j) {
        if (!(k & 1u)) continue;                                     // wave-uniform
        const u64 *P = xs_pow + j * 64;
        u64 y = 0;
#pragma unroll 8
        for (int i = 0; i < 64; ++i) y ^= P[i] & (0ULL - ((h >> i) & 1ULL));
        h = y;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h ^= __shfl_xor(h, off);
    if (g == 0) atomicXor(reinterpret_cast<unsigned long long *>(out1 + t), (unsigned long long)h);
}

__global__ void k_iota_keys_plain(u32 *__restrict__ idx, i64 T) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) idx[t] = (u32)t;
}

__global__ void k_pair_keys(const u64 *__restrict__ hI, i64 Ni, const u64 *__restrict__ hO, i64 T, u64 *__restrict__ keys, u32 *__restrict__ idx) {
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (i64)gridDim.x * blockDim.x) {
        const i64 o = t / Ni, i = t - o * Ni;
        keys[t] = hI[i] ^ hO[o];
        idx[t] = (u32)t;
    }
}

// PACKED pair keys (written by product.hip's k_mul_coeff<.., KEYS>): one u64 per pair
//     [hash: 64-F bits][e: 2 bits][o: bo bits][i: bi bits],   F = bi + bo + 2,  bi/bo = bits of Ni-1 / No-1  (bi + bo <= 32)
// The radix sort then moves 8 instead of 12 bytes per element and pass, nothing on the path divides by Ni, the 16-byte pair
// coefficient is never materialised (c_i * c_o * i^e is rebuilt from e and the two cache-resident operand tables when the
// sorted order is known), and the full 64-bit key of a sorted element is recomputed from its (i, o) fields with two lookups
// in the per-operand hash tables whenever it is needed.  The LSD sort only touches hash bits and is stable, so equal keys
// stay in ascending pair-index order exactly as with separate index values.
struct PackedLayout {
    int bi, bo;
    __host__ __device__ int F() const { return bi + bo + 2; }
    __device__ __forceinline__ u32 i(u64 k) const { return (u32)(k & ((1ULL << bi) - 1ULL)); }
    __device__ __forceinline__ u32 o(u64 k) const { return (u32)((k >> bi) & ((1ULL << bo) - 1ULL)); }
    __device__ __forceinline__ int e(u64 k) const { return (int)((k >> (bi + bo)) & 3ULL); }
    __device__ __forceinline__ u32 fields(u64 k) const { return (u32)(k & ((1ULL << (bi + bo)) - 1ULL)); }   // (o << bi) | i
    __device__ __forceinline__ u64 full_key(const u64 *__restrict__ hI, const u64 *__restrict__ hO, u64 k) const { return hI[i(k)] ^ hO[o(k)]; }
};

// head flags + exact verification of equal-key neighbours + coefficient gather into sorted order.
// PAIR: row(t) = inner[t % Ni] ^ outer[t / Ni].  One wavefront owns 64 consecutive sorted positions; the positions whose
// key equals their predecessor's are verified COOPERATIVELY with 16-byte loads: G = pow2 >= W/2 lanes (<= 64) cover the
// 16-byte chunks of the two rows, the 64/G lane groups each walk the candidates of their own lane range, so 64/G
// comparisons (4 row reads each in PAIR mode) are in flight per step.  A mismatch (two different rows with one 64-bit
// hash) only raises the collision flag: the caller reseeds the hash and redoes the pass, so exactness never rests on the
// hash.  cg[s] = coeff[idx[s]] turns the segment sums into sequential reads.
__device__ __forceinline__ bool differs(u32x4 a, u32x4 b) {
    const u32x4 d = a ^ b;
    return (d.x | d.y | d.z | d.w) != 0u;
}
// PACKED (implies PAIR): keys are packed pair keys; idx and coeff are unused: the input index, the full key and the pair
// coefficient c_i * c_o * i^e come from the key's fields and the operand tables hI / hO / ci / co.
//
// The SAME kernel forms the segment sums, so that neither the head flags nor the coefficients in sorted order are ever
// written to memory: every wavefront owns a contiguous range of 64-position chunks; head lanes add the coefficients of the
// non-head lanes that follow them one shuffle at a time — SEQUENTIALLY in ascending input order (the sort is stable), exactly
// np.add.at's order (utils.py:273-274) — and a segment that runs past the end of a chunk is carried (wave-uniform
// accumulator) into the next chunks, past the end of the wave's own range if necessary (those chunks are only decoded, their
// owner verifies them); the leading non-head positions of a range therefore belong to the previous wave and are skipped.
// A term that survives the strict |c| > thr test sets the bit of its first input index in `markbits` (T bits: 12.5 MB for
// 1e8 terms, cache resident) and files its sum under that index in `sum_of` for the output stage.
// squared mode: slot of the pair (o, i), i >= o, in pair-index order, and back
__device__ __forceinline__ u32 tri_slot(u32 o, u32 i, u32 N) { return (u32)((u64)o * N - (u64)o * (o - 1) / 2 - o + i); }   // o*(o-1)/2 = 0 for o = 0 (u64 wraps twice)
__device__ __forceinline__ void tri_pair(u32 p, u32 N, u32 &o, u32 &i) {
    // rows o start at off(o) = o*N - o(o-1)/2: largest o with off(o) <= p; float estimate, then exact correction
    const double b = 2.0 * N + 1.0;
    i64 oo = (i64)((b - sqrt(b * b - 8.0 * (double)p)) * 0.5);
    if (oo < 0) oo = 0;
    if (oo > (i64)N - 1) oo = (i64)N - 1;
    auto off = [&](i64 x) { return x * (i64)N - x * (x - 1) / 2; };
    while (oo > 0 && off(oo) > (i64)p) --oo;
    while (oo + 1 < (i64)N && off(oo + 1) <= (i64)p) ++oo;
    o = (u32)oo;
    i = (u32)((i64)p - off(oo) + oo);
}


// ---- terms that merge with nothing ("singles") never pass through the sorted order ------------------------------------------------
// An operator product without repeated rows — the common case, and the benchmark's — merges nothing: every sorted key is a segment of
// its own.  Filing 2.5e7 sums under their first index from the SORTED order is a random 16-byte scatter plus a bitmap atomic per
// term (WRITE_SIZE 1.56 GB for 0.4 GB of sums: 1.0 of k_heads_sums' 1.36 ms at cfg3), and the output stage then reads them back.
// Instead the fate of a term AS IF IT WERE ALONE is decided here, in INDEX order and before the sort — strict |c| > thr on
// 0.0 + c, exactly the sum k_heads_sums forms for a one-element segment — with coalesced bitmap stores; the packed pair key's
// phase exponent goes to two more bitmaps (the sort scrambles the keys).  k_heads_sums then only touches the members of segments
// with MORE than one element: the followers clear their bits, the head files the sum and sets its `patch` bit; k_emit_meta takes a
// patched term's coefficient from the filed sum and rebuilds every other one from the operand tables (cache resident).  Nothing
// about the result changes: same kept set, same order, same sums.
// PACKED: keys[s] is the packed key of index s (pair index, or slot of a squared operator); otherwise coeff[s].
// One workgroup per tile of SORT_TILE indices (= the radix sort's tile).  hist != null (packed keys): the same pass forms the digit
// histograms of the sort's first pass, which reads these very keys in this very order (one HBM pass over the keys less).
// smallest max(|re|, |im|) over the terms of an operand (0 if a component is not a number), as the bit pattern of a non-negative double
// (ordered like the unsigned integer): *slot starts as all ones
__global__ __launch_bounds__(1024) void k_coeff_floor(const double *__restrict__ c0, i64 n0, const double *__restrict__ c1, i64 n1,
                                                      unsigned long long *__restrict__ slot0, int one_block) {
    const double *__restrict__ c = blockIdx.y ? c1 : c0;           // one_block: grid (1, 2) = operand 0 / operand 1 -> slot0[0] / slot0[1]
    const i64 n = blockIdx.y ? n1 : n0;
    unsigned long long *slot = slot0 + blockIdx.y;
    __shared__ unsigned long long s_min;
    if (one_block) { if (threadIdx.x == 0) s_min = ~0ULL; __syncthreads(); }
    double m = __builtin_inf();
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 t0 = (i64)blockIdx.x * blockDim.x + threadIdx.x; t0 < n; t0 += 4 * stride) {      // four loads in flight
        double2 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const i64 t = t0 + k * stride; v[k] = reinterpret_cast<const double2 *>(c)[t < n ? t : n - 1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double a = fabs(v[k].x), b = fabs(v[k].y);
            const double mx = (a == a && b == b) ? (a > b ? a : b) : 0.0;
            m = mx < m ? mx : m;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const double o = __shfl_xor(m, d); m = o < m ? o : m; }
    if ((threadIdx.x & 63) == 0) atomicMin(one_block ? &s_min : slot, (unsigned long long)__double_as_longlong(m));
    if (one_block) { __syncthreads(); if (threadIdx.x == 0) *slot = s_min; }
}
// floor_i / floor_o (packed keys; null: none): k_coeff_floor of the two operands.  |c_i c_o| >= floor_i * floor_o, the larger component of
// the computed product is at least 0.7 of that, so when half of it exceeds thr every pair of non-zero weight is kept WHATEVER its
// coefficient: the two table gathers, the complex product and the comparison (50 of the kernel's 70 instructions per key) are skipped
// — the decision is the same, it is just not computed (cfg3: 0.19 -> see DESIGN 3.3).
template <bool PACKED>
__global__ __launch_bounds__(256) void k_mark_singles(const u64 *__restrict__ keys, const double *__restrict__ coeff, i64 space, PackedLayout L,
                                                       const double *__restrict__ ci, const double *__restrict__ co, int squared, double thr, int use_thr,
                                                       u64 *__restrict__ markbits64, u64 *__restrict__ e_lo64, u64 *__restrict__ e_hi64,
                                                       u32 *__restrict__ hist, int hist_shift, i64 n_tiles, const double *__restrict__ floor_i,
                                                       const double *__restrict__ floor_o) {
    __shared__ u32 s_h[256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool all_kept = PACKED && (!use_thr || (floor_i && floor_o && 0.5 * floor_i[0] * floor_o[0] > thr));   // block-uniform
    if (hist) { s_h[threadIdx.x] = 0; __syncthreads(); }
    const i64 tile_base = (i64)blockIdx.x * SORT_TILE;
#pragma unroll 1
    for (int step = 0; step < SORT_TILE / 1024; ++step) {             // four 64-index chunks per wavefront and step, their loads in flight together
        const i64 g0 = tile_base + step * 1024 + wave * 256;
        if (g0 >= space) break;                                       // wave-uniform
        u64 k[4];
        double2 cf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i64 sidx = g0 + 64 * j + lane;
            k[j] = 0ULL; cf[j].x = 0.0; cf[j].y = 0.0;
            if (sidx < space) {
                if (PACKED) k[j] = keys[sidx];
                else cf[j] = reinterpret_cast<const double2 *>(coeff)[sidx];
            }
        }
        // the operand coefficients of all four chunks are fetched before the first one is used (gathers from the cache-resident tables:
        // one after the other they were four dependent round trips per step, and the kernel was bound by them)
        double2 ca[4], cb[4];
        if (PACKED && !all_kept) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 i = L.i(k[j]), o = L.o(k[j]);                // (lanes past the end hold key 0: term 0 of both tables)
                ca[j] = reinterpret_cast<const double2 *>(ci)[i];
                cb[j] = reinterpret_cast<const double2 *>(co)[o];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i64 sidx = g0 + 64 * j + lane;
            if (g0 + 64 * j >= space) break;                          // wave-uniform
            const bool valid = sidx < space;
            double cx = cf[j].x, cy = cf[j].y;
            int e = 0;
            bool keep;
            if (PACKED && all_kept) {
                if (valid && hist) atomicAdd(&s_h[(u32)(k[j] >> hist_shift) & 255u], 1u);
                e = L.e(k[j]);
                // the weight-0 pairs of a squared operator (anticommuting, off the diagonal) are exact zeros: kept only without a threshold
                keep = valid && !(use_thr && squared && (e & 1) && L.i(k[j]) != L.o(k[j]) && !(0.0 > thr));
            } else {
            if (PACKED && valid) {
                if (hist) atomicAdd(&s_h[(u32)(k[j] >> hist_shift) & 255u], 1u);
                const u32 i = L.i(k[j]), o = L.o(k[j]);
                e = L.e(k[j]);
                pair_coefficient(ca[j].x, ca[j].y, cb[j].x, cb[j].y, e, cx, cy);
                if (squared && i != o) {
                    if (e & 1) { cx = 0.0; cy = 0.0; }
                    else { cx = __dadd_rn(cx, cx); cy = __dadd_rn(cy, cy); }
                }
            }
            // strict |c| > thr: a component that alone exceeds thr decides it (hypot is faithfully rounded and >= either component)
            // (and an exact zero — every anticommuting pair of a squared operator — needs no hypot either)
            const bool zero = cx == 0.0 && cy == 0.0;
            keep = valid && (!use_thr || (zero ? 0.0 > thr : (fabs(cx) > thr || fabs(cy) > thr || hypot(__dadd_rn(0.0, cx), __dadd_rn(0.0, cy)) > thr)));
            }
            const u64 mk = __ballot(keep);
            const i64 chunk = (g0 + 64 * j) / 64;
            if (PACKED) {
                const u64 lo = __ballot(e & 1), hi = __ballot(e & 2);
                if (lane == 0) { markbits64[chunk] = mk; e_lo64[chunk] = lo; e_hi64[chunk] = hi; }
            } else if (lane == 0) markbits64[chunk] = mk;
        }
    }
    if (hist) {
        __syncthreads();
        hist[(i64)blockIdx.x * 256 + threadIdx.x] = s_h[threadIdx.x];         // tile-major, as k_rs_hist files it
    }
}

// Round 6: the same marking from ONE BYTE per pair (PairKeyArgs::ebytes: e | (i == o) << 2, written by the key kernel in index order in
// place of the keys) — where the pairs that share a key are found from the operand hash tables (pair_dups.hip) the marking was the only
// reader of the 400 MB of keys (0.14 ms at cfg3: bound by the latency of its 8-byte loads).  A lane takes 16 consecutive indices (one 16-byte
// load), forms its 16 bits of the three bitmaps with byte-parallel arithmetic and stores them as 2 bytes each (a wavefront: 128 contiguous
// bytes per bitmap): 50 MB in, 19 MB out.  When the operands' coefficient floors do not settle the decision (all_kept false) the sixteen
// pairs are decided one by one from the operand tables like k_mark_singles does.
// index -> (i, o) of a pair: general o * Ni + i; squared operator: the compacted slot of PairKeyArgs
__device__ __forceinline__ void pair_of_index(i64 pos, i64 Ni, int squared, i64 &i, i64 &o) {
    if (!squared) { o = pos / Ni; i = pos - o * Ni; return; }
    // row o starts at s(o) = o (2 Ni - o + 1) / 2: the largest o with s(o) <= pos
    const double b = 2.0 * (double)Ni + 1.0;
    i64 oo = (i64)((b - sqrt(b * b - 8.0 * (double)pos)) * 0.5);
    if (oo < 0) oo = 0;
    if (oo >= Ni) oo = Ni - 1;
    while (oo > 0 && oo * (2 * Ni - oo + 1) / 2 > pos) --oo;
    while (oo + 1 < Ni && (oo + 1) * (2 * Ni - oo) / 2 <= pos) ++oo;
    o = oo;
    i = pos - oo * (2 * Ni - oo + 1) / 2 + oo;
}
__device__ __forceinline__ u32 pack_bit0_of_bytes(u32 w) { return ((w & 0x01010101u) * 0x01020408u) >> 24 & 0xFu; }   // bit 0 of the four bytes -> four bits
__global__ __launch_bounds__(256) void k_mark_bytes(const u32x4 *__restrict__ eb, i64 space, i64 n_groups, i64 Ni, const double *__restrict__ ci,
                                                     const double *__restrict__ co, int squared, double thr, int use_thr, unsigned short *__restrict__ mark16,
                                                     unsigned short *__restrict__ lo16, unsigned short *__restrict__ hi16, const double *__restrict__ floor_i,
                                                     const double *__restrict__ floor_o) {
    const bool all_kept = !use_thr || (floor_i && floor_o && 0.5 * floor_i[0] * floor_o[0] > thr);     // uniform
    const bool drop_rule = use_thr && squared && !(0.0 > thr);           // the anticommuting off-diagonal pairs of a squared operator are exact zeros
    for (i64 g = (i64)blockIdx.x * blockDim.x + threadIdx.x; g < n_groups; g += (i64)gridDim.x * blockDim.x) {
        const i64 p0 = g * 16;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (p0 < space) v = __builtin_nontemporal_load(eb + g);
        u32 valid = 0xFFFFu;
        if (p0 + 16 > space) valid = p0 < space ? (1u << (int)(space - p0)) - 1u : 0u;
        u32 lo = 0, hi = 0, drop = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const u32 w = v[q];
            lo |= pack_bit0_of_bytes(w) << (4 * q);
            hi |= pack_bit0_of_bytes(w >> 1) << (4 * q);
            drop |= pack_bit0_of_bytes(w & ~(w >> 2)) << (4 * q);        // e odd and not the diagonal
        }
        lo &= valid; hi &= valid;
        u32 keep;
        if (all_kept) keep = valid & (drop_rule ? ~drop : 0xFFFFu);
        else {
            keep = 0;
            for (int k = 0; k < 16; ++k) {
                if (!((valid >> k) & 1u)) continue;
                i64 i, o;
                pair_of_index(p0 + k, Ni, squared, i, o);
                const int e = (int)(((lo >> k) & 1u) | (((hi >> k) & 1u) << 1));
                double cx, cy;
                pair_coefficient(ci[2 * i], ci[2 * i + 1], co[2 * o], co[2 * o + 1], e, cx, cy);
                if (squared && i != o) {
                    if (e & 1) { cx = 0.0; cy = 0.0; }
                    else { cx = __dadd_rn(cx, cx); cy = __dadd_rn(cy, cy); }
                }
                const bool zero = cx == 0.0 && cy == 0.0;
                const bool kp = !use_thr || (zero ? 0.0 > thr : (fabs(cx) > thr || fabs(cy) > thr || hypot(__dadd_rn(0.0, cx), __dadd_rn(0.0, cy)) > thr));
                keep |= (kp ? 1u : 0u) << k;
            }
        }
        mark16[g] = (unsigned short)keep; lo16[g] = (unsigned short)lo; hi16[g] = (unsigned short)hi;
    }
}
// the flagged pairs' packed keys rebuilt in place from their indices (k_compact_suspects with keys == null leaves those), the byte and the
// operand hash tables; *n_flagged is the device-side count (the host does not know it yet)
__global__ __launch_bounds__(256) void k_keys_of_indices(u64 *__restrict__ out, const u32 *__restrict__ n_flagged, const unsigned char *__restrict__ eb, i64 Ni, int squared,
                                                          PackedLayout L, const u64 *__restrict__ hI, const u64 *__restrict__ hO) {
    const u64 hmask = ~((1ULL << L.F()) - 1ULL);
    const i64 n = n_flagged[0];
    for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
        const i64 pos = (i64)out[t];
        i64 i, o;
        pair_of_index(pos, Ni, squared, i, o);
        const u64 e = eb[pos] & 3u;
        out[t] = ((hI[i] ^ hO[o]) & hmask) | (e << (L.bi + L.bo)) | ((u64)o << L.bi) | (u64)i;
    }
}

// lazy mode, after the sort: which 64-position chunks of the sorted keys hold a member of a segment with more than one element?
// A position that equals its predecessor (the test k_heads_sums makes: prefix, then P * P twins, then the full keys rebuilt from the
// operand hash tables) marks its own chunk and its predecessor's.  Four chunks per wavefront and step, all loads of a step in flight.
template <bool PACKED>
__global__ __launch_bounds__(256) void k_find_merges(const u64 *__restrict__ keys, i64 T, const u32 *__restrict__ zero_len, PackedLayout L,
                                                      const u64 *__restrict__ hI, const u64 *__restrict__ hO, int same_operand, u32 *__restrict__ dirtybits) {
    const i64 ZL = zero_len ? (i64)*zero_len : 0;
    const int lane = threadIdx.x & 63;
    const i64 n_steps = (T + 255) / 256;
    for (i64 g = (i64)blockIdx.x * 4 + (threadIdx.x >> 6); g < n_steps; g += (i64)gridDim.x * 4) {
        const i64 base = g * 256;
        u64 k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const i64 sp = base + 64 * j + lane; k[j] = sp < T ? keys[sp] : 0ULL; }
        const u64 prev = base > 0 ? keys[base - 1] : 0ULL;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i64 sp = base + 64 * j + lane;
            u64 k0 = __shfl_up(k[j], 1);
            const u64 carry_in = j > 0 ? __shfl(k[j > 0 ? j - 1 : 0], 63) : prev;
            if (lane == 0) k0 = carry_in;
            const bool valid = sp < T && sp >= ZL;
            bool eq;
            if (PACKED) {
                eq = valid && sp > 0 && (k[j] >> L.F()) == (k0 >> L.F());
                if (eq && !(same_operand && L.i(k[j]) == L.o(k0) && L.o(k[j]) == L.i(k0))) eq = L.full_key(hI, hO, k[j]) == L.full_key(hI, hO, k0);
            } else eq = valid && sp > 0 && k[j] == k0;
            const u64 b = __ballot(eq);
            if (b != 0ULL && lane == 0) {
                const i64 chunk = base / 64 + j;
                atomicOr(&dirtybits[chunk >> 5], 1u << (chunk & 31));
                if ((b & 1ULL) && chunk > 0) atomicOr(&dirtybits[(chunk - 1) >> 5], 1u << ((chunk - 1) & 31));
            }
        }
    }
}

// ---- the identity segment of a squared operator -------------------------------------------------------------------------------
// P * P has N diagonal pairs (i, i), and every one of them is the identity row: N equal keys — with FULL KEY ZERO, because the row
// hash is linear (h(0) = 0) — i.e. one segment of >= N elements at the very start of the sorted array.  k_heads_sums sums a
// segment sequentially, one wavefront walking chunk after chunk: 157 dependent chunk steps for N = 10,000, 0.8 of the kernel's
// 1.64 ms at cfg3, 40 of 44 us at cfg1.  The squared path already associates its sums differently from the reference (twin first:
// exact for dyadic coefficients, rounding-level otherwise), so the zero-key segment is reduced in parallel here, in a FIXED order
// (per-thread ascending positions, then threads, then blocks in order: deterministic), and k_heads_sums starts behind it.
// Members that are not diagonal pairs (duplicate rows in P, or a 64-bit collision) are verified row against row like everywhere.
constexpr int ZB = 1024;                                              // sorted positions per block
__global__ __launch_bounds__(256) void k_zero_partial(const u64 *__restrict__ keys, i64 Tk, const u64 *__restrict__ hI, const u64 *__restrict__ hO,
                                                       PackedLayout L, const u64 *__restrict__ rows, int W, const double *__restrict__ cf,
                                                       double *__restrict__ part, u32 *__restrict__ part_n, u32 *__restrict__ collision,
                                                       u32 Ni, u32 *__restrict__ lazy_markbits) {
    __shared__ double s_re[256], s_im[256];
    __shared__ u32 s_n[256];
    const i64 base = (i64)blockIdx.x * ZB;
    if (L.full_key(hI, hO, keys[base]) != 0ULL) {                     // uniform: the zero keys are a prefix of the sorted array
        if (threadIdx.x == 0) { part[2 * blockIdx.x] = 0.0; part[2 * blockIdx.x + 1] = 0.0; part_n[blockIdx.x] = 0u; }
        return;
    }
    double re = 0.0, im = 0.0;
    u32 n = 0;
    bool mism = false, offdiag = false;
    for (int j = 0; j < ZB / 256; ++j) {
        const i64 p = base + threadIdx.x + 256 * j;
        if (p >= Tk) break;
        const u64 k = keys[p];
        if (L.full_key(hI, hO, k) != 0ULL) break;                     // behind the segment: so is everything this thread has left
        const u32 i = L.i(k), o = L.o(k);
        const int e = L.e(k);
        double cr, cim;
        pair_coefficient(cf[2 * i], cf[2 * i + 1], cf[2 * o], cf[2 * o + 1], e, cr, cim);
        if (i != o) {
            if (e & 1) { cr = 0.0; cim = 0.0; } else { cr = __dadd_rn(cr, cr); cim = __dadd_rn(cim, cim); }
            for (int w = 0; w < W; ++w) mism |= rows[(i64)i * W + w] != rows[(i64)o * W + w];      // identity row <=> equal factors
            offdiag = true;
        }
        re = __dadd_rn(re, cr);
        im = __dadd_rn(im, cim);
        ++n;
        if (lazy_markbits) {                                          // not a single: whatever k_mark_singles decided is void (k_zero_close files the head)
            const u32 slot = tri_slot(o, i, Ni);
            atomicAnd(&lazy_markbits[slot >> 5], ~(1u << (slot & 31u)));
        }
    }
    s_re[threadIdx.x] = re; s_im[threadIdx.x] = im; s_n[threadIdx.x] = n;
    if (mism) atomicOr(collision, 1u);
    if (offdiag) atomicOr(collision + 2, 1u);                          // a member that is not a diagonal pair: P holds duplicate rows
    __syncthreads();
    if (threadIdx.x == 0) {
        double r = 0.0, q = 0.0;
        u32 c = 0;
        for (int t = 0; t < 256; ++t) { r = __dadd_rn(r, s_re[t]); q = __dadd_rn(q, s_im[t]); c += s_n[t]; }
        part[2 * blockIdx.x] = r; part[2 * blockIdx.x + 1] = q; part_n[blockIdx.x] = c;
    }
}
// The identity coefficient of P * P exactly as the reference forms it when P has no duplicate rows: the N diagonal pairs are then the
// only members of the identity segment, and np.add.at adds their coefficients c_i * c_i (the phase exponent of P_i * P_i is 0) to
// 0.0 one after the other in index order (utils.py:273-274).  A sequential sum is sequential: ONE wavefront, 64 products per step
// staged in LDS, lane 0 adds them in order (two independent chains, re and im).  ~12 cycles per term: 50 us at N = 10,000 — on the
// side stream next to the key generation and the sort of the same call, which take milliseconds.
__global__ __launch_bounds__(64) void k_diag_seq_sum(const double *__restrict__ cf, u32 N, double *__restrict__ out) {
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    __shared__ f64x2 s_p[2][64];
    const int lane = threadIdx.x;
    double re = 0.0, im = 0.0;
    int buf = 0;
    for (u32 base = 0; base < N; base += 64, buf ^= 1) {
        const u32 i = base + lane;
        if (i < N) {
            const f64x2 c = reinterpret_cast<const f64x2 *>(cf)[i];
            double pr, pi;
            pair_coefficient(c.x, c.y, c.x, c.y, 0, pr, pi);
            s_p[buf][lane] = f64x2{pr, pi};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane == 0) {
            const u32 m = N - base < 64u ? N - base : 64u;
            if (m == 64u) {
#pragma unroll
                for (int k = 0; k < 64; ++k) { const f64x2 p = s_p[buf][k]; re = __dadd_rn(re, p.x); im = __dadd_rn(im, p.y); }
            } else {
                for (u32 k = 0; k < m; ++k) { const f64x2 p = s_p[buf][k]; re = __dadd_rn(re, p.x); im = __dadd_rn(im, p.y); }
            }
        }
    }
    if (lane == 0) { out[0] = re; out[1] = im; }
}

// blocks in order; files the identity term under the slot of the segment's first element and tells k_heads_sums where to start
__global__ void k_zero_close(const u64 *__restrict__ keys, const double *__restrict__ part, const u32 *__restrict__ part_n, i64 n_blocks,
                             PackedLayout L, u32 Ni, double thr, int use_thr, u32 *__restrict__ markbits, double *__restrict__ sum_of,
                             u32 *__restrict__ zero_len, u32 *__restrict__ patchbits, const double *__restrict__ diag_seq,
                             const u32 *__restrict__ offdiag) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double re = 0.0, im = 0.0;
    u64 len = 0;
    for (i64 b = 0; b < n_blocks; ++b) {
        const u32 c = part_n[b];
        if (c == 0) break;
        re = __dadd_rn(re, part[2 * b]); im = __dadd_rn(im, part[2 * b + 1]);
        len += c;
        if (c < (u32)ZB) break;
    }
    // exactly the N diagonal pairs: the reference's sequential sum (k_diag_seq_sum) instead of the blocked one
    if (diag_seq && len == (u64)Ni && *offdiag == 0u) { re = diag_seq[0]; im = diag_seq[1]; }
    *zero_len = (u32)len;
    if (len == 0) return;
    if (use_thr && !(hypot(re, im) > thr)) return;
    const u64 k0 = keys[0];                                           // stable sort: the segment's smallest pair index
    const u32 first = tri_slot(L.o(k0), L.i(k0), Ni);
    atomicOr(&markbits[first >> 5], 1u << (first & 31u));
    if (patchbits) atomicOr(&patchbits[first >> 5], 1u << (first & 31u));
    sum_of[2 * (i64)first] = re; sum_of[2 * (i64)first + 1] = im;
}

template <bool PAIR, bool PACKED>
__global__ __launch_bounds__(256) void k_heads_sums(const u64 *__restrict__ keys, const u32 *__restrict__ idx, i64 T, const u64 *__restrict__ rows, int W,
                                                     const u64 *__restrict__ inner, u32 Ni, const u64 *__restrict__ outer, int G,
                                                     const double *__restrict__ coeff, u32 *__restrict__ collision,
                                                     const u64 *__restrict__ hI, const u64 *__restrict__ hO, PackedLayout L,
                                                     const double *__restrict__ ci, const double *__restrict__ co,
                                                     double thr, int use_thr, u32 *__restrict__ markbits, double *__restrict__ sum_of,
                                                     i64 chunks_per_wave, int squared, const u32 *__restrict__ zero_len = nullptr,
                                                     u32 *__restrict__ patchbits = nullptr, const u32 *__restrict__ dirtybits = nullptr) {
    // patchbits != null ("lazy" mode, see k_mark_singles): one-element segments are not touched at all, and wavefront w only works on
    // the chunks whose bit is set in dirtybits[w] (k_find_merges: the chunks that hold a member of a segment of more than one element)
    // zero_len (squared operators): the first *zero_len sorted positions are the identity segment, already reduced by k_zero_partial /
    // k_zero_close — they are treated like positions past the end (they end every run and contribute nothing)
    const i64 ZL = zero_len ? (i64)*zero_len : 0;
    // squared (PACKED only, P * P): the keys are the pairs with i >= o; an off-diagonal pair stands for itself and its twin
    // (o, i), whose coefficient is bit-identical in magnitude (IEEE products and sums commute): it counts twice if the two terms
    // commute (e even) and not at all if they anticommute (e odd) — it then only marks the first occurrence of its row.
    const int lane = threadIdx.x & 63;
    const int gi = lane / G, gl = lane % G;
    const int C = W / 2;                                             // 16-byte chunks per row
    const u64 gmask = G == 64 ? ~0ULL : (((1ULL << G) - 1ULL) << (gi * G));
    const i64 n_chunks = (T + 63) / 64;
    const i64 gw = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
    u32 dirty = 0;
    if (dirtybits) {                                                // eight chunks per wavefront: a quarter of a bitmap word
        if (gw * 8 >= n_chunks) return;
        dirty = (__builtin_amdgcn_readfirstlane(dirtybits[gw >> 2]) >> (8 * (int)(gw & 3))) & 0xFFu;
        if (dirty == 0) return;
    }

    const bool lazy = patchbits != nullptr;
    auto close = [&](u32 first, double re, double im, bool multi) {   // strict threshold, bitmap, sum filed under the first index
        if (lazy && !multi) return;                                   // a single: decided by k_mark_singles, rebuilt by k_emit_meta
        if (use_thr && !(hypot(re, im) > thr)) {
            if (lazy) atomicAnd(&markbits[first >> 5], ~(1u << (first & 31u)));
            return;
        }
        atomicOr(&markbits[first >> 5], 1u << (first & 31u));
        if (lazy) atomicOr(&patchbits[first >> 5], 1u << (first & 31u));
        double2 o; o.x = re; o.y = im;
        reinterpret_cast<double2 *>(sum_of)[first] = o;
    };

    bool open = false;                  // wave-uniform: a segment is carried across chunk boundaries
    double are = 0.0, aim = 0.0;        // its running sum
    u32 afirst = 0;                     // input index of its first (head) element
    bool amulti = false;                // it has more than one element so far
    bool mism = false;
    for (;;) {                          // the wavefront's chunk ranges: one, or — dirtybits — one per set bit
    i64 c0, c1;
    if (dirtybits) {
        if (dirty == 0) break;
        c0 = gw * 8 + __builtin_ctz(dirty);
        dirty &= dirty - 1;
        c1 = c0 + 1;
        open = false;
    } else {
        c0 = gw * chunks_per_wave;
        if (c0 >= n_chunks) break;
        c1 = c0 + chunks_per_wave < n_chunks ? c0 + chunks_per_wave : n_chunks;
    }
    for (i64 chunk = c0;; ++chunk) {
        if (chunk >= c1 && !open) break;
        if (chunk >= n_chunks) {        // the carried segment ends with the data
            if (lane == 0) close(afirst, are, aim, amulti);
            break;
        }
        const bool own = chunk < c1;    // beyond the own range: decode only, to finish the carried segment
        if (!own) {
            // Peek before decoding a foreign chunk: the carried segment only continues if the chunk's FIRST key equals the key before
            // it — almost never (an operator without duplicate rows merges nothing).  Two wave-uniform loads instead of 64 keys,
            // 128 coefficient-table gathers and the sum loop: with one chunk per wavefront this second pass was half of the kernel.
            const i64 s0 = chunk * 64;                                // 0 < s0 < T
            const u64 kb = keys[s0], ka = keys[s0 - 1];
            bool eq0;
            if (PACKED) {
                eq0 = (kb >> L.F()) == (ka >> L.F());
                if (eq0 && !(inner == outer && L.i(kb) == L.o(ka) && L.o(kb) == L.i(ka))) eq0 = (hI[L.i(kb)] ^ hO[L.o(kb)]) == (hI[L.i(ka)] ^ hO[L.o(ka)]);
            } else eq0 = kb == ka;
            if (!eq0) {
                if (lane == 0) close(afirst, are, aim, amulti);
                break;
            }
        }
        const i64 s = chunk * 64 + lane;
        const bool valid = s < T && s >= ZL;
        // this position: key k1, input index t1 (PAIR: as (i1, o1)); predecessor k0 / t0 / (i0, o0) from the neighbour lane
        // (lane 0 reads position s-1 itself)
        u64 k1 = valid ? keys[s] : 0ULL;
        u64 k0 = __shfl_up(k1, 1);
        if (lane == 0 && valid && s > 0) k0 = keys[s - 1];
        u32 t1 = 0, i1 = 0, o1 = 0, t0 = 0, i0 = 0, o0 = 0;
        bool eq;
        if (PACKED) {
            i1 = L.i(k1); o1 = L.o(k1); i0 = L.i(k0); o0 = L.o(k0);
            // index under which the term is filed: the pair index, or — squared mode — its slot in the compacted key order (the
            // same order, half the index space: bitmap and sums stay dense)
            t1 = squared ? tri_slot(o1, i1, Ni) : o1 * Ni + i1;
            // equal 64-bit keys?  Different hash prefixes: no.  P * P twins (i, o) / (o, i): yes, the two hash tables are the
            // same.  Otherwise (about 1 % of the positions) the full keys are rebuilt from the operand hash tables.
            eq = valid && s > 0 && (k1 >> L.F()) == (k0 >> L.F());
            if (eq && !(inner == outer && i1 == o0 && o1 == i0)) eq = (hI[i1] ^ hO[o1]) == (hI[i0] ^ hO[o0]);
        } else {
            if (valid) t1 = idx[s];
            t0 = __shfl_up(t1, 1);
            if (lane == 0 && valid && s > 0) t0 = idx[s - 1];
            if (PAIR) { o1 = t1 / Ni; i1 = t1 - o1 * Ni; o0 = t0 / Ni; i0 = t0 - o0 * Ni; }
            eq = valid && s > 0 && k1 == k0;
        }
        double2 c; c.x = 0.0; c.y = 0.0;
        // lazy: only members of segments with more than one element need their coefficient — a position that equals its predecessor,
        // one whose successor equals it, and lane 63 (its successor is in the next chunk)
        bool need = valid;
        if (lazy) {
            const int eq_next = __shfl_down((int)eq, 1);
            need = valid && (eq || eq_next || lane == 63);
            if (valid && eq) atomicAnd(&markbits[t1 >> 5], ~(1u << (t1 & 31u)));      // a follower is never the first occurrence
        }
        if (need) {
            if (PACKED) {
                pair_coefficient(ci[2 * i1], ci[2 * i1 + 1], co[2 * o1], co[2 * o1 + 1], L.e(k1), c.x, c.y);
                if (squared && i1 != o1) {
                    if (L.e(k1) & 1) { c.x = 0.0; c.y = 0.0; }
                    else { c.x = __dadd_rn(c.x, c.x); c.y = __dadd_rn(c.y, c.y); }
                }
            } else c = reinterpret_cast<const double2 *>(coeff)[t1];
        }
        if (own) {
            // exact verification of the equal-key neighbours.  P * P: row(i, o) == row(o, i) by commutativity of XOR when both
            // operands are the same array — nothing to read
            const bool trivially_equal = PAIR && inner == outer && i1 == o0 && o1 == i0;
            u64 sub = __ballot(eq && !trivially_equal) & gmask;      // this group's candidates
            while (__ballot(sub != 0ULL)) {                          // wave-uniform
                const bool act = sub != 0ULL;
                const int p = act ? __builtin_ctzll(sub) : 0;
                sub &= sub - 1;
                if (PAIR) {
                    const i64 ci1 = __shfl(i1, p), co1 = __shfl(o1, p), ci0 = __shfl(i0, p), co0 = __shfl(o0, p);
                    if (act) {
                        const u32x4 *r1 = reinterpret_cast<const u32x4 *>(inner + ci1 * W), *q1 = reinterpret_cast<const u32x4 *>(outer + co1 * W);
                        const u32x4 *r0 = reinterpret_cast<const u32x4 *>(inner + ci0 * W), *q0 = reinterpret_cast<const u32x4 *>(outer + co0 * W);
                        for (int cc = gl; cc < C; cc += G) mism |= differs(r1[cc] ^ q1[cc], r0[cc] ^ q0[cc]);
                    }
                } else {
                    const i64 a1 = __shfl(t1, p), a0 = __shfl(t0, p);
                    if (act) {
                        const u32x4 *r1 = reinterpret_cast<const u32x4 *>(rows + a1 * W), *r0 = reinterpret_cast<const u32x4 *>(rows + a0 * W);
                        for (int cc = gl; cc < C; cc += G) mism |= differs(r1[cc], r0[cc]);
                    }
                }
            }
        }
        // ---- segment sums on the head flags (positions past the end act as heads: they end every run) ----
        const u64 m = __ballot(!valid || !eq);
        const int lead = m ? __builtin_ctzll(m) : 64;             // leading non-head lanes continue the carried segment
        if (open) {
            for (int k = 0; k < lead; ++k) {
                are = __dadd_rn(are, __shfl(c.x, k));
                aim = __dadd_rn(aim, __shfl(c.y, k));
            }
            if (lead > 0) amulti = true;
        }
        if (m == 0ULL) continue;                                  // no head in this chunk
        if (open) {
            if (lane == 0) close(afirst, are, aim, amulti);
            open = false;
        }
        if (!own) break;                                          // beyond the own range only the carry had to be closed
        const bool is_head = valid && !eq;
        const u64 above = lane == 63 ? 0ULL : (m >> (lane + 1));
        const int run = above ? __builtin_ctzll(above) : 63 - lane;   // non-head lanes that follow this lane in the chunk
        double re = __dadd_rn(0.0, c.x), im = __dadd_rn(0.0, c.y);
        for (int k = 1; __ballot(is_head && k <= run); ++k) {
            const double vx = __shfl_down(c.x, k), vy = __shfl_down(c.y, k);
            if (is_head && k <= run) {
                re = __dadd_rn(re, vx);
                im = __dadd_rn(im, vy);
            }
        }
        // the last head of a full chunk may continue in the next chunk: carry it; everything else closes here
        const int last = 63 - __builtin_clzll(m);
        // a full chunk whose last head is a real position: its run reaches lane 63 and may continue in the next chunk
        const bool carry = chunk * 64 + 64 <= T && ((__ballot(valid) >> last) & 1ULL);
        if (is_head && !(carry && lane == last)) close(t1, re, im, run > 0);
        if (carry) {
            open = true;
            are = __shfl(re, last);
            aim = __shfl(im, last);
            afirst = __shfl(t1, last);
            amulti = __shfl(run, last) > 0;
        }
    }
    if (!dirtybits) break;
    }
    if (__ballot(mism) && lane == 0) atomicOr(collision, 1u);
}

// Truncated sort fix-up.  The radix sort only orders the top `nb` key bits (random hash bits: ~log2(T)+5..12 of them already
// separate almost all distinct keys).  Inside a run of equal prefixes the elements are still in input order; if such a
// run holds more than one distinct key it is re-ordered here by (full key, input order) with a stable insertion sort.
// Runs longer than FIX_MAX that are not uniform raise `fallback`: the caller then redoes a full 64-bit sort.
constexpr int FIX_MAX = 48;
// Round 3: the keys that need a look at all — a key inside a run that differs from its predecessor (round 6: no longer every run's first
// key as well: an input full of repeated rows has a run start at every third position, and walking all those uniform runs was a quarter of a
// plain cleanup, 60 of 237 us at 10^5 rows) — are flagged by a streaming pass (k_fixup_find: four chunks per wavefront and step,
// one 64-bit word of flags per chunk, plain stores: appending to ONE list counter instead serialises 4e5 returning atomics on one
// address, 2.8 ms) and worked off by wavefronts that expand the flags of 4,096 positions into a dense list (k_fixup_work): the thread
// of a flagged key measures its run by the prefixes, raises `fallback` if it is longer than FIX_MAX (a long UNIFORM run — the identity
// segment of a squared operator — has no flagged key and costs nothing) and, if no earlier member of the run is flagged, sorts it.
// Threads of one run may read keys while its first flagged member reorders them: all of them share the prefix, which is all the others
// look at (who is first is read from the flags, which nobody writes here).  (The two launches this replaces walked the runs from inside the streaming pass: 0.33 ms at cfg3, now 0.11.)
// PACKED: keys are packed pair keys (full key recomputed from the (i, o) fields), there is no separate idx array.
template <bool PACKED>
__device__ __forceinline__ bool fixup_differ(u64 k, u64 kp, const u64 *__restrict__ hI, const u64 *__restrict__ hO, const PackedLayout &L, bool same_operand) {
    if (PACKED) {
        if (same_operand && L.i(k) == L.o(kp) && L.o(k) == L.i(kp)) return false;         // P * P twins: equal keys by construction
        return L.full_key(hI, hO, k) != L.full_key(hI, hO, kp);
    }
    return k != kp;
}
template <bool PACKED>
__global__ __launch_bounds__(256) void k_fixup_find(const u64 *__restrict__ keys, i64 T, int shift, const u64 *__restrict__ hI, const u64 *__restrict__ hO,
                                                     PackedLayout L, bool same_operand, u64 *__restrict__ rarebits, u32 *__restrict__ dirtybits) {
    // dirtybits (lazy cleanup): a key that EQUALS its predecessor is a merged term — its chunk and its predecessor's are the ones
    // k_heads_sums has to visit (k_find_merges' job, done here in the same pass; a run that k_fixup_work reorders is marked again there)
    const int lane = threadIdx.x & 63;
    const i64 n_steps = (T + 255) / 256;
    for (i64 g = (i64)blockIdx.x * 4 + (threadIdx.x >> 6); g < n_steps; g += (i64)gridDim.x * 4) {
        const i64 base = g * 256;
        u64 k[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const i64 sp = base + 64 * j + lane; k[j] = sp < T ? keys[sp] : 0ULL; }
        const u64 prev = base > 0 ? keys[base - 1] : 0ULL;
        const u64 next = base + 256 < T ? keys[base + 256] : 0ULL;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const i64 sp = base + 64 * j + lane;
            u64 kp = __shfl_up(k[j], 1), kn = __shfl_down(k[j], 1);
            const u64 in_lo = j > 0 ? __shfl(k[j > 0 ? j - 1 : 0], 63) : prev;
            const u64 in_hi = j < 3 ? __shfl(k[j < 3 ? j + 1 : 3], 0) : next;
            if (lane == 0) kp = in_lo;
            if (lane == 63) kn = in_hi;
            const bool valid = sp < T;
            const bool with_prev = valid && sp > 0 && (kp >> shift) == (k[j] >> shift);
            const bool with_next = valid && sp + 1 < T && (kn >> shift) == (k[j] >> shift);
            (void)with_next;
            const bool rare = with_prev && fixup_differ<PACKED>(k[j], kp, hI, hO, L, same_operand);
            const u64 b = __ballot(rare);
            if (lane == 0 && base + 64 * j < T) rarebits[base / 64 + j] = b;
            if (dirtybits) {
                const u64 m = __ballot(with_prev && !rare);
                if (m != 0ULL && lane == 0) {
                    const i64 chunk = base / 64 + j;
                    atomicOr(&dirtybits[chunk >> 5], 1u << (chunk & 31));
                    if ((m & 1ULL) && chunk > 0) atomicOr(&dirtybits[(chunk - 1) >> 5], 1u << ((chunk - 1) & 31));
                }
            }
        }
    }
}
template <bool PACKED>
__global__ __launch_bounds__(256) void k_fixup_work(u64 *__restrict__ keys, u32 *__restrict__ idx, i64 T, int shift, const u64 *__restrict__ rarebits,
                                                     u32 *__restrict__ fallback, const u64 *__restrict__ hI,
                                                     const u64 *__restrict__ hO, PackedLayout L, bool same_operand, u32 *__restrict__ dirtybits) {
    __shared__ unsigned short s_list[4][4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned short *list = s_list[wave];
    const i64 n_chunks = (T + 63) / 64;
    const i64 cw = ((i64)blockIdx.x * 4 + wave) * 64 + lane;                  // this lane's chunk: 64 chunks = 4,096 positions per wavefront
    u64 bits = cw < n_chunks ? rarebits[cw] : 0ULL;
    const u32 cnt = (u32)__popcll(bits);
    u32 incl = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    const u32 n = __shfl(incl, 63);
    if (n == 0) return;                                                       // wave-uniform
    {
        u32 at = incl - cnt;
        while (bits) { list[at++] = (unsigned short)(lane * 64 + __builtin_ctzll(bits)); bits &= bits - 1; }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const i64 wbase = ((i64)blockIdx.x * 4 + wave) * 4096;
    for (u32 it = lane; it < n; it += 64) {
        const i64 sp = wbase + list[it];
        const u64 k = keys[sp];                                               // (only its prefix is used: the run's first flagged member may be moving keys)
        // the run's start and end (prefixes do not change under the reordering)
        i64 b = sp, e = sp + 1;
        while (b > 0 && sp - b <= FIX_MAX && (keys[b - 1] >> shift) == (k >> shift)) --b;
        while (e < T && e - b <= FIX_MAX && (keys[e] >> shift) == (k >> shift)) ++e;
        if (e - b > FIX_MAX) { atomicOr(fallback, 1u); continue; }           // a long run that is not uniform: the caller sorts completely
        bool first = true;                                                    // the run's FIRST flagged member reorders it (the flags are not written here)
        for (i64 j = b + 1; j < sp; ++j)
            if ((rarebits[j >> 6] >> (j & 63)) & 1ULL) { first = false; break; }
        if (!first) continue;
        for (i64 a = b + 1; a < e; ++a) {                                     // stable insertion sort by full key
            const u64 ka = keys[a];
            if (PACKED) {
                const u64 fa = L.full_key(hI, hO, ka);
                i64 c = a - 1;
                while (c >= b && L.full_key(hI, hO, keys[c]) > fa) { keys[c + 1] = keys[c]; --c; }
                keys[c + 1] = ka;
            } else {
                const u32 ia = idx[a];
                i64 c = a - 1;
                while (c >= b && keys[c] > ka) { keys[c + 1] = keys[c]; idx[c + 1] = idx[c]; --c; }
                keys[c + 1] = ka;
                idx[c + 1] = ia;
            }
        }
        if (dirtybits)                                                        // the merged terms of the reordered run, where they are now
            for (i64 a = b + 1; a < e; ++a)
                if (!fixup_differ<PACKED>(keys[a], keys[a - 1], hI, hO, L, same_operand)) {
                    const i64 ca = a / 64, cb = (a - 1) / 64;
                    atomicOr(&dirtybits[ca >> 5], 1u << (ca & 31));
                    if (cb != ca) atomicOr(&dirtybits[cb >> 5], 1u << (cb & 31));
                }
    }
}

__global__ __launch_bounds__(256) void k_count_bits(const u32 *__restrict__ bits, i64 n_words, u32 *__restrict__ total) {
    u32 c = 0;
    for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (i64)gridDim.x * blockDim.x) c += (u32)__popc(bits[w]);
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(total, c);
}
__global__ void k_popc_words(const u32 *__restrict__ bits, i64 n_words, u32 *__restrict__ counts) {
    for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (i64)gridDim.x * blockDim.x) counts[w] = (u32)__popc(bits[w]);
}

// Output stage without any scatter.  The kept terms are the set bits of `markbits` (input index space), and input order IS the
// output order.  Two kernels per BATCH of 2^17 bitmap words (4M input indices; emit_batch_words()):
//  * k_emit_meta: a wavefront takes up to 64 bitmap words, expands their set bits into a compact list in LDS (output slot of the
//    k-th one = word prefix + k), and every lane files one kept term: its summed coefficient (read from where k_heads_sums filed
//    it, nearly sequential) at its output slot and its source (i, o) — 8 bytes — in the batch's list, coalesced.
//  * k_emit_stream: the rows.  Block b writes the b-th 4 KiB of the batch's output — the sequential pattern of the product's row
//    stream (product.hip), which is what the HBM wants.
// Why batches: a saturated HBM write stream tolerates reads that hit in the L2s or in the Infinity Cache, but HBM READS mixed into
// it halve it (tools/ubench_fused.hip, S3: the same stream kernel writes 6.1-6.3 TB/s while its 8-byte-per-row list comes out of
// the Infinity Cache and 3.7-4.0 TB/s when the list was evicted by 256 MB of other traffic or is read with `nt` loads, which
// bypass that cache).  The first version streamed the rows from the wavefronts that decode the bitmap, next to their HBM reads of
// the filed sums (4.2 TB/s); a whole-operator list written first and streamed afterwards is out of the cache again by the time
// it is read (3.8 TB/s).  A batch's list is 32 MB at most, written by one kernel and read by the next.
// TRI (PAIR only): the index space is the compacted slot order of a squared operator (tri_slot / tri_pair).
static i64 emit_batch_words() {                                  // SYMGPU_EMIT_BATCH = log2(bitmap words per batch), default 17 (4M indices)
    const i64 w = [] { const char *e = SG_TUNE("SYMGPU_EMIT_BATCH"); const int l = e ? atoi(e) : 17; return (i64)1 << (l >= 6 && l <= 26 ? l : 17); }();
    return w;
}
// where k_emit_meta takes a kept term's coefficient from: mode 0 = the filed sums only; 1 / 2 = filed sums for patched terms, the
// operand tables (packed products) / the input coefficients (indexed operators) for all others (k_mark_singles)
struct LazyEmit {
    int mode = 0, squared = 0, no_one_outer = 0;
    const u32 *patchbits = nullptr, *e_lo = nullptr, *e_hi = nullptr;
    const double *ci = nullptr, *co = nullptr, *coeff = nullptr;
};
template <bool PAIR, bool TRI>
__global__ __launch_bounds__(256) void k_emit_meta(const u32 *__restrict__ markbits, const u32 *__restrict__ wordprefix, i64 w_begin, i64 w_end,
                                                    const double *__restrict__ sum_of, int wpw, u32 Ni,
                                                    uint2 *__restrict__ meta, double *__restrict__ out_coeff, LazyEmit lz, u64 *__restrict__ out_first) {
    __shared__ unsigned short s_list[4][2048];                       // offsets inside the chunk (< 2048)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned short *list = s_list[wave];
    const u32 P0 = wordprefix[w_begin];                              // first output slot of the batch
    // wpw (power of two <= 64) bitmap words per wavefront and step: narrow chunks so that enough waves exist
    const i64 n_chunks = (w_end - w_begin + wpw - 1) / wpw;
    for (i64 chunk = (i64)blockIdx.x * 4 + wave; chunk < n_chunks; chunk += (i64)gridDim.x * 4) {
        const i64 w = w_begin + chunk * wpw + lane;
        const u32 bits = (lane < wpw && w < w_end) ? markbits[w] : 0u;
        const u32 cnt = (u32)__popc(bits);
        u32 incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 v = __shfl_up(incl, off);
            if (lane >= off) incl += v;
        }
        const u32 K = __shfl(incl, 63);
        if (K == 0) continue;                                        // wave-uniform
        const u32 p_base = __shfl(wordprefix[w_begin + chunk * wpw], 0);
        {
            u32 b = bits, k = incl - cnt;
            while (b) { list[k++] = (unsigned short)(lane * 32 + __builtin_ctz(b)); b &= b - 1; }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u32 t_base = (u32)((w_begin + chunk * wpw) * 32);
        for (u32 k = lane; k < K; k += 64) {
            const u32 t = t_base + list[k];
            u32 ti = t, to = 0;
            if (PAIR && TRI) tri_pair(t, Ni, to, ti);
            else if (PAIR) { to = t / Ni; ti = t - to * Ni; }
            double2 cf;
            if (lz.mode == 0 || ((lz.patchbits[t >> 5] >> (t & 31u)) & 1u)) cf = reinterpret_cast<const double2 *>(sum_of)[t];
            else if (lz.mode == 1) {                                  // a single of a packed product: c_i * c_o * i^e from the operand tables
                const int e = (int)(((lz.e_lo[t >> 5] >> (t & 31u)) & 1u) | (((lz.e_hi[t >> 5] >> (t & 31u)) & 1u) << 1));
                double cx, cy;
                pair_coefficient(lz.ci[2 * ti], lz.ci[2 * ti + 1], lz.co[2 * to], lz.co[2 * to + 1], e, cx, cy);
                if (lz.squared && ti != to) {
                    if (e & 1) { cx = 0.0; cy = 0.0; }
                    else { cx = __dadd_rn(cx, cx); cy = __dadd_rn(cy, cy); }
                }
                cf.x = __dadd_rn(0.0, cx); cf.y = __dadd_rn(0.0, cy);
            } else {                                                  // a single of an indexed operator: 0.0 + its own coefficient
                const double2 c0 = reinterpret_cast<const double2 *>(lz.coeff)[t];
                cf.x = __dadd_rn(0.0, c0.x); cf.y = __dadd_rn(0.0, c0.y);
            }
            reinterpret_cast<double2 *>(out_coeff)[(i64)p_base + k] = cf;
            meta[(p_base - P0) + k] = make_uint2(ti, to);
            if (out_first) out_first[(i64)p_base + k] = PAIR ? (((u64)to << 32) | ti) : (u64)t;
        }
        __builtin_amdgcn_wave_barrier();                             // the list is rewritten by the next chunk
    }
}

// RC 16-byte chunks per lane, 256 apart: output chunk f = slot * Wq + c  <-  inner[i][c] ^ outer[o][c]   (plain mode: rows[i][c])
// for the slots [*p_begin, *p_end) of the batch; the grid is sized for a batch whose every index is kept, surplus blocks exit.
template <bool PAIR, int RC>
__global__ __launch_bounds__(256) void k_emit_stream(const uint2 *__restrict__ meta, const u32 *__restrict__ p_begin, const u32 *__restrict__ p_end,
                                                      int Wq, int wsh, const u32x4 *__restrict__ rows, const u32x4 *__restrict__ inner,
                                                      const u32x4 *__restrict__ outer, u32x4 *__restrict__ out_rows) {
    const i64 P0 = *p_begin;
    const i64 n_chunks16 = ((i64)*p_end - P0) * Wq;
    const i64 b0 = (i64)blockIdx.x * (256 * RC);
    if (b0 >= n_chunks16) return;
    const i64 f0 = b0 + threadIdx.x;
    const i64 last = n_chunks16 - 1;
    uint2 m[RC];
    int c[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k) {
        const i64 f = f0 + 256 * k < last ? f0 + 256 * k : last;      // clamped: every load is unconditional
        const i64 slot = wsh >= 0 ? f >> wsh : f / Wq;
        c[k] = (int)(f - slot * Wq);
        m[k] = meta[slot];                                            // plain load: served by the Infinity Cache (see above)
    }
    u32x4 v[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k)
        v[k] = PAIR ? (inner[(i64)m[k].x * Wq + c[k]] ^ outer[(i64)m[k].y * Wq + c[k]]) : rows[(i64)m[k].x * Wq + c[k]];
    u32x4 *dst = out_rows + P0 * Wq + f0;
    if (b0 + 256 * RC <= n_chunks16) {
#pragma unroll
        for (int k = 0; k < RC; ++k) __builtin_nontemporal_store(v[k], dst + 256 * k);
    } else {
#pragma unroll
        for (int k = 0; k < RC; ++k)
            if (f0 + 256 * k < n_chunks16) __builtin_nontemporal_store(v[k], dst + 256 * k);
    }
}

// Fused output stage (round 3): ONE launch, one 64-bit bitmap word (64 indices) per wavefront.  The set lanes file their term — source
// (i, o) into LDS, coefficient to its output slot — and the wavefront then streams the K kept rows, 64 consecutive 16-byte chunks per
// step, into [prefix, prefix + K): consecutive wavefronts and workgroups write consecutive pieces (a workgroup's four words make
// ~32 KB at n = 1000), so the chip still writes one moving window.  No list, no batches, no second launch: with the sums of the
// singles no longer read back from HBM (k_mark_singles) nothing but the bitmaps is read beside the row stream.
// NW = bitmap words per wavefront (4: the dependent chain bitmap -> prefix -> phase / patch bits -> operand tables is paid once per 256
// indices; a wavefront per word spent two thirds of its life in it: 4.5 TB/s), U = steps of 64 chunks in flight
template <bool PAIR, bool TRI, int NW, int U>
__global__ __launch_bounds__(256) void k_emit_fused(const u64 *__restrict__ markbits64, const u32 *__restrict__ wordprefix, i64 T, const double *__restrict__ sum_of,
                                                     u32 Ni, int Wq, int wsh, const u32x4 *__restrict__ rows, const u32x4 *__restrict__ inner,
                                                     const u32x4 *__restrict__ outer, u32x4 *__restrict__ out_rows, double *__restrict__ out_coeff, LazyEmit lz,
                                                     u64 *__restrict__ out_first, int pfx_shift /* prefix entries per 64 indices: 1 << pfx_shift */,
                                                     int interleaved) {
    __shared__ u32 s_i[4][64 * NW], s_o[4][64 * NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // XCD-aware order: workgroups go round-robin over the 8 XCDs; every XCD takes a CONTIGUOUS eighth of the bitmap (and of the output) in
    // order instead of every eighth 64 KB piece — 1.13-1.18 -> 1.04 ms at cfg3 (the output stage's write stream reaches 6.5 TB/s) on the
    // arenas that like it and 1.20 ms on those that do not (round 5: the mode follows the memory the driver hands out).  `interleaved`
    // is the plain order (1.10-1.19 ms everywhere); emit_order() measures both on the buffer at hand and keeps the faster.
    const i64 per_xcd = ((i64)gridDim.x + 7) / 8;
    const i64 blk = interleaved ? (i64)blockIdx.x : (i64)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    const i64 w0 = (blk * 4 + wave) * NW;
    if (w0 * 64 >= T) return;
    u64 bits[NW];
    u32 off[NW + 1];
    off[0] = 0;
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const i64 w = w0 + u;
        bits[u] = w * 64 < T ? markbits64[w] : 0ULL;
        if (w * 64 < T && T - w * 64 < 64) bits[u] &= (1ULL << (T - w * 64)) - 1ULL;   // the bitmap's last word may be half written (32-bit words)
        off[u + 1] = off[u] + (u32)__popcll(bits[u]);
    }
    const u32 K = off[NW];
    if (K == 0) return;                                              // wave-uniform
    const i64 p_base = wordprefix[w0 << pfx_shift];
    u32 o_min = 0xFFFFFFFFu, o_max = 0u;                             // this lane's kept terms: range of their outer indices
    // Nothing the coefficients need depends on the bitmap's CONTENT: the pair (i, o) of an index, its patch / phase-exponent bits and its
    // operands' coefficients are addressed by the index alone.  They are therefore fetched for all NW x 64 indices at once, next to the
    // bitmap words and the prefix — one memory round trip per wavefront instead of two dependent ones (the prologue alone took 0.25 ms of the
    // stage's 1.19 at cfg3 and did not overlap the row stream).
    u32 ti_[NW], to_[NW], pw_[NW], el_[NW], eh_[NW];
    double2 ca_[NW], cb_[NW], cs_[NW];
    // The pair of the wavefront's FIRST index costs a square root (triangular slots) or a division; every other index of the wavefront is a
    // few additions away: consecutive indices walk along a row of the (o, i) plane and step to the next row when i reaches Ni (the per-index
    // tri_pair of all NW x 64 indices was most of the prologue's 0.25 ms: ~150 instructions each).
    u32 o_first = 0, i_first = 0;
    if (PAIR) {
        const u32 t0 = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(w0 * 64));
        if (TRI) tri_pair(t0, Ni, o_first, i_first);
        else { o_first = t0 / Ni; i_first = t0 - o_first * Ni; }
        o_first = (u32)__builtin_amdgcn_readfirstlane((int)o_first); i_first = (u32)__builtin_amdgcn_readfirstlane((int)i_first);
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        const i64 tt = (w0 + u) * 64 + lane;
        const u32 t = (u32)(tt < T ? tt : T - 1);                     // clamped: the surplus lanes of the last word load valid addresses
        ti_[u] = t; to_[u] = 0;
        if (PAIR) {
            u32 o = o_first;
            u64 i = (u64)i_first + (t - (u32)(w0 * 64));
            // row o holds the indices i in [TRI ? o : 0, Ni): step rows while i runs past the end (0 or 1 steps except in the last short rows)
            while (i >= Ni) { i -= Ni; ++o; if (TRI) i += o; }
            to_[u] = o; ti_[u] = (u32)i;
        }
        pw_[u] = lz.mode != 0 ? lz.patchbits[t >> 5] : 0xFFFFFFFFu;
        if (lz.mode == 1) {
            el_[u] = lz.e_lo[t >> 5]; eh_[u] = lz.e_hi[t >> 5];
            ca_[u] = reinterpret_cast<const double2 *>(lz.ci)[ti_[u]];
            cb_[u] = reinterpret_cast<const double2 *>(lz.co)[to_[u]];
        } else if (lz.mode == 2) {
            cs_[u] = reinterpret_cast<const double2 *>(lz.coeff)[t];
        }
    }
#pragma unroll
    for (int u = 0; u < NW; ++u) {
        if ((bits[u] >> lane) & 1ULL) {
            const u32 rank = off[u] + (u32)__popcll(bits[u] & ((1ULL << lane) - 1ULL));
            const u32 t = (u32)((w0 + u) * 64 + lane);
            const u32 ti = ti_[u], to = to_[u];
            double2 cf;
            if (lz.mode == 0 || ((pw_[u] >> (t & 31u)) & 1u)) cf = reinterpret_cast<const double2 *>(sum_of)[t];
            else if (lz.mode == 1) {                                  // a single of a packed product: c_i * c_o * i^e from the operand tables
                const int e = (int)(((el_[u] >> (t & 31u)) & 1u) | (((eh_[u] >> (t & 31u)) & 1u) << 1));
                double cx, cy;
                pair_coefficient(ca_[u].x, ca_[u].y, cb_[u].x, cb_[u].y, e, cx, cy);
                if (lz.squared && ti != to) {
                    if (e & 1) { cx = 0.0; cy = 0.0; }
                    else { cx = __dadd_rn(cx, cx); cy = __dadd_rn(cy, cy); }
                }
                cf.x = __dadd_rn(0.0, cx); cf.y = __dadd_rn(0.0, cy);
            } else {                                                  // a single of an indexed operator: 0.0 + its own coefficient
                cf.x = __dadd_rn(0.0, cs_[u].x); cf.y = __dadd_rn(0.0, cs_[u].y);
            }
            reinterpret_cast<double2 *>(out_coeff)[p_base + rank] = cf;
            if (out_first) out_first[p_base + rank] = PAIR ? (((u64)to << 32) | ti) : (u64)t;
            s_i[wave][rank] = ti; s_o[wave][rank] = to;
            o_min = to < o_min ? to : o_min; o_max = to > o_max ? to : o_max;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const u32 n_ch = K * (u32)Wq;
    u32x4 *dst = out_rows + p_base * Wq;
    // One outer row for the whole wavefront (a product's terms are ordered by their outer index, so 256 consecutive indices nearly always
    // share it) and a row length that divides 64: the lane's chunk of that row is loop invariant — one gather and one list read per
    // 16 bytes written instead of two each.
    bool one_outer = false;
    if (PAIR && wsh >= 0 && Wq <= 64 && lz.no_one_outer == 0) {
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) {
            const u32 a = (u32)__shfl_xor((int)o_min, sft), b = (u32)__shfl_xor((int)o_max, sft);
            o_min = a < o_min ? a : o_min; o_max = b > o_max ? b : o_max;
        }
        one_outer = o_min == o_max;                                   // wave-uniform
    }
    if (one_outer) {
        const u32x4 oc = outer[(i64)o_min * Wq + (lane & (Wq - 1))];
        for (u32 f0 = 0; f0 < n_ch; f0 += 64 * U) {
            u32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const u32 f = f0 + 64 * u + lane < n_ch ? f0 + 64 * u + lane : n_ch - 1;
                const u32 r = f >> wsh;
                v[u] = inner[(i64)s_i[wave][r] * Wq + (lane & (Wq - 1))] ^ oc;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (f0 + 64 * u + lane < n_ch) __builtin_nontemporal_store(v[u], dst + f0 + 64 * u + lane);
        }
        return;
    }
    for (u32 f0 = 0; f0 < n_ch; f0 += 64 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32 f = f0 + 64 * u + lane < n_ch ? f0 + 64 * u + lane : n_ch - 1;
            const u32 r = wsh >= 0 ? f >> wsh : f / (u32)Wq;
            const u32 c = f - r * (u32)Wq;
            v[u] = PAIR ? (inner[(i64)s_i[wave][r] * Wq + c] ^ outer[(i64)s_o[wave][r] * Wq + c]) : rows[(i64)s_i[wave][r] * Wq + c];
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (f0 + 64 * u + lane < n_ch) __builtin_nontemporal_store(v[u], dst + f0 + 64 * u + lane);
    }
}

static int grid_for(i64 n, int block = 256, int cap = 8192) {
    i64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

static int pow2_group(int W) {
    int g = 1;
    while (g < W && g < 64) g <<= 1;
    return g;
}

// rows that are only compared with each other (see k_hash_rows_mix); `seed` as for the tables: a reseed changes the function
int hash_rows_any(const u64 *rows, i64 T, int W, u64 seed, u64 *out1, u32 *iota) {
    if (T == 0) return SYMGPU_OK;
    if (W >= 64 * 128) {                                            // very long rows: the segmented kernel of the linear hash
        if (iota) { hipLaunchKernelGGL(k_iota_keys_plain, dim3(grid_for(T)), dim3(256), 0, ctx().stream, iota, T); KERNEL_CHECK(); }
        return hash_rows(rows, T, W, out1);
    }
    u64 s = seed * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL;
    s ^= s >> 29; s *= 0xBF58476D1CE4E5B9ULL; s ^= s >> 32;
    u64 keep = ~0ULL;
    if (const char *e = getenv("SYMGPU_HASH_WEAK_ODD"))             // the tables' test hook: an odd seed keeps 4 bits, so rows collide in bulk
        if (e[0] == '1' && (seed & 1)) keep = 0xF000000000000000ULL;
    const int G = pow2_group(W / 2);                                // lanes per row: one 16-byte chunk each (rows of up to 128 words)
    const int rpb = 4 * (256 / G);
    i64 g = (T + rpb - 1) / rpb;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_hash_rows_mix, dim3((unsigned)g), dim3(256), 0, ctx().stream, rows, T, W, G, s, keep, out1, iota);
    KERNEL_CHECK();
    return SYMGPU_OK;
}
int hash_rows(const u64 *rows, i64 T, int W, u64 *out1) {
    if (T == 0) return SYMGPU_OK;
    if (W >= 64 * 128) {                                            // very long rows: segments in parallel (k_hash_rows_long)
        SG_TRY(ensure_xs_pow());
        hipStream_t st = ctx().stream;
        HIP_TRY(hipMemsetAsync(out1, 0, (size_t)T * sizeof(u64), st));
        const int n_seg = ((W + 63) / 64 + LSEG - 1) / LSEG;
        for (i64 t0 = 0; t0 < T; t0 += 65535) {
            const i64 nt = T - t0 < 65535 ? T - t0 : 65535;
            hipLaunchKernelGGL(k_hash_rows_long, dim3((unsigned)((n_seg + 3) / 4), (unsigned)nt), dim3(256), 0, st, rows, t0, W, ctx().hash_tab, ctx().xs_pow, out1);
            KERNEL_CHECK();
        }
        return SYMGPU_OK;
    }
    const int G = pow2_group(W);
    const int rpb = 4 * (256 / G);                                  // k_hash_rows: HU = 4 row groups per step
    i64 g = (T + rpb - 1) / rpb;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_hash_rows, dim3((unsigned)g), dim3(256), 0, ctx().stream, rows, T, W, G, ctx().hash_tab, out1);
    KERNEL_CHECK();
    return SYMGPU_OK;
}

// Reads every 16-byte chunk of a buffer and keeps nothing: run right before the output stage on the bitmaps it decodes (written two
// gigabytes of traffic earlier, so out of the Infinity Cache again) — a saturated HBM WRITE stream tolerates cache hits, but every HBM
// read mixed into it costs the DRAM a turn-around.
struct TouchMaps { const u32x4 *p[5]; i64 n16[5]; };
__global__ __launch_bounds__(256) void k_touch(TouchMaps tm, u32 *__restrict__ sink) {
    u32 acc = 0;
#pragma unroll
    for (int m = 0; m < 5; ++m)
        for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < tm.n16[m]; i += (i64)gridDim.x * 256) { const u32x4 v = tm.p[m][i]; acc |= v.x & v.y & v.z & v.w; }
    if (acc == 0xDEADBEEFu) *sink = acc;                           // (never true for bitmaps of a real run; keeps the loads alive)
}

// zeroes of up to three buffers in ONE launch (byte counts multiples of 4): hipMemsetAsync is a launch per buffer, two when the size is not a
// multiple of its fill kernel's granule, 5 us each on an otherwise idle queue
__global__ __launch_bounds__(256) void k_zero_two(u32 *__restrict__ a, i64 na, u32 *__restrict__ b, i64 nb, u32 *__restrict__ c, i64 nc) {
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < na; i += (i64)gridDim.x * 256) a[i] = 0u;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < nb; i += (i64)gridDim.x * 256) b[i] = 0u;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < nc; i += (i64)gridDim.x * 256) c[i] = 0u;
}
static int zero_two(void *a, size_t bytes_a, void *b, size_t bytes_b, void *c = nullptr, size_t bytes_c = 0) {
    const i64 na = (i64)(bytes_a / 4), nb = b ? (i64)(bytes_b / 4) : 0, nc = c ? (i64)(bytes_c / 4) : 0;
    if (na + nb + nc == 0) return SYMGPU_OK;
    i64 nmax = na > nb ? na : nb;
    nmax = nmax > nc ? nmax : nc;
    hipLaunchKernelGGL(k_zero_two, dim3(grid_for(nmax, 256, 4096)), dim3(256), 0, ctx().stream, static_cast<u32 *>(a), na, static_cast<u32 *>(b), nb,
                       static_cast<u32 *>(c), nc);
    KERNEL_CHECK();
    return SYMGPU_OK;
}

// T = size of the index space the kept terms are filed under (pair indices, or the slots of a squared operator: `tri`)
// word prefix of the kept-term bitmap (the output slot of every 32 indices' first kept term) + the number of kept terms, on the device
// wide: one prefix entry per 64 indices (all the fused output stage reads) instead of per 32: half the elements to count and scan
__global__ void k_popc_words64_tail(const u64 *__restrict__ bits, i64 n64, i64 n32, u32 *__restrict__ counts) {
    for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < n64; w += (i64)gridDim.x * blockDim.x) {
        u64 b = bits[w];
        if (w == n64 - 1 && (n32 & 1)) b &= 0xFFFFFFFFULL;           // the bitmap is written in 32-bit words: the last upper half may be unwritten
        counts[w] = (u32)__popcll(b);
    }
}
static int emit_prefix(const u32 *markbits_p, i64 T, Scratch &wordprefix, Scratch &total, bool wide) {
    hipStream_t st = ctx().stream;
    const i64 n_words = (T + 31) / 32, n64 = (T + 63) / 64;
    const i64 n = wide ? n64 : n_words;
    SG_TRY(wordprefix.alloc((size_t)n * 4));
    SG_TRY(total.alloc(32));                                       // [0] the count, [2] k_touch's sink
    if (wide && n64 <= POPC_SCAN_SMALL_MAX) return popc_scan_small(reinterpret_cast<const u64 *>(markbits_p), n64, n_words, wordprefix.as<u32>(), total.as<u32>());
    if (wide) hipLaunchKernelGGL(k_popc_words64_tail, dim3(grid_for(n64)), dim3(256), 0, st, reinterpret_cast<const u64 *>(markbits_p), n64, n_words, wordprefix.as<u32>());
    else hipLaunchKernelGGL(k_popc_words, dim3(grid_for(n_words)), dim3(256), 0, st, markbits_p, n_words, wordprefix.as<u32>());
    KERNEL_CHECK();
    return exclusive_scan_u32(wordprefix.as<u32>(), wordprefix.as<u32>(), n, total.as<u32>());
}

// pre (optional): the prefix was formed and the count read back by the caller (cleanup_core reads it with its own status words: one host
// round trip instead of two)
struct EmitPrefix { Scratch wordprefix, total; i64 n_out = -1; bool touched = false, wide = false; };
// one launch over the bitmaps the fused output stage decodes (five launches of 2 us each sat 6 us apart behind the host's read-back)
// Which block order the fused output stage takes on THIS output buffer (the XCD-contiguous order is 15 % faster or 3 %
// slower than the plain one depending on the memory behind the buffer, which a process cannot choose).  A large output (>= 512 MB) is
// written into blocks the allocator hands out again and again (a loop of calls alternates between two of them): per block, the first four
// calls alternate the two orders under a pair of events, from the fifth on the block is written in the order whose faster sample won.
// Both orders write the same bytes.  SYMGPU_EMIT_ORDER=0 / 1 pins the contiguous / plain order (tests).
struct EmitOrder { int interleaved = 0; EmitProbe *probe = nullptr; };
static int emit_order_begin(const void *dst, size_t bytes, EmitOrder &eo) {
    Context &c = ctx();
    if (const char *e = getenv("SYMGPU_EMIT_ORDER")) { eo.interleaved = e[0] == '1'; return SYMGPU_OK; }
    if (bytes < ((size_t)512 << 20)) return SYMGPU_OK;
    EmitProbe *p = nullptr, *oldest = &c.emit_probe[0];
    for (auto &q : c.emit_probe) {
        if (q.key == dst) { p = &q; break; }
        if (q.stamp < oldest->stamp) oldest = &q;
    }
    if (!p) { p = oldest; p->key = dst; p->phase = 0; p->best[0] = p->best[1] = 1e30f; }
    p->stamp = ++c.emit_stamp;
    if (p->phase > 0 && p->phase <= 4) {                          // the sample of this block's previous call (two or more calls ago: complete)
        float ms = 0;
        if (hipEventSynchronize(p->ev[1]) == hipSuccess && hipEventElapsedTime(&ms, p->ev[0], p->ev[1]) == hipSuccess) {
            const int o = (p->phase - 1) & 1;
            if (ms < p->best[o]) p->best[o] = ms;
        } else (void)hipGetLastError();
        if (p->phase == 4) { p->choice = p->best[1] < p->best[0] ? 1 : 0; p->phase = 5; }
    }
    if (p->phase >= 5) { eo.interleaved = p->choice; return SYMGPU_OK; }
    if (!p->ev[0]) { HIP_TRY(hipEventCreate(&p->ev[0])); HIP_TRY(hipEventCreate(&p->ev[1])); }
    eo.interleaved = p->phase & 1;                                // 0, 1, 0, 1
    eo.probe = p;
    HIP_TRY(hipEventRecord(p->ev[0], c.stream));
    return SYMGPU_OK;
}
static void emit_order_end(const EmitOrder &eo) {
    if (!eo.probe) return;
    (void)hipEventRecord(eo.probe->ev[1], ctx().stream);
    ++eo.probe->phase;
}

static bool emit_is_fused(int Wq) { return Wq <= 64 && [] { const char *e = getenv("SYMGPU_EMIT_FUSED"); return !(e && e[0] == '0'); }(); }
static int emit_touch(const u32 *markbits_p, i64 T, const LazyEmit &lz, EmitPrefix &pre) {
    const i64 n16 = (T + 63) / 64 / 2;                              // whole 16-byte chunks of a T-bit map
    pre.touched = true;
    if (n16 <= 0 || T < ((i64)1 << 23)) return SYMGPU_OK;           // (five maps of < 1 MiB each: still cached from the kernels that wrote them)
    TouchMaps tm;
    const void *maps[5] = {markbits_p, pre.wordprefix.p, lz.mode ? (const void *)lz.patchbits : nullptr, lz.mode == 1 ? (const void *)lz.e_lo : nullptr,
                           lz.mode == 1 ? (const void *)lz.e_hi : nullptr};
    for (int m = 0; m < 5; ++m) {
        tm.p[m] = reinterpret_cast<const u32x4 *>(maps[m]);
        tm.n16[m] = maps[m] ? (m == 1 ? (pre.wide ? n16 / 2 : 2 * n16) : n16) : 0;     // (the prefix: 4 bytes per 64 or per 32 indices)
    }
    hipLaunchKernelGGL(k_touch, dim3(grid_for(2 * n16)), dim3(256), 0, ctx().stream, tm, pre.total.as<u32>() + 2);
    KERNEL_CHECK();
    return SYMGPU_OK;
}
int cleanup_finish(u32 *markbits_p, const double *sum_of_p, i64 T, bool pair, const u64 *rows, int W, const u64 *inner, i64 Ni,
                   const u64 *outer, symgpu_op_t *out, int Wq_out, bool tri, const LazyEmit &lz, bool want_first, EmitPrefix *pre = nullptr) {
    hipStream_t st = ctx().stream;
    const i64 n_words = (T + 31) / 32;
    EmitPrefix own;
    if (!pre || pre->n_out < 0) {
        pre = &own;
        own.wide = emit_is_fused(W / 2);
        SG_TRY(emit_prefix(markbits_p, T, own.wordprefix, own.total, own.wide));
        u32 n_out32 = 0;
        SG_TRY(read_back_words(own.total.as<u32>(), 1, nullptr, 0, &n_out32));
        own.n_out = n_out32;
    }
    Scratch &wordprefix = pre->wordprefix, &total = pre->total;
    const i64 n_out = pre->n_out;
    symgpu_op_t res = nullptr;
    SG_TRY(symgpu_op_alloc(n_out > 0 ? n_out : 1, Wq_out, 1, &res));
    res->T = n_out;
    if (want_first) {
        const int rcf = dev_alloc((size_t)(n_out > 0 ? n_out : 1) * 8, (void **)&res->first);
        if (rcf != SYMGPU_OK) { symgpu_op_free(res); return rcf; }
    }
    if (n_out > 0) {
        const int Wq = W / 2;
        const int wsh = (Wq & (Wq - 1)) == 0 ? __builtin_ctz((unsigned)Wq) : -1;
        const u32x4 *pin = reinterpret_cast<const u32x4 *>(inner), *pout = reinterpret_cast<const u32x4 *>(outer);
        u32x4 *dst = reinterpret_cast<u32x4 *>(res->rows);
        // (rows of more than 64 chunks keep the batched stage: there every workgroup of the grid shares the chunks of a row, here a
        // wavefront streams its own rows alone — two 10^8-qubit terms: 2.9 ms against 8.5 ms)
        const bool fused = emit_is_fused(Wq);
        if (fused) {
            const i64 n_w64 = (T + 63) / 64;
            // wavefront shape: bitmap words per wavefront x 64-chunk steps in flight (SYMGPU_EMIT_SHAPE = "NW,U": experiments)
            static const int shape = [] { const char *e = SG_TUNE("SYMGPU_EMIT_SHAPE"); int nw = 2, u = 4; if (e) sscanf(e, "%d,%d", &nw, &u); return nw * 16 + u; }();
            const int NWs = shape / 16, Us = shape % 16;
            const int NWr = (NWs == 1 || NWs == 4 || NWs == 8) ? NWs : 2;   // 2 words per wavefront, 4 steps in flight: 1.20 ms at cfg3 (4,4: 1.30; 1,4: 1.24; 2,8: 1.22)
            const dim3 gfu((unsigned)(((n_w64 + 4 * NWr - 1) / (4 * NWr) + 7) / 8 * 8));     // a multiple of 8: the kernel's XCD-aware order is a bijection then
            static const bool touch_on = [] { const char *e = SG_TUNE("SYMGPU_EMIT_TOUCH"); return !(e && e[0] == '0'); }();
            if (touch_on && !pre->touched) SG_TRY(emit_touch(markbits_p, T, lz, *pre));
            EmitOrder eo;
            SG_TRY(emit_order_begin(dst, (size_t)n_out * (size_t)(16 * Wq + 16), eo));
            ProfScope prof(3);
#define LAUNCH_FUSED_S(P, TR, NWV, UV) hipLaunchKernelGGL((k_emit_fused<P, TR, NWV, UV>), gfu, dim3(256), 0, st, reinterpret_cast<const u64 *>(markbits_p), wordprefix.as<u32>(), T, sum_of_p, \
                                               (u32)(pair ? Ni : 1), Wq, wsh, reinterpret_cast<const u32x4 *>(rows), pin, pout, dst, res->coeff, lz, res->first, pre->wide ? 0 : 1, eo.interleaved)
#define LAUNCH_FUSED_U(P, TR, NWV) do { if (Us == 8) LAUNCH_FUSED_S(P, TR, NWV, 8); else if (Us == 2) LAUNCH_FUSED_S(P, TR, NWV, 2); else LAUNCH_FUSED_S(P, TR, NWV, 4); } while (0)
#define LAUNCH_FUSED(P, TR) do { if (NWr == 1) LAUNCH_FUSED_U(P, TR, 1); else if (NWr == 4) LAUNCH_FUSED_U(P, TR, 4); else if (NWr == 8) LAUNCH_FUSED_U(P, TR, 8); \
                                 else LAUNCH_FUSED_U(P, TR, 2); } while (0)
            if (pair && tri) LAUNCH_FUSED(true, true); else if (pair) LAUNCH_FUSED(true, false); else LAUNCH_FUSED(false, false);
            emit_order_end(eo);
#undef LAUNCH_FUSED_S
#undef LAUNCH_FUSED_U
#undef LAUNCH_FUSED
        } else {
        const int rc_env = [] { const char *e = SG_TUNE("SYMGPU_EMIT_RC"); return e ? atoi(e) : 2; }();     // chunks per lane: 1: 115, 2: 97, 4: 97 us per batch
        const int RCs = rc_env == 1 || rc_env == 4 ? rc_env : 2;
        const i64 EMIT_BATCH_WORDS = emit_batch_words();
        const i64 bw = n_words < EMIT_BATCH_WORDS ? n_words : EMIT_BATCH_WORDS;
        Scratch meta;
        int rc = meta.alloc((size_t)bw * 32 * sizeof(uint2));
        if (rc != SYMGPU_OK) { symgpu_op_free(res); return rc; }
        int wpw = 64;                                            // bitmap words per wavefront: aim at >= 16k wavefronts per batch
        while (wpw > 1 && (bw + wpw - 1) / wpw < 16384) wpw >>= 1;
        for (i64 w0 = 0; w0 < n_words; w0 += EMIT_BATCH_WORDS) {
            const i64 w1 = w0 + EMIT_BATCH_WORDS < n_words ? w0 + EMIT_BATCH_WORDS : n_words;
            i64 ge = ((w1 - w0 + wpw - 1) / wpw + 3) / 4;
            if (ge > 16384) ge = 16384;
#define LAUNCH_META(P, TR) hipLaunchKernelGGL((k_emit_meta<P, TR>), dim3((unsigned)ge), dim3(256), 0, st, markbits_p, wordprefix.as<u32>(), w0, w1, sum_of_p, wpw, \
                                              (u32)(pair ? Ni : 1), meta.as<uint2>(), res->coeff, lz, res->first)
            if (pair && tri) LAUNCH_META(true, true); else if (pair) LAUNCH_META(true, false); else LAUNCH_META(false, false);
#undef LAUNCH_META
            const u32 *p_begin = wordprefix.as<u32>() + w0;
            const u32 *p_end = w1 < n_words ? wordprefix.as<u32>() + w1 : total.as<u32>();
            const i64 gs = ((w1 - w0) * 32 * Wq + 256 * RCs - 1) / (256 * RCs);     // as if every index of the batch were kept
#define LAUNCH_STREAM(P, R) hipLaunchKernelGGL((k_emit_stream<P, R>), dim3((unsigned)gs), dim3(256), 0, st, meta.as<uint2>(), p_begin, p_end, Wq, wsh, \
                                               reinterpret_cast<const u32x4 *>(rows), pin, pout, dst)
            {
                ProfScope prof(3);
                if (pair) { if (RCs == 1) LAUNCH_STREAM(true, 1); else if (RCs == 4) LAUNCH_STREAM(true, 4); else LAUNCH_STREAM(true, 2); }
                else { if (RCs == 1) LAUNCH_STREAM(false, 1); else if (RCs == 4) LAUNCH_STREAM(false, 4); else LAUNCH_STREAM(false, 2); }
            }
#undef LAUNCH_STREAM
        }
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);   // the scratch buffers are freed on return; keep ordering simple
        if (e != hipSuccess) { symgpu_op_free(res); return hip_fail(e, "cleanup emit", __FILE__, __LINE__); }
    }
    res->dup_free = 1;                                          // merged: no two equal rows
    *out = res;
    return SYMGPU_OK;
}

// ---- products without repeated rows: the keys that COULD merge are few — find them after a partial sort, sort only them ---------------
// The radix sort exists to bring equal keys together, but a product of operators without repeated rows merges next to nothing (cfg3:
// the N diagonal pairs of P * P and nothing else), and the lazy flow has already decided every other term in index order
// (k_mark_singles).  So the sort stops early: after `run_bits` / 8 passes the keys are ordered by hash bits [lo, lo + run_bits), a RUN of
// equal bits holds ~Tk / 2^run_bits keys still in ascending index order (LSD passes are stable), and equal FULL keys can only sit in
// one run.  k_find_suspects flags every key that has an equal full key somewhere in its run — both partners — plus every key whose hash
// field is zero (the identity segment of a squared operator, N keys in one run).  The flagged keys are compacted IN ARRAY ORDER (so equal
// keys stay in index order), sorted completely — a few thousand keys instead of 5e7 — and handed to the unchanged segment machinery
// (fix-up, identity segment, k_heads_sums), which only ever acts on segments of more than one key.  Inputs full of repeated rows flag
// most keys: the caller then finishes the remaining passes on the whole array and carries on as before (the partial sort is the old
// sort's first passes, nothing is wasted but the flag pass).
// Round 4 stopped ONE pass early (runs of 3 keys, each key compared with its 12 predecessors in registers).  Round 5 stops when a run
// holds <= 1,024 keys on average (cfg3: two passes of four, runs of 763) and finds the partners inside a run through LDS: one scatter,
// one histogram and one scan pass over 5e7 keys less (0.27 ms), for a flag pass of the same cost.
//
// A workgroup owns the runs that START in its tile of 4,096 positions: it skips the head of the tile that continues the previous tile's
// run and reads on past the end of the tile until its last run ends (the neighbour skips exactly those keys).  The words w = key bits
// [32, 64) of the owned keys go through a pair of LDS bitmaps (2^17 bits each): `seen`, and `dup` for a bit that was already set.  Every
// key whose `dup` bit is set — the keys that have a partner, plus ~6 % chance hits — is listed, the listed words are compared all against
// all from LDS (two lanes per key, no memory access in the loop), and an equal word is followed up with the hash field and then the full
// 64-bit keys rebuilt from the operand hash tables, exactly the test the segment machinery makes.  A list that overflows flags every key
// the workgroup owns; a run longer than SUS_MAX_EXT extension steps raises `giveup` (the caller finishes the sort).  The kernel is bound
// by its LDS atomics and the chain of dependent phases of a workgroup, not by bandwidth (the loads alone: 0.09 ms of its 0.17).
constexpr int SUS_TILE = 4096;
constexpr int SUS_BLOOM = 17;                                          // log2 bits per bitmap: 16 KB each
constexpr int SUS_CAND = 960;                                          // (four workgroups' LDS per CU: 4 x 40.4 KB)
constexpr int SUS_MAX_EXT = 64;                                        // steps of 1,024 keys a workgroup reads past its tile
constexpr int SUS_WAVES = 8;                                           // 512 threads: eight keys of the tile and two of an extension step per lane
constexpr int SUS_ROWS = SUS_TILE / (64 * SUS_WAVES), SUS_XROWS = 1024 / (64 * SUS_WAVES);
constexpr int SUS_RUN_BITS = 16;                                       // the caller's partial sort: key bits [32, 48)
__device__ __forceinline__ u32 wave_shr1(u32 v, u32 lane0) {            // lane L <- v[L - 1], lane 0 <- lane0 (DPP wave_shr:1, one VALU instruction)
    return (u32)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138, 0xf, 0xf, false);
}
// w = the upper half of a key: run bits v below, sixteen more hash bits u above.  Inside a workgroup's range v takes a handful of consecutive
// values: the slot of a word is (u, v mod 2) (one v_alignbit + one v_and).  ~6 % of the keys are listed without having a partner (another
// run of the same parity holds their u).  A second slot (u / 8, v mod 8) in the same bitmaps brings that down to 1 %, but its two LDS
// atomics per key cost more than the longer comparison of the listed words: 0.20 against 0.167 ms.
__device__ __forceinline__ u32 sus_slot(u32 w) { return __builtin_amdgcn_alignbit(w, w, 16) & ((1u << SUS_BLOOM) - 1u); }
static_assert(SUS_BLOOM == 17 && SUS_RUN_BITS == 16, "sus_slot is written for these");
__global__ __launch_bounds__(64 * SUS_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_find_suspects(const u64 *__restrict__ keys, i64 T, PackedLayout L, const u64 *__restrict__ hI,
                                                                   const u64 *__restrict__ hO, u64 *__restrict__ suspect64, u32 *__restrict__ giveup) {
    constexpr int NT = 64 * SUS_WAVES;
    __shared__ u32 s_seen[1 << (SUS_BLOOM - 5)], s_dup[1 << (SUS_BLOOM - 5)];
    __shared__ __attribute__((aligned(16))) u32 s_cw[SUS_CAND];
    __shared__ u32 s_cpos[SUS_CAND];
    __shared__ u32 s_start, s_end[2], s_nc;                             // s_end by step parity: a wavefront is at most one barrier ahead
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 lt_mask = (1ULL << lane) - 1ULL;
    const i64 t0 = (i64)blockIdx.x * SUS_TILE;
    const u64 *kt = keys + t0;
    const u32 n_rel = T - t0 < (i64)0x0fffffff ? (u32)(T - t0) : 0x0fffffffu;          // positions from t0 on, as far as this workgroup could ever reach
    const u32 t1_rel = n_rel < (u32)SUS_TILE ? n_rel : (u32)SUS_TILE;
    const u32 rmask = (1u << SUS_RUN_BITS) - 1u;
    const int F = L.F();
    typedef unsigned long long ull;
    // clamped, unconditional; a 32-bit byte offset from the workgroup's (scalar) base: one address register per load instead of two
    auto at = [&](u32 rel) -> u64 { return *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(kt) + (size_t)((rel < n_rel ? rel : n_rel - 1) << 3)); };
    auto hash0 = [&](u64 k) -> bool { return (k >> F) == 0ULL; };                      // the identity segment
    // the tile — a wavefront holds 512 consecutive keys — and the first 1,024 keys behind it (a run of 763 keys on average ends there):
    // all loads in flight together
    constexpr int NK = SUS_ROWS + SUS_XROWS;
    u64 key[NK];
    const u32 wrel = (u32)wave * (64 * SUS_ROWS), xrel = (u32)SUS_TILE + (u32)wave * (64 * SUS_XROWS);
    auto rel_of = [&](int r) -> u32 { return (r < SUS_ROWS ? wrel + r * 64 : xrel + (r - SUS_ROWS) * 64) + lane; };
#pragma unroll
    for (int r = 0; r < NK; ++r) key[r] = at(rel_of(r));
    // only the upper half of a key (run bits + sixteen hash bits) and one bit "hash field zero" are kept from here on: ten registers less
    u32 w[NK], zmask = 0;
    const u64 kprev = wrel > 0 ? at(wrel - 1) : (t0 > 0 ? keys[t0 - 1] : 0ULL);      // (clamped: a wavefront's segment may lie behind the end)
    const u64 kpe = at(xrel - 1);
    {
        const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < (1 << (SUS_BLOOM - 5)) / (4 * NT); ++k) {
            reinterpret_cast<u32x4 *>(s_seen)[k * NT + threadIdx.x] = z;
            reinterpret_cast<u32x4 *>(s_dup)[k * NT + threadIdx.x] = z;
        }
    }
    if (threadIdx.x == 0) { s_start = ~0u; s_end[0] = ~0u; s_end[1] = ~0u; s_nc = 0u; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < NK; ++r) { w[r] = (u32)(key[r] >> 32); zmask |= (hash0(key[r]) ? 1u : 0u) << r; }
    // first position of the tile where a run starts, and first one behind the tile
    {
        u32 carry = (u32)(kprev >> 32) & rmask;
        u32 first_b = ~0u, first_e = ~0u;
#pragma unroll
        for (int r = 0; r < NK; ++r) {
            if (r == SUS_ROWS) carry = (u32)(kpe >> 32) & rmask;
            const u32 v = w[r] & rmask;
            const u32 pv = wave_shr1(v, carry);
            carry = (u32)__builtin_amdgcn_readlane((int)v, 63);
            const u32 rel = rel_of(r);
            if (r < SUS_ROWS) {
                const u64 bm = __ballot(rel < t1_rel && (v != pv || (t0 == 0 && rel == 0)));
                if (bm && first_b == ~0u) first_b = rel - lane + (u32)__builtin_ctzll(bm);
            } else {
                const u64 bm = __ballot(rel < n_rel && v != pv);
                if (bm && first_e == ~0u) first_e = rel - lane + (u32)__builtin_ctzll(bm);
            }
        }
        if (lane == 0 && first_b != ~0u) atomicMin(&s_start, first_b);
        if (lane == 0 && first_e != ~0u) atomicMin(&s_end[0], first_e);
    }
    __syncthreads();
    const u32 s_rel = s_start;
    if (s_rel == ~0u) return;                                           // the whole tile continues a run that started earlier (block-uniform)
    // the end of the last run that starts in the tile: in the first 1,024 keys behind it, or (rare) further on, in steps of 1,024 keys
    u32 e_rel = t1_rel;
    bool closed = n_rel <= (u32)SUS_TILE;
    if (!closed) {
        const u32 end_now = s_end[0];
        const u32 step_end = (u32)SUS_TILE + 1024u;
        if (end_now != ~0u) { e_rel = end_now; closed = true; }
        else if (step_end >= n_rel) { e_rel = n_rel; closed = true; }
        else e_rel = step_end;
    }
    auto flag_chunk = [&](u64 m, i64 chunk) {
        if (lane == 0 && m) atomicOr(reinterpret_cast<ull *>(suspect64 + chunk), (ull)m);
    };
    // `seen` bits of a lane's keys back to back (a lane that owns nothing ORs a zero), then the `dup` bits
    {
        u32 ins = 0;                                                    // bit r = key r of this lane goes into the bitmaps
#pragma unroll
        for (int r = 0; r < NK; ++r) {
            const u32 rel = rel_of(r);
            const bool own = rel >= s_rel && rel < e_rel;
            const bool zh = own && ((zmask >> r) & 1u);
            flag_chunk(__ballot(zh), (t0 + rel - lane) / 64);
            ins |= ((own && !zh) ? 1u : 0u) << r;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                   // five keys at a time: ten atomics with return in flight
            constexpr int HB = NK / 2;
            u32 old[HB];
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                const int r = h * HB + q;
                const u32 slot = sus_slot(w[r]), in = (ins >> r) & 1u;
                old[q] = atomicOr(&s_seen[slot >> 5], in << (slot & 31));
            }
#pragma unroll
            for (int q = 0; q < HB; ++q) {
                const int r = h * HB + q;
                const u32 slot = sus_slot(w[r]), in = (ins >> r) & 1u;
                atomicOr(&s_dup[slot >> 5], old[q] & (in << (slot & 31)));
            }
        }
    }
    for (int step = 1; !closed; ++step) {
        if (step == SUS_MAX_EXT) {                                      // block-uniform
            if (threadIdx.x == 0) atomicOr(giveup, 1u);
            return;
        }
        const u32 q0 = xrel + (u32)step * 1024u;
        u64 kq[SUS_XROWS];
#pragma unroll
        for (int r = 0; r < SUS_XROWS; ++r) kq[r] = at(q0 + r * 64 + lane);
        const u64 kp = at(q0 - 1);
        u32 carry = (u32)(kp >> 32) & rmask, first_e = ~0u;
#pragma unroll
        for (int r = 0; r < SUS_XROWS; ++r) {
            const u32 rel = q0 + r * 64 + lane;
            const u32 v = (u32)(kq[r] >> 32) & rmask;
            const u32 pv = wave_shr1(v, carry);
            carry = (u32)__builtin_amdgcn_readlane((int)v, 63);
            const u64 bm = __ballot(rel < n_rel && v != pv);
            if (bm && first_e == ~0u) first_e = q0 + r * 64 + (u32)__builtin_ctzll(bm);
        }
        if (lane == 0 && first_e != ~0u) atomicMin(&s_end[step & 1], first_e);
        __syncthreads();
        const u32 end_now = s_end[step & 1];
        const u32 step_end = (u32)SUS_TILE + (u32)(step + 1) * 1024u;
        if (end_now != ~0u) { e_rel = end_now; closed = true; }
        else if (step_end >= n_rel) { e_rel = n_rel; closed = true; }
        else e_rel = step_end;
#pragma unroll
        for (int r = 0; r < SUS_XROWS; ++r) {
            const u32 rel = q0 + r * 64 + lane;
            const bool own = rel < e_rel;
            const bool zh = own && hash0(kq[r]);
            flag_chunk(__ballot(zh), (t0 + q0) / 64 + r);
            const u32 wq = (u32)(kq[r] >> 32), slot = sus_slot(wq);
            const u32 bit = (own && !zh) ? 1u << (slot & 31) : 0u;
            const u32 old = atomicOr(&s_seen[slot >> 5], bit);
            atomicOr(&s_dup[slot >> 5], old & bit);
        }
    }
    __syncthreads();
    // the owned keys whose bit was set twice: one counter update per wavefront
    auto listed = [&](u32 rel, u32 wk, bool zh) -> bool {
        const u32 slot = sus_slot(wk);
        return rel >= s_rel && rel < e_rel && ((s_dup[slot >> 5] >> (slot & 31)) & 1u) && !zh;
    };
    {
        u64 m[NK];
        u32 total = 0;
#pragma unroll
        for (int r = 0; r < NK; ++r) {
            m[r] = __ballot(listed(rel_of(r), w[r], (zmask >> r) & 1u));
            total += (u32)__popcll(m[r]);
        }
        if (total) {                                                    // wave-uniform
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&s_nc, total);
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
#pragma unroll
            for (int r = 0; r < NK; ++r) {
                const u32 n = base + (u32)__popcll(m[r] & lt_mask);
                if (((m[r] >> lane) & 1ULL) && n < (u32)SUS_CAND) { s_cpos[n] = rel_of(r); s_cw[n] = w[r]; }
                base += (u32)__popcll(m[r]);
            }
        }
    }
    for (u32 rel = (u32)SUS_TILE + 1024u + threadIdx.x; rel < e_rel; rel += NT) {      // (the steps that were not kept in registers)
        const u64 k = kt[rel];
        if (listed(rel, (u32)(k >> 32), hash0(k))) {
            const u32 n = atomicAdd(&s_nc, 1u);
            if (n < (u32)SUS_CAND) { s_cpos[n] = rel; s_cw[n] = (u32)(k >> 32); }
        }
    }
    __syncthreads();
    const u32 nc = s_nc;
    if (nc > (u32)SUS_CAND) {                                           // (repeated rows all over: the caller gives the partial sort up anyway)
        // whole flag words (a bit at a time this path took 1.5 ms on the hash-partitioned products, where every key has its twin)
        const i64 g0 = t0 + s_rel, g1 = t0 + e_rel - 1;                // first and last owned position
        for (i64 wd = (g0 >> 6) + threadIdx.x; wd <= (g1 >> 6); wd += NT) {
            u64 m = ~0ULL;
            if (wd == (g0 >> 6)) m &= ~0ULL << (g0 & 63);
            if (wd == (g1 >> 6)) m &= ~0ULL >> (63 - (g1 & 63));
            atomicOr(reinterpret_cast<ull *>(suspect64 + wd), (ull)m);
        }
        return;
    }
    // listed keys all against all by their words (a broadcast 16-byte read serves four) — no memory access in the loop: a lane that
    // followed an equal word up with loads made the whole wavefront wait for every such lane in turn (a run of 763 keys holds four pairs
    // with equal 32-bit words), and that was two thirds of the kernel.  One partner: the rest of the hash field, then the 64-bit keys
    // rebuilt from the operand hash tables decide, all lanes at once (a run holds 0.02 triples of equal words on average: 3,000 keys of
    // 5e7, and flagged without looking they cost the fix-up pass of the flagged keys 0.15 ms); more than two: flagged without looking.
    // (two lanes per listed key, each on half of the list: the tail of a workgroup is a handful of lanes at work)
    const u32 nc4 = (nc + 3u) & ~3u, half = ((nc4 / 4 + 1) / 2) * 4;
    for (u32 a0 = 0; a0 < nc; a0 += NT / 2) {
        const u32 a = a0 + threadIdx.x / 2, part = threadIdx.x & 1u;
        const bool live = a < nc;
        const u32 wa = live ? s_cw[a] : 0u;
        u32 n_eq = 0, first = 0, last = 0;
        const u32 b0 = part ? half : 0u, b1 = part ? nc : (half < nc ? half : nc);
#pragma unroll 4
        for (u32 b = b0; b < b1; b += 4) {
            const u32x4 wl = *reinterpret_cast<const u32x4 *>(&s_cw[b]);
#pragma unroll
            for (u32 j = 0; j < 4; ++j)
                if (live && wl[j] == wa && b + j != a && b + j < b1) {
                    if (n_eq == 0) first = b + j;
                    last = b + j;
                    ++n_eq;
                }
        }
        {   // both halves together: with up to two partners (first, last) are all of them
            const u32 n_o = (u32)__shfl_xor((int)n_eq, 1), f_o = (u32)__shfl_xor((int)first, 1), l_o = (u32)__shfl_xor((int)last, 1);
            const u32 n_lo = part ? n_o : n_eq, f_lo = part ? f_o : first, l_lo = part ? l_o : last;
            const u32 n_hi = part ? n_eq : n_o, f_hi = part ? first : f_o, l_hi = part ? last : l_o;
            first = n_lo ? f_lo : f_hi;
            last = n_hi ? l_hi : l_lo;
            n_eq = n_lo + n_hi;
        }
        bool hit = n_eq > 2;                                            // (four equal 32-bit words in one run: flagged without looking)
        if (n_eq >= 1 && n_eq <= 2 && part == 0) {
            const u64 ka = kt[s_cpos[a]], kb = kt[s_cpos[first]], kc = kt[s_cpos[last]];      // (from the L2: the lists hold words only)
            const u64 fa = L.full_key(hI, hO, ka);
            hit = (((ka ^ kb) >> F) == 0ULL && fa == L.full_key(hI, hO, kb)) || (((ka ^ kc) >> F) == 0ULL && fa == L.full_key(hI, hO, kc));
        }
        if (hit && part == 0) {
            const i64 p = t0 + s_cpos[a];
            atomicOr(reinterpret_cast<ull *>(suspect64 + (p >> 6)), 1ULL << (p & 63));
        }
    }
}
__global__ void k_popc_words64(const u64 *__restrict__ bits, i64 n_words, u32 *__restrict__ counts) {
    for (i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x; w < n_words; w += (i64)gridDim.x * blockDim.x) counts[w] = (u32)__popcll(bits[w]);
}
// flagged keys, in array order, to the front of `out`
__global__ __launch_bounds__(256) void k_compact_suspects(const u64 *__restrict__ keys, const u64 *__restrict__ suspect64, const u32 *__restrict__ prefix,
                                                           i64 n_chunks, u64 *__restrict__ out) {
    // sixteen flag words per wavefront (lanes 0..15 load them), then the few words that have a flag set one after the other with the whole
    // wavefront (a product without repeated rows flags a few thousand of 5e7 keys: a wavefront per word spent 41 us reading zeros; 64 words
    // per wavefront left the 10^4 consecutive flagged keys of a squared operator's identity segment to three wavefronts: 31 us)
    const int lane = threadIdx.x & 63;
    for (i64 base = ((i64)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16; base < n_chunks; base += (i64)gridDim.x * 64) {
        const i64 mine = base + (lane & 15);
        const u64 b = mine < n_chunks ? suspect64[mine] : 0ULL;
        u64 nz = __ballot(b != 0ULL) & 0xFFFFULL;
        while (nz) {                                                             // wave-uniform
            const int l = __builtin_ctzll(nz);
            nz &= nz - 1;
            const u64 bb = __shfl(b, l);
            const i64 c = base + l;
            if ((bb >> lane) & 1ULL) out[(i64)prefix[c] + __popcll(bb & ((1ULL << lane) - 1ULL))] = keys ? keys[c * 64 + lane] : (u64)(c * 64 + lane);   // (no keys: the index)
        }
    }
}

// plain mode: rows/coeff of T terms.  pair mode (inner != null): T = Ni*No, term t = o*Ni + i is inner[i] ^ outer[o] with
// coefficient ci[i] * co[o] * i^e (product.hip); coeff is unused.  *out is a fresh operator with the cleaned result.
int cleanup_core(const u64 *rows, const double *coeff, i64 T, int W, const u64 *inner, i64 Ni, const u64 *outer, i64 No,
                 double thr, int use_thr, symgpu_op_t *out, int Wq_out, const double *ci, const double *co, int inner_is_left, bool want_first) {
    hipStream_t st = ctx().stream;
    const bool pair = inner != nullptr;
    const bool same_rows = pair && outer == inner && No == Ni;
    const u64 *hO_p = nullptr;
    if (T >= ((i64)1 << 32) - 1) {
        set_error("cleanup: %lld terms exceed the 2^32-2 limit of the 32-bit index sort", (long long)T);
        return SYMGPU_E_INVALID;
    }
    symgpu_op_t res = nullptr;
    if (T == 0) {
        SG_TRY(symgpu_op_alloc(1, Wq_out, 1, &res));
        res->T = 0;
        res->dup_free = 1;
        *out = res;
        return SYMGPU_OK;
    }
    // pair mode sorts PACKED keys (hash | e | o | i) unless the index fields need more than 32 bits or a long mixed prefix
    // run forced the 64-bit fallback (then the pair coefficients are materialised and sorted by separate index values)
    PackedLayout L;
    L.bi = 0; L.bo = 0;
    while (pair && ((i64)1 << L.bi) < Ni) ++L.bi;
    while (pair && ((i64)1 << L.bo) < No) ++L.bo;
    bool packed = pair && L.bi + L.bo <= 32 && L.bi < 32;
    {   // SYMGPU_CLEANUP_UNPACKED=1 forces the fallback path (tests)
        const char *e = getenv("SYMGPU_CLEANUP_UNPACKED");
        if (e && e[0] == '1') packed = false;
    }
    // P * P with one device operand for both factors: keys only for the pairs with i >= o (product.hip, KM = 2), weighted 1 / 2 / 0
    // — valid when exact zeros are dropped anyway (strict |c| > thr with thr >= 0): without a threshold the reference keeps the
    // rows of anticommuting pairs with coefficient 0, and the sum x + y - x of a row shared with other pairs need not equal y
    // (and thr > 0: the twin-first sums differ from the reference's sequential ones by rounding for coefficients that are not dyadic, and
    // against thr = 0 a residue of 1e-33 instead of an exact 0 keeps or drops a ROW — tools/stress_cleanup.py, tiny planted coefficients)
    bool squared = packed && inner == outer && Ni == No && ci == co && use_thr && thr > 0.0;
    {
        const char *e = getenv("SYMGPU_CLEANUP_NOSQUARE");             // tests: the general pair path on a squared operator
        if (e && e[0] == '1') squared = false;
    }
    i64 Tk = T;                                                         // number of keys that are sorted (T index space stays)
    Scratch keys, keys2, idx, idx2, fixlist, collision, hI, hO, pair_coeff, markbits, sum_of, zpart, zcount, patchbits, e_lo, e_hi, dirtybits, sort_hist, cfloor;
    // singles decided in index order, only merged terms filed from the sorted order (k_mark_singles); SYMGPU_CLEANUP_LAZY=0: every term filed
    // Default: products only.  A plain cleanup is what follows `A + B` or a rotation — inputs full of repeated rows, where every chunk
    // holds merged terms and the extra passes buy nothing (10^6 terms of 1,000 qubits, 2.7 copies of every row: 0.89 ms lazy against
    // 0.55 ms filed); SYMGPU_CLEANUP_LAZY=1 forces the lazy flow for plain cleanups too (tests), 0 switches it off everywhere.
    const int lazy_env = [] { const char *e = getenv("SYMGPU_CLEANUP_LAZY"); return e ? (e[0] == '0' ? 0 : 1) : -1; }();
    const bool lazy = lazy_env == 1 || (lazy_env == -1 && pair);
    const size_t bitmap_bytes = (size_t)((T + 63) / 64) * 8;         // whole 64-bit words: k_mark_singles stores one per wavefront
    SG_TRY(keys.alloc((size_t)T * 8));
    SG_TRY(keys2.alloc((size_t)T * 8));
    SG_TRY(markbits.alloc(bitmap_bytes));                            // kept terms by first input index (one bit each)
    if (lazy) {
        SG_TRY(patchbits.alloc(bitmap_bytes));                       // ... whose coefficient is a filed sum
        if (pair) { SG_TRY(e_lo.alloc(bitmap_bytes)); SG_TRY(e_hi.alloc(bitmap_bytes)); }   // phase exponents of the pairs, by index
    }
    SG_TRY(sum_of.alloc((size_t)T * 16));                           // their summed coefficients, indexed the same way
    SG_TRY(collision.alloc(16));
    if (pair) {
        SG_TRY(hI.alloc((size_t)Ni * 8));
        if (!same_rows) SG_TRY(hO.alloc((size_t)No * 8));
        hO_p = same_rows ? hI.as<u64>() : hO.as<u64>();
    } else {
        SG_TRY(idx.alloc((size_t)T * 4));
        SG_TRY(idx2.alloc((size_t)T * 4));
    }
    u64 *ks = nullptr;
    u32 *is = nullptr;
    u64 seed = ctx().hash_tab ? ctx().hash_seed : 1;
    bool ok = false;
    if (squared) Tk = Ni * (Ni + 1) / 2;                                // the pairs with i >= o
    // number of (top) key bits the radix sort orders; the rest is handled by k_fixup_mark / k_fixup_sort
    int nb = 64;
    {
        int lg = 0;
        while (((i64)1 << lg) < Tk) ++lg;
        const int want = (lg + 5 + 7) / 8 * 8;   // ~1-3 % of the keys then share a prefix with another key: cheap local fix-up
        if (want < 64) nb = want;
    }
    // P * P: the identity coefficient in the reference's own (sequential) order, see k_diag_seq_sum.  Large operators: on the side stream,
    // joined just before k_zero_close reads it; small ones inline (two event operations cost more than the kernel).
    Scratch diag_seq;
    bool diag_side = false;
    struct JoinSide {                                                  // whatever way the function is left: the main stream is ordered behind the side
        bool &pending;                                                 // stream's kernel before its 16-byte result buffer goes back to the allocator
        ~JoinSide() { if (pending) (void)hipStreamWaitEvent(ctx().stream, ctx().ev_join, 0); }
    } join_side{diag_side};
    if (squared) {
        Context &c = ctx();
        SG_TRY(diag_seq.alloc(16));
        diag_side = Ni > 2048;
        if (diag_side) {
            HIP_TRY(hipEventRecord(c.ev_fork, st));
            HIP_TRY(hipStreamWaitEvent(c.stream2, c.ev_fork, 0));
        }
        hipLaunchKernelGGL(k_diag_seq_sum, dim3(1), dim3(64), 0, diag_side ? c.stream2 : st, ci, (u32)Ni, diag_seq.as<double>());
        KERNEL_CHECK();
        if (diag_side) HIP_TRY(hipEventRecord(c.ev_join, c.stream2));
    }
    bool lazy_final = false;
    EmitPrefix pre;
    for (int attempt = 0; attempt < 6 && !ok; ++attempt) {
        // the lazy flow pays off on big key sets (its passes are fixed costs, the scatter it avoids only hurts at scale): from 2^18 keys for
        // general products (2.5e5 keys: 0.178 -> 0.164 ms, 1.5e6: 0.32 -> 0.25, 4e6: 0.60 -> 0.42), from 2^20 for squared operators (5e5 keys:
        // 0.222 against 0.231 lazy; 1.1e6: 0.31 -> 0.30, 3.1e6: 0.42 -> 0.37) — round 4's gate was 2^22 for both
        const bool lazy_a = lazy && (lazy_env == 1 || Tk >= ((i64)1 << (squared ? 20 : 18)));
        lazy_final = lazy_a;
        SG_TRY(ensure_hash_tables(seed));
        const int hash_bits = packed ? 64 - L.F() : 64;       // a packed key carries 64 - F >= 30 hash bits
        const int nbits = nb > hash_bits ? hash_bits : nb;
        bool in_tmp = false;
        // what the stages after the sort see: `Tsort` keys ordered by their top `fix_bits` bits — all Tk keys, or (sus_active) only the keys
        // that k_find_suspects flagged
        i64 Tsort = Tk;
        int fix_bits = nbits;
        u64 *ks_sorted = nullptr;
        bool sus_active = false, sus_coop = false;
        if (pair) {
            SG_TRY(hash_rows(inner, Ni, W, hI.as<u64>()));
            if (!same_rows) SG_TRY(hash_rows(outer, No, W, hO.as<u64>()));     // (one operand used twice: hO_p is hI)
            if (packed) {
                PairKeyArgs ka;
                ka.hI = hI.as<u64>(); ka.hO = hO_p; ka.keys = keys.as<u64>(); ka.bi = L.bi; ka.bo = L.bo; ka.o_base = 0;
                ka.squared = squared ? 1 : 0;
                // products whose keys mostly merge with nothing: stop the sort one pass early and sort only the keys that have a partner
                // (k_find_suspects).  Applies when a run of the partial order is short (<= 8 keys on average) and the operands are not
                // one array used twice without the squared-operator compaction (then EVERY key has its twin).
                // SYMGPU_CLEANUP_SUSPECTS: 0 = complete sort of all keys; 2 = tests: behave as if most keys were flagged (the remaining passes are finished on the whole array)
                const bool sus_env = [] { const char *e = getenv("SYMGPU_CLEANUP_SUSPECTS"); return !(e && e[0] == '0'); }();
                const bool sus_giveup = [] { const char *e = getenv("SYMGPU_CLEANUP_SUSPECTS"); return e && e[0] == '2'; }();
                // two passes before the flag pass, on key bits [32, 48): a run then holds <= 2,048 keys on average (products of up to 2^27 keys)
                const int sus_pass = SUS_RUN_BITS / 8;
                const bool sus_try = lazy_a && sus_env && nbits == 32 && (Tk >> SUS_RUN_BITS) <= 2048 && !(inner == outer && !squared);
                // round 6: where the flag pass works from the operand hash tables (pair_dups.hip) nobody reads the keys but the marking of
                // the single terms, and all it reads of them is the phase exponent and "is the diagonal": the key kernel then writes ONE BYTE
                // per pair (into the key buffer) and k_mark_bytes marks from those; the few flagged keys are rebuilt at their compaction.
                // Should the flag pass give up, the keys are generated after all.  SYMGPU_CLEANUP_KEYBYTES=0: 8-byte keys + k_mark_singles.
                const bool key_bytes = sus_try && pair_dups_fits(Ni, No, squared, Tk, nullptr) && !wide_pairs_worthwhile(Ni, No, W / 2) &&
                                       !(getenv("SYMGPU_CLEANUP_KEYBYTES") && getenv("SYMGPU_CLEANUP_KEYBYTES")[0] == '0');
                if (key_bytes) ka.ebytes = keys.as<unsigned char>();
                SG_TRY(mul_keys_dev(inner, Ni, outer, No, W / 2, inner_is_left, ka));
                ka.ebytes = nullptr;
                u32 *first_hist = nullptr;
                if (lazy_a) {                                          // the keys are still in index order
                    const i64 n_tiles = (Tk + SORT_TILE - 1) / SORT_TILE;
                    // how small the operands' coefficients get: decides whether the marking has to look at them at all
                    SG_TRY(cfloor.alloc(16));
                    const bool one_block = Ni <= 65536 && No <= 65536;       // a single workgroup stores its minimum: nothing to initialise
                    if (!one_block) HIP_TRY(hipMemsetAsync(cfloor.p, 0xFF, 16, st));
                    const bool two_ops = co != ci || No != Ni;
                    if (one_block)                                         // (blockIdx.y selects the operand: one launch for both)
                        hipLaunchKernelGGL(k_coeff_floor, dim3(1, two_ops ? 2 : 1), dim3(1024), 0, st, ci, Ni, co, No, cfloor.as<unsigned long long>(), 1);
                    else {
                        hipLaunchKernelGGL(k_coeff_floor, dim3(grid_for(Ni, 256, 256)), dim3(256), 0, st, ci, Ni, ci, Ni, cfloor.as<unsigned long long>(), 0);
                        if (two_ops)
                            hipLaunchKernelGGL(k_coeff_floor, dim3(grid_for(No, 256, 256)), dim3(256), 0, st, co, No, co, No, cfloor.as<unsigned long long>() + 1, 0);
                    }
                    const double *fl_i = cfloor.as<double>(), *fl_o = (co != ci || No != Ni) ? cfloor.as<double>() + 1 : cfloor.as<double>();
                    if (getenv("SYMGPU_CLEANUP_NOFLOOR")) fl_i = fl_o = nullptr;     // tests: every coefficient looked at
                    if (key_bytes) {
                        const i64 n_groups = (Tk + 63) / 64 * 4;           // whole 64-bit words of the bitmaps
                        hipLaunchKernelGGL(k_mark_bytes, dim3((unsigned)grid_for(n_groups, 256, 1 << 16)), dim3(256), 0, st, keys.as<u32x4>(), Tk, n_groups, Ni, ci, co,
                                           squared ? 1 : 0, thr, use_thr, markbits.as<unsigned short>(), e_lo.as<unsigned short>(), e_hi.as<unsigned short>(), fl_i, fl_o);
                    } else {
                        SG_TRY(sort_hist.alloc((size_t)n_tiles * 256 * sizeof(u32)));
                        first_hist = sort_hist.as<u32>();
                        hipLaunchKernelGGL(k_mark_singles<true>, dim3((unsigned)n_tiles), dim3(256), 0, st, keys.as<u64>(), (const double *)nullptr, Tk, L, ci, co,
                                           squared ? 1 : 0, thr, use_thr, markbits.as<u64>(), e_lo.as<u64>(), e_hi.as<u64>(), first_hist, 64 - nbits, n_tiles, fl_i, fl_o);
                    }
                    KERNEL_CHECK();
                }
                if (!sus_try) {
                    // (small products — up to 2e5 keys, no first-pass histograms at hand —: the sort in ONE launch; its passes were three launches each)
                    bool coop_done = false;
                    if (!first_hist) SG_TRY(radix_sort_keys_u64_small(keys.as<u64>(), keys2.as<u64>(), Tk, 64 - nbits, 64, &in_tmp, &coop_done));
                    if (coop_done) sus_coop = true;
                    else SG_TRY(radix_sort_keys_u64(keys.as<u64>(), keys2.as<u64>(), Tk, 64 - nbits, 64, &in_tmp, first_hist));
                } else {
                    const int lo = 64 - nbits, hi = lo + 8 * sus_pass;
                    const i64 n_sc = (Tk + 63) / 64;
                    Scratch susbits, susprefix;
                    SG_TRY(susbits.alloc((size_t)n_sc * 8 + 16));
                    SG_TRY(susprefix.alloc((size_t)n_sc * 4));
                    u32 *sustotal = susbits.as<u32>() + 2 * n_sc;          // [0] flagged keys, [1] the flag pass gave up (one memset with the flags)
                    SG_TRY(zero_two(susbits.p, (size_t)n_sc * 8 + 16, nullptr, 0));
                    // round 6: operands whose bucketed hash words fit a workgroup's LDS — the flags come from the operand hash tables, the
                    // keys are never sorted (pair_dups.hip); they stay in index order and the flags are indexed likewise.
                    // (Measured and dropped: the marking pass folded into the key kernel — three bitmaps ORed from its epilogue, an atomic per
                    // 64 indices, or per 256 with the inner words stored: the conditional memory operations of the epilogue drain the kernel's
                    // memory counter one by one, 0.28 -> 0.48 / 0.58 ms for the 0.14 ms pass it would replace; and the marking pass without the
                    // first-pass histograms, which only the fall-back needs: 0.139 -> 0.141 ms, it is bound by reading the keys.)
                    bool direct = false;
                    SG_TRY(pair_dups_dev(hI.as<u64>(), Ni, hO_p, No, squared, Tk, susbits.as<u64>(), sustotal + 1, &direct));
                    u64 *part = keys.as<u64>(), *spare = keys2.as<u64>();
                    if (!direct) {
                        if (key_bytes) SG_TRY(mul_keys_dev(inner, Ni, outer, No, W / 2, inner_is_left, ka));     // (the flag pass was refused: keys after all)
                        SG_TRY(radix_sort_keys_u64(keys.as<u64>(), keys2.as<u64>(), Tk, lo, hi, &in_tmp, first_hist));
                        part = in_tmp ? keys2.as<u64>() : keys.as<u64>(); spare = in_tmp ? keys.as<u64>() : keys2.as<u64>();
                        hipLaunchKernelGGL(k_find_suspects, dim3((unsigned)((Tk + SUS_TILE - 1) / SUS_TILE)), dim3(64 * SUS_WAVES), 0, st, part, Tk, L, hI.as<u64>(),
                                           hO_p, susbits.as<u64>(), sustotal + 1);
                    }
                    if (n_sc <= POPC_SCAN_SMALL_MAX) SG_TRY(popc_scan_small(susbits.as<u64>(), n_sc, -1, susprefix.as<u32>(), sustotal));
                    else {
                        hipLaunchKernelGGL(k_popc_words64, dim3(grid_for(n_sc)), dim3(256), 0, st, susbits.as<u64>(), n_sc, susprefix.as<u32>());
                        KERNEL_CHECK();
                        SG_TRY(exclusive_scan_u32(susprefix.as<u32>(), susprefix.as<u32>(), n_sc, sustotal));
                    }
                    // the compaction does not need the count: it is queued behind the count's way home and runs while the host waits for it (in
                    // the rare give-up case its output is simply overwritten)
                    u32 h_sus2[2] = {0, 0};
                    {
                        ReadBack rb;
                        SG_TRY(read_back_post(sustotal, 2, nullptr, 0, &rb));
                        const bool from_bytes = key_bytes && direct;       // only bytes in the key buffer: the indices are compacted, the keys rebuilt from them
                        hipLaunchKernelGGL(k_compact_suspects, dim3((unsigned)grid_for((n_sc + 63) / 64, 1, 1 << 16)), dim3(256), 0, st, from_bytes ? (const u64 *)nullptr : part,
                                           susbits.as<u64>(), susprefix.as<u32>(), n_sc, spare);
                        if (from_bytes)
                            hipLaunchKernelGGL(k_keys_of_indices, dim3(64), dim3(256), 0, st, spare, sustotal, keys.as<unsigned char>(), Ni, squared ? 1 : 0, L, hI.as<u64>(), hO_p);
                        KERNEL_CHECK();
                        SG_TRY(read_back_wait(&rb, h_sus2));
                    }
                    const u32 h_sus = h_sus2[0];
                    if ((i64)h_sus * 16 > Tk || sus_giveup || h_sus2[1]) {
                        // repeated rows all over: the last pass on the whole array after all (LSD: the order so far is its first passes; the
                        // direct flag pass left the keys in index order: all passes)
                        bool in_tmp2 = false;
                        if (key_bytes && direct) SG_TRY(mul_keys_dev(inner, Ni, outer, No, W / 2, inner_is_left, ka));   // (only bytes so far: the keys after all)
                        SG_TRY(radix_sort_keys_u64(part, spare, Tk, direct ? lo : hi, 64, &in_tmp2, direct ? first_hist : nullptr));
                        if (in_tmp2) in_tmp = !in_tmp;
                    } else {
                        sus_active = true;
                        Tsort = h_sus;
                        if (Tsort > 0) {
                            // the flagged keys, sorted completely (the same rule for the number of sorted bits, now for a few thousand keys);
                            // `part` is not needed any more and serves as the sort's second buffer
                            int lgs = 0;
                            while (((i64)1 << lgs) < Tsort) ++lgs;
                            const int want_s = (lgs + 5 + 7) / 8 * 8;
                            fix_bits = want_s < hash_bits ? want_s : hash_bits;
                            // the compaction kept the array order, i.e. the flagged keys are ordered by key bits [lo, hi) already: when the bits to
                            // order start inside that range only the passes above it are left (LSD: stable passes on more significant bits)
                            const int sort_from = (!direct && 64 - fix_bits >= lo && 64 - fix_bits < hi) ? hi : 64 - fix_bits;
                            bool in_tmp_s = false, coop_done = false;
                            SG_TRY(radix_sort_keys_u64_coop(spare, part, Tsort, sort_from, 64, &in_tmp_s, &coop_done));
                            sus_coop = coop_done;
                            if (!coop_done) SG_TRY(radix_sort_keys_u64(spare, part, Tsort, sort_from, 64, &in_tmp_s));
                            ks_sorted = in_tmp_s ? part : spare;
                        }
                    }
                }
            } else {
                if (!idx.p) {
                    SG_TRY(idx.alloc((size_t)T * 4));
                    SG_TRY(idx2.alloc((size_t)T * 4));
                }
                if (!pair_coeff.p) {
                    SG_TRY(pair_coeff.alloc((size_t)T * 16));
                    SG_TRY(mul_coeff_dev(inner, ci, Ni, outer, co, 0, No, W / 2, inner_is_left, pair_coeff.as<double>()));
                    coeff = pair_coeff.as<double>();
                }
                hipLaunchKernelGGL(k_pair_keys, dim3(grid_for(T)), dim3(256), 0, st, hI.as<u64>(), Ni, hO_p, T, keys.as<u64>(), idx.as<u32>());
                KERNEL_CHECK();
            }
        } else {
            SG_TRY(hash_rows_any(rows, T, W, seed, keys.as<u64>(), idx.as<u32>()));      // (and idx[t] = t, the array the sort carries)
        }
        if (!packed && lazy_a) {
            hipLaunchKernelGGL(k_mark_singles<false>, dim3((unsigned)((T + SORT_TILE - 1) / SORT_TILE)), dim3(256), 0, st, (const u64 *)nullptr, coeff, T, L, (const double *)nullptr,
                               (const double *)nullptr, 0, thr, use_thr, markbits.as<u64>(), (u64 *)nullptr, (u64 *)nullptr, (u32 *)nullptr, 0, (i64)0, (const double *)nullptr,
                               (const double *)nullptr);
            KERNEL_CHECK();
        }
        if (!packed) {
            // (plain cleanups of up to 1.3e5 rows: the index sort in ONE launch — its three passes were nine launches, launch bound)
            bool coop_done = false;
            SG_TRY(radix_sort_pairs_u64_u32_coop(keys.as<u64>(), idx.as<u32>(), keys2.as<u64>(), idx2.as<u32>(), Tk, 64 - nbits, 64, &in_tmp, &coop_done));
            if (coop_done) sus_coop = true;
            else SG_TRY(radix_sort_pairs_u64_u32(keys.as<u64>(), idx.as<u32>(), keys2.as<u64>(), idx2.as<u32>(), Tk, 64 - nbits, 64, &in_tmp));
        }
        ks = ks_sorted ? ks_sorted : (in_tmp ? keys2.as<u64>() : keys.as<u64>());
        is = packed ? nullptr : (in_tmp ? idx2.as<u32>() : idx.as<u32>());
        bool merges_found = false, patch_zeroed = false;               // lazy: dirtybits already filled by the fix-up passes
        // (no key has a partner — Tsort == 0 —: every term is a single, decided by k_mark_singles; the patch bitmap is zeroed in the same launch)
        // (and without the lazy flow the kept-term bitmap, which the segment sums then fill: one launch for both)
        const size_t space_now = (size_t)((squared && packed) ? Tk : T);
        const bool mark_zeroed = !lazy_a && Tsort > 0;
        if (!(fix_bits < 64 && Tsort > 0 && lazy_a))
            SG_TRY(zero_two(collision.p, 16, Tsort == 0 ? patchbits.p : nullptr, (space_now + 63) / 64 * 8, mark_zeroed ? markbits.p : nullptr, (space_now + 31) / 32 * 4));
        if (fix_bits < 64 && Tsort > 0) {
            const i64 n_ch = (Tsort + 63) / 64;
            SG_TRY(fixlist.alloc((size_t)n_ch * 8 + 16));                         // one word of flags per 64 positions
            u32 *dirty_fx = nullptr;
            if (lazy_a) {                                                           // the chunks with merged terms are found in the same pass
                const i64 n_dw = (n_ch + 31) / 32;
                SG_TRY(dirtybits.alloc((size_t)n_dw * 4 + 16));
                // (with the flagged keys alone in the sorted array the lazy flow is never given up below: its patch bitmap is zeroed here too)
                patch_zeroed = sus_active && patchbits.p != nullptr;
                SG_TRY(zero_two(collision.p, 16, dirtybits.p, (size_t)n_dw * 4 + 16, patch_zeroed ? patchbits.p : nullptr,
                                (size_t)((((squared && packed) ? Tk : T) + 63) / 64) * 8));
                dirty_fx = dirtybits.as<u32>();
                merges_found = true;
            }
            const dim3 gff((unsigned)grid_for((Tsort + 255) / 256, 4, 8192));
            const dim3 gfw((unsigned)((n_ch + 255) / 256));
            if (packed) {
                hipLaunchKernelGGL(k_fixup_find<true>, gff, dim3(256), 0, st, ks, Tsort, 64 - fix_bits, hI.as<u64>(), hO_p, L, inner == outer, fixlist.as<u64>(), dirty_fx);
                hipLaunchKernelGGL(k_fixup_work<true>, gfw, dim3(256), 0, st, ks, (u32 *)nullptr, Tsort, 64 - fix_bits, fixlist.as<u64>(), collision.as<u32>() + 1,
                                   hI.as<u64>(), hO_p, L, inner == outer, dirty_fx);
            } else {
                hipLaunchKernelGGL(k_fixup_find<false>, gff, dim3(256), 0, st, ks, Tsort, 64 - fix_bits, (const u64 *)nullptr, (const u64 *)nullptr, L, false, fixlist.as<u64>(), dirty_fx);
                hipLaunchKernelGGL(k_fixup_work<false>, gfw, dim3(256), 0, st, ks, is, Tsort, 64 - fix_bits, fixlist.as<u64>(), collision.as<u32>() + 1,
                                   (const u64 *)nullptr, (const u64 *)nullptr, L, false, dirty_fx);
            }
            KERNEL_CHECK();
        }
        if (Tsort == 0) {
            // nothing can merge (no key has a partner): every term is a single, decided by k_mark_singles (patch bitmap zeroed above)
        } else {
            int G = 1;                                   // lanes per verified candidate: one 16-byte chunk each
            while (G < W / 2 && G < 64) G <<= 1;
            const i64 n_chunks = (Tsort + 63) / 64;
            // The kernel is latency bound (dependent key load -> operand table gathers -> store per 64-position chunk; rocprofv3: 6 % of
            // the wave cycles issue, 53 % wait on memory), so it wants many short waves rather than few long ones: cfg3 6.39 / 6.16 /
            // 6.10 / 6.04 ms at 2^15 / 2^17 / 2^19 / 2^21 wavefronts (a wave also decodes the chunk after its range to close the
            // segment it carries, so one chunk per wave reads the keys twice — still the fastest).
            const i64 HS_WAVES = [] { const char *e = SG_TUNE("SYMGPU_HS_WAVES"); return e ? atoll(e) : (i64)1 << 21; }();
            const i64 cpw = (n_chunks + HS_WAVES - 1) / HS_WAVES;     // <= HS_WAVES wavefronts, each on a contiguous range of chunks
            const i64 n_waves = (n_chunks + cpw - 1) / cpw;
            const dim3 gs((unsigned)((n_waves + 3) / 4));
            const u64 *nul = nullptr;
            const double *nud = nullptr;
            const i64 space = (squared && packed) ? Tk : T;               // index space of markbits / sum_of
            // Adaptive: when many chunks hold merged terms — an input full of repeated rows: the followers' bitmap atomics of the lazy
            // flow then cost more than the heads' scatter it saves (10^8 pairs with ~5 copies of every row: 6.2 ms lazy against 4.5 ms
            // filed) — every term is filed from the sorted order after all; k_mark_singles' pass was wasted (one count read-back).
            bool lazy_now = lazy_a;
            if (lazy_a && merges_found && lazy_env != 1 && !sus_active) {       // (sus_active: the sorted keys ARE the merged terms)
                u32 *dcount = collision.as<u32>() + 3;
                hipLaunchKernelGGL(k_count_bits, dim3(grid_for((n_chunks + 31) / 32)), dim3(256), 0, st, dirtybits.as<u32>(), (n_chunks + 31) / 32, dcount);
                KERNEL_CHECK();
                u32 h_dirty = 0;
                SG_TRY(read_back_words(dcount, 1, nullptr, 0, &h_dirty));
                if ((i64)h_dirty * 8 > n_chunks) lazy_now = false;
            }
            lazy_final = lazy_now;
            if (lazy_now) { if (!patch_zeroed) SG_TRY(zero_two(patchbits.p, (size_t)((space + 63) / 64) * 8, nullptr, 0)); }
            else if (!mark_zeroed) SG_TRY(zero_two(markbits.p, (size_t)((space + 31) / 32) * 4, nullptr, 0));
            u32 *patch_p = lazy_now ? patchbits.as<u32>() : nullptr;
            const u32 *zero_len_p = nullptr;
            const bool zero_on = [] { const char *e = SG_TUNE("SYMGPU_CLEANUP_ZEROSEG"); return !(e && e[0] == '0'); }();
            if (squared && packed && zero_on) {
                // the identity segment (the N diagonal pairs and whatever else multiplies to the identity) in parallel, see k_zero_partial
                const i64 n_zb = (Tsort + ZB - 1) / ZB;
                SG_TRY(zpart.alloc((size_t)n_zb * 16));
                SG_TRY(zcount.alloc((size_t)n_zb * 4 + 16));
                u32 *zl = zcount.as<u32>() + n_zb;
                hipLaunchKernelGGL(k_zero_partial, dim3((unsigned)n_zb), dim3(256), 0, st, ks, Tsort, hI.as<u64>(), hO_p, L, inner, W, ci,
                                   zpart.as<double>(), zcount.as<u32>(), collision.as<u32>(), (u32)Ni, lazy_now ? markbits.as<u32>() : (u32 *)nullptr);
                if (diag_side) { HIP_TRY(hipStreamWaitEvent(st, ctx().ev_join, 0)); diag_side = false; }
                hipLaunchKernelGGL(k_zero_close, dim3(1), dim3(64), 0, st, ks, zpart.as<double>(), zcount.as<u32>(), n_zb, L, (u32)Ni, thr, use_thr,
                                   markbits.as<u32>(), sum_of.as<double>(), zl, patch_p, diag_seq.as<double>(), collision.as<u32>() + 2);
                zero_len_p = zl;
            }
            const u32 *dirty_p = nullptr;
            dim3 gsl = gs;
            if (lazy_now) {
                // the chunks that hold a member of a segment of more than one element; k_heads_sums then works on those only
                const i64 n_dw = (n_chunks + 31) / 32;
                if (!merges_found) {
                SG_TRY(dirtybits.alloc((size_t)n_dw * 4 + 16));
                HIP_TRY(hipMemsetAsync(dirtybits.p, 0, (size_t)n_dw * 4 + 16, st));
                const dim3 gf((unsigned)grid_for((Tsort + 255) / 256, 4, 8192));
                if (packed) hipLaunchKernelGGL(k_find_merges<true>, gf, dim3(256), 0, st, ks, Tsort, zero_len_p, L, hI.as<u64>(), hO_p, inner == outer ? 1 : 0, dirtybits.as<u32>());
                else hipLaunchKernelGGL(k_find_merges<false>, gf, dim3(256), 0, st, ks, Tsort, zero_len_p, L, nul, nul, 0, dirtybits.as<u32>());
                KERNEL_CHECK();
                }
                dirty_p = dirtybits.as<u32>();
                gsl = dim3((unsigned)(((n_chunks + 7) / 8 + 3) / 4));
            }
            if (packed)
                hipLaunchKernelGGL((k_heads_sums<true, true>), gsl, dim3(256), 0, st, ks, (const u32 *)nullptr, Tsort, nul, W, inner, (u32)Ni, outer, G, nud,
                                   collision.as<u32>(), hI.as<u64>(), hO_p, L, ci, co, thr, use_thr, markbits.as<u32>(), sum_of.as<double>(), cpw, squared && packed ? 1 : 0,
                                   zero_len_p, patch_p, dirty_p);
            else if (pair)
                hipLaunchKernelGGL((k_heads_sums<true, false>), gsl, dim3(256), 0, st, ks, is, Tsort, nul, W, inner, (u32)Ni, outer, G, coeff,
                                   collision.as<u32>(), nul, nul, L, nud, nud, thr, use_thr, markbits.as<u32>(), sum_of.as<double>(), cpw, squared && packed ? 1 : 0,
                                   (const u32 *)nullptr, patch_p, dirty_p);
            else
                hipLaunchKernelGGL((k_heads_sums<false, false>), gsl, dim3(256), 0, st, ks, is, Tsort, rows, W, nul, 1u, nul, G, coeff,
                                   collision.as<u32>(), nul, nul, L, nud, nud, thr, use_thr, markbits.as<u32>(), sum_of.as<double>(), cpw, squared && packed ? 1 : 0,
                                   (const u32 *)nullptr, patch_p, dirty_p);
        }
        KERNEL_CHECK();
        // the output stage's prefix over the kept-term bitmap is formed BEFORE this attempt's status words are read: the count of kept terms
        // comes back with them (an attempt that has to be repeated throws the prefix away)
        pre.n_out = -1;
        pre.wide = emit_is_fused(W / 2);
        SG_TRY(emit_prefix(markbits.as<u32>(), (squared && packed) ? Tk : T, pre.wordprefix, pre.total, pre.wide));
        pre.touched = false;
        if (emit_is_fused(W / 2) && !SG_TUNE("SYMGPU_EMIT_TOUCH")) {      // (the output stage's bitmaps back into the cache: queued ahead of the read-back, not behind it)
            LazyEmit lzt;
            if (lazy_final) { lzt.mode = packed ? 1 : 2; lzt.patchbits = patchbits.as<u32>(); lzt.e_lo = e_lo.as<u32>(); lzt.e_hi = e_hi.as<u32>(); }
            SG_TRY(emit_touch(markbits.as<u32>(), (squared && packed) ? Tk : T, lzt, pre));
        }
        // (ONE trip to the host for the count and the status words.  Measured and dropped, round 6: small results allocated for every index
        // and the output stage queued before the count is in — same-box A/B 3 - 14 us SLOWER per call, 72 -> 75 us at 10^3 rows.)
        u32 hback[6] = {0, 0, 0, 0, 0, 0};
        {
            u32 hb[4] = {0, 0, 0, 0};
            SG_TRY(read_back_words(pre.total.as<u32>(), 1, collision.as<u32>(), 2, hb, sus_coop ? radix_sort_coop_flag() : nullptr));
            hback[0] = hb[0]; hback[4] = hb[1]; hback[5] = hb[2]; hback[1] = hb[3];
        }
        const u32 hflags[2] = {hback[4], hback[5]};
        pre.n_out = hback[0];
        if (sus_coop) {                                            // the one-launch sort of the flagged keys gave up at a barrier (GPU shared): its
            bool timed_out = false;                                // output is garbage; the form is off now, the next attempt sorts with launches
            radix_sort_coop_note(hback[1], &timed_out);
            if (timed_out) continue;
        }
        if (hflags[1]) { nb = 64; packed = false; squared = false; Tk = T; continue; }   // a long mixed prefix run: redo with a full 64-bit sort over all pairs, same seed
        ok = (hflags[0] == 0);
        if (!ok) { ++seed; ++g_hash_reseeds; }      // genuine 64-bit hash collision: reseed and retry
    }
    if (!ok) {
        set_error("cleanup: 64-bit row-hash collision survived 4 reseeds");
        return SYMGPU_E_COLLISION;
    }
    const bool tri = squared && packed;
    LazyEmit lz;
    lz.no_one_outer = SG_TUNE("SYMGPU_EMIT_NO_ONE_OUTER") ? 1 : 0;
    if (lazy_final) {
        lz.mode = packed ? 1 : 2; lz.squared = tri ? 1 : 0;
        lz.patchbits = patchbits.as<u32>(); lz.e_lo = e_lo.as<u32>(); lz.e_hi = e_hi.as<u32>();
        lz.ci = ci; lz.co = co; lz.coeff = coeff;
    }
    return cleanup_finish(markbits.as<u32>(), sum_of.as<double>(), tri ? Tk : T, pair, rows, W, inner, Ni, outer, out, Wq_out, tri, lz, want_first, &pre);
}

}  // namespace symgpu

using namespace symgpu;

extern "C" {

int symgpu_cleanup_dev(symgpu_op_t in, double thr, int use_thr, symgpu_op_t *out) {
    SG_ENTER(in);
    SG_REQUIRE(in && out, "cleanup_dev: null handle");
    SG_REQUIRE(in->coeff || in->T == 0, "cleanup_dev: operator has no coefficients");
    return cleanup_core(in->rows, in->coeff, in->T, 2 * in->Wq, nullptr, 0, nullptr, 0, thr, use_thr, out, in->Wq, nullptr, nullptr, 1);
}

int symgpu_mul_cleanup_dev(symgpu_op_t inner, symgpu_op_t outer, int inner_is_left, double thr, int use_thr, symgpu_op_t *out) {
    SG_ENTER(inner, outer);
    SG_REQUIRE(inner && outer && out, "mul_cleanup_dev: null handle");
    SG_REQUIRE(inner->Wq == outer->Wq, "mul_cleanup_dev: operands must share Wq");
    const i64 Ni = inner->T, No = outer->T;
    const i64 T = Ni * No;
    if (T == 0) return cleanup_core(nullptr, nullptr, 0, 2 * inner->Wq, nullptr, 0, nullptr, 0, thr, use_thr, out, inner->Wq, nullptr, nullptr, 1);
    SG_REQUIRE(inner->coeff && outer->coeff, "mul_cleanup_dev: operands have no coefficients");
    SG_REQUIRE(No == 0 || Ni < ((i64)1 << 32) / No, "mul_cleanup_dev: Ni*No must stay below 2^32 (tile the outer operand)");
    return cleanup_core(nullptr, nullptr, T, 2 * inner->Wq, inner->rows, Ni, outer->rows, No, thr, use_thr, out, inner->Wq, inner->coeff, outer->coeff,
                        inner_is_left);
}

// The same two calls, with the first-occurrence index of every output term kept on the result (symgpu_op_first_index): what a caller needs
// to merge cleaned partial results of ONE product in the reference's order (symmer_amd/parallel.py: hash-partitioned multi-GPU cleanup).
int symgpu_cleanup_indexed_dev(symgpu_op_t in, double thr, int use_thr, symgpu_op_t *out) {
    SG_ENTER(in);
    SG_REQUIRE(in && out, "cleanup_indexed_dev: null handle");
    SG_REQUIRE(in->coeff || in->T == 0, "cleanup_indexed_dev: operator has no coefficients");
    return cleanup_core(in->rows, in->coeff, in->T, 2 * in->Wq, nullptr, 0, nullptr, 0, thr, use_thr, out, in->Wq, nullptr, nullptr, 1, true);
}

int symgpu_mul_cleanup_indexed_dev(symgpu_op_t inner, symgpu_op_t outer, int inner_is_left, double thr, int use_thr, symgpu_op_t *out) {
    SG_ENTER(inner, outer);
    SG_REQUIRE(inner && outer && out, "mul_cleanup_indexed_dev: null handle");
    SG_REQUIRE(inner->Wq == outer->Wq, "mul_cleanup_indexed_dev: operands must share Wq");
    const i64 Ni = inner->T, No = outer->T;
    const i64 T = Ni * No;
    if (T == 0) return cleanup_core(nullptr, nullptr, 0, 2 * inner->Wq, nullptr, 0, nullptr, 0, thr, use_thr, out, inner->Wq, nullptr, nullptr, 1, true);
    SG_REQUIRE(inner->coeff && outer->coeff, "mul_cleanup_indexed_dev: operands have no coefficients");
    SG_REQUIRE(Ni < ((i64)1 << 32) / No, "mul_cleanup_indexed_dev: Ni*No must stay below 2^32 (tile the outer operand)");
    return cleanup_core(nullptr, nullptr, T, 2 * inner->Wq, inner->rows, Ni, outer->rows, No, thr, use_thr, out, inner->Wq, inner->coeff, outer->coeff,
                        inner_is_left, true);
}

int symgpu_op_first_index(symgpu_op_t op, uint64_t *first_host, int64_t capacity) {
    SG_ENTER(op);
    SG_REQUIRE(op && (first_host || op->T == 0), "op_first_index: null argument");
    SG_REQUIRE(op->first || op->T == 0, "op_first_index: the operator does not come from an *_indexed cleanup");
    if (capacity < op->T) { set_error("op_first_index: capacity %lld < %lld rows", (long long)capacity, (long long)op->T); return SYMGPU_E_CAPACITY; }
    if (op->T > 0) {
        HIP_TRY(hipMemcpyAsync(first_host, op->first, (size_t)op->T * 8, hipMemcpyDeviceToHost, ctx().stream));
        count_d2h((size_t)op->T * 8);
        HIP_TRY(hipStreamSynchronize(ctx().stream));
    }
    return SYMGPU_OK;
}

static int finish_to_host(symgpu_op_t res, uint64_t *out_rows, double *out_coeff, int64_t capacity, int64_t *n_out) {
    if (n_out) *n_out = res->T;
    int rc = SYMGPU_OK;
    if (res->T > capacity) {
        set_error("output capacity %lld < %lld rows", (long long)capacity, (long long)res->T);
        rc = SYMGPU_E_CAPACITY;
    } else if (res->T > 0) {
        rc = symgpu_op_download(res, out_rows, out_coeff, capacity);
    }
    symgpu_op_free(res);
    return rc;
}

int symgpu_cleanup(const uint64_t *rows, const double *coeff, int64_t T, int W, double thr, int use_thr, uint64_t *out_rows,
                   double *out_coeff, int64_t capacity, int64_t *n_out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(T >= 0 && W >= 2 && (W % 2) == 0 && capacity >= 0, "cleanup: sizes (W must be 2*Wq)");
    SG_REQUIRE(T == 0 || (rows && coeff), "cleanup: null input");
    symgpu_op_t in = nullptr, res = nullptr;
    SG_TRY(symgpu_op_upload(rows, coeff, T, W / 2, &in));
    if (T == 0 && !in->coeff) { /* empty upload has no coeff buffer; cleanup_core handles T == 0 first */ }
    int rc = cleanup_core(in->rows, in->coeff, T, W, nullptr, 0, nullptr, 0, thr, use_thr, &res, W / 2, nullptr, nullptr, 1);
    symgpu_op_free(in);
    if (rc != SYMGPU_OK) return rc;
    return finish_to_host(res, out_rows, out_coeff, capacity, n_out);
}

int symgpu_mul_cleanup(const uint64_t *inner, const double *ci, int64_t Ni, const uint64_t *outer, const double *co, int64_t No,
                       int Wq, int inner_is_left, double thr, int use_thr, uint64_t *out_rows, double *out_coeff, int64_t capacity,
                       int64_t *n_out) {
    SG_TRY(require_ctx());
    SG_REQUIRE(Ni >= 0 && No >= 0 && Wq >= 1 && capacity >= 0, "mul_cleanup: sizes");
    if (Ni == 0 || No == 0) { if (n_out) *n_out = 0; return SYMGPU_OK; }
    SG_REQUIRE(inner && outer && ci && co, "mul_cleanup: null input");
    symgpu_op_t a = nullptr, b = nullptr, res = nullptr;
    int rc = symgpu_op_upload(inner, ci, Ni, Wq, &a);
    if (rc == SYMGPU_OK) rc = symgpu_op_upload(outer, co, No, Wq, &b);
    if (rc == SYMGPU_OK) rc = symgpu_mul_cleanup_dev(a, b, inner_is_left, thr, use_thr, &res);
    symgpu_op_free(a); symgpu_op_free(b);
    if (rc != SYMGPU_OK) return rc;
    return finish_to_host(res, out_rows, out_coeff, capacity, n_out);
}

}  // extern "C"
