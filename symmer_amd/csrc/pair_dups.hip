// pair_dups.hip — which pairs of a product share their row with another pair, WITHOUT sorting the pair keys (round 6).
//
// The fused product + cleanup (cleanup.hip; reference: PauliwordOp.__mul__ -> cleanup, symmer/operators/base.py:821-859, utils.py:230-279)
// keys the product row of the pair (i, o) with hI[i] ^ hO[o], a GF(2)-linear row hash.  A product of operators without repeated rows merges
// next to nothing (BASELINE cfg3: the 10^4 diagonal pairs of P * P and a few hundred others out of 5e7), every other pair is decided in index
// order (k_mark_singles), and all the sort is for is to FIND the few pairs that have a partner.  Rounds 4-5 found them behind two radix passes
// over all 5e7 keys (k_find_suspects: 0.67 ms of scatter / histogram / scan / flag pass at cfg3).
//
// Here the keys are never moved.  The terms of each operand are bucketed by the top B bits of their hash (k_pd_bucket: 10^4 terms, one
// workgroup); the pairs whose key starts with beta are then exactly the TILES (bucket a of one operand) x (bucket a ^ beta of the other), for
// all a.  A persistent workgroup takes the product buckets beta one after the other, with the bucketed hash words of the operands in its LDS:
//   tiles   per a: the size of tile (a, a ^ beta); a block scan gives every tile its first pair number (P * P: only the tiles a < a ^ beta —
//           the tile (a ^ beta, a) holds the same pairs; beta = 0, pairs inside one bucket, takes a loop of its own);
//   count   every lane takes q = ceil(P / 1024) consecutive pair numbers: finds its first tile by bisection and walks on from there — one
//           LDS read, one XOR and one returning LDS add on a 4-bit counter per pair, the key words stay in registers;
//   list    a pair whose counter had been hit before, or reads >= 2 now, is listed (4 % of them: chance hits included);
//   match   the listed words are chained by a 2,048-way LDS hash; a listed pair walks its chain, an equal word is followed up with the 64-bit
//           hashes (bucket-ordered copy in HBM / L2) — equal: the pair's bit is set in the flag bitmap, indexed like the keys in index order.
// A word of zero is listed whatever its count and flagged when its 64-bit hash is zero (the identity: P * P's diagonal and whatever else
// multiplies to it).  The caller compacts the flagged keys from the index-ordered key array (they come out in index order: the order the segment
// machinery wants inside equal keys) and sorts those few thousand.  Anything that does not fit — a bucket too long (operands full of repeated
// rows), a counter that saturates, a list that overflows — raises `giveup` and the caller takes the sorted path; nothing is decided here that
// the 64-bit hashes do not decide there.
#include "common.h"

namespace symgpu {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
constexpr int PD_THREADS = 1024;
constexpr int PD_QUOTA = 16;                        // pairs per lane and product bucket: 16,384 pairs per bucket at most
constexpr int PD_TARGET = 12288;                    // pairs per product bucket the bucket width is chosen for
constexpr int PD_SLOT_BITS = 17;                    // 4-bit counters: 64 KiB
constexpr int PD_CAND = 2048, PD_CHAIN = 2048;
constexpr int PD_MAX_B = 13;
constexpr int PD_MAX_BUCKET = 255;                  // terms per operand bucket (longer: the hashes are not spread — repeated rows)
constexpr size_t PD_LDS_MAX = 160 * 1024 - 256;

__device__ __forceinline__ u32 pd_cnt_word(u32 v) { return v >> (32 - PD_SLOT_BITS + 3); }
__device__ __forceinline__ u32 pd_cnt_shift(u32 v) { return ((v >> (32 - PD_SLOT_BITS)) << 2) & 28u; }

// one workgroup per operand: terms in bucket order (unordered inside a bucket) — hash word behind the bucket bits, full hash, term index
__global__ __launch_bounds__(1024) void k_pd_bucket(const u64 *__restrict__ hI, int nI, const u64 *__restrict__ hO, int nO, int B, u32 *__restrict__ tab_w,
                                                    u64 *__restrict__ tab_h, u32 *__restrict__ tab_idx, unsigned short *__restrict__ start, u32 *__restrict__ giveup) {
    __shared__ u32 s_cnt[(1 << PD_MAX_B) + 1];
    __shared__ u32 s_wsum[16];
    const int side = blockIdx.x, nb = 1 << B, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 *h = side ? hO : hI;
    const int n = side ? nO : nI, base = side ? nI : 0;
    for (int a = tid; a <= nb; a += 1024) s_cnt[a] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&s_cnt[h[i] >> (64 - B)], 1u);
    __syncthreads();
    const int per = (nb + 1023) / 1024;                                     // buckets per lane, consecutive
    u32 mine = 0, longest = 0;
    for (int k = 0; k < per; ++k) { const int a = tid * per + k; if (a < nb) { const u32 c = s_cnt[a]; mine += c; longest = c > longest ? c : longest; } }
    if (longest > (u32)PD_MAX_BUCKET) atomicOr(giveup, 8u);
    u32 inc = mine;
    for (int off = 1; off < 64; off <<= 1) { const u32 t = (u32)__shfl_up((int)inc, off); if (lane >= off) inc += t; }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    u32 run = inc - mine;
    for (int w2 = 0; w2 < wave; ++w2) run += s_wsum[w2];
    __syncthreads();
    unsigned short *st = start + (size_t)side * (nb + 1);
    for (int k = 0; k < per; ++k) {
        const int a = tid * per + k;
        if (a < nb) { const u32 c = s_cnt[a]; s_cnt[a] = run; st[a] = (unsigned short)run; run += c; }
    }
    if (tid == 1023) st[nb] = (unsigned short)n;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const u64 hi = h[i];
        const u32 p = atomicAdd(&s_cnt[hi >> (64 - B)], 1u);
        tab_w[base + p] = (u32)((hi << B) >> 32); tab_h[base + p] = hi; tab_idx[base + p] = (u32)i;
    }
}

struct PairDupArgs {
    const u32 *tab_w; const u64 *tab_h; const u32 *tab_idx; const unsigned short *start;
    int nI, nO, B, squared;                          // squared: ONE table (nO = 0), pairs i >= o
    i64 Ni;                                          // terms of the inner operand (index of a pair: o * Ni + i; squared: PairKeyArgs' compacted slot)
    u64 *flags; u32 *giveup;
};

__device__ __forceinline__ void pd_flag(const PairDupArgs &a, u32 x, u32 y) {
    // x: position in the inner table, y: in the outer one (squared: both in the one table)
    const u32 ix = a.tab_idx[x], iy = a.tab_idx[a.squared ? y : (u32)a.nI + y];
    i64 pos;
    if (a.squared) {
        const i64 i = ix > iy ? ix : iy, o = ix > iy ? iy : ix;
        pos = o * a.Ni - o * (o - 1) / 2 + (i - o);
    } else {
        pos = (i64)iy * a.Ni + ix;
    }
    atomicOr(reinterpret_cast<unsigned long long *>(a.flags + (pos >> 6)), 1ULL << (pos & 63));
}
__device__ __forceinline__ u64 pd_hash(const PairDupArgs &a, u32 xy) {
    return a.tab_h[xy & 0xFFFFu] ^ a.tab_h[(a.squared ? 0u : (u32)a.nI) + (xy >> 16)];
}

__global__ __launch_bounds__(PD_THREADS) void k_pair_dups(const PairDupArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = PD_THREADS;
    const int nb = 1 << a.B, nTab = a.nI + a.nO;
    u32 *s_w = reinterpret_cast<u32 *>(smem);                                  // [nTab]: inner table, then the outer one
    u32 *s_cnt = s_w + ((nTab + 3) & ~3);                                      // 4-bit counters, eight per word
    u32 *s_tiles = s_cnt + (1 << (PD_SLOT_BITS - 3));                          // [nb]: first pair number << 13 | a
    u32 *s_lw = s_tiles + nb;                                                  // listed pairs: word, positions, chain link
    u32 *s_lxy = s_lw + PD_CAND;
    u32 *s_next = s_lxy + PD_CAND;
    u32 *s_head = s_next + PD_CAND;                                            // [PD_CHAIN]
    unsigned short *s_sI = reinterpret_cast<unsigned short *>(s_head + PD_CHAIN);   // [nb + 1] bucket starts of the inner table
    unsigned short *s_sO = a.squared ? s_sI : s_sI + (nb + 2);                 // ... and of the outer one
    __shared__ u32 s_wsum[16], s_wtil[16], s_nc, s_total, s_ntiles, s_over;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (*a.giveup & 8u) return;                                                 // a bucket longer than the positions below can address: not for this path
    const u32 *s_wO = a.squared ? s_w : s_w + a.nI;
    for (int i = tid; i < nTab; i += NT) s_w[i] = a.tab_w[i];
    for (int i = tid; i <= nb; i += NT) { s_sI[i] = a.start[i]; if (!a.squared) s_sO[i] = a.start[nb + 1 + i]; }
    __syncthreads();
    // P * P: the diagonal (the identity, N pairs) is flagged as it is
    if (a.squared)
        for (int x = blockIdx.x * NT + tid; x < a.nI; x += gridDim.x * NT) pd_flag(a, (u32)x, (u32)x);
    for (int beta = blockIdx.x; beta < nb; beta += gridDim.x) {
        {
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int i = tid; i < (1 << (PD_SLOT_BITS - 5)); i += NT) reinterpret_cast<u32x4 *>(s_cnt)[i] = z;
        }
        for (int i = tid; i < PD_CHAIN; i += NT) s_head[i] = 0xFFFFFFFFu;
        if (tid == 0) { s_nc = 0; s_over = 0; }
        const bool inside = a.squared && beta == 0;                             // pairs inside one bucket: no tiles, a loop per bucket
        u32 P = 0, ntl = 0;
        u32 hw[PD_QUOTA], xy[PD_QUOTA], old[PD_QUOTA];
        u32 rem = 0;
        if (!inside) {
            // ---- tiles of this product bucket: sizes, scan, list of the non-empty ones
            const int per = (nb + NT - 1) / NT;
            u32 mysum = 0, mytiles = 0;
            for (int k = 0; k < per; ++k) {
                const int av = tid * per + k;
                if (av < nb) {
                    const int b = av ^ beta;
                    const u32 ca = s_sI[av + 1] - s_sI[av], cb = s_sO[b + 1] - s_sO[b];
                    const u32 c = (!a.squared || av < b) ? ca * cb : 0u;
                    mysum += c; mytiles += c ? 1u : 0u;
                }
            }
            u32 inc_s = mysum, inc_t = mytiles;                                 // (two sums: a tile holds up to 255 x 255 pairs)
            for (int off = 1; off < 64; off <<= 1) {
                const u32 ts = (u32)__shfl_up((int)inc_s, off), tt = (u32)__shfl_up((int)inc_t, off);
                if (lane >= off) { inc_s += ts; inc_t += tt; }
            }
            if (lane == 63) { s_wsum[wave] = inc_s; s_wtil[wave] = inc_t; }
            __syncthreads();
            u32 base_s = 0, base_t = 0;
            for (int w2 = 0; w2 < wave; ++w2) { base_s += s_wsum[w2]; base_t += s_wtil[w2]; }
            if (tid == NT - 1) { s_total = base_s + inc_s; s_ntiles = base_t + inc_t; }
            {
                u32 off = base_s + inc_s - mysum, tix = base_t + inc_t - mytiles;
                for (int k = 0; k < per; ++k) {
                    const int av = tid * per + k;
                    if (av < nb) {
                        const int b = av ^ beta;
                        const u32 ca = s_sI[av + 1] - s_sI[av], cb = s_sO[b + 1] - s_sO[b];
                        const u32 c = (!a.squared || av < b) ? ca * cb : 0u;
                        if (c) { if (off < (1u << 19)) s_tiles[tix] = (off << 13) | (u32)av; ++tix; off += c; }
                    }
                }
            }
            __syncthreads();
            P = s_total; ntl = s_ntiles;
            if (P > (u32)(PD_QUOTA * NT)) {                                     // (block-uniform)
                if (tid == 0) atomicOr(a.giveup, 1u);
                __syncthreads();
                continue;
            }
            if (P == 0) { __syncthreads(); continue; }
            // ---- count: a lane takes q consecutive pairs; words, positions and counter answers stay in registers
            const u32 q = (u32)__builtin_amdgcn_readfirstlane((int)((P + NT - 1) / NT));   // block-uniform
            const u32 p0 = tid * q;
            rem = p0 < P ? (P - p0 < q ? P - p0 : q) : 0u;
            {
                int lo = 0, hi = (int)ntl - 1;                                  // last tile that starts at or before p0
                const u32 ps = p0 < P ? p0 : 0u;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((s_tiles[mid] >> 13) <= ps) lo = mid; else hi = mid - 1; }
                u32 t = (u32)lo;
                u32 td = s_tiles[t];
                u32 av = td & 0x1FFFu, b = av ^ (u32)beta;
                u32 sb = s_sO[b];
                const u32 cb = s_sO[b + 1] - sb;
                const u32 r = ps - (td >> 13);
                const u32 x0 = r / cb, y0 = r - x0 * cb;
                u32 ax = s_sI[av] + x0, ax_end = s_sI[av + 1], ay = sb + y0, ay_end = sb + cb;
                u32 wI = s_w[ax];
#pragma unroll
                for (int k = 0; k < PD_QUOTA; ++k) {
                    if ((u32)k < q) {                                            // (scalar)
                        const u32 v = wI ^ s_wO[ay];
                        hw[k] = v; xy[k] = ax | (ay << 16);
                        const u32 incr = (u32)k < rem ? 1u << pd_cnt_shift(v) : 0u;
                        old[k] = atomicAdd(&s_cnt[pd_cnt_word(v)], incr);
                        ++ay;
                        if (ay == ay_end) {
                            ++ax;
                            if (ax == ax_end) {
                                t = t + 1 < ntl ? t + 1 : t;                      // (behind the last tile the lane has no pairs left: it walks the last tile again, adding zeros)
                                td = s_tiles[t]; av = td & 0x1FFFu; b = av ^ (u32)beta;
                                ax = s_sI[av]; ax_end = s_sI[av + 1]; sb = s_sO[b]; ay_end = s_sO[b + 1];
                            }
                            ay = sb;
                            wI = s_w[ax];
                        }
                    }
                }
            }
        } else {
            // ---- beta = 0 of P * P: the pairs x > y inside a bucket; a lane takes whole buckets (a few pairs each)
            __syncthreads();
            for (int av = tid; av < nb; av += NT) {
                const u32 s0 = s_sI[av], s1 = s_sI[av + 1];
                for (u32 x = s0 + 1; x < s1; ++x)
                    for (u32 y = s0; y < x; ++y) {
                        const u32 v = s_w[x] ^ s_w[y];
                        const u32 o = (atomicAdd(&s_cnt[pd_cnt_word(v)], 1u << pd_cnt_shift(v)) >> pd_cnt_shift(v)) & 15u;
                        if (o == 15u) s_over = 1;
                    }
            }
        }
        u32 live = 0, later = 0;
        if (!inside) {
#pragma unroll
            for (int k = 0; k < PD_QUOTA; ++k) {
                if ((u32)k < rem) {
                    const u32 v = hw[k];
                    live |= 1u << k;
                    const u32 o = (old[k] >> pd_cnt_shift(v)) & 15u;
                    if (o == 15u) s_over = 1;
                    if (o || v == 0u) later |= 1u << k;                          // (a zero word: listed whatever its count — the identity is decided on the list)
                }
            }
        }
        __syncthreads();
        if (s_over) {                                                           // a counter saturated: rows repeated all over
            if (tid == 0) atomicOr(a.giveup, 2u);
            __syncthreads();
            continue;
        }
        // ---- list: pairs whose counter was hit before them or reads >= 2 now
        if (!inside) {
#pragma unroll
            for (int k = 0; k < PD_QUOTA; ++k) {
                bool li = false;
                if ((live >> k) & 1u) li = ((later >> k) & 1u) || ((s_cnt[pd_cnt_word(hw[k])] >> pd_cnt_shift(hw[k])) & 15u) >= 2u;
                const u64 m = __ballot(li);
                if (m) {
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(&s_nc, (u32)__popcll(m));
                    base = (u32)__builtin_amdgcn_readfirstlane((int)base);
                    const u32 n = base + (u32)__popcll(m & ((1ULL << lane) - 1ULL));
                    if (li && n < (u32)PD_CAND) { s_lw[n] = hw[k]; s_lxy[n] = xy[k]; }
                }
            }
        } else {
            for (int av = tid; av < nb; av += NT) {
                const u32 s0 = s_sI[av], s1 = s_sI[av + 1];
                for (u32 x = s0 + 1; x < s1; ++x)
                    for (u32 y = s0; y < x; ++y) {
                        const u32 v = s_w[x] ^ s_w[y];
                        if (v == 0u || ((s_cnt[pd_cnt_word(v)] >> pd_cnt_shift(v)) & 15u) >= 2u) {
                            const u32 n = atomicAdd(&s_nc, 1u);
                            if (n < (u32)PD_CAND) { s_lw[n] = v; s_lxy[n] = x | (y << 16); }
                        }
                    }
            }
        }
        __syncthreads();
        const u32 nc = s_nc;
        if (nc > (u32)PD_CAND) {
            if (tid == 0) atomicOr(a.giveup, 4u);
            __syncthreads();
            continue;
        }
        // ---- match: chains by word, every listed pair walks its chain; an equal word is decided by the 64-bit hashes
        for (u32 c = tid; c < nc; c += NT) {
            const u32 hsh = (s_lw[c] * 2654435761u) >> 21;
            s_next[c] = atomicExch(&s_head[hsh], c);
        }
        __syncthreads();
        for (u32 c = tid; c < nc; c += NT) {
            const u32 wc = s_lw[c], hsh = (wc * 2654435761u) >> 21, xyc = s_lxy[c];
            u64 Hc = 0;
            bool have = false;
            if (wc == 0u) { Hc = pd_hash(a, xyc); have = true; }
            bool hit = have && Hc == 0ULL;                                      // the identity
            for (u32 j = hit ? 0xFFFFFFFFu : s_head[hsh]; j != 0xFFFFFFFFu; j = s_next[j]) {
                if (j != c && s_lw[j] == wc) {
                    if (!have) { Hc = pd_hash(a, xyc); have = true; }
                    if (pd_hash(a, s_lxy[j]) == Hc) { hit = true; break; }
                }
            }
            if (hit) pd_flag(a, xyc & 0xFFFFu, xyc >> 16);
        }
        __syncthreads();
    }
}
static_assert(PD_CHAIN == 2048, "the chain hash keeps 11 bits");

// The flag bitmap `flags` (one bit per pair, indexed like the key array in index order; zeroed by the caller) gets the bit of every pair
// whose 64-bit key equals another pair's, and of every pair whose key is zero.  *applies = false: the operands do not fit this path (nothing
// was launched).  giveup (device word, zeroed by the caller) != 0 afterwards: the flags are incomplete, take the sorted path.
int pair_dups_dev(const u64 *hI, i64 Ni, const u64 *hO, i64 No, bool squared, i64 Tk, u64 *flags, u32 *giveup, bool *applies) {
    *applies = false;
    if (getenv("SYMGPU_CLEANUP_DIRECT") && getenv("SYMGPU_CLEANUP_DIRECT")[0] == '0') return SYMGPU_OK;
    const i64 nI = Ni, nO = squared ? 0 : No;
    if (nI + nO > 65535 || nI < 2 || (!squared && nO < 1)) return SYMGPU_OK;
    int B = 2;
    while (B < PD_MAX_B && (Tk >> B) > PD_TARGET) ++B;
    if ((Tk >> B) > PD_TARGET) return SYMGPU_OK;
    const int nb = 1 << B;
    const size_t lds = (size_t)((nI + nO + 3) & ~3) * 4 + ((size_t)1 << (PD_SLOT_BITS - 3)) * 4 + (size_t)nb * 4 + (size_t)PD_CAND * 12 + (size_t)PD_CHAIN * 4 +
                       (size_t)(squared ? 1 : 2) * (nb + 2) * 2 + 16;
    if (lds > PD_LDS_MAX) return SYMGPU_OK;
    const bool attr = SG_DEVICE_ONCE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pair_dups), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PD_LDS_MAX) == hipSuccess);
    if (!attr) return SYMGPU_OK;
    hipStream_t st = ctx().stream;
    Scratch tab;
    const size_t n = (size_t)(nI + nO);
    const size_t off_h = 0, off_w = n * 8, off_idx = off_w + ((n * 4 + 7) & ~(size_t)7), off_start = off_idx + ((n * 4 + 7) & ~(size_t)7);
    SG_TRY(tab.alloc(off_start + (size_t)2 * (nb + 1) * 2 + 16));
    char *base = tab.as<char>();
    u64 *tab_h = reinterpret_cast<u64 *>(base + off_h);
    u32 *tab_w = reinterpret_cast<u32 *>(base + off_w), *tab_idx = reinterpret_cast<u32 *>(base + off_idx);
    unsigned short *start = reinterpret_cast<unsigned short *>(base + off_start);
    hipLaunchKernelGGL(k_pd_bucket, dim3(squared ? 1 : 2), dim3(1024), 0, st, hI, (int)nI, hO, (int)nO, B, tab_w, tab_h, tab_idx, start, giveup);
    KERNEL_CHECK();
    PairDupArgs a;
    a.tab_w = tab_w; a.tab_h = tab_h; a.tab_idx = tab_idx; a.start = start;
    a.nI = (int)nI; a.nO = (int)nO; a.B = B; a.squared = squared ? 1 : 0; a.Ni = Ni; a.flags = flags; a.giveup = giveup;
    const int P = ctx().num_cu < nb ? ctx().num_cu : nb;
    hipLaunchKernelGGL(k_pair_dups, dim3((unsigned)P), dim3(PD_THREADS), lds, st, a);
    KERNEL_CHECK();
    *applies = true;
    return SYMGPU_OK;       // (`tab` goes back to the stream-ordered allocator: the kernel above is queued ahead of any reuse)
}

}  // namespace symgpu
