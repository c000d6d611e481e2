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
// all a.  A persistent workgroup takes the product buckets beta one after the other, with the bucketed hash words of the operands in its LDS
// (4,096 buckets of up to 12,288 pairs — 10,030 terms squared —, or, for squared operators, 8,192 of up to 9,216 with half the counters):
//   tiles   per a: the size of tile (a, a ^ beta); a block scan gives every tile its first pair number (P * P: only the tiles a < a ^ beta —
//           the tile (a ^ beta, a) holds the same pairs; beta = 0, pairs inside one bucket, takes a loop of its own);
//   count   every lane takes q = ceil(P / 1024) consecutive pair numbers: the tile's owner has left it its first tile and place, it walks on from there — one
//           LDS read, one XOR and one returning LDS add on a 4-bit counter per pair, the key words stay in registers;
//   list    a pair whose counter reads >= 2 now is listed (9 % of them: chance hits included);
//   match   the listed words are chained by a 1,024-way LDS hash; a listed pair walks its chain, an equal word is followed up with the 64-bit
//           hashes (bucket-ordered copy in HBM / L2) — equal: the pair's bit is set in the flag bitmap, indexed like the keys in index order.
// A word of zero is listed whatever its count and flagged when its 64-bit hash is zero (the identity: P * P's diagonal and whatever else
// multiplies to it).  The caller compacts the flagged keys from the index-ordered key array (they come out in index order: the order the segment
// machinery wants inside equal keys) and sorts those few thousand.  Anything that does not fit — a bucket too long (operands full of repeated
// rows), a counter that saturates, a list that overflows — raises `giveup` and the caller takes the sorted path; nothing is decided here that
// the 64-bit hashes do not decide there.
//
// cfg3 (10^4 terms squared, 4,096 product buckets of 12,208 pairs, 16 per workgroup): k_pd_bucket 22 us + k_pair_dups 236 us against the 0.67 ms
// above.  The kernel is bound by its LDS operations (per pair: 1.8 reads on the walk, a returning add, a counter read; per bucket and lane: the
// 64 KiB of counters cleared at ~32 B per cycle, tile sizes, two scans, the tiles filed): phase stamps (-DSYMGPU_PD_STAMPS + SYMGPU_PD_STAMPS=1,
// cycles per product bucket of one workgroup) clear 2,400 / tile sizes 1,900 / scans 1,300 / tiles filed 3,600 / walk 3,900 / adds 5,900 /
// list 5,500 / chains + match 5,800.  Measured and dropped: 512 lanes x 32 pairs (fewer wavefronts hide less: 29 -> 41 thousand cycles per
// bucket); `seen` / `dup` bitmaps instead of counters (half the clearing, but a conditional second atomic per pair: +6 %); bisection for a lane's
// first tile instead of the owner's note (+1,500 cycles); __shfl_up scans (ds_bpermute round trips) instead of DPP.
#include "common.h"

namespace symgpu {

typedef u32 u32x4 __attribute__((ext_vector_type(4)));
constexpr int PD_THREADS = 1024;
constexpr int PD_QUOTA = 16;                        // pairs per lane and product bucket: 16,384 pairs per bucket at most
constexpr int PD_TARGET = 12288;                    // pairs per product bucket the bucket width is chosen for
constexpr int PD_SLOT_BITS = 17;                    // 4-bit counters: 64 KiB (two 1-bit maps, `seen` and `dup`, were measured: 6 % slower)
constexpr int PD_CAND = 1792, PD_CHAIN = 1024;         // (cfg3 lists 1,140 +- 50 pairs per product bucket, 1,269 at most over its 4,096 buckets)
constexpr int PD_MIN_B = 8;
constexpr int PD_MAX_B = 13;                        // 8,192 buckets: eight per lane (two groups of four); up to 12: 4,096, one group
constexpr int PD_TARGET_16 = 9216;                  // pairs per product bucket with the 32 KiB counter table of the 8,192-bucket form
constexpr int PD_MAX_BUCKET = 255;                  // terms per operand bucket (longer: the hashes are not spread — repeated rows)
constexpr size_t PD_LDS_MAX = 160 * 1024 - 512;     // (the kernel's static LDS — wavefront sums, two counters — comes on top)

// inclusive prefix sum over the wavefront: four row_shr steps inside the rows of 16 lanes, then lane 15 of rows 0 and 2 into rows 1 and 3
// (row_bcast:15) and lane 31 into the upper half (row_bcast:31) — DPP modifiers, no LDS round trips (__shfl_up is a ds_bpermute each)
__device__ __forceinline__ u32 pd_wave_scan(u32 x) {
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
// the 4-bit counter of a word: index of its 32-bit word, shift of its field
// (sb = log2 of the counters: 17, 64 KiB — or 16 with 8,192 buckets, whose tile list and bucket starts need the other 32 KiB)
__device__ __forceinline__ u32 pd_cnt_word(u32 v, int sb) { return v >> (32 - sb + 3); }
__device__ __forceinline__ u32 pd_cnt_shift(u32 v, int sb) { return ((v >> (32 - sb)) << 2) & 28u; }

// one workgroup per operand: terms in bucket order (unordered inside a bucket) — hash word behind the bucket bits, full hash, term index
__global__ __launch_bounds__(1024) void k_pd_bucket(const u64 *__restrict__ hI, int nI, const u64 *__restrict__ hO, int nO, int B, u32 *__restrict__ tab_w,
                                                    u64 *__restrict__ tab_h, u32 *__restrict__ tab_idx, unsigned short *__restrict__ start, u32 *__restrict__ giveup) {
    __shared__ u32 s_cnt[(1 << PD_MAX_B) + 1];
    __shared__ u32 s_wsum[16];
    const int side = blockIdx.x, nb = 1 << B, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 *h = side ? hO : hI;
    const int n = side ? nO : nI, base = side ? nI : 0;
    // the first 16,384 hashes stay in registers between the count and the scatter (all loads in flight together); longer operands: read again
    constexpr int REG = 16;
    u64 hv[REG];
#pragma unroll
    for (int k = 0; k < REG; ++k) { const int i = tid + k * 1024; hv[k] = i < n ? h[i] : 0ULL; }
    for (int a = tid; a <= nb; a += 1024) s_cnt[a] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < REG; ++k) if (tid + k * 1024 < n) atomicAdd(&s_cnt[hv[k] >> (64 - B)], 1u);
    for (int i = tid + REG * 1024; i < n; i += 1024) atomicAdd(&s_cnt[h[i] >> (64 - B)], 1u);
    __syncthreads();
    const int per = (nb + 1023) / 1024;                                     // buckets per lane, consecutive
    u32 mine = 0, longest = 0;
    for (int k = 0; k < per; ++k) { const int a = tid * per + k; if (a < nb) { const u32 c = s_cnt[a]; mine += c; longest = c > longest ? c : longest; } }
    if (longest > (u32)PD_MAX_BUCKET) atomicOr(giveup, 8u);
    const u32 inc = pd_wave_scan(mine);
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
    auto place = [&](int i, u64 hi) {
        const u32 p = atomicAdd(&s_cnt[hi >> (64 - B)], 1u);
        tab_w[base + p] = (u32)((hi << B) >> 32); tab_h[base + p] = hi; tab_idx[base + p] = (u32)i;
    };
#pragma unroll
    for (int k = 0; k < REG; ++k) if (tid + k * 1024 < n) place(tid + k * 1024, hv[k]);
    for (int i = tid + REG * 1024; i < n; i += 1024) place(i, h[i]);
}

struct PairDupArgs {
    const u32 *tab_w; const u64 *tab_h; const u32 *tab_idx; const unsigned short *start;
    int nI, nO, B, squared;                          // squared: ONE table (nO = 0), pairs i >= o
    int sb;                                          // log2 of the 4-bit counters (17 or 16)
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

// LDS of k_pair_dups, in this order (pd_lds_bytes is the host's copy of the sum)
//   s_w      [nI + nO] u32   hash words, bucket order: inner table, then the outer one
//   s_cnt    64 KiB          4-bit counters, eight per word
//   s_tiles  [cap] 2 x u32   non-empty tiles: {first inner position | end << 16, first outer position | end << 16}
//   s_lw, s_lxy, s_next [PD_CAND] u32, s_head [PD_CHAIN] u32      listed pairs: word, positions, chain link; chain heads
//   s_lane   [1024] u32      where a lane's first pair lies: tile << 16 | pair number inside the tile
//   s_sI, s_sO [nb + 4] u16  bucket starts of the two tables (P * P: one)
__host__ __device__ inline size_t pd_tile_cap(int nb, int squared) { return squared ? (size_t)nb / 2 + 4 : (size_t)nb; }
__host__ __device__ inline size_t pd_lds_bytes(int nTab, int nb, int squared, int sb) {
    return (size_t)((nTab + 3) & ~3) * 4 + ((size_t)1 << (sb - 3)) * 4 + pd_tile_cap(nb, squared) * 8 + (size_t)PD_CAND * 12 + (size_t)PD_CHAIN * 4 +
           (size_t)PD_THREADS * 4 + (size_t)(squared ? 1 : 2) * (nb + 4) * 2;
}

#ifdef SYMGPU_PD_STAMPS
__device__ u64 g_pd_stamps[12];
#define PD_STAMP(i) do { const u64 tn_ = __builtin_readcyclecounter(); acc_[i] += tn_ - ts_; ts_ = tn_; } while (0)
#else
#define PD_STAMP(i) do { } while (0)
#endif
__global__ __launch_bounds__(PD_THREADS) void k_pair_dups(const PairDupArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NT = PD_THREADS;
    const int nb = 1 << a.B, nTab = a.nI + a.nO;
    u32 *s_w = reinterpret_cast<u32 *>(smem);
    u32 *s_cnt = s_w + ((nTab + 3) & ~3);
    const int cbits = a.sb;                                                    // log2 of the counters
    uint2 *s_tiles = reinterpret_cast<uint2 *>(s_cnt + (1 << (cbits - 3)));
    u32 *s_lw = reinterpret_cast<u32 *>(s_tiles + pd_tile_cap(nb, a.squared));
    u32 *s_lxy = s_lw + PD_CAND;
    u32 *s_next = s_lxy + PD_CAND;
    u32 *s_head = s_next + PD_CAND;
    u32 *s_lane = s_head + PD_CHAIN;
    unsigned short *s_sI = reinterpret_cast<unsigned short *>(s_lane + NT);
    unsigned short *s_sO = a.squared ? s_sI : s_sI + (nb + 4);
    __shared__ __attribute__((aligned(16))) u32 s_wsum[64];                     // inclusive sums per wavefront: pairs, tiles (and the same of the second groups)
    __shared__ u32 s_nc, s_over;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (*a.giveup & 8u) return;                                                 // a bucket longer than the positions below can address: not for this path
    const u32 *s_wO = a.squared ? s_w : s_w + a.nI;
    for (int i = tid; i < nTab; i += NT) s_w[i] = a.tab_w[i];
    for (int i = tid; i < nb + 4; i += NT) {                                    // (padded with the table's length: the groups of four below read one start past theirs)
        s_sI[i] = i <= nb ? a.start[i] : (unsigned short)a.nI;
        if (!a.squared) s_sO[i] = i <= nb ? a.start[nb + 1 + i] : (unsigned short)a.nO;
    }
    __syncthreads();
    // P * P: the diagonal (the identity, N pairs) is flagged as it is
    if (a.squared)
        for (int x = blockIdx.x * NT + tid; x < a.nI; x += gridDim.x * NT) pd_flag(a, (u32)x, (u32)x);
#ifdef SYMGPU_PD_STAMPS
    u64 acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ts_ = __builtin_readcyclecounter();
#endif
    for (int beta = blockIdx.x; beta < nb; beta += gridDim.x) {
        PD_STAMP(5);
        {
            const u32x4 z = {0u, 0u, 0u, 0u};
            for (int i = tid; i < (1 << (cbits - 5)); i += NT) reinterpret_cast<u32x4 *>(s_cnt)[i] = z;
        }
        for (int i = tid; i < PD_CHAIN; i += NT) s_head[i] = 0xFFFFFFFFu;
        s_lane[tid] = 0u;
        if (tid == 0) { s_nc = 0; s_over = 0; }
        PD_STAMP(6);
        const bool inside = a.squared && beta == 0;                             // pairs inside one bucket: no tiles, a loop per bucket
        u32 hw[PD_QUOTA], xy[PD_QUOTA];
        u32 rem = 0, sat = 0;
        if (!inside) {
            // ---- tiles of this product bucket: sizes (a lane: groups of four consecutive buckets a — their partners a ^ beta are a group of
            //      four as well), scan, list of the non-empty ones with the lanes whose first pair lies in them
            auto group = [&](int a0, u32 (&tsz)[4], u32 (&tax)[4], u32 (&tby)[4]) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { tsz[j] = 0; tax[j] = 0; tby[j] = 0; }
                if (a0 < nb) {
                    const int b0 = a0 ^ (beta & ~3), bj = beta & 3;
                    const uint2 ia = *reinterpret_cast<const uint2 *>(s_sI + a0), ib = *reinterpret_cast<const uint2 *>(s_sO + b0);
                    const u32 sa[5] = {ia.x & 0xFFFFu, ia.x >> 16, ia.y & 0xFFFFu, ia.y >> 16, (u32)s_sI[a0 + 4]};
                    const u32 sbv[5] = {ib.x & 0xFFFFu, ib.x >> 16, ib.y & 0xFFFFu, ib.y >> 16, (u32)s_sO[b0 + 4]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int jb = j ^ bj;
                        u32 lo_b = sbv[0], hi_b = sbv[1];
                        if (jb == 1) { lo_b = sbv[1]; hi_b = sbv[2]; } else if (jb == 2) { lo_b = sbv[2]; hi_b = sbv[3]; } else if (jb == 3) { lo_b = sbv[3]; hi_b = sbv[4]; }
                        tsz[j] = (!a.squared || (a0 + j) < (b0 + jb)) ? (sa[j + 1] - sa[j]) * (hi_b - lo_b) : 0u;
                        tax[j] = sa[j] | (sa[j + 1] << 16); tby[j] = lo_b | (hi_b << 16);
                    }
                }
            };
            // (the lane's first four buckets stay in registers across the scan; with 8,192 buckets its second four, a = 4,096 + 4 tid ..., are
            // formed again behind it)
            const bool two_groups = nb > 4 * NT;
            u32 tsz[4], tax[4], tby[4];
            group(4 * tid, tsz, tax, tby);
            PD_STAMP(7);
            u32 mysum = 0, mytiles = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { mysum += tsz[j]; mytiles += tsz[j] ? 1u : 0u; }
            // tile order = pair order: all first groups (a < 4,096) come before all second groups, so the two are scanned one after the other
            u32 mysum2 = 0, mytiles2 = 0;
            if (two_groups) {
                u32 t2[4], x2[4], y2[4];
                group(4 * (tid + NT), t2, x2, y2);
#pragma unroll
                for (int j = 0; j < 4; ++j) { mysum2 += t2[j]; mytiles2 += t2[j] ? 1u : 0u; }
            }
            const u32 inc_s = pd_wave_scan(mysum), inc_t = pd_wave_scan(mytiles);     // (two sums: a tile holds up to 255 x 255 pairs)
            u32 inc_s2 = 0, inc_t2 = 0;
            if (two_groups) { inc_s2 = pd_wave_scan(mysum2); inc_t2 = pd_wave_scan(mytiles2); }
            if (lane == 63) {
                s_wsum[wave] = inc_s; s_wsum[16 + wave] = inc_t;
                if (two_groups) { s_wsum[32 + wave] = inc_s2; s_wsum[48 + wave] = inc_t2; }
            }
            __syncthreads();
            PD_STAMP(8);
            u32 base_s = 0, base_t = 0, P = 0, ntl = 0, base_s2 = 0, base_t2 = 0, P2 = 0, ntl2 = 0;
            {
                const u32x4 *ws4 = reinterpret_cast<const u32x4 *>(s_wsum);
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) {
                    const u32x4 vs = ws4[w4], vt = ws4[4 + w4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        P += vs[j]; ntl += vt[j];
                        base_s += (w4 * 4 + j) < wave ? vs[j] : 0u; base_t += (w4 * 4 + j) < wave ? vt[j] : 0u;
                    }
                }
                if (two_groups) {
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) {
                        const u32x4 vs = ws4[8 + w4], vt = ws4[12 + w4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            P2 += vs[j]; ntl2 += vt[j];
                            base_s2 += (w4 * 4 + j) < wave ? vs[j] : 0u; base_t2 += (w4 * 4 + j) < wave ? vt[j] : 0u;
                        }
                    }
                }
            }
            const u32 P1 = P, ntl1 = ntl;
            P += P2; ntl += ntl2;
            if (P > (u32)(PD_QUOTA * NT)) {                                     // (block-uniform)
                if (tid == 0) atomicOr(a.giveup, 1u);
                __syncthreads();
                continue;
            }
            if (P == 0) { __syncthreads(); continue; }
            const u32 q = (u32)__builtin_amdgcn_readfirstlane((int)((P + NT - 1) / NT));   // pairs per lane, block-uniform
            const u32 qinv = ((1u << 20) + q - 1) / q;                         // x / q = (x * qinv) >> 20 for x < 2^15, q <= 32
            {
                auto file = [&](u32 off, u32 tix, const u32 (&tsz_)[4], const u32 (&tax_)[4], const u32 (&tby_)[4]) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const u32 c = tsz_[j];
                        if (c) {
                            s_tiles[tix] = uint2{tax_[j], tby_[j]};
                            for (u32 L = ((off + q - 1) * qinv) >> 20; L * q < off + c; ++L) s_lane[L] = (tix << 16) | (L * q - off);
                            ++tix; off += c;
                        }
                    }
                };
                file(base_s + inc_s - mysum, base_t + inc_t - mytiles, tsz, tax, tby);
                if (two_groups) {
                    u32 t2[4], x2[4], y2[4];
                    group(4 * (tid + NT), t2, x2, y2);
                    file(P1 + base_s2 + inc_s2 - mysum2, ntl1 + base_t2 + inc_t2 - mytiles2, t2, x2, y2);
                }
            }
            __syncthreads();
            PD_STAMP(0);
            // ---- count: a lane takes q consecutive pairs.  First the words and positions of all of them (reads only), then the returning adds
            //      back to back: LDS answers come back in order, a read behind an add would wait for it
            const u32 p0 = tid * q;
            rem = p0 < P ? (P - p0 < q ? P - p0 : q) : 0u;
            {
                const u32 e = s_lane[tid];
                u32 t = e >> 16;
                uint2 d = s_tiles[t];
                u32 ax_end = d.x >> 16, sb = d.y & 0xFFFFu, ay_end = d.y >> 16;
                const u32 cb = ay_end - sb, r = e & 0xFFFFu;
                const u32 x0 = (u32)__fdividef((float)r + 0.5f, (float)cb);
                u32 ax = (d.x & 0xFFFFu) + x0, ay = sb + (r - x0 * cb);
                u32 wI = s_w[ax];
#pragma unroll
                for (int k = 0; k < PD_QUOTA; ++k) {
                    if ((u32)k < q) {                                            // (scalar)
                        hw[k] = wI ^ s_wO[ay]; xy[k] = ax | (ay << 16);
                        ++ay;
                        if (ay == ay_end) {
                            ++ax;
                            if (ax == ax_end) {
                                t = t + 1 < ntl ? t + 1 : t;                      // (behind the last tile the lane has no pairs left: it walks the last tile again, adding zeros)
                                d = s_tiles[t];
                                ax = d.x & 0xFFFFu; ax_end = d.x >> 16; sb = d.y & 0xFFFFu; ay_end = d.y >> 16;
                            }
                            ay = sb;
                            wI = s_w[ax];
                        }
                    }
                }
                PD_STAMP(9);
#pragma unroll
                for (int h = 0; h < 2; ++h) {                                   // eight returning adds in flight
                    constexpr int HQ = PD_QUOTA / 2;
                    u32 old[HQ];
#pragma unroll
                    for (int k = 0; k < HQ; ++k)
                        if ((u32)(h * HQ + k) < q) old[k] = atomicAdd(&s_cnt[pd_cnt_word(hw[h * HQ + k], cbits)], (u32)(h * HQ + k) < rem ? 1u << pd_cnt_shift(hw[h * HQ + k], cbits) : 0u);
#pragma unroll
                    for (int k = 0; k < HQ; ++k)
                        if ((u32)(h * HQ + k) < q) sat |= ((old[k] >> pd_cnt_shift(hw[h * HQ + k], cbits)) & 15u) == 15u ? 1u : 0u;
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
                        const u32 o = (atomicAdd(&s_cnt[pd_cnt_word(v, cbits)], 1u << pd_cnt_shift(v, cbits)) >> pd_cnt_shift(v, cbits)) & 15u;
                        sat |= o == 15u ? 1u : 0u;
                    }
            }
        }
        if (sat) s_over = 1;                                                   // (a counter at 15: the next hit would wrap it)
        __syncthreads();
        PD_STAMP(1);
        if (s_over) {                                                           // rows repeated all over
            if (tid == 0) atomicOr(a.giveup, 2u);
            __syncthreads();
            continue;
        }
        // ---- list: pairs whose counter reads >= 2 now (4 % of them), and zero words whatever their count (the identity is decided on the list)
        if (!inside) {
            u32 lm = 0;
#pragma unroll
            for (int k = 0; k < PD_QUOTA; ++k)
                if ((u32)k < rem && (hw[k] == 0u || ((s_cnt[pd_cnt_word(hw[k], cbits)] >> pd_cnt_shift(hw[k], cbits)) & 15u) >= 2u)) lm |= 1u << k;
            const u32 mine = (u32)__popc(lm);
            const u32 inc = pd_wave_scan(mine);
            const u32 total = (u32)__builtin_amdgcn_readlane((int)inc, 63);
            if (total) {                                                        // (wave-uniform)
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&s_nc, total);
                u32 n = (u32)__builtin_amdgcn_readfirstlane((int)base) + inc - mine;
#pragma unroll
                for (int k = 0; k < PD_QUOTA; ++k)
                    if ((lm >> k) & 1u) { if (n < (u32)PD_CAND) { s_lw[n] = hw[k]; s_lxy[n] = xy[k]; } ++n; }
            }
        } else {
            for (int av = tid; av < nb; av += NT) {
                const u32 s0 = s_sI[av], s1 = s_sI[av + 1];
                for (u32 x = s0 + 1; x < s1; ++x)
                    for (u32 y = s0; y < x; ++y) {
                        const u32 v = s_w[x] ^ s_w[y];
                        if (v == 0u || ((s_cnt[pd_cnt_word(v, cbits)] >> pd_cnt_shift(v, cbits)) & 15u) >= 2u) {
                            const u32 n = atomicAdd(&s_nc, 1u);
                            if (n < (u32)PD_CAND) { s_lw[n] = v; s_lxy[n] = x | (y << 16); }
                        }
                    }
            }
        }
        __syncthreads();
        PD_STAMP(2);
        const u32 nc = s_nc;
        if (nc > (u32)PD_CAND) {
            if (tid == 0) atomicOr(a.giveup, 4u);
            __syncthreads();
            continue;
        }
        // ---- match: chains by word, every listed pair walks its chain; an equal word is decided by the 64-bit hashes
        for (u32 c = tid; c < nc; c += NT) {
            const u32 hsh = (s_lw[c] * 2654435761u) >> 22;
            s_next[c] = atomicExch(&s_head[hsh], c);
        }
        __syncthreads();
        PD_STAMP(11);
        for (u32 c = tid; c < nc; c += NT) {
            const u32 wc = s_lw[c], hsh = (wc * 2654435761u) >> 22, xyc = s_lxy[c];
            u64 Hc = 0;
            bool have = false;
            if (wc == 0u) { Hc = pd_hash(a, xyc); have = true; }
            bool hit = have && Hc == 0ULL;                                      // the identity
            u32 steps = 0;                                                     // (a chain holds at most nc pairs: the walk ends whatever the links say)
            for (u32 j = hit ? 0xFFFFFFFFu : s_head[hsh]; j < nc && steps < nc; j = s_next[j], ++steps) {
                if (j != c && s_lw[j] == wc) {
                    if (!have) { Hc = pd_hash(a, xyc); have = true; }
                    if (pd_hash(a, s_lxy[j]) == Hc) { hit = true; break; }
                }
            }
            if (hit) pd_flag(a, xyc & 0xFFFFu, xyc >> 16);
        }
        __syncthreads();
        PD_STAMP(3);
    }
#ifdef SYMGPU_PD_STAMPS
    if (tid == 0 && blockIdx.x == 7) for (int i = 0; i < 12; ++i) g_pd_stamps[i] = acc_[i];
#endif
}
static_assert(PD_CHAIN == 1024, "the chain hash keeps 10 bits");

// The flag bitmap `flags` (one bit per pair, indexed like the key array in index order; zeroed by the caller) gets the bit of every pair
// whose 64-bit key equals another pair's, and of every pair whose key is zero.  *applies = false: the operands do not fit this path (nothing
// was launched).  giveup (device word, zeroed by the caller) != 0 afterwards: the flags are incomplete, take the sorted path.
// pair_dups_fits: does this path take the product?  (host-side: sizes only; *B_out: the bucket width)
bool pair_dups_fits(i64 Ni, i64 No, bool squared, i64 Tk, int *B_out, int *sb_out) {
    if (getenv("SYMGPU_CLEANUP_DIRECT") && getenv("SYMGPU_CLEANUP_DIRECT")[0] == '0') return false;
    const i64 nI = Ni, nO = squared ? 0 : No;
    if (nI + nO > 65535 || nI < 2 || (!squared && nO < 1)) return false;
    int B = 2, sb = PD_SLOT_BITS;
    while (B < 12 && (Tk >> B) > PD_TARGET) ++B;
    if ((Tk >> B) > PD_TARGET) {
        // 8,192 buckets: their tile list and bucket starts take 24 KiB more, which the counters give up — fewer pairs per bucket for the smaller table
        B = 13; sb = 16;
        if ((Tk >> B) > PD_TARGET_16) return false;
    }
    if (B < PD_MIN_B) return false;                  // (below ~1.6e6 keys the sorted flag pass is as fast: P * P of 1,500 terms 0.29 against 0.30 ms)
    if (pd_lds_bytes((int)(nI + nO), 1 << B, squared ? 1 : 0, sb) > PD_LDS_MAX) return false;
    if (B_out) *B_out = B;
    if (sb_out) *sb_out = sb;
    return true;
}

int pair_dups_dev(const u64 *hI, i64 Ni, const u64 *hO, i64 No, bool squared, i64 Tk, u64 *flags, u32 *giveup, bool *applies) {
    *applies = false;
    int B = 0, sb = PD_SLOT_BITS;
    if (!pair_dups_fits(Ni, No, squared, Tk, &B, &sb)) return SYMGPU_OK;
    const i64 nI = Ni, nO = squared ? 0 : No;
    const int nb = 1 << B;
    const size_t lds = pd_lds_bytes((int)(nI + nO), nb, squared ? 1 : 0, sb);
    const bool attr = SG_DEVICE_ONCE(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pair_dups), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PD_LDS_MAX) == hipSuccess);
    if (!attr) { (void)hipGetLastError(); return SYMGPU_OK; }
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
    a.nI = (int)nI; a.nO = (int)nO; a.B = B; a.sb = sb; a.squared = squared ? 1 : 0; a.Ni = Ni; a.flags = flags; a.giveup = giveup;
    const int P = ctx().num_cu < nb ? ctx().num_cu : nb;
    hipLaunchKernelGGL(k_pair_dups, dim3((unsigned)P), dim3(PD_THREADS), lds, st, a);
    KERNEL_CHECK();
#ifdef SYMGPU_PD_STAMPS
    if (getenv("SYMGPU_PD_STAMPS")) {
        u64 h[12];
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pd_stamps), sizeof h));
        const double nbeta = (double)((nb - 7 + P - 1) / P);
        fprintf(stderr, "pair_dups workgroup 7, cycles per product bucket: tiles %.0f count %.0f list %.0f match %.0f (loop top %.0f) | clear %.0f sizes %.0f scan+sync %.0f | walk %.0f adds %.0f | chain %.0f\n", h[0] / nbeta, h[1] / nbeta, h[2] / nbeta,
                h[3] / nbeta, h[5] / nbeta, h[6] / nbeta, h[7] / nbeta, h[8] / nbeta, h[9] / nbeta, h[10] / nbeta, h[11] / nbeta);
    }
#endif
    *applies = true;
    return SYMGPU_OK;       // (`tab` goes back to the stream-ordered allocator: the kernel above is queued ahead of any reuse)
}

}  // namespace symgpu
