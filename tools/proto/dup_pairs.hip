// prototype (not part of the library): duplicate product keys of P * P without sorting the 5e7 keys.
// Terms bucketed by the top B bits of their row hash; the keys of product bucket beta are the tiles (a, a ^ beta): a persistent
// workgroup enumerates them from an LDS copy of the bucketed hash words, counts them into 2-bit LDS counters, lists the keys whose
// counter reached 2, matches the listed words through LDS chains and verifies on the 64-bit hashes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
typedef uint64_t u64; typedef uint32_t u32; typedef uint16_t u16; typedef int64_t i64;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NT = 1024, QUOTA = 16, SLOT_BITS = 17, CAND = 2048, CHAIN = 2048;
// 4-bit counters, eight per word
__device__ __forceinline__ u32 cnt_word(u32 v) { return v >> (32 - SLOT_BITS + 3); }
__device__ __forceinline__ u32 cnt_shift(u32 v) { return (v >> (32 - SLOT_BITS)) << 2 & 28u; }

__global__ __launch_bounds__(1024) void k_bucket(const u64 *h, int N, int B, u32 *tab_w, u64 *tab_h, u32 *tab_idx, u16 *start) {
    __shared__ u32 s_cnt[4097];
    const int nb = 1 << B;
    for (int a = threadIdx.x; a <= nb; a += 1024) s_cnt[a] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 1024) atomicAdd(&s_cnt[h[i] >> (64 - B)], 1u);
    __syncthreads();
    if (threadIdx.x == 0) { u32 run = 0; for (int a = 0; a < nb; ++a) { const u32 c = s_cnt[a]; s_cnt[a] = run; start[a] = (u16)run; run += c; } start[nb] = (u16)run; }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 1024) {
        const u64 hi = h[i];
        const u32 p = atomicAdd(&s_cnt[hi >> (64 - B)], 1u);
        tab_w[p] = (u32)((hi << B) >> 32); tab_h[p] = hi; tab_idx[p] = (u32)i;
    }
}

struct Args { const u32 *tab_w; const u64 *tab_h; const u32 *tab_idx; const u16 *start; int N, B; u64 *out; u32 *out_n; u32 *giveup; u32 *stats; };

__global__ __launch_bounds__(NT) void k_dup_pairs(const Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int nb = 1 << a.B, N = a.N;
    u32 *s_w = reinterpret_cast<u32 *>(smem);                                  // [N]
    u32 *s_cnt2 = s_w + ((N + 3) & ~3);                                        // [2^SLOT_BITS / 16]
    u32 *s_tiles = s_cnt2 + (1 << (SLOT_BITS - 3));                            // [nb]: offset << 12 | a
    u32 *s_lw = s_tiles + nb;                                                  // [CAND]
    u32 *s_lxy = s_lw + CAND;                                                  // [CAND]
    u32 *s_next = s_lxy + CAND;                                                // [CAND]
    u32 *s_head = s_next + CAND;                                               // [CHAIN]
    u16 *s_start = reinterpret_cast<u16 *>(s_head + CHAIN);                    // [nb + 1]
    __shared__ u32 s_wsum[16], s_nc, s_total, s_ntiles, s_over;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < N; i += NT) s_w[i] = a.tab_w[i];
    for (int i = tid; i <= nb; i += NT) s_start[i] = a.start[i];
    __syncthreads();
    u64 acc[5] = {0, 0, 0, 0, 0};
    for (int beta = blockIdx.x; beta < nb; beta += gridDim.x) {
        u64 ts = __builtin_readcyclecounter();
#define STAMP(i) do { const u64 tn = __builtin_readcyclecounter(); acc[i] += tn - ts; ts = tn; } while (0)
        for (int i = tid; i < (1 << (SLOT_BITS - 3)); i += NT) s_cnt2[i] = 0;
        for (int i = tid; i < CHAIN; i += NT) s_head[i] = 0xFFFFFFFFu;
        if (tid == 0) { s_nc = 0; s_over = 0; }
        // ---- tiles of this bucket: counts, scan, compact list
        const int per = (nb + NT - 1) / NT;                                     // buckets a per lane, consecutive
        u32 mysum = 0, mytiles = 0;
        for (int k = 0; k < per; ++k) {
            const int av = tid * per + k;
            if (av < nb) {
                const int b = av ^ beta;
                const u32 ca = s_start[av + 1] - s_start[av], cb = s_start[b + 1] - s_start[b];
                const u32 c = av < b ? ca * cb : 0u;
                mysum += c; mytiles += c ? 1u : 0u;
            }
        }
        u32 pk = mysum | (mytiles << 20), inc = pk;
        for (int off = 1; off < 64; off <<= 1) { const u32 t = (u32)__shfl_up((int)inc, off); if (lane >= off) inc += t; }
        if (lane == 63) s_wsum[wave] = inc;
        __syncthreads();
        u32 wbase = 0;
        for (int w2 = 0; w2 < wave; ++w2) wbase += s_wsum[w2];
        if (tid == NT - 1) { const u32 tot = wbase + inc; s_total = tot & 0xFFFFFu; s_ntiles = tot >> 20; }
        u32 excl = wbase + inc - pk;
        {
            u32 off = excl & 0xFFFFFu, tix = excl >> 20;
            for (int k = 0; k < per; ++k) {
                const int av = tid * per + k;
                if (av < nb) {
                    const int b = av ^ beta;
                    const u32 ca = s_start[av + 1] - s_start[av], cb = s_start[b + 1] - s_start[b];
                    const u32 c = av < b ? ca * cb : 0u;
                    if (c) { s_tiles[tix++] = (off << 12) | (u32)av; off += c; }
                }
            }
        }
        __syncthreads();
        const u32 P = s_total, ntl = s_ntiles;
        STAMP(0);
        if (P > (u32)(QUOTA * NT)) { if (tid == 0) atomicOr(a.giveup, 1u); __syncthreads(); continue; }
        // ---- walk: lane takes q consecutive pairs; words, positions and counter answers stay in registers
        const u32 q = (P + NT - 1) / NT;                                        // block-uniform
        const u32 p0 = tid * q;
        u32 hw[QUOTA], xy[QUOTA], old[QUOTA];
        const u32 rem = p0 < P ? (P - p0 < q ? P - p0 : q) : 0u;                // pairs of this lane
        {
            int lo = 0, hi = (int)ntl - 1;                                      // last tile with offset <= p0
            const u32 ps = p0 < P ? p0 : 0u;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if ((s_tiles[mid] >> 12) <= ps) lo = mid; else hi = mid - 1; }
            u32 t = (u32)lo;
            u32 td = s_tiles[t];
            u32 av = td & 0xFFF, b = av ^ (u32)beta;
            u32 sa = s_start[av], sb = s_start[b];
            const u32 cb = s_start[b + 1] - sb;
            const u32 r = ps - (td >> 12);
            const u32 x0 = r / cb, y0 = r - x0 * cb;
            u32 ax = sa + x0, ax_end = s_start[av + 1], ay = sb + y0, ay_end = sb + cb;
            u32 wI = s_w[ax];
#pragma unroll
            for (int k = 0; k < QUOTA; ++k) {
                if ((u32)k < q) {                                                // (scalar)
                    const u32 v = wI ^ s_w[ay];
                    hw[k] = v; xy[k] = ax | (ay << 16);
                    const u32 inc = (u32)k < rem ? 1u << cnt_shift(v) : 0u;
                    old[k] = atomicAdd(&s_cnt2[cnt_word(v)], inc);
                    ++ay;
                    if (ay == ay_end) {
                        ++ax;
                        if (ax == ax_end) {
                            t = t + 1 < ntl ? t + 1 : t;                          // (past the last tile: the lane has no pairs left, it walks the last tile again)
                            td = s_tiles[t]; av = td & 0xFFF; b = av ^ (u32)beta; ax = s_start[av]; ax_end = s_start[av + 1]; sb = s_start[b]; ay_end = s_start[b + 1];
                        }
                        ay = sb;
                        wI = s_w[ax];
                    }
                }
            }
        }
        u32 live = 0, later = 0;
#pragma unroll
        for (int k = 0; k < QUOTA; ++k) {
            if ((u32)k < rem) {
                const u32 v = hw[k];
                live |= 1u << k;
                const u32 o = (old[k] >> cnt_shift(v)) & 15u;
                if (o == 15u) s_over = 1;
                if (o || v == 0u) later |= 1u << k;                              // (a zero word: listed whatever its count — the identity is decided on the list)
            }
        }
        __syncthreads();
        STAMP(1);
        if (!s_over) {
#pragma unroll
            for (int k = 0; k < QUOTA; ++k) {
                bool li = false;
                if ((live >> k) & 1u) li = ((later >> k) & 1u) || ((s_cnt2[cnt_word(hw[k])] >> cnt_shift(hw[k])) & 15u) >= 2u;
                const u64 m = __ballot(li);
                if (m) {
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(&s_nc, (u32)__popcll(m));
                    base = (u32)__builtin_amdgcn_readfirstlane((int)base);
                    const u32 n = base + (u32)__popcll(m & ((1ULL << lane) - 1ULL));
                    if (li && n < (u32)CAND) { s_lw[n] = hw[k]; s_lxy[n] = xy[k]; }
                }
            }
        }
        __syncthreads();
        STAMP(2);
        if (s_over) { if (tid == 0) atomicOr(a.giveup, 2u); __syncthreads(); continue; }
        __syncthreads();
        const u32 nc = s_nc;
        if (tid == 0 && a.stats) { atomicAdd(&a.stats[0], nc); atomicMax(&a.stats[1], nc); atomicMax(&a.stats[2], P); }
        if (nc > (u32)CAND) { if (tid == 0) atomicOr(a.giveup, 4u); __syncthreads(); continue; }
        // ---- chains by word, then every listed key walks its chain
        for (u32 c = tid; c < nc; c += NT) {
            const u32 hsh = (s_lw[c] * 2654435761u) >> (32 - 11);
            s_next[c] = atomicExch(&s_head[hsh], c);
        }
        __syncthreads();
        for (u32 c = tid; c < nc; c += NT) {
            const u32 wc = s_lw[c], hsh = (wc * 2654435761u) >> (32 - 11);
            u64 Hc = 0; bool have = false;
            if (wc == 0u) { const u32 xyc = s_lxy[c]; Hc = a.tab_h[xyc & 0xFFFF] ^ a.tab_h[xyc >> 16]; have = true; }
            bool hit = have && Hc == 0ULL;                                      // the identity: flagged directly
            for (u32 j = hit ? 0xFFFFFFFFu : s_head[hsh]; j != 0xFFFFFFFFu; j = s_next[j]) {
                if (j != c && s_lw[j] == wc) {
                    if (!have) { const u32 xyc = s_lxy[c]; Hc = a.tab_h[xyc & 0xFFFF] ^ a.tab_h[xyc >> 16]; have = true; }
                    const u32 xyj = s_lxy[j];
                    if ((a.tab_h[xyj & 0xFFFF] ^ a.tab_h[xyj >> 16]) == Hc) { hit = true; break; }
                }
            }
            if (hit) {
                const u32 xyc = s_lxy[c];
                const u32 n = atomicAdd(a.out_n, 1u);
                const u32 i0 = a.tab_idx[xyc & 0xFFFF], i1 = a.tab_idx[xyc >> 16];
                a.out[n] = ((u64)(i0 > i1 ? i0 : i1) << 32) | (i0 > i1 ? i1 : i0);
            }
        }
        __syncthreads();
        STAMP(3);
    }
    if (tid == 0 && blockIdx.x == 7 && a.stats) for (int i = 0; i < 4; ++i) a.stats[4 + i] = (u32)(acc[i] / 16);
}

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 10000;
    const int planted = argc > 2 ? atoi(argv[2]) : 100;
    const i64 Tk = (i64)N * (N + 1) / 2;
    int B = 4;
    while (B < 12 && (Tk >> B) > 12288) ++B;
    std::mt19937_64 rng(12345);
    std::vector<u64> h(N);
    for (auto &v : h) v = rng();
    for (int k = 0; k < planted; ++k) {
        int q[4];
        for (int j = 0; j < 4; ++j) q[j] = (int)(rng() % N);
        if (q[0] == q[1] || q[0] == q[2] || q[0] == q[3] || q[1] == q[2] || q[1] == q[3] || q[2] == q[3]) continue;
        h[q[3]] = h[q[0]] ^ h[q[1]] ^ h[q[2]];
    }
    u64 *d_h, *d_tab_h, *d_out; u32 *d_tab_w, *d_tab_idx, *d_misc; u16 *d_start;
    CK(hipMalloc(&d_h, N * 8)); CK(hipMalloc(&d_tab_h, N * 8)); CK(hipMalloc(&d_tab_w, N * 4)); CK(hipMalloc(&d_tab_idx, N * 4));
    CK(hipMalloc(&d_start, (4097) * 2)); CK(hipMalloc(&d_out, (size_t)(4 << 20) * 8)); CK(hipMalloc(&d_misc, 64));
    CK(hipMemcpy(d_h, h.data(), N * 8, hipMemcpyHostToDevice));
    const int nb = 1 << B;
    const size_t lds = (size_t)((N + 3) & ~3) * 4 + (size_t)(1 << (SLOT_BITS - 3)) * 4 + (size_t)nb * 4 + CAND * 12 + CHAIN * 4 + (nb + 1) * 2 + 16;
    printf("N=%d Tk=%lld B=%d keys/bucket=%.0f LDS=%zu\n", N, (long long)Tk, B, (double)Tk / nb, lds);
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dup_pairs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Args a{d_tab_w, d_tab_h, d_tab_idx, d_start, N, B, d_out, d_misc, d_misc + 1, d_misc + 4};
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    int ncu = 256;
    { hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); ncu = pr.multiProcessorCount; }
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(d_misc, 0, 64));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_bucket, dim3(1), dim3(1024), 0, 0, d_h, N, B, d_tab_w, d_tab_h, d_tab_idx, d_start);
        CK(hipEventRecord(e1));
        hipLaunchKernelGGL(k_dup_pairs, dim3(ncu < nb ? ncu : nb), dim3(NT), lds, 0, a);
        CK(hipEventRecord(e2));
        CK(hipDeviceSynchronize());
        float t01, t12; CK(hipEventElapsedTime(&t01, e0, e1)); CK(hipEventElapsedTime(&t12, e1, e2));
        u32 misc[16]; CK(hipMemcpy(misc, d_misc, 64, hipMemcpyDeviceToHost));
        printf("rep %d: bucket %.1f us, pairs %.1f us, flagged %u, giveup %u, listed total %u max %u, max P %u; cycles per beta: scan %u count %u list %u chains %u\n", rep, t01 * 1e3, t12 * 1e3, misc[0], misc[1], misc[4], misc[5], misc[6], misc[8], misc[9], misc[10], misc[11]);
    }
    u32 misc[16]; CK(hipMemcpy(misc, d_misc, 64, hipMemcpyDeviceToHost));
    std::vector<u64> got(misc[0]);
    CK(hipMemcpy(got.data(), d_out, (size_t)misc[0] * 8, hipMemcpyDeviceToHost));
    std::sort(got.begin(), got.end());
    // brute force
    std::vector<std::pair<u64, u64>> all;
    all.reserve(Tk);
    for (int i = 0; i < N; ++i) for (int o = 0; o <= i; ++o) all.push_back({h[i] ^ h[o], ((u64)i << 32) | (u64)o});
    std::sort(all.begin(), all.end());
    std::vector<u64> want;
    for (size_t k = 0; k < all.size(); ++k) {
        const bool dup = (k > 0 && all[k - 1].first == all[k].first) || (k + 1 < all.size() && all[k + 1].first == all[k].first) || all[k].first == 0;
        if (dup) want.push_back(all[k].second);
    }
    std::sort(want.begin(), want.end());
    printf("expected %zu flagged, got %zu: %s\n", want.size(), got.size(), want == got ? "MATCH" : "MISMATCH");
    return want == got ? 0 : 1;
}
