// micro-benchmark: can the row stream of the all-pairs product (k_mul_rows, product.hip) carry the pair coefficients as well?
// n = 1000 qubits: a row is 16 chunks of 16 bytes = one DPP row of 16 lanes (lanes 0-7 hold X words, lanes 8-15 Z words), so
//   Y_out  = sum popc(out & row_ror:8(out))            (x & z of the product row, both halves of the row get the same words)
//   flip   = sum popc(inner & row_ror:8(outer))        (lanes 0-7: x_inner & z_outer, lanes 8-15: z_inner & x_outer)
// and three row_ror adds leave the sum over 8 consecutive lanes in lane 0 (X half) and lane 8 (Z half) of every row.
//   A: rows only (= k_mul_rows, one row per block, grid.x padded to 8)      B: rows + coefficients in one kernel
//   C: the separate coefficient pass is NOT modelled here (0.158 ms per slab in the library)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned int u32;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef long long i64;
typedef unsigned long long u64;

__device__ __forceinline__ u32 ror8(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); }
template <int N> __device__ __forceinline__ u32 rorN(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xf, 0xf, false); }

__device__ __forceinline__ void apply_phase(double re, double im, int e, double &ore, double &oim) {
    const bool swap = e & 1;
    double a = swap ? im : re, b = swap ? re : im;
    const bool neg_a = (e == 1) || (e == 2), neg_b = (e == 2) || (e == 3);
    ore = neg_a ? -a : a;
    oim = neg_b ? -b : b;
}
__device__ __forceinline__ void pair_coefficient(double ar, double ai, double br, double bi, int e, double &ore, double &oim) {
    const double re = __dsub_rn(__dmul_rn(ar, br), __dmul_rn(ai, bi));
    const double im = __dadd_rn(__dmul_rn(ar, bi), __dmul_rn(ai, br));
    apply_phase(re, im, e, ore, oim);
}

// MODE 0: rows only.  MODE 1: rows + coefficients (inner is the left factor).
// yi / yo: Y counts of the operand rows (mod 4 is enough); ci / co: coefficients.
template <int MODE>
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ inner, i64 n_chunks, const u32x4 *__restrict__ outer, i64 o_count,
                                         u32x4 *__restrict__ out, const int *__restrict__ yi, const int *__restrict__ yo,
                                         const f64x2 *__restrict__ ci, const f64x2 *__restrict__ co, f64x2 *__restrict__ outc, i64 Ni) {
    const i64 c0 = (i64)blockIdx.x * 256 + threadIdx.x;
    if ((i64)blockIdx.x * 256 >= n_chunks) return;
    const bool ok = c0 < n_chunks;
    const u32x4 v = ok ? inner[c0] : (u32x4)(0u);
    const int wq = threadIdx.x & 15;
    const i64 o = blockIdx.y;
    const u32x4 r = outer[o * 16 + wq];
    const u32x4 x = v ^ r;
    if (ok) __builtin_nontemporal_store(x, out + o * n_chunks + c0);
    if (MODE == 0) return;
    u32 cy = 0, cf = 0, s = 0;
    if (MODE != 3) {
    cy = __popc(x.x & ror8(x.x));
    cy += __popc(x.y & ror8(x.y));
    cy += __popc(x.z & ror8(x.z));
    cy += __popc(x.w & ror8(x.w));
    cf = __popc(v.x & ror8(r.x));
    cf += __popc(v.y & ror8(r.y));
    cf += __popc(v.z & ror8(r.z));
    cf += __popc(v.w & ror8(r.w));
    s = cy + 2u * cf;
    s += rorN<1>(s);
    s += rorN<2>(s);
    s += rorN<4>(s);
    }
    if (MODE == 3) s = (u32)o;
    if ((threadIdx.x & 15) == 7) {
        const i64 i = c0 >> 4;
        if (MODE == 2) { if (s == 0x12345u) outc[i] = ci[0]; return; }
        if (i < Ni) {
            const int e = (int)((3u * (u32)(yi[i] + yo[o]) + s) & 3u);
            const f64x2 a = ci[i], b = co[o];
            double re, im;
            pair_coefficient(a.x, a.y, b.x, b.y, e, re, im);
            const f64x2 w = {re, im};
            __builtin_nontemporal_store(w, outc + o * Ni + i);
        }
    }
}


// per-lane: s = Y_out + 2 * flip of the row, summed over lanes L-7..L of the 16-lane row (valid in lane 7: X half, lane 15: Z half)
__device__ __forceinline__ u32 row_phase_sum(const u32x4 v, const u32x4 r, const u32x4 x) {
    u32 cy = __popc(x.x & ror8(x.x));
    cy += __popc(x.y & ror8(x.y));
    cy += __popc(x.z & ror8(x.z));
    cy += __popc(x.w & ror8(x.w));
    u32 cf = __popc(v.x & ror8(r.x));
    cf += __popc(v.y & ror8(r.y));
    cf += __popc(v.z & ror8(r.z));
    cf += __popc(v.w & ror8(r.w));
    u32 s = cy + 2u * cf;
    s += rorN<1>(s);
    s += rorN<2>(s);
    s += rorN<4>(s);
    return s;
}

// E: 4 chunks per lane, every wave owns 4 KiB contiguous (16 rows) of the output row and writes their 16 coefficients itself
// (256 contiguous bytes from lanes 0-15, no barrier).  F: the block's 64 coefficients gathered through LDS, written by wave 0
// as ONE 1 KiB store.  G: rows only on E's mapping.
template <int MODE>
__global__ __launch_bounds__(256) void k4(const u32x4 *__restrict__ inner, i64 n_chunks, const u32x4 *__restrict__ outer, i64 o_count,
                                          u32x4 *__restrict__ out, const int *__restrict__ yi, const int *__restrict__ yo,
                                          const f64x2 *__restrict__ ci, const f64x2 *__restrict__ co, f64x2 *__restrict__ outc, i64 Ni) {
    __shared__ u32 se[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 cb = (i64)blockIdx.x * 1024;
    if (cb >= n_chunks) return;
    const i64 o = blockIdx.y;
    const u32x4 r = outer[o * 16 + (lane & 15)];
    u32 sk[4];
    u32x4 v[4];
    const bool full = cb + 1024 <= n_chunks;   // block-uniform
    i64 c[4];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        c[k2] = (MODE == 'F') ? cb + k2 * 256 + threadIdx.x : cb + wave * 256 + k2 * 64 + lane;
        v[k2] = inner[full || c[k2] < n_chunks ? c[k2] : n_chunks - 1];
    }
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        const u32x4 x = v[k2] ^ r;
        if (full) __builtin_nontemporal_store(x, out + o * n_chunks + c[k2]);
        else if (c[k2] < n_chunks) __builtin_nontemporal_store(x, out + o * n_chunks + c[k2]);
        if (MODE != 'G') sk[k2] = row_phase_sum(v[k2], r, x);
    }
    if (MODE == 'G') return;
    if (MODE == 'E') {
        // lane j < 16 takes row j = 4*k + g of the wave: s_k of lane 16*g + 7
        const int src = 16 * (lane & 3) + 7;
        const u32 t0 = __shfl(sk[0], src), t1 = __shfl(sk[1], src), t2 = __shfl(sk[2], src), t3 = __shfl(sk[3], src);
        const int kk = lane >> 2;
        const u32 s = kk == 0 ? t0 : kk == 1 ? t1 : kk == 2 ? t2 : t3;
        const i64 i = ((cb + wave * 256) >> 4) + lane;
        if (lane < 16 && i < Ni) {
            const int e = (int)((3u * (u32)(yi[i] + yo[o]) + s) & 3u);
            const f64x2 a = ci[i], b = co[o];
            double re, im;
            pair_coefficient(a.x, a.y, b.x, b.y, e, re, im);
            const f64x2 w = {re, im};
            __builtin_nontemporal_store(w, outc + o * Ni + i);
        }
    } else {
        if ((lane & 15) == 7) {
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) se[k2 * 16 + wave * 4 + (lane >> 4)] = sk[k2];
        }
        __syncthreads();
        if (wave != 0) return;
        const i64 i = (cb >> 4) + lane;
        if (i < Ni) {
            const int e = (int)((3u * (u32)(yi[i] + yo[o]) + se[lane]) & 3u);
            const f64x2 a = ci[i], b = co[o];
            double re, im;
            pair_coefficient(a.x, a.y, b.x, b.y, e, re, im);
            const f64x2 w = {re, im};
            __builtin_nontemporal_store(w, outc + o * Ni + i);
        }
    }
}
template <int MODE> float run4(const char *name, const u32x4 *in, const u32x4 *outer, u32x4 *out, const int *yi, const int *yo,
                               const f64x2 *ci, const f64x2 *co, f64x2 *outc, i64 Ni, i64 No) {
    const i64 n_chunks = Ni * 16;
    unsigned gx = (unsigned)((n_chunks + 1023) / 1024);
    gx = (gx + 7) / 8 * 8;
    dim3 grid(gx, (unsigned)No);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k4<MODE>, grid, dim3(256), 0, 0, in, n_chunks, outer, No, out, yi, yo, ci, co, outc, Ni);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)n_chunks * No * 16 + (MODE != 'G' ? (double)Ni * No * 16 : 0.0);
    printf("%-40s Ni=%7lld No=%4lld  best %.3f ms  avg %.3f ms  %.2f TB/s (avg, %s)\n", name, (long long)Ni, (long long)No, best, sum / 6,
           bytes / (sum / 6 * 1e-3) / 1e12, MODE != 'G' ? "272 B/pair" : "256 B/pair");
    return sum / 6;
}


// H: A's structure (one 16-byte chunk per lane, BT threads = BT/16 rows of ONE outer row per block) with the block's BT/16
// phase sums gathered through LDS and written by wave 0 as one contiguous store of BT bytes.
template <int BT, bool NTC>
__global__ __launch_bounds__(BT) void kh(const u32x4 *__restrict__ inner, i64 n_chunks, const u32x4 *__restrict__ outer, i64 o_count,
                                         u32x4 *__restrict__ out, const int *__restrict__ yi, const int *__restrict__ yo,
                                         const f64x2 *__restrict__ ci, const f64x2 *__restrict__ co, f64x2 *__restrict__ outc, i64 Ni) {
    __shared__ u32 se[BT / 16];
    const i64 cb = (i64)blockIdx.x * BT;
    if (cb >= n_chunks) return;
    const i64 c0 = cb + threadIdx.x;
    const bool ok = c0 < n_chunks;
    const u32x4 v = ok ? inner[c0] : (u32x4)(0u);
    const i64 o = blockIdx.y;
    const u32x4 r = outer[o * 16 + (threadIdx.x & 15)];
    const u32x4 x = v ^ r;
    if (ok) __builtin_nontemporal_store(x, out + o * n_chunks + c0);
    const u32 s = row_phase_sum(v, r, x);
    if ((threadIdx.x & 15) == 7) se[threadIdx.x >> 4] = s;
    __syncthreads();
    if (threadIdx.x >= BT / 16) return;
    const i64 i = (cb >> 4) + threadIdx.x;
    if (i < Ni) {
        const int e = (int)((3u * (u32)(yi[i] + yo[o]) + se[threadIdx.x]) & 3u);
        const f64x2 a = ci[i], b = co[o];
        double re, im;
        pair_coefficient(a.x, a.y, b.x, b.y, e, re, im);
        const f64x2 w = {re, im};
        if (NTC) __builtin_nontemporal_store(w, outc + o * Ni + i); else outc[o * Ni + i] = w;
    }
}
template <int BT, bool NTC> float runh(const char *name, const u32x4 *in, const u32x4 *outer, u32x4 *out, const int *yi, const int *yo,
                               const f64x2 *ci, const f64x2 *co, f64x2 *outc, i64 Ni, i64 No) {
    const i64 n_chunks = Ni * 16;
    unsigned gx = (unsigned)((n_chunks + BT - 1) / BT);
    gx = (gx + 7) / 8 * 8;
    dim3 grid(gx, (unsigned)No);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((kh<BT, NTC>), grid, dim3(BT), 0, 0, in, n_chunks, outer, No, out, yi, yo, ci, co, outc, Ni);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)n_chunks * No * 16 + (double)Ni * No * 16;
    printf("%-34s BT=%4d Ni=%7lld No=%4lld  best %.3f ms  avg %.3f ms  %.2f TB/s (avg, 272 B/pair)\n", name, BT, (long long)Ni, (long long)No, best, sum / 6,
           bytes / (sum / 6 * 1e-3) / 1e12);
    return sum / 6;
}


// J: A's structure; the 2-bit phase sums (Y_out + 2 flip) mod 4 of a wave's 4 rows leave as ONE byte per wave (plain store: the
// bytes of 32 consecutive blocks of one XCD fill a 128-byte line in that XCD's L2), K expands them to coefficients in a second,
// purely streaming kernel.  e-byte index: o * 4 gx + (bx % 8) * (gx / 8) * 4 + (bx / 8) * 4 + wave.
template <int V>
__global__ __launch_bounds__(256) void kj(const u32x4 *__restrict__ inner, i64 n_chunks, const u32x4 *__restrict__ outer,
                                          u32x4 *__restrict__ out, unsigned char *__restrict__ eb) {
    __shared__ unsigned char sb[4];
    const i64 cb = (i64)blockIdx.x * 256;
    if (cb >= n_chunks) return;
    const i64 c0 = cb + threadIdx.x;
    const bool ok = c0 < n_chunks;
    const u32x4 v = ok ? inner[c0] : (u32x4)(0u);
    const i64 o = blockIdx.y;
    const u32x4 r = outer[o * 16 + (threadIdx.x & 15)];
    const u32x4 x = v ^ r;
    if (ok) __builtin_nontemporal_store(x, out + o * n_chunks + c0);
    const u32 s = row_phase_sum(v, r, x);
    const u64 b0 = __ballot(s & 1u), b1 = __ballot(s & 2u);
    u32 byte = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) byte |= (u32)((b0 >> (16 * g + 7)) & 1ull) << (2 * g) | (u32)((b1 >> (16 * g + 7)) & 1ull) << (2 * g + 1);
    const int wave = threadIdx.x >> 6;
    const i64 gx = gridDim.x;
    const i64 idx = o * 4 * gx + (i64)(blockIdx.x & 7) * (gx >> 3) * 4 + (i64)(blockIdx.x >> 3) * 4;
    if (V == 0) {
        if ((threadIdx.x & 63) == 0) eb[idx + wave] = (unsigned char)byte;
    } else {
        if ((threadIdx.x & 63) == 0) sb[wave] = (unsigned char)byte;
        __syncthreads();
        if (threadIdx.x == 0) *reinterpret_cast<u32 *>(eb + idx) = *reinterpret_cast<const u32 *>(sb);
    }
}
// K: one lane per pair, 16-byte coalesced nt stores
template <int EOK> __global__ __launch_bounds__(256) void kk(const unsigned char *__restrict__ eb, i64 gx, const int *__restrict__ yi, const int *__restrict__ yo,
                                          const f64x2 *__restrict__ ci, const f64x2 *__restrict__ co, f64x2 *__restrict__ outc, i64 Ni) {
    const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= Ni) return;
    const i64 bx = i >> 4;
    const i64 ib = (bx & 7) * (gx >> 3) * 4 + (bx >> 3) * 4 + ((i >> 2) & 3);
    const f64x2 a = ci[i];
    const int y = yi[i];
    const int sh = 2 * (int)(i & 3);
#pragma unroll
    for (i64 o = (i64)blockIdx.y * EOK; o < (i64)blockIdx.y * EOK + EOK; ++o) {
        const u32 sv = eb[o * 4 * gx + ib] >> sh;
        const int e = (int)((3u * (u32)(y + yo[o]) + sv) & 3u);
        const f64x2 b = co[o];
        double re, im;
        pair_coefficient(a.x, a.y, b.x, b.y, e, re, im);
        const f64x2 w = {re, im};
        __builtin_nontemporal_store(w, outc + o * Ni + i);
    }
}
template <int V, int EOK = 16> float runj(const char *name, const u32x4 *in, const u32x4 *outer, u32x4 *out, const int *yi, const int *yo,
                               const f64x2 *ci, const f64x2 *co, f64x2 *outc, i64 Ni, i64 No, unsigned char *eb) {
    const i64 n_chunks = Ni * 16;
    unsigned gx = (unsigned)((n_chunks + 255) / 256);
    gx = (gx + 7) / 8 * 8;
    dim3 grid(gx, (unsigned)No);
    hipEvent_t e0, e1, e2; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventCreate(&e2);
    float sum = 0, sum2 = 0;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kj<V>, grid, dim3(256), 0, 0, in, n_chunks, outer, out, eb);
        (void)hipEventRecord(e1);
        hipLaunchKernelGGL(kk<EOK>, dim3((unsigned)(((Ni + 255) / 256 + 7) / 8 * 8), (unsigned)(No / EOK)), dim3(256), 0, 0, eb, (i64)gx, yi, yo, ci, co, outc, Ni);
        (void)hipEventRecord(e2); (void)hipEventSynchronize(e2);
        float ms, ms2; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipEventElapsedTime(&ms2, e1, e2);
        if (rep >= 2) { sum += ms; sum2 += ms2; }
    }
    const double bytes = (double)n_chunks * No * 16 + (double)Ni * No * 16;
    printf("%-34s Ni=%7lld No=%4lld  rows+e %.3f ms  expand %.3f ms  total %.3f ms  %.2f TB/s (avg, 272 B/pair)\n", name, (long long)Ni, (long long)No,
           sum / 6, sum2 / 6, (sum + sum2) / 6, bytes / ((sum + sum2) / 6 * 1e-3) / 1e12);
    return sum / 6;
}


// S: the cleanup's output stream (k_emit_stream, cleanup.hip) on the same buffers: 1-D grid, chunk f <- inner[(f >> 4) & mask] ^ outer[0]
template <int V>
__global__ __launch_bounds__(256) void ks(const u32x4 *__restrict__ inner, i64 n_chunks16, const u32x4 *__restrict__ outer, u32x4 *__restrict__ out, i64 mask) {
    const i64 f = (i64)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_chunks16) return;
    const i64 slot = f >> 4;
    const int c = (int)(f & 15);
    const i64 row = V == 0 ? (slot & mask) : slot % 99991;
    const u32x4 v = inner[row * 16 + c] ^ outer[c];
    __builtin_nontemporal_store(v, out + f);
}
template <int V> void runs(const char *name, const u32x4 *in, const u32x4 *outer, u32x4 *out, i64 n_chunks16, i64 mask) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(ks<V>, dim3((unsigned)((n_chunks16 + 255) / 256)), dim3(256), 0, 0, in, n_chunks16, outer, out, mask);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) sum += ms;
    }
    printf("%-50s chunks=%lld mask=%lld  avg %.3f ms  %.2f TB/s\n", name, (long long)n_chunks16, (long long)mask, sum / 6, (double)n_chunks16 * 16 / (sum / 6 * 1e-3) / 1e12);
}


// S2: as S, plus one 8-byte list entry per 256-byte output row read from a buffer of `meta_slots` entries (wrapping): is it the
// HBM READ inside a saturated write stream that halves k_emit_stream?  small buffer = L2 hits, 32 MB = Infinity Cache, 200 MB = HBM
__global__ __launch_bounds__(256) void ks2(const u32x4 *__restrict__ inner, i64 n_chunks16, const u32x4 *__restrict__ outer, u32x4 *__restrict__ out,
                                           const u64 *__restrict__ meta, i64 meta_slots) {
    const i64 f = (i64)blockIdx.x * 256 + threadIdx.x;
    if (f >= n_chunks16) return;
    const i64 slot = f >> 4;
    const int c = (int)(f & 15);
    const u64 m = meta[slot % meta_slots];
    const i64 row = (slot + (i64)(m & 1)) & 8191;
    const u32x4 v = inner[row * 16 + c] ^ outer[c];
    __builtin_nontemporal_store(v, out + f);
}
__global__ void k_fill_list(u64 *meta, i64 n, int nt) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (nt) __builtin_nontemporal_store((u64)(i * 2), meta + i); else meta[i] = (u64)(i * 2);
}
static int g_refill = 0;   // 1: the list is rewritten (plain stores) by a kernel right before every stream launch, 2: with nt stores
void runs2(const u32x4 *in, const u32x4 *outer, u32x4 *out, i64 n_chunks16, u64 *meta, i64 meta_slots) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
        if (g_refill) hipLaunchKernelGGL(k_fill_list, dim3((unsigned)((meta_slots + 255) / 256)), dim3(256), 0, 0, meta, meta_slots, g_refill == 2);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(ks2, dim3((unsigned)((n_chunks16 + 255) / 256)), dim3(256), 0, 0, in, n_chunks16, outer, out, meta, meta_slots);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) sum += ms;
    }
    printf("S2 (refill %d) stream + 8-byte list entry per row, list of %9lld entries (%7.1f MB)  avg %.3f ms  %.2f TB/s\n", g_refill, (long long)meta_slots, meta_slots * 8 / 1e6,
           sum / 6, (double)n_chunks16 * 16 / (sum / 6 * 1e-3) / 1e12);
}


// S3: k_emit_stream of cleanup.hip verbatim (RC chunks per lane, clamped index, nt list load)
struct u2 { u32 x, y; };
template <int RC, int VAR = 0>
__global__ __launch_bounds__(256) void ks3(const u2 *__restrict__ meta, i64 n_chunks16, int Wq, int wsh,
                                           const u32x4 *__restrict__ inner, const u32x4 *__restrict__ outer, u32x4 *__restrict__ out_rows) {
    const i64 f0 = (i64)blockIdx.x * (256 * RC) + threadIdx.x;
    const i64 last = n_chunks16 - 1;
    u2 m[RC];
    int c[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k) {
        const i64 f = (VAR & 2) ? f0 + 256 * k : (f0 + 256 * k < last ? f0 + 256 * k : last);
        const i64 slot = (VAR & 8) ? f >> 4 : (wsh >= 0 ? f >> wsh : f / Wq);
        c[k] = (int)(f - slot * Wq);
        const u64 mm = (VAR & 1) ? reinterpret_cast<const u64 *>(meta)[slot] : __builtin_nontemporal_load(reinterpret_cast<const u64 *>(meta) + slot);
        m[k].x = (u32)mm; m[k].y = (u32)(mm >> 32);
        if (VAR & 4) { m[k].x = (u32)(slot & 8191) + (m[k].x & 1u); m[k].y = m[k].y & 1u; }
    }
    u32x4 v[RC];
#pragma unroll
    for (int k = 0; k < RC; ++k) v[k] = inner[(i64)m[k].x * Wq + c[k]] ^ outer[(i64)m[k].y * Wq + c[k]];
    if ((i64)blockIdx.x * (256 * RC) + 256 * RC <= n_chunks16) {
#pragma unroll
        for (int k = 0; k < RC; ++k) __builtin_nontemporal_store(v[k], out_rows + f0 + 256 * k);
    } else {
#pragma unroll
        for (int k = 0; k < RC; ++k)
            if (f0 + 256 * k < n_chunks16) __builtin_nontemporal_store(v[k], out_rows + f0 + 256 * k);
    }
}
// list entries as the cleanup of a squared operator produces them: outer index o fixed over long runs, inner index ascending with gaps
__global__ void k_fill_tri(u64 *meta, i64 n, u32 N) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 o = (u32)(i / (N / 2)) % N, ii = (u32)((i % (N / 2)) * 2 + (i & 1)) % N;
    meta[i] = (u64)ii | ((u64)o << 32);
}
static u64 *g_pollute = nullptr; static i64 g_pollute_n = 0;   // a buffer rewritten between the list fill and the stream
template <int RC, int VAR = 0> void runs3(const char *name, const u32x4 *in, u32x4 *out, i64 n_slots, u64 *meta, int same_ops) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float sum = 0;
    const i64 n_chunks16 = n_slots * 16;
    for (int rep = 0; rep < 8; ++rep) {
        if (same_ops >= 2) hipLaunchKernelGGL(k_fill_tri, dim3((unsigned)((n_slots + 255) / 256)), dim3(256), 0, 0, meta, n_slots, 10000u);
        if (same_ops >= 3) hipLaunchKernelGGL(k_fill_list, dim3((unsigned)((g_pollute_n + 255) / 256)), dim3(256), 0, 0, g_pollute, g_pollute_n, 0);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((ks3<RC, VAR>), dim3((unsigned)((n_chunks16 + 256 * RC - 1) / (256 * RC))), dim3(256), 0, 0, (const u2 *)meta, n_chunks16, 16, 4, in, in, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) sum += ms;
    }
    printf("S3 var %2d %-40s RC=%d slots=%lld  avg %.3f ms  %.2f TB/s\n", VAR, name, RC, (long long)n_slots, sum / 6, (double)n_chunks16 * 16 / (sum / 6 * 1e-3) / 1e12);
}

// checker: one thread per pair
__global__ void k_check(const u64 *inner, const u64 *outer, i64 Ni, i64 No, const f64x2 *ci, const f64x2 *co, const f64x2 *outc,
                        const u64 *outrows, int *bad) {
    const i64 p = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= Ni * No) return;
    const i64 o = p / Ni, i = p % Ni;
    int yi = 0, yo = 0, yout = 0, fl = 0;
    bool rows_ok = true;
    for (int w = 0; w < 16; ++w) {
        const u64 xi = inner[i * 32 + w], zi = inner[i * 32 + 16 + w], xo = outer[o * 32 + w], zo = outer[o * 32 + 16 + w];
        yi += __popcll(xi & zi); yo += __popcll(xo & zo); yout += __popcll((xi ^ xo) & (zi ^ zo)); fl += __popcll(xi & zo);
        rows_ok &= outrows[p * 32 + w] == (xi ^ xo) && outrows[p * 32 + 16 + w] == (zi ^ zo);
    }
    const int e = (3 * (yi + yo) + yout + 2 * fl) & 3;
    double re, im;
    pair_coefficient(ci[i].x, ci[i].y, co[o].x, co[o].y, e, re, im);
    if (!rows_ok || re != outc[p].x || im != outc[p].y) atomicAdd(bad, 1);
}
__global__ void k_ycount(const u64 *rows, i64 T, int *y) {
    const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    int c = 0;
    for (int w = 0; w < 16; ++w) c += __popcll(rows[t * 32 + w] & rows[t * 32 + 16 + w]);
    y[t] = c;
}

template <int MODE> float run(const char *name, const u32x4 *in, const u32x4 *outer, u32x4 *out, const int *yi, const int *yo,
                              const f64x2 *ci, const f64x2 *co, f64x2 *outc, i64 Ni, i64 No) {
    const i64 n_chunks = Ni * 16;
    unsigned gx = (unsigned)((n_chunks + 255) / 256);
    gx = (gx + 7) / 8 * 8;
    dim3 grid(gx, (unsigned)No);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, in, n_chunks, outer, No, out, yi, yo, ci, co, outc, Ni);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)n_chunks * No * 16 + (MODE ? (double)Ni * No * 16 : 0.0);
    printf("%-40s Ni=%7lld No=%4lld  best %.3f ms  avg %.3f ms  %.2f TB/s (avg, %s)\n", name, (long long)Ni, (long long)No, best, sum / 6,
           bytes / (sum / 6 * 1e-3) / 1e12, MODE ? "272 B/pair" : "256 B/pair");
    return sum / 6;
}

int main(int argc, char **argv) {
    const i64 Ni = argc > 1 ? atoll(argv[1]) : 100000, No = 256;
    u64 *in, *outer; u32x4 *out; int *yi, *yo, *bad; f64x2 *ci, *co, *outc;
    (void)hipMalloc(&in, Ni * 256); (void)hipMalloc(&outer, No * 256); (void)hipMalloc(&out, (size_t)No * Ni * 256);
    (void)hipMalloc(&yi, Ni * 4); (void)hipMalloc(&yo, No * 4); (void)hipMalloc(&bad, 4);
    (void)hipMalloc(&ci, Ni * 16); (void)hipMalloc(&co, No * 16); (void)hipMalloc(&outc, (size_t)No * Ni * 16);
    std::vector<u64> h(Ni * 32), ho(No * 32); std::vector<double> hc(Ni * 2), hco(No * 2);
    u64 s = 88172645463325252ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto &w : h) w = rnd() & rnd();
    for (auto &w : ho) w = rnd() & rnd();
    for (auto &c : hc) c = (double)((i64)(rnd() % 17) - 8) / 16.0;
    for (auto &c : hco) c = (double)((i64)(rnd() % 17) - 8) / 16.0;
    (void)hipMemcpy(in, h.data(), Ni * 256, hipMemcpyHostToDevice); (void)hipMemcpy(outer, ho.data(), No * 256, hipMemcpyHostToDevice);
    (void)hipMemcpy(ci, hc.data(), Ni * 16, hipMemcpyHostToDevice); (void)hipMemcpy(co, hco.data(), No * 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_ycount, dim3((unsigned)((Ni + 255) / 256)), dim3(256), 0, 0, in, Ni, yi);
    hipLaunchKernelGGL(k_ycount, dim3((unsigned)((No + 255) / 256)), dim3(256), 0, 0, outer, No, yo);
    unsigned char *eb; (void)hipMalloc(&eb, (size_t)No * (Ni + 4096) / 4 + 4096);
    const u32x4 *pin = (const u32x4 *)in, *po = (const u32x4 *)outer;
    { u64 *meta; (void)hipMalloc(&meta, 26000000ull * 8); (void)hipMemset(meta, 0, 26000000ull * 8);
      for (int rf = 0; rf < 3; ++rf) { g_refill = rf; for (i64 ms : {1000000LL, 4000000LL, 25600000LL}) runs2(pin, po, out, Ni * 16 * No, meta, ms); } }
    { u64 *meta; (void)hipMalloc(&meta, 26000000ull * 8);
      const i64 ns = 24992058;
      hipLaunchKernelGGL(k_fill_list, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, 0, meta, ns, 0);
      // k_fill_list leaves x = 2 i (all rows of a 1e5-row operand... clamp to 8192 rows): rewrite as i & 8191
      hipLaunchKernelGGL(k_fill_tri, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, 0, meta, ns, 8192u);
      runs3<1>("tri-like list over 8192 rows", pin, out, ns, meta, 1);
      runs3<4>("tri-like list over 8192 rows", pin, out, ns, meta, 1);
      hipLaunchKernelGGL(k_fill_tri, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, 0, meta, ns, 10000u);
      runs3<1>("tri-like list over 10000 rows", pin, out, ns, meta, 1);
      runs3<4>("tri-like list over 10000 rows", pin, out, ns, meta, 1);
      runs3<1, 1>("plain load", pin, out, ns, meta, 1);
      runs3<1, 2>("no clamp", pin, out, ns, meta, 1);
      runs3<1, 4>("rows slot & 8191 + bit", pin, out, ns, meta, 1);
      runs3<1, 8>("constant shift", pin, out, ns, meta, 1);
      runs3<1, 15>("all four", pin, out, ns, meta, 1);
      runs3<1, 7>("plain, no clamp, rows slot", pin, out, ns, meta, 1);
      runs3<1, 6>("no clamp, rows slot", pin, out, ns, meta, 1);
      runs3<1, 1>("plain load, list refilled before", pin, out, ns, meta, 2);
      (void)hipMalloc(&g_pollute, 1ull << 30); g_pollute_n = (1ll << 30) / 8;
      runs3<1, 1>("plain load, refilled, then 1 GB written", pin, out, ns, meta, 3);
      g_pollute_n = (256ll << 20) / 8;
      runs3<1, 1>("plain load, refilled, then 256 MB written", pin, out, ns, meta, 3);
      g_pollute_n = (64ll << 20) / 8;
      runs3<1, 1>("plain load, refilled, then 64 MB written", pin, out, ns, meta, 3);
      for (i64 nsl : {250000LL, 1000000LL, 4000000LL, 8000000LL, 16000000LL}) { runs3<1, 0>("nt load, short list/output", pin, out, nsl, meta, 1); runs3<1, 1>("plain load, short list/output", pin, out, nsl, meta, 1); } }
    runs<0>("S 1-D stream, rows slot & 8191", pin, po, out, Ni * 16 * No, 8191);
    runs<0>("S 1-D stream, rows slot & 65535", pin, po, out, Ni * 16 * No, 65535);
    runs<1>("S 1-D stream, rows slot % 99991", pin, po, out, Ni * 16 * No, 0);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("A rows only", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        run<2>("C rows + phase exponents, no coeff store", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        run<3>("D rows + coeff store, no phase exponent", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        run<1>("B rows + coefficients fused", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        run4<'G'>("G rows only, 4 chunks/lane wave-contig", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        run4<'F'>("F fused, 1 KiB coeff store per block", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        runh<256, true>("H fused, LDS gather, nt coeff", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        runh<256, false>("H fused, LDS gather, plain coeff", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        runh<512, true>("H fused, LDS gather, nt coeff", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        runh<1024, true>("H fused, LDS gather, nt coeff", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        runj<1, 1>("J dword per block + K expand EO=1", pin, po, out, yi, yo, ci, co, outc, Ni, No, eb);
        runj<1, 2>("J dword per block + K expand EO=2", pin, po, out, yi, yo, ci, co, outc, Ni, No, eb);
        runj<1, 4>("J dword per block + K expand EO=4", pin, po, out, yi, yo, ci, co, outc, Ni, No, eb);
        runj<1, 8>("J dword per block + K expand EO=8", pin, po, out, yi, yo, ci, co, outc, Ni, No, eb);
        runj<1, 32>("J dword per block + K expand EO=32", pin, po, out, yi, yo, ci, co, outc, Ni, No, eb);
        runj<1>("J dword per block + K expand", pin, po, out, yi, yo, ci, co, outc, Ni, No, eb);
        run4<'E'>("E fused, 256 B coeff store per wave", pin, po, out, yi, yo, ci, co, outc, Ni, No);
        (void)hipMemset(outc, 0, (size_t)No * Ni * 16);
        runj<1>("J dword per block + K expand", pin, po, out, yi, yo, ci, co, outc, Ni, No, eb);
    }
    (void)hipMemset(bad, 0, 4);
    const i64 np = Ni * No;
    hipLaunchKernelGGL(k_check, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, 0, in, outer, Ni, No, ci, co, outc, (const u64 *)out, bad);
    int hb = -1; (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("check: %d mismatching pairs of %lld\n", hb, (long long)np);
    return hb != 0;
}
