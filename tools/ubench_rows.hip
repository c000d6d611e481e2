// micro-benchmark: the write pattern of k_mul_rows (product.hip) taken apart.  Output = 256 outer rows x (1e5 inner rows x 256 B).
//   0: one-shot fill of the same bytes (every thread one 16-byte nt store)           -> the ceiling
//   1: the kernel's pattern, stores only (block = 4 KiB chunk x 12 outer rows)       -> is the pattern the limit?
//   2: + the inner chunk loaded once                                                  3: + the outer chunk load per store (= k_mul_rows)
//   4: as 3, the 12 outer chunks loaded before the first store                       5: as 1 with plain (temporal) stores
//   6: as 1 with "sc1 nt" stores   7: as 1 with "sc0 sc1" stores
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef long long i64;
template <int MODE>
__global__ __launch_bounds__(256) void k(const u32x4 *__restrict__ inner, i64 n_chunks, const u32x4 *__restrict__ outer, int Wq, i64 o_count,
                                         u32x4 *__restrict__ out, int rto) {
    const i64 c0 = (i64)blockIdx.x * 256 + threadIdx.x;
    if (MODE == 0) {
        const i64 i = ((i64)blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x;
        if (i < n_chunks * o_count) { u32x4 x = (u32x4)((unsigned)i); __builtin_nontemporal_store(x, out + i); }
        return;
    }
    if (c0 >= n_chunks) return;
    u32x4 v = MODE >= 2 && MODE <= 4 ? inner[c0] : (u32x4)((unsigned)c0);
    const int wq = (int)(c0 % Wq);
    const i64 ob = (i64)blockIdx.y * rto, oe = ob + rto < o_count ? ob + rto : o_count;
    if (MODE == 4) {
        u32x4 r[12];
#pragma unroll
        for (int k2 = 0; k2 < 12; ++k2) r[k2] = (ob + k2 < oe) ? outer[(ob + k2) * Wq + wq] : (u32x4)(0u);
#pragma unroll
        for (int k2 = 0; k2 < 12; ++k2) if (ob + k2 < oe) __builtin_nontemporal_store(v ^ r[k2], out + (ob + k2) * n_chunks + c0);
        return;
    }
    for (i64 o = ob; o < oe; ++o) {
        u32x4 r = v;
        if (MODE == 3) r = v ^ outer[o * Wq + wq];
        else r.x ^= (unsigned)o;
        u32x4 *dst = out + o * n_chunks + c0;
        if (MODE == 5) *dst = r;
        else if (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(dst), "v"(r) : "memory");
        else if (MODE == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(r) : "memory");
        else __builtin_nontemporal_store(r, dst);
    }
}
// persistent variant: one workgroup per CU owns a contiguous segment of the inner operand (held in LDS) and writes that segment of
// every outer row in turn: all CUs write the same output row at (roughly) the same time, like a sequential fill.
template <bool LDSIN>
__global__ __launch_bounds__(1024) void k_persist(const u32x4 *__restrict__ inner, i64 n_chunks, const u32x4 *__restrict__ outer, int Wq, i64 o_count,
                                                   u32x4 *__restrict__ out) {
    extern __shared__ u32x4 seg[];
    const i64 per = (n_chunks + gridDim.x - 1) / gridDim.x;
    const i64 c_lo = (i64)blockIdx.x * per, c_hi = c_lo + per < n_chunks ? c_lo + per : n_chunks;
    if (LDSIN) {
        for (i64 c = c_lo + threadIdx.x; c < c_hi; c += 1024) seg[c - c_lo] = inner[c];
        __syncthreads();
    }
    for (i64 o = 0; o < o_count; ++o) {
        const u32x4 *orow = outer + o * Wq;
        u32x4 *dst = out + o * n_chunks;
        for (i64 c = c_lo + threadIdx.x; c < c_hi; c += 1024) {
            u32x4 v = LDSIN ? seg[c - c_lo] : (u32x4)((unsigned)c);
            if (LDSIN) v ^= orow[c % Wq]; else v.x ^= (unsigned)o;
            __builtin_nontemporal_store(v, dst + c);
        }
    }
}
template <bool LDSIN> void run_persist(const char *name, const u32x4 *in, const u32x4 *outer, u32x4 *out, int blocks) {
    const i64 Ni = 100000, No = 256; const int Wq = 16;
    const i64 n_chunks = Ni * Wq;
    const size_t lds = LDSIN ? (size_t)((n_chunks + blocks - 1) / blocks) * 16 : 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_persist<LDSIN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_persist<LDSIN>, dim3(blocks), dim3(1024), lds, 0, in, n_chunks, outer, Wq, No, out);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%-52s blocks=%4d lds=%6zu  %.3f ms  %.2f TB/s\n", name, blocks, lds, best, (double)n_chunks * No * 16 / (best * 1e-3) / 1e12);
}
// stores only, the block's `rto` rows are o = by + j * gridDim.y (interleaved row groups) instead of consecutive rows
__global__ __launch_bounds__(256) void k_strided(i64 n_chunks, i64 o_count, u32x4 *__restrict__ out, int rto) {
    const i64 c0 = (i64)blockIdx.x * 256 + threadIdx.x;
    if (c0 >= n_chunks) return;
    u32x4 v = (u32x4)((unsigned)c0);
    for (int j = 0; j < rto; ++j) {
        const i64 o = (i64)blockIdx.y + (i64)j * gridDim.y;
        if (o >= o_count) break;
        v.x ^= (unsigned)o;
        __builtin_nontemporal_store(v, out + o * n_chunks + c0);
    }
}
void run_strided(u32x4 *out, i64 Ni, int rto) {
    const i64 No = 256; const int Wq = 16;
    const i64 n_chunks = Ni * Wq;
    dim3 grid((unsigned)((n_chunks + 255) / 256), (unsigned)((No + rto - 1) / rto));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_strided, grid, dim3(256), 0, 0, n_chunks, No, out, rto);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("strided rows, stores only  Ni=%7lld rto=%3d  %.3f ms  %.2f TB/s\n", (long long)Ni, rto, best, (double)n_chunks * No * 16 / (best * 1e-3) / 1e12);
}
static int g_pad8 = 0;      // pad grid.x to a multiple of 8: workgroups go round-robin to the 8 XCDs by linear id, so chunk bx then
                            // always lands on XCD bx % 8 and each XCD's L2 only ever sees its own eighth of the inner operand
template <int MODE> void run(const char *name, const u32x4 *in, const u32x4 *outer, u32x4 *out, int rto = 12, i64 Ni = 100000) {
    const i64 No = 256; const int Wq = 16;
    const i64 n_chunks = Ni * Wq;
    unsigned gx = (unsigned)((n_chunks + 255) / 256);
    if (g_pad8) gx = (gx + 7) / 8 * 8;
    dim3 grid(gx, MODE == 0 ? (unsigned)No : (unsigned)((No + rto - 1) / rto));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, grid, dim3(256), 0, 0, in, n_chunks, outer, Wq, No, out, rto);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    printf("%-52s Ni=%7lld rto=%3d pad8=%d  %.3f ms  %.2f TB/s\n", name, (long long)Ni, rto, g_pad8, best, (double)n_chunks * No * 16 / (best * 1e-3) / 1e12);
}
int main() {
    u32x4 *in, *outer, *out;
    (void)hipMalloc(&in, 100000ull * 256); (void)hipMalloc(&outer, 256ull * 256); (void)hipMalloc(&out, 256ull * 102400 * 256);
    (void)hipMemset(in, 1, 100000ull * 256); (void)hipMemset(outer, 2, 256ull * 256);
    run<0>("0 one-shot fill", in, outer, out);
    run<1>("1 pattern, stores only", in, outer, out);
    run<2>("2 + inner chunk load", in, outer, out);
    run<3>("3 + outer chunk load per store (= k_mul_rows)", in, outer, out);
    run<4>("4 outer chunks loaded before the stores", in, outer, out);
    run<5>("5 pattern, plain stores", in, outer, out);
    run<6>("6 pattern, sc1 nt stores", in, outer, out);
    run<7>("7 pattern, sc0 sc1 stores", in, outer, out);
    for (int pad = 0; pad < 2; ++pad) { g_pad8 = pad; for (int rto : {1, 2, 3, 4, 6, 8, 12}) run<3>("3 full kernel", in, outer, out, rto); }
    g_pad8 = 1; for (i64 Ni : {65536LL, 98304LL}) for (int rto : {1, 2, 4, 12}) run<3>("3 full kernel", in, outer, out, rto, Ni);
    g_pad8 = 0;
    return 0;
    for (i64 Ni : {100000LL, 98304LL, 99999LL, 100352LL, 102400LL, 65536LL, 81920LL}) run<1>("1 pattern, stores only", in, outer, out, 12, Ni);
    for (i64 Ni : {100000LL, 98304LL}) for (int rto : {4, 12, 32}) run_strided(out, Ni, rto);
    return 0;
    run_persist<false>("8 persistent, stores only", in, outer, out, 256);
    run_persist<false>("8 persistent, stores only", in, outer, out, 512);
    run_persist<false>("8 persistent, stores only", in, outer, out, 1024);
    run_persist<true>("9 persistent, inner segment in LDS", in, outer, out, 256);
    run_persist<true>("9 persistent, inner segment in LDS", in, outer, out, 512);
    for (int rto : {1, 2, 3, 4, 6, 8, 16, 32, 64, 256}) run<1>("1 pattern, stores only", in, outer, out, rto);
    for (int rto : {1, 2, 4, 8, 32, 256}) run<3>("3 full kernel", in, outer, out, rto);
    return 0;
}
