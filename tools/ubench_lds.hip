// micro-benchmark: ds_read_b128 table look-ups as in commute_m4r.hip (4 row slots x 16 lanes, 256-byte entries, 64 KiB table),
// 8 waves per CU, one workgroup per CU (128 KiB of LDS), rolling window of LOOK reads in flight.
//   MODE 0: reads only (every result folded into ONE accumulator pair: 4 v_xor per read but no per-row registers)
//   MODE 1: reads + per-row accumulators (4 v_xor per read into acc[R][4]) + v_perm address (the real inner loop)
//   MODE 2: as 1 but the address comes from a precomputed VGPR (no v_perm)
//   MODE 3: ds_write_b64 table build only (16 entries per lane per iteration)
//   MODE 8: TWO 32 KiB tables (7-bit groups) read per row and folded with v_bitop3 XOR3: 2 ds_read_b128 + 2 v_perm + 4 v_bitop3 per row (3 VALU per read)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32;
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) u64x2 lds_u64x2;
constexpr int LDS = 128 * 1024;
template <int MODE, int R, int LOOK>
__global__ __launch_bounds__(512) void k(u64 *out, const u32 *idx_in, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < LDS / 8; i += 512) reinterpret_cast<u64 *>(lds)[i] = i * 0x9E3779B97F4A7C15ULL;
    __syncthreads();
    u32 idx[R / 4];
    for (int q = 0; q < R / 4; ++q) idx[q] = idx_in[(threadIdx.x >> 4) * (R / 4) + q];
    const u32 base = (lane & 15) * 16;
    u64 acc[R][2];
    for (int j = 0; j < R; ++j) acc[j][0] = acc[j][1] = 0;
    u64x2 one = {0, 0};
    if (MODE >= 4 && MODE != 8) {
        // pure store rate, values precomputed: MODE 4 = 16 x ds_write_b64 (entry stride 256 B), 5 = 8 x ds_write_b128 (16 lanes per entry),
        // 6 = 8 x ds_write_b128 with both halves... , 7 = 16 x b64 but only even waves write (half the waves)
        u64 e[16];
        for (int s = 0; s < 16; ++s) e[s] = lane * 0x9E37ULL + s;
        for (int it = 0; it < iters; ++it) {
            if (MODE == 4 || (MODE == 7 && ((threadIdx.x >> 6) & 1) == 0)) {
                uint8_t *dst = lds + (it & 1) * 65536 + ((threadIdx.x >> 5) * 16) * 256 + (lane & 31) * 8;
#pragma unroll
                for (int s = 0; s < 16; ++s) *reinterpret_cast<u64 *>(dst + s * 256) = e[s];
            } else if (MODE == 5) {
                // lane (quarter = lane / 16, wp = lane % 16) writes 16 B of 8 entries
                uint8_t *dst = lds + (it & 1) * 65536 + ((threadIdx.x >> 4) * 8) * 256 + (lane & 15) * 16;
#pragma unroll
                for (int s = 0; s < 8; ++s) { u64x2 v = {e[2 * s], e[2 * s + 1]}; *reinterpret_cast<u64x2 *>(dst + s * 256) = v; }
            }
            asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
            __syncthreads();
        }
        out[blockIdx.x * 512 + threadIdx.x] = e[0] + lds[threadIdx.x];
        return;
    }
    if (MODE == 3) {
        u64 e = lane;
        for (int it = 0; it < iters; ++it) {
            uint8_t *dst = lds + (it & 1) * 65536 + ((threadIdx.x >> 5) * 16) * 256 + (lane & 31) * 8;
#pragma unroll
            for (int s = 0; s < 16; ++s) { e ^= e << 1; *reinterpret_cast<u64 *>(dst + s * 256) = e; }
            __syncthreads();
        }
        out[blockIdx.x * 512 + threadIdx.x] = e + lds[threadIdx.x];
        return;
    }
    if (MODE == 8) {
        u32 idx2[R / 4];
        for (int q = 0; q < R / 4; ++q) idx2[q] = idx_in[4096 + (threadIdx.x >> 4) * (R / 4) + q] & 0x7f7f7f7fu;
        for (int q = 0; q < R / 4; ++q) idx[q] &= 0x7f7f7f7fu;
        for (int it = 0; it < iters; ++it) {
            const u32 b = base | ((it & 1) << 16);
            auto readA = [&](int j) -> u64x2 {
                const u32 addr = __builtin_amdgcn_perm(idx[j / 4], b, 0x0c020000u | ((4u + (j % 4)) << 8));
                return *reinterpret_cast<const lds_u64x2 *>((uintptr_t)addr);
            };
            auto readB = [&](int j) -> u64x2 {
                const u32 addr = __builtin_amdgcn_perm(idx2[j / 4], b, 0x0c020000u | ((4u + (j % 4)) << 8));
                return *reinterpret_cast<const lds_u64x2 *>((uintptr_t)addr + 32768);
            };
            u64x2 va[LOOK], vb[LOOK];
#pragma unroll
            for (int q = 0; q < LOOK; ++q) { va[q] = readA(q); vb[q] = readB(q); }
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const u64x2 a = va[j % LOOK], c = vb[j % LOOK];
                const u32 l0 = __builtin_amdgcn_bitop3_b32((u32)acc[j][0], (u32)a.x, (u32)c.x, 0x96);
                const u32 h0 = __builtin_amdgcn_bitop3_b32((u32)(acc[j][0] >> 32), (u32)(a.x >> 32), (u32)(c.x >> 32), 0x96);
                const u32 l1 = __builtin_amdgcn_bitop3_b32((u32)acc[j][1], (u32)a.y, (u32)c.y, 0x96);
                const u32 h1 = __builtin_amdgcn_bitop3_b32((u32)(acc[j][1] >> 32), (u32)(a.y >> 32), (u32)(c.y >> 32), 0x96);
                acc[j][0] = ((u64)h0 << 32) | l0;
                acc[j][1] = ((u64)h1 << 32) | l1;
                asm volatile("" : "+v"(acc[j][0]), "+v"(acc[j][1]));
                if (j + LOOK < R) { va[j % LOOK] = readA(j + LOOK); vb[j % LOOK] = readB(j + LOOK); }
            }
#pragma unroll
            for (int q = 0; q < LOOK; ++q) { __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
#pragma unroll
            for (int j = 0; j < R - LOOK; ++j) { __builtin_amdgcn_sched_group_barrier(0x002, 6, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
            __builtin_amdgcn_sched_group_barrier(0x002, 4 * LOOK, 0);
        }
        u64 r = 0;
        for (int j = 0; j < R; ++j) r ^= acc[j][0] ^ acc[j][1];
        out[blockIdx.x * 512 + threadIdx.x] = r;
        return;
    }
    for (int it = 0; it < iters; ++it) {
        const u32 b = base | ((it & 1) << 16);
        auto read = [&](int j) -> u64x2 {
            u32 addr;
            if (MODE == 2) addr = (idx[j / 4] & 0xff00u) | b;          // hoisted by the compiler: one VGPR per 4 rows... keep it cheap
            else addr = __builtin_amdgcn_perm(idx[j / 4], b, 0x0c020000u | ((4u + (j % 4)) << 8));
            return *reinterpret_cast<const lds_u64x2 *>((uintptr_t)addr);
        };
        u64x2 v[LOOK];
#pragma unroll
        for (int q = 0; q < LOOK; ++q) v[q] = read(q);
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (MODE == 0) { one.x ^= v[j % LOOK].x; one.y ^= v[j % LOOK].y; asm volatile("" : "+v"(one)); }
            else { acc[j][0] ^= v[j % LOOK].x; acc[j][1] ^= v[j % LOOK].y; asm volatile("" : "+v"(acc[j][0]), "+v"(acc[j][1])); }
            if (j + LOOK < R) v[j % LOOK] = read(j + LOOK);
        }
#pragma unroll
        for (int q = 0; q < LOOK; ++q) { __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
        for (int j = 0; j < R - LOOK; ++j) { __builtin_amdgcn_sched_group_barrier(0x002, 5, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        __builtin_amdgcn_sched_group_barrier(0x002, 4 * LOOK, 0);
    }
    u64 r = one.x ^ one.y;
    for (int j = 0; j < R; ++j) r ^= acc[j][0] ^ acc[j][1];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int MODE, int R, int LOOK> void run(const char *name, u64 *out, u32 *idx) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE, R, LOOK>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    const int iters = 2000, blocks = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, R, LOOK>), dim3(blocks), dim3(512), LDS, 0, out, idx, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, R, LOOK>), dim3(blocks), dim3(512), LDS, 0, out, idx, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_iter_us = ms * 1e3 / iters / (blocks / 256.0);
    const double instr = MODE == 8 ? 16.0 * R : (MODE >= 3 ? (MODE == 5 ? 8 * 8 : (MODE == 7 ? 4 * 16 : 8 * 16)) : 8.0 * R);
    printf("%-44s %8.3f ms  %.3f us per iteration per CU  = %.0f cycles @2.3GHz, %.2f cycles per DS wave-instr\n", name, ms, per_iter_us, per_iter_us * 2300,
           per_iter_us * 2300 / instr);
}
int main() {
    u64 *out; u32 *idx; hipMalloc(&out, 1024 * 512 * 8); hipMalloc(&idx, 1 << 16);   // idx: 16384 dwords (MODE 8 reads a second set from +4096)
    u32 *h = (u32 *)malloc(1 << 16); for (int i = 0; i < (1 << 14); ++i) h[i] = (u32)rand() * 2654435761u;
    hipMemcpy(idx, h, 1 << 16, hipMemcpyHostToDevice);
    run<0, 40, 8>("reads only, R=40 LOOK=8", out, idx);
    run<0, 40, 12>("reads only, R=40 LOOK=12", out, idx);
    run<1, 40, 8>("reads + 4 xor + perm, R=40 LOOK=8", out, idx);
    run<1, 40, 12>("reads + 4 xor + perm, R=40 LOOK=12", out, idx);
    run<1, 48, 6>("reads + 4 xor + perm, R=48 LOOK=6", out, idx);
    run<2, 40, 8>("reads + 4 xor (addr without perm), R=40", out, idx);
    run<1, 16, 8>("reads + 4 xor + perm, R=16 LOOK=8", out, idx);
    run<8, 48, 3>("2 tables, xor3: R=48, 3 pairs in flight", out, idx);
    run<8, 48, 4>("2 tables, xor3: R=48, 4 pairs in flight", out, idx);
    run<8, 40, 4>("2 tables, xor3: R=40, 4 pairs in flight", out, idx);
    run<8, 40, 6>("2 tables, xor3: R=40, 6 pairs in flight", out, idx);
    run<3, 16, 8>("table build: 16 x ds_write_b64 per lane", out, idx);
    run<4, 16, 8>("64 KiB: 16 x ds_write_b64 per lane, no VALU", out, idx);
    run<5, 16, 8>("64 KiB: 8 x ds_write_b128 per lane, no VALU", out, idx);
    run<7, 16, 8>("32 KiB: 16 x ds_write_b64, even waves only", out, idx);
    return 0;
}
