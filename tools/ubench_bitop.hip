// micro-benchmark: issue rate of v_bitop3_b32 (VGPR,VGPR,VGPR) vs (VGPR,SGPR,VGPR) vs v_and+v_xor, and v_bcnt, on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32;
template <int MODE>
__global__ __launch_bounds__(256) void k(u32 *out, const u32 *in, int iters, u32 s0, u32 s1) {
    u32 a[16], b[4];
    for (int i = 0; i < 16; ++i) a[i] = in[threadIdx.x + 256 * i];
    for (int i = 0; i < 4; ++i) b[i] = in[threadIdx.x * 3 + i];
    u32 sa = __builtin_amdgcn_readfirstlane(s0), sb = __builtin_amdgcn_readfirstlane(s1);
    for (int it = 0; it < iters; ++it) {
        u32 va, vb;
        asm volatile("v_mov_b32 %0, %1" : "=v"(va) : "s"(sa));
        asm volatile("v_mov_b32 %0, %1" : "=v"(vb) : "s"(sb));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) a[i] = __builtin_amdgcn_bitop3_b32(a[i], b[i & 3], b[(i + 1) & 3], 0x78);
            if (MODE == 1) a[i] = __builtin_amdgcn_bitop3_b32(a[i], (i & 1) ? sa : sb, b[i & 3], 0x78);
            if (MODE == 2) a[i] ^= (b[i & 3] & b[(i + 1) & 3]);
            if (MODE == 3) a[i] = __builtin_popcount(b[i & 3] ^ a[i]) + a[i];
            if (MODE == 4) a[i] = a[i] * 3u + b[i & 3];
            if (MODE == 5) { u32 t = a[i] ^ b[i & 3]; asm volatile("" : "+v"(t)); a[i] = t; }
            if (MODE == 6) a[i] = __builtin_amdgcn_bitop3_b32(a[i], (i & 1) ? va : vb, b[i & 3], 0x78);
            if (MODE == 7) { u32 t = a[i] ^ ((i & 1) ? sa : sb); asm volatile("" : "+v"(t)); a[i] = t; }
        }
        sa += 1; sb ^= sa;
    }
    u32 r = 0;
    for (int i = 0; i < 16; ++i) r ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int MODE> void run(const char *name, u32 *out, u32 *in) {
    int iters = 4096, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, 16, 1u, 2u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, in, iters, 1u, 2u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 256 * iters * 16 * (MODE == 2 ? 2 : (MODE == 3 ? 3 : 1));
    printf("%-28s %8.3f ms  %.3e lane-instr/s (%.1f%% of 256CU*128*2.4GHz)\n", name, ms, ops / (ms * 1e-3), 100 * ops / (ms * 1e-3) / 7.864e13);
}
int main() {
    u32 *out, *in; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 1 << 20); hipMemset(in, 1, 1 << 20);
    run<0>("bitop3 v,v,v", out, in); run<1>("bitop3 v,s,v", out, in); run<2>("and+xor (2 instr)", out, in);
    run<3>("xor+bcnt+add (3)", out, in); run<4>("mad_u32", out, in); run<5>("xor v,v (VOP2)", out, in); run<6>("bitop3 v,(s->v mov),v", out, in); run<7>("xor v,s (VOP2)", out, in);
    return 0;
}
