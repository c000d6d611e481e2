// Non-temporal write bandwidth of 512 MB regions of one 8 GiB hipMalloc block, two block orders: workgroup b writes piece b ("interleaved":
// consecutive 64 KB pieces go to the 8 XCDs round robin) or XCD (b & 7) walks the (b & 7)-th contiguous eighth of the region ("contiguous").
// hipcc --offload-arch=gfx950 -O3 tools/ubench_xcdwrite.hip -o tools/ubench_xcdwrite && tools/ubench_xcdwrite
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool CONTIG>
__global__ __launch_bounds__(256) void k_write(u32x4 *dst, long n_pieces) {           // a piece = 64 KB = 4096 chunks of 16 bytes
    long b = blockIdx.x;
    if (CONTIG) { const long per = (gridDim.x + 7) / 8; b = (b & 7) * per + (b >> 3); }
    if (b >= n_pieces) return;
    u32x4 *p = dst + b * 4096 + threadIdx.x;
    const u32x4 v = {1u, 2u, 3u, (unsigned)b};
#pragma unroll
    for (int k = 0; k < 16; ++k) __builtin_nontemporal_store(v, p + k * 256);
}
int main() {
    const size_t total = (size_t)8 << 30, region = (size_t)512 << 20;
    char *d; if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; ++pass)
        for (int mode = 0; mode < 2; ++mode) {
            printf("%s:", mode ? "contiguous " : "interleaved");
            for (size_t off = 0; off < total; off += region) {
                const long n_pieces = region / 65536;
                const unsigned grid = (unsigned)((n_pieces + 7) / 8 * 8);
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    hipEventRecord(e0, 0);
                    if (mode) hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, (u32x4 *)(d + off), n_pieces);
                    else hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, (u32x4 *)(d + off), n_pieces);
                    hipEventRecord(e1, 0); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
                }
                printf(" %.2f", region / best / 1e9);
            }
            printf("  TB/s per 512 MB region\n");
        }
    // the whole block at once
    for (int mode = 0; mode < 2; ++mode) {
        const long n_pieces = total / 65536; const unsigned grid = (unsigned)((n_pieces + 7) / 8 * 8);
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, 0);
            if (mode) hipLaunchKernelGGL(k_write<true>, dim3(grid), dim3(256), 0, 0, (u32x4 *)d, n_pieces);
            else hipLaunchKernelGGL(k_write<false>, dim3(grid), dim3(256), 0, 0, (u32x4 *)d, n_pieces);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("%s whole 8 GiB: %.2f TB/s\n", mode ? "contiguous " : "interleaved", total / best / 1e9);
    }
    return 0;
}
