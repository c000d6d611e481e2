// micro-benchmark: pure HBM write ceiling on gfx950 (16-B stores, plain vs non-temporal, grid-stride vs one-shot)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_fill(u32x4 *out, size_t n, unsigned v) {
    u32x4 x = (u32x4)(v);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        x.x ^= (unsigned)i;
        if (NT) __builtin_nontemporal_store(x, out + i); else out[i] = x;
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(const u32x4 *in, u32x4 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u32x4 x = in[i];
        if (NT) __builtin_nontemporal_store(x, out + i); else out[i] = x;
    }
}
int main() {
    size_t bytes = (size_t)8 << 30, n = bytes / 16;
    u32x4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(b, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int grids[] = {2048, 8192, 65536, 0};
    for (int gi = 0; gi < 4; ++gi) {
        int g = grids[gi] ? grids[gi] : (int)(n / 256);
        for (int nt = 0; nt < 2; ++nt) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL(k_fill<true>, dim3(g), dim3(256), 0, 0, a, n, 7u); else hipLaunchKernelGGL(k_fill<false>, dim3(g), dim3(256), 0, 0, a, n, 7u);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("fill  grid=%8d nt=%d: %.3f ms  %.2f TB/s written\n", g, nt, ms, bytes / (ms * 1e-3) / 1e12);
        }
    }
    for (int nt = 0; nt < 2; ++nt) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (nt) hipLaunchKernelGGL(k_copy<true>, dim3(8192), dim3(256), 0, 0, b, a, n); else hipLaunchKernelGGL(k_copy<false>, dim3(8192), dim3(256), 0, 0, b, a, n);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy  nt=%d: %.3f ms  %.2f TB/s (read+write)\n", nt, ms, 2.0 * bytes / (ms * 1e-3) / 1e12);
    }
    hipEventRecord(e0); hipMemsetAsync(a, 0, bytes, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("hipMemsetAsync: %.3f ms %.2f TB/s\n", ms, bytes / (ms * 1e-3) / 1e12);
    return 0;
}
