// which XCD does workgroup b of a 1-D grid land on?  hipcc --offload-arch=gfx950 -O2 tools/ubench_xcc.hip -o tools/ubench_xcc && tools/ubench_xcc
// (k_rs_scatter and k_emit_fused give every XCD a contiguous eighth of their tiles through blockIdx.x & 7; that is only right while the
// dispatcher deals workgroups round-robin over the XCDs)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));   // HW_REG_XCC_ID, bits [3:0]
}
int main() {
    const int n = 4096;
    unsigned *d, h[n];
    hipMalloc(&d, n * 4);
    hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    printf("first 32 workgroups:");
    for (int b = 0; b < 32; ++b) printf(" %u", h[b]);
    int ok = 0;
    for (int b = 0; b < n; ++b) ok += (h[b] == (unsigned)(b & 7));
    printf("\nworkgroups with XCC_ID == blockIdx & 7: %d of %d\n", ok, n);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    return 0;
}
