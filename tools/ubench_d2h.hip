// micro-benchmark: device -> pageable host copy rate with 1..16 host threads (one stream each), fresh vs pre-touched destination
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
int main() {
    const size_t bytes = (size_t)2 << 30;
    char *d; hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipDeviceSynchronize();
    for (int pre = 0; pre < 2; ++pre)
    for (int nt : {1, 2, 4, 8, 16}) {
        char *h = (char *)malloc(bytes);
        if (pre) memset(h, 0, bytes);
        std::vector<hipStream_t> st(nt);
        for (auto &s : st) hipStreamCreate(&s);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        const size_t chunk = bytes / nt;
        for (int k = 0; k < nt; ++k)
            th.emplace_back([&, k] { hipMemcpyAsync(h + k * chunk, d + k * chunk, chunk, hipMemcpyDeviceToHost, st[k]); hipStreamSynchronize(st[k]); });
        for (auto &t : th) t.join();
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %2d %s destination: %.1f ms  %.1f GB/s\n", nt, pre ? "touched" : "fresh  ", s * 1e3, bytes / s / 1e9);
        for (auto &s2 : st) hipStreamDestroy(s2);
        free(h);
    }
    // pinned reference
    char *p; hipHostMalloc(&p, bytes);
    auto t0 = std::chrono::steady_clock::now();
    hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost);
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("pinned destination: %.1f ms  %.1f GB/s\n", s * 1e3, bytes / s / 1e9);
    return 0;
}
