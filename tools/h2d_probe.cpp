// h2d_probe -- where the time of a host-memory fit goes: raw PCIe rates (pinned and pageable hipMemcpy), the
// staging pipeline of pls_hip_group_upload, hipMalloc/hipFree of the resident matrix, the device-resident fit.
// Usage: h2d_probe [N K]            hipcc -O2 -o tools/h2d_probe tools/h2d_probe.cpp -Iinclude -Lpls_amd/csrc -lpls_hip
#include <hip/hip_runtime.h>
#include <sched.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "pls_hip.h"

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
    const long N = argc > 1 ? atol(argv[1]) : (1L << 20), K = argc > 2 ? atol(argv[2]) : 512;
    const size_t bytes = (size_t)N * K * 8;
    std::printf("cpu of this thread: %d\n", sched_getcpu());
    double *hp = (double *)malloc(bytes);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 8; ++t)
            th.emplace_back([&, t] { for (size_t i = t * (bytes / 8 / 8); i < (t + 1) * (bytes / 8 / 8); ++i) hp[i] = (double)(i % 977) * 1e-3; });
        for (auto &t : th) t.join();
    }
    void *d = nullptr, *pin = nullptr;
    double t0 = now_ms();
    hipMalloc(&d, bytes);
    std::printf("hipMalloc %.2f GB: %.2f ms\n", bytes / 1e9, now_ms() - t0);
    const size_t pb = (size_t)1 << 30;
    hipHostMalloc(&pin, pb, hipHostMallocDefault);
    memset(pin, 1, pb);
    for (int r = 0; r < 3; ++r) {
        t0 = now_ms();
        hipMemcpy(d, pin, pb, hipMemcpyHostToDevice);
        const double ms = now_ms() - t0;
        std::printf("pinned   H2D 1 GiB: %.2f ms = %.1f GB/s\n", ms, pb / 1e6 / ms);
    }
    for (int r = 0; r < 2; ++r) {
        t0 = now_ms();
        hipMemcpy(d, hp, bytes, hipMemcpyHostToDevice);
        const double ms = now_ms() - t0;
        std::printf("pageable H2D %.2f GB: %.2f ms = %.1f GB/s\n", bytes / 1e9, ms, bytes / 1e6 / ms);
    }
    {   // host memcpy rate pageable -> pinned with 8 threads (what the staging threads do)
        t0 = now_ms();
        std::vector<std::thread> th;
        for (int t = 0; t < 8; ++t)
            th.emplace_back([&, t] { memcpy((char *)pin + t * (pb / 8), (char *)hp + t * (pb / 8), pb / 8); });
        for (auto &t : th) t.join();
        const double ms = now_ms() - t0;
        std::printf("host memcpy pageable->pinned 1 GiB, 8 threads: %.2f ms = %.1f GB/s\n", ms, pb / 1e6 / ms);
    }
    t0 = now_ms();
    hipFree(d);
    std::printf("hipFree: %.2f ms\n", now_ms() - t0);
    hipHostFree(pin);

    pls_hip_group g = nullptr;
    int dev0 = 0;
    if (pls_hip_group_create(&g, 1, &dev0) != 0) { std::printf("group_create failed\n"); return 1; }
    std::vector<double> Y((size_t)N);
    for (long i = 0; i < N; ++i) Y[i] = hp[i] + hp[(size_t)i + N];
    for (int r = 0; r < 3; ++r) {
        pls_hip_matrix mX = nullptr, mY = nullptr, mT = nullptr;
        t0 = now_ms();
        pls_hip_group_upload(g, hp, N, N, K, PLS_HIP_F64, &mX);
        const double t_up = now_ms() - t0;
        pls_hip_group_upload(g, Y.data(), N, N, 1, PLS_HIP_F64, &mY);
        double t1 = now_ms();
        pls_hip_group_alloc(g, N, 20, PLS_HIP_F64, &mT);
        std::vector<double> W(K * 20), P(K * 20), R(K * 20), Q(20);
        const int rc = pls_hip_group_fit(g, mX, mY, 20, PLS_HIP_KERNEL_TYPE1, W.data(), P.data(), Q.data(), R.data(), mT, nullptr);
        const double t_fit = now_ms() - t1;
        t1 = now_ms();
        pls_hip_group_free(g, mX); pls_hip_group_free(g, mY); pls_hip_group_free(g, mT);
        std::printf("group_upload X: %.2f ms = %.1f GB/s | alloc T + fit (rc %d): %.2f ms | free: %.2f ms\n", t_up,
                    bytes / 1e6 / t_up, rc, t_fit, now_ms() - t1);
    }
    pls_hip_group_destroy(g);
    {   // the plain C-ABI entry of INTEGRATION.md section B: pls_hip_fit on host pointers (upload, fit, T read back)
        pls_hip_handle h = nullptr;
        if (pls_hip_create(&h, 0, nullptr) == PLS_HIP_OK) {
            std::vector<double> W(K * 20), P(K * 20), R(K * 20), Q(20);
            double *T = (double *)malloc((size_t)N * 20 * 8);
            memset(T, 0, (size_t)N * 20 * 8);
            for (int r = 0; r < 3; ++r) {
                t0 = now_ms();
                const int rc = pls_hip_fit(h, hp, N, Y.data(), N, N, K, 1, 20, PLS_HIP_KERNEL_TYPE1, PLS_HIP_F64, PLS_HIP_MEM_HOST,
                                           W.data(), P.data(), Q.data(), R.data(), T, N, nullptr);
                std::printf("pls_hip_fit(MEM_HOST), KERNEL plan, T read back (rc %d): %.2f ms\n", rc, now_ms() - t0);
            }
            free(T);
            pls_hip_destroy(h);
        }
    }
    free(hp);
    return 0;
}
