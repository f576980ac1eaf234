// host_entry_time -- the reference's entry point, timed through the C++ drop-in itself:
//     PLS::Model(const Mat2D &X, const Mat2D &Y, KERNEL_TYPE1, A)      (reference src/pls.cpp:340-353)
// with X, Y in ordinary (pageable) host memory, at BASELINE config 3 by default.  Everything the constructor does is
// inside the timed region: the transfer of X and Y to the device(s) through the pinned staging pipeline, the fit,
// the read-back of W, P, Q, R.  (The scores stay on the device until print_state() asks for them.)
// Usage: host_entry_time [N K M A reps]      PLS_HIP_DEVICES / PLS_HIP_ALGO / PLS_HIP_COPY_THREADS as for the library
#include <PLS/pls.h>
#include <pls_hip.h>
#include <sched.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    const long N = argc > 1 ? std::atol(argv[1]) : (1L << 20), K = argc > 2 ? std::atol(argv[2]) : 512;
    const long M = argc > 3 ? std::atol(argv[3]) : 1, A = argc > 4 ? std::atol(argv[4]) : 20;
    const int reps = argc > 5 ? std::atoi(argv[5]) : 5;
    Mat2D X(N, K), Y(N, M);
    {  // cheap structured data (8 latent factors + noise), filled by a few threads; values are irrelevant to the timing
        const int nt = 8;
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                for (long k = t; k < K; k += nt)
                    for (long i = 0; i < N; ++i) {
                        const uint64_t h = mix64((uint64_t)i * 1315423911ull + (uint64_t)k);
                        const double z = (double)(int64_t)(mix64((uint64_t)i) >> 40) * (1.0 / 8388608.0) - 1.0;
                        X(i, k) = z * (double)((k % 5) - 2) + (double)(int64_t)(h >> 40) * (0.25 / 8388608.0) - 0.25;
                    }
            });
        for (auto &t : th) t.join();
        for (long j = 0; j < M; ++j)
            for (long i = 0; i < N; ++i) {
                const double z = (double)(int64_t)(mix64((uint64_t)i) >> 40) * (1.0 / 8388608.0) - 1.0;
                Y(i, j) = z + (double)(int64_t)(mix64((uint64_t)i * 977 + j) >> 40) * (0.125 / 8388608.0) - 0.125;
            }
    }
    const double gb = (double)N * K * 8 / 1e9;
    double best = 1e30, sum = 0;
    double b00 = 0;
    for (int r = 0; r < reps + 1; ++r) {  // first construction also creates the device context and the staging buffers
        const auto t0 = std::chrono::steady_clock::now();
        PLS::Model m(X, Y, PLS::KERNEL_TYPE1, (size_t)A);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (r == 0) {
            std::printf("first construction (context + staging buffers created): %.1f ms\n", ms);
            b00 = std::real(m.coefficients()(0, 0));
            continue;
        }
        best = std::min(best, ms);
        sum += ms;
        std::printf("Model(X, Y, KERNEL_TYPE1, %ld): %.2f ms  = %.1f components/s  (%.2f GB of X: %.1f GB/s if it were all transfer)\n",
                    A, ms, A / (ms * 1e-3), gb, gb / (ms * 1e-3));
    }
    {   // the same steps through the C-ABI alone, in this process (same NUMA placement): transfer | fit
        pls_hip_group g = nullptr;
        int dev0 = 0;
        if (pls_hip_group_create(&g, 1, &dev0) == PLS_HIP_OK) {
            for (int r = 0; r < 2; ++r) {
                pls_hip_matrix mX = nullptr, mY = nullptr, mT = nullptr;
                auto t0 = std::chrono::steady_clock::now();
                pls_hip_group_upload(g, X.data(), N, N, K, PLS_HIP_F64, &mX);
                pls_hip_group_upload(g, Y.data(), N, N, M, PLS_HIP_F64, &mY);
                auto t1 = std::chrono::steady_clock::now();
                pls_hip_group_alloc(g, N, A, PLS_HIP_F64, &mT);
                std::vector<double> W(K * A), P(K * A), R(K * A), Q(M * A);
                pls_hip_group_fit(g, mX, mY, A, PLS_HIP_KERNEL_TYPE1, W.data(), P.data(), Q.data(), R.data(), mT, nullptr);
                auto t2 = std::chrono::steady_clock::now();
                pls_hip_group_free(g, mX); pls_hip_group_free(g, mY); pls_hip_group_free(g, mT);
                std::printf("C-ABI alone (cpu %d): upload %.2f ms (%.1f GB/s) | alloc T + fit %.2f ms\n", sched_getcpu(),
                            std::chrono::duration<double, std::milli>(t1 - t0).count(),
                            gb / std::chrono::duration<double>(t1 - t0).count(),
                            std::chrono::duration<double, std::milli>(t2 - t1).count());
            }
            pls_hip_group_destroy(g);
        }
    }
    std::printf("RESULT N=%ld K=%ld M=%ld A=%ld  best %.2f ms (%.1f components/s)  mean %.2f ms   PCIe bound at 63 GB/s: %.1f ms   B[0,0]=%.6g\n",
                N, K, M, A, best, A / (best * 1e-3), sum / reps, gb / 63.0 * 1e3, b00);
    return 0;
}
