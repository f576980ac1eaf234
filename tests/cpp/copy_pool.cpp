// CPU-only exerciser of the host side of the staging pipeline (pls_amd/csrc/host_pipeline.hpp): the copy pool's
// epoch handshake (every job of every parallel_for runs exactly once, whatever the job and thread counts) and the
// tile repacking in both directions on strided column-major matrices.  Built with -fsanitize=thread by
// tests/test_host_sanitized.py; no HIP call is made.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../pls_amd/csrc/host_pipeline.hpp"

int main() {
    int bad = 0;
    for (int threads : {1, 2, 3, 8}) {
        plsh::CopyPool pool(threads);
        for (int round = 0; round < 3000; ++round) {
            const int njobs = 1 + (round * 7) % 67;
            std::vector<std::atomic<int>> hit(njobs);
            for (auto &h : hit) h.store(0);
            pool.parallel_for(njobs, [&](int j) { hit[j].fetch_add(1); });
            for (int j = 0; j < njobs; ++j) bad += (hit[j].load() != 1);
        }
    }
    // repack: host matrix (ld > rows) -> tile (ld = tile rows) -> another host matrix
    plsh::CopyPool pool(4);
    for (size_t es : {(size_t)4, (size_t)8}) {
        const int64_t rows = 70001, cols = 13, ld = rows + 5;
        std::vector<char> src((size_t)ld * cols * es), dst((size_t)ld * cols * es, 0), tile((size_t)20000 * cols * es);
        for (size_t i = 0; i < src.size(); ++i) src[i] = (char)((i * 2654435761u) >> 13);
        for (int64_t r0 = 0; r0 < rows; r0 += 20000) {
            const int64_t rbn = std::min<int64_t>(20000, rows - r0);
            plsh::repack(pool, tile.data(), src.data(), ld, r0, 0, rbn, cols, es, true);
            plsh::repack(pool, tile.data(), dst.data(), ld, r0, 0, rbn, cols, es, false);
        }
        for (int64_t c = 0; c < cols; ++c)
            for (int64_t i = 0; i < rows * (int64_t)es; ++i) bad += (src[(size_t)c * ld * es + i] != dst[(size_t)c * ld * es + i]);
        for (int64_t c = 0; c < cols; ++c)  // the padding rows were never touched
            for (int64_t i = rows * (int64_t)es; i < ld * (int64_t)es; ++i) bad += (dst[(size_t)c * ld * es + i] != 0);
    }
    const plsh::Tiling t1(1 << 20, 512, 8), t2(16777216, 64, 8), t3(10, 3, 4);
    bad += !(t1.rb == (1 << 20) && t1.cb == 4) + !(t2.rb == (int64_t)(plsh::STAGE_BYTES / 8) && t2.cb == 1) + !(t3.rb == 10 && t3.cb == 3);
    std::printf(bad ? "FAILED %d\n" : "ok %d\n", bad);
    return bad ? 1 : 0;
}
