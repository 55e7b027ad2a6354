// hostcopy_pool.cpp -- gem_amd/csrc/gem_hostcopy.hpp on its own (no GPU): every byte of every segment arrives, for thread counts
// 1 .. 8, segment sizes around the piece size, back-to-back jobs (workers still polling) and jobs after a pause (workers asleep),
// and with several threads calling run() at once (one gets the pool, the others copy inline).
#include "../../gem_amd/csrc/gem_hostcopy.hpp"

#include <cstdio>
#include <random>

static bool one_job(std::mt19937& rng, int threads)
{
    const int n = 1 + (int)(rng() % 9);
    std::vector<std::vector<unsigned char>> src(n), dst(n);
    std::vector<gem::CopySeg> segs(n);
    const size_t sizes[] = {0, 1, 4095, 65535, 65536, 65537, 131072, 524288, 1440000, 3000001};
    for (int i = 0; i < n; ++i) {
        const size_t b = sizes[rng() % 10];
        src[i].resize(b + 16); dst[i].assign(b + 16, 0xAB);
        for (size_t k = 0; k < b; ++k) src[i][k] = (unsigned char)(rng() >> 8);
        segs[i] = {dst[i].data(), src[i].data(), b};
    }
    gem::CopyPool::get().run(segs.data(), n, threads);
    for (int i = 0; i < n; ++i) {
        if (std::memcmp(dst[i].data(), src[i].data(), segs[i].bytes) != 0) return false;
        for (size_t k = segs[i].bytes; k < segs[i].bytes + 16; ++k) if (dst[i][k] != 0xAB) return false;      // nothing past the end
    }
    return true;
}

int main()
{
    std::mt19937 rng(1234);
    for (int round = 0; round < 60; ++round) {
        const int threads = 1 + round % 8;
        if (!one_job(rng, threads)) { std::printf("mismatch (round %d, %d threads)\n", round, threads); return 1; }
        if (round % 20 == 19) std::this_thread::sleep_for(std::chrono::milliseconds(3));        // workers go to sleep
    }
    std::atomic<int> bad{0};
    std::vector<std::thread> callers;
    for (int t = 0; t < 4; ++t)
        callers.emplace_back([t, &bad] { std::mt19937 r(99 + t); for (int k = 0; k < 12; ++k) if (!one_job(r, 4)) ++bad; });
    for (auto& c : callers) c.join();
    if (bad) { std::printf("mismatch with concurrent callers\n"); return 1; }
    std::printf("ok\n");
    return 0;
}
