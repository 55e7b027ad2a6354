// gem_hostcopy.hpp -- the CPU half of moving caller-owned PAGEABLE arrays to and from the device.
//
// The node hands the entry points of gpu_process.cu:1096-1141 / :1165-1192 / :1283-1291 plain heap or stack arrays, 0.5 - 1.4 MB
// each.  Handed to the runtime one by one they cost ~12 us per array before the first byte moves (MI355X box, tools/ubench/pcie.hip,
// tools/dbg/pcie_rates.py: 0.5 MB pageable 15-19 GB/s, 13 MB in one piece 53 GB/s).  The library moves them through ITS pinned
// staging buffer (one per handle) instead: kernels that read / write that buffer over the link, or DMA commands that span several
// arrays, on the device side (gem_capi.cpp: upload_arrays, download_arrays, gem_process_points) -- and on the CPU side a memcpy
// between the buffer and the caller's arrays, which one thread does at ~21 GB/s when the data comes fresh from the device: a few
// threads, a group of arrays at a time, so that the link works on one group while the threads move its neighbour.  This file is
// only that thread pool: memcpy of a list of segments, cut into pieces, on the calling thread and `threads - 1` workers.
//
// Workers are process-lifetime (the library must stay loaded once a call with host arrays has been made), detached, asleep on a
// condition variable between frames; after a job they poll for the next one
// for kSpinNs (a frame's calls follow each other within that time) so that the wake-up latency is paid once per frame.  Each job
// is its own object held by a shared_ptr: a worker that wakes late works on the job it was woken for -- by then empty -- and never
// sees the next one half built.
#pragma once

#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace gem {

struct CopySeg { void* dst; const void* src; size_t bytes; };

class CopyPool {
public:
    static constexpr int    kMaxThreads = 16;
    static constexpr size_t kPiece = 64u << 10;
#ifndef GEM_COPY_SPIN_NS
#define GEM_COPY_SPIN_NS 300000
#endif
    static constexpr long long kSpinNs = GEM_COPY_SPIN_NS;

    static CopyPool& get() { static CopyPool* pool = new CopyPool(); return *pool; }   // never destroyed: the workers outlive main()

    // memcpy of every segment; returns when the last byte is written.  threads <= 1, or another thread's job in flight: inline.
    void run(const CopySeg* segs, int n, int threads)
    {
        size_t total = 0;
        for (int i = 0; i < n; ++i) total += segs[i].bytes;
        if (threads > kMaxThreads) threads = kMaxThreads;
        if (threads <= 1 || total < 2 * kPiece || !job_mu_.try_lock()) {
            for (int i = 0; i < n; ++i) if (segs[i].bytes) std::memcpy(segs[i].dst, segs[i].src, segs[i].bytes);
            return;
        }
        std::lock_guard<std::mutex> one_job(job_mu_, std::adopt_lock);
        auto job = std::make_shared<Job>();
        for (int i = 0; i < n; ++i)
            for (size_t o = 0; o < segs[i].bytes; o += kPiece)
                job->pieces.push_back({static_cast<unsigned char*>(segs[i].dst) + o, static_cast<const unsigned char*>(segs[i].src) + o,
                                       segs[i].bytes - o < kPiece ? segs[i].bytes - o : kPiece});
        job->workers = threads - 1;
        grow(threads - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            current_ = job;
            gen_.fetch_add(1, std::memory_order_release);
        }
        cv_.notify_all();
        help(*job);
        while (job->done.load(std::memory_order_acquire) < job->pieces.size()) cpu_relax();
    }

private:
    struct Job {
        std::vector<CopySeg> pieces;
        std::atomic<size_t> next{0}, done{0};
        int workers = 0;
    };

    static void cpu_relax()
    {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield");
#endif
    }

    static void help(Job& j)
    {
        const size_t n = j.pieces.size();
        for (;;) {
            const size_t i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= n) return;
            std::memcpy(j.pieces[i].dst, j.pieces[i].src, j.pieces[i].bytes);
            j.done.fetch_add(1, std::memory_order_release);
        }
    }

    // The CPUs that share a last-level cache with the calling thread (its CCD: /sys/.../cache/index3/shared_cpu_list), within what
    // the process may use.  Workers are kept there: on a two-socket host a worker on the other socket copies at half the rate, and
    // the staging buffer was first touched from here.  Empty if it cannot be read (the workers then run wherever the scheduler puts them).
    static bool near_cpus(cpu_set_t& out)
    {
        CPU_ZERO(&out);
        const int cpu = sched_getcpu();
        if (cpu < 0) return false;
        char path[96];
        std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
        FILE* f = std::fopen(path, "r");
        if (!f) return false;
        char buf[512];
        const bool got = std::fgets(buf, sizeof buf, f) != nullptr;
        std::fclose(f);
        if (!got) return false;
        cpu_set_t allowed;
        if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return false;
        int n = 0;
        for (const char* p = buf; *p;) {                               // "0-7,128-135"
            char* end;
            long a = std::strtol(p, &end, 10), b = a;
            if (end == p) break;
            if (*end == '-') { p = end + 1; b = std::strtol(p, &end, 10); if (end == p) break; }
            for (long c = a; c <= b && c < CPU_SETSIZE; ++c) if (c >= 0 && CPU_ISSET(c, &allowed)) { CPU_SET(c, &out); ++n; }
            p = *end == ',' ? end + 1 : end;
            if (*end != ',') break;
        }
        return n >= 2;
    }

    void grow(int workers)
    {
        if (started_ < workers && !near_known_) { near_ok_ = near_cpus(near_); near_known_ = true; }
        while (started_ < workers) {
            const int idx = started_++;
            std::thread t([this, idx] { work(idx); });
            if (near_ok_) pthread_setaffinity_np(t.native_handle(), sizeof near_, &near_);
            t.detach();
        }
    }

    void work(int idx)
    {
        using clock = std::chrono::steady_clock;
        uint64_t seen = 0;
        auto last = clock::now() - std::chrono::seconds(1);
        for (;;) {
            while (gen_.load(std::memory_order_acquire) == seen &&
                   std::chrono::duration_cast<std::chrono::nanoseconds>(clock::now() - last).count() < kSpinNs)
                cpu_relax();
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_.load(std::memory_order_acquire) != seen; });
                seen = gen_.load(std::memory_order_acquire);
                job = current_;
            }
            if (job && idx < job->workers) help(*job);
            last = clock::now();
        }
    }

    std::mutex job_mu_;                 // one job at a time
    std::mutex mu_;                     // current_ / gen_ / cv_
    std::condition_variable cv_;
    std::shared_ptr<Job> current_;
    std::atomic<uint64_t> gen_{0};
    int started_ = 0;                   // under job_mu_
    cpu_set_t near_;                    // where the workers run (near_cpus at the first job), if near_ok_
    bool near_known_ = false, near_ok_ = false;
};

}  // namespace gem
