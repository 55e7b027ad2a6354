// gem_transport.hpp -- what carries the collectives of the multi-GPU path (internal header of gem_capi.cpp).
//
// The reference has no collectives at all; the tiling is BASELINE.json's north star: "frames shard naturally by
// spatial tile across the 8 GPUs of one node with a RCCL all-gather over xGMI of the fused submap".  A handle that joined a
// communicator talks through TWO of these objects -- one for the exchange of a step's sorted records, one for the all-gather of
// the fused layers -- each with its own stream, so that the two kinds of traffic of consecutive steps do not queue behind each
// other (gem_capi.cpp: gem_add_sharded_device).
//
//   RcclTransport      one process per GPU: ncclAllGather / grouped ncclSend + ncclRecv over xGMI -- the product.
//   LoopbackTransport  W handles of ONE process on ONE device (include/gem_hip_debug.h: gem_comm_init_loopback), one host thread
//                      per handle: the same calls become device-to-device copies ordered by events, with the stream semantics
//                      of the real thing (a transfer starts when both sides' streams have reached it, and work enqueued behind
//                      it on either stream comes after it).  It exists so that the EXACT host code of the W > 1 paths -- the
//                      count / offset / base arithmetic, the buffer rotation, the cross-stream ordering -- runs in the GPU suite
//                      on a one-GPU box; it also checks what RCCL cannot: that every send meets a receive of the same size.
#pragma once

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace gem {

constexpr int kTransportMaxRanks = 8;

struct Transport {
    int nranks = 1, rank = 0;
    std::string err;
    virtual ~Transport() {}
    // every rank contributes `words` 32-bit words; recv holds nranks x words in rank order
    virtual bool all_gather(const uint32_t* send, uint32_t* recv, size_t words, hipStream_t st) = 0;
    // one group of point-to-point transfers, sizes in 32-bit words: EVERY rank of the communicator opens and closes the group
    // (possibly empty) -- that is how both users below call it, and what the loopback's rendezvous relies on
    virtual bool group_begin() = 0;
    virtual bool send(const void* p, size_t words, int peer, hipStream_t st) = 0;
    virtual bool recv(void* p, size_t words, int peer, hipStream_t st) = 0;
    virtual bool group_end(hipStream_t st) = 0;
    // a rank that cannot go on (an allocation failed between two collectives of a step): the peers' pending calls fail instead of hanging
    virtual void abort() = 0;
};

// ---- RCCL ----------------------------------------------------------------------------------------------------------------------
struct RcclTransport final : Transport {
    ncclComm_t comm = nullptr;
    bool owns = true;
    ncclResult_t r_group = ncclSuccess;

    ~RcclTransport() override { if (comm && owns) ncclCommDestroy(comm); }
    bool fail(ncclResult_t r) { err = ncclGetErrorString(r); return false; }
    bool all_gather(const uint32_t* send, uint32_t* recv, size_t words, hipStream_t st) override
    {
        if (!comm) { err = "communicator aborted"; return false; }
        const ncclResult_t r = ncclAllGather(send, recv, words, ncclUint32, comm, st);
        return r == ncclSuccess ? true : fail(r);
    }
    bool group_begin() override
    {
        if (!comm) { err = "communicator aborted"; r_group = ncclInvalidUsage; return false; }
        r_group = ncclGroupStart(); return r_group == ncclSuccess ? true : fail(r_group);
    }
    bool send(const void* p, size_t words, int peer, hipStream_t st) override
    {
        if (r_group == ncclSuccess) r_group = ncclSend(p, words, ncclUint32, peer, comm, st);
        return r_group == ncclSuccess ? true : fail(r_group);
    }
    bool recv(void* p, size_t words, int peer, hipStream_t st) override
    {
        if (r_group == ncclSuccess) r_group = ncclRecv(p, words, ncclUint32, peer, comm, st);
        return r_group == ncclSuccess ? true : fail(r_group);
    }
    bool group_end(hipStream_t) override
    {
        if (!comm) return false;                                       // (group_begin refused: no group is open)
        const ncclResult_t r2 = ncclGroupEnd();                        // (always closed, also after a failed send / recv)
        if (r_group != ncclSuccess) return fail(r_group);
        return r2 == ncclSuccess ? true : fail(r2);
    }
    // (a transport that only BORROWS its communicator -- the gather transport when ncclCommSplit is not available -- leaves the
    //  abort to the owner and just lets go of the pointer: one ncclCommAbort per communicator)
    void abort() override { if (comm && owns) ncclCommAbort(comm); comm = nullptr; }
};

// ---- loopback ------------------------------------------------------------------------------------------------------------------
struct LoopChannel {
    struct Xfer { const void* src; void* dst; size_t words; int peer; };
    struct Post {
        int kind = 0;                                                  // 1 = all-gather, 2 = group
        const uint32_t* ag_send = nullptr; uint32_t* ag_recv = nullptr; size_t ag_words = 0;
        std::vector<Xfer> sends, recvs;
        hipEvent_t ready = nullptr, done = nullptr;                    // created by the rank's transport, alive as long as it
        bool joined = false;
    };
    std::mutex mu;
    std::condition_variable cv;
    int W = 0, arrived = 0;
    unsigned long long gen = 0;
    bool poisoned = false;
    Post post[kTransportMaxRanks];

    // all W ranks arrive, or the channel is poisoned / the wait times out (a peer returned early with an error of its own)
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (poisoned) return false;
        const unsigned long long g = gen;
        if (++arrived == W) { arrived = 0; ++gen; cv.notify_all(); return true; }
        const bool ok = cv.wait_for(lk, std::chrono::seconds(30), [&] { return gen != g || poisoned; });
        if (!ok || poisoned) { poisoned = true; cv.notify_all(); return false; }
        return true;
    }
    void poison() { std::lock_guard<std::mutex> lk(mu); poisoned = true; cv.notify_all(); }
};

struct LoopWorld {
    int W = 0;
    LoopChannel ch[2];                                                 // 0: record exchange, 1: layer all-gather
};

inline std::shared_ptr<LoopWorld> loop_world(long long id, int W, std::string* why)
{
    static std::mutex mu;
    static std::map<long long, std::weak_ptr<LoopWorld>> worlds;
    std::lock_guard<std::mutex> lk(mu);
    std::shared_ptr<LoopWorld> w = worlds[id].lock();
    if (!w) {
        w = std::make_shared<LoopWorld>();
        w->W = W; w->ch[0].W = W; w->ch[1].W = W;
        worlds[id] = w;
    } else if (w->W != W) { if (why) *why = "loopback world: joined with a different number of ranks"; return nullptr; }
    return w;
}

struct LoopbackTransport final : Transport {
    std::shared_ptr<LoopWorld> world;
    LoopChannel* ch = nullptr;
    bool in_group = false;

    ~LoopbackTransport() override
    {
        if (!ch) return;
        LoopChannel::Post& me = ch->post[rank];
        { std::lock_guard<std::mutex> lk(ch->mu); me.joined = false; }
        if (me.ready) hipEventDestroy(me.ready);
        if (me.done) hipEventDestroy(me.done);
        me.ready = me.done = nullptr;
    }
    bool join(std::shared_ptr<LoopWorld> w, int channel, int r)
    {
        world = std::move(w); ch = &world->ch[channel]; nranks = world->W; rank = r;
        LoopChannel::Post& me = ch->post[rank];
        {
            std::lock_guard<std::mutex> lk(ch->mu);
            if (me.joined) { err = "loopback world: rank already taken"; ch = nullptr; return false; }
            me.joined = true;
        }
        if (hipEventCreateWithFlags(&me.ready, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&me.done, hipEventDisableTiming) != hipSuccess) { err = "hipEventCreate"; return false; }
        return true;
    }
    bool hip(hipError_t e, const char* what) { if (e == hipSuccess) return true; err = std::string("loopback: ") + what + ": " + hipGetErrorString(e); ch->poison(); return false; }
    bool sync_fail(const char* what) { err = std::string("loopback: ") + what; return false; }

    // The rendezvous both collectives share.  post -> [barrier] -> every rank enqueues, on ITS stream, a wait for each source's
    // `ready` event and the copies into its own buffers, then records `done` -> [barrier] -> every rank's stream waits for the
    // `done` of the ranks that read from it: whatever it enqueues next may overwrite what it sent.
    bool run(hipStream_t st)
    {
        LoopChannel::Post& me = ch->post[rank];
        if (!hip(hipEventRecord(me.ready, st), "hipEventRecord")) return false;
        if (!ch->barrier()) return sync_fail("a peer did not arrive at the collective (it failed on its own, or the ranks disagree on the order of collectives)");
        bool ok = true;
        std::vector<int> readers;
        for (int p = 0; p < nranks && ok; ++p) {
            const LoopChannel::Post& o = ch->post[p];
            if (o.kind != me.kind) { ok = sync_fail("the ranks are in different collectives"); break; }
            if (me.kind == 1) {
                if (o.ag_words != me.ag_words) { ok = sync_fail("all-gather: the ranks contribute different sizes"); break; }
                if (p != rank) { ok = hip(hipStreamWaitEvent(st, o.ready, 0), "hipStreamWaitEvent"); readers.push_back(p); }
                if (ok && me.ag_words) ok = hip(hipMemcpyAsync(me.ag_recv + (size_t)p * me.ag_words, o.ag_send, me.ag_words * 4, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
                continue;
            }
            // group: my k-th receive from p meets p's k-th send to me
            size_t si = 0; bool waited = false;
            for (const LoopChannel::Xfer& rv : me.recvs) {
                if (rv.peer != p) continue;
                while (si < o.sends.size() && o.sends[si].peer != rank) ++si;
                if (si == o.sends.size()) { ok = sync_fail("a receive without a matching send"); break; }
                if (o.sends[si].words != rv.words) { ok = sync_fail("a receive and its send differ in size"); break; }
                if (!waited && p != rank) { ok = hip(hipStreamWaitEvent(st, o.ready, 0), "hipStreamWaitEvent"); waited = true; if (!ok) break; }
                if (rv.words) ok = hip(hipMemcpyAsync(rv.dst, o.sends[si].src, rv.words * 4, hipMemcpyDeviceToDevice, st), "hipMemcpyAsync");
                if (!ok) break;
                ++si;
            }
            if (ok) { while (si < o.sends.size() && o.sends[si].peer != rank) ++si; if (si != o.sends.size()) ok = sync_fail("a send without a matching receive"); }
            // who reads what I send: their `done` is what my stream waits for below
            for (const LoopChannel::Xfer& sd : me.sends) if (sd.peer == p && p != rank) { readers.push_back(p); break; }
        }
        if (!ok) { ch->poison(); return false; }
        if (!hip(hipEventRecord(me.done, st), "hipEventRecord")) return false;
        if (!ch->barrier()) return sync_fail("a peer failed inside the collective");
        for (int p : readers) if (!hip(hipStreamWaitEvent(st, ch->post[p].done, 0), "hipStreamWaitEvent")) return false;
        return true;
    }

    bool all_gather(const uint32_t* send, uint32_t* recv, size_t words, hipStream_t st) override
    {
        LoopChannel::Post& me = ch->post[rank];
        me.kind = 1; me.ag_send = send; me.ag_recv = recv; me.ag_words = words;
        me.sends.clear(); me.recvs.clear();
        return run(st);
    }
    bool group_begin() override
    {
        LoopChannel::Post& me = ch->post[rank];
        me.kind = 2; me.ag_send = nullptr; me.ag_recv = nullptr; me.ag_words = 0; me.sends.clear(); me.recvs.clear();
        in_group = true;
        return true;
    }
    bool send(const void* p, size_t words, int peer, hipStream_t) override
    {
        if (!in_group || peer < 0 || peer >= nranks) return sync_fail("send outside a group / bad peer");
        ch->post[rank].sends.push_back(LoopChannel::Xfer{p, nullptr, words, peer});
        return true;
    }
    bool recv(void* p, size_t words, int peer, hipStream_t) override
    {
        if (!in_group || peer < 0 || peer >= nranks) return sync_fail("recv outside a group / bad peer");
        ch->post[rank].recvs.push_back(LoopChannel::Xfer{nullptr, p, words, peer});
        return true;
    }
    bool group_end(hipStream_t st) override { in_group = false; return run(st); }
    void abort() override { if (ch) ch->poison(); }
};

} // namespace gem
