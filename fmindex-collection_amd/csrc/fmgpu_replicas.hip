// fmgpu_replicas.hip — one index replicated on several GPUs of a node, a query batch sharded over them (SURVEY 8b's `fmgpu_set_devices`, 8e): for callers
// that are ONE process (the C++ mirror, a reference build with the binding of INTEGRATION.md).  The path shards by independent queries and the index is
// read-only, so there is no exchange between the replicas: every replica searches a contiguous range of the batch on its own device, from its own host
// thread (a persistent worker per replica), and writes its results into its range of the caller's HOST arrays — the "gather" is that write.  (Ranks of a torch.distributed job gather with
// RCCL instead: bench.py / parallel.py.)  Built on the public entry points only; no kernel lives here.
#include "fmgpu_common.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace fmgpu {
// one host thread per replica, started by fmgpu_replicas_load and joined by _destroy: the per-thread call scratch of the library (frame stacks, counters, events:
// fmgpu_common.h) lives as long as the replica does — threads made per call would allocate and free it per call, and hipFree synchronises the device (with a
// device listed twice it stalls the sibling replica)
struct Worker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has = false, quit = false, done = false;
    int rc = 0; std::string msg;
    void run(int device) {
        hipError_t e = hipSetDevice(device);
        const int dev_rc = e != hipSuccess ? hip_fail(e, "hipSetDevice") : 0;
        const std::string dev_msg = dev_rc ? last_error_cstr() : "";
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has || quit; });
            if (quit) return;
            std::function<int()> f = std::move(job);
            has = false;
            lk.unlock();
            int r = dev_rc; std::string why = dev_msg;
            if (!r) {
                try { r = f(); if (r) why = last_error_cstr(); }
                catch (const std::bad_alloc&) { r = FMGPU_ERR_NOMEM; why = "out of host memory"; }
                catch (const std::exception& ex) { r = FMGPU_ERR_INVALID; why = ex.what(); }
            }
            lk.lock();
            rc = r; msg = why; done = true;
            cv.notify_all();
        }
    }
    void post(std::function<int()> f) { { std::lock_guard<std::mutex> g(m); job = std::move(f); has = true; done = false; } cv.notify_all(); }
    void wait() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return done; }); }
    void stop() { { std::lock_guard<std::mutex> g(m); quit = true; } cv.notify_all(); if (th.joinable()) th.join(); }
};
struct Replicas {
    std::vector<fmgpu_index_t> index;
    std::vector<int> device;
    std::vector<std::unique_ptr<Worker>> worker;
    std::mutex calls;                      // one sharded call at a time per replica set (a worker holds one job)
    int peer_copies = 0;                   // replicas that were made by a device-to-device copy of the first one (the rest read the file)
    ~Replicas() { for (auto& w : worker) if (w) w->stop(); }
};
namespace {
struct DeviceGuard {                       // the calling thread keeps its current device
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
// runs fn(i) for every replica on the replica's own thread (its device current); the first error (code + text) is the call's
template <class F>
int on_every_replica(Replicas& r, F&& fn) {
    std::lock_guard<std::mutex> one(r.calls);
    const size_t n = r.index.size();
    for (size_t i = 0; i < n; ++i) r.worker[i]->post([&fn, i]() -> int { return fn(i); });
    for (size_t i = 0; i < n; ++i) r.worker[i]->wait();
    for (size_t i = 0; i < n; ++i) if (r.worker[i]->rc) return fail(r.worker[i]->rc, "replica " + std::to_string(i) + " (device " + std::to_string(r.device[i]) + "): " + r.worker[i]->msg);
    return 0;
}
// no C++ exception crosses the ABI: what the host side of a sharded call can throw (vector growth, thread start) becomes an error code
template <class F>
int guarded(F&& f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(FMGPU_ERR_NOMEM, "out of host memory"); }
    catch (const std::exception& ex) { return fail(FMGPU_ERR_INVALID, std::string("replica set: ") + ex.what()); }
}
struct Shard { uint64_t first, count; std::vector<uint64_t> qoff; };
// contiguous ranges of the batch, offsets rebased to each range's first query
std::vector<Shard> shards_of(const uint64_t* qoff, uint64_t nq, size_t n) {
    std::vector<Shard> s(n);
    for (size_t i = 0; i < n; ++i) {
        s[i].first = nq / n * i + std::min<uint64_t>(i, nq % n);
        s[i].count = nq / n + (i < nq % n ? 1 : 0);
        s[i].qoff.resize(s[i].count + 1);
        for (uint64_t k = 0; k <= s[i].count; ++k) s[i].qoff[k] = qoff[s[i].first + k] - qoff[s[i].first];
    }
    return s;
}
void add_stats(fmgpu_stats* total, const fmgpu_stats& s) {
    total->lf_steps += s.lf_steps; total->hits += s.hits; total->table_bytes += s.table_bytes; total->table_accesses += s.table_accesses; total->table_steps += s.table_steps;
    total->kernel_ms = std::max(total->kernel_ms, s.kernel_ms); total->prepass_ms = std::max(total->prepass_ms, s.prepass_ms);
}
int host_only(const void* p, const char* what) {
    if (p && is_device_pointer(p)) return fail(FMGPU_ERR_INVALID, std::string(what) + " lives in one device's memory: the replicas take and fill host buffers");
    return 0;
}
}  // namespace
}  // namespace fmgpu

using namespace fmgpu;

extern "C" {

static int replicas_load(const char* path, const int32_t* devices, int32_t ndev, fmgpu_replicas_t* out) {
    if (!path || !out || (ndev > 0 && !devices)) return fail(FMGPU_ERR_INVALID, "path / devices / out is null");
    *out = nullptr;
    int have = 0;
    if (hipGetDeviceCount(&have) != hipSuccess || have == 0) { (void)hipGetLastError(); return fail(FMGPU_ERR_NO_DEVICE, "no HIP device visible — the product path has no CPU fallback"); }
    auto r = std::unique_ptr<Replicas>(new (std::nothrow) Replicas);
    if (!r) return fail(FMGPU_ERR_NOMEM, "out of host memory");
    if (ndev <= 0) for (int d = 0; d < have; ++d) r->device.push_back(d);             // every visible device
    else for (int32_t i = 0; i < ndev; ++i) {
        if (devices[i] < 0 || devices[i] >= have) return fail(FMGPU_ERR_INVALID, "device " + std::to_string(devices[i]) + " is not one of the " + std::to_string(have) + " visible devices");
        r->device.push_back(devices[i]);                                                // (the same device twice is allowed: two replicas share its HBM)
    }
    r->index.assign(r->device.size(), nullptr);
    DeviceGuard keep;
    const std::string file = path;
    for (size_t i = 0; i < r->device.size(); ++i) {
        r->worker.emplace_back(new Worker);
        Worker* w = r->worker.back().get();
        const int d = r->device[i];
        w->th = std::thread([w, d] { w->run(d); });
    }
    // the file is read once, onto the first listed device; the other replicas are copies made device to device (over xGMI where the devices are peers), each by
    // its own thread; a replica whose copy fails reads the file itself
    int rc = 0;
    {
        hipError_t e = hipSetDevice(r->device[0]);
        rc = e != hipSuccess ? hip_fail(e, "hipSetDevice") : fmgpu_index_load(file.c_str(), &r->index[0]);
    }
    r->peer_copies = 0;
    if (!rc && r->index.size() > 1) {
        std::atomic<int> copied{0};
        rc = on_every_replica(*r, [&](size_t i) {
            if (i == 0) return 0;
            if (fmgpu_index_clone(r->index[0], &r->index[i]) == 0) { ++copied; return 0; }
            return fmgpu_index_load(file.c_str(), &r->index[i]);
        });
        r->peer_copies = copied.load();
    }
    if (rc) {
        const std::string why = last_error_cstr();
        for (size_t i = 0; i < r->index.size(); ++i) if (r->index[i]) { (void)hipSetDevice(r->device[i]); (void)fmgpu_index_destroy(r->index[i]); }
        return fail(rc, why);
    }
    *out = reinterpret_cast<fmgpu_replicas_t>(r.release());
    return 0;
}

int fmgpu_replicas_load(const char* path, const int32_t* devices, int32_t ndev, fmgpu_replicas_t* out) {
    return guarded([&] { return replicas_load(path, devices, ndev, out); });
}

int fmgpu_replicas_peer_copies(fmgpu_replicas_t rh, int32_t* count) {
    if (!rh || !count) return fail(FMGPU_ERR_INVALID, "null replica set / count");
    *count = reinterpret_cast<Replicas*>(rh)->peer_copies;
    return 0;
}

int fmgpu_replicas_destroy(fmgpu_replicas_t rh) {
    if (!rh) return 0;
    auto* r = reinterpret_cast<Replicas*>(rh);
    DeviceGuard keep;
    for (auto& w : r->worker) if (w) w->stop();                    // (a worker's call scratch goes with its thread, before the handles)
    int rc = 0;
    for (size_t i = 0; i < r->index.size(); ++i) {
        if (hipSetDevice(r->device[i]) != hipSuccess) { (void)hipGetLastError(); continue; }
        int e = fmgpu_index_destroy(r->index[i]);
        if (e && !rc) rc = e;
    }
    delete r;
    return rc;
}

int fmgpu_replicas_info(fmgpu_replicas_t rh, int32_t* count, int32_t* devices, int32_t capacity, fmgpu_index_t* first) {
    if (!rh) return fail(FMGPU_ERR_INVALID, "null replica set");
    auto* r = reinterpret_cast<Replicas*>(rh);
    if (count) *count = (int32_t)r->index.size();
    if (devices) for (int32_t i = 0; i < capacity && (size_t)i < r->device.size(); ++i) devices[i] = r->device[i];
    if (first) *first = r->index[0];
    return 0;
}

int fmgpu_replicas_search_exact(fmgpu_replicas_t rh, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, uint64_t* out_lb, uint64_t* out_len, fmgpu_stats* stats) {
    if (!rh) return fail(FMGPU_ERR_INVALID, "null replica set");
    if (nq && (!qoff || !out_lb || !out_len)) return fail(FMGPU_ERR_INVALID, "null buffer");
    int rc;
    if ((rc = host_only(qbuf, "qbuf")) || (rc = host_only(qoff, "qoff")) || (rc = host_only(out_lb, "out_lb")) || (rc = host_only(out_len, "out_len"))) return rc;
    auto* r = reinterpret_cast<Replicas*>(rh);
    if (stats) *stats = fmgpu_stats{};
    if (nq == 0) return 0;
    return guarded([&] {
        const auto sh = shards_of(qoff, nq, r->index.size());
        std::vector<fmgpu_stats> st(sh.size());
        DeviceGuard keep;
        int rc2 = on_every_replica(*r, [&](size_t i) {
            if (sh[i].count == 0) return 0;
            return fmgpu_search_exact(r->index[i], qbuf + qoff[sh[i].first], sh[i].qoff.data(), sh[i].count, out_lb + sh[i].first, out_len + sh[i].first, stats ? &st[i] : nullptr, nullptr);
        });
        if (rc2) return rc2;
        if (stats) for (const auto& s_ : st) add_stats(stats, s_);
        return 0;
    });
}

}  // extern "C"

namespace fmgpu {
namespace {
// a search that produces hit records, sharded: call(i, qbuf, qoff, nq, out, cap, &count, stats) runs it on replica i
template <class Call>
int sharded_hits(Replicas* r, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats, Call&& call) {
    if (stats) *stats = fmgpu_stats{};
    *out_count = 0;
    if (nq == 0) return 0;
    const auto sh = shards_of(qoff, nq, r->index.size());
    std::vector<fmgpu_stats> st(sh.size());
    std::vector<std::vector<fmgpu_hit>> part(sh.size());
    std::vector<uint64_t> got(sh.size(), 0);
    DeviceGuard keep;
    int rc = on_every_replica(*r, [&](size_t i) {
        if (sh[i].count == 0) return 0;
        // a replica's records first go to a buffer of its own (its share of the caller's capacity, grown once if the shard holds more: the records of a
        // batch are not spread evenly — one read of a satellite repeat has thousands)
        uint64_t cap = std::min<uint64_t>(capacity, capacity / sh.size() + capacity / (4 * sh.size()) + 1024);
        for (int attempt = 0; attempt < 2; ++attempt) {
            part[i].resize(cap);
            int e = call(i, qbuf + qoff[sh[i].first], sh[i].qoff.data(), sh[i].count, part[i].data(), cap, &got[i], stats ? &st[i] : nullptr);
            if (e != FMGPU_ERR_CAPACITY) return e;
            if (got[i] > capacity || attempt) return e;            // more than the whole call may return: the caller's to grow
            cap = got[i];
        }
        return 0;
    });
    uint64_t total = 0;
    for (uint64_t g : got) total += g;
    *out_count = total;                                             // (records produced, also when they do not fit: like the single-handle calls)
    if (rc) return rc;
    if (total > capacity) return fail(FMGPU_ERR_CAPACITY, std::to_string(total) + " hit records, capacity " + std::to_string(capacity));
    uint64_t at = 0;
    for (size_t i = 0; i < sh.size(); ++i) {
        for (uint64_t k = 0; k < got[i]; ++k) { fmgpu_hit h = part[i][k]; h.qidx += sh[i].first; out[at++] = h; }     // query numbers of the whole batch
        if (stats) add_stats(stats, st[i]);
    }
    return 0;
}
}  // namespace
}  // namespace fmgpu

extern "C" {

int fmgpu_replicas_search_scheme(fmgpu_replicas_t rh, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_scheme* scheme, uint64_t max_hits_per_query,
                                 fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats) {
    if (!rh) return fail(FMGPU_ERR_INVALID, "null replica set");
    if (!scheme || !out_count || (nq && !qoff) || (capacity && !out)) return fail(FMGPU_ERR_INVALID, "null buffer");
    int rc;
    if ((rc = host_only(qbuf, "qbuf")) || (rc = host_only(qoff, "qoff")) || (rc = host_only(out, "out"))) return rc;
    auto* r = reinterpret_cast<Replicas*>(rh);
    return guarded([&] { return sharded_hits(r, qbuf, qoff, nq, out, capacity, out_count, stats, [&](size_t i, const uint8_t* qb, const uint64_t* qo, uint64_t n, fmgpu_hit* o, uint64_t cap, uint64_t* cnt, fmgpu_stats* st) {
        return fmgpu_search_scheme(r->index[i], qb, qo, n, scheme, max_hits_per_query, o, cap, cnt, st, nullptr);
    }); });
}

int fmgpu_replicas_search_ng21(fmgpu_replicas_t rh, const uint8_t* qbuf, const uint64_t* qoff, uint64_t nq, const fmgpu_expanded_scheme* scheme, uint64_t max_hits_per_query,
                               fmgpu_hit* out, uint64_t capacity, uint64_t* out_count, fmgpu_stats* stats) {
    if (!rh) return fail(FMGPU_ERR_INVALID, "null replica set");
    if (!scheme || !out_count || (nq && !qoff) || (capacity && !out)) return fail(FMGPU_ERR_INVALID, "null buffer");
    int rc;
    if ((rc = host_only(qbuf, "qbuf")) || (rc = host_only(qoff, "qoff")) || (rc = host_only(out, "out"))) return rc;
    auto* r = reinterpret_cast<Replicas*>(rh);
    return guarded([&] { return sharded_hits(r, qbuf, qoff, nq, out, capacity, out_count, stats, [&](size_t i, const uint8_t* qb, const uint64_t* qo, uint64_t n, fmgpu_hit* o, uint64_t cap, uint64_t* cnt, fmgpu_stats* st) {
        return fmgpu_search_ng21(r->index[i], qb, qo, n, scheme, max_hits_per_query, o, cap, cnt, st, nullptr);
    }); });
}

// FMIndex::locate of `count` rows, sharded like a batch of queries (rows are rows of the index, the same on every replica)
int fmgpu_replicas_locate(fmgpu_replicas_t rh, const uint64_t* rows, uint64_t count, uint64_t* out_seq, uint64_t* out_pos, uint64_t* out_steps, fmgpu_stats* stats) {
    if (!rh) return fail(FMGPU_ERR_INVALID, "null replica set");
    if (count && (!rows || !out_seq || !out_pos || !out_steps)) return fail(FMGPU_ERR_INVALID, "null buffer");
    int rc;
    if ((rc = host_only(rows, "rows")) || (rc = host_only(out_seq, "out_seq")) || (rc = host_only(out_pos, "out_pos")) || (rc = host_only(out_steps, "out_steps"))) return rc;
    auto* r = reinterpret_cast<Replicas*>(rh);
    if (stats) *stats = fmgpu_stats{};
    if (count == 0) return 0;
    return guarded([&] {
        const size_t n = r->index.size();
        std::vector<fmgpu_stats> st(n);
        DeviceGuard keep;
        int rc2 = on_every_replica(*r, [&](size_t i) {
            const uint64_t first = count / n * i + std::min<uint64_t>(i, count % n), mine = count / n + (i < count % n ? 1 : 0);
            if (mine == 0) return 0;
            return fmgpu_locate(r->index[i], rows + first, mine, out_seq + first, out_pos + first, out_steps + first, stats ? &st[i] : nullptr, nullptr);
        });
        if (rc2) return rc2;
        if (stats) for (const auto& s_ : st) add_stats(stats, s_);
        return 0;
    });
}

}  // extern "C"
