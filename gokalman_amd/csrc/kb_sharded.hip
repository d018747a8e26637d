// kb_sharded.hip -- the native multi-device driver: ONE process, one kb_batch + one host thread + one
// HIP stream per device, the filter index range split into contiguous shards -- GPU g owns [g N / G, (g + 1) N / G) -- and NO
// collective in the update path (filters share nothing: vanilla.go:216-218).  The one exchange the path has, the Monte-Carlo /
// chi-square statistics (montecarlo.go:18-59, chisquare.go:85-94: steps x 2n resp. steps x 2 doubles), is ONE
// ncclAllReduce(ncclSum, ncclDouble) over RCCL / xGMI on the shards' device buffers, communicators from ncclCommInitAll (single
// process); RCCL is loaded at run time (dlopen) the first time a sharded batch with more than one DISTINCT device needs it.
// Shards that share a device (a test box with one GPU) or a box without librccl fall back to adding the shards on the host, in
// shard order.
//
// This is what a Go / C++ host -- the reference's audience -- calls to use a whole node; the Python bench drives the same
// engine as one process per GPU over torch.distributed (gokalman_amd/dist.py), which is the other shape of the same sharding.
#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "kb_internal.h"

// RCCL's types and entry points, bound with dlsym (no link-time dependency: a process that never shards never loads it)
extern "C" {
typedef struct ncclComm *kb_ncclComm_t;
typedef int (*kb_ncclCommInitAll_t)(kb_ncclComm_t *, int, const int *);
typedef int (*kb_ncclCommDestroy_t)(kb_ncclComm_t);
typedef int (*kb_ncclAllReduce_t)(const void *, void *, size_t, int /* ncclDataType_t */, int /* ncclRedOp_t */, kb_ncclComm_t, hipStream_t);
typedef int (*kb_ncclGroup_t)(void);
typedef const char *(*kb_ncclGetErrorString_t)(int);
}

namespace kb {

constexpr int KB_NCCL_DOUBLE = 8;   // ncclFloat64 (rccl.h: ncclInt8 = 0, ..., ncclFloat32 = 7, ncclFloat64 = 8)
constexpr int KB_NCCL_SUM = 0;      // ncclSum

struct Rccl {
    void *lib = nullptr;
    kb_ncclCommInitAll_t init_all = nullptr;
    kb_ncclCommDestroy_t destroy = nullptr;
    kb_ncclAllReduce_t all_reduce = nullptr;
    kb_ncclGroup_t group_start = nullptr, group_end = nullptr;
    kb_ncclGetErrorString_t error_string = nullptr;
    bool ok = false;
    static Rccl &get() {
        static Rccl r;
        static std::once_flag once;
        std::call_once(once, [] {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (r.lib) break;
            }
            if (!r.lib) return;
            r.init_all = (kb_ncclCommInitAll_t)dlsym(r.lib, "ncclCommInitAll");
            r.destroy = (kb_ncclCommDestroy_t)dlsym(r.lib, "ncclCommDestroy");
            r.all_reduce = (kb_ncclAllReduce_t)dlsym(r.lib, "ncclAllReduce");
            r.group_start = (kb_ncclGroup_t)dlsym(r.lib, "ncclGroupStart");
            r.group_end = (kb_ncclGroup_t)dlsym(r.lib, "ncclGroupEnd");
            r.error_string = (kb_ncclGetErrorString_t)dlsym(r.lib, "ncclGetErrorString");
            r.ok = r.init_all && r.destroy && r.all_reduce && r.group_start && r.group_end;
        });
        return r;
    }
};

// One host thread per shard: every call on shard g's handle runs on thread g (which selected its device once).
class Worker {
   public:
    Worker() : th_([this] { loop(); }) {}
    ~Worker() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        th_.join();
    }
    void post(std::function<int()> job) {
        std::lock_guard<std::mutex> lk(mu_);
        job_ = std::move(job);
        busy_ = true;
        cv_.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [this] { return !busy_; });
        return rc_;
    }
    std::string error;   // kb_last_error() of the worker thread after a failed job (the message is thread-local)

   private:
    void loop() {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            cv_.wait(lk, [this] { return stop_ || busy_; });
            if (stop_) return;
            std::function<int()> job = std::move(job_);
            lk.unlock();
            const int rc = job();
            std::string err = rc ? kb_last_error() : "";
            lk.lock();
            rc_ = rc;
            error = std::move(err);
            busy_ = false;
            cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool busy_ = false, stop_ = false;
    int rc_ = 0;
    std::thread th_;
};

}  // namespace kb

using namespace kb;

struct kb_sharded {
    int kind = 0, n = 0, p = 0, m = 0, dtype = 0;
    int64_t N = 0;
    std::vector<kb_batch *> shard;
    std::vector<int64_t> first;           // first[g] = global index of shard g's first filter; first[G] = N
    std::vector<int> device;
    std::vector<std::unique_ptr<Worker>> worker;
    std::vector<kb_ncclComm_t> comm;      // one per shard when RCCL is in use
    bool rccl_tried = false, rccl_on = false;
    int last_reduce_rccl = 0;
};

namespace {

// run fn(g) on every shard's thread, wait for all; the first failure wins (its message becomes this thread's kb_last_error)
int on_all(kb_sharded *s, const std::function<int(int)> &fn) {
    const int G = (int)s->shard.size();
    for (int g = 0; g < G; g++) s->worker[g]->post([&fn, g] { return fn(g); });
    int rc = KB_OK;
    for (int g = 0; g < G; g++) {
        const int r = s->worker[g]->wait();
        if (r && !rc) { rc = r; set_error("shard %d: %s", g, s->worker[g]->error.c_str()); }
    }
    return rc;
}

// reduce_sum / kb_sharded_mc_run touch several devices from the CALLER's thread: its current device is put back on return
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

bool distinct_devices(const kb_sharded *s) {
    for (size_t i = 0; i < s->device.size(); i++)
        for (size_t j = i + 1; j < s->device.size(); j++)
            if (s->device[i] == s->device[j]) return false;
    return true;
}

// communicators on first use; false = host-sum fallback (shards sharing a device, or no usable librccl).  A one-shard batch goes
// through RCCL as well (a one-rank all-reduce): it costs nothing and it is how a one-GPU box exercises this code path.
bool ensure_rccl(kb_sharded *s) {
    if (s->rccl_tried) return s->rccl_on;
    s->rccl_tried = true;
    if (!distinct_devices(s)) return false;
    Rccl &r = Rccl::get();
    if (!r.ok) return false;
    s->comm.assign(s->shard.size(), nullptr);
    if (r.init_all(s->comm.data(), (int)s->shard.size(), s->device.data()) != 0) { s->comm.clear(); return false; }
    s->rccl_on = true;
    return true;
}

// Sum `count` doubles held by every shard at d_buf[g] (its own device, its own stream): in place over RCCL (result on every
// shard), then shard 0's copy goes to the host; or on the host, in shard order.
int reduce_sum(kb_sharded *s, const std::vector<double *> &d_buf, size_t count, double *host_out) {
    const int G = (int)s->shard.size();
    const DeviceGuard guard;
    if (ensure_rccl(s)) {
        Rccl &r = Rccl::get();
        int nrc = r.group_start();
        for (int g = 0; g < G && nrc == 0; g++)
            nrc = r.all_reduce(d_buf[g], d_buf[g], count, KB_NCCL_DOUBLE, KB_NCCL_SUM, s->comm[g], s->shard[g]->stream);
        const int erc = r.group_end();
        if (nrc == 0) nrc = erc;
        if (nrc != 0) { set_error("ncclAllReduce failed: %s", r.error_string ? r.error_string(nrc) : "?"); return KB_ERR_HIP; }
        s->last_reduce_rccl = 1;
        KB_HIP(hipSetDevice(s->device[0]));
        KB_HIP(hipMemcpyAsync(host_out, d_buf[0], count * sizeof(double), hipMemcpyDeviceToHost, s->shard[0]->stream));
        KB_HIP(hipStreamSynchronize(s->shard[0]->stream));
        for (int g = 1; g < G; g++) {   // the other shards' streams have the collective in them: drain before their buffers are reused
            KB_HIP(hipSetDevice(s->device[g]));
            KB_HIP(hipStreamSynchronize(s->shard[g]->stream));
        }
        return KB_OK;
    }
    s->last_reduce_rccl = 0;
    std::vector<double> part(count);
    for (size_t i = 0; i < count; i++) host_out[i] = 0.0;
    for (int g = 0; g < G; g++) {
        KB_HIP(hipSetDevice(s->device[g]));
        KB_HIP(hipMemcpyAsync(part.data(), d_buf[g], count * sizeof(double), hipMemcpyDeviceToHost, s->shard[g]->stream));
        KB_HIP(hipStreamSynchronize(s->shard[g]->stream));
        for (size_t i = 0; i < count; i++) host_out[i] += part[i];
    }
    return KB_OK;
}

}  // namespace

extern "C" {

int kb_sharded_create(kb_sharded **out, int kind, int n, int p, int m, int64_t nfilters, int dtype, const int *devices, int ndev, unsigned flags) {
    if (!out) { set_error("out is NULL"); return KB_ERR_INVALID; }
    *out = nullptr;
    if (ndev < 1 || ndev > 64) { set_error("ndev must be 1..64"); return KB_ERR_INVALID; }
    if (nfilters < ndev) { set_error("fewer filters (%lld) than shards (%d)", (long long)nfilters, ndev); return KB_ERR_INVALID; }
    auto *s = new kb_sharded();
    s->kind = kind; s->n = n; s->p = p; s->m = m; s->dtype = dtype; s->N = nfilters;
    s->first.resize((size_t)ndev + 1);
    for (int g = 0; g <= ndev; g++) s->first[(size_t)g] = (int64_t)(((__int128)nfilters * g) / ndev);   // GPU g owns [g N / G, (g + 1) N / G)
    for (int g = 0; g < ndev; g++) {
        s->device.push_back(devices ? devices[g] : g);
        s->worker.emplace_back(new Worker());
    }
    s->shard.assign((size_t)ndev, nullptr);
    const int rc = on_all(s, [&](int g) {
        return kb_create(&s->shard[(size_t)g], kind, n, p, m, s->first[(size_t)g + 1] - s->first[(size_t)g], dtype, s->device[(size_t)g], flags);
    });
    if (rc) { kb_sharded_destroy(s); return rc; }
    *out = s;
    return KB_OK;
}

void kb_sharded_destroy(kb_sharded *s) {
    if (!s) return;
    if (!s->comm.empty()) {
        Rccl &r = Rccl::get();
        for (kb_ncclComm_t c : s->comm)
            if (c) (void)r.destroy(c);
    }
    if (!s->worker.empty())
        (void)on_all(s, [&](int g) { kb_destroy(s->shard[(size_t)g]); return KB_OK; });
    delete s;
}

int kb_sharded_num_shards(const kb_sharded *s) { return s ? (int)s->shard.size() : 0; }
kb_batch *kb_sharded_shard(kb_sharded *s, int g) { return (s && g >= 0 && g < (int)s->shard.size()) ? s->shard[(size_t)g] : nullptr; }
int64_t kb_sharded_first(const kb_sharded *s, int g) { return (s && g >= 0 && g <= (int)s->shard.size()) ? s->first[(size_t)g] : -1; }
int kb_sharded_used_rccl(const kb_sharded *s) { return s ? s->last_reduce_rccl : 0; }

// kb_set on every shard: a per-filter array [N][E] is cut at the shard boundaries, a shared one goes to every shard
int kb_sharded_set(kb_sharded *s, int field, const double *host, int64_t count, int broadcast, int p_rows, int64_t elems_per_filter) {
    if (!s || !host) { set_error("null argument"); return KB_ERR_INVALID; }
    if (!broadcast && count != s->N) { set_error("count must be 1 with broadcast or N=%lld without", (long long)s->N); return KB_ERR_INVALID; }
    return on_all(s, [&](int g) {
        const int64_t lo = s->first[(size_t)g], cnt = s->first[(size_t)g + 1] - lo;
        return broadcast ? kb_set(s->shard[(size_t)g], field, host, 1, 1, p_rows)
                         : kb_set(s->shard[(size_t)g], field, host + lo * elems_per_filter, cnt, 0, p_rows);
    });
}
int kb_sharded_set_noise_kind(kb_sharded *s, int noise_kind, uint64_t seed) {
    if (!s) { set_error("null argument"); return KB_ERR_INVALID; }
    return on_all(s, [&](int g) { return kb_set_noise_kind(s->shard[(size_t)g], noise_kind, seed); });
}
int kb_sharded_init(kb_sharded *s) {
    if (!s) { set_error("null argument"); return KB_ERR_INVALID; }
    return on_all(s, [&](int g) { return kb_init(s->shard[(size_t)g]); });
}
int kb_sharded_reset(kb_sharded *s) {
    if (!s) { set_error("null argument"); return KB_ERR_INVALID; }
    return on_all(s, [&](int g) { return kb_reset(s->shard[(size_t)g]); });
}
int kb_sharded_synchronize(kb_sharded *s) {
    if (!s) { set_error("null argument"); return KB_ERR_INVALID; }
    return on_all(s, [&](int g) { return kb_synchronize(s->shard[(size_t)g]); });
}

// LDKF.Update for every filter of every shard, the shards in parallel (host measurements [N][rows], controls [N][rows] or NULL)
int kb_sharded_update(kb_sharded *s, const double *meas, int meas_rows, const double *ctrl, int ctrl_rows) {
    if (!s || !meas) { set_error("null argument"); return KB_ERR_INVALID; }
    return on_all(s, [&](int g) {
        const int64_t lo = s->first[(size_t)g];
        return kb_update(s->shard[(size_t)g], meas + lo * meas_rows, meas_rows, ctrl ? ctrl + lo * ctrl_rows : nullptr, ctrl_rows);
    });
}
// Same with every shard's measurements already in ITS device's memory: meas[g] planar with leading dimension ld[g] (kb_update_dev)
int kb_sharded_update_dev(kb_sharded *s, const void *const *meas, const int64_t *ld_meas, const void *const *ctrl, const int64_t *ld_ctrl) {
    if (!s || !meas || !ld_meas) { set_error("null argument"); return KB_ERR_INVALID; }
    return on_all(s, [&](int g) {
        return kb_update_dev(s->shard[(size_t)g], meas[g], ld_meas[g], ctrl ? ctrl[g] : nullptr, ld_ctrl ? ld_ctrl[g] : 0);
    });
}

int kb_sharded_get(kb_sharded *s, int field, double *host, int64_t first, int64_t count, int64_t elems_per_filter) {
    if (!s || !host) { set_error("null argument"); return KB_ERR_INVALID; }
    if (first < 0 || count < 0 || first + count > s->N) { set_error("range outside the batch"); return KB_ERR_INVALID; }
    return on_all(s, [&](int g) {
        const int64_t lo = std::max(first, s->first[(size_t)g]), hi = std::min(first + count, s->first[(size_t)g + 1]);
        if (hi <= lo) return (int)KB_OK;
        return kb_get(s->shard[(size_t)g], field, host + (lo - first) * elems_per_filter, lo - s->first[(size_t)g], hi - lo);
    });
}
int kb_sharded_get_status(kb_sharded *s, uint32_t *host, int64_t first, int64_t count) {
    if (!s || !host) { set_error("null argument"); return KB_ERR_INVALID; }
    if (first < 0 || count < 0 || first + count > s->N) { set_error("range outside the batch"); return KB_ERR_INVALID; }
    return on_all(s, [&](int g) {
        const int64_t lo = std::max(first, s->first[(size_t)g]), hi = std::min(first + count, s->first[(size_t)g + 1]);
        if (hi <= lo) return (int)KB_OK;
        return kb_get_status(s->shard[(size_t)g], host + (lo - first), lo - s->first[(size_t)g], hi - lo);
    });
}

// NewMonteCarloRuns over the whole node (montecarlo.go:92-119): shard g runs the runs [first[g], first[g + 1]) of ONE ensemble (a
// run's noise depends only on its global index), the per-step sums are all-reduced (montecarlo.go:18-59's gather over runs).
// sums[steps][3][n] as kb_mc_run, over ALL runs.
int kb_sharded_mc_run(kb_sharded *s, int steps, const double *controls, int ncontrols, double *sums, unsigned mc_flags) {
    if (!s || !sums) { set_error("null argument"); return KB_ERR_INVALID; }
    const int G = (int)s->shard.size(), n = s->n;
    std::vector<double *> folded((size_t)G, nullptr), shift((size_t)G, nullptr);
    int rc = on_all(s, [&](int g) {
        return mc_run_device(*s->shard[(size_t)g], steps, controls, ncontrols, s->first[(size_t)g], mc_flags, &folded[(size_t)g], &shift[(size_t)g]);
    });
    if (rc) return rc;
    std::vector<double> tot((size_t)steps * 2 * n), sh((size_t)steps * n);
    if ((rc = reduce_sum(s, folded, tot.size(), tot.data()))) return rc;
    const DeviceGuard guard;
    KB_HIP(hipSetDevice(s->device[0]));
    KB_HIP(hipMemcpyAsync(sh.data(), shift[0], sh.size() * sizeof(double), hipMemcpyDeviceToHost, s->shard[0]->stream));   // identical on every shard
    KB_HIP(hipStreamSynchronize(s->shard[0]->stream));
    for (int t = 0; t < steps; t++)
        for (int i = 0; i < n; i++) {
            sums[((size_t)t * 3 + 0) * n + i] = tot[((size_t)t * 2 + 0) * n + i];
            sums[((size_t)t * 3 + 1) * n + i] = tot[((size_t)t * 2 + 1) * n + i];
            sums[((size_t)t * 3 + 2) * n + i] = sh[(size_t)t * n + i];
        }
    return KB_OK;
}

// NewChiSquare over the whole node (chisquare.go:16-95): sums[steps][2] = { sum NIS, sum NEES } over ALL runs
int kb_sharded_chisquare(kb_sharded *truth, kb_sharded *kf, int steps, const double *controls, int ncontrols, int replay_last_mc,
                         int with_nees, int with_nis, double *sums) {
    if (!truth || !kf || !sums) { set_error("null argument"); return KB_ERR_INVALID; }
    if (truth->shard.size() != kf->shard.size() || truth->first != kf->first || truth->device != kf->device) {
        set_error("truth and filter must be sharded alike");
        return KB_ERR_DIMS;
    }
    const int G = (int)truth->shard.size();
    std::vector<double *> folded((size_t)G, nullptr);
    int rc = on_all(truth, [&](int g) {
        return chisq_run_device(*truth->shard[(size_t)g], *kf->shard[(size_t)g], steps, controls, ncontrols, truth->first[(size_t)g], replay_last_mc,
                                with_nees, with_nis, &folded[(size_t)g]);
    });
    if (rc) return rc;
    return reduce_sum(truth, folded, (size_t)steps * 2, sums);
}

}  // extern "C"
