// kb_api.hip -- the C ABI (include/gokalman_amd.h): handle lifetime, uploads, the
// Update entry points, Estimate getters.  Host-side control only; all arithmetic on
// filter data happens in the HIP kernels (there is no CPU path).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include <chrono>

#include <cxxabi.h>

#include "kb_internal.h"

namespace kb {
bool squareroot_fused_ok(const Batch &b, const StepArgs &a);   // kb_squareroot_reg.hip

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int hip_fail(hipError_t e, const char *what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return KB_ERR_HIP;
}

static bool is_ldkf(int k) { return k == KB_VANILLA || k == KB_VANILLA_PREDICT || k == KB_SQUAREROOT || k == KB_INFORMATION; }
static bool is_nldkf(int k) { return k == KB_SRIF || k == KB_HYBRID || k == KB_BATCH_LS; }

Layout make_layout(int kind, int n, int pmax, int m, unsigned flags) {
    Layout L;
    L.n = n; L.pmax = pmax; L.m = m;
    L.st_vec = 0; L.st_mat = n;
    L.st_mat_full = (kind == KB_SRIF || kind == KB_BATCH_LS);
    L.st_elems = n + (L.st_mat_full ? n * n : tri(n));
    int o = 0;
    L.es_ppred = o; o += L.st_mat_full ? n * n : tri(n);
    L.es_gain = o;  o += n * pmax;
    L.es_innov = o; o += pmax;
    L.es_yhat = o;  o += pmax;
    L.es_dobs = o;  o += pmax;
    L.es_elems = o;
    (void)flags;
    o = 0;
    L.nq = (kind == KB_HYBRID) ? m : 0;
    L.mo_F = o; o += n * n;
    L.mo_H = o; o += pmax * n;
    L.mo_Q = o; o += (kind == KB_HYBRID) ? tri(m > 0 ? m : 1) : tri(n);
    L.mo_R = o; o += tri(pmax);
    L.mo_G = o; o += n * (m > 0 ? m : 0);
    L.mo_LQ = o; o += tri(n);
    L.mo_LR = o; o += tri(pmax);
    if (kind == KB_INFORMATION) {
        L.mo_Finv = o; o += n * n;
        L.mo_Qinv = o; o += tri(n);      // Q^-1 and R^-1 are inverses of symmetric matrices: stored packed (upper triangle of the
        L.mo_Rinv = o; o += tri(pmax);   // computed inverse, mirrored on use), 8 (n(n-1) + p(p-1)) / 2 bytes per filter-step less
    }
    L.mo_elems = o;
    return L;
}

// ---- dev_alloc / dev_free (kb_internal.h) -------------------------------------------------------------
static bool fence_mode() {
    static const bool on = [] { const char *e = getenv("KB_DEBUG_FENCE"); return e && *e && *e != '0'; }();
    return on;
}
static std::mutex g_fence_mu;
static std::unordered_map<void *, void *> g_fence_base;   // handed-out pointer -> hipMalloc'ed base

hipError_t dev_alloc(void **p, size_t bytes) {
    if (!fence_mode() || bytes == 0) return hipMalloc(p, bytes);
    const size_t G = size_t(2) << 20, rounded = (bytes + G - 1) / G * G;   // (scripts/diag_fence.hip: the first byte behind a 2 MB multiple faults, behind a 4 / 64 KB multiple not)
    void *base = nullptr;
    const hipError_t e = hipMalloc(&base, rounded);
    if (e != hipSuccess) return e;
    void *q = (char *)base + ((rounded - bytes) & ~size_t(255));
    std::lock_guard<std::mutex> lk(g_fence_mu);
    g_fence_base[q] = base;
    *p = q;
    return hipSuccess;
}

hipError_t dev_free(void *p) {
    if (fence_mode()) {
        std::lock_guard<std::mutex> lk(g_fence_mu);
        const auto it = g_fence_base.find(p);
        if (it != g_fence_base.end()) { p = it->second; g_fence_base.erase(it); }
    }
    return hipFree(p);
}

int ensure_stage(Batch &b, size_t bytes) {
    if (b.stage_bytes >= bytes) return KB_OK;
    if (b.d_stage) KB_HIP(dev_free(b.d_stage));
    b.d_stage = nullptr; b.stage_bytes = 0;
    KB_HIP(dev_alloc(&b.d_stage, bytes));
    b.stage_bytes = bytes;
    return KB_OK;
}

int ensure_pin(Batch &b) {
    if (b.h_pin) return KB_OK;
    KB_HIP(hipHostMalloc(&b.h_pin, KB_PIN_FLAG_OFF + 64, hipHostMallocMapped));
    *(volatile uint32_t *)((char *)b.h_pin + KB_PIN_FLAG_OFF) = 0;
    b.pin_seq = 0;
    if (hipHostGetDevicePointer(&b.d_pin, b.h_pin, 0) != hipSuccess) {
        (void)hipHostFree(b.h_pin);
        b.h_pin = b.d_pin = nullptr;
        set_error("hipHostGetDevicePointer failed");
        return KB_ERR_HIP;
    }
    return KB_OK;
}

int ensure_xp(Batch &b) {
    if (b.d_xp) return KB_OK;
    KB_HIP(dev_alloc(&b.d_xp, b.block_bytes(b.n + tri(b.n))));
    return KB_OK;
}

// ---- HeavyScope (kb_internal.h) -------------------------------------------------------------------
static hipStream_t heavy_stream(int device) {
    static std::mutex mu;
    static hipStream_t streams[64] = {nullptr};
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!streams[device]) {
        if (hipStreamCreateWithFlags(&streams[device], hipStreamNonBlocking) != hipSuccess) streams[device] = nullptr;
    }
    return streams[device];
}

HeavyScope::HeavyScope(int device, hipStream_t user, bool heavy, hipEvent_t *cached) : stream(user), user_(user) {
    if (!heavy) return;
    hipStream_t hs = heavy_stream(device);
    if (!hs) return;   // could not create it: stay on the caller's stream
    for (int i = 0; i < 2; i++) {
        if (cached && cached[i]) { ev_[i] = cached[i]; continue; }
        if (hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming) != hipSuccess) { ev_[i] = nullptr; return; }
        if (cached) cached[i] = ev_[i];
    }
    owned_ = cached == nullptr;
    if (hipEventRecord(ev_[0], user_) != hipSuccess || hipStreamWaitEvent(hs, ev_[0], 0) != hipSuccess) return;
    heavy_ = true;
    stream = hs;
}

HeavyScope::~HeavyScope() {
    if (heavy_) {
        (void)hipEventRecord(ev_[1], stream);
        (void)hipStreamWaitEvent(user_, ev_[1], 0);
    }
    if (owned_)
        for (int i = 0; i < 2; i++)
            if (ev_[i]) (void)hipEventDestroy(ev_[i]);   // released by the runtime once the recorded work has completed
}

// After a synchronisation of the handle's stream: a KB_SRIF batch that ran the dense Update kernel learns here whether any
// filter failed in it (Batch::srif_leftover); if none did, the steady-state kernel runs alone again.
void after_sync(Batch &b) {
    if (b.srif_leftover && b.h_srif_fail && *b.h_srif_fail == 0u) b.srif_leftover = 0;
}

int use_device(const Batch &b) {
    KB_HIP(hipSetDevice(b.device));
    return KB_OK;
}

// element maps ---------------------------------------------------------------------
// full row-major n x n -> packed symmetric (upper triangle only is taken)
static void map_sym_in(int n, int dst_off, int16_t *map) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) map[i * n + j] = (j >= i) ? (int16_t)(dst_off + symi(i, j)) : (int16_t)-1;
}
// packed symmetric -> full row-major (mirror)
static void map_sym_out(int n, int src_off, int16_t *map) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) map[i * n + j] = (int16_t)(src_off + symi(i, j));
}
static void map_dense(int rows, int cols, int off, int ld, int16_t *map) {
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < cols; j++) map[i * cols + j] = (int16_t)(off + i * ld + j);
}
// lower-triangular factor stored packed (L[i][k], k<=i at symi(k,i)) -> full
static void map_lower_out(int n, int src_off, int16_t *map) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) map[i * n + j] = (j <= i) ? (int16_t)(src_off + symi(j, i)) : (int16_t)-1;
}
static void map_upper_out(int n, int src_off, int16_t *map) {
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) map[i * n + j] = (j >= i) ? (int16_t)(src_off + symi(i, j)) : (int16_t)-1;
}

struct FieldTarget { void *block; int block_elems; int src_elems; int16_t map[KB_MAX_DIM * KB_MAX_DIM]; };

// where an INPUT field lands; p_rows = rows the caller's H/R has
static int input_target(Batch &b, int field, int p_rows, FieldTarget &t) {
    const int n = b.n, m = b.m;
    const Layout &L = b.L;
    switch (field) {
    case KB_X:
        t.block = b.d_state; t.block_elems = L.st_elems; t.src_elems = n;
        map_dense(1, n, L.st_vec, n, t.map);
        return KB_OK;
    case KB_P:
        t.block = b.d_state; t.block_elems = L.st_elems; t.src_elems = n * n;
        if (L.st_mat_full) map_dense(n, n, L.st_mat, n, t.map); else map_sym_in(n, L.st_mat, t.map);
        return KB_OK;
    case KB_F:
        t.block = b.d_model; t.block_elems = L.mo_elems; t.src_elems = n * n;
        map_dense(n, n, L.mo_F, n, t.map);
        return KB_OK;
    case KB_G:
        if (m <= 0) { set_error("batch was created with m = 0: no input control"); return KB_ERR_INVALID; }
        t.block = b.d_model; t.block_elems = L.mo_elems; t.src_elems = n * m;
        map_dense(n, m, L.mo_G, m, t.map);
        return KB_OK;
    case KB_H:
        if (p_rows < 1 || p_rows > b.pmax) { set_error("H has %d rows; the batch was created for at most %d", p_rows, b.pmax); return KB_ERR_DIMS; }
        t.block = b.d_model; t.block_elems = L.mo_elems; t.src_elems = p_rows * n;
        map_dense(p_rows, n, L.mo_H, n, t.map);
        return KB_OK;
    case KB_Q: {
        const int q = (b.kind == KB_HYBRID) ? m : n;
        if (q <= 0) { set_error("no process noise dimension"); return KB_ERR_INVALID; }
        t.block = b.d_model; t.block_elems = L.mo_elems; t.src_elems = q * q;
        map_sym_in(q, L.mo_Q, t.map);
        return KB_OK;
    }
    case KB_R:
        if (p_rows < 1 || p_rows > b.pmax) { set_error("R is %dx%d; the batch was created for at most %d", p_rows, p_rows, b.pmax); return KB_ERR_DIMS; }
        t.block = b.d_model; t.block_elems = L.mo_elems; t.src_elems = p_rows * p_rows;
        map_sym_in(p_rows, L.mo_R, t.map);
        return KB_OK;
    }
    set_error("field %d is not an input field", field);
    return KB_ERR_INVALID;
}

template <typename T>
__global__ void pack_planar_kernel(const T *__restrict__ src, int64_t ld, int src_elems, int64_t N,
                                   T *__restrict__ dst, int dst_elems, const int16_t *__restrict__ map) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    T *d = dst + (i / KB_TILE) * ((int64_t)KB_TILE * dst_elems) + (i % KB_TILE);
    for (int e = 0; e < src_elems; e++) {
        const int de = map[e];
        if (de >= 0) d[(int64_t)de * KB_TILE] = src[(int64_t)e * ld + i];
    }
}

void fill_step_args(const Batch &b, StepArgs &a) {
    memset(&a, 0, sizeof(a));
    a.state = b.d_state; a.est = b.d_est; a.model = b.d_model; a.status = b.d_status; a.lag = b.d_lag;
    a.N = b.N; a.ntiles = b.ntiles; a.nsteps = 1;
    a.mo_ts = b.per_filter_model ? (int64_t)KB_TILE * b.L.mo_elems : 0;
    a.stream_state = b.block_bytes(b.L.st_elems) > KB_MALL_BYTES ? 1 : 0;
    a.n = b.n; a.p = b.p; a.m = b.m; a.pmax = b.pmax; a.L = b.L; a.flags = b.flags;
    a.need_ctrl = b.need_ctrl; a.rinv_p = b.rinv_p; a.sqrt_p = b.sqrt_p; a.srif_tri = b.srif_tri;
    a.srif_leftover = b.srif_leftover; a.srif_dense_fail = b.d_srif_fail; a.srif_dense = b.d_srif_dense;
    a.ekf = b.ekf; a.snc = b.snc; a.predict = (b.kind == KB_VANILLA_PREDICT);
    a.noise_kind = b.noise_kind; a.seed = b.seed; a.epoch = b.epoch; a.step0 = b.step; a.first_filter = 0;
    a.bn_proc = b.d_bn_proc; a.bn_meas = b.d_bn_meas; a.bn_p = b.bn_p;
}

// ---- kb_last_kernel: which instantiation(s) served the last step (KB_LAUNCH, kb_internal.h) ----------------------------------------
static thread_local const std::type_info *g_noted[KB_MAX_NOTED];
static thread_local int g_nnoted = 0;
void note_kernel(const std::type_info &tag) { if (g_nnoted < KB_MAX_NOTED) g_noted[g_nnoted++] = &tag; }
void begin_kernel_record() { g_nnoted = 0; }
void end_kernel_record(Batch &b) {
    b.n_last_kernels = g_nnoted;
    for (int i = 0; i < g_nnoted; i++) b.last_kernels[i] = g_noted[i];
}

static int launch_step(Batch &b, const StepArgs &a, bool fused) {
    begin_kernel_record();
    int rc;
    switch (b.kind) {
    case KB_VANILLA:
    case KB_VANILLA_PREDICT: rc = launch_vanilla(b, a, fused); break;
    case KB_SQUAREROOT: rc = launch_squareroot(b, a, fused); break;
    case KB_INFORMATION: rc = launch_information(b, a); break;
    default:
        set_error("kind %d has no LDKF update", b.kind);
        return KB_ERR_UNSUPPORTED;
    }
    end_kernel_record(b);
    return rc;
}

// host [N][rows] -> AoSoA staging block with `rows` elements per filter
int stage_host_vec(Batch &b, const double *host, int rows, void **dblock, int slot, const void **tile) {
    if (b.N <= KB_TILE) {   // one tile: the host transposes straight into the pinned buffer, no copy, no pack launch
        int rc = ensure_pin(b);
        if (rc) return rc;
        char *dst = (char *)b.h_pin + (size_t)slot * KB_PIN_TILE_BYTES;
        for (int e = 0; e < rows; e++)
            for (int64_t f = 0; f < b.N; f++) {
                if (b.dtype == KB_F64) ((double *)dst)[e * KB_TILE + f] = host[f * rows + e];
                else ((float *)dst)[e * KB_TILE + f] = (float)host[f * rows + e];
            }
        *tile = (const char *)b.d_pin + (size_t)slot * KB_PIN_TILE_BYTES;
        return KB_OK;
    }
    const size_t bytes = (size_t)b.N * rows * sizeof(double);
    int rc = ensure_stage(b, bytes);
    if (rc) return rc;
    KB_HIP(hipMemcpyAsync(b.d_stage, host, bytes, hipMemcpyHostToDevice, b.stream));
    if (!*dblock) {
        KB_HIP(dev_alloc(dblock, b.block_bytes(KB_MAX_DIM)));
        KB_HIP(hipMemsetAsync(*dblock, 0, b.block_bytes(KB_MAX_DIM), b.stream));
    }
    int16_t map[KB_MAX_DIM * KB_MAX_DIM];
    map_dense(1, rows, 0, rows, map);
    *tile = *dblock;
    return launch_pack(b, b.d_stage, rows, b.N, false, *dblock, rows, map);
}

}  // namespace kb

using namespace kb;

// =====================================================================================
// lifetime
// =====================================================================================
extern "C" {

const char *kb_last_error(void) { return g_err; }

// "vanilla_reg_kernel<double, 6, 3, 0, false, false, false, false, false, false>" (two kernels of one step joined by " + "), "" before
// the first step.  The string lives in the handle until the next call of this function on it.
const char *kb_last_kernel(kb_batch *b) {
    if (!b) return "";
    b->last_kernel_text.clear();
    for (int i = 0; i < b->n_last_kernels; i++) {
        int st = 0;
        char *d = abi::__cxa_demangle(b->last_kernels[i]->name(), nullptr, nullptr, &st);
        std::string t = (st == 0 && d) ? d : b->last_kernels[i]->name();
        free(d);
        // "kb::KernelTag<&(void kb::NAME<ARGS>(kb::StepArgs))>" -> "NAME<ARGS>"
        const size_t v = t.find("(void kb::");
        if (v != std::string::npos) t = t.substr(v + 10);
        const size_t e = t.rfind(">(");
        if (e != std::string::npos) t = t.substr(0, e + 1);
        else { const size_t e2 = t.rfind("("); if (e2 != std::string::npos) t = t.substr(0, e2); }
        if (i) b->last_kernel_text += " + ";
        b->last_kernel_text += t;
    }
    return b->last_kernel_text.c_str();
}
const char *kb_version(void) { return "gokalman_amd 0.1 (gfx950)"; }

int kb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int kb_create(kb_batch **out, int kind, int n, int p, int m, int64_t nfilters, int dtype, int device, unsigned flags) {
    if (!out) { set_error("out is NULL"); return KB_ERR_INVALID; }
    *out = nullptr;
    if (kind < KB_VANILLA || kind > KB_BATCH_LS) { set_error("unknown filter kind %d", kind); return KB_ERR_INVALID; }
    if (n < 1 || n > KB_MAX_DIM || p < 1 || p > KB_MAX_DIM || m < 0 || m > KB_MAX_DIM) {
        set_error("dimensions out of range: n=%d p=%d m=%d (1..%d)", n, p, m, KB_MAX_DIM);
        return KB_ERR_DIMS;
    }
    if (nfilters < 1) { set_error("nfilters must be >= 1"); return KB_ERR_INVALID; }
    if (dtype != KB_F64 && dtype != KB_F32) { set_error("unknown dtype %d", dtype); return KB_ERR_INVALID; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        set_error("no HIP device visible: gokalman_amd has no CPU fallback");
        return KB_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d visible)", device, ndev); return KB_ERR_INVALID; }
    kb_batch *b = new kb_batch();
    b->kind = kind; b->n = n; b->pmax = p; b->p = p; b->m = m; b->dtype = dtype; b->device = device; b->flags = flags;
    b->N = nfilters; b->ntiles = (nfilters + KB_TILE - 1) / KB_TILE;
    b->L = make_layout(kind, n, p, m, flags);
    b->locked = is_nldkf(kind) ? 1 : 0;
    int rc = KB_OK;
    auto fail = [&](int code) { kb_destroy(b); return code; };
    if ((rc = use_device(*b))) return fail(rc);
#define KB_TRY(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(hip_fail(e__, #call)); } while (0)
    KB_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    KB_TRY(dev_alloc(&b->d_state, b->block_bytes(b->L.st_elems)));
    KB_TRY(dev_alloc(&b->d_state0, b->block_bytes(b->L.st_elems)));
    KB_TRY(dev_alloc(&b->d_model, b->block_bytes(b->L.mo_elems)));
    KB_TRY(dev_alloc((void **)&b->d_status, (size_t)b->ntiles * KB_TILE * sizeof(uint32_t)));
    if (b->ntiles == 1) {   // kf.step of a drop-in (one-tile) batch is read by the host without touching the device: pinned, device-mapped
        KB_TRY(hipHostMalloc((void **)&b->h_lag, KB_TILE * sizeof(uint32_t), hipHostMallocMapped));
        memset(b->h_lag, 0, KB_TILE * sizeof(uint32_t));
        KB_TRY(hipHostGetDevicePointer((void **)&b->d_lag, b->h_lag, 0));
    } else {
        KB_TRY(dev_alloc((void **)&b->d_lag, (size_t)b->ntiles * KB_TILE * sizeof(uint32_t)));
        KB_TRY(hipMemsetAsync(b->d_lag, 0, (size_t)b->ntiles * KB_TILE * sizeof(uint32_t), b->stream));
    }
    KB_TRY(hipMemsetAsync(b->d_state, 0, b->block_bytes(b->L.st_elems), b->stream));
    KB_TRY(hipMemsetAsync(b->d_state0, 0, b->block_bytes(b->L.st_elems), b->stream));
    KB_TRY(hipMemsetAsync(b->d_model, 0, b->block_bytes(b->L.mo_elems), b->stream));
    KB_TRY(hipMemsetAsync(b->d_status, 0, (size_t)b->ntiles * KB_TILE * sizeof(uint32_t), b->stream));
    if (kind == KB_SRIF) {
        KB_TRY(hipHostMalloc((void **)&b->h_srif_fail, sizeof(uint32_t), hipHostMallocMapped));
        *b->h_srif_fail = 0u;
        KB_TRY(hipHostGetDevicePointer((void **)&b->d_srif_fail, b->h_srif_fail, 0));
        KB_TRY(dev_alloc((void **)&b->d_srif_dense, (size_t)b->ntiles * 2 * sizeof(uint32_t)));
        KB_TRY(hipMemsetAsync(b->d_srif_dense, 0, (size_t)b->ntiles * 2 * sizeof(uint32_t), b->stream));
    }
    if (flags & KB_FLAG_FULL_ESTIMATE) {
        KB_TRY(dev_alloc(&b->d_est, b->block_bytes(b->L.es_elems)));
        KB_TRY(hipMemsetAsync(b->d_est, 0, b->block_bytes(b->L.es_elems), b->stream));
    }
    KB_TRY(hipStreamSynchronize(b->stream));
#undef KB_TRY
    *out = b;
    return KB_OK;
}

void kb_destroy(kb_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    void *ptrs[] = {b->d_state, b->d_state0, b->d_est, b->d_model, b->d_status,
                    b->d_stage, b->d_y, b->d_u, b->d_y2, b->d_xp, b->d_flags, b->d_mc, b->d_chi_table, b->d_ctrl, b->d_bn_proc, b->d_bn_meas, b->d_traj,
                    b->h_lag ? nullptr : (void *)b->d_lag, (void *)b->d_srif_dense};
    for (void *p : ptrs)
        if (p) (void)dev_free(p);
    if (b->h_lag) (void)hipHostFree(b->h_lag);
    if (b->h_srif_fail) (void)hipHostFree(b->h_srif_fail);
    if (b->h_pin) (void)hipHostFree(b->h_pin);
    for (hipEvent_t e : b->ev_heavy)
        if (e) (void)hipEventDestroy(e);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

// =====================================================================================
// inputs
// =====================================================================================
static int after_set(kb_batch *b, int field, int p_rows, const double *host_for_nil, int64_t host_elems) {
    if (field <= KB_R) b->have[field] = true;
    if (field == KB_H) b->p = p_rows;
    if (field == KB_R) b->r_p = p_rows;
    if (field == KB_G && !b->initialized && host_for_nil) {
        // needCtrl = !IsNil(G) at construction (vanilla.go:39, helper.go:49-62); the setter does not refresh it
        int nz = 0;
        for (int64_t i = 0; i < host_elems; i++) if (host_for_nil[i] != 0.0) { nz = 1; break; }
        b->need_ctrl = nz;
    }
    if (b->initialized && (field == KB_F || field == KB_Q || field == KB_R)) {
        int not_pd = 0;
        int rc = launch_refresh(*b, field, &not_pd);
        if (rc) return rc;
        if (not_pd) { set_error("matrix is not positive definite (Cholesky failed for %d filter(s))", not_pd); return KB_ERR_NOT_PD; }
    }
    return KB_OK;
}

int kb_set(kb_batch *b, int field, const double *host, int64_t count, int broadcast, int p_rows) {
    if (!b || !host) { set_error("null argument"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    if (broadcast ? count != 1 : count != b->N) {
        set_error("count must be 1 with broadcast or N=%lld without (got %lld)", (long long)b->N, (long long)count);
        return KB_ERR_INVALID;
    }
    if (b->initialized && (field == KB_X || field == KB_P)) {
        set_error("x0/P0 are constructor arguments: set them before kb_init");
        return KB_ERR_INVALID;
    }
    FieldTarget t;
    if ((rc = input_target(*b, field, p_rows, t))) return rc;
    const size_t bytes = (size_t)count * t.src_elems * sizeof(double);
    if ((rc = ensure_stage(*b, bytes))) return rc;
    KB_HIP(hipMemcpyAsync(b->d_stage, host, bytes, hipMemcpyHostToDevice, b->stream));
    if ((rc = launch_pack(*b, b->d_stage, t.src_elems, count, broadcast != 0, t.block, t.block_elems, t.map))) return rc;
    if (t.block == b->d_model && field >= 0 && field < 32) {   // does every filter still have the same model? (StepArgs::mo_ts)
        if (broadcast) b->per_filter_model &= ~(1u << field); else b->per_filter_model |= 1u << field;
    }
    if (field == KB_X || field == KB_P) {
        if (broadcast) b->per_filter_init &= ~(1u << field); else b->per_filter_init |= 1u << field;
    }
    if ((rc = after_set(b, field, p_rows, host, count * t.src_elems))) { (void)hipStreamSynchronize(b->stream); return rc; }
    KB_HIP(hipStreamSynchronize(b->stream));
    return KB_OK;
}

int kb_set_dev(kb_batch *b, int field, const void *src, int64_t ld, int p_rows) {
    if (!b || !src) { set_error("null argument"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    if (ld < b->N) { set_error("ld (%lld) < N (%lld)", (long long)ld, (long long)b->N); return KB_ERR_INVALID; }
    if (b->initialized && (field == KB_X || field == KB_P)) {
        set_error("x0/P0 are constructor arguments: set them before kb_init");
        return KB_ERR_INVALID;
    }
    FieldTarget t;
    if ((rc = input_target(*b, field, p_rows, t))) return rc;
    // the element map travels through the staging buffer (device-visible)
    if ((rc = ensure_stage(*b, sizeof(t.map)))) return rc;
    KB_HIP(hipMemcpyAsync(b->d_stage, t.map, sizeof(t.map), hipMemcpyHostToDevice, b->stream));
    KB_HIP(hipStreamSynchronize(b->stream));  // t.map is on the stack
    const unsigned blocks = (unsigned)((b->N + 255) / 256);
    if (b->dtype == KB_F64)
        hipLaunchKernelGGL(pack_planar_kernel<double>, dim3(blocks), dim3(256), 0, b->stream, (const double *)src, ld,
                           t.src_elems, b->N, (double *)t.block, t.block_elems, (const int16_t *)b->d_stage);
    else
        hipLaunchKernelGGL(pack_planar_kernel<float>, dim3(blocks), dim3(256), 0, b->stream, (const float *)src, ld,
                           t.src_elems, b->N, (float *)t.block, t.block_elems, (const int16_t *)b->d_stage);
    KB_HIP(hipGetLastError());
    if (t.block == b->d_model && field >= 0 && field < 32) b->per_filter_model |= 1u << field;   // per-filter by construction (StepArgs::mo_ts)
    if (field == KB_X || field == KB_P) b->per_filter_init |= 1u << field;
    if (field == KB_G && !b->initialized) b->need_ctrl = 1;  // device-side G: assumed non-nil
    if ((rc = after_set(b, field, p_rows, nullptr, 0))) return rc;
    KB_HIP(hipStreamSynchronize(b->stream));  // the map buffer may be reused by the next call
    return KB_OK;
}

int kb_init(kb_batch *b) {
    if (!b) { set_error("null batch"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    if (b->initialized) { set_error("kb_init called twice"); return KB_ERR_INVALID; }
    const bool need_fh = is_ldkf(b->kind);
    const char *names[] = {"x0", "P0", "F", "G", "H", "Q", "R"};
    for (int f = KB_X; f <= KB_R; f++) {
        bool required = (f == KB_X || f == KB_P || f == KB_R);
        if (b->kind == KB_BATCH_LS) required = (f == KB_R);  // NewBatchKF(numMeasurements, noise): Lambda, N start at zero
        if (need_fh && (f == KB_F || f == KB_H || f == KB_Q)) required = true;
        if (required && !b->have[f]) { set_error("kb_init: %s has not been set", names[f]); return KB_ERR_INVALID; }
    }
    int not_pd = 0;
    if ((rc = launch_init(*b, &not_pd))) return rc;
    if (not_pd) {
        set_error("constructor: matrix is not positive definite (Cholesky failed for %d filter(s))", not_pd);
        return KB_ERR_NOT_PD;
    }
    KB_HIP(hipMemcpyAsync(b->d_state0, b->d_state, b->block_bytes(b->L.st_elems), hipMemcpyDeviceToDevice, b->stream));
    KB_HIP(hipStreamSynchronize(b->stream));
    b->initialized = true;
    b->step = 0;
    return KB_OK;
}

int kb_reset(kb_batch *b) {
    if (!b || !b->initialized) { set_error("batch not initialised"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    KB_HIP(hipMemcpyAsync(b->d_state, b->d_state0, b->block_bytes(b->L.st_elems), hipMemcpyDeviceToDevice, b->stream));
    if (b->d_est) KB_HIP(hipMemsetAsync(b->d_est, 0, b->block_bytes(b->L.es_elems), b->stream));
    KB_HIP(hipMemsetAsync(b->d_status, 0, (size_t)b->ntiles * KB_TILE * sizeof(uint32_t), b->stream));
    if (!b->h_lag) KB_HIP(hipMemsetAsync(b->d_lag, 0, (size_t)b->ntiles * KB_TILE * sizeof(uint32_t), b->stream));
    KB_HIP(hipStreamSynchronize(b->stream));
    if (b->h_lag) memset(b->h_lag, 0, KB_TILE * sizeof(uint32_t));
    b->step = 0;   // vanilla.go:123
    b->calls++;
    b->epoch++;  // AWGN.Reset re-seeds (noise.go:145-146)
    if (is_nldkf(b->kind)) { b->locked = 1; b->snc = 0; b->srif_tri = 1; b->srif_leftover = 0; }
    if (b->h_srif_fail) *b->h_srif_fail = 0u;
    return KB_OK;
}

// =====================================================================================
// the hot path
// =====================================================================================
static int check_update_dims(kb_batch *b, int meas_rows, int ctrl_rows, bool have_ctrl) {
    // vanilla.go:129-135 (same in squareroot.go:131-136, information.go:154-160)
    if (b->need_ctrl) {
        if (!have_ctrl) ctrl_rows = 0;
        if (ctrl_rows != b->m) {
            set_error("dimensions must agree: control (u)(%dx...) G(...x%d)", ctrl_rows, b->m);
            return KB_ERR_DIMS;
        }
    }
    if (meas_rows != b->p) {
        set_error("dimensions must agree: measurement (y)(%dx...) H(%dx...)", meas_rows, b->p);
        return KB_ERR_DIMS;
    }
    return KB_OK;
}

static int check_batch_noise(kb_batch *b, int nsteps) {
    if (b->noise_kind != KB_NOISE_BATCH) return KB_OK;
    // the largest kf.step any filter of the batch may be at (exact for one-tile batches, whose failed-step counts the host sees)
    int64_t step = b->step;
    if (b->h_lag) {
        if (hipStreamQuery(b->stream) != hipSuccess) KB_HIP(hipStreamSynchronize(b->stream));   // the words of an asynchronous step still running
        uint32_t least = b->h_lag[0];
        for (int64_t i = 1; i < b->N; i++) least = b->h_lag[i] < least ? b->h_lag[i] : least;
        step -= least;
    }
    const int64_t last = step + nsteps - 1;
    if (last >= b->bn_nproc) { set_error("no process noise defined at step k=%lld", (long long)(step < b->bn_nproc ? b->bn_nproc : step)); return KB_ERR_INVALID; }
    if (last >= b->bn_nmeas) { set_error("no measurement noise defined at step k=%lld", (long long)(step < b->bn_nmeas ? b->bn_nmeas : step)); return KB_ERR_INVALID; }
    if (b->bn_p != b->p) { set_error("dimensions must agree: measurement noise(%dx...) H(%dx...)", b->bn_p, b->p); return KB_ERR_DIMS; }
    return KB_OK;
}

static int ready_ldkf(kb_batch *b) {
    if (!b) { set_error("null batch"); return KB_ERR_INVALID; }
    if (!b->initialized) { set_error("kb_init has not been called"); return KB_ERR_INVALID; }
    if (!is_ldkf(b->kind)) { set_error("kb_update is the LDKF entry point; use kb_update_nl for SRIF / Hybrid"); return KB_ERR_INVALID; }
    if (b->kind == KB_INFORMATION && b->rinv_p != 1 && b->rinv_p != b->p) {
        // information.go:197-203: H^T R^-1 with the R^-1 cached at construction (SetNoise never refreshes it, :136-138);
        // only a 1x1 R^-1 takes the scalar branch, anything else is a mat64 shape panic in the reference
        set_error("matrix: dimension mismatch: H^T(...x%d) R^-1(%dx%d) (R^-1 is cached at construction)", b->p, b->rinv_p, b->rinv_p);
        return KB_ERR_DIMS;
    }
    return use_device(*b);
}

static int update_host(kb_batch *b, const double *meas, int meas_rows, const double *ctrl, int ctrl_rows, bool sync) {
    int rc = ready_ldkf(b);
    if (rc) return rc;
    if (!meas) { set_error("measurement is NULL"); return KB_ERR_INVALID; }
    if ((rc = check_update_dims(b, meas_rows, ctrl_rows, ctrl != nullptr))) return rc;
    if ((rc = check_batch_noise(b, 1))) return rc;
    const void *ytile = nullptr, *utile = nullptr;
    if ((rc = stage_host_vec(*b, meas, meas_rows, &b->d_y, 0, &ytile))) return rc;
    if (b->need_ctrl) {
        // the staging buffer is reused: order the two packs on the stream
        if ((rc = stage_host_vec(*b, ctrl, ctrl_rows, &b->d_u, 1, &utile))) return rc;
    }
    StepArgs a;
    fill_step_args(*b, a);
    a.y = ytile; a.y_es = KB_TILE; a.y_ts = (int64_t)KB_TILE * meas_rows; a.y_step = 0;
    if (b->need_ctrl) { a.u = utile; a.u_es = KB_TILE; a.u_ts = (int64_t)KB_TILE * ctrl_rows; a.u_step = 0; }
    if ((rc = launch_step(*b, a, false))) return rc;
    if (sync) KB_HIP(hipStreamSynchronize(b->stream));
    b->step++;
    b->calls++;
    return KB_OK;
}

int kb_update(kb_batch *b, const double *meas, int meas_rows, const double *ctrl, int ctrl_rows) {
    return update_host(b, meas, meas_rows, ctrl, ctrl_rows, true);
}

// Update + the Estimate it returns (vanilla.go:216-218) with ONE synchronisation: the step and the snapshot are enqueued back to
// back and the host waits once (kb_update followed by kb_get_estimate waits twice: 32 us per step for one filter, 22 us this way)
int kb_update_estimate(kb_batch *b, const double *meas, int meas_rows, const double *ctrl, int ctrl_rows, int64_t first, int64_t count,
                       kb_estimate_view *view) {
    if (!view) { set_error("null argument"); return KB_ERR_INVALID; }
    int rc = update_host(b, meas, meas_rows, ctrl, ctrl_rows, false);
    if (rc) return rc;
    rc = kb_get_estimate(b, first, count, view);
    if (rc) (void)hipStreamSynchronize(b->stream);   // the step is enqueued either way: leave no work behind on an error
    return rc;
}

static int update_dev_common(kb_batch *b, const void *meas, int64_t ld_meas, const void *ctrl, int64_t ld_ctrl,
                             int nsteps, bool fused) {
    int rc = ready_ldkf(b);
    if (rc) return rc;
    if (!meas && b->kind != KB_VANILLA_PREDICT) { set_error("measurement is NULL"); return KB_ERR_INVALID; }
    if (ld_meas < b->N) { set_error("ld_meas (%lld) < N (%lld)", (long long)ld_meas, (long long)b->N); return KB_ERR_INVALID; }
    if (b->need_ctrl && (!ctrl || ld_ctrl < b->N)) { set_error("control required (needCtrl) with ld_ctrl >= N"); return KB_ERR_DIMS; }
    if (nsteps < 1) { set_error("nsteps must be >= 1"); return KB_ERR_INVALID; }
    if ((rc = check_batch_noise(b, nsteps))) return rc;
    StepArgs a;
    fill_step_args(*b, a);
    a.nsteps = nsteps;
    a.y = meas; a.y_es = ld_meas; a.y_ts = KB_TILE; a.y_step = (int64_t)b->p * ld_meas;
    if (b->need_ctrl) { a.u = ctrl; a.u_es = ld_ctrl; a.u_ts = KB_TILE; a.u_step = (int64_t)b->m * ld_ctrl; }
    const bool vanilla = b->kind == KB_VANILLA || b->kind == KB_VANILLA_PREDICT;
    const bool have_fused = (vanilla && vanilla_fused_ok(*b, a)) || (b->kind == KB_SQUAREROOT && squareroot_fused_ok(*b, a));
    if (fused && !(b->flags & KB_FLAG_STATEMENT_KERNELS) && !have_fused) {
        // no time-fused register kernel for this kind / shape / noise: one single-step launch per step, back to back on the stream
        // (the multi-step statement kernel is 16-30x slower than that)
        for (int t = 0; t < nsteps; t++) {
            StepArgs s = a;
            s.nsteps = 1;
            s.step0 = a.step0 + t;
            s.y = meas ? (const char *)meas + (size_t)t * (size_t)a.y_step * b->esize() : nullptr;
            if (b->need_ctrl) s.u = (const char *)ctrl + (size_t)t * (size_t)a.u_step * b->esize();
            if ((rc = launch_step(*b, s, false))) return rc;
        }
    } else if ((rc = launch_step(*b, a, fused))) return rc;
    b->step += nsteps;
    b->calls++;
    return KB_OK;
}

int kb_update_dev(kb_batch *b, const void *meas, int64_t ld_meas, const void *ctrl, int64_t ld_ctrl) {
    return update_dev_common(b, meas, ld_meas, ctrl, ld_ctrl, 1, false);
}

int kb_update_steps_dev(kb_batch *b, const void *meas, int64_t ld_meas, const void *ctrl, int64_t ld_ctrl, int nsteps) {
    return update_dev_common(b, meas, ld_meas, ctrl, ld_ctrl, nsteps, true);
}

// =====================================================================================
// results
// =====================================================================================
// Where an OUTPUT field lives: the AoSoA block, its element count, the element map to the host layout and the number of
// doubles per filter on the host side.  Lazy getters (SquareRoot / Information / SRIF covariance, Information / SRIF
// state) enqueue their materialise kernel here, into the handle's scratch block.
struct OutputSource { const void *block = nullptr; int block_elems = 0, out_elems = 0; int16_t map[KB_MAX_DIM * KB_MAX_DIM]; };
static int resolve_output(kb_batch *b, int field, OutputSource &o) {
    int rc = KB_OK;
    const int n = b->n, p = b->p;
    const Layout &L = b->L;
    int16_t (&map)[KB_MAX_DIM * KB_MAX_DIM] = o.map;
    const void *&block = o.block; int &block_elems = o.block_elems, &out_elems = o.out_elems;
    const bool full = (b->flags & KB_FLAG_FULL_ESTIMATE) != 0;
    const bool lazy = (b->kind == KB_SQUAREROOT || b->kind == KB_INFORMATION || b->kind == KB_SRIF || b->kind == KB_BATCH_LS);
    auto need_full = [&]() -> int {
        if (!full) { set_error("field %d needs a batch created with KB_FLAG_FULL_ESTIMATE", field); return KB_ERR_INVALID; }
        return KB_OK;
    };
    void *tmp = nullptr;
    switch (field) {
    case KB_X: case KB_RAW_VEC: case KB_STATE:
        if (field == KB_STATE && (b->kind == KB_INFORMATION || b->kind == KB_SRIF || b->kind == KB_BATCH_LS)) {
            if ((rc = ensure_xp(*b))) return rc;
            tmp = b->d_xp;
            if ((rc = launch_materialise(*b, b->d_state, false, tmp))) return rc;
            block = tmp; block_elems = n + tri(n); out_elems = n; map_dense(1, n, 0, n, map);
        } else {
            block = b->d_state; block_elems = L.st_elems; out_elems = n; map_dense(1, n, L.st_vec, n, map);
        }
        break;
    case KB_P: case KB_COVAR: case KB_PRED_COVAR:
        if (field == KB_PRED_COVAR && (rc = need_full())) return rc;
        if (lazy) {
            if ((rc = ensure_xp(*b))) return rc;
            tmp = b->d_xp;
            const bool pred = field == KB_PRED_COVAR;
            if ((rc = launch_materialise(*b, pred ? b->d_est : b->d_state, pred, tmp))) return rc;
            block = tmp; block_elems = n + tri(n); out_elems = n * n; map_sym_out(n, n, map);
        } else if (field == KB_PRED_COVAR) {
            block = b->d_est; block_elems = L.es_elems; out_elems = n * n; map_sym_out(n, L.es_ppred, map);
        } else {
            block = b->d_state; block_elems = L.st_elems; out_elems = n * n; map_sym_out(n, L.st_mat, map);
        }
        break;
    case KB_RAW_MAT: case KB_RAW_PRED_MAT: {
        const bool pred = field == KB_RAW_PRED_MAT;
        if (pred && (rc = need_full())) return rc;
        block = pred ? b->d_est : b->d_state; block_elems = pred ? L.es_elems : L.st_elems; out_elems = n * n;
        const int off = pred ? L.es_ppred : L.st_mat;
        if (L.st_mat_full) map_dense(n, n, off, n, map);
        else if (b->kind == KB_SQUAREROOT) { if (pred) map_upper_out(n, off, map); else map_lower_out(n, off, map); }
        else map_sym_out(n, off, map);
        break;
    }
    case KB_GAIN:
        if ((rc = need_full())) return rc;
        block = b->d_est; block_elems = L.es_elems; out_elems = n * p; map_dense(n, p, L.es_gain, b->pmax, map);
        break;
    case KB_INNOVATION:
        if (b->kind == KB_INFORMATION || b->kind == KB_SRIF) {
            block = b->d_state; block_elems = L.st_elems; out_elems = n; map_dense(1, n, L.st_vec, n, map);
        } else {
            if ((rc = need_full())) return rc;
            block = b->d_est; block_elems = L.es_elems; out_elems = p; map_dense(1, p, L.es_innov, p, map);
        }
        break;
    case KB_MEASUREMENT:
        if ((rc = need_full())) return rc;
        block = b->d_est; block_elems = L.es_elems; out_elems = p; map_dense(1, p, L.es_yhat, p, map);
        break;
    case KB_F: block = b->d_model; block_elems = L.mo_elems; out_elems = n * n; map_dense(n, n, L.mo_F, n, map); break;
    case KB_H: block = b->d_model; block_elems = L.mo_elems; out_elems = p * n; map_dense(p, n, L.mo_H, n, map); break;
    case KB_G:
        if (b->m <= 0) { set_error("no input control"); return KB_ERR_INVALID; }
        block = b->d_model; block_elems = L.mo_elems; out_elems = n * b->m; map_dense(n, b->m, L.mo_G, b->m, map); break;
    case KB_Q: {
        const int q = b->kind == KB_HYBRID ? b->m : n;
        block = b->d_model; block_elems = L.mo_elems; out_elems = q * q; map_sym_out(q, L.mo_Q, map); break;
    }
    case KB_R: block = b->d_model; block_elems = L.mo_elems; out_elems = p * p; map_sym_out(p, L.mo_R, map); break;
    default:
        set_error("unknown field %d", field);
        return KB_ERR_INVALID;
    }
    return KB_OK;
}

int kb_get(kb_batch *b, int field, double *host, int64_t first, int64_t count) {
    if (!b || !host) { set_error("null argument"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    if (first < 0 || count < 0 || first + count > b->N) { set_error("range [%lld,+%lld) outside the batch", (long long)first, (long long)count); return KB_ERR_INVALID; }
    if (count == 0) return KB_OK;
    OutputSource o;
    if ((rc = resolve_output(b, field, o))) return rc;
    const void *block = o.block; const int block_elems = o.block_elems, out_elems = o.out_elems;
    const int16_t *map = o.map;
    const size_t bytes = (size_t)count * out_elems * sizeof(double);
    if (bytes <= KB_PIN_OUT_BYTES && !ensure_pin(*b)) {   // small read-back: the kernel writes the pinned host buffer itself
        const size_t off = 3 * KB_PIN_TILE_BYTES;
        rc = launch_unpack(*b, block, block_elems, map, out_elems, (double *)((char *)b->d_pin + off), first, count);
        if (!rc) {
            const hipError_t e = hipStreamSynchronize(b->stream);
            if (e != hipSuccess) return hip_fail(e, "kb_get");
            memcpy(host, (const char *)b->h_pin + off, bytes);
        }
        return rc;
    }
    if ((rc = ensure_stage(*b, bytes))) return rc;
    rc = launch_unpack(*b, block, block_elems, map, out_elems, (double *)b->d_stage, first, count);
    if (!rc) {
        hipError_t e = hipMemcpyAsync(host, b->d_stage, bytes, hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e != hipSuccess) rc = hip_fail(e, "kb_get copy");
    }
    return rc;
}

int kb_get_dev(kb_batch *b, int field, void *dst, int64_t ld) {
    if (!b || !dst) { set_error("null argument"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    if (ld < b->N) { set_error("ld < N"); return KB_ERR_INVALID; }
    const int n = b->n;
    int16_t map[KB_MAX_DIM * KB_MAX_DIM];
    if (field == KB_RAW_VEC || (field == KB_STATE && b->kind != KB_INFORMATION && b->kind != KB_SRIF)) {
        map_dense(1, n, b->L.st_vec, n, map);
        return launch_unpack_planar(*b, b->d_state, b->L.st_elems, map, n, dst, ld);
    }
    if ((field == KB_COVAR || field == KB_RAW_MAT) && !b->L.st_mat_full &&
        (field == KB_RAW_MAT ? b->kind != KB_SQUAREROOT : (b->kind == KB_VANILLA || b->kind == KB_VANILLA_PREDICT || b->kind == KB_HYBRID))) {
        map_sym_out(n, b->L.st_mat, map);
        return launch_unpack_planar(*b, b->d_state, b->L.st_elems, map, n * n, dst, ld);
    }
    set_error("kb_get_dev: field %d not available on the device path for kind %d", field, b->kind);
    return KB_ERR_UNSUPPORTED;
}

// One snapshot of the Estimate of filters [first, first+count): every requested member is unpacked into ONE staging
// area (the pinned, device-mapped buffer when it fits: no copy at all) and handed over after ONE stream synchronisation.
// Completion of a flagged snapshot without the runtime: the kernel's last store is the sequence number, to pinned host memory; the
// host polls that word.  hipStreamSynchronize costs 11.5 us for an empty kernel on this stack, the polled word 7.5
// (scripts/diag_launch_latency.hip): 3-4 us per one-filter Update-with-estimate.  Bounded: after 200 us (a busy stream, or a launch
// that never ran) the caller falls back to hipStreamSynchronize, which also reports an asynchronous error.
static bool wait_snapshot(const Batch &b) {
    const uint32_t *flag = (const uint32_t *)((const char *)b.h_pin + KB_PIN_FLAG_OFF);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; spins++) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == b.pin_seq) return true;
        if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) return false;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        asm volatile("yield");
#endif
    }
}

int kb_get_estimate(kb_batch *b, int64_t first, int64_t count, kb_estimate_view *v) {
    if (!b || !v) { set_error("null argument"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    if (!b->initialized) { set_error("kb_init has not been called"); return KB_ERR_INVALID; }
    if (first < 0 || count < 0 || first + count > b->N) { set_error("range [%lld,+%lld) outside the batch", (long long)first, (long long)count); return KB_ERR_INVALID; }
    if (count == 0) return KB_OK;
    struct Want { int field; double *dst; size_t off, bytes; };
    Want want[6] = {{KB_STATE, v->state, 0, 0}, {KB_COVAR, v->covariance, 0, 0}, {KB_PRED_COVAR, v->pred_covariance, 0, 0},
                    {KB_GAIN, v->gain, 0, 0}, {KB_INNOVATION, v->innovation, 0, 0}, {KB_MEASUREMENT, v->measurement, 0, 0}};
    // sizes first (the staging area must exist before anything is enqueued into it)
    const int n = b->n, p = b->p;
    const bool info = b->kind == KB_INFORMATION || b->kind == KB_SRIF;
    const int elems[6] = {n, n * n, n * n, n * p, info ? n : p, p};
    size_t total = 0;
    for (int i = 0; i < 6; i++)
        if (want[i].dst) { want[i].off = total; want[i].bytes = (size_t)count * elems[i] * sizeof(double); total += want[i].bytes; }
    const size_t st_off = total;
    if (v->status) total += (size_t)count * sizeof(uint32_t);
    if (total == 0) return KB_OK;
    char *d_area = nullptr; const char *h_area = nullptr;
    const bool pinned = total <= KB_PIN_OUT_BYTES && !ensure_pin(*b);
    if (pinned) {
        d_area = (char *)b->d_pin + 3 * KB_PIN_TILE_BYTES;
        h_area = (const char *)b->h_pin + 3 * KB_PIN_TILE_BYTES;
    } else {
        if ((rc = ensure_stage(*b, total))) return rc;
        d_area = (char *)b->d_stage;
    }
    // One launch for every member that can be read in place; a lazy getter's member (SquareRoot / Information / SRIF covariance,
    // Information / SRIF state) first gets its materialise kernel, and pred_covar of those kinds -- which re-uses the scratch
    // block of state / covar -- goes into a second snapshot launch, stream-ordered behind the first.
    const bool lazy = (b->kind == KB_SQUAREROOT || b->kind == KB_INFORMATION || b->kind == KB_SRIF || b->kind == KB_BATCH_LS);
    // The status words travel (and are cleared) with the LAST launch: a member that cannot be delivered (pred_covariance of a
    // batch without KB_FLAG_FULL_ESTIMATE, a launch error) fails the call before anything has been cleared.
    const int status_pass = (lazy && v->pred_covariance) ? 1 : 0;
    // which pass is the last one that launches anything (status travels with status_pass; pass 1 only exists for lazy pred_covariance)
    const int last_pass = (lazy && v->pred_covariance) ? 1 : 0;
    const bool flagged = pinned && count <= KB_SNAP_FLAG_MAX;
    bool flag_armed = false;
    for (int pass = 0; pass < 2; pass++) {
        SnapArgs sa;
        sa.nmembers = 0;
        for (int i = 0; i < 6; i++) {
            if (!want[i].dst) continue;
            const bool second = lazy && want[i].field == KB_PRED_COVAR;
            if (second != (pass == 1)) continue;
            OutputSource o;
            if ((rc = resolve_output(b, want[i].field, o))) return rc;
            if (o.out_elems != elems[i]) { set_error("internal: member %d has %d elements, expected %d", want[i].field, o.out_elems, elems[i]); return KB_ERR_INVALID; }
            const int m = sa.nmembers++;
            sa.block[m] = o.block; sa.block_elems[m] = o.block_elems; sa.out_elems[m] = o.out_elems; sa.off[m] = (int64_t)want[i].off;
            memcpy(sa.map[m], o.map, sizeof(sa.map[m]));
        }
        const bool with_status = v->status && pass == status_pass;
        if (sa.nmembers == 0 && !with_status) continue;
        // the LAST launch of a small snapshot raises a completion word in the pinned block (see wait_snapshot)
        uint32_t *done = nullptr;
        if (flagged && pass == last_pass) { done = (uint32_t *)((char *)b->d_pin + KB_PIN_FLAG_OFF); ++b->pin_seq; }
        if ((rc = launch_snapshot(*b, sa, first, count, d_area, with_status ? b->d_status : nullptr, (int64_t)st_off, v->clear_status ? 1 : 0, done, b->pin_seq))) return rc;
        if (done) flag_armed = true;
    }
    if (pinned) {
        if (!(flag_armed && wait_snapshot(*b))) KB_HIP(hipStreamSynchronize(b->stream));
        else KB_HIP(hipGetLastError());   // the polled word says the snapshot ran; an error raised by this handle's launches is reported here, not later
        for (int i = 0; i < 6; i++)
            if (want[i].dst) memcpy(want[i].dst, h_area + want[i].off, want[i].bytes);
        if (v->status) memcpy(v->status, h_area + st_off, (size_t)count * sizeof(uint32_t));
    } else {
        for (int i = 0; i < 6; i++)
            if (want[i].dst) KB_HIP(hipMemcpyAsync(want[i].dst, d_area + want[i].off, want[i].bytes, hipMemcpyDeviceToHost, b->stream));
        if (v->status) KB_HIP(hipMemcpyAsync(v->status, d_area + st_off, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
        KB_HIP(hipStreamSynchronize(b->stream));
    }
    return KB_OK;
}

int kb_get_status(kb_batch *b, uint32_t *host, int64_t first, int64_t count) {
    if (!b || !host) { set_error("null argument"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    if (first < 0 || count < 0 || first + count > b->N) { set_error("range outside the batch"); return KB_ERR_INVALID; }
    KB_HIP(hipMemcpyAsync(host, b->d_status + first, (size_t)count * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    KB_HIP(hipStreamSynchronize(b->stream));
    return KB_OK;
}

int kb_clear_status(kb_batch *b) {
    if (!b) { set_error("null batch"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    KB_HIP(hipMemsetAsync(b->d_status, 0, (size_t)b->ntiles * KB_TILE * sizeof(uint32_t), b->stream));
    KB_HIP(hipStreamSynchronize(b->stream));
    return KB_OK;
}

int kb_is_within_nsigma(kb_batch *b, double nsigma, uint8_t *host, int64_t first, int64_t count) {
    if (!b || !host) { set_error("null argument"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    if (first < 0 || count < 0 || first + count > b->N) { set_error("range outside the batch"); return KB_ERR_INVALID; }
    const int n = b->n;
    void *tmp = nullptr;
    const void *xp = b->d_state;
    const bool lazy = (b->kind == KB_SQUAREROOT || b->kind == KB_INFORMATION || b->kind == KB_SRIF || b->kind == KB_BATCH_LS);
    if (lazy) {
        if ((rc = ensure_xp(*b))) return rc;
        tmp = b->d_xp;
        if ((rc = launch_materialise(*b, b->d_state, false, tmp))) return rc;
        xp = tmp;
    }
    if (!b->d_flags) KB_HIP(dev_alloc((void **)&b->d_flags, (size_t)b->ntiles * KB_TILE));
    uint8_t *d_out = b->d_flags;
    hipError_t e = hipSuccess;
    rc = launch_within_nsigma(*b, xp, nsigma, d_out);
    if (!rc) {
        e = hipMemcpyAsync(host, d_out + first, (size_t)count, hipMemcpyDeviceToHost, b->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
        if (e != hipSuccess) rc = hip_fail(e, "copy");
    }
    return rc;
}

// kf.step: a failed Update does not advance it (vanilla.go:164-167 returns before :218).  One-tile batches: exact, filter 0's
// counter (the host sees the failed-step counts); larger batches: the counter of a filter that never failed.
// The failed-step words are written by the step kernels: with asynchronous work outstanding on the handle's stream (kb_update_dev,
// kb_update_steps_dev) they are read only once it has drained.
static void settle_lag(const Batch &b) {
    if (b.h_lag && b.stream && hipStreamQuery(b.stream) != hipSuccess) {
        (void)hipSetDevice(b.device);
        (void)hipStreamSynchronize(b.stream);
    }
}
int64_t kb_step(const kb_batch *b) {
    if (!b) return -1;
    settle_lag(*b);
    return b->step - (b->h_lag ? (int64_t)b->h_lag[0] : 0);
}
int64_t kb_calls(const kb_batch *b) { return b ? b->calls : -1; }
int kb_filter_step(kb_batch *b, int64_t filter, int64_t *step) {
    if (!b || !step) { set_error("null argument"); return KB_ERR_INVALID; }
    if (filter < 0 || filter >= b->N) { set_error("filter %lld outside the batch", (long long)filter); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    uint32_t lag = 0;
    if (b->h_lag) {
        KB_HIP(hipStreamSynchronize(b->stream));
        lag = b->h_lag[filter];
    } else {
        KB_HIP(hipMemcpyAsync(&lag, b->d_lag + filter, sizeof(lag), hipMemcpyDeviceToHost, b->stream));
        KB_HIP(hipStreamSynchronize(b->stream));
    }
    *step = b->step - (int64_t)lag;
    return KB_OK;
}
int kb_need_ctrl(const kb_batch *b) { return b ? b->need_ctrl : 0; }
int kb_meas_dim(const kb_batch *b) { return b ? b->p : 0; }
int64_t kb_num_filters(const kb_batch *b) { return b ? b->N : 0; }
void *kb_stream(const kb_batch *b) { return b ? (void *)b->stream : nullptr; }
int kb_synchronize(kb_batch *b) {
    if (!b) { set_error("null batch"); return KB_ERR_INVALID; }
    int rc = use_device(*b);
    if (rc) return rc;
    KB_HIP(hipStreamSynchronize(b->stream));
    return KB_OK;
}

int kb_set_ekf(kb_batch *b, int enabled) {
    if (!b) { set_error("null batch"); return KB_ERR_INVALID; }
    if (b->kind == KB_HYBRID) b->ekf = enabled ? 1 : 0;  // srif.go:62-72: no-ops for SRIF
    return KB_OK;
}
int kb_ekf_enabled(const kb_batch *b) { return (b && b->kind == KB_HYBRID) ? b->ekf : 0; }

}  // extern "C"
