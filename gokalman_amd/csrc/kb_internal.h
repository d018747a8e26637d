// kb_internal.h -- host-side handle, HBM layout description and kernel launch entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <typeinfo>

#include "../../include/gokalman_amd.h"
#include "kb_device.h"

namespace kb {

// ---------------------------------------------------------------------------
// HBM layout of one batch (all blocks AoSoA-64, see kb_device.h).
//
//   state block : raw vector [n] | raw matrix (packed sym or full, per kind)
//   est   block : P- / S- / I- / Rbar | K [n][pmax] | innovation [pmax] | yhat [pmax] | dobs [pmax]   (FULL_ESTIMATE)
//   model block : F [n*n] | H [pmax*n] | Q packed | R packed(pmax) | G [n*m] | kind-specific
// ---------------------------------------------------------------------------
struct Layout {
    int n = 0, pmax = 0, m = 0;
    // state block
    int st_vec = 0, st_mat = 0, st_elems = 0;
    bool st_mat_full = false;  // full n*n (S, R factors) vs packed symmetric
    // est block
    int es_ppred = 0, es_gain = 0, es_innov = 0, es_yhat = 0, es_dobs = 0, es_elems = 0;
    // model block
    int mo_F = 0, mo_H = 0, mo_Q = 0, mo_R = 0, mo_G = 0;
    // derived (constructor / SetNoise / SetStateTransition products)
    int mo_LQ = 0, mo_LR = 0;                   // chol_L(Q) packed [tri(n)], chol_L(R) packed [tri(pmax)]  (AWGN, SQUAREROOT, SRIF)
    int mo_Finv = 0, mo_Qinv = 0, mo_Rinv = 0;  // INFORMATION: F^-1 full n*n; Q^-1, R^-1 packed symmetric [tri(n)], [tri(pmax)]
    int nq = 0;                                  // HYBRID: dimension of the SNC noise (Q is q x q, Gamma n x q)
    int mo_elems = 0;
};

struct Batch {
    int kind = 0, n = 0, pmax = 0, p = 0, m = 0, dtype = 0, device = 0;
    unsigned flags = 0;
    int64_t N = 0, ntiles = 0;
    Layout L;
    hipStream_t stream = nullptr;
    void *d_state = nullptr, *d_state0 = nullptr;  // current / initial estimate
    void *d_est = nullptr;
    void *d_model = nullptr;
    uint32_t *d_status = nullptr;
    const std::type_info *last_kernels[4] = {nullptr, nullptr, nullptr, nullptr};   // KB_LAUNCH / kb_last_kernel
    int n_last_kernels = 0;
    std::string last_kernel_text;
    // kf.step per filter.  The reference returns from a failed Update BEFORE `kf.step++` (vanilla.go:164-167 / :207-215 against
    // :218; srif.go:112-114; hybrid.go:150-152): such a filter's step counter -- the index of its BatchNoise vectors and the k of
    // its error messages -- falls one behind the number of Update calls.  lag[i] = calls that failed for filter i that way;
    // kf.step of filter i = step - lag[i].  One-tile batches (the drop-in use, N = 1) keep the words in pinned, device-mapped
    // host memory so that kb_step() is exact without a device read.
    uint32_t *d_lag = nullptr, *h_lag = nullptr;
    int64_t calls = 0;         // Update / Predict / Reset calls accepted so far (monotone: stale-view detection in the host mirrors)
    void *d_stage = nullptr;   // AoS staging for host <-> device transfers
    size_t stage_bytes = 0;
    void *d_y = nullptr, *d_u = nullptr;  // AoSoA staging of host measurements / controls
    // Batches of at most one tile (the reference's own use: one filter) skip the staging copies: the host writes / reads a
    // pinned, device-mapped buffer the kernels access directly (3 input tiles + one read-back area).
    unsigned per_filter_model = 0;   // bit f set: model field f (KB_F .. ) was last uploaded per filter; 0 = one model for the whole batch
    unsigned per_filter_init = 0;    // same for the constructor arguments x0 (KB_X) / P0 (KB_P): the Monte-Carlo ensembles are N copies of ONE filter
    void *h_pin = nullptr, *d_pin = nullptr;
    uint32_t pin_seq = 0;   // sequence number of the last flagged snapshot (the word at KB_PIN_FLAG_OFF of the pinned block)
    void *d_y2 = nullptr;
    void *d_xp = nullptr;      // cached getter scratch: materialised State() | Covariance() (x[n] | P packed) per filter
    uint8_t *d_flags = nullptr;  // cached IsWithinNsigma output
    double *d_mc = nullptr; size_t mc_bytes = 0;
    void *d_chi_table = nullptr; size_t chi_table_bytes = 0;   // kb_chisquare with one filter fanned out: K | inverse(P+) | inverse(S) per step (kb_chisq.hip)
    // Monte-Carlo runs kept on the device (kb_mc_run_ex with KB_MC_KEEP_RUNS): traj[(t * (n + p) + e) * mc_ld + run] = State() element
    // e < n, Measurement() element e - n of run `run` at step t, in the batch dtype
    void *d_traj = nullptr; size_t traj_bytes = 0; int mc_steps = 0, mc_p = 0; int64_t mc_ld = 0, mc_first_run = 0, mc_epoch = -1;
    bool initialized = false;
    bool have[8] = {false, false, false, false, false, false, false, false};  // KB_X..KB_R staged
    int need_ctrl = 0;
    int64_t step = 0;          // step calls since construction / Reset (kf.step of a filter that never failed)
    // KB_SRIF: some filter may hold a dense R although srif_tri is set (it failed the Update that followed a Predict()); the dense
    // kernel raises *h_srif_fail (pinned, device-mapped) when that happens, and the flag is dropped at the next host synchronisation
    // that finds the word at zero (kb_srif_pair.h)
    int srif_leftover = 0;
    uint32_t *h_srif_fail = nullptr, *d_srif_fail = nullptr;
    // one word per half-tile (32 filters), written by the dense Update kernel only: non-zero = a filter of this half-tile failed in it
    // and may still hold a dense R.  Both Update kernels of a step pick their half-tiles from these words, which the steady-state
    // kernel never writes: ownership of a half-tile cannot change between the two launches of one step.
    uint32_t *d_srif_dense = nullptr;
    int srif_tri = 1;      // KB_SRIF: R is upper triangular (constructor / measurement update wrote it; Predict() stores the full RBar)
    int rinv_p = 0;        // KB_INFORMATION: dimension R^-1 was computed for (stale-Rinv quirk)
    int sqrt_p = 0;        // KB_SQUAREROOT: dimension of chol(R)
    int r_p = 0;           // dimension of the R last given to kb_set
    void *d_ctrl = nullptr; size_t ctrl_bytes = 0;  // Monte-Carlo control sequence
    // NLDKF
    int ekf = 0, locked = 1, snc = 0;
    const void *ext_phi = nullptr, *ext_h = nullptr; int64_t ext_ld = 0;  // kb_prepare_dev: caller's planar Phi / Htilde (zero-copy)
    // noise
    int noise_kind = KB_NOISE_NOISELESS;
    uint64_t seed = 0;
    int64_t epoch = 0;
    void *d_bn_proc = nullptr, *d_bn_meas = nullptr; int bn_nproc = 0, bn_nmeas = 0, bn_p = 0;  // BatchNoise sequences (batch dtype)
    mutable hipEvent_t ev_heavy[2] = {nullptr, nullptr};   // HeavyScope: hand-over events between this handle's stream and the heavy stream
    size_t esize() const { return dtype == KB_F64 ? 8 : 4; }
    size_t block_bytes(int elems) const { return (size_t)ntiles * KB_TILE * (size_t)elems * esize(); }
};

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
// Device memory of a batch.  Plain hipMalloc / hipFree -- unless the environment holds KB_DEBUG_FENCE=1 (tests/test_fence_gpu.py): then every
// block is rounded up to the 2 MB the driver maps and handed out so that it ENDS where the mapping ends, and a kernel that reads or
// writes behind its last tile (a padded shape's stand-in element, a tail tile's masked lane) takes a memory fault instead of going unseen.
hipError_t dev_alloc(void **p, size_t bytes);

// Which kernel(s) served the last step of a handle (kb_last_kernel, a debugging / reporting aid: scripts/dispatch_table.py walks kind x n x
// p x noise x flags with it).  Every STEP-kernel launch goes through KB_LAUNCH, which notes the instantiation -- as the type_info of
// KernelTag<&kernel<...>>, demangled only when somebody asks -- in a per-thread record that launch_step / launch_nl move into the Batch.
template <auto K> struct KernelTag {};
constexpr int KB_MAX_NOTED = 4;
void note_kernel(const std::type_info &tag);
void begin_kernel_record();
void end_kernel_record(struct Batch &b);
#define KB_LAUNCH(kern, ...) do { ::kb::note_kernel(typeid(::kb::KernelTag<&kern>)); hipLaunchKernelGGL(kern, __VA_ARGS__); } while (0)
hipError_t dev_free(void *p);
#define KB_HIP(call)                                                          \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) return ::kb::hip_fail(e__, #call);             \
    } while (0)

// Arguments common to the step kernels (kept POD so it travels as kernarg).
struct StepArgs {
    void *state, *est, *model;
    uint32_t *status;
    uint32_t *lag;                               // per-filter failed-step count (Batch::d_lag)
    int64_t mo_ts;   // elements between the model blocks of consecutive tiles: 64 * L.mo_elems, or 0 when every filter has the SAME model
                     // (all model fields uploaded with broadcast = 1: every wave reads tile 0's block, which stays in the L2)
    const void *y; int64_t y_es, y_ts, y_step;   // element stride, tile stride, step stride (elements)
    const void *y2; int64_t y2_es, y2_ts, y2_step;   // NLDKF: computed observation
    const void *u; int64_t u_es, u_ts, u_step;
    const void *ext_phi, *ext_h; int64_t ext_ld;   // NLDKF zero-copy model (planar, element e of filter i at ptr[e*ld + i])
    int64_t ext_phi_step, ext_h_step;              // kb_update_nl_steps_dev: elements between the arrays of consecutive steps
    int64_t N, ntiles;
    int nsteps;
    int stream_state;                            // the state block cannot stay in the Infinity Cache: read / write it non-temporally (kb_vanilla_reg.h)
    int n, p, m, pmax;
    Layout L;
    unsigned flags;
    int need_ctrl;
    int rinv_p, sqrt_p;
    int srif_tri;                               // KB_SRIF: R is upper triangular (last writer: constructor or a measurement update)
    int srif_leftover; uint32_t *srif_dense_fail;   // KB_SRIF: see Batch::srif_leftover
    uint32_t *srif_dense;                           // KB_SRIF: Batch::d_srif_dense
    int ekf, snc, predict;
    int noise_kind; uint64_t seed; int64_t epoch; int64_t step0; int64_t first_filter;
    const void *bn_proc, *bn_meas; int bn_p;     // BatchNoise: [step][n], [step][bn_p]
};

// A step that fails the way the reference's Update returns (nil, err) BEFORE kf.step++ (vanilla.go:164-167, :207-215; srif.go:112-114;
// hybrid.go:150-152): the status bits are reported and the filter's step counter stays behind by the steps that were not applied.
__device__ __forceinline__ void fail_step(const StepArgs &a, int64_t fi, unsigned bits, unsigned nsteps = 1u) {
    atomicOr(a.status + fi, bits);
    a.lag[fi] += nsteps;   // the filter is owned by this lane: no atomic needed
}

// kb_pack.hip
struct SnapArgs {   // kernel argument of the one-launch Estimate snapshot (kb_get_estimate): at most 6 members
    const void *block[6];
    int block_elems[6], out_elems[6];
    int64_t off[6];
    int16_t map[6][KB_MAX_DIM * KB_MAX_DIM];
    int nmembers;
};
// done != nullptr (count <= KB_SNAP_FLAG_MAX): one block copies every filter, then stores `seq` to *done (pinned host memory, system scope)
int launch_snapshot(const Batch &b, const SnapArgs &sa, int64_t first, int64_t count, void *area, uint32_t *status, int64_t status_off, int clear,
                    uint32_t *done = nullptr, uint32_t seq = 0);
int launch_pack(const Batch &b, const void *src_aos, int src_elems, int64_t count, bool broadcast,
                void *dst_block, int dst_elems, const int16_t *map /* [src_elems] -> dst elem or -1 */);
int launch_unpack(const Batch &b, const void *src_block, int src_elems, const int16_t *map /* [dst_elems] -> src elem or -1 (0.0) */,
                  int dst_elems, double *dst_aos, int64_t first, int64_t count);
int launch_unpack_planar(const Batch &b, const void *src_block, int src_elems, const int16_t *map, int dst_elems,
                         void *dst, int64_t ld);
int launch_replicate(const Batch &dst, const void *src_block, int elems, int64_t src_filter, void *dst_block);
int launch_traj_unpack(const Batch &b, int64_t first, int64_t count, double *d_states, double *d_meas);

// kb_vanilla.hip
int launch_vanilla(const Batch &b, const StepArgs &a, bool fused);
bool vanilla_fused_ok(const Batch &b, const StepArgs &a);   // a time-fused register kernel exists for this batch
// kb_getters.hip (materialise State/Covariance for the kinds with a lazy getter)
int launch_materialise(const Batch &b, const void *state_block, bool pred, void *out_block /* x[n] | P packed */);
int launch_within_nsigma(const Batch &b, const void *xp_block, double nsigma, uint8_t *d_out);
int launch_smooth(const Batch &b, const void *xp_block, int xp_elems, int vec_off, int mat_off, const void *phis, int64_t ld, int steps,
                  void *x_out, void *P_out);
// kb_init.hip (constructor arithmetic per kind)
int launch_init(Batch &b, int *not_pd);
int launch_refresh(Batch &b, int field, int *not_pd);
// other kinds
int launch_squareroot_gen(const Batch &b, const StepArgs &a);
int launch_information(const Batch &b, const StepArgs &a);   // kb_information_reg.hip (falls back to _gen)
int launch_information_gen(const Batch &b, const StepArgs &a);
bool launch_information_split(const Batch &b, const StepArgs &a);       // kb_information_split12.hip: 6 < n <= 16, one filter over four / eight lanes
bool launch_information_split8(const Batch &b, const StepArgs &a);      // kb_information_split8.hip: n <= 8, p <= 4 (called by launch_information_split)
bool launch_information_split_full(const Batch &b, const StepArgs &a);  // kb_information_split12f.hip: the same with KB_FLAG_FULL_ESTIMATE
int launch_srif_gen(const Batch &b, const StepArgs &a);
int launch_hybrid_gen(const Batch &b, const StepArgs &a);
int launch_squareroot(const Batch &b, const StepArgs &a, bool fused);   // kb_squareroot_reg.hip (falls back to _gen)
bool launch_squareroot_split12(const Batch &b, const StepArgs &a);      // kb_squareroot_split12.hip: 6 < n <= 12, one filter over four lanes
bool launch_squareroot_split16(const Batch &b, const StepArgs &a);      // kb_squareroot_split16.hip: 12 < n <= 16, eight lanes
bool launch_squareroot_split12_plain(const Batch &b, const StepArgs &a);   // kb_squareroot_split12p.hip / 16p.hip: padded shapes, Noiseless, state only
bool launch_squareroot_split16_plain(const Batch &b, const StepArgs &a);
int launch_srif(const Batch &b, const StepArgs &a);
int launch_hybrid(const Batch &b, const StepArgs &a);
int launch_batch_ls(const Batch &b, const StepArgs &a);
bool hybrid_reg_ok(const Batch &b, const StepArgs &a);
bool launch_hybrid_padded(const Batch &b, const StepArgs &a);    // kb_hybrid_pad.hip: n <= 4 / p <= 2, n <= 6 / p <= 4
bool launch_hybrid_padded8(const Batch &b, const StepArgs &a);   // kb_hybrid_pad8.hip: n <= 8 / p <= 4
bool hybrid_split_ok(const Batch &b, const StepArgs &a);
bool launch_hybrid_split(const Batch &b, const StepArgs &a);     // kb_hybrid_split.hip: 8 < n <= 16, p <= 6, on the split-lane Vanilla kernel (HYB)
bool launch_hybrid_fused(const Batch &b, const StepArgs &a);    // kb_hybrid_fused.hip: kb_update_nl_steps_dev, 6 / 1..3 fp64, the caller loop in one launch (round 6)
bool launch_hybrid_strict(const Batch &b, const StepArgs &a);   // kb_hybrid_strict.hip: KB_FLAG_STRICT_SYMCHECK on registers (6 / 1..3, fp64)
bool srif_reg_ok(const Batch &b, const StepArgs &a);
// kb_srif_split_*.hip: fp64, one filter over 4 (n <= 12) / 8 (n <= 16) lanes, the state dimension at compile time, p at run time (kb_srif_split.h)
void launch_srif_split_n1(const Batch &b, const StepArgs &a);
void launch_srif_split_n2(const Batch &b, const StepArgs &a);
void launch_srif_split_n3(const Batch &b, const StepArgs &a);
void launch_srif_split_n4(const Batch &b, const StepArgs &a);
void launch_srif_split_n5(const Batch &b, const StepArgs &a);
void launch_srif_split_n6(const Batch &b, const StepArgs &a);
void launch_srif_split_n7(const Batch &b, const StepArgs &a);
void launch_srif_split_n8(const Batch &b, const StepArgs &a);
void launch_srif_split_n9(const Batch &b, const StepArgs &a);
void launch_srif_split_n10(const Batch &b, const StepArgs &a);
void launch_srif_split_n11(const Batch &b, const StepArgs &a);
void launch_srif_split_n12(const Batch &b, const StepArgs &a);
void launch_srif_split_n13(const Batch &b, const StepArgs &a);
void launch_srif_split_n14(const Batch &b, const StepArgs &a);
void launch_srif_split_n15(const Batch &b, const StepArgs &a);
void launch_srif_split_n16(const Batch &b, const StepArgs &a);
// kb_srif_split_f32*.hip: fp32, four lanes per filter: the shapes the two-lane fp32 kernels do not serve
void launch_srif_split_f32_n1(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n2(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n3(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n4(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n5(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n7(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n9(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n11(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n12(const Batch &b, const StepArgs &a);   // (diagnostic builds only: -DKB_DIAG_SRIF_F32_N12)
void launch_srif_split_f32_n13(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n14(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n15(const Batch &b, const StepArgs &a);
void launch_srif_split_f32_n16(const Batch &b, const StepArgs &a);
Layout make_layout(int kind, int n, int pmax, int m, unsigned flags);   // kb_api.hip
bool launch_srif_pair_f32(const Batch &b, const StepArgs &a);   // kb_srif_pair32.hip: Update with two lanes per filter; false = shape not covered
bool launch_srif_pair_f64(const Batch &b, const StepArgs &a);   // kb_srif_pair64.hip
bool launch_srif_pair_f32_fused(const Batch &b, const StepArgs &a);   // kb_srif_pair32h.hip: config E time-fused (kb_update_nl_steps_dev)
bool launch_srif_pair_f32b(const Batch &b, const StepArgs &a);  // kb_srif_pair32b.hip / 64b.hip: 8 / 10 states, p = 2 / 4
bool launch_srif_pair_f32c(const Batch &b, const StepArgs &a);  // kb_srif_pair32c.hip / 64c.hip: 12 states, p = 1 .. 5
bool launch_srif_pair_f32d(const Batch &b, const StepArgs &a);  // kb_srif_pair32d.hip / 64d.hip: 6, 8, 10 states, p = 5 / 6
bool launch_srif_pair_f32e(const Batch &b, const StepArgs &a);  // kb_srif_pair32e.hip / 64e.hip: 6, 8, 10 (fp32: and 12) states, p = 7 / 8
bool launch_srif_pair_f32f(const Batch &b, const StepArgs &a);  // kb_srif_pair32f/g.hip, 64f/g.hip: 14 and 16 states, p = 1 .. 6
bool launch_srif_pair_f32g(const Batch &b, const StepArgs &a);
// traj != nullptr: keep every run's State() and Measurement() per step (Batch::d_traj layout)
int launch_mc(const Batch &b, const StepArgs &a, const void *d_controls, int ncontrols, double *d_sums, void *traj, int64_t traj_ld);
int mc_repl();
int launch_fold(hipStream_t stream, const double *src, int repl, int64_t per, double *out);
// The device half of kb_mc_run_ex / kb_chisquare: everything up to the folded per-step sums in Batch::d_mc (asynchronous on the
// handle's stream); *folded = [steps][2][n] (MC) or [steps][2] (chi-square) doubles on the device, *shift = [steps][n] (MC only).
int mc_run_device(Batch &b, int steps, const double *controls, int ncontrols, int64_t first_run, unsigned mc_flags, double **folded, double **shift);
int chisq_run_device(Batch &truth, Batch &kf, int steps, const double *controls, int ncontrols, int64_t first_run, int replay_last_mc,
                     int with_nees, int with_nis, double **folded);
// shared host helpers (kb_api.hip)
int use_device(const Batch &b);
void after_sync(Batch &b);   // call after every hipStreamSynchronize of the handle's stream in a host-facing entry point
int ensure_stage(Batch &b, size_t bytes);
int ensure_xp(Batch &b);  // allocates Batch::d_xp on first use
int stage_host_vec(Batch &b, const double *host, int rows, void **dblock, int slot, const void **tile);  // *tile: the AoSoA tile(s) the step kernel reads
int ensure_pin(Batch &b);
constexpr size_t KB_PIN_TILE_BYTES = (size_t)KB_MAX_DIM * KB_TILE * sizeof(double);
constexpr size_t KB_PIN_OUT_BYTES = (size_t)KB_TILE * KB_MAX_DIM * KB_MAX_DIM * sizeof(double);
constexpr size_t KB_PIN_FLAG_OFF = 3 * KB_PIN_TILE_BYTES + KB_PIN_OUT_BYTES;   // completion word of a small snapshot, then padding to a cache line
constexpr int KB_SNAP_FLAG_MAX = 16;   // snapshots of up to this many filters run as ONE block that raises the completion word itself
void fill_step_args(const Batch &b, StepArgs &a);
int upload_field(Batch &b, int field, const double *host, int64_t count, int broadcast, int p_rows);

// The run-time-dimension kernels with a leading dimension of 16 keep 10-31 KB of private arrays per lane.  The runtime
// sizes a queue's scratch for the whole device (bytes per lane x 64 x 8192 wave slots = up to 15 GiB) the first time such
// a kernel runs on it and keeps it for the life of the queue, and a process may hold about 64 GiB of scratch in total:
// with one stream per handle, a handful of handles that each ran one big-shape generic step exhaust it and the runtime
// aborts the process (HSA_STATUS_ERROR_OUT_OF_RESOURCES).  So every scratch-heavy launch goes to ONE stream per device,
// ordered after the handle's stream and before its later work by a pair of events.  The register kernels (no scratch) and
// the small-shape generic kernels (<= 8 KB per lane) stay on the handle's own stream.
struct HeavyScope {
    hipStream_t stream;            // launch on this
    HeavyScope(const Batch &b, bool heavy) : HeavyScope(b.device, b.stream, heavy, b.ev_heavy) {}
    HeavyScope(int device, hipStream_t user, bool heavy, hipEvent_t *cached = nullptr);
    ~HeavyScope();
    HeavyScope(const HeavyScope &) = delete;
    HeavyScope &operator=(const HeavyScope &) = delete;
   private:
    hipStream_t user_ = nullptr;
    hipEvent_t ev_[2] = {nullptr, nullptr};
    bool heavy_ = false, owned_ = false;
};
constexpr size_t KB_MALL_BYTES = (size_t)256 << 20;   // Infinity Cache (MALL) of one MI355X
constexpr int KB_HEAVY_LD = 16;   // leading dimension from which a generic kernel counts as scratch-heavy

inline dim3 tile_grid(int64_t ntiles) { return dim3((unsigned)((ntiles + 3) / 4)); }

}  // namespace kb

struct kb_batch : kb::Batch {};
