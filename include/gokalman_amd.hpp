// gokalman_amd.hpp -- header-only C++ host mirror of gokalman's filter interface over the C ABI
// (gokalman_amd.h).  The reference's host language is Go; no Go toolchain exists in the build
// image, so this is the compiled-language host layer: same type and method names, argument
// meaning and error behaviour as kalman.go:35-72, noise.go:13-20 and the constructors, with every
// object being a batch of N independent filters (N = 1 reproduces a reference filter object).
//
//   gokalman::Noiseless noise(Q, R);
//   auto [kf, est0] = gokalman::NewVanilla(x0, P0, F, G, H, noise);      // vanilla.go:21
//   gokalman::Estimate est = kf->Update(measurement, control);            // vanilla.go:128
//   est.State(); est.Covariance(); est.IsWithinNσ(2);
//
// Matrices are row-major std::vector<double> wrapped in gokalman::Matrix {rows, cols, data};
// for N > 1 `data` holds N matrices back to back (or one, shared by all filters).
// Errors: the reference returns (nil, error) or panics; here every failure throws
// gokalman::Error carrying the kb_status code and the reference's message (gokalman::StepError for the
// numerical failure of one Update: the filter keeps its previous estimate and the next call runs normally).
// Update / Predict return an Estimate VALUE that owns its data (see class Estimate).
#pragma once
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "gokalman_amd.h"

namespace gokalman {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
    if (rc != KB_OK) throw Error(rc, kb_last_error());
}

struct Matrix {
    int rows = 0, cols = 0;
    std::vector<double> data;  // rows*cols (shared) or N*rows*cols (per filter)
    Matrix() = default;
    Matrix(int r, int c, std::vector<double> d) : rows(r), cols(c), data(std::move(d)) {}
    Matrix(int r, int c) : rows(r), cols(c), data((size_t)r * c, 0.0) {}
    bool shared() const { return data.size() == (size_t)rows * cols; }
    int64_t count() const { return (rows && cols) ? (int64_t)(data.size() / ((size_t)rows * cols)) : 0; }
    double At(int i, int j, int64_t filter = 0) const { return data[(size_t)filter * rows * cols + (size_t)i * cols + j]; }
};
using Vector = Matrix;  // cols == 1
inline Vector NewVector(int n, std::vector<double> d = {}) { return d.empty() ? Matrix(n, 1) : Matrix(n, 1, std::move(d)); }
inline Matrix ScaledIdentity(int n, double s) {  // helper.go:13-23
    Matrix m(n, n);
    for (int i = 0; i < n; i++) m.data[(size_t)i * n + i] = s;
    return m;
}
inline Matrix Identity(int n) { return ScaledIdentity(n, 1.0); }  // helper.go:44-46
inline bool IsNil(const Matrix &m) {                               // helper.go:49-62
    for (double v : m.data) if (v != 0.0) return false;
    return true;
}

// String() support.  The reference prints through gonum's mat64.Formatted(m, mat64.Prefix(p)) with fmt's %v: box-drawing
// brackets (square ones for a single row), every element right-aligned to the widest, two spaces between columns, the
// prefix in front of every line but the first.  Restated from gonum's documented behaviour (gonum is not available here:
// byte equality with a Go run is unverified); the labels, order and prefixes are the reference's format strings.
inline std::string go_v(double x) {   // fmt %v of a float64 = strconv 'g', shortest round-trip digits: %e form for exponents < -4 or >= 6 (1e6 prints 1e+06)
    if (std::isnan(x)) return "NaN";
    if (std::isinf(x)) return x > 0 ? "+Inf" : "-Inf";
    if (x == 0.0) return std::signbit(x) ? "-0" : "0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), std::fabs(x), std::chars_format::scientific);
    std::string sci(buf, r.ptr);   // d[.ddd]e[+-]XX
    const size_t e = sci.find('e');
    std::string digits = sci.substr(0, e);
    digits.erase(std::remove(digits.begin(), digits.end(), '.'), digits.end());
    const int e10 = std::stoi(sci.substr(e + 1));
    const std::string sign = x < 0 ? "-" : "";
    if (e10 < -4 || e10 >= 6) {
        char eb[16];
        std::snprintf(eb, sizeof(eb), "e%c%02d", e10 < 0 ? '-' : '+', std::abs(e10));
        return sign + digits.substr(0, 1) + (digits.size() > 1 ? "." + digits.substr(1) : "") + eb;
    }
    if (e10 >= 0) {
        if ((int)digits.size() <= e10 + 1) return sign + digits + std::string((size_t)(e10 + 1 - (int)digits.size()), '0');
        return sign + digits.substr(0, (size_t)e10 + 1) + "." + digits.substr((size_t)e10 + 1);
    }
    return sign + "0." + std::string((size_t)(-e10 - 1), '0') + digits;
}
inline std::string Formatted(const Matrix &m, const std::string &prefix, int64_t filter = 0) {
    if (m.rows == 0 || m.cols == 0 || m.data.empty()) return "<nil>";
    const int64_t f = m.shared() ? 0 : filter;
    std::vector<std::string> cells;
    size_t width = 0;
    for (int i = 0; i < m.rows; i++)
        for (int j = 0; j < m.cols; j++) { cells.push_back(go_v(m.At(i, j, f))); width = std::max(width, cells.back().size()); }
    std::string out;
    for (int i = 0; i < m.rows; i++) {
        if (i) out += "\n" + prefix;
        out += m.rows == 1 ? "[" : (i == 0 ? "\u23a1" : (i == m.rows - 1 ? "\u23a3" : "\u23a2"));
        for (int j = 0; j < m.cols; j++) {
            const std::string &c = cells[(size_t)i * m.cols + j];
            out += (j ? "  " : "") + std::string(width - c.size(), ' ') + c;
        }
        out += m.rows == 1 ? "]" : (i == 0 ? "\u23a4" : (i == m.rows - 1 ? "\u23a6" : "\u23a5"));
    }
    return out;
}

// noise.go:13-20.  Process/Measurement sampling happens on the device; the host object carries Q, R
// and which implementation to use.
struct Noise {
    Matrix Q, R;
    int kind = KB_NOISE_NOISELESS;
    uint64_t seed = 0;
    const Matrix &ProcessMatrix() const { return Q; }
    const Matrix &MeasurementMatrix() const { return R; }
    std::string String() const {   // noise.go:62-64, :104-106, :162-164
        if (kind == KB_NOISE_BATCH) return "BatchNoise";
        return std::string(kind == KB_NOISE_AWGN ? "AWGN" : "Noiseless") + "{\nQ=" + Formatted(Q, "  ") + "\nR=" + Formatted(R, "  ") + "}\n";
    }
};
inline Noise NewNoiseless(Matrix Q, Matrix R) { return Noise{std::move(Q), std::move(R), KB_NOISE_NOISELESS, 0}; }       // noise.go:29-37
inline Noise NewAWGN(Matrix Q, Matrix R, uint64_t seed = 0) { return Noise{std::move(Q), std::move(R), KB_NOISE_AWGN, seed}; }  // noise.go:117-121

// The reference's per-call error of an Update that failed NUMERICALLY for a filter (vanilla.go:164-167, :207-215,
// hybrid.go:150-152, srif.go:112-114): thrown for batches of one filter, exactly where the reference returns
// (nil, err); the filter keeps its previous estimate and the next Update runs normally.  Larger batches report the
// per-filter status words through Estimate::Status() instead.
struct StepError : Error {
    uint32_t status;
    StepError(uint32_t st, const std::string &m) : Error(KB_ERR_INVALID, m), status(st) {}
};

class Batch {
   public:
    Batch(int kind, int n, int p, int m, int64_t N, unsigned flags = KB_FLAG_FULL_ESTIMATE, int dtype = KB_F64, int device = 0)
        : kind_(kind), n_(n), N_(N), flags_(flags) {
        check(kb_create(&h_, kind, n, p, m, N, dtype, device, flags));
    }
    // N copies of filter `filter` of an initialised batch (kb_replicate): its model, its INITIAL estimate, its noise selection
    static std::shared_ptr<Batch> Replicate(const Batch &src, int64_t filter, int64_t N, unsigned flags) {
        kb_batch *h = nullptr;
        check(kb_replicate(src.h_, filter, N, flags, &h));
        return std::shared_ptr<Batch>(new Batch(h, src.kind_, src.n_, N, flags | (src.flags_ & (KB_FLAG_INFO_FROM_STATE | KB_FLAG_SRIF_NON_TRI_R))));
    }
    ~Batch() { kb_destroy(h_); }
    Batch(const Batch &) = delete;
    Batch &operator=(const Batch &) = delete;
    kb_batch *handle() const { return h_; }
    int kind() const { return kind_; }
    int n() const { return n_; }
    int64_t N() const { return N_; }
    unsigned flags() const { return flags_; }
    void set(int field, const Matrix &m, int p_rows = 0) {
        check(kb_set(h_, field, m.data.data(), m.shared() ? 1 : N_, m.shared() ? 1 : 0, p_rows));
    }
    Matrix get(int field, int rows, int cols) const {
        Matrix out(rows, cols);
        out.data.assign((size_t)N_ * rows * cols, 0.0);
        check(kb_get(h_, field, out.data.data(), 0, N_));
        return out;
    }

   private:
    Batch(kb_batch *adopted, int kind, int n, int64_t N, unsigned flags) : h_(adopted), kind_(kind), n_(n), N_(N), flags_(flags) {}
    kb_batch *h_ = nullptr;
    int kind_, n_;
    int64_t N_;
    unsigned flags_;
};

// Batches up to this size get an owning Estimate from every Update (the reference's semantics); larger ones get a
// guarded view (copying a million estimates to the host every step is what the device path exists to avoid).
constexpr int64_t kSnapshotMaxFilters = 4096;

// kalman.go:64-72.  An Estimate is an immutable VALUE: the reference's Update returns a freshly allocated estimate and
// the callers keep it (vanilla.go:216-218; consumed later through a channel in examples/jerkcar/main.go:71-90, stored per
// step in montecarlo.go:108-117).  Here it owns one snapshot taken with a single kb_get_estimate call, shared by its
// copies like the Go pointer is.  For batches above kSnapshotMaxFilters it is a view instead, and every getter throws
// once the batch has moved on (never a silent read of a later step); Freeze() turns a live view into a snapshot.
class Estimate {
   public:
    struct Snapshot {
        int n = 0, p = 0;
        int64_t N = 0;
        bool has_full = false, has_innovation = false, has_gain = false;
        Matrix state, covariance, pred_covariance, gain, innovation, measurement;
        std::vector<uint32_t> status;
    };
    Estimate() = default;
    // snapshot == true: download now (clear_status: read-and-clear the status words, the per-call error semantics)
    Estimate(std::shared_ptr<Batch> b, bool snapshot, bool clear_status = false) : b_(std::move(b)) {
        if (snapshot) snap_ = download(*b_, clear_status, nullptr);
        step_ = kb_step(b_->handle());
        calls_ = kb_calls(b_->handle());
    }
    // The estimate of a step that `step_call` runs: kb_update_estimate & co. enqueue the step and the snapshot back to back and the
    // host waits ONCE (the reference's `est, err := kf.Update(y, u)`)
    using StepCall = std::function<int(int64_t first, int64_t count, kb_estimate_view *view)>;
    Estimate(std::shared_ptr<Batch> b, const StepCall &step_call, bool clear_status) : b_(std::move(b)) {
        snap_ = download(*b_, clear_status, &step_call);
        step_ = kb_step(b_->handle());
        calls_ = kb_calls(b_->handle());
    }
    bool Owning() const { return snap_ != nullptr; }
    Estimate &Freeze() {
        if (!snap_) { live(); snap_ = download(*b_, false, nullptr); }
        return *this;
    }
    Vector State() const { return snap_ ? snap_->state : (live(), b_->get(KB_STATE, b_->n(), 1)); }
    Matrix Covariance() const { return snap_ ? snap_->covariance : (live(), b_->get(KB_COVAR, b_->n(), b_->n())); }
    Vector Measurement() const {
        if (snap_) { need(snap_->has_full, "Measurement"); return snap_->measurement; }
        live();
        return b_->get(KB_MEASUREMENT, kb_meas_dim(b_->handle()), 1);
    }
    Vector Innovation() const {
        if (snap_) { need(snap_->has_innovation, "Innovation"); return snap_->innovation; }
        live();
        const bool info = b_->kind() == KB_INFORMATION || b_->kind() == KB_SRIF;
        return b_->get(KB_INNOVATION, info ? b_->n() : kb_meas_dim(b_->handle()), 1);
    }
    Matrix PredCovariance() const {
        if (snap_) { need(snap_->has_full, "PredCovariance"); return snap_->pred_covariance; }
        live();
        return b_->get(KB_PRED_COVAR, b_->n(), b_->n());
    }
    Matrix Gain() const {
        if (snap_) { need(snap_->has_gain, "Gain"); return snap_->gain; }
        live();
        return b_->get(KB_GAIN, b_->n(), kb_meas_dim(b_->handle()));
    }
    // per-filter status bits of the step this estimate belongs to (0 = the reference's err == nil)
    std::vector<uint32_t> Status() const {
        if (snap_) return snap_->status;
        live();
        std::vector<uint32_t> st((size_t)b_->N());
        check(kb_get_status(b_->handle(), st.data(), 0, b_->N()));
        return st;
    }
    // vanilla.go:231-239: |x_i| <= N sqrt(P_ii) for every component
    std::vector<uint8_t> IsWithinNσ(double N) const {
        std::vector<uint8_t> out;
        if (snap_) {
            const int n = snap_->n;
            out.assign((size_t)snap_->N, 1);
            for (int64_t f = 0; f < snap_->N; f++)
                for (int i = 0; i < n; i++) {
                    const double ns = N * std::sqrt(snap_->covariance.At(i, i, f)), x = snap_->state.At(i, 0, f);
                    if (x > ns || x < -ns) out[(size_t)f] = 0;
                }
            return out;
        }
        live();
        out.assign((size_t)b_->N(), 0);
        check(kb_is_within_nsigma(b_->handle(), N, out.data(), 0, b_->N()));
        return out;
    }
    std::vector<uint8_t> IsWithin2σ() const { return IsWithinNσ(2); }
    // <Kind>Estimate.String() of filter f (vanilla.go:276-284, squareroot.go:347-355, hybrid.go:300-308; information.go:318-325
    // has no gain, srif.go:283-289 neither gain nor innovation); members the batch does not keep print as Go's nil
    std::string String(int64_t f = 0) const {
        auto opt = [&](auto getter, const char *prefix) {
            try { return Formatted((this->*getter)(), prefix, f); } catch (const Error &) { return std::string("<nil>"); }
        };
        const std::string s = opt(&Estimate::State, "  "), y = opt(&Estimate::Measurement, "  "), P = opt(&Estimate::Covariance, "  "),
                          Pm = opt(&Estimate::PredCovariance, "   ");
        const int kind = b_ ? b_->kind() : KB_VANILLA;
        if (kind == KB_SRIF) return "{\ns=" + s + "\ny=" + y + "\nP=" + P + "\nP-=" + Pm + "\n}";
        const std::string i = opt(&Estimate::Innovation, "  ");
        if (kind == KB_INFORMATION) return "{\ns=" + s + "\ny=" + y + "\nP=" + P + "\nP-=" + Pm + "\ni=" + i + "\n}";
        return "{\ns=" + s + "\ny=" + y + "\nP=" + P + "\nK=" + opt(&Estimate::Gain, "  ") + "\nP-=" + Pm + "\ni=" + i + "\n}";
    }

   private:
    static std::shared_ptr<const Snapshot> download(Batch &b, bool clear_status, const StepCall *step_call) {
        auto s = std::make_shared<Snapshot>();
        const int n = b.n(), p = kb_meas_dim(b.handle());
        const int64_t N = b.N();
        const int kind = b.kind();
        const bool info = kind == KB_INFORMATION || kind == KB_SRIF;
        const bool full = (b.flags() & KB_FLAG_FULL_ESTIMATE) != 0;
        const bool lazy = kind == KB_SQUAREROOT || kind == KB_INFORMATION || kind == KB_SRIF || kind == KB_BATCH_LS;
        s->n = n; s->p = p; s->N = N; s->has_full = full; s->has_innovation = info || full;
        s->has_gain = full && (!lazy || kind == KB_SQUAREROOT);
        auto alloc = [&](Matrix &m, int r, int c) { m = Matrix(r, c); m.data.assign((size_t)N * r * c, 0.0); return m.data.data(); };
        kb_estimate_view v{};
        v.state = alloc(s->state, n, 1);
        v.covariance = alloc(s->covariance, n, n);
        if (full) { v.pred_covariance = alloc(s->pred_covariance, n, n); v.measurement = alloc(s->measurement, p, 1); }
        if (s->has_gain) v.gain = alloc(s->gain, n, p);
        if (s->has_innovation) v.innovation = alloc(s->innovation, info ? n : p, 1);
        s->status.assign((size_t)N, 0u);
        v.status = s->status.data();
        v.clear_status = clear_status ? 1 : 0;
        check(step_call ? (*step_call)(0, N, &v) : kb_get_estimate(b.handle(), 0, N, &v));
        return s;
    }
    void live() const {
        if (!b_) throw Error(KB_ERR_INVALID, "empty Estimate");
        if (kb_calls(b_->handle()) != calls_)   // kb_step is kf.step, which a failed Update does not advance: the call counter is monotone
            throw Error(KB_ERR_INVALID, "this Estimate is a view of step " + std::to_string(step_) + " but the batch is at step " +
                                            std::to_string(kb_step(b_->handle())) + ": Freeze() it before the next Update to keep it");
    }
    static void need(bool have, const char *what) {
        if (!have) throw Error(KB_ERR_INVALID, std::string(what) + "() needs a batch created with KB_FLAG_FULL_ESTIMATE");
    }
    std::shared_ptr<Batch> b_;
    std::shared_ptr<const Snapshot> snap_;
    int64_t step_ = 0, calls_ = 0;
};

// The estimate of the step that just ran, with the reference's per-call error behaviour.  at_k: srif.go:113 and hybrid.go:151
// print the step ("... at k=%d: ..."), vanilla.go:166 does not.  kf.step is not advanced by the failed call (kb_step), so it
// still is the k of the step that failed.
inline Estimate step_estimate(const std::shared_ptr<Batch> &b, const char *what_failed, bool at_k, const Estimate::StepCall &step_call,
                              const std::function<int()> &step_only) {
    if (b->N() > kSnapshotMaxFilters) { check(step_only()); return Estimate(b, false); }
    Estimate est(b, step_call, /*clear_status=*/true);
    if (b->N() == 1) {
        const uint32_t st = est.Status()[0];
        if (st & KB_ST_SINGULAR)
            throw StepError(st, std::string("could not invert ") + what_failed + (at_k ? " at k=" + std::to_string(kb_step(b->handle())) : std::string()) +
                                    ": matrix singular or near-singular");
        if (st & KB_ST_ASYMMETRIC) throw StepError(st, "matrix is not symmetric");                       // helper.go:76
        if (st & KB_ST_NONFINITE) throw StepError(st, "matrix is not symmetric (non-finite covariance)");  // NaN fails helper.go:75's comparison
    }
    return est;
}

// kalman.go:35-47
class LDKF {
   public:
    virtual ~LDKF() = default;
    // Update(measurement, control *mat64.Vector) (Estimate, error)
    Estimate Update(const Vector &measurement, const Vector &control) {
        const std::vector<double> y = expand(measurement), u = expand(control);
        const double *up = control.rows ? u.data() : nullptr;
        kb_batch *h = b_->handle();
        return step_estimate(b_, "`H*P_kp1_minus*H' + R`", false,   // vanilla.go:166
                             [&](int64_t first, int64_t count, kb_estimate_view *v) { return kb_update_estimate(h, y.data(), measurement.rows, up, control.rows, first, count, v); },
                             [&] { return kb_update(h, y.data(), measurement.rows, up, control.rows); });
    }
    const Noise &GetNoise() const { return noise_; }
    const Matrix &GetStateTransition() const { return F_; }
    const Matrix &GetInputControl() const { return G_; }
    const Matrix &GetMeasurementMatrix() const { return H_; }
    void SetStateTransition(const Matrix &F) { F_ = F; b_->set(KB_F, F); }
    void SetInputControl(const Matrix &G) { G_ = G; b_->set(KB_G, G); }
    void SetMeasurementMatrix(const Matrix &H) { H_ = H; b_->set(KB_H, H, H.rows); }
    void SetNoise(const Noise &n) {
        noise_ = n;
        b_->set(KB_Q, n.Q);
        b_->set(KB_R, n.R, n.R.rows);
    }
    void Reset() { check(kb_reset(b_->handle())); }
    // vanilla.go:76-78, squareroot.go:65-67 (filter f of the batch; shared matrices print as they were given)
    std::string String(int64_t f = 0) const {
        return "F=" + Formatted(F_, "  ", f) + "\nG=" + Formatted(G_, "  ", f) + "\nH=" + Formatted(H_, "  ", f) + "\n" + noise_.String();
    }
    std::string LastKernel() const { return kb_last_kernel(b_->handle()); }   // which instantiation(s) served the last step (debugging aid)
    int64_t Step() const { return kb_step(b_->handle()); }
    std::shared_ptr<Batch> batch() const { return b_; }

   protected:
    LDKF(int kind, const Vector &x0, const Matrix &P0, const Matrix &F, const Matrix &G, const Matrix &H, const Noise &noise,
         int64_t N, int pmax, unsigned flags)
        : F_(F), G_(G), H_(H), noise_(noise) {
        // checkMatDims at construction (vanilla.go:23-31), same messages
        if (x0.rows != P0.cols) throw Error(KB_ERR_DIMS, dim2("x0", x0.rows, "Covar0", P0.cols));
        if (F.rows != P0.cols) throw Error(KB_ERR_DIMS, dim2("F", F.rows, "Covar0", P0.cols));
        if (H.cols != x0.rows) throw Error(KB_ERR_DIMS, "dimensions must agree: H(...x" + std::to_string(H.cols) + ") x0(" + std::to_string(x0.rows) + "x...)");
        const int m = G.cols;  // needCtrl = !IsNil(G) is evaluated by kb_set(KB_G) (vanilla.go:39)
        b_ = std::make_shared<Batch>(kind, x0.rows, pmax > H.rows ? pmax : H.rows, m, N, flags);
        b_->set(KB_X, x0); b_->set(KB_P, P0); b_->set(KB_F, F);
        if (m > 0) b_->set(KB_G, G);
        b_->set(KB_H, H, H.rows); b_->set(KB_Q, noise.Q); b_->set(KB_R, noise.R, noise.R.rows);
        if (noise.kind != KB_NOISE_NOISELESS) check(kb_set_noise_kind(b_->handle(), noise.kind, noise.seed));
        check(kb_init(b_->handle()));
    }
    static std::string dim2(const char *a, int ra, const char *b, int cb) {
        return std::string("dimensions must agree: ") + a + "(" + std::to_string(ra) + "x...) " + b + "(...x" + std::to_string(cb) + ")";
    }
    std::vector<double> expand(const Vector &v) const {  // one vector for every filter, or N vectors
        if (v.rows == 0) return {};
        if (!v.shared() || b_->N() == 1) return v.data;
        std::vector<double> out;
        out.reserve((size_t)b_->N() * v.rows);
        for (int64_t i = 0; i < b_->N(); i++) out.insert(out.end(), v.data.begin(), v.data.end());
        return out;
    }
    std::shared_ptr<Batch> b_;
    Matrix F_, G_, H_;
    Noise noise_;
};

struct Vanilla : LDKF {  // vanilla.go:65-74
    Vanilla(const Vector &x0, const Matrix &P0, const Matrix &F, const Matrix &G, const Matrix &H, const Noise &n, bool predictOnly,
            int64_t N, int pmax, unsigned flags)
        : LDKF(predictOnly ? KB_VANILLA_PREDICT : KB_VANILLA, x0, P0, F, G, H, n, N, pmax, flags) {}
};
struct SquareRoot : LDKF {  // squareroot.go:53-63
    SquareRoot(const Vector &x0, const Matrix &P0, const Matrix &F, const Matrix &G, const Matrix &H, const Noise &n, int64_t N, int pmax, unsigned flags)
        : LDKF(KB_SQUAREROOT, x0, P0, F, G, H, n, N, pmax, flags) {}
};
struct Information : LDKF {  // information.go:84-95
    Information(const Vector &i0, const Matrix &I0, const Matrix &F, const Matrix &G, const Matrix &H, const Noise &n, bool fromState,
                int64_t N, int pmax, unsigned flags)
        : LDKF(KB_INFORMATION, i0, I0, F, G, H, n, N, pmax, flags | (fromState ? KB_FLAG_INFO_FROM_STATE : 0u)) {}
};

using VanillaPair = std::pair<std::shared_ptr<Vanilla>, Estimate>;
// NewVanilla(x0, Covar0, F, G, H, noise) (*Vanilla, *VanillaEstimate, error)   vanilla.go:21-40
inline VanillaPair NewVanilla(const Vector &x0, const Matrix &P0, const Matrix &F, const Matrix &G, const Matrix &H, const Noise &noise,
                              int64_t N = 1, int pmax = 0, unsigned flags = KB_FLAG_FULL_ESTIMATE) {
    auto kf = std::make_shared<Vanilla>(x0, P0, F, G, H, noise, false, N, pmax, flags);
    return {kf, Estimate(kf->batch(), true)};
}
// NewPurePredictorVanilla   vanilla.go:43-62
inline VanillaPair NewPurePredictorVanilla(const Vector &x0, const Matrix &P0, const Matrix &F, const Matrix &G, const Matrix &H, const Noise &noise,
                                           int64_t N = 1, int pmax = 0, unsigned flags = KB_FLAG_FULL_ESTIMATE) {
    auto kf = std::make_shared<Vanilla>(x0, P0, F, G, H, noise, true, N, pmax, flags);
    return {kf, Estimate(kf->batch(), true)};
}
// NewSquareRoot   squareroot.go:21-50
inline std::pair<std::shared_ptr<SquareRoot>, Estimate> NewSquareRoot(const Vector &x0, const Matrix &P0, const Matrix &F, const Matrix &G, const Matrix &H,
                                                                      const Noise &noise, int64_t N = 1, int pmax = 0, unsigned flags = KB_FLAG_FULL_ESTIMATE) {
    auto kf = std::make_shared<SquareRoot>(x0, P0, F, G, H, noise, N, pmax, flags);
    return {kf, Estimate(kf->batch(), true)};
}
// NewInformation(i0, I0, ...)   information.go:20-53
inline std::pair<std::shared_ptr<Information>, Estimate> NewInformation(const Vector &i0, const Matrix &I0, const Matrix &F, const Matrix &G, const Matrix &H,
                                                                        const Noise &noise, int64_t N = 1, int pmax = 0, unsigned flags = KB_FLAG_FULL_ESTIMATE) {
    auto kf = std::make_shared<Information>(i0, I0, F, G, H, noise, false, N, pmax, flags);
    return {kf, Estimate(kf->batch(), true)};
}
// NewInformationFromState(x0, P0, ...)   information.go:65-81
inline std::pair<std::shared_ptr<Information>, Estimate> NewInformationFromState(const Vector &x0, const Matrix &P0, const Matrix &F, const Matrix &G, const Matrix &H,
                                                                                 const Noise &noise, int64_t N = 1, int pmax = 0, unsigned flags = KB_FLAG_FULL_ESTIMATE) {
    auto kf = std::make_shared<Information>(x0, P0, F, G, H, noise, true, N, pmax, flags);
    return {kf, Estimate(kf->batch(), true)};
}

// kalman.go:51-60
class NLDKF {
   public:
    virtual ~NLDKF() = default;
    void Prepare(const Matrix &Phi, const Matrix &Htilde) {
        check(kb_prepare(b_->handle(), Phi.data.data(), Htilde.data.data(), Phi.shared() ? 1 : b_->N(), Phi.shared() ? 1 : 0));
    }
    Estimate Predict() {
        kb_batch *h = b_->handle();
        return step_estimate(b_, what_failed(), true, [&](int64_t first, int64_t count, kb_estimate_view *v) { return kb_predict_nl_estimate(h, first, count, v); },
                             [&] { return kb_predict_nl(h); });
    }
    Estimate Update(const Vector &realObservation, const Vector &computedObservation) {
        const std::vector<double> r = expand(realObservation), c = expand(computedObservation);
        kb_batch *h = b_->handle();
        return step_estimate(b_, what_failed(), true,
                             [&](int64_t first, int64_t count, kb_estimate_view *v) {
                                 return kb_update_nl_estimate(h, r.data(), realObservation.rows, c.data(), computedObservation.rows, first, count, v);
                             },
                             [&] { return kb_update_nl(h, r.data(), realObservation.rows, c.data(), computedObservation.rows); });
    }
    int64_t Step() const { return kb_step(b_->handle()); }
    bool EKFEnabled() const { return kb_ekf_enabled(b_->handle()) != 0; }
    void EnableEKF() { check(kb_set_ekf(b_->handle(), 1)); }
    void DisableEKF() { check(kb_set_ekf(b_->handle(), 0)); }
    void PreparePNT(const Matrix &Gamma) {
        check(kb_prepare_pnt(b_->handle(), Gamma.data.data(), Gamma.shared() ? 1 : b_->N(), Gamma.shared() ? 1 : 0));
    }
    std::shared_ptr<Batch> batch() const { return b_; }

   protected:
    // srif.go:113 "could not invert `Φ` at k=%d", hybrid.go:151 "could not invert `H*P_kp1_minus*H' + R` at k=%d"
    const char *what_failed() const { return b_->kind() == KB_SRIF ? "`Φ`" : "`H*P_kp1_minus*H' + R`"; }
    std::vector<double> expand(const Vector &v) const {
        if (!v.shared() || b_->N() == 1) return v.data;
        std::vector<double> out;
        for (int64_t i = 0; i < b_->N(); i++) out.insert(out.end(), v.data.begin(), v.data.end());
        return out;
    }
    std::shared_ptr<Batch> b_;
};

struct SRIF : NLDKF {  // NewSRIF(x0, P0, measSize, nonTriR, noise)   srif.go:14-49
    SRIF(const Vector &x0, const Matrix &P0, int measSize, bool nonTriR, const Noise &n, int64_t N = 1, unsigned flags = KB_FLAG_FULL_ESTIMATE) {
        if (x0.rows != P0.cols) throw Error(KB_ERR_DIMS, "dimensions must agree: x0(" + std::to_string(x0.rows) + "x...) P0(...x" + std::to_string(P0.cols) + ")");
        (void)measSize;  // the reference only uses it to size Predict()'s zero vectors
        b_ = std::make_shared<Batch>(KB_SRIF, x0.rows, n.R.rows, 0, N, flags | (nonTriR ? KB_FLAG_SRIF_NON_TRI_R : 0u));
        b_->set(KB_X, x0); b_->set(KB_P, P0); b_->set(KB_R, n.R, n.R.rows);
        check(kb_init(b_->handle()));
    }
    void SetNoise(const Noise &) { throw Error(KB_ERR_UNSUPPORTED, "noise not yet supported for SRIF"); }  // srif.go:76-78 (a panic there)
};
struct HybridKF : NLDKF {  // NewHybridKF(x0, P0, noise, measSize)   hybrid.go:23-34
    HybridKF(const Vector &x0, const Matrix &P0, const Noise &n, int measSize, int64_t N = 1, unsigned flags = KB_FLAG_FULL_ESTIMATE) {
        if (x0.rows != P0.cols) throw Error(KB_ERR_DIMS, "dimensions must agree: x0(" + std::to_string(x0.rows) + "x...) Covar0(...x" + std::to_string(P0.cols) + ")");
        b_ = std::make_shared<Batch>(KB_HYBRID, x0.rows, measSize, n.Q.rows, N, flags);
        b_->set(KB_X, x0); b_->set(KB_P, P0); b_->set(KB_R, n.R, n.R.rows);
        if (n.Q.rows > 0) b_->set(KB_Q, n.Q);
        check(kb_init(b_->handle()));
    }
    void SetNoise(const Noise &n) { b_->set(KB_R, n.R, n.R.rows); if (n.Q.rows > 0) b_->set(KB_Q, n.Q); }  // hybrid.go:68-70
};

// ---- Monte-Carlo runs (montecarlo.go:11-124) and the chi-square tests (chisquare.go:16-95) -------------------------------
// Everything the reference's MonteCarloRuns holds, behind the same names.  The runs live on the device: per-step sums for
// Mean / StdDev always, every run's State() / Measurement() per step (Runs, AsCSV) when the ensemble was kept.
struct MonteCarloData {
    int64_t runs = 0;        // all runs of the ensemble (montecarlo.go:12)
    int steps = 0, n = 0, p = 0;
    std::vector<double> mean, stddev;            // [steps][n]
    std::shared_ptr<Batch> truth;                // one pure-predictor AWGN filter per run
    std::vector<double> controls; int ncontrols = 0;
    bool kept = false;
    mutable std::vector<double> states, meas;    // [runs][steps][n], [runs][steps][p]: downloaded on first use
    mutable std::vector<double> ppred, gain;     // [steps][n][n], [steps][n][p]: identical for every run
    void need_kept() const {
        if (!kept) throw Error(KB_ERR_INVALID, "these Monte-Carlo runs were not kept (NewMonteCarloRuns(..., keepRuns = true)): only Mean / StdDev / NewChiSquare are available");
    }
    void download() const {
        need_kept();
        if (!states.empty()) return;
        const int64_t N = truth->N();
        states.assign((size_t)N * steps * n, 0.0);
        meas.assign((size_t)N * steps * p, 0.0);
        check(kb_mc_get_runs(truth->handle(), 0, N, states.data(), meas.data()));
    }
    // P-_k and K_k do not see the noise: one Noiseless copy of the filter stepped through the controls with KB_FLAG_FULL_ESTIMATE
    void shared() const {
        need_kept();
        if (!ppred.empty()) return;
        auto one = Batch::Replicate(*truth, 0, 1, KB_FLAG_FULL_ESTIMATE);
        check(kb_set_noise_kind(one->handle(), KB_NOISE_NOISELESS, 0));
        const int m = ncontrols ? (int)(controls.size() / (size_t)ncontrols) : 0;
        std::vector<double> y0((size_t)p, 0.0), u0((size_t)(m > 0 ? m : 1), 0.0), P((size_t)steps * n * n), K((size_t)steps * n * p);
        const bool ctrl = kb_need_ctrl(one->handle()) != 0;
        for (int t = 0; t < steps; t++) {
            const double *u = !ctrl ? nullptr : (ncontrols == 1 ? u0.data() : controls.data() + (size_t)t * m);
            check(kb_update(one->handle(), y0.data(), p, u, ctrl ? m : 0));
            kb_estimate_view v{};
            v.pred_covariance = P.data() + (size_t)t * n * n;
            v.gain = K.data() + (size_t)t * n * p;
            check(kb_get_estimate(one->handle(), 0, 1, &v));
        }
        ppred = std::move(P); gain = std::move(K);
    }
};

// MonteCarloRun.Estimates[k] (montecarlo.go:108-117): what a pure-predictor Vanilla returns (vanilla.go:170-179):
// {x-, yhat, 0, sym(P-), sym(P-), K}
class MonteCarloEstimate {
   public:
    MonteCarloEstimate(std::shared_ptr<const MonteCarloData> d, int64_t run, int step) : d_(std::move(d)), r_(run), k_(step) {}
    Vector State() const { d_->download(); return slice(d_->states, ((size_t)r_ * d_->steps + k_) * d_->n, d_->n, 1); }
    Vector Measurement() const { d_->download(); return slice(d_->meas, ((size_t)r_ * d_->steps + k_) * d_->p, d_->p, 1); }
    Vector Innovation() const { return Matrix(d_->p, 1); }
    Matrix Covariance() const { return PredCovariance(); }
    Matrix PredCovariance() const { d_->shared(); return slice(d_->ppred, (size_t)k_ * d_->n * d_->n, d_->n, d_->n); }
    Matrix Gain() const { d_->shared(); return slice(d_->gain, (size_t)k_ * d_->n * d_->p, d_->n, d_->p); }
    bool IsWithinNσ(double N) const {
        const Vector x = State();
        const Matrix P = Covariance();
        for (int i = 0; i < d_->n; i++) {
            const double ns = N * std::sqrt(P.At(i, i));
            if (x.data[(size_t)i] > ns || x.data[(size_t)i] < -ns) return false;
        }
        return true;
    }
    bool IsWithin2σ() const { return IsWithinNσ(2); }

   private:
    static Matrix slice(const std::vector<double> &v, size_t off, int r, int c) {
        return Matrix(r, c, std::vector<double>(v.begin() + (std::ptrdiff_t)off, v.begin() + (std::ptrdiff_t)(off + (size_t)r * c)));
    }
    std::shared_ptr<const MonteCarloData> d_;
    int64_t r_;
    int k_;
};
struct MonteCarloRun {   // montecarlo.go:122-124
    std::vector<MonteCarloEstimate> Estimates;
};

struct MonteCarloRuns {   // montecarlo.go:11-15
    int64_t runs = 0;
    int steps = 0;
    std::vector<MonteCarloRun> Runs;   // empty when the ensemble was not kept
    std::shared_ptr<const MonteCarloData> data;
    std::vector<double> Mean(int step) const {     // montecarlo.go:18-37
        return {data->mean.begin() + (std::ptrdiff_t)((size_t)step * data->n), data->mean.begin() + (std::ptrdiff_t)((size_t)(step + 1) * data->n)};
    }
    std::vector<double> StdDev(int step) const {   // montecarlo.go:40-59 (stat.StdDev: n - 1)
        return {data->stddev.begin() + (std::ptrdiff_t)((size_t)step * data->n), data->stddev.begin() + (std::ptrdiff_t)((size_t)(step + 1) * data->n)};
    }
    // AsCSV(headers) (montecarlo.go:62-89): one string per state component: "h-0,h-1,...,h-mean,h-stddev", then per step every
    // run's value, the mean and the standard deviation, all %f
    std::vector<std::string> AsCSV(const std::vector<std::string> &headers) const {
        data->download();
        const MonteCarloData &d = *data;
        const int64_t N = d.truth->N();
        auto f = [](double x) {   // Go's %f
            if (std::isnan(x)) return std::string("NaN");
            if (std::isinf(x)) return std::string(x > 0 ? "+Inf" : "-Inf");
            char buf[400];
            std::snprintf(buf, sizeof(buf), "%f", x);
            return std::string(buf);
        };
        std::vector<std::string> rtn((size_t)d.n);
        for (int i = 0; i < d.n; i++) {
            const std::string &h = headers.at((size_t)i);
            std::string out;
            for (int64_t r = 0; r < N; r++) out += h + "-" + std::to_string(r) + ",";
            out += h + "-mean," + h + "-stddev";
            for (int k = 0; k < d.steps; k++) {
                out += "\n";
                for (int64_t r = 0; r < N; r++) out += f(d.states[((size_t)r * d.steps + k) * d.n + i]) + ",";
                out += f(d.mean[(size_t)k * d.n + i]) + "," + f(d.stddev[(size_t)k * d.n + i]);
            }
            rtn[(size_t)i] = std::move(out);
        }
        return rtn;
    }
};

constexpr size_t kMonteCarloAutoKeepBytes = (size_t)256 << 20;   // keepRuns < 0 keeps ensembles of the reference's size, not the benchmark's

// NewMonteCarloRuns(samples, steps, rowsH, controls, kf) MonteCarloRuns   montecarlo.go:92-119.
// kf is the reference's argument: ONE pure-predictor Vanilla (AWGN noise).  The `samples` runs the reference performs one after
// the other on it, Reset() in between, are `samples` copies of it in one launch (kb_replicate + kb_mc_run_ex), and kf is left
// Reset() as montecarlo.go:116 leaves it.  A kf that already is a batch of `samples` filters is used as it is.
// keepRuns: 1 keeps every run for Runs / AsCSV (refused above KB_MC_KEEP_MAX_BYTES), 0 only the statistics, -1 decides by size.
// firstRun: global index of this process's first run when an ensemble is sharded over GPUs.
inline MonteCarloRuns NewMonteCarloRuns(int64_t samples, int steps, int rowsH, const std::vector<Vector> &controls, Vanilla &kf,
                                        int keepRuns = -1, int64_t firstRun = 0) {
    if (kf.batch()->kind() != KB_VANILLA_PREDICT)
        throw Error(KB_ERR_INVALID, "the Kalman filter needed for the Monte Carlo runs must be a pure predictor");   // montecarlo.go:93-95 (a panic)
    if ((int)controls.size() != 1 && (int)controls.size() != steps)
        throw Error(KB_ERR_INVALID, "must provide as much control vectors as steps, or just one control vector");    // montecarlo.go:105-107 (a panic)
    const int p = kb_meas_dim(kf.batch()->handle());
    if (rowsH != p)   // montecarlo.go:111 hands Update a zero vector of rowsH rows; vanilla.go:133-135 rejects any other size
        throw Error(KB_ERR_DIMS, "dimensions must agree: measurement (y)(" + std::to_string(rowsH) + "x...) H(" + std::to_string(p) + "x...)");
    auto d = std::make_shared<MonteCarloData>();
    d->runs = samples; d->steps = steps; d->n = kf.batch()->n(); d->p = p;
    for (const auto &c : controls) d->controls.insert(d->controls.end(), c.data.begin(), c.data.end());
    d->ncontrols = (int)controls.size();
    const bool single = kf.batch()->N() == 1 && samples > 1;
    d->truth = single ? Batch::Replicate(*kf.batch(), 0, samples, 0u) : kf.batch();
    if (keepRuns < 0) keepRuns = (size_t)steps * (size_t)(d->n + p) * (size_t)d->truth->N() * sizeof(double) <= kMonteCarloAutoKeepBytes ? 1 : 0;
    std::vector<double> sums((size_t)steps * 3 * d->n);
    check(kb_mc_run_ex(d->truth->handle(), steps, d->controls.data(), d->ncontrols, firstRun, sums.data(), keepRuns ? KB_MC_KEEP_RUNS : 0u));
    d->kept = keepRuns != 0;
    if (single) kf.Reset();
    d->mean.assign((size_t)steps * d->n, 0.0);
    d->stddev.assign((size_t)steps * d->n, 0.0);
    check(kb_mc_stats(sums.data(), steps, d->n, samples, d->mean.data(), d->stddev.data()));
    MonteCarloRuns mc{samples, steps, {}, d};
    if (d->kept) {
        mc.Runs.resize((size_t)d->truth->N());
        for (int64_t r = 0; r < d->truth->N(); r++) {
            mc.Runs[(size_t)r].Estimates.reserve((size_t)steps);
            for (int k = 0; k < steps; k++) mc.Runs[(size_t)r].Estimates.emplace_back(d, r, k);
        }
    }
    return mc;
}

// NewChiSquare(kf LDKF, runs MonteCarloRuns, controls, withNEES, withNIS) (NISmeans, NEESmeans, error)   chisquare.go:16-95.
// kf is the Vanilla filter under test -- one filter, which the reference Reset()s for every run (chisquare.go:39): every run of
// `runs` is replayed against its own copy of kf in one launch (kb_chisquare with replay_last_mc; the truth's states and
// measurements are regenerated from the runs' noise streams, they need not have been kept).
inline std::pair<std::vector<double>, std::vector<double>> NewChiSquare(LDKF &kf, const MonteCarloRuns &runs, const std::vector<Vector> &controls,
                                                                        bool withNEES, bool withNIS) {
    if (!withNEES && !withNIS) throw Error(KB_ERR_INVALID, "Chi Square requires either NEES or NIS or both");   // chisquare.go:17-19
    const int steps = runs.steps;
    if ((int)controls.size() != 1 && (int)controls.size() != steps)
        throw Error(KB_ERR_INVALID, "must provide as much control vectors as steps, or just one control vector");   // chisquare.go:35
    std::vector<double> ctrl;
    for (const auto &c : controls) ctrl.insert(ctrl.end(), c.data.begin(), c.data.end());
    const auto &truth = runs.data->truth;
    auto kfb = (kf.batch()->N() == 1 && truth->N() > 1) ? Batch::Replicate(*kf.batch(), 0, truth->N(), 0u) : kf.batch();
    std::vector<double> sums((size_t)steps * 2);
    check(kb_chisquare(truth->handle(), kfb->handle(), steps, ctrl.data(), (int)controls.size(), 0, 1, withNEES, withNIS, sums.data()));
    std::vector<double> nis((size_t)steps), nees((size_t)steps);
    const double n_runs = (double)truth->N();
    for (int t = 0; t < steps; t++) { nis[(size_t)t] = sums[(size_t)t * 2] / n_runs; nees[(size_t)t] = sums[(size_t)t * 2 + 1] / n_runs; }
    return {nis, nees};
}

// ---- one process, every GPU of the node (kb_sharded_* in gokalman_amd.h) ---------------------------
// N LDKF filters split into contiguous shards over the visible devices -- GPU g owns [g N / G, (g + 1) N / G) -- one handle, host
// thread and stream per device; Update has no collective, the Monte-Carlo / chi-square statistics are ONE ncclAllReduce over RCCL
// (host sum when the shards share a device).  Matrices as everywhere: one shared (rows x cols) or N of them back to back.
class ShardedBatch {
   public:
    ShardedBatch(int kind, const Vector &x0, const Matrix &P0, const Matrix &F, const Matrix &G, const Matrix &H, const Noise &noise,
                 int64_t N, std::vector<int> devices = {}, unsigned flags = 0)
        : n_(x0.rows), p_(H.rows), m_(G.cols), N_(N) {
        if (devices.empty())
            for (int g = 0; g < kb_device_count(); g++) devices.push_back(g);
        check(kb_sharded_create(&s_, kind, n_, p_, m_, N, KB_F64, devices.data(), (int)devices.size(), flags));
        auto set = [&](int field, const Matrix &mtx, int p_rows) {
            if (mtx.data.empty()) return;
            const int64_t per = (int64_t)mtx.rows * mtx.cols;
            check(kb_sharded_set(s_, field, mtx.data.data(), mtx.shared() ? 1 : N, mtx.shared() ? 1 : 0, p_rows, per));
        };
        set(KB_X, x0, 0); set(KB_P, P0, 0); set(KB_F, F, 0);
        if (m_ > 0) set(KB_G, G, 0);
        set(KB_H, H, H.rows); set(KB_Q, noise.Q, 0); set(KB_R, noise.R, noise.R.rows);
        if (noise.kind != KB_NOISE_NOISELESS) check(kb_sharded_set_noise_kind(s_, noise.kind, noise.seed));
        check(kb_sharded_init(s_));
    }
    ~ShardedBatch() { kb_sharded_destroy(s_); }
    ShardedBatch(const ShardedBatch &) = delete;
    ShardedBatch &operator=(const ShardedBatch &) = delete;
    kb_sharded *handle() const { return s_; }
    int Shards() const { return kb_sharded_num_shards(s_); }
    int64_t First(int g) const { return kb_sharded_first(s_, g); }
    int64_t N() const { return N_; }
    // LDKF.Update(measurement, control) on every filter, the shards in parallel: measurements [N][p] (or one vector for all)
    void Update(const Vector &measurement, const Vector &control = Vector()) {
        const std::vector<double> y = expand(measurement), u = expand(control);
        check(kb_sharded_update(s_, y.data(), measurement.rows, control.rows ? u.data() : nullptr, control.rows));
    }
    void Reset() { check(kb_sharded_reset(s_)); }
    Matrix State() const { return get(KB_STATE, n_, 1); }
    Matrix Covariance() const { return get(KB_COVAR, n_, n_); }
    std::vector<uint32_t> Status() const {
        std::vector<uint32_t> st((size_t)N_);
        check(kb_sharded_get_status(s_, st.data(), 0, N_));
        return st;
    }
    // NewMonteCarloRuns over the whole node (montecarlo.go:92-119): mean / stddev per step over ALL runs; usedRccl = how they were reduced
    struct Stats { int steps, n; std::vector<double> mean, stddev; bool usedRccl; };
    Stats MonteCarlo(int steps, const std::vector<Vector> &controls) {
        std::vector<double> ctrl;
        for (const auto &c : controls) ctrl.insert(ctrl.end(), c.data.begin(), c.data.end());
        std::vector<double> sums((size_t)steps * 3 * n_);
        check(kb_sharded_mc_run(s_, steps, ctrl.data(), (int)controls.size(), sums.data(), 0u));
        Stats st{steps, n_, std::vector<double>((size_t)steps * n_), std::vector<double>((size_t)steps * n_), kb_sharded_used_rccl(s_) != 0};
        check(kb_mc_stats(sums.data(), steps, n_, N_, st.mean.data(), st.stddev.data()));
        return st;
    }
    // NewChiSquare over the whole node (chisquare.go:16-95): (NISmeans, NEESmeans); `*this` is the truth (pure predictor, AWGN)
    std::pair<std::vector<double>, std::vector<double>> ChiSquare(ShardedBatch &kf, int steps, const std::vector<Vector> &controls, bool replayLastMC,
                                                                   bool withNEES = true, bool withNIS = true) {
        std::vector<double> ctrl;
        for (const auto &c : controls) ctrl.insert(ctrl.end(), c.data.begin(), c.data.end());
        std::vector<double> sums((size_t)steps * 2);
        check(kb_sharded_chisquare(s_, kf.s_, steps, ctrl.data(), (int)controls.size(), replayLastMC, withNEES, withNIS, sums.data()));
        std::vector<double> nis((size_t)steps), nees((size_t)steps);
        for (int t = 0; t < steps; t++) { nis[(size_t)t] = sums[(size_t)t * 2] / (double)N_; nees[(size_t)t] = sums[(size_t)t * 2 + 1] / (double)N_; }
        return {nis, nees};
    }

   private:
    Matrix get(int field, int rows, int cols) const {
        Matrix out(rows, cols);
        out.data.assign((size_t)N_ * rows * cols, 0.0);
        check(kb_sharded_get(s_, field, out.data.data(), 0, N_, (int64_t)rows * cols));
        return out;
    }
    std::vector<double> expand(const Vector &v) const {
        if (v.rows == 0) return {};
        if (!v.shared()) return v.data;
        std::vector<double> out;
        out.reserve((size_t)N_ * v.rows);
        for (int64_t i = 0; i < N_; i++) out.insert(out.end(), v.data.begin(), v.data.end());
        return out;
    }
    kb_sharded *s_ = nullptr;
    int n_, p_, m_;
    int64_t N_;
};

// VanLoan(A, Gamma, W, dt) (F, Q, error)   c2d.go:13-75.  The reference returns its Nyquist error NEXT to valid F and Q;
// here `nyquist` carries it ("gokalman: Nyquist sampling criterion not fulfilled with dt=...").
struct VanLoanResult { Matrix F, Q; bool nyquist = false; };
inline VanLoanResult VanLoan(const Matrix &A, const Matrix &Gamma, const Matrix &W, double dt, int device = 0) {
    if (A.rows != A.cols || Gamma.rows != A.rows || W.rows != W.cols || W.rows != Gamma.cols)
        throw Error(KB_ERR_DIMS, "dimensions must agree: A(nxn) Gamma(nxq) W(qxq)");
    VanLoanResult r{Matrix(A.rows, A.rows), Matrix(A.rows, A.rows), false};
    uint32_t st = 0;
    check(kb_van_loan(device, KB_F64, A.rows, Gamma.cols, 1, A.data.data(), Gamma.data.data(), W.data.data(), &dt, 15, r.F.data.data(),
                      r.Q.data.data(), &st));
    r.nyquist = (st & KB_ST_NYQUIST) != 0;
    return r;
}

}  // namespace gokalman
