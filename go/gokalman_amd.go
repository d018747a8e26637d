// Package gokalman_amd is the cgo shim a gokalman maintainer adds to route the predict/update
// hot path to the MI355X engine.  It binds exactly the C ABI in include/gokalman_amd.h and
// implements gokalman's own interfaces (kalman.go:35-72), so `examples/*/main.go` and the tests
// keep calling kf.Update(measurement, control).
//
// NOT COMPILED in this repository's CI: the build image has no Go toolchain and gonum is not
// vendored by the reference.  Build (on a box with Go >= 1.10, gonum and ROCm):
//   CGO_CFLAGS="-I${REPO}/include" CGO_LDFLAGS="-L${REPO}/gokalman_amd -lgokalman_amd" go build
package gokalman_amd

/*
#cgo LDFLAGS: -lgokalman_amd
#include <stdlib.h>
#include <string.h>
#include "gokalman_amd.h"

// The estimate view is built on the C side: a Go-allocated kb_estimate_view holding pointers into Go slices would be "a Go
// pointer to Go memory that contains unpinned Go pointers", which the cgo pointer rules forbid (cgocheck panics).  Passed as
// separate arguments, each slice pointer is an ordinary cgo argument and stays pinned for the duration of the call.
static int kbgo_get_estimate(kb_batch *b, int64_t first, int64_t count, double *state, double *covariance,
                             double *pred_covariance, double *gain, double *innovation, double *measurement,
                             uint32_t *status, int clear_status) {
    kb_estimate_view v;
    v.state = state; v.covariance = covariance; v.pred_covariance = pred_covariance; v.gain = gain;
    v.innovation = innovation; v.measurement = measurement; v.status = status; v.clear_status = clear_status;
    return kb_get_estimate(b, first, count, &v);
}
// The step and the estimate it returns in ONE call and one synchronisation (kb_update_estimate & co.); which = 0: LDKF.Update,
// 1: NLDKF.Update, 2: NLDKF.Predict
static int kbgo_step_estimate(kb_batch *b, int which, const double *a, int arows, const double *c, int crows, double *state,
                              double *covariance, double *pred_covariance, double *gain, double *innovation,
                              double *measurement, uint32_t *status) {
    kb_estimate_view v;
    v.state = state; v.covariance = covariance; v.pred_covariance = pred_covariance; v.gain = gain;
    v.innovation = innovation; v.measurement = measurement; v.status = status; v.clear_status = 1;
    if (which == 0) return kb_update_estimate(b, a, arows, c, crows, 0, 1, &v);
    if (which == 1) return kb_update_nl_estimate(b, a, arows, c, crows, 0, 1, &v);
    return kb_predict_nl_estimate(b, 0, 1, &v);
}
// kb_last_error() is thread-local: copy it out in the same C call frame's thread (see kbCall)
static void kbgo_last_error(char *dst, size_t n) {
    strncpy(dst, kb_last_error(), n - 1);
    dst[n - 1] = 0;
}
*/
import "C"

import (
	"errors"
	"fmt"
	"math"
	"runtime"
	"strings"
	"time"
	"unsafe"

	"github.com/ChristopherRabotin/gokalman"
	"github.com/gonum/matrix/mat64"
)

// kbCall runs one C-ABI call and, on failure, fetches the thread-local kb_last_error() from the SAME OS thread: the
// goroutine is pinned for both calls (without the lock the scheduler may move it in between and the message is lost).
func kbCall(call func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := call(); rc != C.KB_OK {
		var buf [512]C.char
		C.kbgo_last_error(&buf[0], C.size_t(len(buf)))
		return errors.New(C.GoString(&buf[0]))
	}
	return nil
}

func rowMajor(m mat64.Matrix) []float64 {
	r, c := m.Dims()
	out := make([]float64, r*c)
	for i := 0; i < r; i++ {
		for j := 0; j < c; j++ {
			out[i*c+j] = m.At(i, j)
		}
	}
	return out
}

func ptr(v []float64) *C.double {
	if len(v) == 0 {
		return nil
	}
	return (*C.double)(unsafe.Pointer(&v[0]))
}

// batch is one kb_batch holding N filters (N == 1 for the drop-in types below).
type batch struct {
	h       *C.kb_batch
	n, p, m int
	N       int64
}

func newBatch(kind C.int, n, p, m int, N int64, flags C.uint) (*batch, error) {
	b := &batch{n: n, p: p, m: m, N: N}
	if err := kbCall(func() C.int { return C.kb_create(&b.h, kind, C.int(n), C.int(p), C.int(m), C.int64_t(N), C.KB_F64, 0, flags) }); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(b, func(b *batch) { C.kb_destroy(b.h) })
	return b, nil
}

func (b *batch) set(field C.int, m mat64.Matrix, pRows int) error {
	v := rowMajor(m)
	if len(v) == 0 { // an n x 0 input-control matrix: nothing to upload, needCtrl stays false (vanilla.go:39)
		return nil
	}
	return kbCall(func() C.int { return C.kb_set(b.h, field, ptr(v), 1, 1, C.int(pRows)) })
}

func (b *batch) get(field C.int, rows, cols int) []float64 {
	out := make([]float64, rows*cols)
	if err := kbCall(func() C.int { return C.kb_get(b.h, field, ptr(out), 0, 1) }); err != nil {
		panic(err)
	}
	return out
}

// Estimate implements gokalman.Estimate (kalman.go:64-72).  It is an immutable VALUE, as in the reference, whose Update
// returns a freshly allocated estimate that callers keep (vanilla.go:216-218; examples/jerkcar/main.go:71-90 sends it
// through a channel to another goroutine, montecarlo.go:108-117 stores one per step): every member is copied out of HBM
// once, by ONE kb_get_estimate call, when the estimate is created.
type Estimate struct {
	n, p                   int
	kind                   C.int
	state, meas, innov     []float64
	covar, predCovar, gain []float64
	status                 uint32
}

// snapshot downloads the current estimate of filter 0 of b and reads-and-clears its status word, so that one failed
// Update does not poison the next call (vanilla.go:164-167 returns an error and leaves prevEst alone).
func snapshot(b *batch, kind C.int) (*Estimate, error) { return stepSnapshot(b, kind, -1, nil, nil) }

// stepSnapshot runs one step (which = 0 LDKF.Update(a, c), 1 NLDKF.Update(a, c), 2 NLDKF.Predict(); -1: no step) and snapshots the
// estimate it produced, with ONE device synchronisation for both.
func stepSnapshot(b *batch, kind C.int, which int, a, c []float64) (*Estimate, error) {
	n, p := b.n, int(C.kb_meas_dim(b.h))
	info := kind == C.KB_INFORMATION || kind == C.KB_SRIF
	lazy := info || kind == C.KB_SQUAREROOT
	ni := p
	if info {
		ni = n // Innovation() returns the information vector (information.go:272, srif.go:237)
	}
	e := &Estimate{n: n, p: p, kind: kind, state: make([]float64, n), covar: make([]float64, n*n), predCovar: make([]float64, n*n),
		meas: make([]float64, p), innov: make([]float64, ni)}
	if !lazy || kind == C.KB_SQUAREROOT {
		e.gain = make([]float64, n*p)
	}
	var st C.uint32_t
	// every slice pointer is a direct cgo argument (pinned for the call); the view struct itself lives in C (see the preamble)
	if err := kbCall(func() C.int {
		if which < 0 {
			return C.kbgo_get_estimate(b.h, 0, 1, ptr(e.state), ptr(e.covar), ptr(e.predCovar), ptr(e.gain), ptr(e.innov), ptr(e.meas), &st, 1)
		}
		return C.kbgo_step_estimate(b.h, C.int(which), ptr(a), C.int(len(a)), ptr(c), C.int(len(c)), ptr(e.state), ptr(e.covar),
			ptr(e.predCovar), ptr(e.gain), ptr(e.innov), ptr(e.meas), &st)
	}); err != nil {
		return nil, err // "dimensions must agree: ...", "kf is locked (call Prepare() first)"
	}
	e.status = uint32(st)
	return e, nil
}

func (e *Estimate) State() *mat64.Vector            { return mat64.NewVector(e.n, e.state) }
func (e *Estimate) Measurement() *mat64.Vector      { return mat64.NewVector(e.p, e.meas) }
func (e *Estimate) Innovation() *mat64.Vector       { return mat64.NewVector(len(e.innov), e.innov) }
func (e *Estimate) Covariance() mat64.Symmetric     { return mat64.NewSymDense(e.n, e.covar) }
func (e *Estimate) PredCovariance() mat64.Symmetric { return mat64.NewSymDense(e.n, e.predCovar) }
func (e *Estimate) Gain() mat64.Matrix              { return mat64.NewDense(e.n, e.p, e.gain) }

// isWithin: vanilla.go:231-239.
func isWithin(state, covar []float64, n int, N float64) bool {
	for i := 0; i < n; i++ {
		nσ := N * math.Sqrt(covar[i*n+i])
		if state[i] > nσ || state[i] < -nσ {
			return false
		}
	}
	return true
}
func (e *Estimate) IsWithinNσ(N float64) bool { return isWithin(e.state, e.covar, e.n, N) }
func (e *Estimate) IsWithin2σ() bool          { return e.IsWithinNσ(2) }

// String prints what the reference's estimate of the same kind prints (vanilla.go:276-284, squareroot.go:347-355,
// hybrid.go:300-308; information.go:318-325 without the gain; srif.go:283-289 without gain and innovation).
func (e *Estimate) String() string {
	state := mat64.Formatted(e.State(), mat64.Prefix("  "))
	meas := mat64.Formatted(e.Measurement(), mat64.Prefix("  "))
	covar := mat64.Formatted(e.Covariance(), mat64.Prefix("  "))
	predp := mat64.Formatted(e.PredCovariance(), mat64.Prefix("   "))
	switch e.kind {
	case C.KB_SRIF:
		return fmt.Sprintf("{\ns=%v\ny=%v\nP=%v\nP-=%v\n}", state, meas, covar, predp)
	case C.KB_INFORMATION:
		innov := mat64.Formatted(e.Innovation(), mat64.Prefix("  "))
		return fmt.Sprintf("{\ns=%v\ny=%v\nP=%v\nP-=%v\ni=%v\n}", state, meas, covar, predp, innov)
	}
	gain := mat64.Formatted(e.Gain(), mat64.Prefix("  "))
	innov := mat64.Formatted(e.Innovation(), mat64.Prefix("  "))
	return fmt.Sprintf("{\ns=%v\ny=%v\nP=%v\nK=%v\nP-=%v\ni=%v\n}", state, meas, covar, gain, predp, innov)
}

// stepError turns the status word of the step that just ran into the reference's error value.  vanilla.go:166 prints no
// step ("could not invert `H*P_kp1_minus*H' + R`: %s"); srif.go:113 and hybrid.go:151 do ("... at k=%d: %s").  The engine
// does not advance kf.step on a failed call (kb_step), exactly like the reference, so `step` IS the k of the failed step.
func stepError(st uint32, what string, withStep bool, step int64) error {
	if st&C.KB_ST_SINGULAR != 0 {
		if withStep {
			return fmt.Errorf("could not invert %s at k=%d: matrix singular or near-singular", what, step)
		}
		return fmt.Errorf("could not invert %s: matrix singular or near-singular", what)
	}
	if st&(C.KB_ST_ASYMMETRIC|C.KB_ST_NONFINITE) != 0 {
		return errors.New("matrix is not symmetric") // helper.go:76
	}
	return nil
}

// ldkf implements gokalman.LDKF (kalman.go:35-47) with the device engine behind it; Vanilla, SquareRoot and Information are
// the reference's three LDKF types over it.
type ldkf struct {
	b       *batch
	kind    C.int
	F, G, H mat64.Matrix
	Noise   gokalman.Noise
}

// Vanilla mirrors gokalman.Vanilla (vanilla.go:65-74), SquareRoot squareroot.go:53-63, Information information.go:84-95.
type Vanilla struct{ ldkf }
type SquareRoot struct{ ldkf }
type Information struct{ ldkf }

// handle gives NewChiSquare / NewMonteCarloRuns the device batch behind any of the LDKF types of this package.
type batchHolder interface{ handle() *batch }

func (kf *ldkf) handle() *batch { return kf.b }

// LastKernel names the kernel instantiation(s) the last Update of this filter ran on (kb_last_kernel: a debugging / reporting aid with
// no counterpart in kalman.go:35-72).
func LastKernel(kf batchHolder) string { return C.GoString(C.kb_last_kernel(kf.handle().h)) }

// Update implements LDKF.Update (vanilla.go:128-220, squareroot.go:129-274, information.go:153-227): one launch of the HIP
// step kernel, then one snapshot.
func (kf *ldkf) Update(measurement, control *mat64.Vector) (gokalman.Estimate, error) {
	y := rowMajor(measurement)
	var u []float64
	if control != nil {
		u = rowMajor(control)
	}
	est, err := stepSnapshot(kf.b, kf.kind, 0, y, u) // kb_update_estimate: the step and its estimate, one synchronisation
	if err != nil {
		return nil, err // "dimensions must agree: ..." (vanilla.go:129-135)
	}
	if err := stepError(est.status, "`H*P_kp1_minus*H' + R`", false, 0); err != nil {
		return nil, err // the filter kept its previous estimate and its kf.step; the next Update runs normally
	}
	return est, nil
}
func (kf *ldkf) GetNoise() gokalman.Noise            { return kf.Noise }
func (kf *ldkf) GetStateTransition() mat64.Matrix    { return kf.F }
func (kf *ldkf) GetInputControl() mat64.Matrix       { return kf.G }
func (kf *ldkf) GetMeasurementMatrix() mat64.Matrix  { return kf.H }
func (kf *ldkf) SetStateTransition(F mat64.Matrix)   { kf.F = F; kf.b.set(C.KB_F, F, 0) }
func (kf *ldkf) SetInputControl(G mat64.Matrix)      { kf.G = G; kf.b.set(C.KB_G, G, 0) }
func (kf *ldkf) SetMeasurementMatrix(H mat64.Matrix) { kf.H = H; p, _ := H.Dims(); kf.b.set(C.KB_H, H, p) }
func (kf *ldkf) SetNoise(n gokalman.Noise) {
	kf.Noise = n
	p, _ := n.MeasurementMatrix().Dims()
	kf.b.set(C.KB_Q, n.ProcessMatrix(), 0)
	kf.b.set(C.KB_R, n.MeasurementMatrix(), p)
}
func (kf *ldkf) Reset() { C.kb_reset(kf.b.h) }

// String is vanilla.go:76-78 / squareroot.go:65-67 (information.go:96-98 prints inv(F) instead of F).
func (kf *ldkf) String() string {
	if kf.kind == C.KB_INFORMATION {
		var finv mat64.Dense
		if err := finv.Inverse(kf.F); err == nil {
			return fmt.Sprintf("inv(F)=%v\nG=%v\nH=%v\n%s", mat64.Formatted(&finv, mat64.Prefix("      ")), mat64.Formatted(kf.G, mat64.Prefix("  ")), mat64.Formatted(kf.H, mat64.Prefix("  ")), kf.Noise)
		}
	}
	return fmt.Sprintf("F=%v\nG=%v\nH=%v\n%s", mat64.Formatted(kf.F, mat64.Prefix("  ")), mat64.Formatted(kf.G, mat64.Prefix("  ")), mat64.Formatted(kf.H, mat64.Prefix("  ")), kf.Noise)
}

var _ gokalman.LDKF = (*Vanilla)(nil)
var _ gokalman.LDKF = (*SquareRoot)(nil)
var _ gokalman.LDKF = (*Information)(nil)
var _ gokalman.Estimate = (*Estimate)(nil)

// newLDKF is the shared constructor body of the LDKF kinds (vanilla.go:21, squareroot.go:21,
// information.go:20/65): kind selects the device kernels, flags carries INFO_FROM_STATE.
func newLDKF(kind C.int, flags C.uint, x0 *mat64.Vector, P0 mat64.Symmetric, F, G, H mat64.Matrix, noise gokalman.Noise) (ldkf, *Estimate, error) {
	n, _ := x0.Dims()
	p, _ := H.Dims()
	_, m := G.Dims()
	b, err := newBatch(kind, n, p, m, 1, C.KB_FLAG_FULL_ESTIMATE|flags)
	if err != nil {
		return ldkf{}, nil, err
	}
	for _, s := range []struct {
		f C.int
		m mat64.Matrix
		p int
	}{{C.KB_X, x0, 0}, {C.KB_P, P0, 0}, {C.KB_F, F, 0}, {C.KB_G, G, 0}, {C.KB_H, H, p},
		{C.KB_Q, noise.ProcessMatrix(), 0}, {C.KB_R, noise.MeasurementMatrix(), p}} {
		if err := b.set(s.f, s.m, s.p); err != nil {
			return ldkf{}, nil, err
		}
	}
	if _, isAWGN := noise.(*gokalman.AWGN); isAWGN {
		if err := kbCall(func() C.int { return C.kb_set_noise_kind(b.h, C.KB_NOISE_AWGN, C.uint64_t(time.Now().UnixNano())) }); err != nil {
			return ldkf{}, nil, err // "process noise invalid" (noise.go:148-156 panics there)
		}
	}
	if err := kbCall(func() C.int { return C.kb_init(b.h) }); err != nil {
		return ldkf{}, nil, err
	}
	est0, err := snapshot(b, kind)
	if err != nil {
		return ldkf{}, nil, err
	}
	return ldkf{b, kind, F, G, H, noise}, est0, nil
}

// NewVanilla mirrors gokalman.NewVanilla (vanilla.go:21-40).
func NewVanilla(x0 *mat64.Vector, Covar0 mat64.Symmetric, F, G, H mat64.Matrix, noise gokalman.Noise) (*Vanilla, *Estimate, error) {
	kf, est0, err := newLDKF(C.KB_VANILLA, 0, x0, Covar0, F, G, H, noise)
	if err != nil {
		return nil, nil, err
	}
	return &Vanilla{kf}, est0, nil
}

// NewPurePredictorVanilla mirrors vanilla.go:43-62.
func NewPurePredictorVanilla(x0 *mat64.Vector, P0 mat64.Symmetric, F, G, H mat64.Matrix, n gokalman.Noise) (*Vanilla, *Estimate, error) {
	kf, est0, err := newLDKF(C.KB_VANILLA_PREDICT, 0, x0, P0, F, G, H, n)
	if err != nil {
		return nil, nil, err
	}
	return &Vanilla{kf}, est0, nil
}

// NewSquareRoot mirrors squareroot.go:21-50.
func NewSquareRoot(x0 *mat64.Vector, P0 mat64.Symmetric, F, G, H mat64.Matrix, n gokalman.Noise) (*SquareRoot, *Estimate, error) {
	kf, est0, err := newLDKF(C.KB_SQUAREROOT, 0, x0, P0, F, G, H, n)
	if err != nil {
		return nil, nil, err
	}
	return &SquareRoot{kf}, est0, nil
}

// NewInformation mirrors information.go:20-53 (i0, I0); NewInformationFromState information.go:65-81 (x0, P0).
func NewInformation(i0 *mat64.Vector, I0 mat64.Symmetric, F, G, H mat64.Matrix, n gokalman.Noise) (*Information, *Estimate, error) {
	kf, est0, err := newLDKF(C.KB_INFORMATION, 0, i0, I0, F, G, H, n)
	if err != nil {
		return nil, nil, err
	}
	return &Information{kf}, est0, nil
}
func NewInformationFromState(x0 *mat64.Vector, P0 mat64.Symmetric, F, G, H mat64.Matrix, n gokalman.Noise) (*Information, *Estimate, error) {
	kf, est0, err := newLDKF(C.KB_INFORMATION, C.KB_FLAG_INFO_FROM_STATE, x0, P0, F, G, H, n)
	if err != nil {
		return nil, nil, err
	}
	return &Information{kf}, est0, nil
}

// nldkf implements gokalman.NLDKF (kalman.go:51-60); SRIF and HybridKF are the reference's two types over it.
type nldkf struct {
	b     *batch
	kind  C.int
	Noise gokalman.Noise
}
type SRIF struct{ nldkf }
type HybridKF struct{ nldkf }

func newNLDKF(kind C.int, x0 *mat64.Vector, P0 mat64.Symmetric, noise gokalman.Noise, measSize int, flags C.uint) (nldkf, *Estimate, error) {
	n, _ := x0.Dims()
	q := 0
	if kind == C.KB_HYBRID {
		q, _ = noise.ProcessMatrix().Dims()
	}
	b, err := newBatch(kind, n, measSize, q, 1, C.KB_FLAG_FULL_ESTIMATE|flags)
	if err != nil {
		return nldkf{}, nil, err
	}
	if err := b.set(C.KB_X, x0, 0); err != nil {
		return nldkf{}, nil, err
	}
	if err := b.set(C.KB_P, P0, 0); err != nil {
		return nldkf{}, nil, err
	}
	if err := b.set(C.KB_R, noise.MeasurementMatrix(), measSize); err != nil {
		return nldkf{}, nil, err
	}
	if q > 0 {
		if err := b.set(C.KB_Q, noise.ProcessMatrix(), 0); err != nil {
			return nldkf{}, nil, err
		}
	}
	if err := kbCall(func() C.int { return C.kb_init(b.h) }); err != nil {
		return nldkf{}, nil, err
	}
	est0, err := snapshot(b, kind)
	if err != nil {
		return nldkf{}, nil, err
	}
	return nldkf{b: b, kind: kind, Noise: noise}, est0, nil
}

// NewSRIF mirrors gokalman.NewSRIF (srif.go:14-49); NewHybridKF mirrors hybrid.go:23-34.
func NewSRIF(x0 *mat64.Vector, P0 mat64.Symmetric, measSize int, nonTriR bool, n gokalman.Noise) (*SRIF, *Estimate, error) {
	var fl C.uint
	if nonTriR {
		fl = C.KB_FLAG_SRIF_NON_TRI_R
	}
	p, _ := n.MeasurementMatrix().Dims()
	_ = measSize // only sizes Predict()'s zero vectors in the reference
	kf, est0, err := newNLDKF(C.KB_SRIF, x0, P0, n, p, fl)
	if err != nil {
		return nil, nil, err
	}
	return &SRIF{kf}, est0, nil
}
func NewHybridKF(x0 *mat64.Vector, P0 mat64.Symmetric, n gokalman.Noise, measSize int) (*HybridKF, *Estimate, error) {
	kf, est0, err := newNLDKF(C.KB_HYBRID, x0, P0, n, measSize, 0)
	if err != nil {
		return nil, nil, err
	}
	return &HybridKF{kf}, est0, nil
}

func (kf *nldkf) Prepare(Φ, Htilde *mat64.Dense) { // srif.go:82-86, hybrid.go:78-82
	phi, h := rowMajor(Φ), rowMajor(Htilde)
	if err := kbCall(func() C.int { return C.kb_prepare(kf.b.h, ptr(phi), ptr(h), 1, 1) }); err != nil {
		panic(err)
	}
}
func (kf *nldkf) PreparePNT(Γ *mat64.Dense) { // hybrid.go:86-89
	g := rowMajor(Γ)
	if err := kbCall(func() C.int { return C.kb_prepare_pnt(kf.b.h, ptr(g), 1, 1) }); err != nil {
		panic(err)
	}
}
func (kf *nldkf) whatFailed() string {
	if kf.kind == C.KB_SRIF {
		return "`Φ`" // srif.go:113
	}
	return "`H*P_kp1_minus*H' + R`" // hybrid.go:151
}
func (kf *nldkf) stepEstimate(which int, a, c []float64) (gokalman.Estimate, error) {
	est, err := stepSnapshot(kf.b, kf.kind, which, a, c)
	if err != nil {
		return nil, err // "kf is locked (call Prepare() first)", "dimensions must agree: ..."
	}
	// kb_step is kf.step: the failed call did not advance it, so it is the k srif.go:113 / hybrid.go:151 print
	if err := stepError(est.status, kf.whatFailed(), true, int64(C.kb_step(kf.b.h))); err != nil {
		return nil, err
	}
	return est, nil
}
func (kf *nldkf) Update(realObservation, computedObservation *mat64.Vector) (gokalman.Estimate, error) { // srif.go:90, hybrid.go:93
	return kf.stepEstimate(1, rowMajor(realObservation), rowMajor(computedObservation))
}
func (kf *nldkf) Predict() (gokalman.Estimate, error) { // srif.go:96, hybrid.go:99
	return kf.stepEstimate(2, nil, nil)
}
func (kf *nldkf) EKFEnabled() bool { return C.kb_ekf_enabled(kf.b.h) != 0 }
func (kf *nldkf) EnableEKF()       { C.kb_set_ekf(kf.b.h, 1) }
func (kf *nldkf) DisableEKF()      { C.kb_set_ekf(kf.b.h, 0) }

// String is hybrid.go:63-65 (the reference's SRIF has no String of its own).
func (kf *nldkf) String() string {
	if kf.kind == C.KB_HYBRID {
		return fmt.Sprintf("HybridKF [k=%d]\n%s", int64(C.kb_step(kf.b.h)), kf.Noise)
	}
	return fmt.Sprintf("SRIF [k=%d]", int64(C.kb_step(kf.b.h)))
}
func (kf *nldkf) SetNoise(n gokalman.Noise) {
	if kf.kind == C.KB_SRIF {
		panic("noise not yet supported for SRIF") // srif.go:76-78
	}
	kf.Noise = n
	p, _ := n.MeasurementMatrix().Dims()
	kf.b.set(C.KB_R, n.MeasurementMatrix(), p)
}

var _ gokalman.NLDKF = (*SRIF)(nil)
var _ gokalman.NLDKF = (*HybridKF)(nil)

// ---- Monte-Carlo runs (montecarlo.go:11-124) and chi-square (chisquare.go:16-95) ------------------------------------------

// mcData is what the runs of one ensemble share.  The runs live on the device; the host holds the per-step statistics and,
// when the ensemble was kept (kb_mc_run_ex with KB_MC_KEEP_RUNS), downloads every run's State() / Measurement() once, on
// first use.  Covariance / PredCovariance / Gain of a pure predictor's estimates do not depend on the noise: they come from
// one Noiseless copy of the filter stepped through the controls.
type mcData struct {
	runs, steps, n, p int
	mean, stddev      []float64 // [steps][n]
	truth             *batch    // one pure-predictor AWGN filter per run
	controls          []float64
	ncontrols         int
	kept              bool
	states, meas      []float64 // [runs][steps][n], [runs][steps][p]
	ppred, gain       []float64 // [steps][n][n], [steps][n][p]
}

func (d *mcData) download() {
	if !d.kept {
		panic("these Monte-Carlo runs were not kept (the ensemble is above the size NewMonteCarloRuns keeps): only Mean / StdDev / NewChiSquare are available")
	}
	if d.states != nil {
		return
	}
	N := int(d.truth.N)
	st, me := make([]float64, N*d.steps*d.n), make([]float64, N*d.steps*d.p)
	if err := kbCall(func() C.int { return C.kb_mc_get_runs(d.truth.h, 0, C.int64_t(N), ptr(st), ptr(me)) }); err != nil {
		panic(err)
	}
	d.states, d.meas = st, me
}

func (d *mcData) shared() {
	if d.ppred != nil {
		return
	}
	one := &batch{n: d.n, p: d.p, N: 1}
	if err := kbCall(func() C.int { return C.kb_replicate(d.truth.h, 0, 1, C.KB_FLAG_FULL_ESTIMATE, &one.h) }); err != nil {
		panic(err)
	}
	defer C.kb_destroy(one.h)
	C.kb_set_noise_kind(one.h, C.KB_NOISE_NOISELESS, 0)
	m := 0
	if d.ncontrols > 0 {
		m = len(d.controls) / d.ncontrols
	}
	ctrl := C.kb_need_ctrl(one.h) != 0
	y0, u0 := make([]float64, d.p), make([]float64, m+1)
	P, K := make([]float64, d.steps*d.n*d.n), make([]float64, d.steps*d.n*d.p)
	for t := 0; t < d.steps; t++ {
		var up *C.double
		nu := 0
		if ctrl {
			up, nu = ptr(u0), m
			if d.ncontrols != 1 {
				up = ptr(d.controls[t*m : (t+1)*m])
			}
		}
		if err := kbCall(func() C.int { return C.kb_update(one.h, ptr(y0), C.int(d.p), up, C.int(nu)) }); err != nil {
			panic(err)
		}
		if err := kbCall(func() C.int {
			return C.kbgo_get_estimate(one.h, 0, 1, nil, nil, ptr(P[t*d.n*d.n:(t+1)*d.n*d.n]), ptr(K[t*d.n*d.p:(t+1)*d.n*d.p]), nil, nil, nil, 0)
		}); err != nil {
			panic(err)
		}
	}
	d.ppred, d.gain = P, K
}

// mcEstimate is MonteCarloRun.Estimates[k] (montecarlo.go:108-117): what a pure-predictor Vanilla returns
// (vanilla.go:170-179): {x-, yhat, 0, sym(P-), sym(P-), K}.
type mcEstimate struct {
	d    *mcData
	r, k int
}

func (e *mcEstimate) State() *mat64.Vector {
	e.d.download()
	o := (e.r*e.d.steps + e.k) * e.d.n
	return mat64.NewVector(e.d.n, e.d.states[o:o+e.d.n])
}
func (e *mcEstimate) Measurement() *mat64.Vector {
	e.d.download()
	o := (e.r*e.d.steps + e.k) * e.d.p
	return mat64.NewVector(e.d.p, e.d.meas[o:o+e.d.p])
}
func (e *mcEstimate) Innovation() *mat64.Vector   { return mat64.NewVector(e.d.p, nil) }
func (e *mcEstimate) Covariance() mat64.Symmetric { return e.PredCovariance() }
func (e *mcEstimate) PredCovariance() mat64.Symmetric {
	e.d.shared()
	nn := e.d.n * e.d.n
	return mat64.NewSymDense(e.d.n, e.d.ppred[e.k*nn:(e.k+1)*nn])
}
func (e *mcEstimate) Gain() mat64.Matrix {
	e.d.shared()
	np := e.d.n * e.d.p
	return mat64.NewDense(e.d.n, e.d.p, e.d.gain[e.k*np:(e.k+1)*np])
}
func (e *mcEstimate) IsWithinNσ(N float64) bool {
	e.d.download()
	e.d.shared()
	o, nn := (e.r*e.d.steps+e.k)*e.d.n, e.d.n*e.d.n
	return isWithin(e.d.states[o:o+e.d.n], e.d.ppred[e.k*nn:(e.k+1)*nn], e.d.n, N)
}
func (e *mcEstimate) String() string { // vanilla.go:276-284
	return fmt.Sprintf("{\ns=%v\ny=%v\nP=%v\nK=%v\nP-=%v\ni=%v\n}", mat64.Formatted(e.State(), mat64.Prefix("  ")),
		mat64.Formatted(e.Measurement(), mat64.Prefix("  ")), mat64.Formatted(e.Covariance(), mat64.Prefix("  ")),
		mat64.Formatted(e.Gain(), mat64.Prefix("  ")), mat64.Formatted(e.PredCovariance(), mat64.Prefix("   ")),
		mat64.Formatted(e.Innovation(), mat64.Prefix("  ")))
}

var _ gokalman.Estimate = (*mcEstimate)(nil)

// MonteCarloRun stores the results of an MC run (montecarlo.go:122-124).
type MonteCarloRun struct {
	Estimates []gokalman.Estimate
}

// MonteCarloRuns stores MC runs (montecarlo.go:11-15).  Runs is empty when the ensemble is too large to be kept (Mean, StdDev
// and NewChiSquare do not need it).
type MonteCarloRuns struct {
	runs, steps int
	Runs        []MonteCarloRun
	d           *mcData
}

// Mean / StdDev: montecarlo.go:18-59 (stat.Mean, stat.StdDev = the unbiased estimator), reduced on the device.
func (mc MonteCarloRuns) Mean(step int) []float64   { return mc.d.mean[step*mc.d.n : (step+1)*mc.d.n] }
func (mc MonteCarloRuns) StdDev(step int) []float64 { return mc.d.stddev[step*mc.d.n : (step+1)*mc.d.n] }

// AsCSV is used as a CSV serializer. Does not include the header (montecarlo.go:62-89, same strings).
func (mc MonteCarloRuns) AsCSV(headers []string) []string {
	mc.d.download()
	d := mc.d
	N := int(d.truth.N)
	rtn := make([]string, d.n)
	for i := 0; i < d.n; i++ {
		header := headers[i]
		lines := make([]string, d.steps+1)
		var sb strings.Builder
		for rNo := 0; rNo < N; rNo++ {
			fmt.Fprintf(&sb, "%s-%d,", header, rNo)
		}
		lines[0] = sb.String() + header + "-mean," + header + "-stddev"
		for k := 0; k < d.steps; k++ {
			sb.Reset()
			for rNo := 0; rNo < N; rNo++ {
				fmt.Fprintf(&sb, "%f,", d.states[(rNo*d.steps+k)*d.n+i])
			}
			fmt.Fprintf(&sb, "%f,%f", d.mean[k*d.n+i], d.stddev[k*d.n+i])
			lines[k+1] = sb.String()
		}
		rtn[i] = strings.Join(lines, "\n")
	}
	return rtn
}

func flattenControls(controls []*mat64.Vector) []float64 {
	var out []float64
	for _, c := range controls {
		out = append(out, rowMajor(c)...)
	}
	return out
}

// mcAutoKeepBytes: ensembles whose trajectories fit are kept for Runs / AsCSV (the reference's 50 x 120 and 15 x 1086 are a few
// hundred KB); above it only the statistics are.
const mcAutoKeepBytes = 256 << 20

// NewMonteCarloRuns run monte carlos on the provided filter: gokalman.NewMonteCarloRuns(samples, steps, rowsH, controls, kf)
// (montecarlo.go:92-119), same signature, same panics.  kf is ONE pure-predictor Vanilla; the `samples` runs the reference
// performs one after the other on it (Reset() in between) are `samples` copies of it on the device (kb_replicate) advanced by
// one launch (kb_mc_run_ex), and kf is left Reset() as montecarlo.go:116 leaves it.
func NewMonteCarloRuns(samples, steps, rowsH int, controls []*mat64.Vector, kf *Vanilla) MonteCarloRuns {
	if kf.kind != C.KB_VANILLA_PREDICT {
		panic("the Kalman filter needed for the Monte Carlo runs must be a pure predictor") // montecarlo.go:93-95
	}
	if len(controls) != 1 && len(controls) != steps {
		panic("must provide as much control vectors as steps, or just one control vector") // montecarlo.go:105-107
	}
	p := int(C.kb_meas_dim(kf.b.h))
	if rowsH != p { // montecarlo.go:111 hands Update a zero vector of rowsH rows; vanilla.go:133-135 rejects any other size
		panic(fmt.Sprintf("dimensions must agree: measurement (y)(%dx...) H(%dx...)", rowsH, p))
	}
	truth := &batch{n: kf.b.n, p: p, m: kf.b.m, N: int64(samples)}
	if err := kbCall(func() C.int { return C.kb_replicate(kf.b.h, 0, C.int64_t(samples), 0, &truth.h) }); err != nil {
		panic(err)
	}
	runtime.SetFinalizer(truth, func(b *batch) { C.kb_destroy(b.h) })
	d := &mcData{runs: samples, steps: steps, n: kf.b.n, p: p, truth: truth, controls: flattenControls(controls), ncontrols: len(controls)}
	d.kept = steps*(d.n+p)*samples*8 <= mcAutoKeepBytes
	var flags C.uint
	if d.kept {
		flags = C.KB_MC_KEEP_RUNS
	}
	sums := make([]float64, steps*3*d.n)
	if err := kbCall(func() C.int {
		return C.kb_mc_run_ex(truth.h, C.int(steps), ptr(d.controls), C.int(d.ncontrols), 0, ptr(sums), flags)
	}); err != nil {
		panic(err) // e.g. "Monte-Carlo runs need AWGN noise"
	}
	kf.Reset() // montecarlo.go:116
	d.mean, d.stddev = make([]float64, steps*d.n), make([]float64, steps*d.n)
	if err := kbCall(func() C.int {
		return C.kb_mc_stats(ptr(sums), C.int(steps), C.int(d.n), C.int64_t(samples), ptr(d.mean), ptr(d.stddev))
	}); err != nil {
		panic(err)
	}
	mc := MonteCarloRuns{runs: samples, steps: steps, d: d}
	if d.kept {
		mc.Runs = make([]MonteCarloRun, samples)
		for r := range mc.Runs {
			mc.Runs[r].Estimates = make([]gokalman.Estimate, steps)
			for k := 0; k < steps; k++ {
				mc.Runs[r].Estimates[k] = &mcEstimate{d, r, k}
			}
		}
	}
	return mc
}

// NewChiSquare runs the Chi square tests from the MonteCarlo runs: gokalman.NewChiSquare(kf, runs, controls, withNEES,
// withNIS) (chisquare.go:16-95), same signature and return order (NISmeans, NEESmeans, error).  kf must be one of this
// package's Vanilla filters; the reference Reset()s it for every run (chisquare.go:39): here every run of `runs` is replayed
// against its own copy of kf in one launch (kb_chisquare with replay_last_mc: the truth's states and measurements are
// regenerated from the runs' noise streams).
func NewChiSquare(kf gokalman.LDKF, runs MonteCarloRuns, controls []*mat64.Vector, withNEES, withNIS bool) ([]float64, []float64, error) {
	if !withNEES && !withNIS {
		return nil, nil, errors.New("Chi Square requires either NEES or NIS or both") // chisquare.go:17-19
	}
	steps := runs.steps
	if len(controls) != 1 && len(controls) != steps {
		return nil, nil, errors.New("must provide as much control vectors as steps, or just one control vector") // chisquare.go:35
	}
	holder, ok := kf.(batchHolder)
	if !ok {
		return nil, nil, errors.New("NewChiSquare: kf must be a filter of package gokalman_amd")
	}
	truth := runs.d.truth
	var kfb *C.kb_batch
	if err := kbCall(func() C.int { return C.kb_replicate(holder.handle().h, 0, C.int64_t(truth.N), 0, &kfb) }); err != nil {
		return nil, nil, err
	}
	defer C.kb_destroy(kfb)
	ctrl := flattenControls(controls)
	sums := make([]float64, steps*2)
	b2i := func(v bool) C.int {
		if v {
			return 1
		}
		return 0
	}
	if err := kbCall(func() C.int {
		return C.kb_chisquare(truth.h, kfb, C.int(steps), ptr(ctrl), C.int(len(controls)), 0, 1, b2i(withNEES), b2i(withNIS), ptr(sums))
	}); err != nil {
		return nil, nil, err
	}
	nis, nees := make([]float64, steps), make([]float64, steps)
	for k := 0; k < steps; k++ {
		nis[k], nees[k] = sums[2*k]/float64(truth.N), sums[2*k+1]/float64(truth.N)
	}
	return nis, nees, nil
}

// BatchLDKF is N independent LDKF filters sharing one model (or per-filter models through SetPerFilter) behind one
// handle: the reference's `for _, kf := range filters { kf.Update(y, u) }` as one launch.  Measurements are [N][p].
type BatchLDKF struct {
	b    *batch
	kind C.int
}

func NewBatchLDKF(kind C.int, N int64, x0 *mat64.Vector, P0 mat64.Symmetric, F, G, H mat64.Matrix, noise gokalman.Noise) (*BatchLDKF, error) {
	n, _ := x0.Dims()
	p, _ := H.Dims()
	_, m := G.Dims()
	b, err := newBatch(kind, n, p, m, N, 0) // state-only outputs: the batch path reads what it needs with Estimates()
	if err != nil {
		return nil, err
	}
	for _, s := range []struct {
		f C.int
		m mat64.Matrix
		p int
	}{{C.KB_X, x0, 0}, {C.KB_P, P0, 0}, {C.KB_F, F, 0}, {C.KB_G, G, 0}, {C.KB_H, H, p},
		{C.KB_Q, noise.ProcessMatrix(), 0}, {C.KB_R, noise.MeasurementMatrix(), p}} {
		if err := b.set(s.f, s.m, s.p); err != nil {
			return nil, err
		}
	}
	if _, isAWGN := noise.(*gokalman.AWGN); isAWGN {
		if err := kbCall(func() C.int { return C.kb_set_noise_kind(b.h, C.KB_NOISE_AWGN, C.uint64_t(time.Now().UnixNano())) }); err != nil {
			return nil, err
		}
	}
	if err := kbCall(func() C.int { return C.kb_init(b.h) }); err != nil {
		return nil, err
	}
	return &BatchLDKF{b, kind}, nil
}

// SetPerFilter uploads one matrix per filter (values holds N matrices back to back, row-major).
func (kf *BatchLDKF) SetPerFilter(field C.int, values []float64, pRows int) error {
	return kbCall(func() C.int { return C.kb_set(kf.b.h, field, ptr(values), C.int64_t(kf.b.N), 0, C.int(pRows)) })
}

// Update runs LDKF.Update for every filter: measurements [N][p], controls [N][m] or nil.
func (kf *BatchLDKF) Update(measurements, controls []float64) error {
	var up *C.double
	m := 0
	if len(controls) > 0 {
		up, m = ptr(controls), len(controls)/int(kf.b.N)
	}
	return kbCall(func() C.int { return C.kb_update(kf.b.h, ptr(measurements), C.int(len(measurements)/int(kf.b.N)), up, C.int(m)) })
}

// Estimates snapshots State() and Covariance() of filters [first, first+count) and their status words (read and
// cleared): states [count][n], covariances [count][n][n].
func (kf *BatchLDKF) Estimates(first, count int64) (states, covars []float64, status []uint32, err error) {
	n := kf.b.n
	states, covars = make([]float64, int(count)*n), make([]float64, int(count)*n*n)
	status = make([]uint32, count)
	err = kbCall(func() C.int {
		return C.kbgo_get_estimate(kf.b.h, C.int64_t(first), C.int64_t(count), ptr(states), ptr(covars), nil, nil, nil, nil,
			(*C.uint32_t)(unsafe.Pointer(&status[0])), 1)
	})
	return
}

// FilterStep is kf.step of one filter of the batch (kb_filter_step): a failed Update does not advance it (vanilla.go:164-167).
func (kf *BatchLDKF) FilterStep(filter int64) (int64, error) {
	var st C.int64_t
	err := kbCall(func() C.int { return C.kb_filter_step(kf.b.h, C.int64_t(filter), &st) })
	return int64(st), err
}

// ShardedBatch is N LDKF filters split into contiguous shards over the GPUs of the node from this ONE process (kb_sharded_*,
// SURVEY section 8e): GPU g owns the filters [g N / G, (g + 1) N / G), one handle + host thread + stream per device.  Update has
// no collective; MonteCarlo / ChiSquare combine the per-shard sums with ONE ncclAllReduce over RCCL (host sum when shards share a
// device).  devices = nil uses every visible GPU.
type ShardedBatch struct {
	s       *C.kb_sharded
	n, p, m int
	N       int64
}

func NewShardedBatch(kind C.int, N int64, devices []int, x0 *mat64.Vector, P0 mat64.Symmetric, F, G, H mat64.Matrix, noise gokalman.Noise) (*ShardedBatch, error) {
	n, _ := x0.Dims()
	p, _ := H.Dims()
	_, m := G.Dims()
	sb := &ShardedBatch{n: n, p: p, m: m, N: N}
	if len(devices) == 0 {
		for g := 0; g < int(C.kb_device_count()); g++ {
			devices = append(devices, g)
		}
	}
	devs := make([]C.int, len(devices))
	for i, d := range devices {
		devs[i] = C.int(d)
	}
	if err := kbCall(func() C.int {
		return C.kb_sharded_create(&sb.s, kind, C.int(n), C.int(p), C.int(m), C.int64_t(N), C.KB_F64, &devs[0], C.int(len(devs)), 0)
	}); err != nil {
		return nil, err
	}
	runtime.SetFinalizer(sb, func(b *ShardedBatch) { C.kb_sharded_destroy(b.s) })
	for _, f := range []struct {
		f C.int
		m mat64.Matrix
		p int
	}{{C.KB_X, x0, 0}, {C.KB_P, P0, 0}, {C.KB_F, F, 0}, {C.KB_G, G, 0}, {C.KB_H, H, p},
		{C.KB_Q, noise.ProcessMatrix(), 0}, {C.KB_R, noise.MeasurementMatrix(), p}} {
		v := rowMajor(f.m)
		if len(v) == 0 {
			continue
		}
		if err := kbCall(func() C.int { return C.kb_sharded_set(sb.s, f.f, ptr(v), 1, 1, C.int(f.p), C.int64_t(len(v))) }); err != nil {
			return nil, err
		}
	}
	if _, isAWGN := noise.(*gokalman.AWGN); isAWGN {
		if err := kbCall(func() C.int { return C.kb_sharded_set_noise_kind(sb.s, C.KB_NOISE_AWGN, C.uint64_t(time.Now().UnixNano())) }); err != nil {
			return nil, err
		}
	}
	if err := kbCall(func() C.int { return C.kb_sharded_init(sb.s) }); err != nil {
		return nil, err
	}
	return sb, nil
}

// SetPerFilter uploads one matrix per filter (values: N matrices of elems doubles back to back); it is cut at the shard boundaries.
func (sb *ShardedBatch) SetPerFilter(field C.int, values []float64, pRows, elems int) error {
	return kbCall(func() C.int { return C.kb_sharded_set(sb.s, field, ptr(values), C.int64_t(sb.N), 0, C.int(pRows), C.int64_t(elems)) })
}

// Update runs LDKF.Update on every filter, the shards in parallel: measurements [N][p], controls [N][m] or nil.
func (sb *ShardedBatch) Update(measurements, controls []float64) error {
	var up *C.double
	m := 0
	if len(controls) > 0 {
		up, m = ptr(controls), len(controls)/int(sb.N)
	}
	return kbCall(func() C.int {
		return C.kb_sharded_update(sb.s, ptr(measurements), C.int(len(measurements)/int(sb.N)), up, C.int(m))
	})
}

// States / Covariances of the filters [first, first+count): [count][n] and [count][n][n].
func (sb *ShardedBatch) States(first, count int64) ([]float64, error) {
	out := make([]float64, int(count)*sb.n)
	err := kbCall(func() C.int { return C.kb_sharded_get(sb.s, C.KB_STATE, ptr(out), C.int64_t(first), C.int64_t(count), C.int64_t(sb.n)) })
	return out, err
}
func (sb *ShardedBatch) Covariances(first, count int64) ([]float64, error) {
	out := make([]float64, int(count)*sb.n*sb.n)
	err := kbCall(func() C.int {
		return C.kb_sharded_get(sb.s, C.KB_COVAR, ptr(out), C.int64_t(first), C.int64_t(count), C.int64_t(sb.n*sb.n))
	})
	return out, err
}

// MonteCarlo is NewMonteCarloRuns over the whole node (montecarlo.go:92-119): per-step mean and unbiased standard deviation over
// all N runs (mean[steps][n], stddev[steps][n]); usedRCCL tells how the shards' sums were combined.
func (sb *ShardedBatch) MonteCarlo(steps int, controls []*mat64.Vector) (mean, stddev []float64, usedRCCL bool, err error) {
	ctrl := flattenControls(controls)
	sums := make([]float64, steps*3*sb.n)
	if err = kbCall(func() C.int { return C.kb_sharded_mc_run(sb.s, C.int(steps), ptr(ctrl), C.int(len(controls)), ptr(sums), 0) }); err != nil {
		return
	}
	mean, stddev = make([]float64, steps*sb.n), make([]float64, steps*sb.n)
	err = kbCall(func() C.int { return C.kb_mc_stats(ptr(sums), C.int(steps), C.int(sb.n), C.int64_t(sb.N), ptr(mean), ptr(stddev)) })
	usedRCCL = C.kb_sharded_used_rccl(sb.s) != 0
	return
}

// ChiSquare is NewChiSquare over the whole node (chisquare.go:16-95); sb is the truth (pure predictor, AWGN), kf the filter under test.
func (sb *ShardedBatch) ChiSquare(kf *ShardedBatch, steps int, controls []*mat64.Vector, replayLastMC, withNEES, withNIS bool) ([]float64, []float64, error) {
	ctrl := flattenControls(controls)
	sums := make([]float64, steps*2)
	b2i := func(v bool) C.int {
		if v {
			return 1
		}
		return 0
	}
	if err := kbCall(func() C.int {
		return C.kb_sharded_chisquare(sb.s, kf.s, C.int(steps), ptr(ctrl), C.int(len(controls)), b2i(replayLastMC), b2i(withNEES), b2i(withNIS), ptr(sums))
	}); err != nil {
		return nil, nil, err
	}
	nis, nees := make([]float64, steps), make([]float64, steps)
	for k := 0; k < steps; k++ {
		nis[k], nees[k] = sums[2*k]/float64(sb.N), sums[2*k+1]/float64(sb.N)
	}
	return nis, nees, nil
}

// VanLoan computes F and Q from the continuous-time system A, Γ, W and the sampling period Δt
// (gokalman.VanLoan, c2d.go:13-75) on the GPU.
func VanLoan(A, Γ, W *mat64.Dense, Δt float64) (*mat64.Dense, *mat64.SymDense, error) {
	n, _ := A.Dims()
	_, q := Γ.Dims()
	a, g, w := rowMajor(A), rowMajor(Γ), rowMajor(W)
	f, qq := make([]float64, n*n), make([]float64, n*n)
	var st C.uint32_t
	dt := C.double(Δt)
	if err := kbCall(func() C.int {
		return C.kb_van_loan(0, C.KB_F64, C.int(n), C.int(q), 1, ptr(a), ptr(g), ptr(w), &dt, 15, ptr(f), ptr(qq), &st)
	}); err != nil {
		return nil, nil, err
	}
	var err error
	if st&C.KB_ST_NYQUIST != 0 {
		err = fmt.Errorf("gokalman: Nyquist sampling criterion not fulfilled with Δt=%f", Δt)
	}
	var Q *mat64.SymDense
	if st&C.KB_ST_ASYMMETRIC == 0 { // QSym, _ := AsSymDense(&Q): nil when asymmetric (c2d.go:73)
		Q = mat64.NewSymDense(n, qq)
	}
	return mat64.NewDense(n, n, f), Q, err
}
