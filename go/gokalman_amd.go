// Package gokalman_amd is the cgo shim a gokalman maintainer adds to route the predict/update
// hot path to the MI355X engine.  It binds exactly the C ABI in include/gokalman_amd.h and
// implements gokalman's own interfaces (kalman.go:35-72), so `examples/*/main.go` and the tests
// keep calling kf.Update(measurement, control).
//
// NOT COMPILED in this repository's CI: the build image has no Go toolchain and gonum is not
// vendored by the reference.  Build (on a box with Go >= 1.7, gonum and ROCm):
//   CGO_CFLAGS="-I${REPO}/include" CGO_LDFLAGS="-L${REPO}/gokalman_amd -lgokalman_amd" go build
package gokalman_amd

/*
#cgo LDFLAGS: -lgokalman_amd
#include <stdlib.h>
#include "gokalman_amd.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"math"
	"runtime"
	"time"
	"unsafe"

	"github.com/ChristopherRabotin/gokalman"
	"github.com/gonum/matrix/mat64"
)

func kbErr(rc C.int) error {
	if rc == C.KB_OK {
		return nil
	}
	return errors.New(C.GoString(C.kb_last_error()))
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
	if err := kbErr(C.kb_create(&b.h, kind, C.int(n), C.int(p), C.int(m), C.int64_t(N), C.KB_F64, 0, flags)); err != nil {
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
	return kbErr(C.kb_set(b.h, field, ptr(v), 1, 1, C.int(pRows)))
}

func (b *batch) get(field C.int, rows, cols int) []float64 {
	out := make([]float64, rows*cols)
	if err := kbErr(C.kb_get(b.h, field, ptr(out), 0, 1)); err != nil {
		panic(err)
	}
	return out
}

// Estimate implements gokalman.Estimate (kalman.go:64-72).  It is an immutable VALUE, as in the reference, whose Update
// returns a freshly allocated estimate that callers keep (vanilla.go:216-218; examples/jerkcar/main.go:71-90 sends it
// through a channel to another goroutine, montecarlo.go:108-117 stores one per step): every member is copied out of HBM
// once, by ONE kb_get_estimate call, when the estimate is created.
type Estimate struct {
	n, p                        int
	kind                        C.int
	state, meas, innov          []float64
	covar, predCovar, gain      []float64
	status                      uint32
}

// snapshot downloads the current estimate of filter 0 of b and reads-and-clears its status word, so that one failed
// Update does not poison the next call (vanilla.go:164-167 returns an error and leaves prevEst alone).
func snapshot(b *batch, kind C.int) (*Estimate, error) {
	n, p := b.n, int(C.kb_meas_dim(b.h))
	info := kind == C.KB_INFORMATION || kind == C.KB_SRIF
	lazy := info || kind == C.KB_SQUAREROOT
	ni := p
	if info {
		ni = n // Innovation() returns the information vector (information.go:272, srif.go:237)
	}
	e := &Estimate{n: n, p: p, kind: kind, state: make([]float64, n), covar: make([]float64, n*n), predCovar: make([]float64, n*n),
		meas: make([]float64, p), innov: make([]float64, ni)}
	var v C.kb_estimate_view
	v.state, v.covariance, v.pred_covariance = ptr(e.state), ptr(e.covar), ptr(e.predCovar)
	v.measurement, v.innovation = ptr(e.meas), ptr(e.innov)
	if !lazy || kind == C.KB_SQUAREROOT {
		e.gain = make([]float64, n*p)
		v.gain = ptr(e.gain)
	}
	var st C.uint32_t
	v.status, v.clear_status = &st, 1
	if err := kbErr(C.kb_get_estimate(b.h, 0, 1, &v)); err != nil {
		return nil, err
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

// IsWithinNσ: vanilla.go:231-239.
func (e *Estimate) IsWithinNσ(N float64) bool {
	for i := 0; i < e.n; i++ {
		nσ := N * math.Sqrt(e.covar[i*e.n+i])
		if e.state[i] > nσ || e.state[i] < -nσ {
			return false
		}
	}
	return true
}
func (e *Estimate) IsWithin2σ() bool { return e.IsWithinNσ(2) }
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

// stepError turns the status word of the step that just ran into the reference's error value.
func stepError(st uint32, what string, step int64) error {
	if st&C.KB_ST_SINGULAR != 0 {
		return fmt.Errorf("could not invert %s at k=%d: matrix singular or near-singular", what, step)
	}
	if st&(C.KB_ST_ASYMMETRIC|C.KB_ST_NONFINITE) != 0 {
		return errors.New("matrix is not symmetric") // helper.go:76
	}
	return nil
}

// Vanilla implements gokalman.LDKF (kalman.go:35-47) with the device engine behind it.
type Vanilla struct {
	b       *batch
	kind    C.int
	F, G, H mat64.Matrix
	Noise   gokalman.Noise
}

// NewVanilla mirrors gokalman.NewVanilla (vanilla.go:21-40).
func NewVanilla(x0 *mat64.Vector, Covar0 mat64.Symmetric, F, G, H mat64.Matrix, noise gokalman.Noise) (*Vanilla, *Estimate, error) {
	return newLDKF(C.KB_VANILLA, 0, x0, Covar0, F, G, H, noise)
}

// Update implements LDKF.Update (vanilla.go:128-220): one launch of the HIP step kernel, then one snapshot.
func (kf *Vanilla) Update(measurement, control *mat64.Vector) (gokalman.Estimate, error) {
	y := rowMajor(measurement)
	var up *C.double
	nu := 0
	if control != nil {
		if u := rowMajor(control); len(u) > 0 {
			up, nu = ptr(u), len(u)
		}
	}
	if err := kbErr(C.kb_update(kf.b.h, ptr(y), C.int(len(y)), up, C.int(nu))); err != nil {
		return nil, err // "dimensions must agree: ..." (vanilla.go:129-135)
	}
	est, err := snapshot(kf.b, kf.kind)
	if err != nil {
		return nil, err
	}
	if err := stepError(est.status, "`H*P_kp1_minus*H' + R`", int64(C.kb_step(kf.b.h))-1); err != nil {
		return nil, err // the filter kept its previous estimate; the next Update runs normally
	}
	return est, nil
}
func (kf *Vanilla) GetNoise() gokalman.Noise            { return kf.Noise }
func (kf *Vanilla) GetStateTransition() mat64.Matrix    { return kf.F }
func (kf *Vanilla) GetInputControl() mat64.Matrix       { return kf.G }
func (kf *Vanilla) GetMeasurementMatrix() mat64.Matrix  { return kf.H }
func (kf *Vanilla) SetStateTransition(F mat64.Matrix)   { kf.F = F; kf.b.set(C.KB_F, F, 0) }
func (kf *Vanilla) SetInputControl(G mat64.Matrix)      { kf.G = G; kf.b.set(C.KB_G, G, 0) }
func (kf *Vanilla) SetMeasurementMatrix(H mat64.Matrix) { kf.H = H; p, _ := H.Dims(); kf.b.set(C.KB_H, H, p) }
func (kf *Vanilla) SetNoise(n gokalman.Noise) {
	kf.Noise = n
	p, _ := n.MeasurementMatrix().Dims()
	kf.b.set(C.KB_Q, n.ProcessMatrix(), 0)
	kf.b.set(C.KB_R, n.MeasurementMatrix(), p)
}
func (kf *Vanilla) Reset()         { C.kb_reset(kf.b.h) }
// String is vanilla.go:76-78 / squareroot.go:65-67 (information.go:96-98 prints inv(F) instead of F).
func (kf *Vanilla) String() string {
	if kf.kind == C.KB_INFORMATION {
		var finv mat64.Dense
		if err := finv.Inverse(kf.F); err == nil {
			return fmt.Sprintf("inv(F)=%v\nG=%v\nH=%v\n%s", mat64.Formatted(&finv, mat64.Prefix("      ")), mat64.Formatted(kf.G, mat64.Prefix("  ")), mat64.Formatted(kf.H, mat64.Prefix("  ")), kf.Noise)
		}
	}
	return fmt.Sprintf("F=%v\nG=%v\nH=%v\n%s", mat64.Formatted(kf.F, mat64.Prefix("  ")), mat64.Formatted(kf.G, mat64.Prefix("  ")), mat64.Formatted(kf.H, mat64.Prefix("  ")), kf.Noise)
}

var _ gokalman.LDKF = (*Vanilla)(nil)
var _ gokalman.Estimate = (*Estimate)(nil)

// newLDKF is the shared constructor body of the LDKF kinds (vanilla.go:21, squareroot.go:21,
// information.go:20/65): kind selects the device kernels, flags carries INFO_FROM_STATE.
func newLDKF(kind C.int, flags C.uint, x0 *mat64.Vector, P0 mat64.Symmetric, F, G, H mat64.Matrix, noise gokalman.Noise) (*Vanilla, *Estimate, error) {
	n, _ := x0.Dims()
	p, _ := H.Dims()
	_, m := G.Dims()
	b, err := newBatch(kind, n, p, m, 1, C.KB_FLAG_FULL_ESTIMATE|flags)
	if err != nil {
		return nil, nil, err
	}
	for _, s := range []struct {
		f C.int
		m mat64.Matrix
		p int
	}{{C.KB_X, x0, 0}, {C.KB_P, P0, 0}, {C.KB_F, F, 0}, {C.KB_G, G, 0}, {C.KB_H, H, p},
		{C.KB_Q, noise.ProcessMatrix(), 0}, {C.KB_R, noise.MeasurementMatrix(), p}} {
		if err := b.set(s.f, s.m, s.p); err != nil {
			return nil, nil, err
		}
	}
	if _, isAWGN := noise.(*gokalman.AWGN); isAWGN {
		if err := kbErr(C.kb_set_noise_kind(b.h, C.KB_NOISE_AWGN, C.uint64_t(time.Now().UnixNano()))); err != nil {
			return nil, nil, err // "process noise invalid" (noise.go:148-156 panics there)
		}
	}
	if err := kbErr(C.kb_init(b.h)); err != nil {
		return nil, nil, err
	}
	est0, err := snapshot(b, kind)
	if err != nil {
		return nil, nil, err
	}
	return &Vanilla{b, kind, F, G, H, noise}, est0, nil
}

// The remaining LDKF constructors differ only in the kind handed to kb_create; the returned value
// satisfies gokalman.LDKF through the methods defined on *Vanilla above.
func NewPurePredictorVanilla(x0 *mat64.Vector, P0 mat64.Symmetric, F, G, H mat64.Matrix, n gokalman.Noise) (*Vanilla, *Estimate, error) {
	return newLDKF(C.KB_VANILLA_PREDICT, 0, x0, P0, F, G, H, n) // vanilla.go:43-62
}
func NewSquareRoot(x0 *mat64.Vector, P0 mat64.Symmetric, F, G, H mat64.Matrix, n gokalman.Noise) (*Vanilla, *Estimate, error) {
	return newLDKF(C.KB_SQUAREROOT, 0, x0, P0, F, G, H, n) // squareroot.go:21-50
}
func NewInformation(i0 *mat64.Vector, I0 mat64.Symmetric, F, G, H mat64.Matrix, n gokalman.Noise) (*Vanilla, *Estimate, error) {
	return newLDKF(C.KB_INFORMATION, 0, i0, I0, F, G, H, n) // information.go:20-53
}
func NewInformationFromState(x0 *mat64.Vector, P0 mat64.Symmetric, F, G, H mat64.Matrix, n gokalman.Noise) (*Vanilla, *Estimate, error) {
	return newLDKF(C.KB_INFORMATION, C.KB_FLAG_INFO_FROM_STATE, x0, P0, F, G, H, n) // information.go:65-81
}

// NLDKF implements gokalman.NLDKF (kalman.go:51-60) for SRIF and HybridKF batches.
type NLDKF struct {
	b     *batch
	kind  C.int
	Noise gokalman.Noise
}

func newNLDKF(kind C.int, x0 *mat64.Vector, P0 mat64.Symmetric, noise gokalman.Noise, measSize int, flags C.uint) (*NLDKF, *Estimate, error) {
	n, _ := x0.Dims()
	q := 0
	if kind == C.KB_HYBRID {
		q, _ = noise.ProcessMatrix().Dims()
	}
	b, err := newBatch(kind, n, measSize, q, 1, C.KB_FLAG_FULL_ESTIMATE|flags)
	if err != nil {
		return nil, nil, err
	}
	if err := b.set(C.KB_X, x0, 0); err != nil {
		return nil, nil, err
	}
	if err := b.set(C.KB_P, P0, 0); err != nil {
		return nil, nil, err
	}
	if err := b.set(C.KB_R, noise.MeasurementMatrix(), measSize); err != nil {
		return nil, nil, err
	}
	if q > 0 {
		if err := b.set(C.KB_Q, noise.ProcessMatrix(), 0); err != nil {
			return nil, nil, err
		}
	}
	if err := kbErr(C.kb_init(b.h)); err != nil {
		return nil, nil, err
	}
	est0, err := snapshot(b, kind)
	if err != nil {
		return nil, nil, err
	}
	return &NLDKF{b: b, kind: kind, Noise: noise}, est0, nil
}

// NewSRIF mirrors gokalman.NewSRIF (srif.go:14-49); NewHybridKF mirrors hybrid.go:23-34.
func NewSRIF(x0 *mat64.Vector, P0 mat64.Symmetric, measSize int, nonTriR bool, n gokalman.Noise) (*NLDKF, *Estimate, error) {
	var fl C.uint
	if nonTriR {
		fl = C.KB_FLAG_SRIF_NON_TRI_R
	}
	p, _ := n.MeasurementMatrix().Dims()
	_ = measSize // only sizes Predict()'s zero vectors in the reference
	return newNLDKF(C.KB_SRIF, x0, P0, n, p, fl)
}
func NewHybridKF(x0 *mat64.Vector, P0 mat64.Symmetric, n gokalman.Noise, measSize int) (*NLDKF, *Estimate, error) {
	return newNLDKF(C.KB_HYBRID, x0, P0, n, measSize, 0)
}

func (kf *NLDKF) Prepare(Φ, Htilde *mat64.Dense) { // srif.go:82-86, hybrid.go:78-82
	phi, h := rowMajor(Φ), rowMajor(Htilde)
	if err := kbErr(C.kb_prepare(kf.b.h, ptr(phi), ptr(h), 1, 1)); err != nil {
		panic(err)
	}
}
func (kf *NLDKF) PreparePNT(Γ *mat64.Dense) { // hybrid.go:86-89
	g := rowMajor(Γ)
	if err := kbErr(C.kb_prepare_pnt(kf.b.h, ptr(g), 1, 1)); err != nil {
		panic(err)
	}
}
func (kf *NLDKF) whatFailed() string {
	if kf.kind == C.KB_SRIF {
		return "`Φ`" // srif.go:113
	}
	return "`H*P_kp1_minus*H' + R`" // hybrid.go:151
}
func (kf *NLDKF) stepEstimate() (gokalman.Estimate, error) {
	est, err := snapshot(kf.b, kf.kind)
	if err != nil {
		return nil, err
	}
	if err := stepError(est.status, kf.whatFailed(), int64(C.kb_step(kf.b.h))-1); err != nil {
		return nil, err
	}
	return est, nil
}
func (kf *NLDKF) Update(realObservation, computedObservation *mat64.Vector) (gokalman.Estimate, error) { // srif.go:90, hybrid.go:93
	r, c := rowMajor(realObservation), rowMajor(computedObservation)
	if err := kbErr(C.kb_update_nl(kf.b.h, ptr(r), C.int(len(r)), ptr(c), C.int(len(c)))); err != nil {
		return nil, err // "kf is locked (call Prepare() first)", "dimensions must agree: ..."
	}
	return kf.stepEstimate()
}
func (kf *NLDKF) Predict() (gokalman.Estimate, error) { // srif.go:96, hybrid.go:99
	if err := kbErr(C.kb_predict_nl(kf.b.h)); err != nil {
		return nil, err
	}
	return kf.stepEstimate()
}
func (kf *NLDKF) EKFEnabled() bool { return C.kb_ekf_enabled(kf.b.h) != 0 }
func (kf *NLDKF) EnableEKF()       { C.kb_set_ekf(kf.b.h, 1) }
func (kf *NLDKF) DisableEKF()      { C.kb_set_ekf(kf.b.h, 0) }
// String is hybrid.go:63-65 (the reference's SRIF has no String of its own).
func (kf *NLDKF) String() string {
	if kf.kind == C.KB_HYBRID {
		return fmt.Sprintf("HybridKF [k=%d]\n%s", int64(C.kb_step(kf.b.h)), kf.Noise)
	}
	return fmt.Sprintf("SRIF [k=%d]", int64(C.kb_step(kf.b.h)))
}
func (kf *NLDKF) SetNoise(n gokalman.Noise) {
	kf.Noise = n
	p, _ := n.MeasurementMatrix().Dims()
	kf.b.set(C.KB_R, n.MeasurementMatrix(), p)
}

var _ gokalman.NLDKF = (*NLDKF)(nil)

// MonteCarloRuns mirrors gokalman.MonteCarloRuns (montecarlo.go:12-59): per-step mean and unbiased standard
// deviation of the state over the runs.  The reference stores samples x steps Estimate objects and reduces them on
// demand; the engine reduces on the device (kb_mc_run) and only steps x 2n sums ever reach the host.
type MonteCarloRuns struct {
	runs          int64
	steps, n      int
	mean, stddev  []float64 // [steps][n]
}

func (mc MonteCarloRuns) Mean(step int) []float64   { return mc.mean[step*mc.n : (step+1)*mc.n] }   // montecarlo.go:18-37
func (mc MonteCarloRuns) StdDev(step int) []float64 { return mc.stddev[step*mc.n : (step+1)*mc.n] } // montecarlo.go:40-59

func flattenControls(controls []*mat64.Vector) []float64 {
	var out []float64
	for _, c := range controls {
		out = append(out, rowMajor(c)...)
	}
	return out
}

// NewMonteCarloRuns mirrors gokalman.NewMonteCarloRuns(samples, steps, rowsH, controls, kf) (montecarlo.go:92-119).
// `kf` is the pure-predictor template (NewPurePredictorVanilla with AWGN noise); the runs are a batch of `samples`
// filters created from the template's model.  firstRun = global index of this process's first run when the ensemble is
// sharded over GPUs (a run's noise depends only on its global index); the per-shard sums are then added by the caller.
func NewMonteCarloRuns(samples, steps, rowsH int, controls []*mat64.Vector, kf *Vanilla, x0 *mat64.Vector, P0 mat64.Symmetric) (MonteCarloRuns, error) {
	if kf.kind != C.KB_VANILLA_PREDICT {
		panic("the Kalman filter needed for the Monte Carlo runs must be a pure predictor") // montecarlo.go:93-95
	}
	if len(controls) != 1 && len(controls) != steps {
		panic("must provide as much control vectors as steps, or just one control vector") // montecarlo.go:105-107
	}
	runs, err := NewBatchLDKF(C.KB_VANILLA_PREDICT, int64(samples), x0, P0, kf.F, kf.G, kf.H, kf.Noise)
	if err != nil {
		return MonteCarloRuns{}, err
	}
	n := runs.b.n
	ctrl := flattenControls(controls)
	sums := make([]float64, steps*3*n)
	if err := kbErr(C.kb_mc_run(runs.b.h, C.int(steps), ptr(ctrl), C.int(len(controls)), 0, ptr(sums))); err != nil {
		return MonteCarloRuns{}, err
	}
	mc := MonteCarloRuns{int64(samples), steps, n, make([]float64, steps*n), make([]float64, steps*n)}
	if err := kbErr(C.kb_mc_stats(ptr(sums), C.int(steps), C.int(n), C.int64_t(samples), ptr(mc.mean), ptr(mc.stddev))); err != nil {
		return MonteCarloRuns{}, err
	}
	_ = rowsH // only sizes the zero measurement vector in the reference (montecarlo.go:111)
	return mc, nil
}

// NewChiSquare mirrors gokalman.NewChiSquare(kf, runs, controls, withNEES, withNIS) (chisquare.go:16-95): returns
// (NISmeans, NEESmeans).  `truth` is the batch that generates the Monte-Carlo runs (one pure-predictor AWGN filter per
// run), `kf` a Vanilla batch of the same size holding the filter under test.
func NewChiSquare(kf, truth *BatchLDKF, steps int, controls []*mat64.Vector, withNEES, withNIS bool) ([]float64, []float64, error) {
	if !withNEES && !withNIS {
		return nil, nil, errors.New("Chi Square requires either NEES or NIS or both") // chisquare.go:17-19
	}
	ctrl := flattenControls(controls)
	sums := make([]float64, steps*2)
	b2i := func(v bool) C.int {
		if v {
			return 1
		}
		return 0
	}
	if err := kbErr(C.kb_chisquare(truth.b.h, kf.b.h, C.int(steps), ptr(ctrl), C.int(len(controls)), 0, 1, b2i(withNEES), b2i(withNIS), ptr(sums))); err != nil {
		return nil, nil, err
	}
	nis, nees := make([]float64, steps), make([]float64, steps)
	for k := 0; k < steps; k++ {
		nis[k], nees[k] = sums[2*k]/float64(truth.b.N), sums[2*k+1]/float64(truth.b.N)
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
		if err := kbErr(C.kb_set_noise_kind(b.h, C.KB_NOISE_AWGN, C.uint64_t(time.Now().UnixNano()))); err != nil {
			return nil, err
		}
	}
	if err := kbErr(C.kb_init(b.h)); err != nil {
		return nil, err
	}
	return &BatchLDKF{b, kind}, nil
}

// SetPerFilter uploads one matrix per filter (values holds N matrices back to back, row-major).
func (kf *BatchLDKF) SetPerFilter(field C.int, values []float64, pRows int) error {
	return kbErr(C.kb_set(kf.b.h, field, ptr(values), C.int64_t(kf.b.N), 0, C.int(pRows)))
}

// Update runs LDKF.Update for every filter: measurements [N][p], controls [N][m] or nil.
func (kf *BatchLDKF) Update(measurements, controls []float64) error {
	var up *C.double
	m := 0
	if len(controls) > 0 {
		up, m = ptr(controls), len(controls)/int(kf.b.N)
	}
	return kbErr(C.kb_update(kf.b.h, ptr(measurements), C.int(len(measurements)/int(kf.b.N)), up, C.int(m)))
}

// Estimates snapshots State() and Covariance() of filters [first, first+count) and their status words (read and
// cleared): states [count][n], covariances [count][n][n].
func (kf *BatchLDKF) Estimates(first, count int64) (states, covars []float64, status []uint32, err error) {
	n := kf.b.n
	states, covars = make([]float64, int(count)*n), make([]float64, int(count)*n*n)
	status = make([]uint32, count)
	var v C.kb_estimate_view
	v.state, v.covariance = ptr(states), ptr(covars)
	v.status, v.clear_status = (*C.uint32_t)(unsafe.Pointer(&status[0])), 1
	err = kbErr(C.kb_get_estimate(kf.b.h, C.int64_t(first), C.int64_t(count), &v))
	return
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
	if rc := C.kb_van_loan(0, C.KB_F64, C.int(n), C.int(q), 1, ptr(a), ptr(g), ptr(w), &dt, 15, ptr(f), ptr(qq), &st); rc != C.KB_OK {
		return nil, nil, errors.New(C.GoString(C.kb_last_error()))
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
