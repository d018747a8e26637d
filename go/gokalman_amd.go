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

func ptr(v []float64) *C.double { return (*C.double)(unsafe.Pointer(&v[0])) }

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
	return kbErr(C.kb_set(b.h, field, ptr(v), 1, 1, C.int(pRows)))
}

func (b *batch) get(field C.int, rows, cols int) []float64 {
	out := make([]float64, rows*cols)
	if err := kbErr(C.kb_get(b.h, field, ptr(out), 0, 1)); err != nil {
		panic(err)
	}
	return out
}

// Estimate implements gokalman.Estimate (kalman.go:64-72) on top of kb_get.
type Estimate struct{ b *batch }

func (e Estimate) State() *mat64.Vector       { return mat64.NewVector(e.b.n, e.b.get(C.KB_STATE, e.b.n, 1)) }
func (e Estimate) Measurement() *mat64.Vector { p := int(C.kb_meas_dim(e.b.h)); return mat64.NewVector(p, e.b.get(C.KB_MEASUREMENT, p, 1)) }
func (e Estimate) Innovation() *mat64.Vector  { p := int(C.kb_meas_dim(e.b.h)); return mat64.NewVector(p, e.b.get(C.KB_INNOVATION, p, 1)) }
func (e Estimate) Covariance() mat64.Symmetric {
	return mat64.NewSymDense(e.b.n, e.b.get(C.KB_COVAR, e.b.n, e.b.n))
}
func (e Estimate) PredCovariance() mat64.Symmetric {
	return mat64.NewSymDense(e.b.n, e.b.get(C.KB_PRED_COVAR, e.b.n, e.b.n))
}
func (e Estimate) IsWithinNσ(N float64) bool {
	var out C.uint8_t
	if err := kbErr(C.kb_is_within_nsigma(e.b.h, C.double(N), &out, 0, 1)); err != nil {
		panic(err)
	}
	return out != 0
}
func (e Estimate) String() string { return "gokalman_amd.Estimate" }

// Vanilla implements gokalman.LDKF (kalman.go:35-47) with the device engine behind it.
type Vanilla struct {
	b       *batch
	F, G, H mat64.Matrix
	Noise   gokalman.Noise
}

// NewVanilla mirrors gokalman.NewVanilla (vanilla.go:21-40).
func NewVanilla(x0 *mat64.Vector, Covar0 mat64.Symmetric, F, G, H mat64.Matrix, noise gokalman.Noise) (*Vanilla, *Estimate, error) {
	n, _ := x0.Dims()
	p, _ := H.Dims()
	_, m := G.Dims()
	b, err := newBatch(C.KB_VANILLA, n, p, m, 1, C.KB_FLAG_FULL_ESTIMATE)
	if err != nil {
		return nil, nil, err
	}
	for _, s := range []struct {
		f C.int
		m mat64.Matrix
		p int
	}{{C.KB_X, x0, 0}, {C.KB_P, Covar0, 0}, {C.KB_F, F, 0}, {C.KB_G, G, 0}, {C.KB_H, H, p},
		{C.KB_Q, noise.ProcessMatrix(), 0}, {C.KB_R, noise.MeasurementMatrix(), p}} {
		if err := b.set(s.f, s.m, s.p); err != nil {
			return nil, nil, err
		}
	}
	if err := kbErr(C.kb_init(b.h)); err != nil {
		return nil, nil, err
	}
	return &Vanilla{b, F, G, H, noise}, &Estimate{b}, nil
}

// Update implements LDKF.Update (vanilla.go:128-220): one launch of the HIP step kernel.
func (kf *Vanilla) Update(measurement, control *mat64.Vector) (gokalman.Estimate, error) {
	y := rowMajor(measurement)
	u := rowMajor(control)
	if err := kbErr(C.kb_update(kf.b.h, ptr(y), C.int(len(y)), ptr(u), C.int(len(u)))); err != nil {
		return nil, err
	}
	var st C.uint32_t
	C.kb_get_status(kf.b.h, &st, 0, 1)
	if st&C.KB_ST_SINGULAR != 0 {
		return nil, errors.New("could not invert `H*P_kp1_minus*H' + R`")
	}
	if st&(C.KB_ST_ASYMMETRIC|C.KB_ST_NONFINITE) != 0 {
		return nil, errors.New("matrix is not symmetric")
	}
	return Estimate{kf.b}, nil
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
func (kf *Vanilla) String() string { return "gokalman_amd.Vanilla" }

var _ gokalman.LDKF = (*Vanilla)(nil)
var _ gokalman.Estimate = Estimate{}
var _ = math.Sqrt

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
	return &Vanilla{b, F, G, H, noise}, &Estimate{b}, nil
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
type NLDKF struct{ b *batch }

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
	return &NLDKF{b}, &Estimate{b}, nil
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
func (kf *NLDKF) Update(realObservation, computedObservation *mat64.Vector) (gokalman.Estimate, error) { // srif.go:90, hybrid.go:93
	r, c := rowMajor(realObservation), rowMajor(computedObservation)
	if err := kbErr(C.kb_update_nl(kf.b.h, ptr(r), C.int(len(r)), ptr(c), C.int(len(c)))); err != nil {
		return nil, err // "kf is locked (call Prepare() first)", "dimensions must agree: ..."
	}
	var st C.uint32_t
	C.kb_get_status(kf.b.h, &st, 0, 1)
	if st&C.KB_ST_SINGULAR != 0 {
		return nil, errors.New("could not invert `H*P_kp1_minus*H' + R`")
	}
	return Estimate{kf.b}, nil
}
func (kf *NLDKF) Predict() (gokalman.Estimate, error) { // srif.go:96, hybrid.go:99
	if err := kbErr(C.kb_predict_nl(kf.b.h)); err != nil {
		return nil, err
	}
	return Estimate{kf.b}, nil
}
func (kf *NLDKF) EKFEnabled() bool { return C.kb_ekf_enabled(kf.b.h) != 0 }
func (kf *NLDKF) EnableEKF()       { C.kb_set_ekf(kf.b.h, 1) }
func (kf *NLDKF) DisableEKF()      { C.kb_set_ekf(kf.b.h, 0) }
func (kf *NLDKF) SetNoise(n gokalman.Noise) {
	p, _ := n.MeasurementMatrix().Dims()
	kf.b.set(C.KB_R, n.MeasurementMatrix(), p)
}

var _ gokalman.NLDKF = (*NLDKF)(nil)

// MonteCarloRuns / NewMonteCarloRuns (montecarlo.go:12-59, 92-119) and NewChiSquare (chisquare.go:16-95)
// bind kb_mc_run + kb_mc_stats and kb_chisquare the same way; the truth filter is a batch created with
// nfilters = samples.

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
