// kb_srif_odd.hip -- SRIF (srif.go:101-160, :298-340) with an ODD number of states (7, 9, 11) or fewer than six.  The two-lanes-per-filter
// kernel (kb_srif_pair.h) splits rows and columns by parity and is built for n = 6, 8, 10, 12; any other filter up to 11 states runs on the
// next instantiation as the filter diag(that filter, uncoupled states) -- with ONE extra state: R' = diag(R, 1), b' = (b, 0), Phi' = diag(Phi, 1), Htilde' = (Htilde, 0).  Phi'^-1 =
// diag(Phi^-1, 1) (the extra pivot is 1), every Householder step k < n sees a zero in the extra row and column (its norm and its
// updates add exact zeros), step n reflects the extra row onto itself -- so the leading n x n block of the result IS the n-state
// filter's, to rounding order.  Per step: the state (unless the shadow still holds it from the step before) and the model (the caller's
// planar Phi / Htilde of kb_prepare_dev are read in place) are copied into the widened shadow blocks, the even kernel steps them, one
// launch copies the state (and the Estimate extras) back: ~2x the bytes of the even kernel, against the statement kernel's scratch arrays two orders of magnitude above it.
#include "kb_internal.h"

namespace kb {

namespace {

struct PadField {
    int src_off, dst_off, rows_s, cols_s, rows_d, cols_d, unit;   // unit: the part of the diagonal the source does not have is 1
    const void *planar;                                          // the source is a caller's planar array (element e of filter i at [e ld + i])
};
struct PadArgs {
    const void *src; void *dst;
    int src_elems, dst_elems, nf;
    int64_t ld, N;
    PadField f[5];
};

// grid (tiles, fields), 256 threads: wave w copies elements w, w + 4, ... of the field, lane f the tile's filter f (512-byte rows)
template <typename T>
__global__ void __launch_bounds__(256) srif_pad_kernel(const PadArgs pa) {
    const int64_t tile = blockIdx.x;
    const PadField fd = pa.f[blockIdx.y];
    const int f = threadIdx.x & 63, w = threadIdx.x >> 6;
    const T *src = (const T *)pa.src + tile * ((int64_t)KB_TILE * pa.src_elems) + f;
    T *dst = (T *)pa.dst + tile * ((int64_t)KB_TILE * pa.dst_elems) + f;
    const T *pl = (const T *)fd.planar;
    const int64_t fi = tile * KB_TILE + f;
    const int total = fd.rows_d * fd.cols_d;
    for (int e = w; e < total; e += 4) {
        const int i = e / fd.cols_d, j = e - i * fd.cols_d;
        T v = (fd.unit && i == j) ? T(1) : T(0);
        if (i < fd.rows_s && j < fd.cols_s) {
            const int es = i * fd.cols_s + j;
            if (pl) v = fi < pa.N ? __builtin_nontemporal_load(pl + ((int64_t)es * pa.ld + fi)) : T(0);
            else v = src[(int64_t)(fd.src_off + es) * KB_TILE];
        }
        dst[(int64_t)(fd.dst_off + e) * KB_TILE] = v;
    }
}

void pad_launch(const Batch &b, const PadArgs &pa) {
    const dim3 grid((unsigned)b.ntiles, (unsigned)pa.nf), block(256);
    if (b.dtype == KB_F64) hipLaunchKernelGGL(srif_pad_kernel<double>, grid, block, 0, b.stream, pa);
    else hipLaunchKernelGGL(srif_pad_kernel<float>, grid, block, 0, b.stream, pa);
}

}  // namespace

// the instantiated size an n-state filter runs on: 6 below six states, the next even number for 7, 9, ... 15; 0: none (n is instantiated itself)
static int srif_widened(int n) { return n < 6 ? 6 : ((n & 1) && n < 16 ? n + 1 : 0); }

bool srif_odd_ok(const Batch &b, const StepArgs &a) {
    if (!srif_widened(a.n) || (a.flags & KB_FLAG_STATEMENT_KERNELS)) return false;
    StepArgs a2 = a;
    a2.n = srif_widened(a.n); a2.ext_phi = a2.ext_h = nullptr; a2.ext_ld = 0;
    return srif_reg_ok(b, a2);
}

int launch_srif_odd(const Batch &b, const StepArgs &a) {
    const int n = a.n, n2 = srif_widened(n), pm = a.pmax;
    const Layout L2 = make_layout(KB_SRIF, n2, pm, a.m, a.flags);
    const bool full = (a.flags & KB_FLAG_FULL_ESTIMATE) != 0;
    if (!b.d_sh_state) KB_HIP(dev_alloc(&b.d_sh_state, b.block_bytes(L2.st_elems)));
    if (!b.d_sh_model) KB_HIP(dev_alloc(&b.d_sh_model, b.block_bytes(L2.mo_elems)));
    if (full && !b.d_sh_est) {
        // a failed filter's Estimate slots are not written by the even kernel and ARE copied back below: defined values (zeros,
        // what a fresh batch's estimate block holds), never uninitialised memory
        KB_HIP(dev_alloc(&b.d_sh_est, b.block_bytes(L2.es_elems)));
        KB_HIP(hipMemsetAsync(b.d_sh_est, 0, b.block_bytes(L2.es_elems), b.stream));
    }
    PadArgs st{};   // state: b, R
    st.src = a.state; st.dst = b.d_sh_state; st.src_elems = a.L.st_elems; st.dst_elems = L2.st_elems; st.ld = 0; st.N = a.N; st.nf = 2;
    st.f[0] = PadField{a.L.st_vec, L2.st_vec, 1, n, 1, n2, 0, nullptr};
    st.f[1] = PadField{a.L.st_mat, L2.st_mat, n, n, n2, n2, 1, nullptr};
    if (!b.sh_state_current) pad_launch(b, st);   // (consecutive steps: the shadow still holds what the last step copied back)
    b.sh_state_current = false;
    PadArgs mo{};   // model: Phi, Htilde (from the model block or the caller's planar arrays), chol(R)
    mo.src = a.model; mo.dst = b.d_sh_model; mo.src_elems = a.L.mo_elems; mo.dst_elems = L2.mo_elems; mo.ld = a.ext_ld; mo.N = a.N; mo.nf = 3;
    mo.f[0] = PadField{a.L.mo_F, L2.mo_F, n, n, n2, n2, 1, a.ext_phi};
    mo.f[1] = PadField{a.L.mo_H, L2.mo_H, a.ext_h ? a.p : pm, n, pm, n2, 0, a.ext_h};
    mo.f[2] = PadField{a.L.mo_LR, L2.mo_LR, 1, pm * (pm + 1) / 2, 1, pm * (pm + 1) / 2, 0, nullptr};
    pad_launch(b, mo);
    StepArgs a2 = a;
    a2.n = n2; a2.L = L2; a2.state = b.d_sh_state; a2.model = b.d_sh_model; a2.est = full ? b.d_sh_est : nullptr;
    a2.mo_ts = (int64_t)KB_TILE * L2.mo_elems;
    a2.ext_phi = a2.ext_h = nullptr; a2.ext_ld = 0;
    const int rc = launch_srif(b, a2);
    if (rc) return rc;
    PadArgs back{};
    back.src = b.d_sh_state; back.dst = a.state; back.src_elems = L2.st_elems; back.dst_elems = a.L.st_elems; back.N = a.N; back.nf = 2;
    back.f[0] = PadField{L2.st_vec, a.L.st_vec, 1, n2, 1, n, 0, nullptr};
    back.f[1] = PadField{L2.st_mat, a.L.st_mat, n2, n2, n, n, 0, nullptr};
    pad_launch(b, back);
    b.sh_state_current = true;
    if (full) {
        PadArgs es{};
        es.src = b.d_sh_est; es.dst = a.est; es.src_elems = L2.es_elems; es.dst_elems = a.L.es_elems; es.N = a.N; es.nf = 5;
        es.f[0] = PadField{L2.es_ppred, a.L.es_ppred, n2, n2, n, n, 0, nullptr};
        es.f[1] = PadField{L2.es_gain, a.L.es_gain, n2, pm, n, pm, 0, nullptr};
        es.f[2] = PadField{L2.es_innov, a.L.es_innov, 1, pm, 1, pm, 0, nullptr};
        es.f[3] = PadField{L2.es_yhat, a.L.es_yhat, 1, pm, 1, pm, 0, nullptr};
        es.f[4] = PadField{L2.es_dobs, a.L.es_dobs, 1, pm, 1, pm, 0, nullptr};
        pad_launch(b, es);
    }
    KB_HIP(hipGetLastError());
    return KB_OK;
}

}  // namespace kb
