// kb_device.h -- device-side building blocks shared by the filter kernels.
//
// Execution model: ONE FILTER PER LANE.  A wavefront (64 lanes) owns one "tile" of
// 64 consecutive filters; every per-filter quantity lives in that lane's VGPRs.
// HBM layout is AoSoA-64: element e of the filter in lane l of tile t sits at
//     block[t * (64 * elems) + e * 64 + l]
// so every wave-level load/store is one fully coalesced 512-byte (f64) row, and a
// tile's whole working set is one contiguous stream (DRAM-page and TLB friendly).
// Symmetric matrices are stored packed (upper triangle, column-major order of the
// triangle: idx(i,j) = j(j+1)/2 + i for i <= j), which is exactly what the
// reference's mat64.SymDense keeps meaningful (helper.go:65-84).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "kb_normal.h"

#define KB_TILE 64

namespace kb {

__host__ __device__ constexpr int tri(int n) { return n * (n + 1) / 2; }
// packed index of symmetric element (i,j), any order of i,j
__host__ __device__ constexpr int symi(int i, int j) { return i <= j ? j * (j + 1) / 2 + i : i * (i + 1) / 2 + j; }

// Tile accessors shared by every kernel: `p` already points at this lane's slot of element 0 of its tile
// (block + tile*64*elems + lane), element e is 64 values further per step.
template <typename T>
__device__ __forceinline__ T ldt(const T *p, int e) { return p[(int64_t)e * KB_TILE]; }
template <typename T>
__device__ __forceinline__ T ldnt(const T *p, int e) { return __builtin_nontemporal_load(p + (int64_t)e * KB_TILE); }  // read-once streams
template <typename T>
__device__ __forceinline__ void stt(T *p, int e, T v) { p[(int64_t)e * KB_TILE] = v; }
// write-once outputs nobody reads on the device (the Estimate extras: P-, K, innovation, yhat): non-temporal, so that they do
// not push the state block out of the Infinity Cache
template <typename T>
__device__ __forceinline__ void stnt(T *p, int e, T v) { __builtin_nontemporal_store(v, p + (int64_t)e * KB_TILE); }

// State-block accesses with the cache policy as a compile-time flag: the kernels pick it per batch through a wave-uniform branch
// (StepArgs::stream_state; the policy and its measurements: kb_vanilla_reg.h)
template <bool NT, typename T>
__device__ __forceinline__ T ldp(const T *p, int e) {
    if constexpr (NT) return ldnt(p, e);
    else return ldt(p, e);
}
template <bool NT, typename T>
__device__ __forceinline__ void stp(T *p, int e, T v) {
    if constexpr (NT) stnt(p, e, v);
    else stt(p, e, v);
}
// (the empty asm statements keep the two arms apart: without them LLVM merges the otherwise identical loads / stores of both
// arms into one copy and drops the non-temporal hint, which is metadata -- measured: no effect at all until they were added)
#define KB_WITH_STATE_POLICY(a, fn)                                    \
    do {                                                               \
        if ((a).stream_state) {                                        \
            asm volatile("; state block: streaming policy" ::: "memory"); \
            fn(std::true_type{});                                      \
            asm volatile("; end of streaming arm" ::: "memory");      \
        } else {                                                       \
            fn(std::false_type{});                                     \
        }                                                              \
    } while (0)

// pin(v): an empty asm that "modifies" v.  The value must exist in a VGPR at this point of the program, so LLVM can neither
// sink the computation that produces it into a later basic block (machine sinking does that across the data-dependent
// branches of the factorisations, and drags the producers' operands along as live registers) nor rematerialise it later.
template <typename T>
__device__ __forceinline__ void pin(T &v) { asm volatile("" : "+v"(v)); }

// Read-once global loads at (wave-uniform pointer) + (32-bit per-lane BYTE offset), in the scalar-base form of global_load
// (`global_load_dword v, v_off, s[base:base+1]`): no 64-bit vector address arithmetic and no address register pair per
// access.  UniformCursor keeps the pointer in SGPRs (it goes through readfirstlane, so the compiler neither folds the lane
// offset into it nor re-derives every address with a 64-bit multiply: advancing costs two scalar adds) and in the global
// address space (a flat load has no scalar-base form).
template <typename T>
struct UniformCursor {
    unsigned lo, hi;
    __device__ __forceinline__ explicit UniformCursor(const T *uniform_ptr) { set((unsigned long long)uniform_ptr); }
    __device__ __forceinline__ void set(unsigned long long v) {
        lo = __builtin_amdgcn_readfirstlane((unsigned)v);
        hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    }
    __device__ __forceinline__ void advance(long long elements) { set((((unsigned long long)hi << 32) | lo) + (unsigned long long)(elements * (long long)sizeof(T))); }
    __device__ __forceinline__ T load_nt(unsigned lane_bytes) const {
        typedef const __attribute__((address_space(1))) T *gptr;
        typedef const __attribute__((address_space(1))) char *gbytes;
        return __builtin_nontemporal_load((gptr)((gbytes)(((unsigned long long)hi << 32) | lo) + lane_bytes));
    }
};
template <typename T>
__device__ __forceinline__ T ld_uniform_nt(const T *uniform_ptr, unsigned lane_bytes) { return UniformCursor<T>(uniform_ptr).load_nt(lane_bytes); }

// compile-time loop: f(integral_constant<int, B>), ..., f(integral_constant<int, E - 1>)
template <int B, int E, class F>
__device__ __forceinline__ void sfor(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        sfor<B + 1, E>(f);
    }
}

// 1 / x from the hardware reciprocal (1 ulp) refined by Newton steps: 3 (fp32) / 5 (fp64) instructions where the IEEE
// division sequence takes 10 / 15.  The result is within an ulp of the correctly rounded quotient; x = 0 gives a non-finite
// value as 1 / 0 does (the callers flag a zero divisor themselves).
__device__ __forceinline__ float recip(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    return fmaf(r, fmaf(-x, r, 1.0f), r);
}
__device__ __forceinline__ double recip(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    return fma(r, fma(-x, r, 1.0), r);
}

// threadIdx.x as a value the optimiser cannot connect to earlier uses: what is derived from it HERE is computed here, not kept alive
// from the top of the kernel (address terms that are needed once, late)
__device__ __forceinline__ unsigned late_lane() {
    unsigned t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// Element (rt + c) of a tile block -- rt wave-uniform at run time, c a compile-time constant -- as (scalar anchor made opaque to the
// optimiser) + (an immediate within +-8 elements), in the global address space; the lane adds ONE unsigned 32-bit element offset:
// the scalar-base form of global_load.  Left alone, instruction selection adds the part of c that does not fit the 13-bit immediate
// to the VECTOR half of the address: a 64-bit VGPR pair and a v_lshl_add_u64 per 8 elements, formed at the top of the kernel and
// carried (or spilled) to the access (kb_vanilla_split.h; the late reads of kb_information_reg.hip).
template <typename T>
__device__ __forceinline__ const __attribute__((address_space(1))) T *anchored(const T *ubase, int rt, int c) {
    typedef const __attribute__((address_space(1))) T *gptr;
    const int anchor = (c >= 0 ? c / 16 : -((-c + 15) / 16)) * 16 + 8;
    unsigned long long s = (unsigned long long)(ubase + (int64_t)(rt + anchor) * KB_TILE);
    asm("" : "+s"(s));
    return (gptr)s + (c - anchor) * KB_TILE;
}

template <typename T> struct Eps;
template <> struct Eps<double> { static constexpr double tiny = 2.2250738585072014e-308; };
template <> struct Eps<float>  { static constexpr float  tiny = 1.17549435e-38f; };

// floats.EqualWithinAbsOrRel(a, b, 1e-6, 1e-2) as used by AsSymDense (helper.go:75)
template <typename T>
__device__ __forceinline__ bool sym_close(T a, T b) {
    if (a == b) return true;
    T d = fabs(a - b);
    if (d <= T(1e-6)) return true;
    if (d <= Eps<T>::tiny) return d <= T(1e-2) * Eps<T>::tiny;
    return d / fmax(fabs(a), fabs(b)) <= T(1e-2);
}

// ---------------------------------------------------------------------------
// mat64.Dense.Inverse restated for a register-resident P x P matrix:
// LU with partial pivoting (row exchanges by select, no dynamic indexing),
// then forward/back substitution against the identity.  Returns true when the
// reference would return a Condition error: exact zero pivot, or
// cond_inf = |A|_inf |A^-1|_inf > 1e16 (gonum matrix.ConditionTolerance), or NaN.
// ---------------------------------------------------------------------------
// `nreal`: rows >= nreal are identity padding (kb_vanilla_reg.h PAD) and stay out of the norms.
// FASTDIV: the reciprocals come from recip() (within an ulp of the IEEE quotient, a third of its instructions): the instruction-bound
// time-fused kernel uses it.
template <typename T, int P, bool FASTDIV = false>
__device__ __forceinline__ bool inverse_lu(const T (&Ain)[P * P], T (&X)[P * P], int nreal = P) {
    T a[P * P], b[P * P];
    T anorm = T(0);
#pragma unroll
    for (int i = 0; i < P; i++) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < P; j++) {
            a[i * P + j] = Ain[i * P + j];
            b[i * P + j] = (i == j) ? T(1) : T(0);
            s += fabs(Ain[i * P + j]);
        }
        if (i < nreal) anorm = (s > anorm || s != s) ? s : anorm;
    }
    bool bad = false;
#pragma unroll
    for (int j = 0; j < P; j++) {
        // bring the largest |a[r][j]|, r >= j, to row j
#pragma unroll
        for (int r = j + 1; r < P; r++) {
            const bool sw = fabs(a[r * P + j]) > fabs(a[j * P + j]);
#pragma unroll
            for (int c = 0; c < P; c++) {
                if (c >= j) {
                    const T t0 = a[j * P + c], t1 = a[r * P + c];
                    a[j * P + c] = sw ? t1 : t0;
                    a[r * P + c] = sw ? t0 : t1;
                }
                const T u0 = b[j * P + c], u1 = b[r * P + c];
                b[j * P + c] = sw ? u1 : u0;
                b[r * P + c] = sw ? u0 : u1;
            }
        }
        const T piv = a[j * P + j];
        bad = bad || (piv == T(0));
        const T rp = FASTDIV ? recip(piv) : T(1) / piv;
#pragma unroll
        for (int r = j + 1; r < P; r++) {
            const T l = a[r * P + j] * rp;
#pragma unroll
            for (int c = j + 1; c < P; c++) a[r * P + c] -= l * a[j * P + c];
#pragma unroll
            for (int c = 0; c < P; c++) b[r * P + c] -= l * b[j * P + c];
        }
    }
    // back substitution U X = b
    T inorm = T(0);
#pragma unroll
    for (int i = P - 1; i >= 0; i--) {
        const T rd = FASTDIV ? recip(a[i * P + i]) : T(1) / a[i * P + i];
#pragma unroll
        for (int c = 0; c < P; c++) {
            T s = b[i * P + c];
#pragma unroll
            for (int k = i + 1; k < P; k++) s -= a[i * P + k] * X[k * P + c];
            X[i * P + c] = s * rd;
        }
    }
#pragma unroll
    for (int i = 0; i < P; i++) {
        T s = T(0);
#pragma unroll
        for (int c = 0; c < P; c++) s += fabs(X[i * P + c]);
        if (i < nreal) inorm = (s > inorm || s != s) ? s : inorm;
    }
    const T cond = anorm * inorm;
    return bad || !(cond <= T(1e16));
}

// Runtime-dimension version on a private array with leading dimension LD
// (generic kernels).  Same algorithm; `a` is destroyed, X receives the inverse.
template <typename T, int LD>
__device__ inline bool inverse_lu_rt(int p, T *a, T *X) {
#pragma clang fp contract(off)   // generic (statement-by-statement) kernels only: see kb_kinds.hip
    T b[LD * LD];
    T anorm = T(0);
    for (int i = 0; i < p; i++) {
        T s = T(0);
        for (int j = 0; j < p; j++) {
            b[i * LD + j] = (i == j) ? T(1) : T(0);
            s += fabs(a[i * LD + j]);
        }
        anorm = (s > anorm || s != s) ? s : anorm;
    }
    bool bad = false;
    for (int j = 0; j < p; j++) {
        int jp = j;
        T best = fabs(a[j * LD + j]);
        for (int r = j + 1; r < p; r++)
            if (fabs(a[r * LD + j]) > best) { best = fabs(a[r * LD + j]); jp = r; }
        if (jp != j)
            for (int c = 0; c < p; c++) {
                T t = a[j * LD + c]; a[j * LD + c] = a[jp * LD + c]; a[jp * LD + c] = t;
                t = b[j * LD + c]; b[j * LD + c] = b[jp * LD + c]; b[jp * LD + c] = t;
            }
        const T piv = a[j * LD + j];
        bad = bad || (piv == T(0));
        const T rp = T(1) / piv;
        for (int r = j + 1; r < p; r++) {
            const T l = a[r * LD + j] * rp;
            for (int c = j + 1; c < p; c++) a[r * LD + c] -= l * a[j * LD + c];
            for (int c = 0; c < p; c++) b[r * LD + c] -= l * b[j * LD + c];
        }
    }
    T inorm = T(0);
    for (int i = p - 1; i >= 0; i--) {
        const T rd = T(1) / a[i * LD + i];
        for (int c = 0; c < p; c++) {
            T s = b[i * LD + c];
            for (int k = i + 1; k < p; k++) s -= a[i * LD + k] * X[k * LD + c];
            X[i * LD + c] = s * rd;
        }
    }
    for (int i = 0; i < p; i++) {
        T s = T(0);
        for (int c = 0; c < p; c++) s += fabs(X[i * LD + c]);
        inorm = (s > inorm || s != s) ? s : inorm;
    }
    const T cond = anorm * inorm;
    return bad || !(cond <= T(1e16));
}

// ---------------------------------------------------------------------------
// Counter-based RNG for AWGN (noise.go:109-164): Philox4x32-10.
// key = (seed_lo, seed_hi); counter = (filter_lo, filter_hi, step, epoch<<8 | draw).
// ---------------------------------------------------------------------------
struct Philox {
    uint32_t c[4];
    __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
        const uint64_t p = (uint64_t)a * b;
        hi = (uint32_t)(p >> 32);
        lo = (uint32_t)p;
    }
    __host__ __device__ static inline void round4(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c[0], hi0, lo0);
        mulhilo(0xCD9E8D57u, c[2], hi1, lo1);
        const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __host__ __device__ static inline void gen(uint64_t seed, uint64_t filter, uint32_t step, uint32_t stream, uint32_t (&out)[4]) {
        uint32_t c[4] = {(uint32_t)filter, (uint32_t)(filter >> 32), step, stream};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; r++) {
            round4(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
    }
};

// two standard normals from one Philox block: Box-Muller on two 53-bit uniforms (two u32 each), u1 in (0, 1], u2 in [0, 1).
// The logarithm and the sine / cosine are kb_normal.h's (~75 instructions together where log + sincospi of the math library take
// ~160), written with explicit FMAs only, so kb_noise_sample on the host and the kernels produce the SAME BITS (sqrt and the
// conversions are correctly rounded on both sides).
__host__ __device__ inline void box_muller(const uint32_t (&r)[4], double &z0, double &z1) {
    const double two53 = 1.0 / 9007199254740992.0;
    const uint64_t a = (((uint64_t)r[0] << 32) | r[1]) >> 11;
    const uint64_t b = (((uint64_t)r[2] << 32) | r[3]) >> 11;
    const double u1 = ((double)a + 1.0) * two53;  // (0,1]
    const double u2 = (double)b * two53;          // [0,1)
    const double rad = sqrt(neg2log(u1));
    double sn, cs;
    sincos2pi(u2, sn, cs);
    z0 = rad * cs;
    z1 = rad * sn;
}

// standard normal number `k` (k = 0,1,2,...) of the vector drawn by filter `filter` at (step, stream)
__host__ __device__ inline double normal_at(uint64_t seed, uint64_t filter, uint32_t step, uint32_t stream, int k) {
    uint32_t r[4];
    Philox::gen(seed, filter, step, (stream << 8) | (uint32_t)(k >> 1), r);
    double z0, z1;
    box_muller(r, z0, z1);
    return (k & 1) ? z1 : z0;
}

}  // namespace kb
