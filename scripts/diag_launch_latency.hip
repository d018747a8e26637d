// Floor of a one-filter step on this stack: how long from the host's launch call until the host SEES the result?
//   a  empty kernel + hipStreamSynchronize
//   b  kernel that writes a sequence number to pinned host memory (system-scope release) + the host spinning on it
//   c  two kernels back to back (step + snapshot) + hipStreamSynchronize         (kb_update_estimate today)
//   d  two kernels back to back, the second one writes the sequence number + host spin
//   e  like b, with hipStreamSynchronize every 64 calls (so the runtime retires its completion signals)
// hipcc --offload-arch=gfx950 -O2 scripts/diag_launch_latency.hip -o scripts/diag_launch_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void empty_k(int) {}
__global__ void work_k(double *buf) { buf[threadIdx.x] = buf[threadIdx.x] * 1.0000001 + 1e-9; }
__global__ void flag_k(double *buf, double *hostbuf, volatile unsigned *flag, unsigned seq) {
    const double v = buf[threadIdx.x] * 1.0000001 + 1e-9;
    buf[threadIdx.x] = v;
    hostbuf[threadIdx.x] = v;   // the "estimate", to pinned host memory
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <class F> static void timeit(const char *name, F &&fn) {
    for (int i = 0; i < 300; i++) fn(i);
    const int reps = 5000;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < reps; i++) fn(300 + i);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    std::printf("%-90s %6.1f us\n", name, us);
}

int main() {
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    double *buf; hipMalloc(&buf, 64 * 8); hipMemset(buf, 0, 64 * 8);
    char *h; hipHostMalloc((void **)&h, 4096, hipHostMallocMapped);
    char *d; hipHostGetDevicePointer((void **)&d, h, 0);
    volatile unsigned *hflag = (volatile unsigned *)h;
    unsigned *dflag = (unsigned *)d;
    double *dhost = (double *)(d + 1024);
    *hflag = 0;
    unsigned seq = 0;
    auto spin = [&](unsigned want) { while (__atomic_load_n((const unsigned *)hflag, __ATOMIC_ACQUIRE) != want) { } };
    timeit("a  empty kernel + hipStreamSynchronize", [&](int) { hipLaunchKernelGGL(empty_k, dim3(1), dim3(64), 0, s, 0); hipStreamSynchronize(s); });
    timeit("a' 64-lane kernel touching device memory + hipStreamSynchronize", [&](int) { hipLaunchKernelGGL(work_k, dim3(1), dim3(64), 0, s, buf); hipStreamSynchronize(s); });
    timeit("b  kernel writes result + sequence number to pinned memory, host spins", [&](int) {
        ++seq; hipLaunchKernelGGL(flag_k, dim3(1), dim3(64), 0, s, buf, dhost, dflag, seq); spin(seq); });
    hipStreamSynchronize(s);
    timeit("c  two kernels + hipStreamSynchronize", [&](int) {
        hipLaunchKernelGGL(work_k, dim3(1), dim3(64), 0, s, buf); hipLaunchKernelGGL(work_k, dim3(1), dim3(64), 0, s, buf); hipStreamSynchronize(s); });
    timeit("d  two kernels, the second writes the sequence number, host spins", [&](int) {
        ++seq; hipLaunchKernelGGL(work_k, dim3(1), dim3(64), 0, s, buf); hipLaunchKernelGGL(flag_k, dim3(1), dim3(64), 0, s, buf, dhost, dflag, seq); spin(seq); });
    hipStreamSynchronize(s);
    timeit("e  like b, hipStreamSynchronize every 64 calls", [&](int i) {
        ++seq; hipLaunchKernelGGL(flag_k, dim3(1), dim3(64), 0, s, buf, dhost, dflag, seq); spin(seq); if ((i & 63) == 63) hipStreamSynchronize(s); });
    hipStreamSynchronize(s);
    std::printf("last value %.9f\n", ((volatile double *)(h + 1024))[0]);
    return 0;
}
