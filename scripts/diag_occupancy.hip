// How many one-wave workgroups with LDS bytes of static LDS does a CU of this GPU hold?  (hipOccupancyMaxActiveBlocksPerMultiprocessor)
// hipcc --offload-arch=gfx950 -O2 scripts/diag_occupancy.hip -o /tmp/diag_occupancy && /tmp/diag_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ void __launch_bounds__(64, 2) k(double *o) {
    __shared__ double s[BYTES / 8];
    s[threadIdx.x] = o[threadIdx.x];
    __syncthreads();
    o[threadIdx.x] = s[(threadIdx.x * 7) % (BYTES / 8)];
}
template <int BYTES>
void one() {
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k<BYTES>, 64, 0);
    printf("static LDS %6d B: %d workgroups (waves) per CU\n", BYTES, nb);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: CUs %d, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu, regsPerBlock %d\n", p.gcnArchName, p.multiProcessorCount, p.sharedMemPerBlock,
           p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
    one<16384>(); one<18432>(); one<19200>(); one<19456>(); one<20480>(); one<20736>(); one<22528>(); one<24576>();
    return 0;
}
