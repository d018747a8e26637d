// kb_vanilla_noise.hip -- register-resident Vanilla kernels (kb_vanilla_reg.h, NOISE = true) for batches whose Noise is AWGN
// or BatchNoise (noise.go:67-164), fp64, one step per launch: the benchmark shapes exactly, and the padded 4 / 2 family
// (every shape of the reference's examples: robot 2/1/1, jerkcar 4/1/1 and 4/2/1, statOD5044 4/2/2).  The larger padded
// members live in kb_vanilla_noise_pad.hip (a separate translation unit: compiled in parallel).
#include "kb_vanilla_reg.h"

namespace kb {

// The caller loop fused into one launch with the noise drawn inside (round 5): state-only outputs, 6 / 3 / no control; x, P and the
// model stay in registers over a.nsteps steps (one wave per SIMD, as the Noiseless time-fused kernel).
bool launch_vanilla_noise_fused(const Batch &b, const StepArgs &a) {
    if (!vanilla_noise_fused_ok(b, a)) return false;
    const dim3 grid((unsigned)((a.ntiles + KB_VANILLA_WPB - 1) / KB_VANILLA_WPB)), block(KB_VANILLA_WPB * 64);
    KB_LAUNCH((vanilla_reg_kernel<double, 6, 3, 0, false, false, true, false, true, false>), grid, block, 0, b.stream, a);
    return true;
}

bool launch_vanilla_noise(const Batch &b, const StepArgs &a) {
    return try_reg<double, 6, 3, 0, false, true>(b, a, false) || try_reg<double, 4, 2, 0, false, true>(b, a, false) ||
           try_pad<double, 4, 2, 0, true>(b, a) || try_pad<double, 4, 2, 2, true>(b, a);
}

}  // namespace kb
