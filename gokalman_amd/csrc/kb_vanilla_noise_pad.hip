// kb_vanilla_noise_pad.hip -- the larger padded members of the AWGN / BatchNoise Vanilla kernel family (kb_vanilla_noise.hip):
// any n <= 8, p <= 4, m <= 2.
#include "kb_vanilla_reg.h"

namespace kb {

bool launch_vanilla_noise_padded(const Batch &b, const StepArgs &a) {
    return try_pad<double, 6, 4, 0, true>(b, a) || try_pad<double, 6, 4, 2, true>(b, a) || try_pad<double, 8, 4, 0, true>(b, a) ||
           try_pad<double, 8, 4, 2, true>(b, a);
}

}  // namespace kb
