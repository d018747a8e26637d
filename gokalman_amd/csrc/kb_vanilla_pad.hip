// kb_vanilla_pad.hip -- padded register-resident Vanilla kernels (kb_vanilla_reg.h, PAD = true), fp64: every shape
// with n <= 8, p <= 4, m <= 2 that has no exact instantiation runs on the next larger one instead of the
// run-time-dimension scratch kernel (vanilla_gen_kernel), which stays for what is larger still.
#include "kb_vanilla_reg.h"

namespace kb {

bool launch_vanilla_padded(const Batch &b, const StepArgs &a) {
    return try_pad<double, 4, 2, 0>(b, a) || try_pad<double, 4, 2, 2>(b, a) || try_pad<double, 4, 4, 0>(b, a) || try_pad<double, 4, 4, 2>(b, a) ||
           try_pad<double, 6, 2, 0>(b, a) || try_pad<double, 6, 2, 2>(b, a) || try_pad<double, 6, 4, 0>(b, a) || try_pad<double, 6, 4, 2>(b, a);
}

}  // namespace kb
