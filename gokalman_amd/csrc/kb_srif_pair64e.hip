// kb_srif_pair64e.hip -- more shapes of the two-lanes-per-filter SRIF Update (kb_srif_pair.h), fp64: 6, 8, 10 and 12 states with 7 or 8
// measurements (7 on the eight-row instantiation with one padded row: kb_srif_pair.h PADM).
#include "kb_srif_pair.h"

namespace kb {
bool launch_srif_pair_f64e(const Batch &b, const StepArgs &a) {
    return srif_pair_launch<double, 6, 8, true>(b, a) || srif_pair_launch<double, 8, 8, true>(b, a) || srif_pair_launch<double, 10, 8, true>(b, a) ||
           srif_pair_launch<double, 12, 8, true>(b, a);   // (208-268 B of scratch at one wave per SIMD: 12 / 6 takes 508 registers)
}
}  // namespace kb
