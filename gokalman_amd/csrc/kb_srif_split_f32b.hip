// kb_srif_split_f32b.hip -- SRIF Update / Predict in fp32, one filter over four lanes (kb_srif_split.h): n = 9 11 -- shapes the two-lane
// fp32 kernels (kb_srif_pair32*.hip) do not serve (odd n, n < 6; at 14 / 16 states Predict() and p = 7, 8); p <= 4, 6 and 8.
#include "kb_srif_split.h"

namespace kb {

KB_SRIF_SPLIT_TU_F32(9)
KB_SRIF_SPLIT_TU_F32(11)
#ifdef KB_DIAG_SRIF_F32_N12   // diagnostic builds only (profiles/NOTES.md round 6): config E on FOUR lanes per filter, against the two-lane kernel
void launch_srif_split_f32_n12(const Batch &b, const StepArgs &a) { srif_split_launch<float, 12, 6>(b, a); }
#endif

}  // namespace kb
