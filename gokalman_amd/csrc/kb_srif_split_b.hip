// kb_srif_split_b.hip -- SRIF Update / Predict in fp64, one filter over four (n <= 12) / eight lanes (kb_srif_split.h): n = 7 8 9 10, p <= 4 and p <= 8.
#include "kb_srif_split.h"

namespace kb {

KB_SRIF_SPLIT_TU(7)
KB_SRIF_SPLIT_TU(8)
KB_SRIF_SPLIT_TU(9)
KB_SRIF_SPLIT_TU(10)

}  // namespace kb
