// kb_srif_split_a.hip -- SRIF Update / Predict in fp64, one filter over four (n <= 12) / eight lanes (kb_srif_split.h): n = 1 2 3 4 5 6, p <= 4 and p <= 8.
#include "kb_srif_split.h"

namespace kb {

KB_SRIF_SPLIT_TU_SMALL(1)
KB_SRIF_SPLIT_TU_SMALL(2)
KB_SRIF_SPLIT_TU_SMALL(3)
KB_SRIF_SPLIT_TU_SMALL(4)
KB_SRIF_SPLIT_TU_SMALL(5)
KB_SRIF_SPLIT_TU(6)

}  // namespace kb
