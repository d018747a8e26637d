// kb_srif_split_e.hip -- SRIF Update / Predict in fp64, one filter over four (n <= 12) / eight lanes (kb_srif_split.h): n = 15 16, p <= 4 and p <= 8.
#include "kb_srif_split.h"

namespace kb {

KB_SRIF_SPLIT_TU(15)
KB_SRIF_SPLIT_TU(16)

}  // namespace kb
