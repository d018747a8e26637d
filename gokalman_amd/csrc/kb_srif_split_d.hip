// kb_srif_split_d.hip -- SRIF Update / Predict in fp64, one filter over four (n <= 12) / eight lanes (kb_srif_split.h): n = 13 14, p <= 4 and p <= 8.
#include "kb_srif_split.h"

namespace kb {

KB_SRIF_SPLIT_TU(13)
KB_SRIF_SPLIT_TU(14)

}  // namespace kb
