// f32 (parity mode) instantiations of the conv kernels.
#define EY_CONV_PART 32
#include "conv_igemm.inc.h"
