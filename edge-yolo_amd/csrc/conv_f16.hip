// f16 (throughput mode) instantiations of the conv kernels + the extern "C" entry points.
#define EY_CONV_PART 16
#include "conv_igemm.inc.h"
