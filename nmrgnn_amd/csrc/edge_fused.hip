// placeholder: fused persistent edge kernels land here (H == 128 fast path)
#include "ng_internal.h"
