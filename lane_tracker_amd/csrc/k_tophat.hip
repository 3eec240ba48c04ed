// placeholder translation unit: the decomposed 29x29 / 55x55 top-hat kernels land here.
#include "lt_internal.h"
