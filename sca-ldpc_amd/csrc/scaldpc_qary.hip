// placeholder translation unit; q-ary kernels follow
#include "scaldpc_common.h"
