// Launcher instantiations, part 3 of 7 (see the end of mfma_dispatch.h).
#define ARL_CONV_PART 3
#include "mfma_conv_impl.h"
