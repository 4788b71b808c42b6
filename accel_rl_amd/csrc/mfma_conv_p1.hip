// Launcher instantiations, part 1 of 7 (see the end of mfma_dispatch.h).
#define ARL_CONV_PART 1
#include "mfma_conv_impl.h"
