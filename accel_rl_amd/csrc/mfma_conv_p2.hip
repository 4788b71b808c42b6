// Launcher instantiations, part 2 of 7 (see the end of mfma_dispatch.h).
#define ARL_CONV_PART 2
#include "mfma_conv_impl.h"
