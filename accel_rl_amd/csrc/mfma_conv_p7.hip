// Launcher instantiations, part 7 of 7 (see the end of mfma_dispatch.h).
#define ARL_CONV_PART 7
#include "mfma_conv_impl.h"
