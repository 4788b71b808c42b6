// Launcher instantiations, part 4 of 7 (see the end of mfma_dispatch.h).
#define ARL_CONV_PART 4
#include "mfma_conv_impl.h"
