// Launcher instantiations, part 6 of 7 (see the end of mfma_dispatch.h).
#define ARL_CONV_PART 6
#include "mfma_conv_impl.h"
