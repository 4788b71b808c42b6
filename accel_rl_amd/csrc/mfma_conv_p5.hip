// Launcher instantiations, part 5 of 7 (see the end of mfma_dispatch.h).
#define ARL_CONV_PART 5
#include "mfma_conv_impl.h"
