// The policy network's dense contractions on the gfx950 matrix cores: kernels and launchers, by family --
//   mfma_common.h    descriptors (GatherDesc / WeightDesc / OutDesc / GemmArgs / WgradArgs), buffer-resource loads, the
//                    exact bf16 split, epilogue stores, XCD-aware tile placement
//   mfma_generic.h   generic fallbacks (any channel count / any K)
//   mfma_igemm.h     scalar-addressed forward / data gradient (igemm_body)
//   mfma_wgrad.h     scalar-addressed weight gradient (wgrad_fast_body)
//   mfma_pair.h      data + weight gradient of one layer in one launch
//   mfma_dispatch.h  launchers, per-call context, instantiation lists
// Entry points: mfma_conv.hip; instantiations: mfma_conv_p1 .. p7.hip.
#pragma once
#include "mfma_common.h"
#include "mfma_generic.h"
#include "mfma_igemm.h"
#include "mfma_wgrad.h"
#include "mfma_pair.h"
#include "mfma_dispatch.h"
