// A layer's data gradient and weight gradient in one launch (bwd_pair_kernel).  Part of mfma_conv_impl.h.
#pragma once
#include "mfma_igemm.h"
#include "mfma_wgrad.h"

namespace arlc {

// One launch for a layer's data gradient AND weight gradient (independent of each other, both read dy):
// workgroups [0, n_ig) run data-gradient tiles, the rest weight-gradient tiles.  One ramp-up and one
// tail instead of two, and the dispatcher fills the CUs the first problem's last wave leaves idle.
template <int DWGM, int DWGN, int DTM, int DTN, int WWGM, int WWGN, int WTM, int WTN, int BK, bool HAS_PAD, int SPLIT = 0>
__global__ __launch_bounds__(256) void bwd_pair_kernel(const GemmArgs a, const WgradArgs w, const int dgx, const int dgy,
                                                       const int n_ig, const int wgx, const int wgy) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int id = blockIdx.x;
    if (a.xcd) id = xcd_chunk(id, (int)gridDim.x);     // uniform
    if (id < n_ig) {
        const int bx = id % dgx, t = id / dgx;          // the row tiles over one weight panel are neighbours
        igemm_body<DWGM, DWGN, DTM, DTN, BK, false, false, HAS_PAD, false, false, SPLIT>(a, bx, t % dgy, t / dgy, smem);
    } else {
        id -= n_ig;
        if (a.xcd) {                                    // ... and so are the row tiles over one panel of the layer's input
            const int by = id % wgy, t = id / wgy;
            wgrad_fast_body<WWGM, WWGN, WTM, WTN, BK, HAS_PAD, false, false, SPLIT>(w, t % wgx, by, t / wgx, smem);
        } else {
            const int bx = id % wgx, t = id / wgx;
            wgrad_fast_body<WWGM, WWGN, WTM, WTN, BK, HAS_PAD, false, false, SPLIT>(w, bx, t % wgy, t / wgy, smem);
        }
    }
}

}  // namespace arlc
