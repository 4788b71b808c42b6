// The data gradients' k-contiguous weight copies (arl_conv2d_dgrad_weights): the permutation as a device function, so that
// it can run as its own launch (mfma_conv.hip) or in extra workgroups of another one (learner.hip: the head kernel of a
// backward pass hosts it -- one launch less per minibatch).
#pragma once
#include "arl_common.h"

namespace arlw {

// Weights (out_c, kh, kw, in_c) -> per input-pixel parity class (ph, pw) of a stride-s data gradient a matrix
// wt[z][c][(ty * taps_x + tx) * out_c + k] = w[k][i0 + s ty][j0 + s tx][c], (i0, j0) = ((ph + pad_h) % s, (pw + pad_w) % s),
// z = ph * s + pw: exactly the element the data gradient's reduction index (tap, k) meets in column c (dgrad_impl).
struct DgradWtItem { const float* w; float* wt; int K, kh, kw, C, st, taps_x, kred, total, i0[4], j0[4]; };
struct DgradWtArgs { DgradWtItem it[ARL_DGRAD_WT_MAX]; int block_start[ARL_DGRAD_WT_MAX + 1]; int n; };

// element idx of item q (one thread)
__device__ __forceinline__ void dgrad_wt_element(const DgradWtItem& q, const int idx) {
    if (idx >= q.total) return;
    const int per = q.C * q.kred;
    const int z = idx / per, rem = idx - z * per;
    const int c = rem / q.kred, r = rem - c * q.kred;
    const int tap = r / q.K, k = r - tap * q.K;
    const int ty = tap / q.taps_x, tx = tap - ty * q.taps_x;
    q.wt[idx] = q.w[((k * q.kh + q.i0[z] + q.st * ty) * q.kw + q.j0[z] + q.st * tx) * q.C + c];
}
// workgroup `block` (256 threads) of the items' joint grid
__device__ __forceinline__ void dgrad_wt_block(const DgradWtArgs& a, const int block, const int tid) {
    int i = 0;
    while (i + 1 < a.n && block >= a.block_start[i + 1]) ++i;            // uniform
    dgrad_wt_element(a.it[i], (block - a.block_start[i]) * 256 + tid);
}

// host: the items' descriptions and their joint grid (0 blocks: nothing to do); geometry checks as arl_conv2d_dgrad_weights
int dgrad_wt_plan(const arl_dgrad_wt* items, int32_t n, DgradWtArgs* out, int* blocks);

}  // namespace arlw
