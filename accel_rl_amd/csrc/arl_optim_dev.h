// Device side of the flat-bucket optimiser update, shared by optim.hip (its own launches) and mfma_conv.hip (a data
// gradient's launch can carry the update of a finished gradient range in extra workgroups: arl_conv_corun_update).
#pragma once

#include "arl_common.h"

namespace arl {

__device__ __forceinline__ double wave_sum_d(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    return x;
}

__device__ __forceinline__ double block_sum_d(double x, double* lds) {
    x = wave_sum_d(x);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) lds[w] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < nw; ++i) s += lds[i];
        lds[0] = s;
    }
    __syncthreads();
    const double r = lds[0];
    __syncthreads();
    return r;
}

template <int METHOD>
__device__ __forceinline__ void update_one(float& p, float g, float& s0, float& s1, float avg,
                                           float cscale, float lr, float a_t, float b1, float b2,
                                           float eps) {
    const float gg = (g * avg) * cscale;                     // util.py:66, then total_norm_constraint
    if (METHOD == ARL_OPT_ADAM) {
        const float m = b1 * s0 + (1.f - b1) * gg;          // :76
        const float v = b2 * s1 + (1.f - b2) * (gg * gg);    // :77
        const float step = a_t * m / (sqrtf(v) + eps);       // :78
        s0 = m; s1 = v;
        p = p - step;
    } else {
        const float acc = b1 * s0 + (1.f - b1) * (gg * gg);  // :24 (rho = b1)
        const float step = lr * gg / sqrtf(acc + eps);       // :28
        s0 = acc;
        p = p - step;
    }
}

// One no-clip update of the float4 ranges [a0, a0 + an) and [b0, b0 + bn) of the flat bucket (the whole bucket: a =
// everything, b empty; "everything but a hole": a below it, b above it; "the hole": a = the hole).  Workgroup `block`
// of `nblocks` leaves its sum of squared gradients at norm_parts[k][block0 + block]; the launch with `finish` set
// also handles the bucket's last n % 4 elements and advances Lasagne's t (step_pp ping-pong: update k reads word
// k & 1, the finishing launch writes word (k + 1) & 1 -- never the word other workgroups are reading).
struct OptSeg {
    arl_opt_state o;
    int method, k;
    float lr_base, avg, b1, b2, eps;
    float* step_pp;
    double* norm_parts;
    long long a0, an, b0, bn;
    int block0, finish;
    int slots;                      // norm-partial slots reserved for this part (>= the workgroups that run it)
    int co_blocks;                  // co-run only: workgroups blockIdx.x < co_blocks of the hosting grid run the update
};

template <int METHOD>
__device__ __forceinline__ void opt_update_block(const OptSeg& c, int block, int nblocks, double* lds) {
    const arl_opt_state& o = c.o;
    const float t = c.step_pp[c.k & 1] + 1.0f;                      // update_methods_stats.py:66
    const float lr = c.lr_base * o.lr_mult[0];
    float a_t = 0.f;
    if (METHOD == ARL_OPT_ADAM)
        a_t = lr * sqrtf(1.f - powf(c.b2, t)) / (1.f - powf(c.b1, t));   // :67
    float4* p4 = reinterpret_cast<float4*>(o.params);
    const float4* g4 = reinterpret_cast<const float4*>(o.grads);
    float4* m4 = reinterpret_cast<float4*>(o.slot0);
    float4* v4 = reinterpret_cast<float4*>(o.slot1);
    // gradient and optimiser slots are streamed (touched once per step): non-temporal, so that they do not push the
    // parameters -- which the next forward pass reads -- out of the caches
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    auto ld_nt = [](const float4* q) {
        const f32x4_t x = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(q));
        return make_float4(x.x, x.y, x.z, x.w);
    };
    auto st_nt = [](float4* q, const float4& x) {
        f32x4_t y = {x.x, x.y, x.z, x.w};
        __builtin_nontemporal_store(y, reinterpret_cast<f32x4_t*>(q));
    };
    double s = 0;
    const long long total = c.an + c.bn, stride = (long long)nblocks * blockDim.x;
    for (long long j = (long long)block * blockDim.x + threadIdx.x; j < total; j += stride) {
        const long long i = j < c.an ? c.a0 + j : c.b0 + (j - c.an);
        float4 p = p4[i];
        const float4 g = ld_nt(g4 + i);
        float4 m = ld_nt(m4 + i);
        float4 v = (METHOD == ARL_OPT_ADAM) ? ld_nt(v4 + i) : make_float4(0, 0, 0, 0);
        s += (double)g.x * g.x + (double)g.y * g.y + (double)g.z * g.z + (double)g.w * g.w;
        update_one<METHOD>(p.x, g.x, m.x, v.x, c.avg, 1.f, lr, a_t, c.b1, c.b2, c.eps);
        update_one<METHOD>(p.y, g.y, m.y, v.y, c.avg, 1.f, lr, a_t, c.b1, c.b2, c.eps);
        update_one<METHOD>(p.z, g.z, m.z, v.z, c.avg, 1.f, lr, a_t, c.b1, c.b2, c.eps);
        update_one<METHOD>(p.w, g.w, m.w, v.w, c.avg, 1.f, lr, a_t, c.b1, c.b2, c.eps);
        p4[i] = p;
        st_nt(m4 + i, m);
        if (METHOD == ARL_OPT_ADAM) st_nt(v4 + i, v);
    }
    const long long n = o.n_params;
    if (c.finish && block == 0 && threadIdx.x < (n & 3)) {
        const long long i = ((n >> 2) << 2) + threadIdx.x;
        const float g = o.grads[i];
        float p = o.params[i], m = o.slot0[i], v = (METHOD == ARL_OPT_ADAM) ? o.slot1[i] : 0.f;
        s += (double)g * g;
        update_one<METHOD>(p, g, m, v, c.avg, 1.f, lr, a_t, c.b1, c.b2, c.eps);
        o.params[i] = p;
        o.slot0[i] = m;
        if (METHOD == ARL_OPT_ADAM) o.slot1[i] = v;
    }
    s = block_sum_d(s, lds);
    if (threadIdx.x == 0) {
        c.norm_parts[(long long)c.k * ARL_OPT_NORM_BLOCKS + c.block0 + block] = s;
        if (c.finish && block == 0) { c.step_pp[(c.k + 1) & 1] = t; o.step_count[0] = t; }
    }
    // fewer workgroups than reserved slots (a host launch runs the part with one workgroup per CU): the others read 0
    if (block == 0)
        for (int i = nblocks + threadIdx.x; i < c.slots; i += blockDim.x)
            c.norm_parts[(long long)c.k * ARL_OPT_NORM_BLOCKS + c.block0 + i] = 0.0;
}

// The workgroup counts of a split update: the rest of the bucket first (slots [0, rest)), the hole behind it
// (slots [rest, rest + hole)); together <= ARL_OPT_NORM_BLOCKS.  hole_count4 = 0: the plain update's grid.
inline void opt_split_plan(long long n_params, long long hole_count, int* rest, int* hole) {
    const long long n4 = n_params >> 2, h4 = hole_count >> 2;
    *rest = (int)stream_grid(n4 - h4, 256);
    if (h4 > 0 && *rest > ARL_OPT_NORM_BLOCKS / 2) *rest = ARL_OPT_NORM_BLOCKS / 2;     // (the loops are grid-stride)
    long long hb = (h4 + 511) / 512;                    // two float4 per thread: small workgroups fill a host's tail
    if (hb > ARL_OPT_NORM_BLOCKS - *rest) hb = ARL_OPT_NORM_BLOCKS - *rest;
    *hole = h4 > 0 ? (int)(hb < 1 ? 1 : hb) : 0;
}

// optim.hip: the segment description of one no-clip update (part 0 = everything but the hole, part 1 = the hole) and
// its own launch
int make_opt_seg(OptSeg* c, const arl_opt_state* opt, int32_t method, float learning_rate, float avg_factor,
                 float beta1_or_rho, float beta2, float epsilon, int32_t k, float* step_pp, double* norm_parts,
                 int64_t hole_first, int64_t hole_count, int part, int* blocks);
int launch_opt_seg(const OptSeg& c, int blocks, hipStream_t s);

}  // namespace arl
