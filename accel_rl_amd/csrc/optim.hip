// Flat-bucket optimiser step for gfx950: one fp32 vector holds every trainable
// parameter (the same flat vector the reference all-reduces,
// accel_rl/optimizers/util.py:35-39), so averaging, global-norm clipping and the
// adam / rmsprop update are two streaming launches over P floats instead of a
// per-tensor kernel zoo.
//
//   launch 1  sumsq_kernel   : partial sums of g^2 in f64 (fixed order => deterministic)
//   launch 2  update_kernel  : every block folds the partials, derives
//                              norm / clip scale / a_t, then streams p, g, m, v.
//
// Replaces (reference root): optimizers/util.py:63-76 (avg_grads_from_flat,
// apply_grad_norm_clip -> lasagne total_norm_constraint), the lasagne update
// restated in optimizers/update_methods_stats.py:11-33 (rmsprop) / :55-87 (adam),
// call order of optimizers/sync/sync_ppo_optimizer.py:27-34.
// Bound: HBM (adam: read p,g,m,v + write p,m,v = 28 B/param; + 4 B/param for the norm).

#include "arl_optim_dev.h"

namespace {

using arl::block_sum_d;
using arl::update_one;

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, int64_t n,
                                                    double* __restrict__ partials,
                                                    float* __restrict__ step_count) {
    __shared__ double lds[8];
    double s = 0;
    const int64_t n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = g4[i];
        s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        s += (double)v * v;
    }
    s = block_sum_d(s, lds);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = s;
        if (blockIdx.x == 0) step_count[0] += 1.0f;       // t = t_prev + 1 (update_methods_stats.py:66)
    }
}

template <int METHOD>
__global__ __launch_bounds__(256) void update_kernel(arl_opt_state o, int n_partials, float lr_base,
                                                     float avg, float clip, float b1, float b2,
                                                     float eps) {
    __shared__ double lds[8];
    double s = 0;
    for (int i = threadIdx.x; i < n_partials; i += blockDim.x) s += o.partials[i];
    s = block_sum_d(s, lds);
    // norm of the AVERAGED gradient (sync_ppo_optimizer.py:28-32: avg, then norm/clip)
    const float norm = avg * (float)sqrt(s);
    float cscale = 1.f;
    if (clip > 0.f) {
        const float target = fminf(fmaxf(norm, 0.f), clip);  // lasagne total_norm_constraint
        cscale = target / (1e-7f + norm);
    }
    const float t = o.step_count[0];
    const float lr = lr_base * o.lr_mult[0];
    float a_t = 0.f;
    if (METHOD == ARL_OPT_ADAM)
        a_t = lr * sqrtf(1.f - powf(b2, t)) / (1.f - powf(b1, t));   // :67
    if (blockIdx.x == 0 && threadIdx.x == 0 && o.grad_norm_log) {
        const int k = ((int)t - 1) % o.norm_log_len;
        o.grad_norm_log[k < 0 ? 0 : k] = norm;
    }
    const int64_t n = o.n_params, n4 = n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    float4* p4 = reinterpret_cast<float4*>(o.params);
    const float4* g4 = reinterpret_cast<const float4*>(o.grads);
    float4* m4 = reinterpret_cast<float4*>(o.slot0);
    float4* v4 = reinterpret_cast<float4*>(o.slot1);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = p4[i];
        const float4 g = g4[i];
        float4 m = m4[i];
        float4 v = (METHOD == ARL_OPT_ADAM) ? v4[i] : make_float4(0, 0, 0, 0);
        update_one<METHOD>(p.x, g.x, m.x, v.x, avg, cscale, lr, a_t, b1, b2, eps);
        update_one<METHOD>(p.y, g.y, m.y, v.y, avg, cscale, lr, a_t, b1, b2, eps);
        update_one<METHOD>(p.z, g.z, m.z, v.z, avg, cscale, lr, a_t, b1, b2, eps);
        update_one<METHOD>(p.w, g.w, m.w, v.w, avg, cscale, lr, a_t, b1, b2, eps);
        p4[i] = p;
        m4[i] = m;
        if (METHOD == ARL_OPT_ADAM) v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        float p = o.params[i], m = o.slot0[i], v = (METHOD == ARL_OPT_ADAM) ? o.slot1[i] : 0.f;
        update_one<METHOD>(p, o.grads[i], m, v, avg, cscale, lr, a_t, b1, b2, eps);
        o.params[i] = p;
        o.slot0[i] = m;
        if (METHOD == ARL_OPT_ADAM) o.slot1[i] = v;
    }
}

// ---- no norm clipping (PPO's default): nothing needs the global norm before the update, so the sum of squares
// rides along in the update's own pass over the gradient (ONE launch per step, and the bucket is read once):
// block b leaves its partial of update k at norm_parts[k][b]; arl_opt_finish turns the partials of a whole call
// (k = 0 .. n - 1) into logged norms.  Lasagne's t: update k reads step_pp[k & 1] and block 0 writes t to
// step_pp[(k + 1) & 1] (and to the public step_count) -- never the word the other blocks are reading.
template <int METHOD>
__global__ __launch_bounds__(256) void update_noclip_kernel(const arl::OptSeg c) {
    __shared__ double lds[8];
    arl::opt_update_block<METHOD>(c, (int)blockIdx.x, (int)gridDim.x, lds);
}

// one block per update of the call: fold its partials in block order, log the norm; block 0 also levels the
// two step words so that the next call's update 0 finds t whatever the parity of this call's length
__global__ __launch_bounds__(256) void opt_finish_kernel(arl_opt_state o, int n_updates, int n_blocks, float avg,
                                                         float* step_pp, const double* __restrict__ norm_parts) {
    __shared__ double lds[8];
    const int k = blockIdx.x;
    double s = 0;
    for (int i = threadIdx.x; i < n_blocks; i += blockDim.x) s += norm_parts[(int64_t)k * ARL_OPT_NORM_BLOCKS + i];
    s = block_sum_d(s, lds);
    if (threadIdx.x == 0) {
        if (o.grad_norm_log) o.grad_norm_log[k % o.norm_log_len] = avg * (float)sqrt(s);
        if (k == 0) { const float t = step_pp[n_updates & 1]; step_pp[0] = t; step_pp[1] = t; o.step_count[0] = t; }
    }
}

int check_opt(const arl_opt_state* opt, int32_t method) {
    ARL_REQUIRE(opt, ARL_E_ARG, "null state");
    ARL_REQUIRE(opt->params && opt->grads && opt->slot0 && opt->step_count && opt->lr_mult, ARL_E_ARG,
                "null pointer in state");
    ARL_REQUIRE(method == ARL_OPT_ADAM || method == ARL_OPT_RMSPROP, ARL_E_ARG, "unknown method");
    ARL_REQUIRE(method != ARL_OPT_ADAM || opt->slot1, ARL_E_ARG, "adam needs slot1");
    ARL_REQUIRE(opt->n_params > 0, ARL_E_ARG, "n_params <= 0");
    ARL_REQUIRE(!opt->grad_norm_log || opt->norm_log_len > 0, ARL_E_ARG, "norm_log_len <= 0");
    ARL_REQUIRE(arl::aligned16(opt->params) && arl::aligned16(opt->grads) && arl::aligned16(opt->slot0) &&
                    (!opt->slot1 || arl::aligned16(opt->slot1)), ARL_E_ALIGN, "flat buffers must be 16-byte aligned");
    return 0;
}

}  // namespace

namespace {
int check_noclip(const arl_opt_state* opt, int32_t method, int32_t k, const float* step_pp, const double* norm_parts) {
    int rc = check_opt(opt, method);
    if (rc) return rc;
    ARL_REQUIRE(step_pp && norm_parts, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(k >= 0 && k < ARL_OPT_NORM_SLOTS, ARL_E_RANGE, "update index outside the call's slots");
    return 0;
}

}  // namespace

namespace arl {
int launch_opt_seg(const OptSeg& c, int blocks, hipStream_t s) {
    if (c.method == ARL_OPT_ADAM)
        hipLaunchKernelGGL((update_noclip_kernel<ARL_OPT_ADAM>), dim3((unsigned)blocks), dim3(256), 0, s, c);
    else
        hipLaunchKernelGGL((update_noclip_kernel<ARL_OPT_RMSPROP>), dim3((unsigned)blocks), dim3(256), 0, s, c);
    return check_launch("update_noclip_kernel");
}


// the segment description of one no-clip update: part 0 = everything but the hole (the whole bucket when the hole is
// empty), part 1 = the hole
int make_opt_seg(OptSeg* c, const arl_opt_state* opt, int32_t method, float learning_rate, float avg_factor,
                 float beta1_or_rho, float beta2, float epsilon, int32_t k, float* step_pp, double* norm_parts,
                 int64_t hole_first, int64_t hole_count, int part, int* blocks) {
    int rc = check_noclip(opt, method, k, step_pp, norm_parts);
    if (rc) return rc;
    ARL_REQUIRE(hole_first >= 0 && hole_count >= 0 && hole_first + hole_count <= opt->n_params &&
                    (hole_first & 3) == 0 && (hole_count & 3) == 0, ARL_E_ARG,
                "hole: a float4-aligned range inside the bucket");
    ARL_REQUIRE(part == 0 || hole_count > 0, ARL_E_ARG, "part 1 needs a hole");
    int rest, hole;
    opt_split_plan(opt->n_params, hole_count, &rest, &hole);
    *c = OptSeg{};
    c->o = *opt; c->method = method; c->k = k;
    c->lr_base = learning_rate; c->avg = avg_factor; c->b1 = beta1_or_rho; c->b2 = beta2; c->eps = epsilon;
    c->step_pp = step_pp; c->norm_parts = norm_parts;
    const long long n4 = opt->n_params >> 2, h0 = hole_first >> 2, hn = hole_count >> 2;
    if (part == 0) {
        c->a0 = 0; c->an = h0; c->b0 = h0 + hn; c->bn = n4 - (h0 + hn);
        c->block0 = 0; c->finish = 1; c->slots = rest; *blocks = rest;
    } else {
        c->a0 = h0; c->an = hn; c->b0 = 0; c->bn = 0;
        c->block0 = rest; c->finish = 0; c->slots = hole; *blocks = hole;
    }
    return 0;
}
}  // namespace arl

extern "C" int arl_opt_step_noclip(const arl_opt_state* opt, int32_t method, float learning_rate, float avg_factor,
                                   float beta1_or_rho, float beta2, float epsilon, int32_t k, float* step_pp,
                                   double* norm_parts, void* stream) {
    return arl_opt_step_noclip_split(opt, method, learning_rate, avg_factor, beta1_or_rho, beta2, epsilon, k, step_pp,
                                     norm_parts, 0, 0, 0, stream);
}

extern "C" int arl_opt_step_noclip_split(const arl_opt_state* opt, int32_t method, float learning_rate,
                                         float avg_factor, float beta1_or_rho, float beta2, float epsilon, int32_t k,
                                         float* step_pp, double* norm_parts, int64_t hole_first, int64_t hole_count,
                                         int32_t part, void* stream) {
    arl::OptSeg c;
    int blocks = 0;
    int rc = arl::make_opt_seg(&c, opt, method, learning_rate, avg_factor, beta1_or_rho, beta2, epsilon, k, step_pp,
                               norm_parts, hole_first, hole_count, part, &blocks);
    if (rc) return rc;
    return arl::launch_opt_seg(c, blocks, (hipStream_t)stream);
}

extern "C" int arl_opt_finish(const arl_opt_state* opt, int32_t n_updates, float avg_factor, float* step_pp,
                              const double* norm_parts, void* stream) {
    return arl_opt_finish_split(opt, n_updates, avg_factor, step_pp, norm_parts, 0, stream);
}

extern "C" int arl_opt_finish_split(const arl_opt_state* opt, int32_t n_updates, float avg_factor, float* step_pp,
                                    const double* norm_parts, int64_t hole_count, void* stream) {
    ARL_REQUIRE(opt && opt->step_count && step_pp && norm_parts, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(n_updates >= 1 && n_updates <= ARL_OPT_NORM_SLOTS, ARL_E_RANGE, "n_updates outside 1 .. ARL_OPT_NORM_SLOTS");
    ARL_REQUIRE(hole_count >= 0 && hole_count <= opt->n_params && (hole_count & 3) == 0, ARL_E_ARG, "hole size");
    int rest, hole;
    arl::opt_split_plan(opt->n_params, hole_count, &rest, &hole);
    hipLaunchKernelGGL(opt_finish_kernel, dim3((unsigned)n_updates), dim3(256), 0, (hipStream_t)stream, *opt,
                       (int)n_updates, rest + hole, avg_factor, step_pp, norm_parts);
    return arl::check_launch("opt_finish_kernel");
}

extern "C" int arl_opt_step(const arl_opt_state* opt, int32_t method, float learning_rate,
                            float avg_factor, float clip, float beta1_or_rho, float beta2,
                            float epsilon, void* stream) {
    ARL_REQUIRE(opt, ARL_E_ARG, "null state");
    ARL_REQUIRE(opt->params && opt->grads && opt->slot0 && opt->step_count && opt->lr_mult &&
                    opt->partials, ARL_E_ARG, "null pointer in state");
    ARL_REQUIRE(method == ARL_OPT_ADAM || method == ARL_OPT_RMSPROP, ARL_E_ARG, "unknown method");
    ARL_REQUIRE(method != ARL_OPT_ADAM || opt->slot1, ARL_E_ARG, "adam needs slot1");
    ARL_REQUIRE(opt->n_params > 0, ARL_E_ARG, "n_params <= 0");
    ARL_REQUIRE(!opt->grad_norm_log || opt->norm_log_len > 0, ARL_E_ARG, "norm_log_len <= 0");
    ARL_REQUIRE(arl::aligned16(opt->params) && arl::aligned16(opt->grads) && arl::aligned16(opt->slot0) &&
                    (!opt->slot1 || arl::aligned16(opt->slot1)), ARL_E_ALIGN, "flat buffers must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    int64_t nb = ((opt->n_params >> 2) + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > ARL_OPT_PARTIALS) nb = ARL_OPT_PARTIALS;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)nb), dim3(256), 0, s, opt->grads, opt->n_params,
                       opt->partials, opt->step_count);
    int rc = arl::check_launch("sumsq_kernel");
    if (rc) return rc;
    const unsigned grid = arl::stream_grid(opt->n_params >> 2, 256);
    if (method == ARL_OPT_ADAM)
        hipLaunchKernelGGL((update_kernel<ARL_OPT_ADAM>), dim3(grid), dim3(256), 0, s, *opt, (int)nb,
                           learning_rate, avg_factor, clip, beta1_or_rho, beta2, epsilon);
    else
        hipLaunchKernelGGL((update_kernel<ARL_OPT_RMSPROP>), dim3(grid), dim3(256), 0, s, *opt, (int)nb,
                           learning_rate, avg_factor, clip, beta1_or_rho, beta2, epsilon);
    return arl::check_launch("update_kernel");
}
