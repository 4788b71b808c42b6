// Return / advantage scans for gfx950 (MI355X).
//
//   arl_gae_scan      <- gen_adv_est      accel_rl/algos/pg/util.py:6-23
//   arl_nstep_return  <- discount_returns accel_rl/algos/pg/util.py:26-37 (+ aac_base.py:121)
//   arl_valids_mask   <- update_valids / zero_after_reset  util.py:40-63
//
// The batch is env-major ([n_env][horizon], horizon = 5 in every reference
// config), so one env's segment is `horizon` consecutive floats.  The scan is a
// reverse-time recurrence that is independent across envs and must follow the
// reference's rounding order, so the short-horizon kernel keeps it sequential:
// one lane per env.  What would otherwise be a 20-byte-stride access pattern is
// staged through LDS: a workgroup copies its tile of EPB*horizon contiguous
// elements with 16-byte coalesced global loads into LDS, lanes then walk their
// own segment in LDS (stride `horizon` words; a +1-word skew per 32 words makes
// even horizons bank-conflict free), results go back through the same LDS tile
// and leave as 16-byte coalesced stores.  HBM traffic is exactly the algorithmic
// 17 B per (env, t) + 4 B per env.  Bound: HBM bandwidth.
//
// This file is compiled with -ffp-contract=off: the dtype walk below reproduces
// numpy's (unfused) arithmetic bit for bit.

#include "arl_common.h"

namespace {

__device__ __forceinline__ int skewed(int idx, int skew_mask) {
    return idx + ((idx >> 5) & skew_mask);
}

// One lane's reverse scan over its segment held in LDS.  out0/out1 overwrite
// the r/v tiles in place: GAE -> (adv, ret); NSTEP -> (ret, adv).
template <bool NSTEP, int PROMO>
__device__ __forceinline__ void lane_scan_lds(float* sr, float* sv, const uint8_t* sd,
                                              int lane_env, int T, int skew_mask,
                                              float last_v, double gamma, double gl) {
    const float g32 = (float)gamma;
    if (NSTEP) {
        if (PROMO == ARL_PROMO_NEP50) {
            float run = last_v;                       // util.py:29, np.float32 scalar
            for (int t = T - 1; t >= 0; --t) {
                const int idx = lane_env * T + t;
                const int p = skewed(idx, skew_mask);
                const float rr = sr[p], vv = sv[p];
                run = sd[idx] ? rr : (run * g32) + rr; // util.py:31-35 (f32 *=, +=)
                sr[p] = run;
                sv[p] = run - vv;                     // aac_base.py:121
            }
        } else {
            double run = (double)last_v;
            for (int t = T - 1; t >= 0; --t) {
                const int idx = lane_env * T + t;
                const int p = skewed(idx, skew_mask);
                const float rr = sr[p], vv = sv[p];
                run = sd[idx] ? (double)rr : (run * gamma) + (double)rr;
                const float ret = (float)run;
                sr[p] = ret;
                sv[p] = ret - vv;
            }
        }
    } else {
        double carry = 0.0;                           // util.py:13
        float v_next = last_v;                        // util.py:9
        // the step's operands are read one step ahead, so the LDS latency hides under the f64 chain
        int idx = lane_env * T + T - 1, p = skewed(idx, skew_mask);
        float rr_n = sr[p], vv_n = sv[p];
        uint8_t dd_n = sd[idx];
        for (int t = T - 1; t >= 0; --t) {
            const float rr = rr_n, vv = vv_n;
            const uint8_t dd = dd_n;
            const int pw = p;
            if (t > 0) {
                idx -= 1;
                p = skewed(idx, skew_mask);
                rr_n = sr[p]; vv_n = sv[p]; dd_n = sd[idx];
            }
            const double nd = dd ? 0.0 : 1.0;         // util.py:8 (int64 -> f64)
            const double gv = (PROMO == ARL_PROMO_NEP50) ? (double)(g32 * v_next)
                                                         : gamma * (double)v_next;
            const double delta = ((double)rr + gv * nd) - (double)vv;   // util.py:15
            carry = delta + (gl * nd) * carry;                          // util.py:16-17
            const float a = (float)carry;
            sr[pw] = a;
            sv[pw] = a + vv;                                            // util.py:21
            v_next = vv;
        }
    }
}

// The same walk over ONE chunk of a longer segment: `len` steps at LDS index lane_env * len + t, state
// (carry / running return, next value) handed from chunk to chunk in registers.
template <bool NSTEP, int PROMO>
__device__ __forceinline__ void lane_scan_chunk(float* sr, float* sv, const uint8_t* sd, int lane_env, int len,
                                                int skew_mask, double gamma, double gl, double& carry,
                                                float& run32, float& v_next) {
    const float g32 = (float)gamma;
    int idx = lane_env * len + len - 1, p = skewed(idx, skew_mask);
    float rr_n = sr[p], vv_n = sv[p];
    uint8_t dd_n = sd[idx];
    for (int t = len - 1; t >= 0; --t) {
        const float rr = rr_n, vv = vv_n;
        const uint8_t dd = dd_n;
        const int pw = p;
        if (t > 0) {
            idx -= 1;
            p = skewed(idx, skew_mask);
            rr_n = sr[p]; vv_n = sv[p]; dd_n = sd[idx];
        }
        if (NSTEP) {
            if (PROMO == ARL_PROMO_NEP50) {
                run32 = dd ? rr : (run32 * g32) + rr;                   // util.py:31-35 (f32 *=, +=)
                sr[pw] = run32;
                sv[pw] = run32 - vv;                                    // aac_base.py:121
            } else {
                carry = dd ? (double)rr : (carry * gamma) + (double)rr;
                const float ret = (float)carry;
                sr[pw] = ret;
                sv[pw] = ret - vv;
            }
        } else {
            const double nd = dd ? 0.0 : 1.0;                           // util.py:8 (int64 -> f64)
            const double gv = (PROMO == ARL_PROMO_NEP50) ? (double)(g32 * v_next) : gamma * (double)v_next;
            const double delta = ((double)rr + gv * nd) - (double)vv;   // util.py:15
            carry = delta + (gl * nd) * carry;                          // util.py:16-17
            const float a = (float)carry;
            sr[pw] = a;
            sv[pw] = a + vv;                                            // util.py:21
            v_next = vv;
        }
    }
}

// Streamed once: non-temporal loads / stores keep the tile traffic from displacing useful lines.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_load4(const float4* p) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float4* p, float4 v) {
    f32x4_t t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4_t*>(p));
}

// EPB envs per workgroup, THREADS >= EPB threads (all of them stream the tile in and out, the first EPB
// walk one segment each).  Requires 16-byte aligned r/v/out and 4-byte aligned dones (checked by the
// host wrapper).
template <bool NSTEP, int PROMO, int EPB, int THREADS>
__global__ __launch_bounds__(THREADS) void scan_lds_kernel(
    const float* __restrict__ r, const float* __restrict__ v, const uint8_t* __restrict__ d,
    const float* __restrict__ lv, double gamma, double gl, int64_t n_env, int T,
    int skew_mask, int cap, float* __restrict__ out0, float* __restrict__ out1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sr = reinterpret_cast<float*>(smem);
    float* sv = sr + cap;
    uint8_t* sd = reinterpret_cast<uint8_t*>(sv + cap);
    const int tid = threadIdx.x;
    const int64_t n_tiles = (n_env + EPB - 1) / EPB;

    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t e0 = tile * EPB;
        const int n_here = (int)((n_env - e0) < EPB ? (n_env - e0) : EPB);
        const int elems = n_here * T;
        const int64_t base = e0 * T;
        const float4* gr4 = reinterpret_cast<const float4*>(r + base);
        const float4* gv4 = reinterpret_cast<const float4*>(v + base);
        const uint32_t* gd4 = reinterpret_cast<const uint32_t*>(d + base);
        const int nq = elems >> 2;
        // prefetch this lane's bootstrap value while the tile streams in
        const float last_v = (tid < n_here) ? lv[e0 + tid] : 0.f;

        for (int q = tid; q < nq; q += THREADS) {
            const float4 a = nt_load4(gr4 + q);
            const float4 b = nt_load4(gv4 + q);
            const uint32_t dd = __builtin_nontemporal_load(gd4 + q);
            const int p = skewed(q << 2, skew_mask);   // 4 elems never straddle a 32-group
            sr[p] = a.x; sr[p + 1] = a.y; sr[p + 2] = a.z; sr[p + 3] = a.w;
            sv[p] = b.x; sv[p + 1] = b.y; sv[p + 2] = b.z; sv[p + 3] = b.w;
            *reinterpret_cast<uint32_t*>(sd + (q << 2)) = dd;
        }
        for (int i = (nq << 2) + tid; i < elems; i += THREADS) {
            const int p = skewed(i, skew_mask);
            sr[p] = r[base + i];
            sv[p] = v[base + i];
            sd[i] = d[base + i];
        }
        __syncthreads();

        if (tid < n_here)
            lane_scan_lds<NSTEP, PROMO>(sr, sv, sd, tid, T, skew_mask, last_v, gamma, gl);
        __syncthreads();

        float4* go0 = reinterpret_cast<float4*>(out0 + base);
        float4* go1 = reinterpret_cast<float4*>(out1 + base);
        for (int q = tid; q < nq; q += THREADS) {
            const int p = skewed(q << 2, skew_mask);
            nt_store4(go0 + q, make_float4(sr[p], sr[p + 1], sr[p + 2], sr[p + 3]));
            nt_store4(go1 + q, make_float4(sv[p], sv[p + 1], sv[p + 2], sv[p + 3]));
        }
        for (int i = (nq << 2) + tid; i < elems; i += THREADS) {
            const int p = skewed(i, skew_mask);
            out0[base + i] = sr[p];
            out1[base + i] = sv[p];
        }
        __syncthreads();   // tile reuse
    }
}

// Long horizons (T % 4 == 0): the tile is EPB envs x TC steps, and a workgroup walks its envs' segments
// backwards chunk by chunk, the scan state staying in the lanes' registers.  LDS per workgroup stays
// ~20 KB whatever T is (whole segments of T = 128 fit only ~128 per CU, and the lane-sequential f64
// chain then cannot hide its latency); every chunk row is one 4 TC-byte run, read and written with
// 16-byte accesses.
template <bool NSTEP, int PROMO, int EPB, int THREADS, int TC>
__global__ __launch_bounds__(THREADS) void scan_lds_chunked_kernel(
    const float* __restrict__ r, const float* __restrict__ v, const uint8_t* __restrict__ d,
    const float* __restrict__ lv, double gamma, double gl, int64_t n_env, int T,
    float* __restrict__ out0, float* __restrict__ out1) {
    constexpr int RAW = EPB * TC, CAP = ((RAW + (RAW >> 5) + 4) + 3) & ~3, QPR = TC / 4;
    __shared__ __attribute__((aligned(16))) float sr[CAP];
    __shared__ __attribute__((aligned(16))) float sv[CAP];
    __shared__ __attribute__((aligned(16))) uint8_t sd[RAW];
    static_assert(TC % 4 == 0 && THREADS >= EPB, "tile shape");
    const int tid = threadIdx.x;
    const int skew_mask = ~0;                         // TC is even
    const int64_t e0 = (int64_t)blockIdx.x * EPB;
    const int n_here = (int)((n_env - e0) < EPB ? (n_env - e0) : EPB);
    double carry = 0.0;
    float run32 = (tid < n_here) ? lv[e0 + tid] : 0.f, v_next = run32;
    if (NSTEP && PROMO != ARL_PROMO_NEP50) carry = (double)run32;
    for (int hi = T; hi > 0; hi -= TC) {              // chunk = steps [lo, hi)
        const int lo = hi - TC > 0 ? hi - TC : 0, len = hi - lo, qpr = len >> 2, nq = n_here * qpr;
        for (int q = tid; q < nq; q += THREADS) {
            const int env = q / qpr, j = q - env * qpr;
            const int64_t g = (e0 + env) * T + lo + 4 * j;
            const float4 a = nt_load4(reinterpret_cast<const float4*>(r + g));
            const float4 b = nt_load4(reinterpret_cast<const float4*>(v + g));
            const uint32_t dd = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(d + g));
            const int i = env * len + 4 * j, p = skewed(i, skew_mask);
            sr[p] = a.x; sr[p + 1] = a.y; sr[p + 2] = a.z; sr[p + 3] = a.w;
            sv[p] = b.x; sv[p + 1] = b.y; sv[p + 2] = b.z; sv[p + 3] = b.w;
            *reinterpret_cast<uint32_t*>(sd + i) = dd;
        }
        __syncthreads();
        if (tid < n_here)
            lane_scan_chunk<NSTEP, PROMO>(sr, sv, sd, tid, len, skew_mask, gamma, gl, carry, run32, v_next);
        __syncthreads();
        for (int q = tid; q < nq; q += THREADS) {
            const int env = q / qpr, j = q - env * qpr;
            const int64_t g = (e0 + env) * T + lo + 4 * j;
            const int p = skewed(env * len + 4 * j, skew_mask);
            nt_store4(reinterpret_cast<float4*>(out0 + g), make_float4(sr[p], sr[p + 1], sr[p + 2], sr[p + 3]));
            nt_store4(reinterpret_cast<float4*>(out1 + g), make_float4(sv[p], sv[p + 1], sv[p + 2], sv[p + 3]));
        }
        __syncthreads();                              // the tile is reused by the next chunk
    }
}

// Any horizon, any alignment: one lane per env straight on global memory.
template <bool NSTEP, int PROMO>
__global__ __launch_bounds__(256) void scan_direct_kernel(
    const float* __restrict__ r, const float* __restrict__ v, const uint8_t* __restrict__ d,
    const float* __restrict__ lv, double gamma, double gl, int64_t n_env, int T,
    float* __restrict__ out0, float* __restrict__ out1) {
    const float g32 = (float)gamma;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_env;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = e * T;
        if (NSTEP) {
            if (PROMO == ARL_PROMO_NEP50) {
                float run = lv[e];
                for (int t = T - 1; t >= 0; --t) {
                    const float rr = r[base + t];
                    run = d[base + t] ? rr : (run * g32) + rr;
                    out0[base + t] = run;
                    out1[base + t] = run - v[base + t];
                }
            } else {
                double run = (double)lv[e];
                for (int t = T - 1; t >= 0; --t) {
                    const float rr = r[base + t];
                    run = d[base + t] ? (double)rr : (run * gamma) + (double)rr;
                    const float ret = (float)run;
                    out0[base + t] = ret;
                    out1[base + t] = ret - v[base + t];
                }
            }
        } else {
            double carry = 0.0;
            float v_next = lv[e];
            for (int t = T - 1; t >= 0; --t) {
                const float rr = r[base + t], vv = v[base + t];
                const double nd = d[base + t] ? 0.0 : 1.0;
                const double gv = (PROMO == ARL_PROMO_NEP50) ? (double)(g32 * v_next)
                                                             : gamma * (double)v_next;
                const double delta = ((double)rr + gv * nd) - (double)vv;
                carry = delta + (gl * nd) * carry;
                const float a = (float)carry;
                out0[base + t] = a;
                out1[base + t] = a + vv;
                v_next = vv;
            }
        }
    }
}

// ARL_PROMO_ASSOC: the recurrence as a wavefront suffix scan.  Both scans are chains of affine maps
//   GAE     carry_t = delta_t + (gamma lambda nd_t) carry_{t+1},  delta_t = r_t + gamma v_{t+1} nd_t - v_t   (util.py:13-17)
//   n-step  run_t   = r_t     + (gamma nd_t)        run_{t+1}                                             (util.py:29-35)
// and affine maps compose associatively: (c1, d1) o (c2, d2) = (c1 c2, d1 + c1 d2).  A segment of T steps sits on
// SEG lanes of a wave (E consecutive steps per lane, SEG * E >= T; 64 / SEG segments per wave): every lane composes
// its own steps, a log2(SEG)-step shuffle scan composes the lanes' maps from the right, and each lane re-walks its
// steps from the carry that enters it.  All in f64 with the LEGACY promotion's operand types, so the result differs
// from the exact walks only by the reassociation of f64 sums (<= 1e-5 asserted in the tests, usually bit-identical
// after the cast to f32).  T steps cost T / (64 E) x (E + log2 SEG) dependent f64 steps per wave instead of T per
// lane, and the loads are 4 E contiguous bytes per lane straight from HBM (no LDS staging, no transposition).
// NCH: segment groups per wave.  A wave owns NCH consecutive groups of 64 E steps and issues the loads of ALL of them
// before it scans the first, so the later groups' loads fly under the earlier groups' f64 work (one group per wave = a
// wave per 4 KB, 262 144 waves at 2^26 elements and T = 128).  Measured at 2^26 elements (tools/wave_scan_probe.py):
// T = 128 0.686 (1) / 0.687 (2) / 0.700 (4) of the HBM peak, T = 256 0.681 / 0.693 / 0.693 -- neither the wave count
// nor the bytes in flight per wave is what holds this kernel at 0.70 (38-82 registers: 6-8 waves per SIMD either way).
template <bool NSTEP, int E, int NCH>
__global__ __launch_bounds__(256) void scan_wave_kernel(
    const float* __restrict__ r, const float* __restrict__ v, const uint8_t* __restrict__ d,
    const float* __restrict__ lv, double gamma, double gl, int64_t n_env, int T, int seg, int vec_ok,
    float* __restrict__ out0, float* __restrict__ out1) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6) * NCH;
    const int sl = lane & (seg - 1);
    const int t0 = sl * E;
    float rr[NCH][E], vv[NCH][E], last_v[NCH];
    uint8_t dd[NCH][E];
    int64_t base[NCH];
    bool live[NCH], vec[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int64_t env = (wave0 + c) * (64 / seg) + lane / seg;
        live[c] = env < n_env;
        base[c] = (live[c] ? env : 0) * (int64_t)T;
        // vector form: the lane's E steps lie inside the segment and start on an E-float boundary (T % E == 0)
        vec[c] = vec_ok && live[c] && t0 < T;
        const int64_t bs = base[c];
        if (E >= 4 && vec[c]) {
#pragma unroll
            for (int q = 0; q < E / 4; ++q) {
                const float4 a = nt_load4(reinterpret_cast<const float4*>(r + bs + t0 + 4 * q));
                const float4 b = nt_load4(reinterpret_cast<const float4*>(v + bs + t0 + 4 * q));
                const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(d + bs + t0 + 4 * q));
                rr[c][(4 * q) % E] = a.x; rr[c][(4 * q + 1) % E] = a.y; rr[c][(4 * q + 2) % E] = a.z; rr[c][(4 * q + 3) % E] = a.w;
                vv[c][(4 * q) % E] = b.x; vv[c][(4 * q + 1) % E] = b.y; vv[c][(4 * q + 2) % E] = b.z; vv[c][(4 * q + 3) % E] = b.w;
#pragma unroll
                for (int k = 0; k < 4; ++k) dd[c][(4 * q + k) % E] = (uint8_t)(w >> (8 * k));
            }
        } else if (E == 2 && vec[c]) {
            const float2 a = *reinterpret_cast<const float2*>(r + bs + t0);
            const float2 b = *reinterpret_cast<const float2*>(v + bs + t0);
            const uint16_t w = *reinterpret_cast<const uint16_t*>(d + bs + t0);
            rr[c][0] = a.x; rr[c][1 % E] = a.y; vv[c][0] = b.x; vv[c][1 % E] = b.y;
            dd[c][0] = (uint8_t)w; dd[c][1 % E] = (uint8_t)(w >> 8);
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k) {
                const bool in = live[c] && t0 + k < T;
                rr[c][k] = in ? r[bs + t0 + k] : 0.f;
                vv[c][k] = in ? v[bs + t0 + k] : 0.f;
                dd[c][k] = in ? d[bs + t0 + k] : (uint8_t)0;
            }
        }
        last_v[c] = live[c] ? lv[env] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        // value that follows the lane's last step: the next lane's first value, the bootstrap value after step T - 1
        float v_after = __shfl_down(vv[c][0], 1, 64);
        if (t0 + E >= T) v_after = last_v[c];
        // ---- the lane's own steps as one affine map x -> D + C x (x = what enters after its last step)
        double cc[E], dl[E];
        double C = 1.0, D = 0.0;
#pragma unroll
        for (int k = E - 1; k >= 0; --k) {
            const bool in = t0 + k < T;
            const double nd = dd[c][k] ? 0.0 : 1.0;                             // util.py:8
            const float vn = (t0 + k + 1 >= T) ? last_v[c] : (k == E - 1) ? v_after : vv[c][(k + 1) % E];
            if (NSTEP) { cc[k] = gamma * nd; dl[k] = (double)rr[c][k]; }
            else { cc[k] = gl * nd; dl[k] = ((double)rr[c][k] + (gamma * (double)vn) * nd) - (double)vv[c][k]; }
            if (!in) { cc[k] = 1.0; dl[k] = 0.0; }                              // identity beyond the segment
            D = dl[k] + cc[k] * D;
            C = cc[k] * C;
        }
        // ---- suffix scan over the segment's lanes: (C, D) <- (C, D) o (C, D)[lane + off]
        for (int off = 1; off < seg; off <<= 1) {
            const double Co = __shfl_down(C, off, 64), Do = __shfl_down(D, off, 64);
            if (sl + off < seg) { D = D + C * Do; C = C * Co; }
        }
        // what enters this lane = the map of all lanes to its right applied to the end value
        const double x_end = NSTEP ? (double)last_v[c] : 0.0;
        const double Cn = __shfl_down(C, 1, 64), Dn = __shfl_down(D, 1, 64);
        double x = (sl + 1 < seg) ? Dn + Cn * x_end : x_end;
        float o0[E], o1[E];
#pragma unroll
        for (int k = E - 1; k >= 0; --k) {
            x = dl[k] + cc[k] * x;
            const float a = (float)x;
            o0[k] = a;
            o1[k] = NSTEP ? a - vv[c][k] : a + vv[c][k];                        // aac_base.py:121 / util.py:21
        }
        if (!live[c]) continue;
        const int64_t bs = base[c];
        if (E >= 4 && vec[c]) {
#pragma unroll
            for (int q = 0; q < E / 4; ++q) {
                nt_store4(reinterpret_cast<float4*>(out0 + bs + t0 + 4 * q),
                          make_float4(o0[(4 * q) % E], o0[(4 * q + 1) % E], o0[(4 * q + 2) % E], o0[(4 * q + 3) % E]));
                nt_store4(reinterpret_cast<float4*>(out1 + bs + t0 + 4 * q),
                          make_float4(o1[(4 * q) % E], o1[(4 * q + 1) % E], o1[(4 * q + 2) % E], o1[(4 * q + 3) % E]));
            }
        } else if (E == 2 && vec[c]) {
            *reinterpret_cast<float2*>(out0 + bs + t0) = make_float2(o0[0], o0[1 % E]);
            *reinterpret_cast<float2*>(out1 + bs + t0) = make_float2(o1[0], o1[1 % E]);
        } else {
#pragma unroll
            for (int k = 0; k < E; ++k)
                if (t0 + k < T) { out0[bs + t0 + k] = o0[k]; out1[bs + t0 + k] = o1[k]; }
        }
    }
}

bool g_force_wave = false;
int g_wave_nch = 0;             // arl_dev_scan_wave_groups: groups per wave of the wave scan (0 = chosen by size)      // arl_dev_scan_force_wave: tests run the wave scan at every horizon <= 512

// E steps per lane (1, 2, 4 or 8) x SEG lanes per segment (a power of two <= 64) for horizons T <= 512;
// returns -1 beyond that (the caller falls back to the exact walk)
template <bool NSTEP>
int launch_wave(const float* r, const float* v, const uint8_t* d, const float* lv, double gamma, double gl,
                int64_t n_env, int T, float* o0, float* o1, hipStream_t s) {
    // Below ~100 steps the exact LDS-tile walk is the faster kernel (T = 32: 0.76 vs 0.71 of the HBM peak, T = 64: 0.75
    // vs 0.71; T = 128: 0.66 vs 0.70, T = 256: 0.68 vs 0.73): the tolerance mode then simply runs it.
    if (T > 512 || (T < 96 && !g_force_wave)) return -1;
    const bool al = arl::aligned16(r) && arl::aligned16(v) && arl::aligned16(o0) && arl::aligned16(o1) && arl::aligned4(d);
    int e = T > 256 ? 8 : T > 128 ? 4 : T > 64 ? 2 : 1;            // fewest steps per lane that fit 64 lanes
    // more steps per lane where that makes the accesses 16 (8) bytes wide and still leaves >= 8 lanes per segment
    // (measured at 2^26 elements, fraction of the HBM peak -- T = 32: E = 1 0.33, 2 0.62, 4 0.71, 8 0.70; T = 128: 2 0.57,
    //  4 0.70, 8 0.66; T = 256: 4 0.73, 8 0.69)
    if (al && e < 4 && T % 4 == 0 && T / 4 >= 8) e = 4;
    else if (al && e < 2 && T % 2 == 0 && T / 2 >= 8) e = 2;
    const int vec_ok = al && T % e == 0 && e > 1;
    int seg = 1;
    while (seg * e < T) seg <<= 1;
    const int64_t groups = (n_env + (64 / seg) - 1) / (64 / seg);
    // groups per wave (ARL_SCAN_NCH overrides: measurement): one while the launch would not fill the chip otherwise
    int nch = g_wave_nch > 0 ? g_wave_nch : (groups >= 8 * 4096 ? 4 : groups >= 2 * 4096 ? 2 : 1);
    if (e == 8 && !g_wave_nch) nch = 1;                             // (T > 256: 0.64 vs 0.62 of the HBM peak with 2)
    if (e == 8 && nch > 2) nch = 2;                                 // (registers)
    const int64_t waves = (groups + nch - 1) / nch;
    const unsigned grid = (unsigned)((waves + 3) / 4);
#define ARL_WAVE_SCAN_N(E_, N_) hipLaunchKernelGGL((scan_wave_kernel<NSTEP, E_, N_>), dim3(grid), dim3(256), 0, s, r, v, d, lv, gamma, gl, n_env, T, seg, vec_ok, o0, o1)
#define ARL_WAVE_SCAN(E_) do { if (nch >= 4) ARL_WAVE_SCAN_N(E_, 4); else if (nch == 2) ARL_WAVE_SCAN_N(E_, 2); else ARL_WAVE_SCAN_N(E_, 1); } while (0)
    if (e == 1) ARL_WAVE_SCAN(1);
    else if (e == 2) ARL_WAVE_SCAN(2);
    else if (e == 4) ARL_WAVE_SCAN(4);
    else { if (nch == 2) ARL_WAVE_SCAN_N(8, 2); else ARL_WAVE_SCAN_N(8, 1); }
#undef ARL_WAVE_SCAN
#undef ARL_WAVE_SCAN_N
    return arl::check_launch("scan_wave_kernel");
}

// valids[e,t] = (t <= first set flag); zero adv/ret/value past it.
__global__ __launch_bounds__(256) void valids_kernel(
    const uint8_t* __restrict__ flags, int64_t n_env, int T, int8_t* __restrict__ valids,
    float* __restrict__ adv, float* __restrict__ ret, float* __restrict__ val) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_env;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t base = e * T;
        bool ok = true;                       // util.py:58-63
        for (int t = 0; t < T; ++t) {
            valids[base + t] = ok ? 1 : 0;
            if (!ok) {                        // util.py:44-46
                if (adv) adv[base + t] = 0.f;
                if (ret) ret[base + t] = 0.f;
                if (val) val[base + t] = 0.f;
            }
            if (flags[base + t]) ok = false;
        }
    }
}

template <bool NSTEP, int PROMO, int EPB, int THREADS = EPB>
int launch_lds(const float* r, const float* v, const uint8_t* d, const float* lv, double gamma,
               double gl, int64_t n_env, int T, float* o0, float* o1, hipStream_t s) {
    const int skew_mask = (T & 1) ? 0 : ~0;
    const int raw = EPB * T;
    const int cap = ((raw + (raw >> 5) + 4) + 3) & ~3;       // floats, 16-B multiple
    const size_t lds = (size_t)cap * 8 + (size_t)raw + 16;
    const int64_t tiles = (n_env + EPB - 1) / EPB;
    // one tile per workgroup (the loop in the kernel only matters beyond 2^30 tiles): a persistent grid
    // serialises load -> scan -> store inside each workgroup and ends ragged (5.6 vs 6.25 TB/s at 2^26 elements)
    const unsigned grid = (unsigned)(tiles < ((int64_t)1 << 30) ? tiles : ((int64_t)1 << 30));
    hipLaunchKernelGGL((scan_lds_kernel<NSTEP, PROMO, EPB, THREADS>), dim3(grid), dim3(THREADS), lds, s, r, v,
                       d, lv, gamma, gl, n_env, T, skew_mask, cap, o0, o1);
    return arl::check_launch("scan_lds_kernel");
}

template <bool NSTEP, int PROMO>
int dispatch(const float* r, const float* v, const uint8_t* d, const float* lv, double gamma,
             double gl, int64_t n_env, int T, float* o0, float* o1, hipStream_t s) {
    const bool vec_ok = arl::aligned16(r) && arl::aligned16(v) && arl::aligned16(o0) &&
                        arl::aligned16(o1) && arl::aligned4(d);
    // Tile = EPB envs.  LDS per tile ~ 9.3 * EPB * T bytes: cap it near 20 KB so 8
    // workgroups stay resident per CU; for small batches prefer narrower tiles so the
    // launch still spreads over many CUs (a 256-env batch = 4 x 64-env tiles).
    // Longer horizons keep the ~40 KB budget with fewer envs per tile; the tile is still streamed by 256
    // threads (a 64-thread workgroup left the load / store phases with one wave).
    if (vec_ok && T <= 17) {
        int epb = (T <= 8) ? 256 : 128;
        while (epb > 64 && (n_env + epb - 1) / epb < 512) epb >>= 1;
        if (epb == 256) return launch_lds<NSTEP, PROMO, 256>(r, v, d, lv, gamma, gl, n_env, T, o0, o1, s);
        if (epb == 128) return launch_lds<NSTEP, PROMO, 128>(r, v, d, lv, gamma, gl, n_env, T, o0, o1, s);
        return launch_lds<NSTEP, PROMO, 64>(r, v, d, lv, gamma, gl, n_env, T, o0, o1, s);
    }
    if (vec_ok && T <= 34) return launch_lds<NSTEP, PROMO, 64, 256>(r, v, d, lv, gamma, gl, n_env, T, o0, o1, s);
    if (vec_ok && (T & 3) == 0) {                     // any longer horizon that is a multiple of 4
        // 32 envs x 64-step chunks (measured at T = 128: 128x16 0.28, 64x32 0.58, 64x64 0.67, 32x64 0.73 of the HBM
        // peak; whole segments, 32 x 128, 0.51): rows of 256 B keep the accesses wide, ~20 KB keeps 8 tiles resident
        hipLaunchKernelGGL((scan_lds_chunked_kernel<NSTEP, PROMO, 32, 256, 64>), dim3((unsigned)((n_env + 31) / 32)),
                           dim3(256), 0, s, r, v, d, lv, gamma, gl, n_env, T, o0, o1);
        return arl::check_launch("scan_lds_chunked_kernel");
    }
    if (vec_ok && T <= 136) return launch_lds<NSTEP, PROMO, 32, 256>(r, v, d, lv, gamma, gl, n_env, T, o0, o1, s);
    if (vec_ok && T <= 544) return launch_lds<NSTEP, PROMO, 8, 256>(r, v, d, lv, gamma, gl, n_env, T, o0, o1, s);
    hipLaunchKernelGGL((scan_direct_kernel<NSTEP, PROMO>), dim3(arl::stream_grid(n_env, 256)),
                       dim3(256), 0, s, r, v, d, lv, gamma, gl, n_env, T, o0, o1);
    return arl::check_launch("scan_direct_kernel");
}

int check_scan_args(const void* a, const void* b, const void* c, const void* d, const void* e,
                    const void* f, int64_t n_env, int32_t T, int32_t promo) {
    if (!a || !b || !c || !d || !e || !f) { arl::set_error("scan: null pointer"); return ARL_E_ARG; }
    if (n_env < 0 || T <= 0) { arl::set_error("scan: bad n_env/horizon"); return ARL_E_ARG; }
    if (n_env * (int64_t)T > ((int64_t)1 << 40)) { arl::set_error("scan: too large"); return ARL_E_RANGE; }
    if (promo != ARL_PROMO_NEP50 && promo != ARL_PROMO_LEGACY && promo != ARL_PROMO_ASSOC) { arl::set_error("scan: bad promo"); return ARL_E_ARG; }
    return 0;
}

}  // namespace

extern "C" void arl_dev_scan_force_wave(int32_t on) { g_force_wave = on != 0; }
extern "C" void arl_dev_scan_wave_groups(int32_t n) { g_wave_nch = (n == 1 || n == 2 || n == 4) ? n : 0; }

extern "C" int arl_gae_scan(const float* rewards, const float* values, const uint8_t* dones,
                            const float* last_values, double discount, double gae_lambda,
                            int64_t n_env, int32_t horizon, int32_t promo, float* advantages,
                            float* returns, void* stream) {
    int rc = check_scan_args(rewards, values, dones, last_values, advantages, returns, n_env, horizon, promo);
    if (rc) return rc;
    if (n_env == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const double gl = discount * gae_lambda;          // util.py:17, python floats
    if (promo == ARL_PROMO_ASSOC) {
        const int rc2 = launch_wave<false>(rewards, values, dones, last_values, discount, gl, n_env, horizon, advantages, returns, s);
        if (rc2 >= 0) return rc2;
        promo = ARL_PROMO_LEGACY;                     // horizon beyond the wave scan: the exact walk with the same operand types
    }
    if (promo == ARL_PROMO_NEP50)
        return dispatch<false, ARL_PROMO_NEP50>(rewards, values, dones, last_values, discount, gl, n_env, horizon, advantages, returns, s);
    return dispatch<false, ARL_PROMO_LEGACY>(rewards, values, dones, last_values, discount, gl, n_env, horizon, advantages, returns, s);
}

extern "C" int arl_nstep_return(const float* rewards, const uint8_t* dones, const float* values,
                                const float* last_values, double discount, int64_t n_env,
                                int32_t horizon, int32_t promo, float* returns, float* advantages,
                                void* stream) {
    int rc = check_scan_args(rewards, values, dones, last_values, advantages, returns, n_env, horizon, promo);
    if (rc) return rc;
    if (n_env == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (promo == ARL_PROMO_ASSOC) {
        const int rc2 = launch_wave<true>(rewards, values, dones, last_values, discount, 0.0, n_env, horizon, returns, advantages, s);
        if (rc2 >= 0) return rc2;
        promo = ARL_PROMO_LEGACY;
    }
    if (promo == ARL_PROMO_NEP50)
        return dispatch<true, ARL_PROMO_NEP50>(rewards, values, dones, last_values, discount, 0.0, n_env, horizon, returns, advantages, s);
    return dispatch<true, ARL_PROMO_LEGACY>(rewards, values, dones, last_values, discount, 0.0, n_env, horizon, returns, advantages, s);
}

extern "C" int arl_valids_mask(const uint8_t* reset_flags, int64_t n_env, int32_t horizon,
                               int8_t* valids, float* advantages, float* returns, float* values,
                               void* stream) {
    ARL_REQUIRE(reset_flags && valids, ARL_E_ARG, "null pointer");
    ARL_REQUIRE(n_env >= 0 && horizon > 0, ARL_E_ARG, "bad n_env/horizon");
    if (n_env == 0) return 0;
    hipLaunchKernelGGL(valids_kernel, dim3(arl::stream_grid(n_env, 256)), dim3(256), 0,
                       (hipStream_t)stream, reset_flags, n_env, (int)horizon, valids, advantages,
                       returns, values);
    return arl::check_launch("valids_kernel");
}
