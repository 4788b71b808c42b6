/* Development hooks of libaccel_rl_hip.so: process-global switches for the parity tests and the measurement tools
 * (tests/, tools/).  NOT part of the drop-in boundary -- include/accel_rl_hip.h, which keeps no state between calls --
 * and not thread-safe: set them from the one thread that drives the device, around the calls they are meant for.     */
#ifndef ACCEL_RL_HIP_DEV_H
#define ACCEL_RL_HIP_DEV_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic (tools/conv_trace.py): while a device buffer of u64[workgroups][8] is set, the forward / data-gradient /
 * weight-gradient kernels record per-workgroup shader-clock timestamps (start, main loop begin, main loop end, end),
 * two 100 MHz wall-clock samples, HW_ID and XCC_ID.  NULL (the default) disables it.                                */
void arl_dev_conv_trace_buffer(void* device_u64_or_null);

/* Tests: route every following conv / dense call to the generic (any channel count / any K) kernels instead of the
 * scalar-addressed fast path, so that both are covered by the parity tests.                                        */
void arl_dev_conv_force_generic(int32_t on);

/* Tests / A-B measurements: tile shape of the split-route forward (and, for 0 / 1, data-gradient) kernels of layers with
 * 33 .. 64 output columns -- 0: 128 x 64 (one wave per 32-row tile, both column halves), 1: 128 x 32 (two column tiles
 * per row tile), 2: 64 x 64 (forward only: both operands through LDS); 6 / 7: the unsplit forward of >= 128-column layers
 * on 64 x 64 tiles always / never (else by how the tiles fill the CUs); every value gives the same results bit for bit.
 * -1 (default) = chosen by the launch's size: the column split while every half tile gets a CU of its own.           */
void arl_dev_fwd_tile(int32_t v);

/* Tests / A-B measurements: 0 = data gradients ignore the k-contiguous weights they are handed (wt_or_null of
 * arl_conv2d_bwd_data / arl_conv2d_bwd_pair) and gather from w as before ABI 4; same results bit for bit.  Default 1. */
void arl_dev_dgrad_wt(int32_t on);

/* A-B measurements: split count from which a fold (arl_fold_many, the dense forward's fold) sums an output with 64
 * threads instead of 16 (csrc/mfma_conv.hip, FOLD_WIDE).  0 = the default (128).  Changes the association of the sums of
 * the items it moves across the threshold (still a fixed order), nothing else.                                       */
void arl_dev_fold_wide_from(int32_t splits);

/* Tests / A-B measurements: bit 0 set = the image-stationary kernels (csrc/img_conv.hip) are not used; the tap-gathering
 * kernels they replace run instead (same results bit for bit).                                                      */
void arl_dev_conv_variant(int32_t v);

/* Tests: with ARL_PROMO_ASSOC run the wave suffix scan at EVERY horizon <= 512 (not only where it is the faster
 * kernel).                                                                                                          */
void arl_dev_scan_force_wave(int32_t on);
/* Tuning / tests: segment groups (64 lanes x E steps each) a wave of the wave suffix scan owns -- 1, 2 or 4; every
 * value gives the same results bit for bit.  0 (default) = chosen by the launch's size.                             */
void arl_dev_scan_wave_groups(int32_t n);

/* Measurement (tools/env_step_bound.sh): timing knock-outs of arl_env_step's kernel, results WRONG -- 1: every env
 * reads raw frame 0 of the bank (the bank reads all hit one 33 KB line set); 2: the three older planes of the stacked
 * observation are not stored; 3: ... nor loaded.  0 (default) = the product kernel.                                 */
void arl_dev_env_variant(int32_t v);

#ifdef __cplusplus
}
#endif
#endif /* ACCEL_RL_HIP_DEV_H */
