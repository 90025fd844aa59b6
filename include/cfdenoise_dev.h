/* libcfdenoise -- developer / test hooks (NOT part of the drop-in boundary: nothing in the reference corresponds to them).
 * Used by tests/ (kernel-level parity, stage-wise taps), tools/ (micro-benchmarks) and nothing in convofusion_amd's product path.
 * The product boundary is include/cfdenoise.h. */
#ifndef CFDENOISE_DEV_H
#define CFDENOISE_DEV_H
#include "cfdenoise.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Test hook: D[j][i] = sum_k X[i][k] Y[j][k] through the split-pair (fp16 hi/lo, 3 MFMAs per product) kernel.
 * X dev float32 [I][K], Y dev float32 [J][K], out dev float32 [J][I]; K % 32 == 0, I % 4 == 0.
 * tile_cfg: 0 = chosen from the shape; 1 = 128x128 (2-stage), 30 = 128x128 with the asymmetric ring (weights 2 stages,
 * activations 3 stages), 6 = 128x112, 19 = 64x64 (3-stage), 20 = 32x128 (3-stage), any other value = 128x16
 * (csrc/gemm_sp.hpp: launch_gemm). */
int cfd_test_gemm(cfd_handle h, const float* X, const float* Y, float* out, int I, int J, int K, int tile_cfg,
                  void* stream);

/* Test hooks: stop the forward pipeline after tap point `stage` (0 = off; 1 = after the latent embedding;
 * 2+4l / 3+4l / 4+4l / 5+4l = layer l after self-attention / time block 1 / cross-attention / the layer),
 * and read an internal float32 buffer ("x" residual stream [M][512], "temb", "ss", "eps", "sc", "ssc"). */
int cfd_debug_stop_stage(cfd_handle h, int stage);
/* Micro-benchmark: average ms of `iters` launches of the [J x K] x [512 x K]^T residual GEMM (I must be 512). */
int cfd_bench_gemm(cfd_handle h, int I, int J, int K, int tile_cfg, int iters, float* ms_out);
int cfd_debug_read(cfd_handle h, const char* what, float* dst_dev, size_t numel);

#ifdef __cplusplus
}
#endif
#endif
