/* lft_hip.h -- C ABI of liblft_hip.so: the LFT forward hot path on MI355X (gfx950).
 *
 * This library replaces, for CUDA/HIP tensors, the arithmetic of the reference's
 *   model/LFT.py:52-83   get_model.forward  (and everything it calls: :86-115, :118-191, :194-238, :255-266)
 * behind the reference's own plugin surface (model.LFT.get_model(args).forward(lr)); the Python module
 * lft_amd/module.py binds these entry points with ctypes.  INTEGRATION.md shows the binding a
 * reference maintainer would add.
 *
 * Conventions
 *  - extern "C", plain pointers and ints; no torch types.  All pointers are DEVICE pointers unless noted.
 *  - The library never allocates, frees or retains device memory: the caller owns every buffer
 *    (packed weights, workspace, inputs, outputs) and sizes them with the *_bytes() queries.
 *  - Every call only enqueues work on `stream` (a hipStream_t passed as void*) and returns without
 *    synchronising; calls are graph-capturable.
 *  - Return value: 0 ok; <0 argument error (LFT_ERR_*); >0 a hipError_t.  lft_last_error() gives a
 *    thread-local message.  Nothing is thrown across the boundary.
 *  - prec selects the MFMA operand type: LFT_PREC_F32 = exact fp32 (v_mfma_f32_32x32x2_f32),
 *    LFT_PREC_BF16 = bf16 operands / fp32 accumulate (v_mfma_f32_32x32x16_bf16), LFT_PREC_F16 = IEEE half operands /
 *    fp32 accumulate (v_mfma_f32_32x32x16_f16: same kernels, layouts and speed as bf16, 11 significant bits instead
 *    of 8, but a range of 65504 -- activations beyond it become inf; every such event reaches a LayerNorm or the output as
 *    inf / NaN, where the kernels set a sticky flag in the workspace that lft_status_read reports: an overflow is a loud error,
 *    not a silently wrong image).
 *    Activations between kernels are stored in the same type (float, __bf16 or _Float16, channels-last [B, A*A, h, w, C]).
 *  - Shapes: A = angRes (A*A <= 128 views; 5x5 and 9x9 are the tested ones), h x w = LR view size, s = scale factor (2 or 4),
 *    channels fixed to 64 (reference option.py --channels default, LFT.py:11).
 */
#ifndef LFT_HIP_H
#define LFT_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LFT_ABI_VERSION 5
#define LFT_PREC_F32 0
#define LFT_PREC_BF16 1
#define LFT_PREC_F16 2
#define LFT_NUM_PARAMS 78
/* GEMM arithmetic of the training step: exact fp32 MFMA, or fp32 operands split into bf16 hi + lo with three bf16
 * MFMAs per product (~2^-16 relative per product, 5x the matrix-pipe rate).  Forward and backward of one step must
 * use the same mode (the packed weights in the tape are in the mode's format). */
#define LFT_MATH_F32 0
#define LFT_MATH_BF16X3 1
#define LFT_MATH_BF16X6 2   /* fp32 operands as THREE bf16 numbers (x = a + b + c exactly), six bf16 MFMAs per product (the three smallest of the
                             * nine terms dropped: <= 2^-23 relative, fp32-class) at 6/16 of the fp32-MFMA cost; every GEMM of the step */

#define LFT_ERR_ARG (-1)         /* null pointer / bad enum */
#define LFT_ERR_SHAPE (-2)       /* shape outside what this build supports */
#define LFT_ERR_UNSUPPORTED (-3) /* valid for the reference, not implemented here yet */
#define LFT_ERR_CALLBACK (-4)    /* a caller-supplied callback asked to stop (lft_train_backward_buckets) */
/* Returned by lft_status_read (never by the enqueue-only calls): a non-finite activation or output value was seen since the last
 * lft_status_reset -- on the LFT_PREC_F16 path that is what a range overflow (|x| > 65504) turns into. */
#define LFT_STATUS_NONFINITE 1001

int lft_version(void);
const char* lft_last_error(void);

/* Size of the packed-weight buffer for a model with (A, h, w, s) and of the per-forward workspace. */
int lft_packed_bytes(int A, int h, int w, int s, int prec, size_t* out_bytes);
int lft_workspace_bytes(int B, int A, int h, int w, int s, int prec, size_t* out_bytes);

/* Re-arrange the reference's 78 fp32 parameter tensors (device pointers, in state_dict order -- the
 * order of lft_amd/params.py:param_table, which is the reference's registration order, LFT.py:23-44,
 * :125-145, :199-214) into MFMA-fragment streams, and precompute the input-independent tables
 * (angular / spatial sinusoids LFT.py:86-115, and the embedded spatial position tokens LFT.py:180).
 * `params` is a HOST array of 78 device pointers.  Must be re-run whenever a weight, h or w changes. */
int lft_pack_weights(const float* const* params, int nparams, void* packed,
                     int A, int h, int w, int s, int prec, void* stream);

/* get_model.forward (reference LFT.py:52-83).  lr: fp32 [B,1,A*h,A*w]; out: fp32 [B,1,A*h*s,A*w*s]. */
int lft_forward(const void* packed, const float* lr, float* out, void* workspace,
                int B, int A, int h, int w, int s, int prec, void* stream);

/* Sticky status word of a workspace (its last 256 bytes).  The kernels of lft_forward / the per-stage entry points that take a
 * workspace set bit 0 when a token's LayerNorm variance or an output pixel is not finite (inf / NaN): an fp16 range overflow
 * anywhere in the network, or non-finite input data.  Nothing clears it but lft_status_reset, so one read covers every
 * forward since the last reset -- also graph replays, which the host never sees individually.
 *   lft_status_reset: enqueue the clearing of the word on `stream` (call once after allocating a workspace, and after a read
 *                     that reported something).
 *   lft_status_read : copy the word to *host_flags (may be NULL), SYNCHRONISES `stream`; returns 0 when clear,
 *                     LFT_STATUS_NONFINITE when set (lft_last_error() names the cause), or an error code. */
int lft_status_reset(void* workspace, int B, int A, int h, int w, int s, int prec, void* stream);
int lft_status_read(const void* workspace, int B, int A, int h, int w, int s, int prec, void* stream, unsigned* host_flags);

/* Profiling aid, NOT for the hot path: same as lft_forward but records a HIP event on `stream` after every
 * kernel, SYNCHRONISES the stream, and returns per-kernel milliseconds (host arrays ms_out / names_out of
 * max_records entries; names are static strings).  bench.py uses it for the roofline of the dominant kernel. */
int lft_forward_profiled(const void* packed, const float* lr, float* out, void* workspace,
                         int B, int A, int h, int w, int s, int prec, void* stream,
                         int max_records, float* ms_out, const char** names_out, int* n_out);

/* Measurement aid for bench.py's roofline: mean milliseconds of ONE kernel of the forward ("k_conv64", "k_ang", "k_spa1",
 * "k_spa_b"), launched `reps` times back to back between two HIP events on `stream` -- no event between the launches,
 * so the figure is comparable with a rocprofv3 kernel trace.  The kernel reads what a previous lft_forward left in
 * `workspace`.  Synchronises `stream`. */
int lft_kernel_time(const char* kernel, const void* packed, void* workspace, int B, int A, int h, int w, int s, int prec,
                    int reps, void* stream, float* ms_out);

/* ---- per-stage entry points (unit tests, profiling).  `act` buffers are channels-last
 * [B, A*A, h, w, 64] in the activation type of `prec` (float or __bf16). ---- */

/* interpolate(), reference LFT.py:255-266: per-view bicubic of the LR mosaic. */
int lft_bicubic_fwd(const float* lr, float* out, int B, int A, int h, int w, int s, void* stream);
/* conv_init0 + conv_init + residual, reference LFT.py:65-66.  Needs the workspace for 3 temporaries. */
int lft_init_features_fwd(const void* packed, const float* lr, void* act_out, void* workspace,
                          int B, int A, int h, int w, int s, int prec, void* stream);
/* AngTrans.forward of layer `layer`, reference LFT.py:225-238. */
int lft_ang_block_fwd(const void* packed, int layer, const void* act_in, void* act_out,
                      int B, int A, int h, int w, int s, int prec, void* stream);
/* SpaTrans.forward of layer `layer`, reference LFT.py:176-191; skip (may be NULL) is added to the
 * output (the global residual of LFT.py:76 when layer == 3). */
int lft_spa_block_fwd(const void* packed, int layer, const void* act_in, const void* skip, void* act_out,
                      void* workspace, int B, int A, int h, int w, int s, int prec, void* stream);
/* upsampling + bicubic skip, reference LFT.py:79-81: act_in [B,V,h,w,64] -> out fp32 [B,1,A*h*s,A*w*s]. */
int lft_upsample_fwd(const void* packed, const void* act_in, const float* lr, float* out, void* workspace,
                     int B, int A, int h, int w, int s, int prec, void* stream);
/* ---- scene tiling around the hot path (reference utils/utils.py:91-157; caller = test.py:83-101) ----
 * lft_scene_counts   : numU, numV of LFdivide for a scene of A x A views of h0 x w0 (utils.py:95-105).
 * lft_scene_divide   : LFdivide -- scene mosaic fp32 [A*h0, A*w0] -> patches fp32 [numU*numV, 1, A*patch, A*patch]
 *                      (mirror-extended by (patch-stride)/2, zero fill beyond), ready to be fed to lft_forward as a batch.
 * lft_scene_integrate: LFintegrate + re-mosaic (test.py:97-101) -- SR patches [numU*numV, 1, A*patch*s, A*patch*s]
 *                      -> SR scene mosaic [A*h0*s, A*w0*s], keeping the central stride*s region of every patch. */
int lft_scene_counts(int h0, int w0, int patch, int stride, int* num_u, int* num_v);
int lft_scene_divide(const float* scene, float* patches, int A, int h0, int w0, int patch, int stride, void* stream);
int lft_scene_integrate(const float* sr_patches, float* sr_scene, int A, int h0, int w0, int patch, int stride, int s, void* stream);

/* ---- fp32 training step (what PyTorch autograd records / replays for reference LFT.py:52-83 under train.py:89-107) ----
 * The 78 parameters are read IN PLACE (device pointers in state_dict order, HOST array), nothing is packed.
 * `tape` (lft_train_tape_bytes) holds every activation the backward pass re-reads plus its scratch; it must stay
 * untouched between lft_train_forward and lft_train_backward of the same step.
 * lft_train_forward : lr [B,1,A*h,A*w] -> out [B,1,A*h*s,A*w*s] (same function as lft_forward, unfused fp32 kernels).
 * lft_train_backward: dout [B,1,A*h*s,A*w*s] -> grads = ONE flat fp32 buffer (lft_train_grad_floats) holding the 78
 *                     parameter gradients back to back in state_dict order, fully overwritten (not accumulated).
 *                     A data-parallel job all-reduces this one buffer.  No gradient flows to lr.
 *                     Every kernel of the pass runs on `stream` (its gradient tensors live in an arena of the tape and are re-used as
 *                     they die; ABI 4 still carried the second stream of round 2 as an ignored argument -- gone in ABI 5).
 * lft_train_tape_offset: float offset of a saved activation inside the tape, for tests ("feat", "ang0.y", "spa2.tok", ...). */
int lft_train_tape_bytes(int B, int A, int h, int w, int s, size_t* out_bytes);
int lft_train_grad_floats(int s, size_t* out_floats);
int lft_train_tape_offset(const char* name, int B, int A, int h, int w, int s, size_t* out_float_offset);
int lft_train_forward(const float* const* params, int nparams, const float* lr, float* out, void* tape,
                      int B, int A, int h, int w, int s, int math, void* stream);
int lft_train_backward(const float* const* params, int nparams, const float* lr, void* tape, const float* dout, float* grads,
                       int B, int A, int h, int w, int s, int math, void* stream);
/* The same pass for data-parallel training (the reference's DP recipe, SURVEY.md section 8e: gradients summed over ranks):
 * the flat gradient buffer is finished in LFT_GRAD_BUCKETS contiguous ranges, in this order --
 *   bucket 0: altblock.2, altblock.3, upsampling   (after the backward of layer 2)
 *   bucket 1: altblock.0, altblock.1               (after layer 0)
 *   bucket 2: conv_init0, conv_init                (end of the pass)
 * -- and on_bucket(user, bucket, first_float, n_floats) is called ON THE HOST, from inside this call, right after the last
 * kernel writing that range has been enqueued on `stream`.  The caller orders a
 * communication stream after `stream` there and starts the bucket's all-reduce, which then runs beside the kernels of the
 * remaining buckets; or, while capturing, ends one graph and begins the next (lft_amd/train.py does the latter).  Nothing
 * enqueued after the callback touches the bucket's range.  The callback returns 0 to continue; any other value stops the pass
 * right there (nothing further is enqueued) and the call returns LFT_ERR_CALLBACK -- e.g. a failed collective or capture.  lft_train_backward is this function without notifications;
 * both produce identical bits.  lft_train_grad_bucket gives the ranges (floats) without running anything. */
#define LFT_GRAD_BUCKETS 3
typedef int (*lft_bucket_fn)(void* user, int bucket, size_t first_float, size_t n_floats);
int lft_train_backward_buckets(const float* const* params, int nparams, const float* lr, void* tape, const float* dout, float* grads,
                               int B, int A, int h, int w, int s, int math, void* stream,
                               lft_bucket_fn on_bucket, void* user);
int lft_train_grad_bucket(int s, int bucket, size_t* first_float, size_t* n_floats);
/* The backward pass of ONE block, for unit tests of the backward kernels (the `_bwd` counterparts of the per-stage forward entry
 * points above).  `tape` must hold a complete lft_train_forward of the same inputs.  The block's incoming
 * gradient d_out and outgoing gradient d_in are caller buffers; only the block's own parameter gradients are written to `grads`
 * (the flat buffer of lft_train_backward; every other range is left untouched).
 *   LFT_BLOCK_UPSAMPLE : d_out = d loss / d output image [B,1,A*h*s,A*w*s]; d_in = gradient of the body features [N,64]
 *                        (reference LFT.py:79-81; gradients of upsampling.0.weight, upsampling.3.weight)
 *   LFT_BLOCK_SPA      : SpaTrans of `layer` (LFT.py:176-191): d_out, d_in [N,64]
 *   LFT_BLOCK_ANG      : AngTrans of `layer` (LFT.py:225-238): d_out, d_in [N,64]
 *   LFT_BLOCK_INIT     : conv_init0 + conv_init + residual (LFT.py:65-66): d_out = gradient of the features [N,64]; d_in unused (may be NULL)
 * N = B*A*A*h*w tokens, channels-last [B, A*A, h, w, 64] fp32. */
#define LFT_BLOCK_UPSAMPLE 0
#define LFT_BLOCK_SPA 1
#define LFT_BLOCK_ANG 2
#define LFT_BLOCK_INIT 3
int lft_train_block_backward(const float* const* params, int nparams, const float* lr, void* tape, int block, int layer,
                             const float* d_out, float* d_in, float* grads,
                             int B, int A, int h, int w, int s, int math, void* stream);
/* Profiling aid, NOT for the hot path (bench.py's `train.roofline`): lft_train_forward + lft_train_backward on ONE stream with a HIP
 * event after every kernel; SYNCHRONISES the stream and returns per-kernel milliseconds in launch order (host arrays of max_records
 * entries, names are static strings; a step has about 450 launches). */
int lft_train_step_profiled(const float* const* params, int nparams, const float* lr, float* out, void* tape, const float* dout, float* grads,
                            int B, int A, int h, int w, int s, int math, void* stream,
                            int max_records, float* ms_out, const char** names_out, int* n_out);
/* get_loss (reference LFT.py:269-277, torch.nn.L1Loss): *loss = mean |sr - hr|; if dsr != NULL also
 * dsr = gscale * sign(sr - hr) (gscale = 1/n for d loss / d sr).  scratch1024: 1024 floats of device scratch. */
int lft_l1_loss(const float* sr, const float* hr, long long n, float* dsr, float gscale, float* loss, float* scratch1024, void* stream);
/* torch.optim.Adam step (train.py:77-83: betas (0.9, 0.999), eps 1e-8, weight_decay = --decay_rate, default 0) on one
 * flat fp32 buffer; step counts from 1; the gradient is multiplied by gscale first (1/world_size after a sum all-reduce),
 * then weight_decay * p is added (torch's L2 form); bias corrections are computed in double, as torch does. */
int lft_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                  int step, float gscale, float weight_decay, void* stream);

/* ---- per-view quality metrics (reference utils/utils.py:56-88 cal_metrics, which calls scikit-image) ----
 * label, out: fp32 mosaics [B,1,A*h,A*w]; psnr, ssim: fp32 [B*A*A] in (b, u, v) order.  PSNR = 10 log10(R^2 / MSE) with
 * R = 1 when min(label view) >= 0 else 2; SSIM = mean over the view minus a 5-pixel border of the Gaussian-window
 * (sigma 1.5, 11x11, sample covariance, K1 0.01, K2 0.03) SSIM map with data range `ssim_range` (2 reproduces the
 * scikit-image releases contemporary with the reference; 1 is the physical range).  fp64 accumulation. */
int lft_view_metrics_scratch_bytes(int B, int A, int h, int w, size_t* out_bytes);
int lft_view_metrics(const float* label, const float* out, int B, int A, int h, int w, float ssim_range, float* psnr, float* ssim,
                     void* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LFT_HIP_H */
