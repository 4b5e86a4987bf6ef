/* emadapt.h -- C ABI of libemadapt_hip.so: the MI355X (gfx950) implementation of the
 * EM-Adapt training hot path of xtudbxk/em-adapt-tensorflow.
 *
 * Every entry point is extern "C", takes plain pointers and sizes (no torch / framework
 * types) and -- except e_step / e_step_perm / emadapt_estep_host, which take HOST buffers
 * like the reference's own native entry -- DEVICE pointers plus a HIP stream handle passed
 * as void* (NULL = the default stream).  All device entries are asynchronous on that
 * stream, allocate nothing and never synchronise, so a caller may capture them in a hipGraph.
 * They return EMADAPT_OK (0) or a negative EMADAPT_ERR_* code; emadapt_strerror() names it.
 *
 * Each declaration cites the reference interface (file:line in xtudbxk/em-adapt-tensorflow)
 * that it replaces.  Tensors are NHWC fp32, filters HWIO fp32, exactly as in the reference.
 */
#ifndef EMADAPT_H_
#define EMADAPT_H_

#include <stddef.h>
#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define EMADAPT_OK 0
#define EMADAPT_ERR_INVALID (-1)     /* bad argument (null pointer, bad size, k out of range ...) */
#define EMADAPT_ERR_UNSUPPORTED (-2) /* shape outside what the kernels are built for             */
#define EMADAPT_ERR_HIP (-3)         /* a HIP runtime call failed (see emadapt_last_hip_error)   */
#define EMADAPT_ERR_WORKSPACE (-4)   /* workspace too small                                      */
#define EMADAPT_ERR_NODEVICE (-5)    /* no usable gfx950 device                                  */
#define EMADAPT_ERR_DEVICE_STATUS (-6) /* a kernel raised a bit of the device status word (emadapt_status_poll) */

/* ABI history (emadapt_version() returns the library's EMADAPT_ABI_VERSION; 10000*major + 100*minor + patch):
 *   100  rounds 1-2.
 *   200  round 3 changed the E-step workspace contract WITHOUT a bump (advisor finding, fixed here): for geometries whose
 *        channels do not fit in LDS (65x65x41) emadapt_estep_workspace_bytes() grew by B*C*H*W floats (class-major home of the
 *        non-resident channels) and emadapt_estep() returns EMADAPT_ERR_WORKSPACE below that; map_out is output only and
 *        emadapt_estep_needs_map() is 0 for every supported geometry.  A caller written against 100 that sized its workspace
 *        by the constant it knew must re-query emadapt_estep_workspace_bytes().  Added: emadapt_prof_collect2 (executed
 *        FLOPs), emadapt_conv_*: tensors of 2 GiB and more (batch-sliced launches).
 *   201  round 4, no signature changed: what emadapt_conv_{fwd,dgrad}[_x6p]_workspace_bytes() return for the stream-K launches
 *        (conv3-conv5, fc6, fc7's last rows) is now arrival words + one 256 KB accumulator image per workgroup (64 MB + 1 KB on
 *        256 CUs) instead of up to three [M][N] images (881 MB for fc6): a caller that re-queries per shape needs no change; one
 *        that cached a size per shape from 200 over-allocates, never under-allocates.
 *   202  round 4, added: emadapt_conv_wgrad_bias (+ _fused, _workspace_bytes): filter gradient and bias gradient in one launch.
 *        Also in 201/202 and NOT recorded then (advisor finding, round 4): emadapt_estep_workspace_bytes() grew again for the
 *        geometries whose class images overflow the LDS -- 256 + B*(C+3)*roundup4(H*W)*4 + align16(4*B) bytes (class images,
 *        per-pixel maxima, partial argmax value / class, the images' shifts) instead of 256 + B*C*roundup4(H*W)*4; a caller
 *        that sized the workspace from the old comment gets EMADAPT_ERR_WORKSPACE: re-query.
 *   203  round 5, added: the device status word (emadapt_status_word / emadapt_status_poll, EMADAPT_ERR_DEVICE_STATUS): a
 *        stream-K launch whose bounded wait for another workgroup's partial tile ran out raises EMADAPT_STATUS_STREAMK_TIMEOUT
 *        there (the tile is also poisoned with NaNs, which the fused ReLU / ReLU-gradient gate now PROPAGATE like
 *        tf.nn.relu), and emadapt_sgd_momentum / emadapt_sgd_momentum_payload leave weights, momentum and accumulators
 *        untouched while the word is non-zero.  Added: the gradient-exchange staging entries (emadapt_grad_pack,
 *        emadapt_grad_owner_sum, emadapt_sgd_momentum_payload), emadapt_sgd_accumulate_momentum.
 *   204  round 5, added: emadapt_conv_wgrad_bias_bf16 (+ _fused, _workspace_bytes): bf16 filter gradient and bias gradient in one
 *        launch (and one second-stage launch for both); emadapt_conv_bf16t_pays; emadapt_filter_tiles_bf16_both; tune "bf16t_sk".  No
 *        signature changed, but launch plans did: emadapt_conv_{fwd,dgrad}_bf16 take 3 x 3, 64 -> 64 channels on bf16-stored tensors
 *        to a direct kernel (emadapt_conv_bf16_plan: plan[0] == 4), the forward of 3 x 3, 4 (f32-stored) -> 64 (bf16-stored) channels
 *        to another (5), a 1 x 1 data gradient with at most 32 gathered channels to a third (6); emadapt_conv_wgrad[_bias]_bf16 take
 *        3 x 3, 64 -> 64 channels on bf16-stored tensors to a direct kernel whose workspace is one slab per workgroup (re-query
 *        emadapt_conv_wgrad[_bias]_bf16_workspace_bytes: 37.8 MB for conv1_2 at batch 16 where the split-M plan needed 15 MB).
 *   205  round 5, added: emadapt_maxpool_fwd_gated (the index byte carries the ReluGrad gate of the pooled tensor; the backward entries
 *        decode it, so the model's max-pool gradients no longer read the gate tensor); tune "pool_rows".  The stride-2 max-pool
 *        gradient runs by 2 x 2 pixel quads (same results).  emadapt_sgd_accumulate_momentum_tiles (+ _supported) and
 *        emadapt_sgd_momentum_payload_tiles: the optimizer pass of a layer that also writes the layer's two tiled bf16 filter images; emadapt_sgd_accumulate_momentum_multi: the pass
 *        over several layers' segments in one launch; emadapt_weights_to_bf16_multi: several filters' bf16 images in one launch.
 *   206  round 6, added: emadapt_voc_preprocess (the input pipeline of a batch in one launch), emadapt_host_register / _unregister / emadapt_memcpy_h2d_async; emadapt_status_raise; emadapt_status_collective (data-parallel jobs: the failure travels with the exchanged gradient instead of
 *        being decided from each rank's own word), EMADAPT_STATUS_NAN_GRADIENT. */
#define EMADAPT_ABI_VERSION 206
int emadapt_version(void);
const char *emadapt_strerror(int code);
int emadapt_last_hip_error(void);          /* hipError_t of the last failing HIP call on this thread */
int emadapt_device_count(void);            /* does not initialise a context */

/* Device status word: one sticky 32-bit word per device in which kernels record failures the host cannot see from a launch's
 * return code (every entry is asynchronous).  Bits:
 *   EMADAPT_STATUS_STREAMK_TIMEOUT  a head workgroup of a stream-K launch (conv3-conv5, fc6, fc7's last rows) gave up waiting
 *                                   for the partial tile of the workgroup behind it (bounded spin, ~1.7 s per publisher: only
 *                                   a co-tenant kernel that never ends, or a hung queue, gets there); that tile of the output
 *                                   holds NaNs.
 * While the word is non-zero the optimiser entries (emadapt_sgd_momentum, emadapt_sgd_momentum_payload,
 * emadapt_sgd_accumulate_momentum) are no-ops ON THE DEVICE -- no host synchronisation is needed to keep a poisoned gradient
 * out of the weights.  emadapt_status_poll copies the word to *flags on `stream`, SYNCHRONISES that stream, optionally clears
 * the word, and returns EMADAPT_ERR_DEVICE_STATUS when it was non-zero (EMADAPT_OK when clean): call it where a
 * synchronisation is acceptable (the reference's loop logs every 500 iterations, deeplab.py:273).  emadapt_status_word returns
 * the device address (tests, graph captures).  Replaces nothing in the reference: TF raises from session.run. */
#define EMADAPT_STATUS_STREAMK_TIMEOUT 1u
#define EMADAPT_STATUS_NAN_GRADIENT 2u   /* collective mode: an optimiser entry met a NaN in the summed gradient and left that element alone */
int emadapt_status_word(void **device_ptr);
int emadapt_status_poll(unsigned *flags, int clear, void *stream);
/* Collective mode (ABI 206; process-wide, returns the previous setting; off by default).  In a data-parallel job "my word is raised"
 * must not decide anything a peer decides differently: a rank whose stream-K wait ran out would skip its update while its NaN tile
 * travels through the exchange and the peers, whose words are clean, apply it -- the ranks diverge.  With the mode on the failure
 * TRAVELS WITH THE GRADIENT: while the local word is raised emadapt_sgd_accumulate, emadapt_sgd_accumulate_pack and
 * emadapt_grad_pack write NaN for every element they produce (sums over ranks keep it); emadapt_sgd_momentum,
 * emadapt_sgd_momentum_payload and emadapt_sgd_momentum_payload_tiles no longer look at the local word but leave every element
 * whose summed gradient is NaN alone (weight and momentum kept, accumulator zeroed) and raise EMADAPT_STATUS_NAN_GRADIENT on their
 * own device.  All ranks read the same summed gradient, so all skip the same elements and all report the failure at their next
 * emadapt_status_poll.  The entries without an exchange (emadapt_sgd_accumulate_momentum*) are not affected. */
int emadapt_status_collective(int on);
/* Raise bits of the word from the host, in stream order (ABI 206): a caller that found a failure of its own (a non-finite loss, a
 * failed read) and wants the optimiser entries -- and, in collective mode, its peers -- to see it; the tests. */
int emadapt_status_raise(unsigned bits, void *stream);

/* ------------------------------------------------------------------------------------------
 * E-step (weak-label assignment)
 * ---------------------------------------------------------------------------------------- */

/* Drop-in for the reference's native entry
 *   extern "C" float *e_step(float*,const int*,const int*,const int*,bool,int,float,float,float)
 *   -- estep/weaklabel.cpp:6-7, bound by estep.py:14-15 and called at estep.py:30.
 * Same signature, same ownership (caller owns everything, `map` is updated in place and
 * returned), HOST pointers.  shape = int32[4] {B,H,W,C}; stride = int32[4] in BYTES (numpy
 * strides_as(c_int)); label = int32 [B,H,W] contiguous.
 * Semantics are those of estep.py:33-84 (estep_py, the path deeplab.py:114 uses), which
 * the reference's C++ only approximates: batch-safe, label set from
 * every pixel, running max initialised correctly, one class order per iteration shared by
 * the batch.  The class order comes from an internal generator (emadapt_estep_seed); use
 * e_step_perm for an explicit order.  Returns NULL on error (the reference never does). */
float *e_step(float *map, const int *shape, const int *stride, const int *label,
              const bool suppress_others, const int num_iter, const float margin_others,
              float bg, float fg);

/* e_step with the visiting order as an input: perms = int32 [num_iter, C-1], each row a
 * permutation of 1..C-1 (class 0 is always visited first, estep.py:64-66). */
float *e_step_perm(float *map, const int *shape, const int *stride, const int *label,
                   const int *perms, const bool suppress_others, const int num_iter,
                   const float margin_others, float bg, float fg);

/* Seed of the generator e_step draws its class orders from (replaces the reference's
 * unseeded rand()/random_shuffle, weaklabel.cpp:83). */
void emadapt_estep_seed(uint64_t seed);

/* Label-kind selector for the entries below. */
#define EMADAPT_LABEL_F32 0      /* float pixel labels [B, label_len]; cast to uint8 like estep.py:35 */
#define EMADAPT_LABEL_I32 1      /* int32 pixel labels [B, label_len]                               */
#define EMADAPT_LABEL_MULTIHOT 2 /* uint8 image-level multi-hot [B, C] ("synthetic image-level labels") */

/* Host-buffer E-step with explicit thresholds: what the Python operator estep()
 * (estep.py:18-27) binds.  map: float32 [B,H,W,C] contiguous, updated in place (float map
 * of estep.py:82); labels_out: optional int64 [B,H,W] = argmax over C, first max
 * (deeplab.py:123).  k_bg = int(H*W*bg_p), k_fg = int(H*W*fg_p) (estep.py:58-59). */
int emadapt_estep_host(float *map, int B, int H, int W, int C, const void *label, int label_kind,
                       int64_t label_len, const int32_t *perms, int suppress_others, int num_iter,
                       float margin_others, int k_bg, int k_fg, int64_t *labels_out);

/* Bytes of device workspace emadapt_estep needs: 256, plus -- when H*W*C is too large for every class that may be present in
 * an image to stay in LDS (513x513 / 41 classes) -- B*(C+3)*roundup4(H*W) floats (every class image class-major, the per-pixel
 * maxima, the partial argmax's value and class planes) and align16(4*B) bytes (the images' shifts):
 *     256 + B*(C+3)*roundup4(H*W)*4 + align16(4*B).   Always query; do not hard-code (ABI history 200-202). */
size_t emadapt_estep_workspace_bytes(int B, int H, int W, int C);

/* 0 for every supported geometry (map_out may always be NULL), <0 for an unsupported one.  The first version returned 1 when
 * the float map doubled as workspace for the channels that do not fit in LDS; those now live in the workspace. */
int emadapt_estep_needs_map(int H, int W, int C);

/* Device E-step: replaces deeplab.py:109-124 (py_func(estep) + argmax) with no host round trip.
 *   logits     f32 [B*H*W rows][ldl], first C entries of each row used (not modified)
 *   label      per label_kind; label_len = elements per image for the pixel kinds
 *   perms      int32 [num_iter, C-1] on the device
 *   map_out    f32 [B,H,W,C] float map of estep.py:82, or NULL when only labels are wanted
 *   labels_out int64 [B,H,W]
 * One workgroup per image; present channels and the running max live in LDS. */
int emadapt_estep(const float *logits, int64_t ldl, const void *label, int label_kind,
                  int64_t label_len, const int32_t *perms, int B, int H, int W, int C,
                  int suppress_others, int num_iter, float margin_others, int k_bg, int k_fg,
                  float *map_out, int64_t *labels_out, void *workspace, size_t workspace_bytes,
                  void *stream);

/* tf.image.resize_nearest_neighbor(label,[oh,ow]) + squeeze, align_corners=False -- deeplab.py:110.
 * in: f32 [B,H,W] (the trailing singleton channel of the placeholder dropped), out: f32 [B,oh,ow]. */
int emadapt_resize_nearest(const float *in, int B, int H, int W, float *out, int oh, int ow, void *stream);

/* Inference head, network.py:39-41: pred = argmax_c( resize_bilinear(logits, [OH,OW]) ), align_corners=False with
 * TF1's legacy source mapping src = dst*in/out.  logits f32 [B,IH,IW][ldl] (first C used); pred int64 [B,OH,OW];
 * rescale_out f32 [B,OH,OW,C] receives the interpolated logits when non-NULL (net["rescale_output"]); pred may be
 * NULL when only the resized tensor is wanted (plain tf.image.resize_bilinear). */
int emadapt_pred_bilinear_argmax(const float *logits, int64_t ldl, int B, int IH, int IW, int C, int OH, int OW,
                                 int64_t *pred, float *rescale_out, void *stream);

/* Input pipeline tail, dataset.py:175-181,187-199 (next row f1): RGB -> BGR, minus the per-channel mean
 * (dataset.py:15-18), optional left-right flip of image and label (flip: uint8 [B], may be NULL).  The two resizes in
 * front of it (dataset.py:153-173) are emadapt_pred_bilinear_argmax with pred == NULL (plain resize_bilinear of a
 * C-channel image) and emadapt_resize_nearest.  img_rgb f32 [B,H,W,3]; label f32 [B,H,W] (may be NULL). */
int emadapt_bgr_mean_flip(const float *img_rgb, const float *label, int B, int H, int W, const float mean_bgr[3],
                          const uint8_t *flip, float *img_out, float *label_out, void *stream);

/* The whole of dataset.image_preprocess (dataset.py:147-185) behind the host's JPEG / PNG decode, for a BATCH, in one launch
 * (ABI 206): uint8 -> f32, the "train" branch's random scale (bilinear image / nearest label to h1 x w1; h1 = w1 = 0 skips it: the
 * "val" branch), the resize to OH x OW, RGB -> BGR, minus mean, left-right flip.  staging (device) holds the decoded samples as the
 * decoder leaves them: sample b's image as uint8 [h0][w0][3] RGB at img_off, its class-index map as uint8 [h0][w0] at lab_off.
 * samples is a HOST array (validated here, passed to the kernel by value).  img_out f32 [B,OH,OW,3], label_out f32 [B,OH,OW] (may
 * be NULL).  Results are bit for bit those of the op-by-op path above (emadapt_pred_bilinear_argmax twice, emadapt_resize_nearest
 * twice, emadapt_bgr_mean_flip): one thread per output pixel evaluates the same fp32 operations in the same order. */
typedef struct emadapt_voc_sample {
    int64_t img_off, lab_off;   /* byte offsets into staging */
    int32_t h0, w0;             /* decoded size */
    int32_t h1, w1;             /* size after the random scale: int(float(h0) * scale), int(float(w0) * scale) -- dataset.py:154-155; 0, 0 = none */
    int32_t flip;               /* != 0: left-right flip of image and label */
    int32_t reserved;
} emadapt_voc_sample;
int emadapt_voc_preprocess(const uint8_t *staging, int64_t staging_bytes, const emadapt_voc_sample *samples, int B, int OH, int OW,
                           const float mean_bgr[3], float *img_out, float *label_out, void *stream);
/* Staging of that batch (ABI 206): emadapt_host_register pins a host block the caller owns (the shared-memory ring the decode
 * processes write into; hipHostRegister), emadapt_memcpy_h2d_async queues one host-to-device copy on `stream` (asynchronous when
 * the source is pinned, staged by the runtime otherwise -- correct either way), emadapt_host_unregister undoes the pinning. */
int emadapt_host_register(void *ptr, size_t bytes);
int emadapt_host_unregister(void *ptr);
int emadapt_memcpy_h2d_async(void *dst_device, const void *src_host, size_t bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Convolution stack (DeepLab VGG-16 with dilation): implicit-GEMM on MFMA
 * ---------------------------------------------------------------------------------------- */

typedef struct emadapt_conv_desc {
    int B, H, W;   /* input = output spatial size: every conv is stride 1, SAME (deeplab.py:58,65,92,95) */
    int Cin, Cout; /* both multiples of 4 (callers pad 3->4 input channels and C->round4 classes)          */
    int KH, KW;    /* 3x3, 4x4 (fc6) or 1x1                                                                */
    int dil;       /* atrous rate: 1, 2 (conv5_x, deeplab.py:65), 4 (fc6, deeplab.py:92), 12 (LargeFOV)    */
} emadapt_conv_desc;

#define EMADAPT_EPI_NONE 0       /* y = conv                                   */
#define EMADAPT_EPI_BIAS 1       /* y = conv + b              (fc8)            */
#define EMADAPT_EPI_BIAS_RELU 2  /* y = max(conv + b, 0)      (conv*, fc6/7)   */

/* Forward: tf.nn.conv2d / tf.nn.atrous_conv2d + tf.nn.bias_add (+ tf.nn.relu (+ tf.nn.dropout))
 * -- deeplab.py:58-59,65-66,71,92-96,100,104.   x f32 [B,H,W,Cin]; w f32 HWIO [KH,KW,Cin,Cout];
 * bias f32 [Cout]; y f32 [B,H,W,Cout].  Dropout (only with EPI_BIAS_RELU): keep_prob < 1 multiplies kept
 * values by the f32 number 1.0f / keep_prob (tf.nn.dropout as of TF 1.13; equal to x / keep_prob bit for bit
 * when keep_prob is a power of two, e.g. the reference's 0.5); the keep mask is drop_mask (uint8 [B,H,W,Cout], 1 = keep)
 * when non-NULL, else a counter-based hash of (drop_seed, element index). */
size_t emadapt_conv_fwd_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_fwd(const emadapt_conv_desc *d, const float *x, const float *w, const float *bias,
                     float *y, int epilogue, float keep_prob, const uint8_t *drop_mask,
                     uint64_t drop_seed, void *workspace, size_t workspace_bytes, void *stream);
/* workspace (fwd and dgrad): optional.  Layers whose output has only a few hundred 128x128 tiles are run
 * split-K (fp32 partial slabs in the workspace + a fused, fixed-order reduce/epilogue kernel) so that the
 * workgroups fill whole rounds of the 256 CUs; with workspace == NULL (or too small) the same result is
 * produced unsplit.  *_workspace_bytes returns what the split plan for this shape needs (0 = none).
 * Stream-K launches (layers whose 256 x 256 tiles are not a whole number of rounds: conv3-conv5, fc6, the tails of fc7) keep
 * one arrival word and one 256 KB accumulator image per workgroup there -- 64 MB + 1 KB on 256 CUs whatever the layer -- and
 * combine the tiles that several workgroups share INSIDE the launch, in ascending k (deterministic; since round 4, ABI
 * unchanged: rounds 2-3 asked for up to three [M][N] images and ran a second kernel).  The words are zeroed by a memset node the
 * entry point enqueues ahead of the launch; the caller provides the bytes and nothing else. */

/* Data gradient (Conv2DBackpropInput of the autodiff in deeplab.py:192):
 *   dx[b,h,w,ci] = sum_{taps,co} dy[b,h-off,w-off,co] * w[tap,ci,co]
 * fused with the gradient of what precedes the conv: when `gate` is non-NULL (f32 [B,H,W,Cin], the
 * stored post-ReLU(/dropout) activation that fed this conv) dx = gate>0 ? dx*gate_scale : 0
 * (ReluGrad, and dropout's 1/keep_prob when the gate is a dropped activation). */
size_t emadapt_conv_dgrad_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_dgrad(const emadapt_conv_desc *d, const float *dy, const float *w, float *dx,
                       const float *gate, float gate_scale, void *workspace, size_t workspace_bytes,
                       void *stream);

/* Filter gradient (Conv2DBackpropFilter): dw[tap,ci,co] = sum_{b,h,w} x[b,h+off,w+off,ci]*dy[b,h,w,co].
 * Deterministic: the pixel dimension is split over workgroups into fp32 partial slabs in `workspace`
 * that a second kernel sums in a fixed order (no atomics).  dw f32 HWIO. */
size_t emadapt_conv_wgrad_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_wgrad(const emadapt_conv_desc *d, const float *x, const float *dy, float *dw,
                       void *workspace, size_t workspace_bytes, void *stream);
/* The same launch also returning BiasAddGrad, db[co] = sum_{b,h,w} dy[b,h,w,co] (the bias half of the gradients of
 * deeplab.py:192), where the filter-gradient kernel holds every dy row in registers anyway -- the layers on the round-1
 * kernel: conv1_x, conv2_x, conv3_1, fc8, whose gradients are 80 % of the bytes emadapt_bias_grad re-reads per step.
 * emadapt_conv_wgrad_bias_fused(d): 1 where this entry takes the shape (else EMADAPT_ERR_UNSUPPORTED: call emadapt_conv_wgrad
 * and emadapt_bias_grad).  Deterministic (per producer wave and pixel split, added in a fixed order); dw is bit-identical to
 * emadapt_conv_wgrad's, db differs from emadapt_bias_grad's in the order of the fp32 sum only.  (ABI 202) */
int emadapt_conv_wgrad_bias_fused(const emadapt_conv_desc *d);
size_t emadapt_conv_wgrad_bias_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_wgrad_bias(const emadapt_conv_desc *d, const float *x, const float *dy, float *dw, float *db,
                            void *workspace, size_t workspace_bytes, void *stream);

/* ---- bf16-compute variants (configurations C3/C4: bf16 compute, fp32 master weights
 * and fp32 accumulation).  Same semantics, shapes and epilogues as the fp32 entry points above; the operands
 * of every inner product are bf16 (round-to-nearest-even), products and sums are fp32
 * (v_mfma_f32_32x32x16_bf16).
 *
 * Storage types.  Each activation / activation-gradient tensor is EMADAPT_DT_F32 or EMADAPT_DT_BF16 in memory
 * (the *_dtype argument next to its pointer).  An f32 tensor is rounded to bf16 by the kernel that consumes
 * it; a bf16 tensor was rounded once by the kernel that produced it (after bias/ReLU/dropout resp. the
 * ReLU-gradient gate, which are evaluated in fp32) -- the same values either way, at half the bytes.  bf16
 * tensors need a channel count that is a multiple of 8 (16-byte vectors).  Bias, filter gradients, logits
 * consumed by the E-step and all optimizer state are always f32.
 *
 * The filters are read from two bf16 images that emadapt_weights_to_bf16 derives from the f32 HWIO master copy
 * (once per optimizer step):
 *   which = 0, forward image  [KH*KW][Cout][round8(Cin)]   (reduction axis ci contiguous, zero padded)
 *   which = 1, dgrad image    [KH*KW][Cin][round8(Cout)]   (the HWIO filter with co padded to 8)
 * The filter gradient needs no image; dw is written f32 HWIO. */
#define EMADAPT_DT_F32 0
#define EMADAPT_DT_BF16 1
size_t emadapt_weights_bf16_bytes(const emadapt_conv_desc *d, int which);
int emadapt_weights_to_bf16(const emadapt_conv_desc *d, const float *w, void *w_fwd_bf16, void *w_dgrad_bf16,
                            void *stream);
/* emadapt_weights_to_bf16 of up to 16 filters in one launch (ABI 205): arrays of n descriptors, filters and image pointers (an
 * image pointer may be NULL as above, not both of a layer). */
int emadapt_weights_to_bf16_multi(int n, const emadapt_conv_desc *const *d, const float *const *w, void *const *w_fwd_bf16,
                                  void *const *w_dgrad_bf16, void *stream); /* either image pointer may be NULL */
size_t emadapt_conv_fwd_bf16_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_fwd_bf16(const emadapt_conv_desc *d, const void *x, int x_dtype, const void *w_fwd_bf16,
                          const float *bias, void *y, int y_dtype, int epilogue, float keep_prob,
                          const uint8_t *drop_mask, uint64_t drop_seed, void *workspace, size_t workspace_bytes,
                          void *stream);
size_t emadapt_conv_dgrad_bf16_workspace_bytes(const emadapt_conv_desc *d);
/* gate (the stored activation that fed the conv) has the storage type of dx */
int emadapt_conv_dgrad_bf16(const emadapt_conv_desc *d, const void *dy, int dy_dtype, const void *w_dgrad_bf16,
                            void *dx, int dx_dtype, const void *gate, float gate_scale, void *workspace,
                            size_t workspace_bytes, void *stream);
size_t emadapt_conv_wgrad_bf16_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_wgrad_bf16(const emadapt_conv_desc *d, const void *x, int x_dtype, const void *dy, int dy_dtype,
                            float *dw, void *workspace, size_t workspace_bytes, void *stream);
/* Filter gradient AND BiasAddGrad (db[n] = sum over pixels of dy[m][n], f32 [Cout], 16-byte aligned) in one launch: the loader
 * waves add up dy pieces that pass through them anyway -- the workgroups that stream the same pixels and columns (one per filter-row
 * tile) divide the K steps among them -- and the same second-stage launch that sums the split-M partial filter gradients sums the
 * partial bias rows (fixed order: bit-reproducible; the separate emadapt_bias_grad_dt costs two launches and one more pass over dy
 * per layer).  Needs dy stored bf16: emadapt_conv_wgrad_bias_bf16_fused() answers 1 where it applies (else 0 and the entry returns
 * EMADAPT_ERR_UNSUPPORTED; tune "wgrad_bias" = 1 makes it answer 0 everywhere).  (ABI 204) */
int emadapt_conv_wgrad_bias_bf16_fused(const emadapt_conv_desc *d, int x_dtype, int dy_dtype);
size_t emadapt_conv_wgrad_bias_bf16_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_wgrad_bias_bf16(const emadapt_conv_desc *d, const void *x, int x_dtype, const void *dy, int dy_dtype,
                                 float *dw, float *db, void *workspace, size_t workspace_bytes, void *stream);
/* The plan a bf16 launch will take for this shape (kind 0 forward, 1 data gradient, 2 filter gradient): plan[0] x plan[1] = the
 * output tile (forward / data gradient: pixels x channels, 256 x 256 or 128 x 128; filter gradient: filter rows x columns),
 * plan[2] = split-K / split-M factor.  The parity tests assert through it that the benchmark shapes run the big tiles.  (ABI 200)
 * (ABI 203-205 carried an opt-in patch kernel, tune "bf16_tile" = 3 -> plan[0] == 3; deleted in ABI 206: the key value is accepted and ignored.)
 * ABI 204: plan[0] == 4 (forward / data gradient, automatic -- any forced "bf16_tile" switches it off) = the DIRECT kernel
 * (csrc/conv_d64.h) for 3 x 3, rate 1, 64 -> 64 channels on bf16-stored tensors without dropout (conv1_2): plan[1] x plan[2] = its
 * output tile in pixels; the filter stays in registers / LDS for the whole launch, the input patch of a tile is fetched once.
 * plan[0] == 5 (forward, automatic): the direct kernel for 3 x 3, rate 1, 4 -> 64 channels from an f32-stored input into a bf16-stored
 * output without dropout (csrc/conv_d4.h: conv1_1 on the 3 + 1-channel image).
 * plan[0] == 6 (data gradient, automatic): 1 x 1 with at most 32 gathered channels and a multiple of 256 output channels on bf16-stored
 * tensors (csrc/conv_k32.h: fc8's data gradient): plan[1] pixels x plan[2] channels per workgroup, operands straight from memory. */
int emadapt_conv_bf16_plan(const emadapt_conv_desc *d, int kind, int plan[3]);

/* ---- bf16 forward / data gradient on LDS-DMA (csrc/igemm_b16t.h; round 5: 8 MFMA wavefronts on v_mfma_f32_16x16x32_bf16 + 4 DMA
 * wavefronts, what the model runs for its layers with 256 and more output columns).  For bf16-STORED activations whose gathered
 * channel count is a multiple of 32 (every layer from conv1_2 on; kind 0: Cin % 32 == 0, kind 1: Cout % 32 == 0) the operand tiles
 * go global -> LDS by DMA with the filter pre-tiled in LDS order:
 *     tiled image  [tap][k / 32][row, zero rows up to a multiple of 256][4 x 16 B, slot s of row r = k chunk s ^ ((-(r >> 2)) & 3)]
 *     (ABI 203: round 4's image swizzled with (r >> 2) & 3 -- an image built by a 202 library is not valid for a 203 one)
 *     dgrad = 0: rows = Cout, k = Cin;  dgrad = 1: rows = Cin, k = Cout     (emadapt_filter_tiles_bf16, once per optimizer step)
 * Same arithmetic as emadapt_conv_fwd_bf16 / _dgrad_bf16 (bf16 operands rounded RNE, exact products, fp32 sums; the K order of
 * the sum differs), same epilogues, bitwise reproducible.  y_dtype / dx_dtype: EMADAPT_DT_*; the gate has dx's storage type.
 * emadapt_conv_bf16t_supported: 1 when the shape qualifies.  emadapt_conv_bf16t_plan: plan[0] rows of the first launch (whole
 * rounds of 256 x 256 tiles on the CUs), [1] 1 = that launch is stream-K, [2] the most workgroups that share one of its tiles (0:
 * every tile whole), [3] the same for the stream-K launch over the remaining rows (-1: plain tiles; 0 with plan[0] == M: there is
 * none; 224 with plan[0] == M (ABI 206): the one plain launch runs 224 x 256 tiles -- a launch whose 256-row tiles fill one partial
 * round and whose 224-row tiles still fit one round, e.g. M = 26,896, N = 512: 242 workgroups of 7 / 8 the work instead of 212; tune
 * "bf16t_sk" = 3 keeps the 256-row tiles; same results bit for bit).  Stream-K launches with shared tiles combine them INSIDE the launch (ABI 203; round 4 ran a second kernel over slab
 * levels): emadapt_conv_bf16t_workspace_bytes is then arrival words + one 256 KB accumulator image per workgroup (64 MB + 1 KB on 256
 * CUs), 0 otherwise.  (ABI 200) */
int emadapt_conv_bf16t_supported(const emadapt_conv_desc *d, int kind, int y_dtype);
/* supported AND measured faster than emadapt_conv_{fwd,dgrad}_bf16 for this shape (four and more rounds of 256 x 256 tiles on the
 * CUs: fc6 / fc7 at batch 16; one and a half rounds and more with K >= 2304: conv3_2 / conv3_3, LargeFOV's fc6 forward); what the
 * model asks per layer and batch size.  (ABI 203) */
int emadapt_conv_bf16t_pays(const emadapt_conv_desc *d, int kind, int y_dtype);
size_t emadapt_filter_tiles_bf16_bytes(const emadapt_conv_desc *d, int dgrad);
int emadapt_filter_tiles_bf16(const emadapt_conv_desc *d, const float *w, void *tiles, int dgrad, void *stream);
/* both images from one pass over the f32 filter (ABI 204; one kernel where Cin % 256 == 0 and Cout % 256 == 0, else the two calls above) */
int emadapt_filter_tiles_bf16_both(const emadapt_conv_desc *d, const float *w, void *tiles_fwd, void *tiles_dgrad, void *stream);
size_t emadapt_conv_bf16t_workspace_bytes(const emadapt_conv_desc *d, int kind);
int emadapt_conv_bf16t_plan(const emadapt_conv_desc *d, int kind, int plan[4]);
int emadapt_conv_fwd_bf16t(const emadapt_conv_desc *d, const void *x_bf16, const void *w_tiles, const float *bias, void *y,
                           int y_dtype, int epilogue, float keep_prob, const uint8_t *drop_mask, uint64_t drop_seed,
                           void *workspace, size_t workspace_bytes, void *stream);
int emadapt_conv_dgrad_bf16t(const emadapt_conv_desc *d, const void *dy_bf16, const void *w_tiles, void *dx, int dx_dtype,
                             const void *gate, float gate_scale, void *workspace, size_t workspace_bytes, void *stream);

/* ---- fp32-grade GEMMs on the bf16 matrix pipe ("bf16x6", opt-in).  Every fp32 operand is split exactly into
 * three bf16 pieces (x = p0 + p1 + p2: 8+8+8 mantissa bits) and each product is formed from the six piece pairs
 * that matter (the dropped ones are below 2^-25 of the product), accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16: fp32-grade results (same parity bar as the native fp32 kernels) at 16/6 of the native
 * fp32 matrix rate.  All tensors are f32 in memory, shapes / epilogues / determinism as for the fp32 entry points.
 * The forward pass reads the filter from a k-contiguous f32 image [KH*KW][Cout][Cin] (emadapt_filter_kmajor, same
 * byte size as the HWIO filter, refreshed when the weights change); dgrad and wgrad read the HWIO master copy. */
int emadapt_filter_kmajor(const emadapt_conv_desc *d, const float *w, float *w_kmajor, void *stream);
size_t emadapt_conv_fwd_x6_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_fwd_x6(const emadapt_conv_desc *d, const float *x, const float *w_kmajor, const float *bias,
                        float *y, int epilogue, float keep_prob, const uint8_t *drop_mask, uint64_t drop_seed,
                        void *workspace, size_t workspace_bytes, void *stream);
size_t emadapt_conv_dgrad_x6_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_dgrad_x6(const emadapt_conv_desc *d, const float *dy, const float *w, float *dx, const float *gate,
                          float gate_scale, void *workspace, size_t workspace_bytes, void *stream);
size_t emadapt_conv_wgrad_x6_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_wgrad_x6(const emadapt_conv_desc *d, const float *x, const float *dy, float *dw, void *workspace,
                          size_t workspace_bytes, void *stream);

/* "bf16x6" on PRE-SPLIT operands (csrc/conv_x6p.hip).  The split costs vector-ALU work per element; the entry points above do
 * it inside the kernels, once per tile and filter tap.  Here it is done once per tensor into a PLANE IMAGE
 *     [rows][C / 16][plane 0..2][16 x bf16]      (C % 16 == 0; 6 bytes per element; x = p0 + p1 + p2 exactly)
 * and the kernels (256 x 256 tiles, LDS-DMA, one workgroup per CU) read plane images only: same arithmetic, same error bound,
 * same epilogues / determinism as the x6 entry points, fp32 outputs.  Operands:
 *   forward          planes of x [B*H*W][Cin]   and the TILED plane image of the filter, rows = Cout, k = Cin
 *   data gradient    planes of dy [B*H*W][Cout] and the tiled plane image of the filter, rows = Cin, k = Cout
 *   filter gradient  planes of x and of dy
 * Tiled filter image (emadapt_filter_tiles(d, w_hwio, tiles, dgrad); emadapt_filter_tiles_bytes gives its size, 0 when the
 * reduction's channel count is not a multiple of 16): [tap][k / 16][row, zero rows up to a multiple of 256][6 chunks of 16 B],
 * chunk (2 plane + half) of a row at position chunk ^ (bit 3 of the row) -- a 256-row tile's K step is 24 contiguous KB in the
 * order the kernel keeps it in LDS, so the filter side of the kernel is plain copies.  Refresh it when the weights change.
 * emadapt_conv_x6p_supported(d, kind) (kind 0 forward, 1 data gradient, 2 filter gradient) says whether a convolution is taken
 * -- forward / data gradient: reduction channels (Cin resp. Cout) in whole groups of 16 and at least 256 output columns (Cout
 * resp. Cin; a multiple of 4), at most 32 taps; filter gradient: Cin a multiple of 256, Cout >= 128 and a multiple of 16; all:
 * plane images below 2 GiB -- the others stay on the x6 entry points and return EMADAPT_ERR_UNSUPPORTED here.  Workspace: the stream-K slabs
 * (+ the row table of packed rows) resp. split-M slabs, as for the fp32 kernels; EMADAPT_ERR_WORKSPACE when it is too small.
 * emadapt_conv_x6p_plan: plan[0] rows of the first launch, [1] 1 = stream-K, [2] the most workgroups sharing one of its tiles, [3] pixel classes of packed
 * rows (0 = off), [4] the same count for the stream-K launch over the remaining rows (-1: a plain launch). */
size_t emadapt_planes_bytes(long long rows, int C);
int emadapt_split_planes(const float *x, void *planes, long long rows, int C, void *stream);
size_t emadapt_filter_tiles_bytes(const emadapt_conv_desc *d, int dgrad);
int emadapt_filter_tiles(const emadapt_conv_desc *d, const float *w, void *tiles, int dgrad, void *stream);
int emadapt_conv_x6p_supported(const emadapt_conv_desc *d, int kind);
int emadapt_conv_x6p_pays(const emadapt_conv_desc *d, int kind);   /* supported AND enough work to fill the part (else: the x6 entry points) */
int emadapt_conv_x6p_plan(const emadapt_conv_desc *d, int dgrad, int plan[5]);
size_t emadapt_conv_fwd_x6p_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_fwd_x6p(const emadapt_conv_desc *d, const void *x_planes, const void *w_tiles, const float *bias,
                         float *y, int epilogue, float keep_prob, const uint8_t *drop_mask, uint64_t drop_seed,
                         void *workspace, size_t workspace_bytes, void *stream);
size_t emadapt_conv_dgrad_x6p_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_dgrad_x6p(const emadapt_conv_desc *d, const void *dy_planes, const void *w_tiles_dgrad, float *dx,
                           const float *gate, float gate_scale, void *workspace, size_t workspace_bytes, void *stream);
size_t emadapt_conv_wgrad_x6p_workspace_bytes(const emadapt_conv_desc *d);
int emadapt_conv_wgrad_x6p(const emadapt_conv_desc *d, const void *x_planes, const void *dy_planes, float *dw,
                           void *workspace, size_t workspace_bytes, void *stream);

/* The HBM-bound kernels of the step on either storage type (dtype = EMADAPT_DT_*; same semantics as the f32
 * entry points below, arithmetic in fp32 registers, results rounded to nearest-even when the output is bf16). */
int emadapt_maxpool_fwd_dt(const void *x, int dtype, int B, int H, int W, int C, int stride, void *y, uint8_t *idx,
                           void *stream);
int emadapt_maxpool_bwd_dt(const void *dy, const uint8_t *idx, const void *gate, int dtype, int B, int H, int W,
                           int C, int stride, void *dx, void *stream);
/* emadapt_maxpool_fwd_dt whose index byte also carries the ReluGrad gate of x when relu_gate != 0 (ABI 205): bit 4 (0x10) is set
 * where the window's winner is not > 0.  ReluGrad of the layer that produced x only matters at a pixel that wins a window, and
 * there x IS the window's maximum -- so emadapt_maxpool_bwd[_dt] on such an index with gate = NULL returns exactly what it
 * returns on a plain index with gate = x (a NaN gradient passes a closed gate either way), without reading x: 211 of the 502 MB
 * of pool1's gradient at batch 16.  The backward entries decode bit 4 of every index they are given (a plain index never has it
 * set); an explicit gate is applied on top.  idx must not be NULL when relu_gate != 0. */
int emadapt_maxpool_fwd_gated(const void *x, int dtype, int B, int H, int W, int C, int stride, void *y, uint8_t *idx,
                              int relu_gate, void *stream);
int emadapt_bias_grad_dt(const void *dy, int dtype, int64_t M, int N, float *db, void *workspace,
                         size_t workspace_bytes, void *stream);
int emadapt_softmax_ce_dt(const float *logits, int64_t ldl, const int64_t *labels, int64_t M, int C,
                          float grad_scale, void *dlogits, int dlogits_dtype, float *loss_sum, void *workspace,
                          size_t workspace_bytes, void *stream);

/* BiasAddGrad: db[n] = sum_m dy[m,n]; dy f32 [M,N].  Deterministic two-stage reduction. */
size_t emadapt_bias_grad_workspace_bytes(int64_t M, int N);
int emadapt_bias_grad(const float *dy, int64_t M, int N, float *db, void *workspace,
                      size_t workspace_bytes, void *stream);

/* tf.nn.max_pool ksize 3x3, stride 2 or 1, SAME (deeplab.py:77,82); padding never wins.
 * x f32 [B,H,W,C] -> y f32 [B,OH,OW,C], OH = ceil(H/stride); idx uint8 [B,OH,OW,C] = winning tap
 * (0..8, first maximum in window scan order) kept for the backward pass. */
int emadapt_maxpool_fwd(const float *x, int B, int H, int W, int C, int stride, float *y, uint8_t *idx,
                        void *stream);
/* MaxPoolGrad fused with the ReluGrad of the layer below: dx[p] = sum over windows whose winner is p
 * of dy, then dx = gate>0 ? dx : 0 when gate (f32 [B,H,W,C], = the pool input) is non-NULL -- or, with an index written by
 * emadapt_maxpool_fwd_gated(relu_gate = 1), the same from the index alone (gate = NULL). */
int emadapt_maxpool_bwd(const float *dy, const uint8_t *idx, const float *gate, int B, int H, int W,
                        int C, int stride, float *dx, void *stream);

/* Zero-pad / strip the channel dimension: dst[m, 0:Cdst] = src[m, 0:min(Csrc,Cdst)], rest 0. */
int emadapt_copy_pad_channels(const float *src, int64_t rows, int Csrc, float *dst, int Cdst, void *stream);

/* ------------------------------------------------------------------------------------------
 * Loss and optimiser
 * ---------------------------------------------------------------------------------------- */

/* tf.nn.sparse_softmax_cross_entropy_with_logits + tf.reduce_mean (deeplab.py:182) and its gradient.
 *   logits f32 [M][ldl] (first C used), labels int64 [M]
 *   dlogits f32 [M][ldl] = (softmax - onehot) * grad_scale in the first C columns, 0 in the padding
 *           columns (may be NULL); grad_scale = 1/M for the mean
 *   loss_sum f32[1] = sum_m CE_m  (deterministic two-stage reduction; may be NULL) */
size_t emadapt_softmax_ce_workspace_bytes(int64_t M);
int emadapt_softmax_ce(const float *logits, int64_t ldl, const int64_t *labels, int64_t M, int C,
                       float grad_scale, float *dlogits, float *loss_sum, void *workspace,
                       size_t workspace_bytes, void *stream);

/* tf.nn.l2_loss: out[0] = sum(x^2)/2 (deeplab.py:184; logging only). */
size_t emadapt_l2_loss_workspace_bytes(int64_t n);
int emadapt_l2_loss(const float *x, int64_t n, float *out, void *workspace, size_t workspace_bytes,
                    void *stream);

/* Gradient accumulation, deeplab.py:202 with the weight-decay term of deeplab.py:184-185 folded in:
 *   accum[i] += (g[i] + (i < n_decay ? weight_decay * w[i] : 0)) * scale      (scale = 1/accum_num)
 * The first n_decay elements of the segment are filter weights, the rest biases. */
int emadapt_sgd_accumulate(float *accum, const float *g, const float *w, int64_t n, int64_t n_decay,
                           float weight_decay, float scale, void *stream);
/* tf.train.MomentumOptimizer.apply_gradients on the accumulators + the clean op
 * (deeplab.py:203-205): v = momentum*v + accum*accum_scale; w -= lr*v; accum = 0. */
int emadapt_sgd_momentum(float *w, float *v, float *accum, int64_t n, float lr, float momentum,
                         float accum_scale, void *stream);
/* The two ops above for ONE micro-batch in one pass (accum_num == 1, no gradient exchange; the accumulator is not touched and
 * stays zero): v = momentum*v + (0 + (g + [i < n_decay] weight_decay*w) * gscale) * ascale; w -= lr*v -- operation for
 * operation what emadapt_sgd_accumulate followed by emadapt_sgd_momentum compute on a zero accumulator, so the weights are
 * bit-identical (tests/test_gpu_kernels.py); 20 bytes per parameter instead of 40.  (ABI 203) */
int emadapt_sgd_accumulate_momentum(float *w, float *v, const float *g, int64_t n, int64_t n_decay, float weight_decay,
                                    float gscale, float lr, float momentum, float ascale, void *stream);
/* The same pass over up to 16 disjoint segments [start[k], start[k] + n[k]) of w / v / g in ONE launch (ABI 205), the first
 * n_decay[k] elements of each decayed: the small layers' launches are all ramp and tail.  start, n, n_decay are HOST arrays of
 * multiples of 4 (else EMADAPT_ERR_UNSUPPORTED: take such a segment to the one-segment entry); bit-identical results. */
int emadapt_sgd_accumulate_momentum_multi(float *w, float *v, const float *g, int nseg, const int64_t *start, const int64_t *n,
                                          const int64_t *n_decay, float weight_decay, float gscale, float lr, float momentum,
                                          float ascale, void *stream);
/* The same pass for ONE layer whose forward and data gradient both run on the LDS-DMA bf16 kernel (ABI 205): w / v / g hold the
 * layer's KH*KW*Cin*Cout filter (HWIO, all of it decayed) followed by n_bias biases (not decayed); the kernel also writes both
 * tiled bf16 images of the NEW filter (emadapt_filter_tiles_bf16_both), which the refresh pass would otherwise build from a second
 * read of the weights.  Weights and momentum come out bit for bit as from emadapt_sgd_accumulate_momentum, the images as from
 * emadapt_filter_tiles_bf16_both on those weights.  _supported: Cin % 256 == 0 and Cout % 256 == 0 (the shapes the one-pass image
 * kernel takes); elsewhere the entry returns EMADAPT_ERR_UNSUPPORTED.  A raised status word leaves weights, momentum AND images
 * untouched. */
int emadapt_sgd_accumulate_momentum_tiles_supported(const emadapt_conv_desc *d);
int emadapt_sgd_accumulate_momentum_tiles(const emadapt_conv_desc *d, float *w, float *v, const float *g, int64_t n_bias,
                                          float weight_decay, float gscale, float lr, float momentum, float ascale,
                                          void *tiles_fwd, void *tiles_dgrad, void *stream);
/* ... and the data-parallel job's Momentum pass on the exchange's gathered payload (emadapt_sgd_momentum_payload below) for such a
 * layer, with both images (ABI 205): w / v / accum (may be NULL) / payload point at the layer's [filter | biases] segment; same
 * shapes as _tiles_supported; weights, momentum and accumulator bit for bit emadapt_sgd_momentum_payload's. */
int emadapt_sgd_momentum_payload_tiles(const emadapt_conv_desc *d, float *w, float *v, float *accum, const void *payload,
                                       int payload_dtype, int64_t n_bias, float lr, float momentum, float scale,
                                       void *tiles_fwd, void *tiles_dgrad, void *stream);

/* ---- staging of the data-parallel gradient exchange (dist.py; the reference's accum += g/accum_num, deeplab.py:202, with
 * micro-batches mapped to ranks).  The exchange is reduce-scatter + all-gather written as two all-to-all steps over all xGMI
 * links; these are its only arithmetic, one pass each.  dtype = EMADAPT_DT_* of the payload on the wire.  (ABI 203)
 *   emadapt_grad_pack        send[0:n] = payload(src[0:n]) (bf16: round to nearest even), send[n:npad] = 0
 *   emadapt_grad_owner_sum   out[i] = payload( f32(recv[0][i]) + f32(recv[1][i]) + ... ), fp32 adds in rank order, rounded
 *                            once; recv = [world][shard] as all_to_all_single delivers it; out = this rank's summed shard
 *   emadapt_sgd_momentum_payload   emadapt_sgd_momentum reading the summed gradient from the gathered payload (f32 or bf16)
 *                            instead of the accumulator, which it only zeroes (the clean op): the gathered values never
 *                            travel back into the accumulator. */
int emadapt_grad_pack(const float *src, int64_t n, void *dst, int dst_dtype, int64_t npad, void *stream);
int emadapt_grad_owner_sum(const void *recv, int dtype, int world, int64_t shard, void *out, void *stream);
int emadapt_sgd_momentum_payload(float *w, float *v, float *accum, const void *payload, int payload_dtype, int64_t n,
                                 float lr, float momentum, float scale, void *stream);   /* accum may be NULL (ABI 204): nothing to zero */
/* accum_num == 1 in a data-parallel job (ABI 204): dst[i] = payload type of 0 + (g[i] + weight_decay w[i] (i < n_decay)) scale -- exactly what
 * emadapt_sgd_accumulate leaves in a zeroed accumulator and emadapt_grad_pack then rounds, without the accumulator (dist.py `prepacked`). */
int emadapt_sgd_accumulate_pack(const float *g, const float *w, int64_t n, int64_t n_decay, float weight_decay, float scale,
                                void *dst, int dst_dtype, void *stream);

/* ------------------------------------------------------------------------------------------
 * Launch-plan overrides.  Every convolution picks a kernel variant and a split factor from its shape; these
 * integer knobs force the choice (0 = automatic, the default).  All variants compute the same function -- the
 * parity tests use the knobs to run each variant on small awkward shapes -- so they are a tuning / test aid, not
 * part of the numerics.  Process-wide; returns EMADAPT_ERR_INVALID for an unknown key.
 *   "fp32_tile"  1: 256x64x16 | 2: 256x256x32 for the rows that fill whole rounds of the CUs + a smaller-tile split-K
 *                launch for the rest | 3: 128x128x32 wave-specialised | 4: 256x256x16 second generation (LDS-DMA
 *                loaders, wide fragment reads; needs Cin resp. Cout % 16 == 0, <= 32 taps), rows split like 2 |
 *                5: the same kernel with the stream-K work split (one workgroup per CU walks an equal share of all
 *                tiles' K steps; pieces of a tile are summed in a fixed order by a second pass).
 *                A forced variant the shape is not eligible for is ignored.
 *   "wgrad_gen"  1: 128x128 wave-specialised filter-gradient kernel | 2: 256x256 second generation (LDS-DMA loaders)
 *   "splitk"     split-K factor of the forward / data-gradient launches (ignored when K is too short)
 *   "bf16_tile"  128 | 256 (emadapt_conv_*_bf16: 128 also keeps the 128-wide tile where 128 x 64 would be chosen); 1: the automatic plan
 *                of the implicit-GEMM kernels with the direct kernels (plan[0] 4 / 5 / 6) switched off: the A/B of those kernels
 *   "bf16t_sk"   1: the tiled-filter bf16 launches (emadapt_conv_*_bf16t) run ALL their tiles as one stream-K launch (A/B aid: measured
 *                slower on fc6 / fc7, profiles/r05_bf16t_bench.txt) | 2: four and more rounds of tiles as whole rounds of plain tiles +
 *                a stream-K launch over the last partial round (rounds 4-5; the default is plain tiles all the way there: fc7 -4 %)
 *                | 3: a launch of one partial round keeps 256 x 256 tiles (default since ABI 206: 224 x 256 where they fill more CUs)
 *   "x6_tile"    128 | 256
 *   "pool_rows"  1: emadapt_maxpool_bwd[_dt] at stride 2 runs its row form (one thread per pixel) instead of 2 x 2 pixel quads
 *                (ABI 205; same results bit for bit)
 *   "plan_cus"  number of CUs the launch plans are laid out for (default 256: MI355X)
 *   "fp32_loader" 1: the 256x256 fp32 kernels' loader waves compute their offsets per K step (works for any tap count);
 *                0: loaders specialised for 1x1 / 3x3 / 4x4 filters where the filter has that many taps, and the
 *                filter-gradient kernel walks only the pixels at which a tile's tap is inside the image; 2: like 0
 *                but every pixel is walked (out-of-image taps contribute zeros)
 *   "pack_pct"   the fp32 forward / data-gradient launches run over packed rows (pixel classes, see emadapt_conv_plan) when
 *                that saves at least this many per cent of the dense launch's products (0 = the built-in threshold)
 *   "raster_gm"  M tiles per group of the plain-tile launches' XCD-aware order (default 8): the workgroups resident on one XCD cover
 *                that many M-tiles x the N-tiles that fit (scripts/raster_sweep.py: L2 hit rate and time per setting)
 *   "dma_stagger" 1: the LDS-DMA kernels without loader waves (the bf16x6 plane kernels) issue every wave's DMA
 *                burst at the same point of the K step (round 3's form); 0 (default): the two waves of a SIMD staggered around
 *                the step's barrier (round 4: +3-6 %)
 *   "wgrad_bias" 1: emadapt_conv_wgrad_bias_fused() answers 0 for every shape (callers then use emadapt_conv_wgrad +
 *                emadapt_bias_grad: A/B of the fused bias gradient, tests of the fallback)
 *   "sk_drop_flag" tests only: 1 = the stream-K publishers keep their arrival words down and the heads give up after 2^12 polls
 *                (the bounded wait: a shared tile then comes out as NaNs, the launch ends)
 *   "estep_stop", "ablate": honoured by diagnostic builds only (-DEMADAPT_ABLATE_BUILD), ignored otherwise
 *                ("estep_stop" 1..4: leave the E-step kernel after that phase; bits 16 / 32 / 64 / 128: run its rank selection
 *                without the LDS atomics / the histogram scans / the barriers / after the first pass -- timing only)
 * ---------------------------------------------------------------------------------------- */
int emadapt_tune_set(const char *key, int value);
/* The plan emadapt_conv_fwd (dgrad = 0) / emadapt_conv_dgrad (dgrad = 1) will use for this shape, given enough workspace:
 * plan[0] = tile variant ("fp32_tile" numbering), plan[1] = rows of the main launch, plan[2] = its split-K factor
 * (variant 5: the largest number of workgroups that share a tile, 0 = none shared), plan[3] = split-K factor / slab levels of the launch over the
 * remaining rows (0 = there is none; variant 5 has no second launch: plan[3] is then the number of pixel classes when the
 * launch skips the filter rows / columns that lie outside the image -- its GEMM rows are the pixels class after class and a
 * tile reduces over the union of its classes' tap boxes; the row -> pixel table takes ceil(M / 256) * 256 ints of the
 * workspace behind the stream-K slabs -- else 0). */
int emadapt_conv_plan(const emadapt_conv_desc *d, int dgrad, int plan[4]);
int emadapt_tune_get(const char *key, int *value);
/* Executed / algorithmic FLOPs of the launches a conv entry makes for this shape (host arithmetic, no device needed): kind 0
 * forward, 1 data gradient, 2 filter gradient; arithmetic 0 = fp32 kernels, 1 = bf16 (always 1.0: the dense reduction),
 * 2 = the bf16x6 plane kernels.  < 1 where the launch skips the products with a dilated filter's zero padding; what
 * emadapt_prof_collect2 accumulates per launch.  (ABI 200) */
int emadapt_conv_executed_fraction(const emadapt_conv_desc *d, int kind, int arithmetic, double *fraction);

/* ------------------------------------------------------------------------------------------
 * Measurement support (bench.py): per-kernel HIP-event timing on the launch stream.
 * ---------------------------------------------------------------------------------------- */
#define EMADAPT_K_IGEMM_FWD 0   /* conv forward implicit GEMM           */
#define EMADAPT_K_IGEMM_DGRAD 1 /* conv data-gradient implicit GEMM     */
#define EMADAPT_K_IGEMM_WGRAD 2 /* conv filter-gradient implicit GEMM   */
#define EMADAPT_K_ESTEP 3       /* E-step scan                          */
#define EMADAPT_K_OTHER 4       /* pools, loss, optimiser, reductions   */
#define EMADAPT_K_COUNT 5

void emadapt_prof_enable(int on); /* 1: bracket every launch with hipEventRecord; 0: off (default) */
void emadapt_prof_reset(void);
/* Synchronises the recorded events and returns, per kernel class: launches, total ms, total
 * algorithmic FLOPs (conv classes) and algorithmic bytes (the others). */
int emadapt_prof_collect(int64_t launches[EMADAPT_K_COUNT], double ms[EMADAPT_K_COUNT],
                         double flops[EMADAPT_K_COUNT], double bytes[EMADAPT_K_COUNT]);
/* The same, plus per class the FLOPs the launches really ISSUED (flops_executed, may be NULL): the fp32 and bf16x6 kernels skip
 * the products of a dilated filter's taps with the zero padding (packed pixel classes in forward / data gradient, per-tap pixel
 * boxes in the filter gradient), so executed <= algorithmic; equal for every kernel that walks the dense reduction.  MFMA
 * utilisation is executed FLOPs / time / peak; algorithmic FLOPs / time is the throughput the caller sees.  (ABI 101) */
int emadapt_prof_collect2(int64_t launches[EMADAPT_K_COUNT], double ms[EMADAPT_K_COUNT], double flops[EMADAPT_K_COUNT],
                          double bytes[EMADAPT_K_COUNT], double flops_executed[EMADAPT_K_COUNT]);

#ifdef __cplusplus
}
#endif
#endif /* EMADAPT_H_ */
