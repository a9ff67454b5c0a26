/*
 * planer_hip.h -- C ABI of libplaner_hip.so, the MI355X (gfx950) backend for
 * planer's per-layer forward pass.
 *
 * The reference (Image-Py/planer v0.34) is pure Python; its "backend" is any
 * module that quacks like numpy, swapped in by planer.core() (__init__.py:22-38).
 * Its only native/GPU call site is cupy.cudnn.convolution_forward
 * (util.py:66-77).  This header is what a ctypes binding for a HIP backend
 * binds instead; every entry point cites the reference function it replaces.
 * INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *  - every function returns an int status (PL_OK == 0) and never throws;
 *    pl_last_error() returns a thread-local message for the last failure.
 *  - all tensors are dense fp32, logical layout NCHW, weights OIHW, exactly as
 *    the reference's numpy arrays (layer.py / util.py).
 *  - all pointers named x/y/w/... are DEVICE pointers obtained from pl_alloc;
 *    sizes are element counts unless a name says bytes.
 *  - all work is enqueued on the context's HIP stream and is asynchronous with
 *    respect to the host; pl_sync() waits for it.
 */
#ifndef PLANER_HIP_H
#define PLANER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PL_OK 0
#define PL_EINVAL 1       /* bad argument (null pointer, negative size ...)      */
#define PL_EUNSUPPORTED 2 /* valid in ONNX terms but undefined in the reference, */
                          /* e.g. asymmetric pads (util.py:8), group !| C        */
#define PL_ENOMEM 3
#define PL_EHIP 4         /* a HIP runtime call failed; see pl_last_error()      */
#define PL_ERCCL 5        /* an RCCL call failed                                 */

/* activation codes for the fused conv epilogue */
#define PL_ACT_NONE 0
#define PL_ACT_RELU 1
#define PL_ACT_LEAKY 2
/* OR-ed into an activation code: the fused residual is added AFTER the activation
 * (conv -> batchnorm -> leakyrelu -> add, the order YOLO-v3's blocks use) instead of before it */
#define PL_ACT_RES_AFTER 16

typedef struct pl_ctx pl_ctx;     /* one device + one stream + one memory pool */
typedef struct pl_graph pl_graph; /* a captured forward pass (hipGraphExec)     */
typedef struct pl_plan pl_plan;   /* a whole compiled forward pass read from a plan file (pl_plan_build) */
typedef struct pl_event pl_event; /* hipEvent on the context stream             */

/* ---- runtime --------------------------------------------------------- */
const char *pl_last_error(void);
int pl_version(void);
int pl_device_count(int *count);
int pl_ctx_create(int device, pl_ctx **out);
int pl_ctx_destroy(pl_ctx *ctx);
int pl_ctx_info(pl_ctx *ctx, int *device, int *cu_count, size_t *hbm_bytes,
                char *arch_name, size_t arch_name_len);
/* "domain:bus:device.function" of the context's GPU (what /sys/bus/pci/devices/ is keyed by): lets a host tool
 * read the card's clocks from sysfs without forking a management CLI. */
int pl_ctx_pci_bus_id(pl_ctx *ctx, char *bus_id, size_t len);
int pl_sync(pl_ctx *ctx); /* wait for the context stream */

/* memory: replaces numpy/cupy array allocation (np.zeros, util.py:31-32,88) */
int pl_alloc(pl_ctx *ctx, size_t bytes, void **out); /* pooled, stream-ordered */
int pl_free(pl_ctx *ctx, void *ptr);
int pl_pool_stats(pl_ctx *ctx, size_t *bytes_reserved, size_t *bytes_in_use);
int pl_pool_block(pl_ctx *ctx, const void *ptr, void **base, size_t *bytes); /* the pool block that holds ptr */
int pl_pool_trim(pl_ctx *ctx);                       /* hipFree cached blocks */
/* np.asarray / .get() of net.py:96-100 */
int pl_h2d(pl_ctx *ctx, void *dst, const void *src_host, size_t bytes);
int pl_d2h(pl_ctx *ctx, void *dst_host, const void *src, size_t bytes); /* syncs */
int pl_d2d(pl_ctx *ctx, void *dst, const void *src, size_t bytes);
int pl_memset(pl_ctx *ctx, void *dst, int byte, size_t bytes);
/* The asynchronous halves of the same contract (host ndarray in, host ndarray out: net.py:94-101).
 *   pl_h2d_staged  copies src_host into a ring of pinned buffers, chunk by chunk (library copy threads), and enqueues each
 *                  chunk's DMA on `consumer`'s stream (NULL: ctx's own) while the next is staged.  Returns as soon as
 *                  src_host has been read -- the caller may overwrite it -- and nothing waits for the bytes except what
 *                  `consumer` enqueues afterwards.  dst is a block of ctx's pool; the copy is ordered behind what ctx's stream
 *                  held at the call.  A src_host inside pl_host_alloc memory is NOT staged: the DMA reads it in place, later,
 *                  so the caller keeps it unchanged until `consumer`'s stream has passed the copy (a pl_event tells).
 *   pl_d2h_begin   enqueues device -> pinned buffer on `producer`'s stream (NULL: ctx's); *ticket identifies the buffer
 *                  (-1: every buffer is in flight, use pl_d2h).  No host wait.
 *   pl_d2h_finish  waits for that copy, moves the bytes to dst_host (NULL: drops them), releases the buffer.  The device
 *                  block must stay allocated until then.
 *   pl_host_alloc / pl_host_free   pinned host memory for callers that build batches in place.
 *   pl_copy_threads                worker threads of the staging copies (PLANER_HIP_COPY_THREADS; the caller's thread works too).
 * PLANER_HIP_COPY_STREAMS=1 moves the DMAs onto two dedicated copy streams (measured slower on MI355X: DESIGN 4.8). */
int pl_h2d_staged(pl_ctx *ctx, pl_ctx *consumer, void *dst, const void *src_host, size_t bytes);
/* host -> device at once and on NO stream: the host waits for the DMA, no hardware queue does.  The caller guarantees that no
 * work on the device still uses dst.  (What Net.submit feeds a pipeline's replicas with: DESIGN 4.8.) */
int pl_h2d_direct(pl_ctx *ctx, void *dst, const void *src_host, size_t bytes);
int pl_d2h_begin(pl_ctx *ctx, pl_ctx *producer, const void *src, size_t bytes, int *ticket);
int pl_d2h_finish(pl_ctx *ctx, int ticket, void *dst_host);
int pl_host_alloc(size_t bytes, void **out);
int pl_host_free(void *ptr);
int pl_copy_threads(int *workers);

/* timing: fills Net.timer (net.py:55,67-70) with device time */
int pl_event_create(pl_ctx *ctx, pl_event **out);
int pl_event_record(pl_ctx *ctx, pl_event *ev);
int pl_event_sync(pl_event *ev);        /* host waits until the recorded point of the stream has been reached */
int pl_event_elapsed_ms(pl_event *start, pl_event *stop, float *ms); /* syncs on stop */
int pl_event_destroy(pl_event *ev);

/* fork/join between two contexts (streams) of one device */
int pl_stream_wait(pl_ctx *waiter, pl_ctx *signal);
/* the waiter's stream waits for one recorded point of another stream (Net.submit: a pending result is handed to the
 * caller's stream without waiting for what was queued behind it) */
int pl_stream_wait_event(pl_ctx *waiter, pl_event *ev);
/* Exchange the streams of two idle contexts of one device (each keeps its memory pool, graphs and events).  Which hardware queue
 * a stream runs on is fixed when the runtime creates it; the plan compiler uses this to try assignments of pipeline replicas
 * to streams without re-capturing their graphs.  No counterpart in the reference (net.py has no notion of a stream). */
int pl_ctx_swap_streams(pl_ctx *a, pl_ctx *b);

/* whole-forward capture: the HIP-native replacement for interpreting the flow
 * in Python on every call (net.py:37-72).  Between begin/end every launch and
 * pool allocation on the context is recorded instead of executed. */
int pl_capture_begin(pl_ctx *ctx);
int pl_capture_end(pl_ctx *ctx, pl_graph **out);
int pl_graph_launch(pl_graph *g);
int pl_graph_destroy(pl_graph *g);

/* ---- a compiled forward pass without the Python host ----------------------------------------
 * Replaces the reference's interpreter loop (net.Net.forward, net.py:37-72) for hosts that bind this header directly.
 * A plan file (written once by planer_amd.export.export_plan from a loaded Net and an input shape) holds the FUSED program of
 * the plan compiler -- fused conv epilogues, channel-quad layouts, Winograd stages and chains, paired convs -- as the flat
 * sequence of calls of this ABI, with the constants (weights, prepared filters) and the activation arena's size.
 *   pl_plan_build    parses the file, uploads the constants, runs the sequence once and captures it into a hipGraph
 *   pl_plan_tensor   input / output i: the plan's own device buffer, its size, element type (0 f32, 1 i32, 2 i64, 3 u8) and shape
 *   pl_plan_run      copies inputs[i] (device pointers; NULL = the caller wrote the plan's buffer itself) in, launches the
 *                    graph, copies the outputs to outputs[i] (device pointers; NULL = read the plan's buffer) -- asynchronous
 *                    on the context's stream like every other call (pl_sync waits)
 *   pl_plan_destroy  frees graph, arena and constants */
int pl_plan_build(pl_ctx *ctx, const void *program, size_t program_bytes, pl_plan **out);
int pl_plan_info(pl_plan *plan, int *n_inputs, int *n_outputs, size_t *arena_bytes, size_t *const_bytes, int *n_calls);
int pl_plan_tensor(pl_plan *plan, int output, int index, void **device_ptr, size_t *bytes, int *dtype, int *ndim, int *dims8);
int pl_plan_run(pl_plan *plan, const void *const *inputs, void *const *outputs);
int pl_plan_destroy(pl_plan *plan);

/* ---- MFMA-bound ops --------------------------------------------------- */
/* layer.Conv2d (layer.py:22-26) == util.conv_for (util.py:17-44) + bias add.
 * Implicit-GEMM: (Cout x Cin/g*kh*kw) @ im2col(x), K ordered (cin,kh,kw) like
 * K.reshape(Cout,-1); output Ho = (H+pt+pb-(kh-1)*dh-1+sh)/sh (util.py:25-26).
 * Requires pt==pb and pl==pr (util.pad, util.py:8) else PL_EUNSUPPORTED.
 * bias may be NULL. */
int pl_conv2d_f32(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W,
                  const float *w, int Cout, int kh, int kw, const float *bias,
                  float *y, int sh, int sw, int dh, int dw, int pt, int pl,
                  int pb, int pr, int group);

/* Conv2d with the layers that follow it folded into the epilogue, used by
 * the compiled plan:  y = act( (conv(x,w)+bias) * scale[c] + shift[c] + res )
 * i.e. conv -> batchnorm (layer.py:125-127) -> add (layer.py:93-95) ->
 * relu/leakyrelu (layer.py:44-51).  scale/shift/res/bias may each be NULL.
 * With act | PL_ACT_RES_AFTER:  y = act( (conv+bias)*scale + shift ) + res. */
int pl_conv2d_fused_f32(pl_ctx *ctx, const float *x, int N, int Cin, int H,
                        int W, const float *w, int Cout, int kh, int kw,
                        const float *bias, float *y, int sh, int sw, int dh,
                        int dw, int pt, int pl, int pb, int pr, int group,
                        const float *scale, const float *shift,
                        const float *res, int act, double alpha, int w_layout);

/* w_layout 0: filters exactly as the reference holds them, OIHW (layer.py:22).
 * w_layout 1: "tap-major" filters [Cout][kh*kw][Cin/g] produced once per model
 * by pl_conv2d_prepare_weights_f32 (needs Cin/g % 16 == 0); lets the kernel
 * compute padding validity once per K-chunk instead of once per element.
 * 1x1 filters are identical in both layouts. */
int pl_conv2d_prepare_weights_f32(pl_ctx *ctx, const float *w, int Cout, int Cin_g,
                                  int kh, int kw, float *out);
/* w_layout 3: Winograd F(2x2,3x3) filters U[16][Cout][Cin] (16*Cout*Cin floats) for 3x3 /
 * stride 1 / pad 1 / group 1 convs with Cin % 16 == 0: input transform + 16 GEMMs (one grouped
 * 1x1 conv on the MFMA kernel) + output transform with the fused tail. */
int pl_conv2d_prepare_winograd_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out);
/* ---- channel-quad ("Q4") activations: the compiled plan's internal layout ----
 * A Q4 tensor holds the reference's (N,C,H,W) array as [N][ceil(C/4)][H][W][4]
 * (channel c -> quad c/4, lane c%4; padding lanes are zero), 16-byte aligned.
 * It exists because on gfx950 one b128 load per (pixel, 4 channels) keeps the
 * fp32 MFMA pipe 13-17 % busier than the four dword loads NCHW needs (DESIGN.md
 * section 4).  Semantics stay layer.Conv2d's (layer.py:22-26, util.py:17-44):
 * pl_q4_to_nchw(pl_conv2d_q4(pl_nchw_to_q4(x))) == pl_conv2d_fused(x) up to
 * fp32 summation order.  Conversions are done by the plan at graph inputs /
 * outputs and around layers that have no Q4 kernel. */
int pl_nchw_to_q4_f32(pl_ctx *ctx, const float *x, float *yq, int N, int C, int HW);
int pl_q4_to_nchw_f32(pl_ctx *ctx, const float *xq, float *y, int N, int C, int HW);
/* Filter for pl_conv2d_q4_f32: OIHW -> wq[group][q][Cout/group][4] with
 * q = tap*ceil(Cin_g/4) + cin/4, zero padded to a multiple of 8 k-quads; made
 * once per model.  `elems` = floats the packed filter occupies. */
int pl_conv2d_q4_filter_elems(int Cout, int Cin_g, int kh, int kw, int group, size_t *elems);
int pl_conv2d_prepare_q4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin_g, int kh, int kw,
                             int group, float *out);
/* pl_conv2d_fused_f32 on Q4 tensors: xq, resq, yq are Q4 (Cin / Cout / Cout
 * channels); bias/scale/shift stay plain per-channel arrays.  group > 1 needs
 * Cin/group and Cout/group to be multiples of 4 (else PL_EUNSUPPORTED). */
int pl_conv2d_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W,
                     const float *wq, int Cout, int kh, int kw, const float *bias,
                     float *yq, int sh, int sw, int dh, int dw, int pt, int pl,
                     int pb, int pr, int group, const float *scale,
                     const float *shift, const float *resq, int act, double alpha);

/* Winograd F(2x2,3x3) on Q4 tensors for 3x3 / stride 1 / pad 1 / group 1 convs with
 * Cin % 4 == 0 and Cout % 4 == 0: uq = [16][k-quad][Cout][4] filters made once per
 * model; float4 input/output transforms around one grouped 1x1 Q4 conv. */
int pl_conv2d_winograd_q4_filter_elems(int Cout, int Cin, size_t *elems);
int pl_conv2d_prepare_winograd_q4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out);
int pl_conv2d_winograd_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W,
                              const float *uq, int Cout, const float *bias, float *yq,
                              const float *scale, const float *shift, const float *resq,
                              int act, double alpha);

/* Row-packed convolution for inputs with 1..3 channels (the 7x7 / 3-channel stem): takes the
 * reference's NCHW input directly, re-lays it as a zero-padded NHWC image (one HBM pass, replaces
 * pl_nchw_to_q4_f32 for this layer) and runs conv_q4_kernel with K = kh rows x ceil(kw*Cin/4)
 * quads instead of kh*kw taps x 1 padded quad (7x7x3: 168 instead of 196 k-values, no range checks
 * in the gather).  wq from pl_conv2d_prepare_rowpack_f32; output and residual are Q4.
 * group 1, dilation 1, symmetric pads. */
int pl_conv2d_rowpack_filter_elems(int Cout, int Cin, int kh, int kw, size_t *elems);
int pl_conv2d_prepare_rowpack_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, int kh, int kw,
                                  float *out);
int pl_conv2d_rowpack_q4_f32(pl_ctx *ctx, const float *x, int N, int Cin, int H, int W,
                             const float *wq, int Cout, int kh, int kw, const float *bias,
                             float *yq, int sh, int sw, int pt, int pl, const float *scale,
                             const float *shift, const float *resq, int act, double alpha);
/* The two halves of pl_conv2d_rowpack_q4_f32, for callers that own the input buffer of a captured plan: the re-layout
 * (x NCHW -> xp, pl_rowpack_input_elems floats, 16-byte aligned) can then BE the copy that brings a new batch into the
 * plan -- planer_amd.net feeds a plan this way instead of copying the batch and re-laying it inside the graph -- and the
 * convolution reads the packed image.  Same kernels, same results as the one-call form (util.py:17-44). */
int pl_rowpack_input_elems(int N, int Cin, int H, int W, int kw, int sw, int pt, int pl, size_t *elems);
int pl_rowpack_input_f32(pl_ctx *ctx, const float *x, float *xp, int N, int Cin, int H, int W, int kw,
                         int sw, int pt, int pl);
int pl_conv2d_rowpacked_q4_f32(pl_ctx *ctx, const float *xp, int N, int Cin, int H, int W,
                               const float *wq, int Cout, int kh, int kw, const float *bias,
                               float *yq, int sh, int sw, int pt, int pl, const float *scale,
                               const float *shift, const float *resq, int act, double alpha);
/* The row-packed stem conv FOLLOWED BY layer.Maxpool(w = 3x3, strides 2, pads 1) (layer.py:71-72 -> util.pool util.py:79-95:
 * zero padding, running maximum from -1e4) in one kernel that writes only the pooled Q4 tensor [N][Cout/4][Hq][Wq][4]
 * (conv_stem_pool_kernel.h: a persistent workgroup marches down an image strip, conv rows live in LDS only).  Built for the
 * stem of an ImageNet-style net -- 3 channels, 7x7 / stride 2 / pad 3, input width 224; _supported says whether a shape
 * qualifies, anything else is PL_EUNSUPPORTED (the plan compiler then keeps the two kernels).  bias / scale / shift are
 * read as 16-byte channel quads; no residual. */
int pl_conv2d_rowpacked_pool_supported(int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int pt,
                                       int pl, int *ok);
int pl_conv2d_rowpacked_pool_q4_f32(pl_ctx *ctx, const float *xp, int N, int Cin, int H, int W,
                                    const float *wq, int Cout, int kh, int kw, const float *bias,
                                    float *yq, int sh, int sw, int pt, int pl, const float *scale,
                                    const float *shift, int act, double alpha);
/* The same kernel reading the reference's NCHW input itself (no row-packed copy of the batch; util.conv_for util.py:17-44 gathers
 * from the padded NCHW tensor too): 3 channels, 7x7 / stride 2 / pad 3, W % 4 == 0, x 16-byte aligned.  wq = the filter in this
 * kernel's k order, [48][Cout][4] floats (_filter_elems) made by pl_conv2d_prepare_stem_nchw_f32 from OIHW [Cout][3][7][7].
 * strip_rows: pooled rows per workgroup strip -- 0 or 7 (one workgroup per CU at batch 32 / 224 px), or 14: half the workgroups
 * running 15 instead of 2 x 8 conv-row pairs, slower alone and cheaper for a pipelined host. */
int pl_conv2d_stem_pool_nchw_supported(int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int pt,
                                       int pl, int *ok);
int pl_conv2d_stem_nchw_filter_elems(int Cout, size_t *elems);
int pl_conv2d_prepare_stem_nchw_f32(pl_ctx *ctx, const float *w, int Cout, float *out);
int pl_conv2d_stem_pool_nchw_q4_f32(pl_ctx *ctx, const float *x, int N, int H, int W, const float *wq, int Cout,
                                    const float *bias, float *yq, const float *scale, const float *shift,
                                    int act, double alpha, int strip_rows);
/* Winograd F(4x4,3x3) on Q4 tensors (same constraints as the F(2x2,3x3) entry points): 6x6 input
 * tiles, 36 grouped GEMMs, 4x fewer multiplies than the direct conv and less transform traffic
 * than F(2x2,3x3); larger transform constants, error a few 1e-6 of max|y| in fp32.
 * uq = [36][k-quad][Cout][4]. */
int pl_conv2d_winograd4_q4_filter_elems(int Cout, int Cin, size_t *elems);
int pl_conv2d_prepare_winograd4_q4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out);
int pl_conv2d_winograd4_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W,
                               const float *uq, int Cout, const float *bias, float *yq,
                               const float *scale, const float *shift, const float *resq,
                               int act, double alpha);

/* The same F(4x4,3x3) pipeline stage by stage, for plans that chain consecutive Winograd convs
 * (replaces util.conv_for, util.py:17-44, for 3x3 / stride 1 / pad 1 convs; the fused tail is
 * layer.BatchNorm / Add / ReLU / LeakyReLU, layer.py:125-127, 93-95, 44-51).
 * V (transformed input) and M (per-frequency products) are [36][C/4][T][4] with
 * T = N * ceil(H/4) * ceil(W/4); pl_wino4_elems gives their size in floats.
 *   pl_wino4_input_q4_f32    xq (N,C,H,W) -> V
 *   pl_wino4_gemm_q4_f32     V (Cin), uq (prepare_winograd4) -> M (Cout): 36 grouped GEMMs on the MFMA kernel
 *   pl_wino4_output_q4_f32   M -> yq = act((A^T m A + bias)*scale + shift + res)
 *   pl_wino4_chain_q4_f32    M -> yq (may be NULL: nobody else reads it) AND Vnext, the transformed input of
 *                            the next 3x3 conv on yq, in one kernel: yq never makes the round trip through HBM
 * pl_wino4_chain_supported: *ok = 1 when an (N,C,H,W) map fits the LDS transform kernel (whole planes per
 * workgroup); otherwise pl_wino4_chain_q4_f32 returns PL_EUNSUPPORTED and input / output use the register
 * kernels.  Results are bit-identical to pl_conv2d_winograd4_q4_f32 whichever kernels run. */
int pl_wino4_elems(int N, int C, int H, int W, size_t *elems);
int pl_wino4_chain_supported(pl_ctx *ctx, int N, int C, int H, int W, int *ok);
int pl_wino4_input_q4_f32(pl_ctx *ctx, const float *xq, int N, int C, int H, int W, float *V);
int pl_wino4_gemm_q4_f32(pl_ctx *ctx, const float *V, int N, int Cin, int H, int W, const float *uq, int Cout, float *M);
int pl_wino4_output_q4_f32(pl_ctx *ctx, const float *M, int N, int C, int H, int W, const float *bias,
                           const float *scale, const float *shift, const float *resq, int act, double alpha,
                           float *yq);
int pl_wino4_chain_q4_f32(pl_ctx *ctx, const float *M, int N, int C, int H, int W, const float *bias,
                          const float *scale, const float *shift, const float *resq, int act, double alpha,
                          float *yq, float *Vnext);

/* Winograd with MIXED tiles for maps whose sides are 7, 14 or 21 pixels (csrc/wino43_kernels.h): a side of 7a is cut into a
 * segments of 4 (F(4,3), 6 frequencies) and a segments of 3 (F(3,3), 5 frequencies), so no tile hangs over the map's edge --
 * F(4x4,3x3) computes a 14x14 map as 16x16 and a 7x7 map as 8x8.  Four tile classes (6x6, 6x5, 5x6, 5x5 frequencies) with equally
 * many tiles each: 121 per-frequency GEMMs of N a^2 columns, one grouped launch.  Same contract as the pl_wino4_* stages /
 * pl_conv2d_winograd4_q4_f32 (3x3 / stride 1 / pad 1 / group 1, Cin and Cout multiples of 4, fused tail); V / M are
 * [121][C/4][N (H/7) (W/7)][4]; replaces the same reference code (layer.py:22-26 -> util.py:17-44). */
int pl_wino43_supported(int H, int W, int *ok);
int pl_wino43_elems(int N, int C, int H, int W, size_t *elems);
int pl_conv2d_winograd43_q4_filter_elems(int Cout, int Cin, size_t *elems);
int pl_conv2d_prepare_winograd43_q4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out);
int pl_wino43_input_q4_f32(pl_ctx *ctx, const float *xq, int N, int C, int H, int W, float *V);
int pl_wino43_gemm_q4_f32(pl_ctx *ctx, const float *V, int N, int Cin, int H, int W, const float *uq, int Cout, float *M);
int pl_wino43_output_q4_f32(pl_ctx *ctx, const float *M, int N, int C, int H, int W, const float *bias, const float *scale,
                            const float *shift, const float *resq, int act, double alpha, float *yq);
int pl_wino43_chain_q4_f32(pl_ctx *ctx, const float *M, int N, int C, int H, int W, const float *bias, const float *scale,
                           const float *shift, const float *resq, int act, double alpha, float *yq, float *Vnext);
int pl_conv2d_winograd43_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *uq, int Cout,
                                const float *bias, float *yq, const float *scale, const float *shift, const float *resq,
                                int act, double alpha);

/* A 1x1 / stride 1 / group 1 channel-quad convolution with its fused tail (bias, scale, shift, ReLU / LeakyReLU; no residual)
 * whose only reader is a staged Winograd 3x3 convolution: writes that conv's transformed input V straight away --
 * wino = 4: F(4x4,3x3), V as pl_wino4_input_q4_f32 makes it (wino = 2, the F(2x2,3x3) domain, is reserved: PL_EUNSUPPORTED).
 * wq from pl_conv2d_prepare_q4_f32 (group 1).  Replaces, for a Darknet block (1x1 then 3x3), layer.Conv2d + BatchNorm +
 * LeakyReLU (reference layer.py:22-26, 125-127, 48-51) and the first stage of the next conv; plan-internal. */
int pl_conv1x1_wino_in_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W, const float *wq, int Cout,
                              const float *bias, const float *scale, const float *shift, int act, double alpha, int wino,
                              float *V);

/* Two channel-quad convolutions that read the SAME input, in one launch (both with the fused tail
 * y = act((conv + bias) * scale + shift), no residual; group 1, dilation 1, symmetric pads; filters from
 * pl_conv2d_prepare_q4_f32).  Replaces two calls of util.conv_for (util.py:17-44) where a graph forks: ResNet's
 * stride-2 3x3 conv and the 1x1 stride-2 projection beside it -- the projection's tiles fill the tail of the
 * sibling's grid and find the pixels it gathers in the L2.  Results equal the two separate launches bit for bit
 * when those run unsplit with the same tile configuration. */
int pl_conv2d_q4_pair_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W,
                          const float *wq_a, int Cout_a, int kh_a, int kw_a, int sh_a, int sw_a, int pt_a, int pl_a,
                          const float *bias_a, const float *scale_a, const float *shift_a, int act_a, double alpha_a, float *yq_a,
                          const float *wq_b, int Cout_b, int kh_b, int kw_b, int sh_b, int sw_b, int pt_b, int pl_b,
                          const float *bias_b, const float *scale_b, const float *shift_b, int act_b, double alpha_b, float *yq_b);

/* Fully fused Winograd F(4x4,3x3) on Q4 tensors (3x3 / stride 1 / pad 1 / group 1, Cin %% 4 == 0, Cout %% 4 == 0;
 * replaces util.conv_for, util.py:17-44, + the fused tail): one workgroup carries 32 tiles x 64 output channels
 * through all 36 frequencies -- input transform into LDS, v_mfma_f32_16x16x4_f32 with the 36 accumulator blocks in
 * registers, lane-local output transform -- so neither the transformed input nor the products ever reach memory.
 * u = filters laid out [Cout/64][Cin/4][36][4][4][16] by pl_conv2d_prepare_wf4_f32.  bias / scale / shift must be
 * 16-byte aligned (read as one 16-byte load per channel quad). */
int pl_conv2d_wf4_filter_elems(int Cout, int Cin, size_t *elems);
int pl_conv2d_prepare_wf4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out);
int pl_conv2d_wf4_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W,
                         const float *u, int Cout, const float *bias, float *yq,
                         const float *scale, const float *shift, const float *resq,
                         int act, double alpha);

/* Fused 1-D Winograd F(4,3) along W on Q4 tensors (3x3 / stride 1 / pad 1 / group 1, Cin %% 4 == 0): 6 frequencies,
 * 4 outputs per tile, 2x fewer multiplies than the direct conv with NO extra HBM traffic -- the input transform happens
 * between the global load and LDS, the output transform in registers (conv_w1d_kernel.h).
 * uq = [6][k-quad][Cout][4], k-quad = row*Cin/4 + cin/4, made once per model. */
int pl_conv2d_w1d4_q4_filter_elems(int Cout, int Cin, size_t *elems);
int pl_conv2d_prepare_w1d4_q4_f32(pl_ctx *ctx, const float *w, int Cout, int Cin, float *out);
int pl_conv2d_w1d4_q4_f32(pl_ctx *ctx, const float *xq, int N, int Cin, int H, int W,
                          const float *uq, int Cout, const float *bias, float *yq,
                          const float *scale, const float *shift, const float *resq,
                          int act, double alpha);

/* HBM-bound layers on Q4 tensors (same semantics as their NCHW namesakes below:
 * util.pool util.py:79-92, layer.UpSample layer.py:80-82, layer.GlobalAveragePool
 * layer.py:77-78, layer.BatchNorm layer.py:125-127).  pl_gap_q4_f32 writes a
 * plain [N][C] array.  Element-wise layers (relu, leakyrelu, add ...) are layout
 * agnostic and run the ordinary kernels on the padded buffer. */
int pl_pool2d_q4_f32(pl_ctx *ctx, const float *xq, float *yq, int N, int C, int H, int W,
                     int kh, int kw, int sh, int sw, int pt, int pl, int pb, int pr,
                     int mode);
int pl_upsample_nearest_q4_f32(pl_ctx *ctx, const float *xq, float *yq, int N, int C,
                               int H, int W, int fh, int fw);
/* layer.Concatenate (axis 1) of two Q4 tensors in one launch; the first is nearest-upsampled by (fh, fw) on the way
 * (layer.UpSample + layer.Concatenate, layer.py:80-82, 90-91): a is (N, Ca, H/fh, W/fw), b (N, Cb, H, W), y (N, Ca+Cb, H, W). */
int pl_concat2_q4_f32(pl_ctx *ctx, const float *aq, const float *bq, float *yq, int N, int Ca, int Cb, int H, int W, int fh,
                      int fw);
int pl_gap_q4_f32(pl_ctx *ctx, const float *xq, float *y, int N, int C, int HW);
int pl_scale_shift_q4_f32(pl_ctx *ctx, const float *xq, float *yq, const float *scale,
                          const float *shift, int N, int C, int HW);

/* First call for a new conv shape times every applicable tile configuration
 * and remembers the fastest (on by default; PLANER_HIP_AUTOTUNE=0 or 0 here
 * selects the static heuristic). Never runs during graph capture. */
int pl_set_autotune(pl_ctx *ctx, int enabled);
/* Persist / restore the tuned plans of a context (text file). */
int pl_tune_cache_save(pl_ctx *ctx, const char *path);
int pl_tune_cache_load(pl_ctx *ctx, const char *path, int *entries);
/* Force one tile configuration for the conv kernel (tuning / tests).
 * cfg < 0 restores the built-in heuristic. split_k <= 0 means automatic. */
int pl_conv2d_set_config(pl_ctx *ctx, int cfg, int split_k);
/* Full launch plan: tiles [0,dp_tiles) run data-parallel with the fused epilogue,
 * the rest as split_k slices + tile reduce; occupancy>0 pins workgroups per CU. */
int pl_conv2d_set_plan(pl_ctx *ctx, int cfg, int dp_tiles, int split_k, int occupancy);
int pl_conv2d_num_configs(void);
/* How the last convolution enqueued on this context was launched -- kernel family, tile
 * configuration, data-parallel tiles / split-K slices / occupancy pin, e.g.
 * "wino4[q64x64x16 dp=1800 split=1 occ=0]".  For run reports (bench.py config.algos). */
int pl_conv2d_last_plan(pl_ctx *ctx, char *buf, size_t len);
/* The GEMM that launch executed on the matrix cores, padding included: ext4 = {groups (Winograd: frequencies),
 * rows, columns, K} with rows / columns rounded up to whole tiles and K to whole chunks; executed FLOPs =
 * 2 * ext4[0] * ext4[1] * ext4[2] * ext4[3] (bench.py roofline.frac). */
int pl_conv2d_last_extents(pl_ctx *ctx, long long *ext4);
/* Launch-plan cache of this context's device: entries held, and how many conv shapes had to be timed
 * (autotuned) by this context because no entry existed (0 = every launch plan came from a loaded cache). */
int pl_tune_stats(pl_ctx *ctx, int *entries, int *misses);
int pl_conv2d_config_name(int cfg, char *buf, size_t len);

/* layer.Dense (layer.py:15-18): y[M,N] = x[M,K] @ w[N,K]^T + bias[N]  (trans_b=1)
 * layer.MatMul (layer.py:20):   y[M,N] = a[M,K] @ b[K,N]              (trans_b=0) */
int pl_gemm_f32(pl_ctx *ctx, const float *a, int M, int K, const float *b,
                int N, int trans_b, const float *bias, float *y);

/* ---- HBM-bound ops ----------------------------------------------------- */
/* layer.BatchNorm (layer.py:125-127): y = x*scale[c] + shift[c], x (outer,C,inner) */
int pl_scale_shift_f32(pl_ctx *ctx, const float *x, float *y, const float *scale,
                       const float *shift, int outer, int C, int inner);
/* layer.ReLU (layer.py:44-46): y = x*(x>0); the reference works in place (y==x) */
int pl_relu_f32(pl_ctx *ctx, const float *x, float *y, size_t n);
/* layer.LeakyReLU (layer.py:48-51): y = x*((x>0)*(1-alpha)+alpha) */
int pl_leakyrelu_f32(pl_ctx *ctx, const float *x, float *y, size_t n, double alpha);
/* layer.Sigmoid (layer.py:61-64): y = 1/(1+exp(-x)) */
int pl_sigmoid_f32(pl_ctx *ctx, const float *x, float *y, size_t n);
/* layer.Add (layer.py:93-95), same-shape and per-channel-broadcast forms */
int pl_add_f32(pl_ctx *ctx, const float *a, const float *b, float *y, size_t n);
int pl_add_channel_f32(pl_ctx *ctx, const float *a, const float *b_c, float *y,
                       int outer, int C, int inner);
/* layer.Maxpool / AveragePool (layer.py:71-75) -> util.pool (util.py:79-100):
 * zero padding, max accumulator initialised to -1e4; avg divides by kh*kw.
 * mode 0 = max, 1 = average. pads must be symmetric. */
int pl_pool2d_f32(pl_ctx *ctx, const float *x, float *y, int NC, int H, int W,
                  int kh, int kw, int sh, int sw, int pt, int pl, int pb, int pr,
                  int mode);
/* layer.UpSample nearest (layer.py:80-82, util.py:184-192): block replication */
int pl_upsample_nearest_f32(pl_ctx *ctx, const float *x, float *y, int NC, int H,
                            int W, int fh, int fw);
/* layer.Concatenate (layer.py:90-91) building block: copy `rows` rows of
 * `width` floats from src (row pitch src_pitch) to dst (row pitch dst_pitch) */
int pl_copy2d_f32(pl_ctx *ctx, float *dst, size_t dst_pitch, const float *src,
                  size_t src_pitch, size_t width, size_t rows);
/* layer.GlobalAveragePool (layer.py:77-78): y[r] = mean(x[r, 0:inner]) */
int pl_gap_f32(pl_ctx *ctx, const float *x, float *y, int rows, int inner);
/* ---- second-wave operators ---------------------------------- */
/* Exp/Log/Tanh/Sqrt/Reciprocal/HardSigmoid/Clip (layer.py:53,66-69,174-186,247-251).
 * op: 0 exp, 1 log, 2 tanh, 3 sqrt, 4 reciprocal, 5 hardsigmoid (p0=alpha,p1=beta),
 * 6 clip (p0=min,p1=max).  y may alias x (Clip works in place in the reference). */
int pl_unary_f32(pl_ctx *ctx, const float *x, float *y, size_t n, int op, double p0, double p1);
/* Add/Sub/Mul/Div/Pow (layer.py:93-111) on a result viewed as (outer,C,inner).
 * op: 0 add, 1 sub, 2 mul, 3 div, 4 pow.  a_mode/b_mode: 0 full-size operand,
 * 1 one value per channel (C), 2 a single value. */
int pl_binary_f32(pl_ctx *ctx, const float *a, const float *b, float *y, int outer, int C,
                  int inner, int op, int a_mode, int b_mode);
/* Add/Sub/Mul/Div/Pow under general numpy broadcasting (layer.py:93-111): `shape` is the
 * broadcast result (1..6 axes), a_stride / b_stride the operands' element strides per
 * result axis, 0 where the operand is broadcast.  op as in pl_binary_f32. */
int pl_binary_bcast_f32(pl_ctx *ctx, const float *a, const float *b, float *y, int ndim,
                        const int *shape, const long long *a_stride, const long long *b_stride,
                        int op);
/* layer.UpSample / Resize, mode "linear", integer factors (util.py:121-153
 * make_upmat + upsample_blinear): `weights` is a HOST table of terms x fh x fw floats
 * (terms = 4 when both factors exceed 1: lt, rt, lb, rb; else 2), fh * fw <= 64. */
int pl_upsample_linear_f32(pl_ctx *ctx, const float *x, float *y, int NC, int H, int W, int fh,
                           int fw, const float *weights);
/* layer.UpSample / Resize, mode "linear", fractional factors (util.py:194-219
 * upsample_size) on (NC, H, W) planes: ra/ca = lower sample row / column per output
 * row / column, rs/cs = the fractions (device arrays), columns first then rows. */
int pl_resize_linear_f32(pl_ctx *ctx, const float *x, float *y, int NC, int H, int W, int OH,
                         int OW, const int *ra, const float *rs, const int *ca, const float *cs);
/* Softmax / LogSoftmax over the last axis (layer.py:141-153) */
int pl_softmax_f32(pl_ctx *ctx, const float *x, float *y, int rows, int cols, int log_softmax);
/* ReduceSum/Mean/Max/Min over the trailing `cols` elements (layer.py:113-123): op 0..3 */
int pl_reduce_f32(pl_ctx *ctx, const float *x, float *y, int rows, int cols, int op);
/* Transpose (layer.py:194): y = x.transpose(perm), up to 6 axes */
int pl_transpose_f32(pl_ctx *ctx, const float *x, float *y, int ndim, const int *shape, const int *perm);
/* General strided map, up to 6 axes: output index o_d reads input index
 * t = o_d*step[d] + start[d] (wrap[d] = 1: taken modulo extent[d], 2: clamped to the
 * axis, 3 / 4: mirrored at its borders without / with the border sample -- np.pad's
 * 'wrap', 'edge', 'reflect', 'symmetric'; div[d] > 1: only
 * where t %% div[d] == 0, then t / div[d]) at in_stride[d] elements per index,
 * and `fill` wherever an axis falls outside [0, extent[d]).  Serves layer.Slice
 * (layer.py:188-196), layer.Pad in those five modes (:241-245), Tile (:57), Expand
 * (:198-200), Split (:170-172) and the zero-stuffing + filter flip/transpose of
 * layer.ConvTranspose2d (:28-34). */
int pl_strided_map_f32(pl_ctx *ctx, const float *x, float *y, int ndim, const int *out_shape,
                       const long long *in_stride, const int *start, const int *step,
                       const int *div, const int *extent, const int *wrap, double fill);
/* ---- operators of ONNX-exported detection heads (layer.py:155-157, 200-234, 253-258) -------------
 * pl_compare_f32: layer.Equal / Greater / GreaterOrEqual (op 0 / 1 / 2) -> bool bytes; *_one: that operand is one value.
 * pl_where_f32:   layer.Where, np.where(mask, a, b) with bool-byte mask.
 * pl_cast:        layer.Cast between 0 float32, 1 int32, 2 int64, 3 bool (numpy astype: truncation, != 0).
 * pl_gather_f32:  layer.Gather, np.take(x, idx, axis) on x viewed (outer, axis_len, inner); idx int32, negatives wrap.
 * pl_erf_lut_f32: layer.Erf -- clamps x IN PLACE like the reference and looks y up in its 1025-entry table `lut`.
 * pl_instancenorm_f32: layer.InstanceNormalization on (rows = N*C, inner) IN PLACE, scale/bias per channel.
 * pl_scatter_rows_f32: the device side of layer.Scatternd (layer.py:208-212): dst row dst_row[j] (rows of
 *                 row_len floats) = src row src_row[j]; the caller resolves index tuples to rows and keeps
 *                 only the last write to each row (the reference applies updates in order).
 * pl_nonzero_count / pl_nonzero_write: layer.NonZero (layer.py:230) = np.array(np.nonzero(x)) in two steps,
 *                 because the result's shape depends on the data: _count fills `scratch`
 *                 (ceil(n / PL_NONZERO_BLOCK) + 1 int64, device) and returns the number of non-zero elements in
 *                 *total (host; SYNCHRONISES the stream); _write fills out (ndim x total, int64, row-major
 *                 coordinates in ascending flat order).  elem_type as pl_cast.
 * pl_topk_f32:    layer.TopK (layer.py:234-239) on x viewed (outer, n, inner) along the middle axis: values /
 *                 int64 indices (outer, k, inner).  largest = 1: the k greatest, descending; largest = 0: k
 *                 copies of the smallest (the reference's index list is arange(k)*0).  NaN sorts last like
 *                 numpy; ties by ascending index (numpy leaves them unspecified).
 * pl_lstm_cell_f32: one time step of util.lstm (util.py:109-118) after the GEMMs: gates_x = x_t W^T,
 *                 gates_h = h R^T, both (N, 4H) in ONNX i|o|f|c order, bias (8H) = Wb | Rb; writes h, c (N, H). */
#define PL_NONZERO_BLOCK 2048
int pl_compare_f32(pl_ctx *ctx, const float *a, const float *b, unsigned char *y, size_t n, int op, int a_one, int b_one);
int pl_where_f32(pl_ctx *ctx, const unsigned char *mask, const float *a, const float *b, float *y, size_t n,
                 int a_one, int b_one);
int pl_cast(pl_ctx *ctx, const void *src, void *dst, size_t n, int src_type, int dst_type);
int pl_gather_f32(pl_ctx *ctx, const float *x, const int *idx, float *y, int outer, int axis_len, int inner, int n_idx);
int pl_erf_lut_f32(pl_ctx *ctx, float *x, const float *lut, float *y, size_t n);
int pl_instancenorm_f32(pl_ctx *ctx, float *x, const float *scale, const float *bias, int rows, int C, int inner,
                        double eps);
int pl_scatter_rows_f32(pl_ctx *ctx, float *dst, const long long *dst_row, const float *src, const int *src_row,
                        int n_rows, int row_len);
int pl_nonzero_count(pl_ctx *ctx, const void *x, size_t n, int elem_type, long long *scratch, long long *total);
int pl_nonzero_write(pl_ctx *ctx, const void *x, size_t n, int elem_type, const long long *scratch,
                     const long long *shape, int ndim, long long *out, long long total);
int pl_topk_f32(pl_ctx *ctx, const float *x, int outer, int n, int inner, int k, int largest, float *values,
                long long *indices);
int pl_lstm_cell_f32(pl_ctx *ctx, const float *gates_x, const float *gates_h, const float *bias, const float *c_prev,
                     float *h, float *c, int N, int H);
/* ---- tiled large-image inference: the device side of util.tile (util.py:291-348) ----
 * pl_resize_hwc_f32: util.resize (util.py:253-269) on an H x W x C image; ra/rs (OH entries) and
 * ca/cs (OW entries) are the integer sample rows/columns and their fractions, device arrays
 * computed by the host exactly as the reference does (float32 linspace, clip, floor).
 * pl_tile_accumulate_f32: one window's result (h x w x C) into the blend buffers at (r0, c0):
 * buf += rst * wt, count += wt with wt = min(distance to the window border, margin) + 1
 * (util.py:327-343).  pl_tile_normalise_f32: buf /= count (util.py:344). */
int pl_resize_hwc_f32(pl_ctx *ctx, const float *x, float *y, int H, int W, int C, int OH, int OW,
                      const int *ra, const float *rs, const int *ca, const float *cs);
int pl_tile_accumulate_f32(pl_ctx *ctx, const float *rst, float *buf, float *count, int h, int w,
                           int C, int r0, int c0, int OH, int OW, int margin);
int pl_tile_normalise_f32(pl_ctx *ctx, float *buf, const float *count, int OH, int OW, int C);
/* split-K combine + epilogue (internal to conv, exported for tests) */
int pl_splitk_reduce_f32(pl_ctx *ctx, const float *ws, int splits, float *y,
                         int N, int C, int inner, const float *bias,
                         const float *scale, const float *shift,
                         const float *res, int act, double alpha);

/* ---- multi-GPU (RCCL over xGMI): one process per GPU -------------------- */
/* The reference has no distributed code.  The forward pass shards by batch
 * with no collectives; the ONE exchange is the weight blob broadcast at load
 * time (net.load_weights, net.py:83-88). */
#define PL_UNIQUE_ID_BYTES 128
int pl_comm_unique_id(void *id_out);                /* rank 0 */
int pl_comm_init_rank(pl_ctx *ctx, int world, int rank, const void *id);
int pl_comm_bcast(pl_ctx *ctx, void *buf, size_t bytes, int root);
int pl_comm_allreduce_max_f32(pl_ctx *ctx, float *buf, size_t n); /* in place, device */
int pl_comm_allgather(pl_ctx *ctx, const void *send, void *recv, size_t bytes_per_rank);
/* what RCCL itself says about the communicator: ncclCommCount / ncclCommUserRank (run reports: config.rccl_ranks) */
int pl_comm_info(pl_ctx *ctx, int *ranks, int *rank);
int pl_comm_destroy(pl_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* PLANER_HIP_H */
