/*
 * st_amd.h - C ABI of libst_amd.so, the MI355X (gfx950) implementation of the per-iteration hot path
 * of crowsonkb/style-transfer-pytorch.
 *
 * The reference has no FFI layer: its hot path sits behind the Python class `StyleTransfer`
 * (reference style_transfer/style_transfer.py:309-499).  This header is the seam a maintainer binds
 * from that class (ctypes).  Every entry point below names the reference code it
 * replaces.  Conventions:
 *   - plain C, no C++/torch types; all tensors are raw DEVICE pointers to dense fp32, batch 1,
 *     channel-major [C][H][W] (the memory of a contiguous torch NCHW tensor with N == 1);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are asynchronous
 *     on that stream unless stated otherwise;
 *   - return value 0 = success, non-zero = failure with text available from st_last_error();
 *     no C++ exception crosses the boundary;
 *   - handles are opaque and owned by the caller (create/destroy pairs); not thread-safe per handle.
 */
#ifndef ST_AMD_H
#define ST_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define ST_AMD_ABI_VERSION 2

/* VGG-19 `features` indices of the taps (style_transfer.py:316-317). */
#define ST_NUM_CONVS 13
#define ST_NUM_STYLE_LAYERS 5
#define ST_CONTENT_LAYER 22
#define ST_NUM_LOSS_TERMS 7 /* SumLoss order: content, style relu1_1..relu5_1, tv (style_transfer.py:455) */

enum st_pooling { ST_POOL_MAX = 0, ST_POOL_AVERAGE = 1, ST_POOL_L2 = 2 }; /* style_transfer.py:21-22 */

typedef struct st_net st_net;   /* frozen VGG-19 trunk weights, pre-arranged for the kernels      */
typedef struct st_plan st_plan; /* every device buffer for one image size ("one handle per scale") */

/* Text of the last failure on the calling thread ("" if none). */
const char* st_last_error(void);
int st_abi_version(void);
/* Name of the gfx target the kernels were compiled for ("gfx950"). */
const char* st_compiled_arch(void);
/* Build flavour: 1 when the library was built with `build.py --experiments` - it then also holds the code that no default path
 * executes (persistent Newton-Schulz chain kernel, Winograd convolution = conv precision code 5, reproducer and measurement-only
 * kernels) and reads EVERY ST_* switch from the environment; 0 for the default build (the hot path and nothing else). */
int st_has_experiments(void);
/* The ST_* switches a default build reads from the environment (the documented ones, tools/README.md): fills
 * names[0 .. min(count, capacity)) with static strings and returns count.  Every other switch answers to st_set_option only. */
int st_env_switches(const char** names, int capacity);
/*
 * Run-time override of one of the library's ST_* switches (the same names as the environment variables the
 * A/B experiments use: ST_CONV_PC, ST_CONV_PC_XL, ST_NS_FULL_BACKWARD, ...).  Overrides win over the
 * environment; clear != 0 removes the override again.  No counterpart in the reference: it exists so that the
 * parity tests can compare two kernel variants on identical operands inside one process (e.g. the
 * producer / consumer convolution against the single-role kernel, bit for bit).  Not thread-safe with respect
 * to launches in flight on other host threads.
 */
int st_set_option(const char* name, int value, int clear);

/*
 * VGGFeatures.__init__ (style_transfer.py:24-49): truncated vgg19.features[:30], conv1_1 with
 * replicate padding (:39), pooling flavour (:41-46), weights frozen (:48-49).
 * weights[i]: device ptr [Cout][Cin][3][3]; biases[i]: device ptr [Cout]; i = 0..12 in network order.
 * The arrays of pointers themselves live in host memory.  Synchronous (returns after the re-layout).
 */
int st_net_create(st_net** out, const float* const* weights, const float* const* biases, int pooling);
/*
 * As st_net_create, with the arithmetic of the twelve 3x3 trunk convolutions (forward and data gradient):
 *   0 = exact fp32 MFMA (default; bitwise an fp32 FMA chain),
 *   3 = "bf16x6": fp32 operands as three bf16 planes, six bf16 MFMA products, fp32 accumulation (fp32-class accuracy),
 *   2 = "bf16x3": two planes, three products (per-product relative error <= ~2^-16),
 *   4 = "fp16x3": two fp16 planes (22 significant bits: residual <= 2^-24 |x| with round-to-nearest), three
 *       fp16 MFMA products, fp32 accumulation; both operands are pre-scaled by a power of two taken from the
 *       tensor's max |x| (measured on device before each launch), undone exactly in the epilogue: fp32-class
 *       accuracy at half the matrix work of bf16x6.
 * Everything else (conv1_1, Gram, sqrtm chains, losses, optimiser) is fp32 in every mode.
 */
int st_net_create_ex(st_net** out, const float* const* weights, const float* const* biases, int pooling,
                     int conv_precision);
/* fp16x3 networks (conv_precision 4): which of the 13 convolutions the dynamic-range guard moved to bf16x6 (three bf16
 * planes, no per-tensor scale) - forward13[i] / backward13[i] = 1 for conv i's forward / data gradient.  A layer is
 * flagged when one of its input (forward) / output (data gradient) channels carries weights more than 2^8 above the
 * layer's median channel: the signature of weights that compensate a tiny-valued operand channel, which two fp16
 * planes under one scale per tensor cannot hold.  All zero for normalised weights (the synthetic ones included). */
int st_net_wide_layers(const st_net* net, int* forward13, int* backward13);

/* Activation-aware part of the same guard (round 4; no reference counterpart - the reference computes in fp32).  On the
 * cold path, once per scale and image: every unflagged trunk convolution of `plan`'s network is evaluated in bf16x6 on
 * exactly the operand the shipped fp16x3 pass feeds it - forward on the maps of a forward pass over `image`
 * ([3][H][W], the plan's size), data gradient on the gradients of one closure (needs the plan's targets; without them
 * only the forward is checked) - and a layer whose fp16x3 result differs by more than 1e-5 rel-L2 over the map, or on any
 * output channel by more than 16 x what the exact-fp32 kernel differs there, runs bf16x6 from then on (sticky for the NETWORK; st_net_wide_layers shows
 * the union).  forward13 / backward13 receive the layers flagged by THIS call.  Synchronous; unsharded plans. */
int st_plan_range_guard(st_plan* plan, const float* image, int* forward13, int* backward13, void* stream);
/* Adds layers to the network's bf16x6 set from outside (flags only ever accumulate).  Strip-sharded runs: every rank runs the
 * guard on its own rows and the ranks take the union, so that all of them compute a layer in the same arithmetic. */
int st_net_mark_wide(st_net* net, const int* forward13, const int* backward13);
int st_net_destroy(st_net* net);

/* Buffers for an H x W image.  VGGFeatures.forward's size check (:81-83): fails if min(H, W) < 16. */
int st_plan_create(st_plan** out, const st_net* net, int height, int width);
int st_plan_destroy(st_plan* plan);
/* Bytes of device memory held by the plan (for STIterate.gpu_ram style reporting). */
long long st_plan_device_bytes(const st_plan* plan);

/*
 * VGGFeatures.forward (style_transfer.py:78-90), forward only, up to and including `last_layer`
 * (a features index 1..29).  `image` is [3][H][W] in [0,1] (un-normalised; Normalize is fused, :85).
 */
int st_plan_forward(st_plan* plan, const float* image, int last_layer, void* stream);
/* Borrow a tap (any ReLU/pool index computed by the last forward): dense [C][h][w] device pointer. */
int st_plan_feature(const st_plan* plan, int layer, const float** data, int* channels, int* height, int* width);

/*
 * StyleLossW2.get_target (style_transfer.py:162-168) on a tap of the last forward:
 * mean_out[C] = spatial mean, srm_out[C*C] = F F^T / (h*w).  `layer` in {1,6,11,20,29}.
 */
int st_plan_moments(st_plan* plan, int layer, float* mean_out, float* srm_out, void* stream);

/* ContentLossMSE target (style_transfer.py:425-429): copies feat [512][H/8][W/8] into the plan. */
int st_plan_set_content_target(st_plan* plan, const float* feat, void* stream);
/*
 * StyleLossW2.__init__ (style_transfer.py:152-160) for style tap `index` (0..4 = relu1_1..relu5_1):
 * cov = srm - mean mean^T + 1e-4 I, cov_sqrt = sqrtm_ns(cov, 12) (sqrtm.py:9-25).
 * mean[C], srm[C*C] are the (already blended, :442-450) targets.
 */
int st_plan_set_style_target(st_plan* plan, int index, const float* mean, const float* srm, void* stream);
/* Scale factors (style_transfer.py:320-322,366,376,429,453): content, 5 style layers, tv. */
int st_plan_set_loss_weights(st_plan* plan, float content_weight, const float* style_layer_weights, float tv_weight);

/*
 * closure() (style_transfer.py:472-476): forward, SumLoss, backward to the pixels.
 * grad_out [3][H][W]; losses_out: DEVICE array of 8 floats = 7 weighted terms (SumLoss order) + total.
 */
int st_plan_loss_and_grad(st_plan* plan, const float* image, float* grad_out, float* losses_out, void* stream);

/*
 * One full iteration of the hot loop (style_transfer.py:479-486) on caller-owned state:
 *   closure; torch.optim.Adam single-tensor step (betas, eps as given, no weight decay / amsgrad,
 *   bias corrections in double from `step`, torch/optim/adam.py:414-547); image.clamp_(0,1) (:485);
 *   EMA.update (:250-253) with decay `ema_decay` rounded to fp32.
 * `step` is the 1-based Adam step number of THIS update.  All four tensors are [3][H][W], updated in
 * place.  losses_out as above (may be NULL).
 */
int st_plan_step(st_plan* plan, float* image, float* exp_avg, float* exp_avg_sq, float* ema_value,
                 long long step, double lr, double beta1, double beta2, double eps, double ema_decay,
                 float* losses_out, void* stream);

/*
 * Only the optimiser update of st_plan_step (Adam + clamp + EMA) on an externally supplied gradient
 * (used by the strip-sharded driver, where the closure is run phase by phase).
 */
int st_plan_apply_update(st_plan* plan, float* image, const float* grad, float* exp_avg, float* exp_avg_sq,
                         float* ema_value, long long step, double lr, double beta1, double beta2, double eps,
                         double ema_decay, void* stream);

/*
 * Spatial strip sharding (replaces the reference's 2-device layer split, style_transfer.py:326-333).
 * A strip plan owns image rows [row_begin, row_end) of a global_height x width image; row_begin and
 * row_end must be multiples of 16 (row_end may instead equal global_height).  Its closure is a sequence
 * of compute phases separated by exchanges that the caller performs between st_plan_closure_next calls:
 *   kind 1 (halo): send `count` floats at send_up to the rank above (it receives them at ITS recv_down)
 *                  and send_down to the rank below (ITS recv_up); NULL pointers = no neighbour there (the message is
 *                  one boundary row of every channel plus a 16-float trailer that carries the sender's max |row| for the
 *                  receiver's fp16x3 operand scale - opaque to the transport);
 *   kind 2 (all-reduce): sum `count` floats at `buffer` over all ranks, in place;
 *   kind 4 (reduce): sum `count` floats at `buffer` over all ranks into rank `root`'s buffer (the others' contents
 *                  are undefined afterwards);   kind 5 (broadcast): rank `root`'s `buffer` to every rank;
 *   kind 3: nothing to exchange, call again;   kind 0: closure finished.
 * Ordering (ABI version 2): the exchange must be ordered on HIP stream `stream` - behind everything enqueued on it so
 * far, ahead of everything enqueued on it later - or, when `stream` is NULL, on the stream passed to
 * st_plan_closure_next.  The plan's own events tie those streams to the compute stream: a halo exchange travels on a
 * communication stream while the interior rows of the convolution that consumes it are computed (a halo exchange whose
 * consumer is ONE launch has nothing to overlap with and names no stream: it is issued in line, between the kernel that
 * packed the rows and the consumer - round 6), a style head's reduce / broadcast on that head's side stream.  A transport that is not stream-ordered (host-synchronous, or the
 * single-process emulation) may instead complete every exchange before it calls st_plan_closure_next again.
 * `channel`: exchanges on different channels (0 trunk, 1 style heads) must not be serialised against each other by
 * the transport (separate communicators), or a head's broadcast would hold back the trunk's halos.
 * All pointers are device memory owned by the plan, valid until the plan is destroyed.
 */
typedef struct st_exchange {
    int kind;
    long long count;
    float* send_up;
    float* send_down;
    float* recv_up;
    float* recv_down;
    float* buffer;
    int root;
    int channel;
    void* stream;
} st_exchange;
int st_plan_create_strip(st_plan** out, const st_net* net, int global_height, int width, int row_begin,
                         int row_end);
/* Position of this strip among the ranks (default 0 of 1).  With world > 1 each style head's C x C work - covariance,
 * both Newton-Schulz chains (sqrtm.py:9-47), d cov - runs on ONE owner rank ((4 - head) % world) and (Ssym, b, loss
 * term) are broadcast; with world == 1 every plan runs every chain on the all-reduced moments. */
int st_plan_set_rank(st_plan* plan, int rank, int world);
int st_plan_closure_begin(st_plan* plan, const float* image, float* grad_out);
int st_plan_closure_next(st_plan* plan, st_exchange* exchange, void* stream);

/* ---- in-library transport (round 4; csrc/st_fabric.hip): the exchanges of the phase machine issued as RCCL operations by
 * the library itself, on the streams the descriptors name.  Replaces the `.to(devices[i])` transfers of
 * style_transfer.py:87,208 like the descriptor form above, without a Python round trip and a torch.distributed call per
 * exchange, and with the RCCL kernels on the library's own (probed) communication / head streams - c10d launches them on an
 * internal stream that ROCm may deal to the trunk's hardware queue (measured: nothing overlapped).
 * st_fabric_unique_id: 128 opaque bytes from ncclGetUniqueId - call it TWICE on rank 0 (trunk channel, heads' channel)
 * and hand both to every rank (e.g. torch.distributed.broadcast of a byte tensor over any backend).
 * st_fabric_create: ncclCommInitRank of the two communicators (collective over the `world` ranks; the calling thread's
 * current HIP device is the rank's GPU).  self_halo = 1 (world must be 1): the rank is its own upper and lower neighbour -
 * the point-to-point path on one GPU, for tests and measurements.
 * st_plan_closure_run: after st_plan_closure_begin (or st_plan_forward_begin), every remaining phase of the sequence and
 * every exchange between them, enqueued in ONE call; `stream` as in st_plan_closure_next. */
typedef struct st_fabric st_fabric;
int st_fabric_unique_id(unsigned char* id128);
int st_fabric_create(st_fabric** out, const unsigned char* id_trunk128, const unsigned char* id_heads128, int rank, int world,
                     int self_halo);
int st_fabric_destroy(st_fabric* fabric);
/* The same for a fabric whose operations may never complete (a rank left the run early): ncclCommAbort, no waiting. */
int st_fabric_abort(st_fabric* fabric);
/* Pre-flight of a fresh fabric: every operation kind of the phase machine (neighbour send / recv group, all-reduce, reduce,
 * broadcast) once per communicator on 4-float messages with known answers, enqueued on `stream`, awaited on the HOST with a
 * deadline.  0 = the transport works; otherwise st_last_error() says which operation gave what, or that nothing completed
 * within `timeout_ms` (the fabric is then only good for st_fabric_destroy, which aborts its communicators).  Collective:
 * every rank calls it.  No reference counterpart (the reference's `.to(device)` cannot hang). */
int st_fabric_selftest(st_fabric* fabric, void* stream, int timeout_ms);
/* After st_plan_closure_begin (or st_plan_forward_begin): every remaining phase of the plan and every exchange between them,
 * enqueued in one call - the loop over st_plan_closure_next with the exchanges issued on `fabric`.  Operations of one
 * communicator must not run concurrently: the call refuses a stream layout in which the heads' exchanges name different
 * streams (ST_STREAMS_COMPACT=0; use the descriptor form there). */
int st_plan_closure_run(st_plan* plan, st_fabric* fabric, void* stream);
/* Device array of 8 floats (7 weighted terms + total) written by the closure of this plan. */
int st_plan_losses(st_plan* plan, float** losses);
/* Target construction on strips: forward phases only (halo exchanges), then per-layer raw moment sums. */
int st_plan_forward_begin(st_plan* plan, const float* image, int last_layer);
/* Raw (un-normalised) moments of a style tap of this strip: sums[C*C + C] = [F F^T | F 1] over local pixels.
 * All-reduce them and divide by the global pixel count to obtain (srm, mean). */
int st_plan_moment_sums(st_plan* plan, int layer, float* sums, void* stream);

/*
 * The closure's ~430 launches are captured into a hipGraph the second time it is called with the same
 * (image, grad, losses) pointers and replayed afterwards.  Default OFF (measured slower than eager
 * multi-stream launches on ROCm 7.2); 1 = enable, 0 = always launch eagerly.
 */
int st_plan_set_graph(st_plan* plan, int enable);

/*
 * Measurement hooks (bench.py `roofline`): when enabled, every launch of the 3x3 trunk convolution (forward and data
 * gradient; its split-K reduce pass included) is bracketed by hipEvents on its stream.  st_plan_profile_read blocks until the recorded
 * events have completed and returns accumulated {launches, milliseconds, algorithmic FLOPs}.
 */
int st_plan_profile_enable(st_plan* plan, int enable);
int st_plan_profile_read(st_plan* plan, long long* launches, double* millis, double* flops);
/* The same for the step's HBM-bound kernels (bench.py `roofline_hbm`): category 0 conv1_1 forward (+ Normalize),
 * 1 conv1_1 data gradient (+ pad-ring fold), 2 max-pool backward, 3 Adam + clamp + EMA, 4 TV loss + gradient,
 * 5 relu1_1 Gram + mean, 6 content MSE + gradient, 7 the style heads' 1x1 gradient step dF = Ssym F + b (all five taps).
 * Returns {launches, milliseconds, algorithmic bytes} accumulated since
 * the last st_plan_profile_read (call this first: st_plan_profile_read recycles the events). */
int st_plan_profile_read_hbm(st_plan* plan, int category, long long* launches, double* millis, double* bytes);

/* Standalone operators exported for kernel-level parity tests (same code the plan uses). */
/* sqrtm_ns (sqrtm.py:9-25): root[n*n] = NS-12 square root of a[n*n]; n in {64,128,256,512}. */
int st_op_sqrtm_ns(const float* a, float* root, int n, void* stream);
/* _MatrixSquareRootNSLyap.backward (sqrtm.py:36-47): grad_a from root and grad_root. */
int st_op_sqrtm_ns_backward(const float* root, const float* grad_root, float* grad_a, int n, void* stream);
/* The same backward for grad_root = grad_diag * I - the case of StyleLossW2 (style_transfer.py:180: the loss
 * depends on the root through its trace only).  This is the code path the plan runs: the reduced recurrence
 * (the commutator of sqrtm.py:44 vanishes) and, for n = 512, fp16x3 products (csrc/st_nsgemm.hip). */
int st_op_sqrtm_ns_backward_diag(const float* root, float grad_diag, float* grad_a, int n, void* stream);
/* TVLoss (style_transfer.py:187-195) value (device scalar, unweighted) and gradient [3][H][W]. */
int st_op_tv_loss(const float* image, int height, int width, float* loss_out, float* grad_out, void* stream);
/* 3x3 stride-1 zero-padded convolution + bias (+ReLU): the K3/K4 kernel on arbitrary tensors.
 * weight [Cout][Cin][3][3] torch layout (re-laid-out internally, synchronous). Cin % 8 == 0, Cout % 64 == 0. */
int st_op_conv3x3(const float* in, const float* weight, const float* bias, float* out, int cin, int cout,
                  int height, int width, int relu, int precision, void* stream);
/* Data gradient of the same convolution: grad_in[Cin][H][W] from grad_out[Cout][H][W]; if relu_out is
 * non-NULL the incoming gradient is first masked by (relu_out > 0) (threshold_backward). */
int st_op_conv3x3_dgrad(const float* grad_out, const float* relu_out, const float* weight, float* grad_in,
                        int cin, int cout, int height, int width, int precision, void* stream);
/* The same convolution (dgrad == 0: forward with bias / ReLU; dgrad != 0: data gradient, `in` = grad_out with
 * `cout` channels, bias and relu ignored) on a ROW STRIP of a taller tensor, as the strip-sharded plan runs it
 * (SURVEY.md 8(e)): rows -1 and `height` of the operand come from `halo` = [2][C][W] (the neighbour's last row, then
 * the other neighbour's first row; C = the operand's channel count); has_up / has_down == 0 means that side is the
 * global border (zero padding, the halo row is not read). */
int st_op_conv3x3_strip(const float* in, const float* halo, int has_up, int has_down, const float* weight,
                        const float* bias, float* out, int cin, int cout, int height, int width, int relu, int dgrad,
                        int precision, void* stream);
/* The same with the epilogue options of the plan's data-gradient launches - accumulate != 0: out += result (out is
 * read), out_mask (optional [C_out][H][W]): out = out_mask > 0 ? out : 0 afterwards - and, with overlap != 0, in the
 * strip plans' TWO-launch form: the interior rows first (they read no halo row; in a plan the halo exchange is in
 * flight meanwhile), then the first and last rows in one launch behind it (csrc/st_conv_pc.hip, overlap_part).  Fails
 * when the producer / consumer kernel does not take the problem or the strip is too short to cut. */
int st_op_conv3x3_strip_ex(const float* in, const float* halo, int has_up, int has_down, const float* weight,
                           const float* bias, float* out, const float* out_mask, int cin, int cout, int height, int width,
                           int relu, int dgrad, int accumulate, int overlap, int precision, void* stream);

/* 1x1 convolution + bias over [Cin][npix] -> [Cout][npix], weight [Cout][Cin] (no re-layout): the style heads'
 * gradient step dF = Ssym F + b 1^T, i.e. the backward of the einsum / mean in StyleLossW2.get_target
 * (style_transfer.py:162-168).  precision 0 = exact fp32 MFMA, 4 = fp16x3 (operand bounds measured on the
 * device first; the plan gets them from the producing kernels).  Cin % 32 == 0, Cout % 64 == 0.  Synchronous. */
int st_op_conv1x1(const float* in, const float* weight, const float* bias, float* out, int cin, int cout,
                  long long npix, int precision, void* stream);

/* ==================================================================================================================
 * MEASUREMENT AIDS - not part of the drop-in surface (no reference counterpart; a binding of the reference does not
 * need them).  They time the product's own kernels in isolation for tools/ and profiles/: st_op_sqrtm_time,
 * st_op_conv3x3_time, st_op_mfma_rate, st_op_mfma_valu_rate, st_op_grid_barrier_time (and the st_plan_profile_* hooks
 * above, which bench.py's `roofline` uses).
 * ================================================================================================================== */

/* Microbenchmark of the two 12-step recurrences on an n x n SPD matrix (workspace preallocated, HIP events
 * on `stream`): average microseconds per full sqrtm_ns forward chain and per Lyapunov backward chain. */
int st_op_sqrtm_time(int n, int iters, double* fwd_us, double* bwd_us, void* stream);

/* Kernel microbenchmark: average microseconds of `iters` back-to-back launches of the convolution
 * (forward if dgrad == 0, masked data gradient otherwise) on device-resident random operands, timed with
 * HIP events on `stream`; same launch path (tile choice, split-K) as the plan uses. */
int st_op_conv3x3_time(int cin, int cout, int height, int width, int dgrad, int precision, int iters,
                       double* avg_us, void* stream);

/* Measurement aid (csrc/st_diag.hip), no reference counterpart: the rate the 16-bit matrix pipe sustains on this
 * chip under the consumer pattern of the XL convolution tile - one persistent workgroup of `waves` waves per CU, every
 * wave `steps` times {lds_reads (0, 4 or 8) ds_read_b128 operand fetches; 12 v_mfma_f32_32x32x16_f16} - as
 * TFLOP/s (HIP events over `launches` launches on `stream`) and the shader clock it held (MHz).  The roofline in
 * bench.py is quoted against the nominal 2.5 PFLOP/s; this is the measured ceiling next to it. */
int st_op_mfma_rate(int lds_reads, int waves, int steps, int launches, double* tflops, double* mhz, void* stream);

/* The same with `valu_waves` extra waves per CU that execute nothing but fp32 VALU work (32 v_fma_f32 per step,
 * `valu_steps` steps; valu_prio bit 1: at s_setprio 3, bit 2: as the workgroup's first = oldest waves) - the position of the convolution tile's staging waves.  cycles[0] =
 * shader cycles per VALU instruction that wave achieved beside the MFMA streams, cycles[1] = cycles per MFMA of an MFMA
 * wave. */
int st_op_mfma_valu_rate(int lds_reads, int waves, int steps, int launches, int valu_waves, int valu_steps,
                         int valu_prio, double* tflops, double* mhz, double* cycles, void* stream);

/* Measurement aid (csrc/st_diag.hip, profiles/r05_winograd.md), no reference counterpart: the 16-bit matrix rate (TFLOP/s of
 * executed MFMA work) a CU sustains in the consumer pattern of a Winograd F(2x2, 3x3) fp16x3 tile - four waves per CU, per step
 * 16 ds_read_b128 of fresh operands for 12 v_mfma_f32_32x32x16_f16 (0.75 MFMAs per read; the shipped direct tile: 1.5). */
int st_op_winograd_consumer_rate(int steps, int launches, double* tflops, void* stream);

/* Diagnostic (tests/test_tv_hazard_gpu.py), no reference counterpart: after a device synchronise, copy `count` floats of an
 * internal buffer of the plan to the host.  what = 0: the TV kernels' per-workgroup partial sums (4 floats per workgroup:
 * the sums of squares of D1 .. D4 of style_transfer.py:189-192, tv_interior_kernel's workgroups first). */
int st_plan_debug_read(st_plan* plan, int what, float* out, int count);

/* Measurement aid (csrc/st_diag.hip), no reference counterpart: microseconds per round of `rounds` device-wide barriers
 * inside ONE launch of `workgroups` co-resident workgroups (<= the CU count), each round writing `payload_floats` floats
 * per workgroup before the barrier and checking another workgroup's (another XCD's) after it; *errors counts stale reads.
 * groups = 0: one counter for all workgroups; groups = G: two levels (workgroup w arrives at group w % G, the last of a
 * group at the top counter, release through per-group flags) - the form the persistent Newton-Schulz chain kernel uses. */
int st_op_grid_barrier_time(int workgroups, int rounds, int payload_floats, int groups, double* us_per_round, int* errors,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ST_AMD_H */
