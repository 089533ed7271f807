// Shared declarations for the gfx950 kernels of libst_amd.so.  CDNA4 only: 64-lane wavefronts,
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 cycles/SIMD) as the contraction primitive.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <string>

namespace st {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kWave = 64;

// ---- error plumbing (no exceptions across the C ABI) -------------------------------------------
void set_error(const char* fmt, ...);
const char* get_error();

#define ST_HIP(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            st::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                          __LINE__);                                                           \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

#define ST_LAUNCH_CHECK()                                                                      \
    do {                                                                                       \
        hipError_t _e = hipGetLastError();                                                     \
        if (_e != hipSuccess) {                                                                \
            st::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, \
                          __LINE__);                                                           \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

#define ST_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            st::set_error(__VA_ARGS__); \
            return 1;                  \
        }                              \
    } while (0)

// ---- build flavours -----------------------------------------------------------------------------
// The default libst_amd.so holds the hot path and nothing else.  `build.py --experiments` (-DST_EXPERIMENTS) adds the code
// that no default path executes - the persistent Newton-Schulz chain kernel (st_nschain.hip), the Winograd convolution
// (st_conv_wino.hip), the TV hazard's reproducer kernels, measurement-only kernels - and lets EVERY ST_* switch be set from the
// environment; in the default build the environment reaches only the documented switches (kEnvSwitches, tools/README.md), the
// others only answer to st_set_option (parity tests compare kernel variants inside one process).
#if defined(ST_EXPERIMENTS)
constexpr bool kExperiments = true;
#else
constexpr bool kExperiments = false;
#endif

// ---- runtime switches ---------------------------------------------------------------------------
// A/B and diagnostic switches: an ST_* environment variable, overridable at run time through the C ABI's
// st_set_option (parity tests compare kernel variants inside one process).  A call site holds a static Option;
// get() re-reads only when an override changed (generation counter), so the launch path pays one relaxed load.
int option_lookup(const char* name, int dflt);      // override > environment (where the build lets it through) > dflt
const char* option_env(const char* name);           // getenv under the same rule
unsigned option_generation();
struct Option {
    const char* name;
    int dflt;
    // (value, generation) packed into ONE atomic word: two host threads launching at once may both refresh the cache,
    // but neither can pair a new generation with a stale value
    std::atomic<unsigned long long> cached{0};
    Option(const char* n, int d) : name(n), dflt(d) {}
    int get() {
        const unsigned g = option_generation();
        const unsigned long long c = cached.load(std::memory_order_acquire);
        if ((unsigned)(c >> 32) == g) return (int)(unsigned)(c & 0xffffffffull);
        const int v = option_lookup(name, dflt);
        cached.store(((unsigned long long)g << 32) | (unsigned)v, std::memory_order_release);
        return v;
    }
};

// ---- device helpers ----------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Block-wide sum for 256-thread blocks; result valid in thread 0.  `scratch` = 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    __syncthreads();
    return r;
}

// StyleLossW2.forward's scalars for one head (style_transfer.py:178-181): the loss term and the seed of its backward,
// d loss / d root = gdiag * I.  A job rides along in the kernel that opens the Lyapunov backward chain (ns_prepare_kernel /
// ns_backward_entry_kernel) instead of being a launch of its own on the heads' critical path; loss_out == nullptr: no job.
struct W2LossJob {
    const float *mean, *mean_t, *cov, *cov_t, *root;
    int n;
    float weight;
    float* loss_out;
    float* gdiag_out;
};
__device__ __forceinline__ float w2_gdiag(const W2LossJob& j) { return -2.f * (j.weight / (float)j.n); }
// one 256-thread block; `scratch` = 4 floats of LDS
__device__ __forceinline__ void w2_loss_block(const W2LossJob& j, float* scratch) {
#pragma clang fp contract(off)
    float sm = 0.f, sc = 0.f;
    for (int i = threadIdx.x; i < j.n; i += 256) {
        const float d = j.mean[i] - j.mean_t[i];
        sm += d * d;
        const size_t ii = (size_t)i * j.n + i;
        sc += (j.cov_t[ii] + j.cov[ii]) - 2.f * j.root[ii];
    }
    sm = block_sum_256(sm, scratch);
    sc = block_sum_256(sc, scratch);
    if (threadIdx.x == 0) {
        const float fn = (float)j.n;
        j.loss_out[0] = (sm / fn + sc / fn) * j.weight;
        j.gdiag_out[0] = w2_gdiag(j);
    }
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- convolution (st_conv.hip) -----------------------------------------------------------------
// Implicit-GEMM 3x3 (or 1x1) convolution on the fp32 MFMA.  D[co][pixel] = sum_k W[k][co] * X[k][pixel],
// k = (tap, ci).  `wgt` is pre-arranged as [taps][Cin][Cout] (Cout fastest).
struct ConvProblem {
    const float* in;       // [Cin][H][W]
    const float* mask;     // optional [Cin][H][W]: staged operand = (mask > 0) ? in : 0   (ReLU backward, consumer
                           // side: costs a second operand stream + 24 registers; the plan masks on the PRODUCER
                           // side instead, see out_mask, and uses this only where the producer cannot)
    const float* wgt;      // [taps][Cin][Cout]
    const float* bias;     // optional [Cout]
    float* out;            // [Cout][H][W]
    int cin, cout, height, width;
    int taps;              // 9 or 1
    int relu;              // epilogue max(x, 0)
    int accumulate;        // epilogue out += result (after bias), else out = result
    float* scratch;        // optional split-K workspace (kConvScratchFloats floats); nullptr = never split
    // Strip sharding (3x3 only): rows -1 and `height` of the operand live in a halo block
    // [2][Cin][W] (top rows, then bottom rows) filled by the neighbour exchange; a missing neighbour
    // (has_up / has_down == 0) means the global image border, i.e. zero padding.  With `mask`, halo rows
    // arrive already masked by the sender.
    const float* in_halo;
    int has_up, has_down;
    int tune;              // experiment bits (see st_conv.hip); 0 = shipped default
    // split-precision path (st_conv_split.hip): 16-bit planes of the weights, [planes][9][Cin/16][Cout][16]
    // followed by a 256-byte trailer (word 0: bits of max |w|, used by the fp16 mode's power-of-two scale).
    //   planes = 0           -> exact fp32 MFMA kernel
    //   planes = 2, elem = 0 -> bf16x3      planes = 3, elem = 0 -> bf16x6
    //   planes = 2, elem = 1 -> fp16x3: fp16 planes of operands pre-scaled by a power of two taken from an
    //                           upper bound of max |in|, read from the device bound `amax_word` (raw float
    //                           bits in kAmaxSlots slots, see amax_commit / amax_read).  The bound normally comes for free from the PRODUCER of `in` (every
    //                           kernel that finalises a conv operand folds max |out| into `out_amax`, see
    //                           amax_commit); amax_measure = 1 makes the launcher measure `in` itself first
    //                           (standalone operators).  A bound that is 2^k too large costs nothing but k
    //                           bits of fp16's underflow floor (2^-29 below the maximum), so pooled tensors
    //                           reuse their input's word.  Halo rows (strip sharding) are folded in by the
    //                           launcher.
    const void* wgt_split;
    int planes;
    int elem;
    unsigned int* amax_word;
    int amax_measure;
    int halo_amax_folded;  // strip plans: the caller has dealt with max |halo rows| - either folded it into amax_word
                           // or handed over the bounds the SENDERS measured (below)
    // strip plans, round 5: max |row| of the neighbours' halo rows as raw bits, measured by the sender while it packed the
    // rows and shipped in the same message (st_api.hip halo_exchange: a 16-float trailer in front of the top rows / behind
    // the bottom rows).  The kernels take max(amax_word, these) - no launch between the halo's arrival and its consumer.
    const unsigned int* halo_bound_up;
    const unsigned int* halo_bound_down;
    // 1x1 problems in fp16x3 (st_conv1x1.hip): device bound on max |wgt| (plain fp32 [Cout][Cin] weights that
    // change every iteration, split while they are staged); nullptr -> exact fp32 kernel
    const unsigned int* wgt_amax;
    // optional [Cout][H][W]: out = (out_mask > 0) ? result : 0, applied last (after accumulate): the
    // threshold_backward of the NEXT data-gradient convolution, done while the gradient is produced
    const float* out_mask;
    // optional (any precision): fold max |out| of the finished output (after bias / ReLU / accumulate) into
    // this device bound (kAmaxWordUints unsigned ints), for the consumer's fp16 scale.  Zeroed once per pass.
    unsigned int* out_amax;
    // producer / consumer kernel only (st_conv_pc.hip): this launch produces output rows [row_begin, row_end) of the
    // image (0, 0 = all rows); operand rows outside the range are read from the same tensors.  Lets the launcher
    // cover an image with two tile shapes (see choose_pc_tile).
    int row_begin, row_end;
    // ... minus rows [row_skip_begin, row_skip_begin + row_skip_len) (row_skip_len == 0: nothing skipped;
    // row_skip_begin - row_begin must be a multiple of the tile height): the first and the last rows of a strip in ONE
    // launch (see overlap_part).
    int row_skip_begin, row_skip_len;
    // producer / consumer kernel, set by its launcher: Cout tiles dealt to groups of XCDs (1 = every XCD takes all Cout tiles
    // of its pixel tiles; see conv_pc_kernel's tile_of)
    int xcd_co_groups;
    // Strip plans, producer / consumer kernel only: a convolution split so that the halo exchange of its operand
    // overlaps most of it (SURVEY.md 8(e) "overlapped with interior compute").  0 = the whole strip in one go;
    // 1 = the INTERIOR rows [b, H - b), which read no halo row (in_halo must be null: the exchange may still be in
    // flight); 2 = the BOUNDARY rows [0, b) and [H - b, H) (in_halo set, launched after the exchange has landed).
    // b and both tile shapes come from conv_pc_overlap_choice(p), a pure function of the problem's shape.
    int overlap_part;
    // producer / consumer kernel only, forward: also write MaxPool2d(2) of the finished output, [Cout][H/2][W/2]
    // (reference style_transfer.py:21 'max').  Honoured only where conv_pc_fuses_pool(p) says so (tiles in which a
    // wave owns whole 2x2 windows, 16-byte store path); the caller launches the pool kernel otherwise.
    float* pool_out;
    // ... and, when this is non-null too, INSTEAD of the full-resolution output: one byte per pooled element, bits 0-1 =
    // position of the window's first maximum in row-major order (what the pooling backward scatters to), bit 2 = that
    // maximum is > 0 (the ReLU mask).  For conv outputs that nothing but the pool consumes (relu1_2, 2_2, 3_4, 4_4 in the
    // closure): the map is neither written by this launch nor re-read by the pooling backward.
    unsigned char* pool_code;
    // Winograd F(2x2, 3x3) form (st_conv_wino.hip): the layer's transformed weight planes (launch_winograd_weights; the forward
    // or the data-gradient set, matching wgt_split) and when to use them - 0 never, 1 where conv_wino_preferred() says the
    // measured time is lower, 2 wherever the kernel takes the problem (operator precision code 5, A/B runs)
    const void* wgt_wino;
    int wino;
};
// A bound lives in kAmaxSlots slots, one per 256-byte line: workgroup b commits to slot b % kAmaxSlots so that
// the ~2000 waves resident when a kernel starts (all of which see an empty bound) do not serialise on one
// address; the consumer takes the maximum over the slots.  Only the EXPONENT of the bound matters.
constexpr int kAmaxSlots = 32;
constexpr int kAmaxSlotStride = 64;                                  // unsigned ints between slots
constexpr int kAmaxWordUints = kAmaxSlots * kAmaxSlotStride;         // footprint of one bound
#if defined(__HIPCC__)
// Fold a thread's max |v| (raw bits; non-negative floats order like unsigned integers) into the bound: wave
// reduction, then at most one atomic per wave, skipped unless it raises the slot's exponent (the read may be
// stale, but slots only grow).  All 64 lanes must call it.
__device__ __forceinline__ void amax_commit(unsigned int m, unsigned int* bound) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)m, off);
        m = o > m ? o : m;
    }
    unsigned int* slot = bound + (blockIdx.x % kAmaxSlots) * kAmaxSlotStride;
    // plain (cacheable, possibly stale) read: a device-scope load here costs every wave a memory round trip at
    // the very end of its life
    if ((threadIdx.x & 63) == 0 && (m >> 23) > (*slot >> 23)) atomicMax(slot, m);
}
// the bound, wave-uniform; every lane of the wave must call it
__device__ __forceinline__ unsigned int amax_read(const unsigned int* bound) {
    unsigned int m = bound[(threadIdx.x % kAmaxSlots) * kAmaxSlotStride];
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)m, off);
        m = o > m ? o : m;
    }
    return (unsigned int)__builtin_amdgcn_readfirstlane((int)m);
}
// Workgroups are dealt to the 8 XCDs (private L2 each) round-robin by their LINEAR id.  logical_block() turns that id into one
// whose consecutive values run on the SAME XCD (XCD x takes the contiguous range [x total / 8, (x + 1) total / 8)), so that
// workgroups which read the same operand - the Cout tiles of one pixel tile, the tile pairs of one Gram split - share it
// through one L2 instead of fetching it into up to 8 of them (round 6; the 3 x 3 kernels have done this since round 1).
// Placement only: which workgroup computes what is unchanged, results are bit-identical.  remap == 0: the plain order.
__device__ __forceinline__ unsigned int logical_block(int remap) {
    const unsigned int linear = blockIdx.y * gridDim.x + blockIdx.x, total = gridDim.x * gridDim.y;
    if (!remap || (total & 7u) != 0) return linear;
    return (linear & 7u) * (total >> 3) + (linear >> 3);
}
__device__ __forceinline__ unsigned int abs_bits(float v) { return __builtin_bit_cast(unsigned int, v) & 0x7fffffffu; }
// fp16 mode: exponent e with bound * 2^e in [2^13, 2^14), from the bits of the bound on max|x| (0 for 0 /
// denormal / inf / nan).  fp16 overflows at 2^16: two spare bits, one of which the x2 average pooling and the
// L2 pooling (<= 1.56 x their input's maximum) may use when a pooled tensor reuses its input's bound.
// the operand's bound with the neighbours' halo rows' (ConvProblem::halo_bound_up / _down; wave-uniform scalar loads)
__device__ __forceinline__ unsigned int amax_with_halo(unsigned int own, const unsigned int* up, const unsigned int* down) {
    if (up) { const unsigned int u = up[0] & 0x7fffffffu; own = u > own ? u : own; }
    if (down) { const unsigned int d = down[0] & 0x7fffffffu; own = d > own ? d : own; }
    return own;
}
__host__ __device__ __forceinline__ int scale_exp(unsigned int amax_bits) {
    const int ef = (int)((amax_bits >> 23) & 0xffu);
    if (ef == 0 || ef == 255) return 0;
    const int e = 14 - (ef - 126);                    // bound < 2^(ef - 126)
    return e > 120 ? 120 : (e < -120 ? -120 : e);
}
__device__ __forceinline__ float pow2f(int e) { return __builtin_bit_cast(float, (unsigned int)(127 + e) << 23); }
#endif
int launch_conv_split(const ConvProblem& p, hipStream_t stream);
// Winograd F(2x2, 3x3) fp16x3 form of the same convolution (st_conv_wino.hip; builds with --experiments only)
#if defined(ST_EXPERIMENTS)
size_t winograd_weight_bytes(int cin, int cout);
int launch_winograd_weights(const float* w_torch, void* out, int cin, int cout, int dgrad, hipStream_t s);
bool conv_wino_applies(const ConvProblem& p);
bool conv_wino_preferred(const ConvProblem& p);   // ... and expected faster than the direct producer / consumer kernel
int launch_conv_wino(const ConvProblem& p, hipStream_t s);
#else
inline size_t winograd_weight_bytes(int, int) { return 256; }
inline int launch_winograd_weights(const float*, void*, int, int, int, hipStream_t) {
    set_error("the Winograd convolution (precision code 5) needs a library built with build.py --experiments");
    return 1;
}
inline bool conv_wino_applies(const ConvProblem&) { return false; }
inline bool conv_wino_preferred(const ConvProblem&) { return false; }
inline int launch_conv_wino(const ConvProblem&, hipStream_t) { return launch_winograd_weights(nullptr, nullptr, 0, 0, 0, nullptr); }
#endif
// fold max |x[0..n)| into a device bound (single = 0: kAmaxSlots-slot bound; 1: one word, the weight trailer)
int launch_amax(const float* x, long long n, unsigned int* word, int single, hipStream_t s);
// producer / consumer form of the unsharded fp16x3 3x3 convolution (st_conv_pc.hip)
bool conv_pc_applies(const ConvProblem& p);
bool conv_pc_preferred(const ConvProblem& p);      // ... and measured faster than the single-role kernel
int launch_conv_pc(const ConvProblem& p, hipStream_t stream);
// "fat" single-role form for large maps (st_conv_fat.hip): four waves of (32 CB) co x 128 px register tiles, staging inside the MFMA streams
bool conv_fat_applies(const ConvProblem& p);
bool conv_fat_preferred(const ConvProblem& p);     // ... and measured faster than the producer / consumer kernel
int launch_conv_fat(const ConvProblem& p, hipStream_t stream);
bool conv1x1_split_applies(const ConvProblem& p);
int launch_conv1x1_split(const ConvProblem& p, hipStream_t stream);
int launch_conv_splitk_reduce(const ConvProblem& p, int ksplit, hipStream_t stream);
// conv_precision code of the C ABI (0 fp32, 2 bf16x3, 3 bf16x6, 4 fp16x3) -> planes / element type
inline bool conv_precision_valid(int code) { return code == 0 || code == 2 || code == 3 || code == 4; }
inline int conv_precision_planes(int code) { return code == 4 ? 2 : code; }
inline int conv_precision_elem(int code) { return code == 4 ? 1 : 0; }
inline size_t split_weight_bytes(int cin, int cout, int planes) { return (size_t)cin * cout * 9 * 2 * planes + 256; }
// torch [Cout][Cin][3][3] -> 16-bit planes (+ trailer) for launch_conv_split (dgrad: roles swapped, taps rotated)
int launch_relayout_split(const float* w, void* out, int cin, int cout, int dgrad, int planes, int elem, hipStream_t s);
// Split-K: layers whose output has too few 32x32 MFMA tiles to fill 256 CUs (deep layers at small
// images) split the input-channel range over `ksplit` workgroups; raw partial sums go to `scratch`
// and a fixed-order reduce applies bias / ReLU / accumulate.  8M floats covers every case where the
// heuristic splits (ksplit * Cout * H * W <= ~4.2M by construction).
constexpr size_t kConvScratchFloats = (size_t)8 << 20;
int launch_conv(const ConvProblem& p, hipStream_t stream);
double conv_flops(const ConvProblem& p);   // algorithmic 2*taps*Cin*Cout*H*W

// torch [Cout][Cin][3][3] -> forward layout [9][Cin][Cout]
int launch_relayout_fwd(const float* w, float* out, int cin, int cout, hipStream_t stream);
// torch [Cout][Cin][3][3] -> data-gradient layout [9][Cout][Cin] with the taps rotated by 180 degrees
int launch_relayout_dgrad(const float* w, float* out, int cin, int cout, hipStream_t stream);

// Boundary rows of a [C][H][W] map into two contiguous [C][W] send buffers (row 0 -> up, row H-1 ->
// down); with `mask` the rows are multiplied by (mask > 0) (threshold_backward on the sender side).
// bounds != nullptr: also max |packed row| of either direction as raw bits into bounds_up[0] / bounds_down[0] (the message
// trailers); `scratch` = kPackScratchUints words of the plan, the last one a ticket that is zero between launches
constexpr int kPackScratchUints = 2 * 512 * 8 + 64;
int launch_pack_rows(const float* src, const float* mask, int channels, int height, int width, float* out_up,
                     float* out_down, hipStream_t s, unsigned int* bounds_up = nullptr, unsigned int* bounds_down = nullptr,
                     unsigned int* scratch = nullptr);

// ---- first layer (st_conv_first.hip) -----------------------------------------------------------
// conv1_1: Normalize + replicate pad + 3->64 conv + bias + ReLU (style_transfer.py:30-31,39,85-87)
int launch_conv_first_fwd(const float* image, const float* w /*[64][3][3][3]*/, const float* b, float* out,
                          int height, int width, hipStream_t stream, const float* halo = nullptr,
                          int has_up = 0, int has_down = 0, unsigned int* out_amax = nullptr);
// its data gradient incl. ReLU mask (relu_out == nullptr: grad_out is already masked by its producer),
// replicate-pad fold and 1/std; accumulates into grad_image
// dp_scratch: 3 * (height + 2) * (width + 2) floats (dP on the padded domain, folded by a second kernel)
// conv1_1 forward + relu1_1's partial moments in one pass (st_conv_first.hip): *splits workgroups' partials for
// launch_gram_finalize
bool conv_first_gram_applies(int height, int width, const float* image, const float* out, int max_splits);
int launch_conv_first_fwd_gram(const float* image, const float* w, const float* b, float* out, int height, int width,
                               hipStream_t stream, unsigned int* out_amax, float* partial, float* partial_sum, int max_splits,
                               int* splits, float w_l1max, float b_max);
// dp_scratch holds `parts` partial planes of 3 (height + 2) (width + 2) floats; parts = conv_first_dgrad_parts(GLOBAL height,
// width): the channel slices whose partial sums the fold kernel adds in order (1 on large images)
int conv_first_dgrad_parts(int height, int width);
// update (optional, unsharded plans): the fold kernel also applies st_plan_step's Adam + clamp + EMA update to the gradient
// element it has just finished (FoldUpdate, below) - the update is then not a launch of its own
struct FoldUpdate;
int launch_conv_first_dgrad(const float* grad_out, const float* relu_out, const float* w, float* grad_image,
                            float* dp_scratch, int height, int width, int accumulate, hipStream_t stream,
                            const float* ghalo = nullptr, int has_up = 0, int has_down = 0, int parts = 1,
                            const FoldUpdate* update = nullptr);

// ---- pooling (st_pool.hip) ---------------------------------------------------------------------
int launch_pool_fwd(const float* in, float* out, int channels, int height, int width, int mode, hipStream_t s);
bool conv_pc_fuses_pool(const ConvProblem& p);     // st_conv_pc.hip: will launch_conv(p) write p.pool_out?
int fold_halo_amax(const ConvProblem& p, hipStream_t stream);     // st_conv_split.hip
// st_conv_pc.hip: how a strip's convolution is cut into interior + boundary launches (overlap_part), and whether the cost
// model expects that to pay (split cost <= whole cost + the exchange latency it hides).  `p` = the whole problem
// (overlap_part ignored; in_halo may be null).  Returns false when the kernel does not take the problem at all.
struct PcOverlap {
    int rows_b;                 // boundary rows at the top (= the boundary tile's height)
    int rows_bottom;            // boundary rows at the bottom (a multiple of it)
    int shape_i, tw_i;          // interior tile
    int shape_b, tw_b;          // boundary tile (tile height == rows_b)
    double cost_split, cost_whole;      // microseconds (cost model)
    bool pays;
    bool pool;                  // both launches write the fused max pool (p.pool_out != null and both tiles can)
};
bool conv_pc_overlap_choice(const ConvProblem& p, PcOverlap* out);
// grad_in[C][H][W] (fully written, zeros in dropped odd rows/cols) from grad_out[C][H/2][W/2]
int launch_pool_bwd_codes(const unsigned char* code, const float* grad_out, float* grad_in, int channels, int height,
                          int width, hipStream_t s);
int launch_pool_bwd(const float* in, const float* grad_out, float* grad_in, int channels, int height,
                    int width, int mode, hipStream_t s);

// ---- Gram / moments (st_gram.hip) --------------------------------------------------------------
struct GramWorkspace {
    float* partial;      // [splits][C][C]
    float* partial_sum;  // [splits][C]
    int max_splits;
};
int gram_choose_splits(int channels, long long npix, int max_splits);
// partial[s] = F[:, ks:ke] F[:, ks:ke]^T, partial_sum[s] = row sums; F is [C][npix]
// bound != nullptr: fp16x3 arithmetic, scaled by the device bound on max |feat| (see ConvProblem::amax_word)
int launch_gram_partial(const float* feat, int channels, long long npix, int splits, GramWorkspace ws,
                        hipStream_t s, const unsigned int* bound = nullptr);
// mean = sum/npix, srm = sum/npix (fixed-order reduction over splits)
int launch_gram_finalize(GramWorkspace ws, int channels, long long npix, int splits, float* mean, float* srm,
                         hipStream_t s, float* cov = nullptr, float cov_eps = 0.f);   // cov: also srm - mean mean^T + eps I

// ---- small dense algebra for the W2 style loss (st_smallgemm.hip) ------------------------------
enum GemmEpilogue {
    EPI_SCALE = 0,         // D = c * P1
    EPI_IDENT_MINUS = 1,   // D = (cI * I - P1) * c
    EPI_DIFF = 2,          // D = (P1 - P2) * c
    EPI_DEV_SQRT_SCALE = 3 // D = P1 * sqrt(*dev_scalar)
};
struct GemmProblem {
    const float* a1; const float* b1;     // P1 = op(a1) @ op(b1)
    const float* a2; const float* b2; const float* b2sub;  // P2 = op(a2) @ (b2 - b2sub)  (EPI_DIFF only)
    float* d;
    int ta1, tb1, ta2;                    // 1 = operand is used transposed
    int epilogue;
    float c, ci;
    const float* dev_scalar;
    int n;                                // 0: the batch's n; else this problem's own size (mixed batches, n <= 256)
    float* sumsq_partials;                // n = 512 only (gemm_sumsq_fusable): partials[tile] = sum of D^2 over the tile, the
                                          // operand of ns_prepare / the backward chain's entry instead of a launch of its own
};
bool gemm_sumsq_fusable(int n);           // launch_gemm_batch honours GemmProblem::sumsq_partials for this size: (n / 32)^2 partials
struct GemmBatch {
    GemmProblem p[6];                     // (6: both products of a recurrence step for three chains in lockstep)
    int count;
    int n;                                // all matrices n x n, n % 32 == 0 (problems may override it, see GemmProblem::n)
};
int launch_gemm_batch(const GemmBatch& b, hipStream_t s);
// dF = Ssym F + b 1^T for a style tap of <= 1024 pixels in one launch (st_smallgemm.hip); mask: optional [C][npix], out = 0
// where mask <= 0; out_amax: optional bound for an fp16x3 consumer
bool head_dgrad_small_applies(int channels, long long npix);
int launch_head_dgrad_small(const float* ssym, const float* feat, const float* bias, const float* mask, float* out, int channels,
                            long long npix, unsigned int* out_amax, hipStream_t s);

struct NSWorkspace {                      // all n*n unless noted
    float *y0, *y1, *z0, *z1, *t;         // forward iterates
    float *a0, *a1, *q0, *q1, *e, *atq, *qa;  // backward iterates
    float* scalars;                       // [4] device scalars: ||M||_F, ||S||_F, spare
    // fp16x3 chains (st_nsgemm.hip, n >= 256): 5 matrix slots x 2 roles x 2 planes of n*n halves
    // (forward y, y', z, z', t; the backward reuses them for a, a', q, q', E)
    _Float16* planes;
    // persistent chain kernel (st_nschain.hip): two sets of barrier words used by alternate launches (a launch clears the
    // other set) + the error word; zero when the workspace is put to use (ns_workspace_reset)
    unsigned int* chain_sync;
    int chain_launches;
    float* chain_arena;                   // kNsChainArenaMats matrices when ST_NS_CHAIN_L2=2 asked for them at carve time, else null
};
constexpr int kNsChainArenaMats = 132;    // forward 4 + 2 + 11 x 2 + 10 x 4 = 68, backward 3 + 12 x 2 + 11 x 2 + 10 = 59

// ---- persistent Newton-Schulz chains (st_nschain.hip) ------------------------------------------------
struct NsChainJob {
    int n;
    int tile0, tiles;                     // set by launch_ns_chain: first workgroup and workgroup count of the job
    int forward, backward;                // which recurrences run (both: sqrtm.py:9-25 then :36-47 for a seed gdiag * I)
    const float* m;                       // forward: the matrix whose root is taken (its upper triangle is read)
    const float* m_partials;              // optional: m_nparts partial sums of squares of m (left by the product that made it)
    int m_nparts;
    int l2_loads;                         // 1: operands through the L2 (plain loads behind an acquire per barrier) instead of sc1 loads
    int symmetric;                        // 1: only the tile pairs ti <= tj of every product + mirror images (exactly symmetric
                                          // iterates, -47 % work; costs accuracy on ill-conditioned input, see st_nschain.hip);
                                          // 0: every tile, every iterate kept as X and X^T (the reference's products, faithfully)
    float *y0, *y1, *z0, *z1, *t;         // n x n workspace matrices (the backward keeps a in y's slots, q in z's, E in t's)
    float *yt0, *yt1, *zt0, *zt1, *tt;    // full jobs: their transposes
    float* arena;                         // optional, with l2_loads: kNsChainArenaMats n x n matrices - every iterate of every step
                                          // gets its own (no address is read before it is written within a launch: no acquire)
    float* root;                          // forward: result; backward only: operand
    float* grad_m;                        // backward: result, dL/dM
    float* scalars;                       // [0] = ||m||_F, [1] = ||root||_F, [8 ..] tile partial sums
    const float* gdiag_dev;               // the seed: a device scalar, or (null) the host value below, or the W2 job's own
    float gdiag;
    W2LossJob loss;                       // optional: the head's loss scalars ride along (loss_out == nullptr: none)
    unsigned int *sync, *sync_next, *error;
};
struct NsChainLaunch {
    NsChainJob job[3];
    int count;
};
#if defined(ST_EXPERIMENTS)
int ns_chain_mask();                      // ST_NS_CHAIN: bit 0 shallow heads, 1 relu4_1, 2 relu5_1, 3 standalone operators
bool ns_chain_enabled();                  // any bit
int ns_chain_sync_uints();                // barrier words per set
int ns_chain_tiles(int n, bool symmetric); // workgroups of a job
int launch_ns_chain(NsChainLaunch& launch, hipStream_t s);
#else                                     // default build: no persistent chain kernel, the callers' branches fold away
inline int ns_chain_mask() { return 0; }
inline bool ns_chain_enabled() { return false; }
inline int ns_chain_sync_uints() { return 16; }
inline int ns_chain_tiles(int, bool) { return 0; }
inline int launch_ns_chain(NsChainLaunch&, hipStream_t) {
    set_error("the persistent Newton-Schulz chain kernel needs a library built with build.py --experiments");
    return 1;
}
#endif
bool ns_chain_combined();                 // ... and neither ST_NS_FULL_BACKWARD nor ST_NS_F16_FWD asks for another recurrence
// the job of a workspace: matrices, scalars, this launch's barrier words (advances the workspace's launch parity)
NsChainJob ns_chain_job(NSWorkspace& ws, int n);
int ns_workspace_reset(NSWorkspace& ws, hipStream_t s);     // zero the barrier / error words (after ns_workspace_carve)
// forward + backward chain of one head (or up to three heads) in ONE launch; loss[i] may be null
int ns_sqrt_chain(const float* const* m, float* const* root, float* const* grad_m, const int* n, NSWorkspace* const* ws,
                  const int* m_partials, const W2LossJob* loss, int lanes, hipStream_t s);
// after a synchronise: has a chain kernel of this workspace given up waiting (a workgroup never became resident)?
int ns_chain_check(NSWorkspace& ws, const char* what);

// ---- fp16x3 Newton-Schulz products (st_nsgemm.hip) -----------------------------------------------
// scale exponent of a plane pair: stored = value * 2^exp.  Either a host constant, or (num != nullptr) derived on
// the device from the bound |num[0] / den[0]| * mult (the Lyapunov iterate q, whose magnitude is a device scalar).
struct NsScale {
    int exp;
    const float* num;
    const float* den;
    float mult;
};
struct NsPlanes {                          // operand: role-A blocks of A, or role-B blocks of B (see st_nsgemm.hip)
    const _Float16* p0;
    const _Float16* p1;
    NsScale scale;
};
struct NsPlanesOut {                       // result planes in role A and / or role B (nullptr = not needed)
    _Float16 *a0, *a1, *b0, *b1;
    NsScale scale;
};
struct NsGemmProblem {
    NsPlanes a, b;                         // D = A @ B
    float* d32;                            // optional fp32 row-major result
    NsPlanesOut out;
    int epilogue;                          // EPI_SCALE, EPI_IDENT_MINUS or EPI_DEV_SQRT_SCALE
    float c, ci;
    const float* dev_scalar;
};
struct NsGemmBatch {
    NsGemmProblem p[2];
    int count;
    int n;
};
struct NsToPlanesItem {
    const float* src;
    NsPlanesOut out;
};
struct NsToPlanes {
    NsToPlanesItem item[2];
    int count;
};
bool ns_f16_applies(int n);
int launch_ns_gemm_f16(const NsGemmBatch& b, hipStream_t s);
int launch_ns_planes_from_f32(const NsToPlanes& job, int n, hipStream_t s);
int ns_sqrt_forward_f16(const float* m, float* root, int n, NSWorkspace& ws, hipStream_t s);
int ns_sqrt_backward_diag_f16(const float* root, const float* grad_diag, float* grad_m, int n, NSWorkspace& ws,
                              hipStream_t s, const W2LossJob* loss = nullptr, int root_partials = 0);
// csrc/st_diag.hip: shares[i] = 1 if candidates[i] sits on the same hardware queue as `ref` (returns 1 if undecidable)
int probe_queue_sharing(hipStream_t ref, const hipStream_t* candidates, int count, int* shares);
size_t ns_workspace_floats(int n);
void ns_workspace_carve(NSWorkspace& ws, float* base, int n);
// m_partials > 0: ws.scalars + 8 already holds that many partial sums of squares of m (written by the product that made m);
// root_partials != nullptr: the last product leaves root's partial sums there too and reports their count (0: not fused)
int ns_sqrt_forward(const float* m, float* root, int n, NSWorkspace& ws, hipStream_t s, int m_partials = 0,
                    int* root_partials = nullptr);
// up to three fp32 chains of DIFFERENT sizes (n <= 256 each) in lockstep: every recurrence step is one launch for all
int ns_sqrt_forward_lockstep(const float* const* m, float* const* root, const int* n, NSWorkspace* const* ws, int lanes,
                             hipStream_t s);
int ns_sqrt_backward_diag_lockstep(const float* const* root, const float* const* grad_diag, float* const* grad_m, const int* n,
                                   NSWorkspace* const* ws, int lanes, hipStream_t s, const W2LossJob* loss = nullptr);
// grad_diag != nullptr: grad_root = (*grad_diag_value) * I with the value read on device from grad_diag[0].
// loss (diag form only): the head's W2 scalars are computed by the chain's opening kernel, which then also WRITES
// grad_diag[0] (= loss->gdiag_out) instead of reading it.
// root_partials > 0: ws.scalars + 8 holds that many partial sums of squares of root (ns_sqrt_forward's root_partials)
int ns_sqrt_backward(const float* root, const float* grad_root, const float* grad_diag, float* grad_m, int n,
                     NSWorkspace& ws, hipStream_t s, const W2LossJob* loss = nullptr, int root_partials = 0);

// ---- pointwise / reductions (st_pointwise.hip) -------------------------------------------------
int launch_fill(float* p, long long n, float v, hipStream_t s);
int launch_identity(float* p, int n, hipStream_t s);
// cov = srm - mean mean^T + eps I
int launch_cov_from_moments(const float* mean, const float* srm, float* cov, int n, float eps, hipStream_t s);
// out[0] = ||a||_F
int launch_frobenius(const float* a, long long count, float* out, hipStream_t s);
// norm = ||a||_F -> norm_out[0]; a_scaled = a / norm; second = I (g, gdiag null), g / norm, or (gdiag[0] / norm) I.
// `partials`: >= 256 floats of scratch.  Two launches.
int launch_sumsq_partials(const float* a, long long count, float* partials, int* nparts, hipStream_t s);
int launch_ns_prepare(const float* a, int n, float* norm_out, float* partials, float* a_scaled, const float* g,
                      const float* gdiag, float* second, hipStream_t s, bool first_t = false,
                      const W2LossJob* loss = nullptr, int partials_ready = 0);   // > 0: `partials` is filled already
// y = a / *scalar
int launch_div_by_dev_scalar(const float* a, const float* scalar, float* y, long long count, hipStream_t s);
// q = (diag_value[0] / *scalar) * I
int launch_scaled_identity_div(const float* diag_value, const float* scalar, float* q, int n, hipStream_t s);

// ContentLossMSE value + gradient: loss_out[0] = weight * mean((f-t)^2); grad = weight * 2 (f-t) / count
// partials: kStreamBlocks floats (content) / 4 * kStreamBlocks floats (TV).  ticket != nullptr: a zeroed device word;
// the value is then finished by the kernel's last block (one launch instead of two)
constexpr int kStreamBlocks = 2048;
int launch_content_mse(const float* feat, const float* target, long long count, float weight, float* grad,
                       float* partials, float* loss_out, hipStream_t s, unsigned int* ticket = nullptr);
// W2 head scalars after the NS forward.  loss_out[0] = weight * (mean((mu-mu_t)^2) + mean(diag(cov_t + cov - 2 root)))
// gdiag_out[0] = -2 * (weight / n)   (the diagonal value of dL/d root)
int launch_style_loss_value(const float* mean, const float* mean_t, const float* cov, const float* cov_t,
                            const float* root, int n, float weight, float* loss_out, float* gdiag_out,
                            hipStream_t s);
// dcov = at^T dT-product result `g` + (weight/n) I ; then
//   ssym[c][d] = (dcov[c][d] + dcov[d][c]) / npix,  bvec[c] = (2 weight (mu-mu_t)[c]/n - sum_d (dcov+dcov^T)[c][d] mu[d]) / npix
int launch_style_grad_finish(const float* g, const float* mean, const float* mean_t, int n, float weight,
                             long long npix, float* ssym, float* bvec, hipStream_t s,
                             unsigned int* ssym_amax = nullptr);
// TV loss partial sums + gradient (optionally scaled by `weight`): see st_pointwise.hip
int launch_tv(const float* image, int height, int width, float weight, float* grad, float* partials,
              float* loss_out, hipStream_t s, unsigned int* ticket = nullptr);
// Strip of a larger image: local rows [row0, row0 + height) of `global_height`; halo = [2][3][W] rows of the
// neighbours (nullptr / has_* == 0 at the global border).  Writes the gradient and sums[4] = the four
// sums of squared differences owned by this strip (to be all-reduced), no final value.
struct StripInfo {
    int row0, global_height, has_up, has_down;
    const float* halo;
};
int launch_tv_strip(const float* image, int height, int width, StripInfo strip, float weight, float* grad,
                    float* partials, float* sums4, hipStream_t s);
float* tv_debug_buffer();     // diagnostic (ST_TV_VARIANT=3, st_plan_debug_read what = 1)
int launch_tv_final(const float* sums4, int global_height, int width, float weight, float* loss_out, hipStream_t s);
// content MSE on a strip: gradient with the GLOBAL element count, sum of squares into sum_out[0]
int launch_content_mse_strip(const float* feat, const float* target, long long local_count,
                             long long global_count, float weight, float* grad, float* partials,
                             float* sum_out, hipStream_t s);
int launch_content_mse_final(const float* sum, long long global_count, float weight, float* loss_out, hipStream_t s);
// y = a / d (host scalar)
int launch_div_by_scalar(const float* a, float d, float* y, long long count, hipStream_t s);
// total = ((((((l0 + l1) + l2) + l3) + l4) + l5) + l6)   (SumLoss order)
int launch_sum_losses(float* losses8, hipStream_t s, float* copy = nullptr);
struct AdamScalars {
    float lerp_w;        // (float)(1 - beta1)
    float beta2;         // (float)beta2
    float one_m_beta2;   // (float)(1 - beta2)
    float step_size;     // (float)(lr / (1 - beta1^step))
    float bc2_sqrt;      // (float)sqrt(1 - beta2^step)
    float eps;
    float decay;         // fp32 EMA decay
    float one_m_decay;   // 1 - decay evaluated in fp32
};
// Work that used to be launches of its own around the update (round 6: two dependent launches less per iteration on the
// caller's stream): the total of the seven loss terms (sum_losses_kernel's arithmetic, by one thread) and the zeroing of the
// fp16x3 operand bounds for the NEXT forward pass.  All fields optional.
struct AdamTail {
    float* losses8;            // [7 terms | total]: total = 0 + l0 + ... + l6
    float* losses_copy;        // the caller's 8-float result buffer (may be null)
    unsigned int* zero;        // words to clear
    long long zero_count;
};
int launch_adam_clamp_ema(float* image, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                          long long count, AdamScalars sc, hipStream_t s, AdamTail tail = AdamTail{});
// st_plan_step's update handed to conv1_1's fold kernel (launch_conv_first_dgrad): state tensors [3][H][W] like the gradient
struct FoldUpdate {
    float* image;
    float* exp_avg;
    float* exp_avg_sq;
    float* ema;
    AdamScalars sc;
    AdamTail tail;
};
#if defined(__HIPCC__)
// torch.optim.Adam single-tensor step (torch/optim/adam.py:414-547, as configured at style_transfer.py:458) +
// image.clamp_(0, 1) (:485) + EMA.update (:250-253) on one element; every operation rounded on its own (no contraction)
__device__ __forceinline__ void adam_clamp_ema_element(float g, float& m, float& v, float& p, float& e, const AdamScalars& sc) {
#pragma clang fp contract(off)
    m = __builtin_fmaf(sc.lerp_w, g - m, m);                 // exp_avg.lerp_(grad, 1 - beta1)
    v = v * sc.beta2;                                        // exp_avg_sq.mul_(beta2)
    v = v + (sc.one_m_beta2 * g) * g;                        //   .addcmul_(grad, grad, value=1 - beta2)
    const float denom = sqrtf(v) / sc.bc2_sqrt + sc.eps;     // (exp_avg_sq.sqrt() / bc2_sqrt).add_(eps)
    p = p - sc.step_size * (m / denom);                      // param.addcdiv_(exp_avg, denom, value=-step_size)
    p = fminf(fmaxf(p, 0.f), 1.f);                           // image.clamp_(0, 1)
    e = e * sc.decay;                                        // self.value *= self.decay
    e = e + sc.one_m_decay * p;                              // self.value += (1 - self.decay) * input
}
// AdamTail by the threads of a 1-D grid of 256-thread blocks
__device__ __forceinline__ void adam_tail(const AdamTail& tail) {
#pragma clang fp contract(off)
    if (tail.losses8 && blockIdx.x == 0 && threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 7; ++i) t = t + tail.losses8[i];        // Python sum(): 0 + l0 + l1 + ... (SumLoss, :208)
        tail.losses8[7] = t;
        if (tail.losses_copy) {
            for (int i = 0; i < 7; ++i) tail.losses_copy[i] = tail.losses8[i];
            tail.losses_copy[7] = t;
        }
    }
    if (tail.zero)
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < tail.zero_count; i += (long long)gridDim.x * 256) tail.zero[i] = 0u;
}
#endif

}  // namespace st
