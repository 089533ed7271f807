// C ABI of libst_amd.so (include/st_amd.h): handle management and the sequencing of one
// closure / one optimiser iteration.  No autograd: forward, loss heads, hand-derived backward and
// the Adam + clamp + EMA update are explicit kernel launches on the caller's stream.
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/st_amd.h"
#include "st_common.h"

namespace st {

int fabric_apply(st_fabric* f, const st_exchange& ex, hipStream_t fallback);      // st_fabric.hip

static thread_local std::string g_error;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_error = buf;
}
const char* get_error() { return g_error.c_str(); }

// runtime switches (st_common.h): overrides set through st_set_option win over the environment
static std::mutex g_option_mutex;
static std::vector<std::pair<std::string, int>> g_option_override;
static std::atomic<unsigned> g_option_gen{1};
unsigned option_generation() { return g_option_gen.load(std::memory_order_relaxed); }
int option_lookup(const char* name, int dflt) {
    {
        std::lock_guard<std::mutex> lock(g_option_mutex);
        for (const auto& kv : g_option_override)
            if (kv.first == name) return kv.second;
    }
    const char* v = option_env(name);
    return (v && *v) ? atoi(v) : dflt;
}
// The switches a default build reads from the environment (documented in tools/README.md; tests/test_abi_cpu.py compares the
// two lists).  Everything else is an A/B or diagnostic switch: st_set_option only, or any build with --experiments.
static const char* const kEnvSwitches[] = {
    "ST_AMD_TIMELINE",      // time stamps of the closure's phases
    "ST_STREAM_LOG",        // the stream / hardware-queue layout the probe chose
    "ST_RANGE_LOG",         // what the activation-aware range guard measured
    "ST_CONV_RANGE_GUARD",  // 0: no weights-only range guard at st_net_create
    "ST_CONV_RANGE_LOG2",   // its threshold (log2 of the channel gain that flags a layer)
    "ST_NS_F16",            // 0: every Newton-Schulz chain on the fp32 matrix pipe
    "ST_GRAM_F32",          // 1: Gram matrices on the fp32 matrix pipe
    "ST_HEAD_1X1_F32",      // 1: the heads' 1 x 1 gradient step on the fp32 matrix pipe
    "ST_STRIP_OVERLAP",     // strip plans: 0 never / 1 by the cost model / 2 always split a convolution for halo overlap
    "ST_STRIP_OVERLAP_US",  // ... the exchange latency the split may cost
    "ST_STRIP_SPARE_CUS",   // ... CUs left free for RCCL's point-to-point kernels beside an interior launch
    "ST_STRIP_NS_OWNER",    // ... the rank that owns relu5_1's chains
    "ST_STREAM_PROBE",      // 0: skip the hardware-queue probe (fixed stream layout)
};
const char* option_env(const char* name) {
    if (!kExperiments) {
        bool listed = false;
        for (const char* k : kEnvSwitches) listed = listed || strcmp(k, name) == 0;
        if (!listed) return nullptr;
    }
    return getenv(name);
}
static void option_set(const char* name, int value, bool clear) {
    std::lock_guard<std::mutex> lock(g_option_mutex);
    for (size_t i = 0; i < g_option_override.size(); ++i)
        if (g_option_override[i].first == name) {
            if (clear) g_option_override.erase(g_option_override.begin() + i);
            else g_option_override[i].second = value;
            g_option_gen.fetch_add(1);
            return;
        }
    if (!clear) g_option_override.emplace_back(name, value);
    g_option_gen.fetch_add(1);
}

namespace {

// torchvision vgg19 cfg "E" truncated at features[29] (reference style_transfer.py:35)
struct OpDesc {
    int kind;        // 0 = conv(+ReLU), 1 = pool
    int index;       // conv number 0..12 or pool number 0..3
    int feat_index;  // features[] index of the produced tap (the ReLU for convs, the pool itself)
    int cin, cout;
};
const OpDesc kProgram[] = {
    {0, 0, 1, 3, 64},     {0, 1, 3, 64, 64},    {1, 0, 4, 64, 64},    {0, 2, 6, 64, 128},
    {0, 3, 8, 128, 128},  {1, 1, 9, 128, 128},  {0, 4, 11, 128, 256}, {0, 5, 13, 256, 256},
    {0, 6, 15, 256, 256}, {0, 7, 17, 256, 256}, {1, 2, 18, 256, 256}, {0, 8, 20, 256, 512},
    {0, 9, 22, 512, 512}, {0, 10, 24, 512, 512}, {0, 11, 26, 512, 512}, {1, 3, 27, 512, 512},
    {0, 12, 29, 512, 512},
};
constexpr int kNumOps = sizeof(kProgram) / sizeof(kProgram[0]);
const int kStyleFeat[5] = {1, 6, 11, 20, 29};     // style_transfer.py:317
const int kStyleConv[5] = {0, 2, 4, 8, 12};
constexpr int kContentConv = 9;                   // relu4_2 = features[22]
constexpr float kCovEps = 1e-4f;                  // StyleLossW2 eps (style_transfer.py:152)

struct Node {
    float* y = nullptr;   // activation (post-ReLU conv output or pooled map) [c][h][w]
    float* g = nullptr;   // gradient w.r.t. y, same shape (allocated lazily)
    float* yhalo = nullptr;   // strip mode: [2][c][w] rows of the neighbours (only if a conv reads this node)
    float* ghalo = nullptr;   // strip mode: [2][c][w] masked gradient rows of the neighbours (conv outputs)
    int c = 0, h = 0, w = 0;
    int hg = 0;               // global height at this level (== h when not sharded)
    bool pooled_by_conv = false;   // forward, strip plans: this conv's epilogue wrote the following max pool
    unsigned char* pool_code = nullptr;   // conv feeding a max pool: argmax + mask codes of the pooled windows (closure only)
    bool coded = false;            // this pass wrote pool_code INSTEAD of y (ConvProblem::pool_code)
    // device words (raw float bits) bounding max |y| / max |g| for the fp16x3 convolutions' scales: written by
    // the kernels that finalise y / g (amax_commit), zeroed at the start of every forward.  Pooled maps reuse
    // their input's y word, and a conv feeding a pool reuses the pool's g word (see scale_exp's spare bit).
    unsigned int* y_amax = nullptr;
    unsigned int* g_amax = nullptr;
    size_t count() const { return (size_t)c * h * w; }
};

struct StyleHead {
    int n = 0;            // channels
    long long npix = 0;        // GLOBAL pixel count of the tap (normalisation of the moments)
    long long npix_local = 0;  // pixels held by this plan (== npix unless strip-sharded)
    unsigned int* s_amax = nullptr;   // fp16x3: bound on max |ssym| of this pass (one of plan->amax_word's bounds)
    bool target_set = false;
    bool joined_in_build = false;    // phase construction: this head's broadcast + gradient step have been placed
    // targets
    float *mean_t = nullptr, *cov_t = nullptr, *root_t = nullptr;
    // per-iteration
    float *mean = nullptr, *srm = nullptr, *cov = nullptr, *tmat = nullptr, *mmat = nullptr, *root = nullptr,
          *gm = nullptr, *dt = nullptr, *dcov = nullptr, *ssym = nullptr, *bvec = nullptr, *gdiag = nullptr;
    float* conv_scratch = nullptr;     // split-K workspace of the head's 1x1 gradient conv (small taps only)
    NSWorkspace ns{};
    GramWorkspace gram{};
    bool allocated = false;
};

struct ProfileEvent {
    hipEvent_t start, stop;
    double flops;
};

// HBM-bound kernels of the step, timed like the conv launches when profiling is on (bench.py `roofline_hbm`):
// category, algorithmic bytes of the launch (operands read once + results written once)
enum HbmCat { HBM_CONV1_FWD = 0, HBM_CONV1_DGRAD, HBM_POOL_BWD, HBM_ADAM, HBM_TV, HBM_GRAM1, HBM_CONTENT, HBM_HEAD_1X1, HBM_CATS };
struct HbmEvent {
    hipEvent_t start, stop;
    int cat;
    double bytes;
};

}  // namespace
}  // namespace st

using namespace st;

struct st_net {
    int pooling = 0;
    float* w_first = nullptr;        // conv1_1 weight, torch layout [64][3][3][3]
    float w_first_l1max = 0.f;       // max over output channels of sum |w|, and max |bias|: the a-priori bound of relu1_1 per
    float b_first_max = 0.f;         // pixel block that the fused conv1_1 + Gram kernel scales its fp16 planes by
    float* bias[13] = {};
    float* w_fwd[13] = {};           // [9][Cin][Cout]   (convs 1..12)
    float* w_bwd[13] = {};           // [9][Cout][Cin], taps rotated (convs 1..12)
    int conv_planes = 0;             // 0: fp32 MFMA; 2 / 3 planes: split-precision convolutions (st_common.h)
    int conv_elem = 0;               // plane element type: 0 bf16, 1 fp16 (fp16x3)
    void* ws_fwd[13] = {};           // bf16 planes of the forward weights (convs 1..12)
    void* ws_bwd[13] = {};           // bf16 planes of the data-gradient weights
    // fp16x3 networks, dynamic-range guard: a convolution whose weights carry a channel far above the layer's median
    // (the signature of weights that compensate a tiny-valued operand channel) runs in bf16x6 instead - three bf16
    // planes, 8-bit exponents, no per-tensor scale to fall out of (see range_guard in net_fill)
    int wide_fwd[13] = {};           // 1: this layer's forward runs bf16x6
    int wide_bwd[13] = {};           // 1: its data gradient does
    void* wsx_fwd[13] = {};          // bf16x6 planes of the flagged layers
    void* wsx_bwd[13] = {};
    float* w_torch[13] = {};         // the weights as given ([Cout][Cin][3][3]): source of planes built after creation, when
                                     // the activation-aware guard (st_plan_range_guard) flags a layer
    int guard_fwd[13] = {};          // 1: flagged by the activation-aware guard (subset of wide_*)
    int guard_bwd[13] = {};
};

namespace {
// arithmetic of trunk convolution `conv` (forward / data gradient) under the network's mode and the range guard
void conv_arithmetic(const st_net* net, int conv, bool dgrad, ConvProblem& c) {
    const bool wide = dgrad ? net->wide_bwd[conv] : net->wide_fwd[conv];
    if (wide) {
        c.wgt_split = dgrad ? net->wsx_bwd[conv] : net->wsx_fwd[conv];
        c.planes = 3;
        c.elem = 0;
    } else {
        c.wgt_split = dgrad ? net->ws_bwd[conv] : net->ws_fwd[conv];
        c.planes = net->conv_planes;
        c.elem = net->conv_elem;
    }
}
}  // namespace

struct st_plan {
    const st_net* net = nullptr;
    int H = 0, W = 0;
    Node conv[13];
    Node pool[4];
    bool grads_allocated = false;
    float* content_target = nullptr;
    bool content_set = false;
    StyleHead style[5];
    float content_weight = 0.015f;
    float style_weight[5] = {256.f / 341, 64.f / 341, 16.f / 341, 4.f / 341, 1.f / 341};
    float tv_weight = 2.0f;
    float* grad_img = nullptr;       // [3][H][W] internal gradient for st_plan_step
    float* losses = nullptr;         // [8] device
    float* red_partials = nullptr;   // scratch for two-level reductions: TV [0, 4 kStreamBlocks), content MSE after it
    float* guard_scratch[3] = {nullptr, nullptr, nullptr};      // plan_range_guard: three maps of the largest activation ...
    float* guard_sums = nullptr;                                // ... and range_diff_kernel's per-block partial sums
    unsigned int* tickets = nullptr; // zeroed device words of the "last block finishes the sum" kernels (self-resetting)
    float* conv_scratch = nullptr;   // split-K workspace of the trunk convolutions (main stream only)
    float* dp_scratch = nullptr;     // conv1_1 data gradient on the padded domain, dp_parts x 3 (H + 2) (W + 2)
    int dp_parts = 1;                // channel slices of that kernel (conv_first_dgrad_parts of the GLOBAL shape)
    float* amax_word = nullptr;      // 64 bounds of kAmaxWordUints: Node::y_amax [conv], +16 g_amax [conv], +32 g_amax [pool], +48 StyleHead::s_amax
    long long bytes = 0;
    std::vector<void*> allocations;
    // strip sharding; strip == false -> the plan owns the whole image
    bool strip = false;
    int Hg = 0, row0 = 0, has_up = 0, has_down = 0;
    float* img_halo = nullptr;       // [2][3][W]
    // packed boundary rows, 64 * W floats each + the 16-float trailer whose first word is max |row| as raw bits (kHaloTrailer):
    // send_up = [rows | trailer] (lands as the upper neighbour's BOTTOM halo), send_down = [trailer | rows] (the lower
    // neighbour's TOP halo) - so that a halo block [trailer | top rows | bottom rows | trailer] receives either message contiguously
    float* send_up = nullptr;
    float* send_down = nullptr;
    unsigned int* pack_scratch = nullptr;   // launch_pack_rows' block maxima + ticket
    float* lossbuf = nullptr;        // [0] content sum of squares, [1..4] TV sums (all-reduced)
    float* gram_raw[5] = {};         // per head [C*C + C] raw moment sums (all-reduced); one contiguous block
    long long gram_total = 0;        // floats in that block
    // strip closure, device-ordered exchanges: halo rows travel on comm_stream while the interior rows of the consuming
    // convolution are computed on the caller's stream (pack_done: the boundary rows are packed; halo_landed: the
    // exchange has been enqueued behind it)
    hipStream_t comm_stream = nullptr;
    bool comm_stream_borrowed = false;     // from the process-wide probed set (shared_head_streams)
    hipStream_t chain_stream = nullptr;    // strip plans, compact layout: the owned heads' Newton-Schulz chains (else they run
                                           // on the head's own stream)
    hipEvent_t moments_ready[5] = {}, chain_done[5] = {};
    hipEvent_t pack_done = nullptr, halo_landed = nullptr;
    unsigned int* halo_bound = nullptr;     // operand bound of a boundary launch: the operand's own bound + its halo rows'

    int rank = 0, world = 1;         // position of this strip among the ranks (NS-chain ownership)
    float* head_result[5] = {};      // per head [C*C + C + 64]: Ssym | b | weighted loss term - what the owner broadcasts
    struct Phase {
        std::function<int(hipStream_t)> run;
        st_exchange ex;
        const float* halo = nullptr;         // the halo block a kind-1 exchange fills (finish_phases)
    };
    // halo blocks whose consumer is NOT cut into interior + boundary launches: their exchange is issued on the caller's
    // stream, in line between the pack kernel and the consumer (add_strip_conv, finish_phases)
    std::unordered_map<const float*, bool> halo_inline;
    std::vector<Phase> phases;
    size_t phase_pos = 0;
    const float* ph_image = nullptr;
    float* ph_grad = nullptr;
    int ph_last_layer = -1;
    unsigned ph_option_gen = 0;
    // Side streams: the five W2 style heads are ~60 dependent small launches each (latency bound),
    // so each runs on its own stream, forked when its tap is ready in the forward pass and joined
    // just before the backward pass needs that tap's gradient.  They overlap the trunk and each other.
    hipStream_t head_stream[5] = {};
    bool head_stream_owned[5] = {};        // false: borrowed from the process-wide set (shared_head_streams)
    hipEvent_t tap_ready[5] = {};
    hipEvent_t head_done[5] = {};
    bool streams_ready = false;
    int device = 0;
    // hipGraph replay of the closure.  The ~430 launches of one closure (6 streams) are captured once
    // per (image, grad, losses) pointer triple on an internal stream and replayed; the caller's stream
    // (possibly the legacy null stream, which cannot be captured) is bridged with two events.
    // OFF by default: measured on ROCm 7.2 / MI355X the replay of this 6-branch graph is bit-identical
    // but slower than eager launches (512^2: 5.9 vs 5.0 ms per step, 128^2: 3.1 vs 2.1 ms).
    bool graph_enabled = false;
    hipStream_t main_stream = nullptr;
    bool gram1_fused = false;              // this pass's conv1_1 launch left relu1_1's partial moments (run_forward)
    int gram1_splits = 0;
    bool compact_streams = false;          // ensure_streams: only the streams that carry work exist
    bool head4_on_caller = false;          // relu5_1's head runs on the caller's stream (shared_head_streams found no sharer)
    std::vector<hipStream_t> junk_streams;  // ST_STREAM_DUMMIES (experiments)
    hipEvent_t bridge_in = nullptr, bridge_out = nullptr;
    // TV (needs only the image) and the content MSE run beside the trunk on one auxiliary stream
    hipStream_t aux_stream = nullptr;
    hipEvent_t aux_in = nullptr, aux_fwd = nullptr, tv_done = nullptr, content_done = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t graph_exec = nullptr;
    const float* gk_image = nullptr;
    float* gk_grad = nullptr;
    float* gk_losses = nullptr;
    int gk_seen = 0;
    bool capturing = false;
    // ST_AMD_TIMELINE=1: timing events at step start / forward end / each head done / backward end
    bool timeline = false;
    hipEvent_t tl_start = nullptr, tl_fwd = nullptr, tl_head[5] = {}, tl_bwd = nullptr;
    hipEvent_t tl_h4[4] = {};        // relu5_1's head: chain start, after NS forward, after NS backward, (end = tl_head[4])
    hipEvent_t tl_h3[4] = {};        // the same for relu4_1's head
    int tl_count = 0;
    // profiling
    bool profiling = false;
    std::vector<ProfileEvent> events;
    size_t events_used = 0;
    std::vector<HbmEvent> hbm_events;
    size_t hbm_used = 0;
    long long prof_launches = 0;
    double prof_ms = 0, prof_flops = 0;
    // st_plan_step: the losses' total and the clearing of the fp16x3 operand bounds ride in the update kernel (AdamTail)
    bool defer_sum = false;          // loss_and_grad leaves the total to the caller
    bool amax_clean = false;         // the update kernel has cleared amax_word: the next run_forward skips its memset
    const FoldUpdate* fold_update = nullptr;     // st_plan_step: conv1_1's fold kernel applies the update (and the tail)
    bool fold_updated = false;       // ... and has done so in this closure
};

namespace {

int plan_alloc(st_plan* p, float** out, size_t floats) {
    void* ptr = nullptr;
    const size_t bytes = ((floats * sizeof(float) + 255) / 256) * 256;
    hipError_t e = hipMalloc(&ptr, bytes);
    if (e != hipSuccess) {
        set_error("hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
        return 1;
    }
    p->allocations.push_back(ptr);
    p->bytes += (long long)bytes;
    *out = static_cast<float*>(ptr);
    return 0;
}

int conv_launch_profiled(st_plan* p, const ConvProblem& prob, hipStream_t s, double flops_fraction = 1.0) {
    if (!p->profiling) return launch_conv(prob, s);
    if (p->events_used == p->events.size()) {
        ProfileEvent ev{};
        ST_HIP(hipEventCreate(&ev.start));
        ST_HIP(hipEventCreate(&ev.stop));
        p->events.push_back(ev);
    }
    ProfileEvent& ev = p->events[p->events_used++];
    ev.flops = conv_flops(prob) * flops_fraction;
    ST_HIP(hipEventRecord(ev.start, s));
    const int rc = launch_conv(prob, s);
    ST_HIP(hipEventRecord(ev.stop, s));
    return rc;
}

template <class F>
int hbm_profiled(st_plan* p, int cat, double bytes, hipStream_t s, F&& launch) {
    if (!p->profiling) return launch();
    if (p->hbm_used == p->hbm_events.size()) {
        HbmEvent ev{};
        ST_HIP(hipEventCreate(&ev.start));
        ST_HIP(hipEventCreate(&ev.stop));
        p->hbm_events.push_back(ev);
    }
    HbmEvent& ev = p->hbm_events[p->hbm_used++];
    ev.cat = cat;
    ev.bytes = bytes;
    ST_HIP(hipEventRecord(ev.start, s));
    const int rc = launch();
    ST_HIP(hipEventRecord(ev.stop, s));
    return rc;
}

const Node* feature_node(const st_plan* p, int layer) {
    for (int i = 0; i < kNumOps; ++i)
        if (kProgram[i].feat_index == layer)
            return kProgram[i].kind == 0 ? &p->conv[kProgram[i].index] : &p->pool[kProgram[i].index];
    return nullptr;
}

int style_head(st_plan* p, int idx, hipStream_t s);
int style_heads_shallow_lockstep(st_plan* p, hipStream_t s, const int* idx, int lanes);

// (compact layout: a head stream that the shipped configuration does not use - ST_HEAD_LOCKSTEP=0, the bf16 / fp32 modes -
// is created when it is first asked for)
int ensure_head_stream(st_plan* p, int k) {
    if (p->head_stream[k]) return 0;
    ST_HIP(hipStreamCreateWithFlags(&p->head_stream[k], hipStreamNonBlocking));
    p->head_stream_owned[k] = true;
    return 0;
}

void invalidate_graph(st_plan* p) {
    if (p->graph_exec) hipGraphExecDestroy(p->graph_exec);
    if (p->graph) hipGraphDestroy(p->graph);
    p->graph_exec = nullptr;
    p->graph = nullptr;
    p->gk_seen = 0;
}

// ---- the heads' streams: chosen once per process and device ------------------------------------------------------------
// How ROCm deals streams to its hardware queues is not documented (creation order and the number of streams alive both
// matter: with one / two foreign streams created first, three freshly created head streams ran 9 - 14 % slower than with
// none or three - 512^2 435 -> 394 / 373 it/s - because two of this library's chains, or a chain and the caller's trunk,
// had landed on ONE hardware queue, where the barrier bit of every dependent packet makes them run in submission order).
// So the layout is MEASURED instead of assumed: six candidate streams, a probe per hardware-queue class (st_diag.hip:
// a spinning kernel on one stream, two dependent marker kernels on the others - a second marker that does not land shares
// the spinner's queue), then relu4_1's head and the shallow heads' chains get two streams on two different queues that are
// NOT the caller's, and relu5_1's head a third queue - preferably the caller's own: the trunk waits for that head, they
// never run side by side, and it leaves the other queues to the chains that do overlap the trunk.  ~1 - 6 ms, once; every
// plan of the process borrows the same three streams (a plan per scale used to create seven streams each, shifting the
// dealing from scale to scale: the "bimodal" small scales of rounds 2 / 3).  The unused candidates are destroyed.
// ST_STREAM_PROBE=0: no probe, the first three candidates in creation order (experiments).
struct SharedStreams {
    hipStream_t head[5] = {};
    hipStream_t q[3] = {};                 // the three picks: q[0], q[1] on two different hardware queues that are not the
                                           // caller's; q[2] on a third one where the probe found one (else the caller's)
    bool ready = false;
    int classes = 0;                       // hardware-queue classes seen by the probe (0: not probed)
    bool no_sharer = false;                // probed, and no candidate sits on the caller's hardware queue
};

const SharedStreams* shared_head_streams(int device, hipStream_t caller, int order_code) {
    static SharedStreams sets[16];
    static std::mutex guard;
    if (device < 0 || device >= 16) { set_error("device index %d out of range", device); return nullptr; }
    std::lock_guard<std::mutex> lock(guard);
    SharedStreams& set = sets[device];
    if (set.ready) return &set;
    static Option probe_opt("ST_STREAM_PROBE", 1);
    constexpr int N = 6;
    hipStream_t cand[N] = {};
    for (int i = 0; i < N; ++i)
        if (hipStreamCreateWithFlags(&cand[i], hipStreamNonBlocking) != hipSuccess) { set_error("hipStreamCreate failed"); return nullptr; }
    int cls[N];
    for (int i = 0; i < N; ++i) cls[i] = -1;
    int classes = 0;
    bool probed = probe_opt.get() != 0;
    if (probed) {
        int shares[N] = {};
        if (probe_queue_sharing(caller, cand, N, shares) == 0) {           // class 0: the caller's hardware queue
            for (int i = 0; i < N; ++i)
                if (shares[i]) cls[i] = 0;
            classes = 1;
            for (int i = 0; i < N && probed; ++i) {
                if (cls[i] >= 0) continue;
                cls[i] = classes;
                hipStream_t rest[N];
                int idx[N], n = 0;
                for (int j = i + 1; j < N; ++j)
                    if (cls[j] < 0) { rest[n] = cand[j]; idx[n++] = j; }
                if (n > 0) {
                    if (probe_queue_sharing(cand[i], rest, n, shares) != 0) { probed = false; break; }
                    for (int m = 0; m < n; ++m)
                        if (shares[m]) cls[idx[m]] = classes;
                }
                ++classes;
            }
        } else {
            probed = false;
        }
    }
    int pick[3] = {-1, -1, -1};                              // relu5_1's head, relu4_1's head, the shallow heads' chains
    if (probed) {
        for (int r = 1; r < 3; ++r)                          // two chains that overlap the trunk: distinct queues, not the caller's
            for (int i = 0; i < N && pick[r] < 0; ++i)
                if (cls[i] > 0 && i != pick[1] && (pick[1] < 0 || cls[i] != cls[pick[1]])) pick[r] = i;
        // the third stream (strip plans: the communication stream; unsharded plans: relu5_1's head when it does not run on
        // the caller's stream): a third queue that is not the caller's, else the caller's
        for (int want_own = 1; want_own >= 0 && pick[0] < 0; --want_own)
            for (int i = 0; i < N && pick[0] < 0; ++i) {
                const bool other = i != pick[1] && i != pick[2] && (pick[1] < 0 || cls[i] != cls[pick[1]]) &&
                                   (pick[2] < 0 || cls[i] != cls[pick[2]]);
                if (other && (want_own ? cls[i] > 0 : true)) pick[0] = i;
            }
    }
    for (int r = 0; r < 3; ++r)                              // no probe / not enough classes: first free candidates
        for (int i = 0; i < N && pick[r] < 0; ++i)
            if (i != pick[0] && i != pick[1] && i != pick[2]) pick[r] = i;
    // order_code (ST_STREAM_ORDER, experiments): e.g. 234 hands the picks to the heads in another order
    int roles[3] = {(order_code / 100) % 10, (order_code / 10) % 10, order_code % 10};
    for (int r = 0; r < 3; ++r)
        if (roles[r] < 2 || roles[r] > 4) { roles[0] = 4; roles[1] = 3; roles[2] = 2; break; }
    for (int r = 0; r < 3; ++r) set.head[roles[r]] = cand[pick[r]];
    set.q[0] = cand[pick[1]]; set.q[1] = cand[pick[2]]; set.q[2] = cand[pick[0]];
    for (int i = 0; i < N; ++i)
        if (i != pick[0] && i != pick[1] && i != pick[2]) hipStreamDestroy(cand[i]);
    set.classes = probed ? classes : 0;
    set.no_sharer = probed && cls[pick[0]] != 0;
    set.ready = true;
    if (option_env("ST_STREAM_LOG")) {
        fprintf(stderr, "[streams] device %d: hardware-queue class of the six candidates (0 = the caller's): %d %d %d %d %d %d%s; "
                        "relu5_1's head <- candidate %d, relu4_1's <- %d, shallow chains <- %d\n",
                device, cls[0], cls[1], cls[2], cls[3], cls[4], cls[5], probed ? "" : " (not probed)", pick[0], pick[1], pick[2]);
    }
    return &set;
}

int ensure_streams(st_plan* p, hipStream_t caller = nullptr) {
    if (p->streams_ready) return 0;
    ST_HIP(hipGetDevice(&p->device));
    ST_HIP(hipEventCreateWithFlags(&p->bridge_in, hipEventDisableTiming));
    ST_HIP(hipEventCreateWithFlags(&p->bridge_out, hipEventDisableTiming));
    // ROCm maps HIP streams onto GPU_MAX_HW_QUEUES = 4 hardware queues in creation order, and two streams on one hardware
    // queue run in submission order: which head shares a queue with which decides whether relu5_1's ~50 dependent launches
    // (the window in which the trunk idles) run at their isolated speed or 1.6 x slower (profiles/r03_head_window.md
    // section 6: 356 ... 420 it/s at 512^2 over twelve creation orders of the round-3 layout - main, aux, heads 0 1 2 3 4,
    // three of them never used once the shallow heads ran in lockstep).
    // Round 4, COMPACT layout (unsharded plans; ST_STREAMS_COMPACT=0: the round-3 layout): only the streams that carry
    // work exist - relu5_1's head, relu4_1's head, the shallow heads' lockstep chains - so that together with the caller's
    // stream the plan occupies four streams = four hardware queues, nobody shares, and the creation order stops mattering.
    // TV and the content MSE, which had the auxiliary stream, run on the caller's stream right after the forward trunk:
    // that stream waits >= 0.5 ms for relu5_1's head there at every size, so they cost nothing (loss_and_grad).  The
    // graph-replay stream is created when a graph is first requested.  Measured alternatives that were NOT kept
    // (profiles/r02_ns_chains.md): relu5_1's head on the caller's stream, one hipGraph per head, one launcher thread per
    // head, a hipGraph of the whole closure, a high-priority stream for relu5_1's head.
    static Option compact_opt("ST_STREAMS_COMPACT", 1);
    static Option order_opt("ST_STREAM_ORDER", 432);          // experiments: creation order of the head streams, e.g. 234
    static Option dummies_opt("ST_STREAM_DUMMIES", 0);        // experiments: throw-away streams created first (a host
                                                              // application that made streams before loading the library)
    p->compact_streams = compact_opt.get() != 0;
    for (int i = 0; i < dummies_opt.get() && i < 8; ++i) {
        hipStream_t junk = nullptr;
        ST_HIP(hipStreamCreateWithFlags(&junk, hipStreamNonBlocking));
        p->junk_streams.push_back(junk);
    }
    if (p->compact_streams) {
        // the three head streams are chosen ONCE per process and device (shared_head_streams) and borrowed by every plan
        const SharedStreams* set = shared_head_streams(p->device, caller, order_opt.get());
        if (!set) return 1;
        if (p->strip) {
            // Strip plans: the COMMUNICATION stream must not share the trunk's hardware queue (a halo exchange behind the
            // interior launch of the same queue overlaps nothing: traced in round 4 - RCCL's kernel ran on the trunk's queue
            // and the trunk idled 0.5 ms behind a head's chain that shared it).  It takes the third probed queue.  The heads'
            // per-rank work that every rank has (Gram + reduction in forward order, broadcast + 1 x 1 step in backward order)
            // shares ONE stream; the Newton-Schulz chains a rank OWNS run on another, so that no later head's reduction - above
            // all relu5_1's, which every rank's backward waits for - queues behind a chain.
            for (int k = 0; k < 5; ++k) p->head_stream[k] = set->q[0];
            p->chain_stream = set->q[1];
            p->comm_stream = set->q[2];
            p->comm_stream_borrowed = true;
        } else {
            for (int k = 2; k < 5; ++k) p->head_stream[k] = set->head[k];
            static Option h4_opt("ST_HEAD5_ON_CALLER", 1);   // 1 (shipped): always; 0: its own stream; -1: only when the probe
                                                             // found no candidate on the caller's hardware queue
            p->head4_on_caller = h4_opt.get() < 0 ? set->no_sharer : h4_opt.get() != 0;
        }
    } else {
        ST_HIP(hipStreamCreateWithFlags(&p->main_stream, hipStreamNonBlocking));
        ST_HIP(hipStreamCreateWithFlags(&p->aux_stream, hipStreamNonBlocking));
        // (round-3 layout - DO NOT reorder: main, aux, heads 0 1 2 3 4 measured best of twelve orders)
        for (int i = 0; i < 5; ++i) {
            ST_HIP(hipStreamCreateWithFlags(&p->head_stream[i], hipStreamNonBlocking));
            p->head_stream_owned[i] = true;
        }
    }
    for (hipEvent_t* e : {&p->aux_in, &p->aux_fwd, &p->tv_done, &p->content_done})
        ST_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    for (int i = 0; i < 5; ++i) {
        ST_HIP(hipEventCreateWithFlags(&p->tap_ready[i], hipEventDisableTiming));
        ST_HIP(hipEventCreateWithFlags(&p->head_done[i], hipEventDisableTiming));
        ST_HIP(hipEventCreateWithFlags(&p->moments_ready[i], hipEventDisableTiming));
        ST_HIP(hipEventCreateWithFlags(&p->chain_done[i], hipEventDisableTiming));
    }
    const char* tl = option_env("ST_AMD_TIMELINE");
    if (tl && atoi(tl) == 1) {
        p->timeline = true;
        ST_HIP(hipEventCreate(&p->tl_start)); ST_HIP(hipEventCreate(&p->tl_fwd)); ST_HIP(hipEventCreate(&p->tl_bwd));
        for (int i = 0; i < 5; ++i) ST_HIP(hipEventCreate(&p->tl_head[i]));
        for (int i = 0; i < 4; ++i) ST_HIP(hipEventCreate(&p->tl_h4[i]));
        for (int i = 0; i < 4; ++i) ST_HIP(hipEventCreate(&p->tl_h3[i]));
    }
    p->streams_ready = true;
    return 0;
}

// fork_heads: launch each style head on its side stream as soon as its tap has been produced
int run_forward(st_plan* p, const float* image, int last_layer, hipStream_t s, bool fork_heads = false) {
    const st_net* net = p->net;
    const Node* prev = nullptr;
    const bool bounds = net->conv_elem == 1;      // fp16x3: producers leave max |y|, max |g| for the consumers
    p->gram1_fused = false;
    if (bounds && !p->amax_clean) ST_HIP(hipMemsetAsync(p->amax_word, 0, (size_t)64 * kAmaxWordUints * sizeof(float), s));
    p->amax_clean = false;
    bool pooled_by_conv = false;
    for (int i = 0; i < kNumOps; ++i) {
        const OpDesc& op = kProgram[i];
        // features[feat_index - 1] is the conv for conv ops: stop once its ReLU lies beyond last_layer
        if (op.feat_index > last_layer) break;
        if (op.kind == 0) {
            Node& n = p->conv[op.index];
            if (op.index == 0) {
                const double hw4 = 4.0 * p->H * p->W;
                // closure, fp16x3: relu1_1's Gram matrix and mean come out of this launch (conv_first_fwd_gram_kernel) instead
                // of a second pass over the largest tap; ST_CONV1_GRAM=0: the two-kernel form
                StyleHead& h0 = p->style[0];
                static Option fwd_too("ST_CONV1_GRAM_IN_FORWARD", 0);      // tests: st_plan_forward + st_plan_moments as well
                p->gram1_fused = (fork_heads || fwd_too.get() != 0) && last_layer >= 1 && bounds && h0.allocated &&
                                 conv_first_gram_applies(p->H, p->W, image, n.y, h0.gram.max_splits);
                if (hbm_profiled(p, HBM_CONV1_FWD, (3 + 64) * hw4, s, [&] {
                        if (p->gram1_fused)
                            return launch_conv_first_fwd_gram(image, net->w_first, net->bias[0], n.y, p->H, p->W, s, n.y_amax,
                                                              h0.gram.partial, h0.gram.partial_sum, h0.gram.max_splits,
                                                              &p->gram1_splits, net->w_first_l1max, net->b_first_max);
                        return launch_conv_first_fwd(image, net->w_first, net->bias[0], n.y, p->H, p->W, s, nullptr, 0, 0,
                                                     bounds ? n.y_amax : nullptr);
                    }))
                    return 1;
            } else {
                ConvProblem c{};
                c.in = prev->y; c.mask = nullptr; c.wgt = net->w_fwd[op.index]; c.bias = net->bias[op.index];
                c.out = n.y; c.cin = op.cin; c.cout = op.cout; c.height = n.h; c.width = n.w;
                c.taps = 9; c.relu = 1; c.accumulate = 0; c.scratch = p->conv_scratch;
                conv_arithmetic(net, op.index, false, c);
                c.amax_word = prev->y_amax; c.out_amax = bounds ? n.y_amax : nullptr;
                // a max pool that follows is written by this conv's epilogue where the chosen tile can (st_conv_pc.hip)
                const bool pool_next = i + 1 < kNumOps && kProgram[i + 1].kind == 1 && kProgram[i + 1].feat_index <= last_layer &&
                                       net->pooling == 0;
                if (pool_next) c.pool_out = p->pool[kProgram[i + 1].index].y;
                pooled_by_conv = pool_next && conv_pc_fuses_pool(c) && c.planes == 2 && c.elem == 1 && c.wgt_split;
                if (!pooled_by_conv) c.pool_out = nullptr;
                // closure (fork_heads): nothing but the pool reads this map - leave argmax codes instead of writing it
                // (ST_POOL_CODES=0: A/B).  st_plan_forward (targets, feature taps) always writes the map.
                static Option codes_opt("ST_POOL_CODES", 1);
                n.coded = fork_heads && pooled_by_conv && n.pool_code != nullptr && codes_opt.get() != 0;
                if (n.coded) c.pool_code = n.pool_code;
                if (conv_launch_profiled(p, c, s)) return 1;
            }
            prev = &n;
            if (fork_heads) {
                // only mark the tap here; the head's ~66 launches are enqueued after the whole trunk so
                // that the host never delays the trunk's next kernel (launch cost ~3-5 us each)
                for (int k = 0; k < 5; ++k)
                    if (kStyleConv[k] == op.index) ST_HIP(hipEventRecord(p->tap_ready[k], s));
            }
        } else {
            Node& n = p->pool[op.index];
            if (!pooled_by_conv && launch_pool_fwd(prev->y, n.y, prev->c, prev->h, prev->w, net->pooling, s)) return 1;
            pooled_by_conv = false;
            prev = &n;
        }
    }
    return 0;
}

int ensure_style_alloc(st_plan* p, int idx) {
    StyleHead& h = p->style[idx];
    if (h.allocated) return 0;
    const size_t nn = (size_t)h.n * h.n;
    float** mats[] = {&h.cov_t, &h.root_t, &h.srm, &h.cov, &h.tmat, &h.mmat, &h.root,
                      &h.gm,    &h.dt,     &h.dcov, &h.ssym};
    for (float** m : mats)
        if (plan_alloc(p, m, nn)) return 1;
    if (plan_alloc(p, &h.mean_t, h.n) || plan_alloc(p, &h.mean, h.n) || plan_alloc(p, &h.bvec, h.n) ||
        plan_alloc(p, &h.gdiag, 64))
        return 1;
    float* nsbase = nullptr;
    if (plan_alloc(p, &nsbase, ns_workspace_floats(h.n))) return 1;
    ns_workspace_carve(h.ns, nsbase, h.n);
    if (ns_workspace_reset(h.ns, nullptr)) return 1;
    ST_HIP(hipStreamSynchronize(nullptr));        // (the heads' streams are non-blocking: nothing orders them behind the null stream)
    long long splits = (16ll << 20) / (long long)nn;
    if (splits > 1024) splits = 1024;
    if (splits < 8) splits = 8;
    h.gram.max_splits = (int)splits;
    if (plan_alloc(p, &h.gram.partial, (size_t)splits * nn) ||
        plan_alloc(p, &h.gram.partial_sum, (size_t)splits * h.n))
        return 1;
    // the head's 1x1 gradient convolution runs on the head's own stream: its own split-K workspace, needed
    // only while the tap is too small to fill the chip.  launch_conv keeps >= 4 chunks of 8 channels per slice,
    // i.e. splits at most n / 32 ways, and never beyond kConvScratchFloats.
    const long long wg = ((h.npix_local + 255) / 256) * (h.n / 64);
    if (wg < 512) {
        size_t need = (size_t)(h.n / 32) * h.n * (size_t)h.npix_local;
        if (need > kConvScratchFloats) need = kConvScratchFloats;
        if (plan_alloc(p, &h.conv_scratch, need)) return 1;
    }
    h.allocated = true;
    return 0;
}

int ensure_grad_alloc(st_plan* p) {
    if (p->grads_allocated) return 0;
    for (Node& n : p->conv)
        if (plan_alloc(p, &n.g, n.count())) return 1;
    for (Node& n : p->pool)
        if (plan_alloc(p, &n.g, n.count())) return 1;
    if (plan_alloc(p, &p->grad_img, (size_t)3 * p->H * p->W)) return 1;
    p->grads_allocated = true;
    return 0;
}

int moments_of_tap(st_plan* p, int idx, float* mean_out, float* srm_out, hipStream_t s, float* cov_out = nullptr) {
    StyleHead& h = p->style[idx];
    const Node& tap = p->conv[kStyleConv[idx]];
    // (relu1_1 in the closure: conv1_1's launch has left the partials of its workgroups - run_forward)
    const bool fused = idx == 0 && p->gram1_fused;
    const int splits = fused ? p->gram1_splits : gram_choose_splits(h.n, h.npix_local, h.gram.max_splits);
    // (ST_ABLATE_SIDE bit 1: skip the Gram kernel, bit 2: skip the heads' 1x1 gradient kernel - wrong results; measures how
    // much of these HBM-bound side kernels' time is exposed in the iteration, tools/README.md)
    static Option ablate_opt("ST_ABLATE_SIDE", 0);
    // ST_GRAM_F32=1 (attribution runs, tools/grad_attribution.py): the moments on the exact fp32 matrix pipe in every mode
    static Option gram_f32("ST_GRAM_F32", 0);
    auto gram = [&] {
        if (!fused && !(ablate_opt.get() & 1) &&
            launch_gram_partial(tap.y, h.n, h.npix_local, splits, h.gram, s,
                                (p->net->conv_elem == 1 && !gram_f32.get()) ? tap.y_amax : nullptr))
            return 1;
        return launch_gram_finalize(h.gram, h.n, h.npix, splits, mean_out, srm_out, s, cov_out, kCovEps);
    };
    // relu1_1's Gram (C = 64) reads its tap once and has 64 MACs per element on the 16-bit pipe: HBM-bound
    if (idx == 0 && !fused) return hbm_profiled(p, HBM_GRAM1, 4.0 * h.n * (double)h.npix_local, s, gram);
    return gram();
}

// raw sums over the local pixels: sums = [F F^T (C*C) | F 1 (C)]  (strip mode, before the all-reduce)
int moment_sums_of_tap(st_plan* p, int idx, float* sums, hipStream_t s) {
    StyleHead& h = p->style[idx];
    const Node& tap = p->conv[kStyleConv[idx]];
    const int splits = gram_choose_splits(h.n, h.npix_local, h.gram.max_splits);
    if (launch_gram_partial(tap.y, h.n, h.npix_local, splits, h.gram, s, p->net->conv_elem == 1 ? tap.y_amax : nullptr))
        return 1;
    return launch_gram_finalize(h.gram, h.n, /*N=*/1, splits, sums + (size_t)h.n * h.n, sums, s);
}

GemmBatch one_gemm(int n, const float* a, const float* b, float* d, int ta, int tb) {
    GemmBatch g{};
    g.n = n; g.count = 1;
    g.p[0].a1 = a; g.p[0].b1 = b; g.p[0].d = d; g.p[0].ta1 = ta; g.p[0].tb1 = tb;
    g.p[0].epilogue = EPI_SCALE; g.p[0].c = 1.f;
    return g;
}

// everything after the moments (h.mean, h.srm) are known, in two parts: the C x C work - covariance, A cov A, the NS
// forward chain, the loss term, the Lyapunov backward chain, d cov -> (Ssym, b) - and the one step that touches the tap,
// dF = Ssym F + b 1^T.  Unsharded plans run both back to back; under sharding the first part runs on the head's OWNER
// rank only and (Ssym, b, loss term) are broadcast (style_head_result_* below).
int style_head_chain(st_plan* p, int idx, hipStream_t s, bool cov_ready = false);
int style_head_gradient(st_plan* p, int idx, hipStream_t s);

// StyleLossW2.forward + its backward down to the tap's feature gradient.
int style_head(st_plan* p, int idx, hipStream_t s) {
    StyleHead& h = p->style[idx];
    static Option fused_cov("ST_GRAM_FUSED_COV", 1);
    const bool with_cov = fused_cov.get() != 0;
    if (moments_of_tap(p, idx, h.mean, h.srm, s, with_cov ? h.cov : nullptr)) return 1;
    if (style_head_chain(p, idx, s, with_cov)) return 1;
    return style_head_gradient(p, idx, s);
}

int style_head_chain(st_plan* p, int idx, hipStream_t s, bool cov_ready) {
    StyleHead& h = p->style[idx];
    const int n = h.n;
    const float w = p->style_weight[idx];
    const bool tl = p->timeline && (idx == 4 || idx == 3);
    hipEvent_t* tlh = idx == 4 ? p->tl_h4 : p->tl_h3;
    if (tl) ST_HIP(hipEventRecord(tlh[0], s));
    // (cov_ready: the Gram kernel's finalize pass wrote the covariance along with the moments)
    if (!cov_ready && launch_cov_from_moments(h.mean, h.srm, h.cov, n, kCovEps, s)) return 1;
    // sqrt_term = sqrtm(cov_sqrt @ cov @ cov_sqrt)                       (style_transfer.py:179)
    if (launch_gemm_batch(one_gemm(n, h.root_t, h.cov, h.tmat, 0, 0), s)) return 1;
    // (n = 512: the Frobenius norms the two chains open with - of M and of the root, sqrtm.py:16,38 - come out of the
    // products that make those matrices instead of two launches of their own on relu5_1's critical path)
    GemmBatch mm = one_gemm(n, h.tmat, h.root_t, h.mmat, 0, 0);
    const int m_partials = gemm_sumsq_fusable(n) ? (n / 32) * (n / 32) : 0;
    if (m_partials) mm.p[0].sumsq_partials = h.ns.scalars + 8;
    if (launch_gemm_batch(mm, s)) return 1;
    // the loss term (style_transfer.py:178-181) and the seed dL/d root = gdiag * I ride in the backward chain's opening
    // kernel (one launch less on the iteration's critical path); then the Lyapunov recurrence -> dL/dM
    const W2LossJob job{h.mean, h.mean_t, h.cov, h.cov_t, h.root, n, w, p->losses + 1 + idx, h.gdiag};
    if (ns_chain_combined() && (ns_chain_mask() & (idx == 4 ? 4 : idx == 3 ? 2 : 1))) {
        // round 5: both recurrences and the loss scalars in ONE persistent launch (st_nschain.hip)
        const float* mm1[1] = {h.mmat};
        float* r1[1] = {h.root};
        float* g1[1] = {h.gm};
        NSWorkspace* w1[1] = {&h.ns};
        // ST_NS_CHAIN_DELAY=1 (experiment): a persistent chain of a head other than relu5_1 starts when the forward trunk has
        // ended - its resident workgroups then spin beside the idle window, not beside the trunk's convolution workgroups
        static Option delay_opt("ST_NS_CHAIN_DELAY", 0);
        if (delay_opt.get() && idx != 4) ST_HIP(hipStreamWaitEvent(s, p->aux_fwd, 0));
        if (ns_sqrt_chain(mm1, r1, g1, &n, w1, &m_partials, &job, 1, s)) return 1;
        if (tl) ST_HIP(hipEventRecord(tlh[1], s));
    } else {
        int root_partials = 0;
        if (ns_sqrt_forward(h.mmat, h.root, n, h.ns, s, m_partials, &root_partials)) return 1;
        if (tl) ST_HIP(hipEventRecord(tlh[1], s));
        if (ns_sqrt_backward(h.root, nullptr, h.gdiag, h.gm, n, h.ns, s, &job, root_partials)) return 1;
    }
    if (tl) ST_HIP(hipEventRecord(tlh[2], s));
    // M = (A cov) A  with A = cov_sqrt (constant):  d cov = A^T (G A^T)
    if (launch_gemm_batch(one_gemm(n, h.gm, h.root_t, h.dt, 0, 1), s)) return 1;
    if (launch_gemm_batch(one_gemm(n, h.root_t, h.dt, h.dcov, 1, 0), s)) return 1;
    const bool f16 = p->net->conv_elem == 1;
    return launch_style_grad_finish(h.dcov, h.mean, h.mean_t, n, w, h.npix, h.ssym, h.bvec, s, f16 ? h.s_amax : nullptr);
}

// The three shallow heads (relu1_1, relu2_1, relu3_1: C = 64, 128, 256) with their Newton-Schulz chains in LOCKSTEP on one
// stream: every recurrence step - and the A cov A / d cov products around the chains - is ONE launch for the three of them
// (gemm_mixed_kernel).  Their ~170 tiny dependent launches become ~60; they have a millisecond of slack each, what they
// must not do is crowd ROCm's hardware queues while relu5_1's chains run (profiles/r03_head_window.md section 6).
int style_heads_shallow_lockstep(st_plan* p, hipStream_t s, const int* idx, int lanes) {
    // idx: `lanes` (1 ... 3) of the shallow heads, in the order the backward needs them
    static Option fused_cov("ST_GRAM_FUSED_COV", 1);
    const bool with_cov = fused_cov.get() != 0;
    StyleHead* h[3];
    int n[3];
    // ST_GRAM_DEFER_PIXELS=n (experiment, default off): on images of >= n pixels hold the shallow taps' Gram kernels back
    // until the forward trunk has ended, i.e. run these image-sized launches in the window in which the trunk waits for
    // relu5_1's head instead of beside the forward convolutions (where a persistent convolution workgroup never overlaps
    // them).  Measured neutral (same box, 2 rounds: 1024^2 179.6 -> 181.0 it/s, 2048^2 51.55 -> 51.45, 2896 x 2172 33.05 ->
    // 33.15): in the window they delay relu5_1's dependent launches by what they saved before it.
    static Option defer_opt("ST_GRAM_DEFER_PIXELS", 0);
    const bool defer = defer_opt.get() > 0 && (long long)p->H * p->W >= defer_opt.get();
    for (int l = 0; l < lanes; ++l) {
        h[l] = &p->style[idx[l]];
        n[l] = h[l]->n;
        ST_HIP(hipStreamWaitEvent(s, defer ? p->aux_fwd : p->tap_ready[idx[l]], 0));
        if (moments_of_tap(p, idx[l], h[l]->mean, h[l]->srm, s, with_cov ? h[l]->cov : nullptr)) return 1;
        if (!with_cov && launch_cov_from_moments(h[l]->mean, h[l]->srm, h[l]->cov, n[l], kCovEps, s)) return 1;
    }
    auto batch3 = [&](auto make) {
        GemmBatch g{};
        g.n = 64; g.count = lanes;
        for (int l = 0; l < lanes; ++l) g.n = std::max(g.n, n[l]);
        for (int l = 0; l < lanes; ++l) {
            g.p[l] = make(l);
            g.p[l].n = n[l];
        }
        return launch_gemm_batch(g, s);
    };
    // sqrt_term = sqrtm(cov_sqrt @ cov @ cov_sqrt)                       (style_transfer.py:179)
    if (batch3([&](int l) { return one_gemm(n[l], h[l]->root_t, h[l]->cov, h[l]->tmat, 0, 0).p[0]; })) return 1;
    if (batch3([&](int l) { return one_gemm(n[l], h[l]->tmat, h[l]->root_t, h[l]->mmat, 0, 0).p[0]; })) return 1;
    const float* mm[3] = {};
    float* roots[3] = {};
    NSWorkspace* ws[3] = {};
    for (int l = 0; l < lanes; ++l) { mm[l] = h[l]->mmat; roots[l] = h[l]->root; ws[l] = &h[l]->ns; }
    W2LossJob jobs[3];
    for (int l = 0; l < lanes; ++l)
        jobs[l] = W2LossJob{h[l]->mean, h[l]->mean_t, h[l]->cov, h[l]->cov_t, h[l]->root, n[l], p->style_weight[idx[l]],
                            p->losses + 1 + idx[l], h[l]->gdiag};
    const float* croots[3] = {};
    const float* gd[3] = {};
    float* gm[3] = {};
    for (int l = 0; l < lanes; ++l) { croots[l] = h[l]->root; gd[l] = h[l]->gdiag; gm[l] = h[l]->gm; }
    if (ns_chain_combined() && (ns_chain_mask() & 1)) {
        // round 5: the three heads' forward and backward recurrences in ONE persistent launch (49 workgroups, three independent
        // barrier groups) instead of ~47 shared launches
        static Option delay_opt("ST_NS_CHAIN_DELAY", 0);
        if (delay_opt.get()) ST_HIP(hipStreamWaitEvent(s, p->aux_fwd, 0));
        if (ns_sqrt_chain(mm, roots, gm, n, ws, nullptr, jobs, lanes, s)) return 1;
    } else {
        if (ns_sqrt_forward_lockstep(mm, roots, n, ws, lanes, s)) return 1;
        if (ns_sqrt_backward_diag_lockstep(croots, gd, gm, n, ws, lanes, s, jobs)) return 1;
    }
    // M = (A cov) A  with A = cov_sqrt (constant):  d cov = A^T (G A^T)
    if (batch3([&](int l) { return one_gemm(n[l], h[l]->gm, h[l]->root_t, h[l]->dt, 0, 1).p[0]; })) return 1;
    if (batch3([&](int l) { return one_gemm(n[l], h[l]->root_t, h[l]->dt, h[l]->dcov, 1, 0).p[0]; })) return 1;
    const bool f16 = p->net->conv_elem == 1;
    for (int l = 0; l < lanes; ++l) {
        if (launch_style_grad_finish(h[l]->dcov, h[l]->mean, h[l]->mean_t, n[l], p->style_weight[idx[l]], h[l]->npix, h[l]->ssym,
                                     h[l]->bvec, s, f16 ? h[l]->s_amax : nullptr))
            return 1;
        if (style_head_gradient(p, idx[l], s)) return 1;
        ST_HIP(hipEventRecord(p->head_done[idx[l]], s));
        if (p->timeline) ST_HIP(hipEventRecord(p->tl_head[idx[l]], s));
    }
    return 0;
}

// ST_HEAD5_MASK=0: rounds 1 / 2 - conv5_1's data gradient masks its operand while staging it (single-role kernel)
static bool head5_masks_its_gradient() {
    static Option opt("ST_HEAD5_MASK", 1);
    return opt.get() != 0;
}

int style_head_gradient(st_plan* p, int idx, hipStream_t s) {
    StyleHead& h = p->style[idx];
    const int n = h.n;
    Node& tap = p->conv[kStyleConv[idx]];
    const bool f16 = p->net->conv_elem == 1;
    // dF = Ssym F + b 1^T : a 1x1 convolution over the tap; WRITES the tap's gradient buffer
    ConvProblem c{};
    c.in = tap.y; c.mask = nullptr; c.wgt = h.ssym; c.bias = h.bvec; c.out = tap.g;
    c.cin = n; c.cout = n; c.height = tap.h; c.width = tap.w; c.taps = 1; c.relu = 0; c.accumulate = 0;
    c.out_amax = f16 ? tap.g_amax : nullptr;
    static Option head_f32("ST_HEAD_1X1_F32", 0);          // attribution runs: the heads' 1x1 gradient step in exact fp32
    if (f16 && !head_f32.get()) {       // large taps: fp16x3 1x1 kernel (st_conv1x1.hip); launch_conv keeps split-K problems on fp32
        c.planes = 2; c.elem = 1; c.amax_word = tap.y_amax; c.wgt_amax = h.s_amax;
    }
    c.scratch = h.conv_scratch;
    // relu5_1's gradient is final as this launch leaves it (nothing accumulates into the top of the trunk), so its
    // threshold_backward is applied HERE, by the producer, like everywhere else in the backward pass - and conv5_1's data
    // gradient stages one operand stream and runs on the producer / consumer kernel (head5_masks_its_gradient)
    if (idx == 4 && head5_masks_its_gradient()) c.out_mask = tap.y;
    static Option ablate_opt("ST_ABLATE_SIDE", 0);
    if (ablate_opt.get() & 2) return 0;
    const long long npix = (long long)tap.h * tap.w;
    if (head_dgrad_small_applies(n, npix))       // a tap of <= 1024 pixels: one small-GEMM launch instead of split-K + reduce
        return hbm_profiled(p, HBM_HEAD_1X1, 2.0 * n * (double)npix * sizeof(float), s, [&] {
            return launch_head_dgrad_small(h.ssym, tap.y, h.bvec, c.out_mask, tap.g, n, npix, c.out_amax, s);
        });
    // (not part of the `roofline` bracket, which is the 3x3 trunk kernel's: on the large taps this step is HBM-bound -
    // read F, write dF - and reported under roofline_hbm)
    return hbm_profiled(p, HBM_HEAD_1X1, 2.0 * n * (double)tap.h * tap.w * sizeof(float), s, [&] { return launch_conv(c, s); });
}

// Sharded plans: the head's OWNER rank has run style_head_chain; (Ssym | b | loss term) travel in one block.
int style_head_result_pack(st_plan* p, int idx, hipStream_t s) {          // owner, before the broadcast
    StyleHead& h = p->style[idx];
    const size_t nn = (size_t)h.n * h.n;
    float* r = p->head_result[idx];
    ST_HIP(hipMemcpyAsync(r, h.ssym, nn * sizeof(float), hipMemcpyDeviceToDevice, s));
    ST_HIP(hipMemcpyAsync(r + nn, h.bvec, h.n * sizeof(float), hipMemcpyDeviceToDevice, s));
    ST_HIP(hipMemcpyAsync(r + nn + h.n, p->losses + 1 + idx, sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}
int style_head_result_unpack(st_plan* p, int idx, hipStream_t s) {        // every rank, after the broadcast
    StyleHead& h = p->style[idx];
    const size_t nn = (size_t)h.n * h.n;
    const float* r = p->head_result[idx];
    ST_HIP(hipMemcpyAsync(h.ssym, r, nn * sizeof(float), hipMemcpyDeviceToDevice, s));
    ST_HIP(hipMemcpyAsync(h.bvec, r + nn, h.n * sizeof(float), hipMemcpyDeviceToDevice, s));
    ST_HIP(hipMemcpyAsync(p->losses + 1 + idx, r + nn + h.n, sizeof(float), hipMemcpyDeviceToDevice, s));
    // the fp16x3 1x1 kernel scales Ssym by a bound on max |Ssym|: measured here on every rank (the owner's epilogue
    // bound stays on the owner)
    if (p->net->conv_elem == 1 && launch_amax(h.ssym, (long long)nn, h.s_amax, 0, s)) return 1;
    return 0;
}

bool conv_is_tap(int conv_index) {
    if (conv_index == kContentConv) return true;
    for (int k : kStyleConv)
        if (k == conv_index) return true;
    return false;
}

int join_head_for_conv(st_plan* p, int conv_index, hipStream_t s) {
    for (int k = 0; k < 5; ++k) {
        if (kStyleConv[k] != conv_index) continue;
        ST_HIP(hipStreamWaitEvent(s, p->head_done[k], 0));
    }
    return 0;
}

int run_backward(st_plan* p, float* grad_image, hipStream_t s) {
    const st_net* net = p->net;
    for (int i = kNumOps - 1; i >= 0; --i) {
        const OpDesc& op = kProgram[i];
        if (op.kind == 0) {
            Node& n = p->conv[op.index];
            // this conv's output gradient is about to be read: its style head (if any) must be done
            if (join_head_for_conv(p, op.index, s)) return 1;
            if (op.index == 0) {
                ST_HIP(hipStreamWaitEvent(s, p->tv_done, 0));
                // grad_image already holds the TV gradient -> accumulate
                // relu1_1's gradient was masked by conv1_2's data-gradient epilogue (out_mask)
                if (hbm_profiled(p, HBM_CONV1_DGRAD, (64 + 3 + 3) * 4.0 * p->H * p->W, s, [&] {
                        return launch_conv_first_dgrad(n.g, nullptr, net->w_first, grad_image, p->dp_scratch, p->H, p->W, 1, s, nullptr, 0, 0,
                                                       p->dp_parts, p->fold_update);
                    }))
                    return 1;
                p->fold_updated = p->fold_update != nullptr;
                continue;
            }
            const OpDesc& pop = kProgram[i - 1];
            Node& in = (pop.kind == 0) ? p->conv[pop.index] : p->pool[pop.index];
            // ... and this launch ACCUMULATES into the input node's gradient: if that node is a style
            // tap, its head (which WRITES the buffer first) must have finished
            if (pop.kind == 0 && join_head_for_conv(p, pop.index, s)) return 1;
            if (pop.kind == 0 && pop.index == kContentConv)
                ST_HIP(hipStreamWaitEvent(s, p->content_done, 0));
            ConvProblem c{};
            // threshold_backward: every gradient tensor is masked by its PRODUCER (the previous data-gradient
            // conv's out_mask, or pool_bwd), so the staging needs no mask stream - except at the top, where the
            // gradient comes straight from relu5_1's style head (whose 1x1 launch masks it, style_head_gradient)
            c.in = n.g; c.mask = (op.index == kStyleConv[4] && !head5_masks_its_gradient()) ? n.y : nullptr;
            c.out_mask = (pop.kind == 0) ? in.y : nullptr;
            c.wgt = net->w_bwd[op.index]; c.bias = nullptr; c.out = in.g;
            c.cin = op.cout; c.cout = op.cin; c.height = n.h; c.width = n.w; c.taps = 9; c.relu = 0;
            c.accumulate = (pop.kind == 0 && conv_is_tap(pop.index)) ? 1 : 0;
            c.scratch = p->conv_scratch;
            conv_arithmetic(net, op.index, true, c);
            c.amax_word = n.g_amax; c.out_amax = net->conv_elem == 1 ? in.g_amax : nullptr;
            if (conv_launch_profiled(p, c, s)) return 1;
        } else {
            Node& n = p->pool[op.index];
            const OpDesc& pop = kProgram[i - 1];            // always a conv
            Node& in = p->conv[pop.index];
            // reads the saved map (argmax + ReLU mask) - or the codes the forward left instead of it - and the pooled
            // gradient, writes the full-resolution gradient
            const double bytes = in.coded ? (in.count() + 1.25 * n.count()) * 4.0 : (2.0 * in.count() + n.count()) * 4.0;
            if (hbm_profiled(p, HBM_POOL_BWD, bytes, s, [&] {
                    if (in.coded) return launch_pool_bwd_codes(in.pool_code, n.g, in.g, in.c, in.h, in.w, s);
                    return launch_pool_bwd(in.y, n.g, in.g, in.c, in.h, in.w, net->pooling, s);
                }))
                return 1;
        }
    }
    return 0;
}

int loss_and_grad(st_plan* p, const float* image, float* grad_out, float* losses_out, hipStream_t s) {
    ST_REQUIRE(p->content_set, "content target not set (st_plan_set_content_target)");
    for (int i = 0; i < 5; ++i)
        ST_REQUIRE(p->style[i].target_set, "style target %d not set (st_plan_set_style_target)", i);
    if (ensure_grad_alloc(p)) return 1;
    if (ensure_streams(p, s)) return 1;
    if (p->timeline) ST_HIP(hipEventRecord(p->tl_start, s));
    // TVLoss on the un-normalised image (style_transfer.py:376): WRITES grad_out.  It needs nothing but the image,
    // so it runs on the auxiliary stream from the start of the iteration (at 2048^2 it is longer than the style
    // heads' window and used to extend the critical path); joined before conv1_1's data gradient folds into grad_out.
    // Compact stream layout (round 4): no auxiliary stream.  TV is the FIRST thing of the iteration on the shallow heads'
    // stream (idle until relu3_1 exists; the slot beside the first forward convolutions that rounds 1 - 3 gave it on the
    // auxiliary stream), the content MSE (needed by conv4_3's data gradient, which runs after relu5_1's head) the last
    // thing on relu4_1's head stream - both off the caller's stream and off relu5_1's critical chain.
    // (TV at the TAIL of a head stream - beside the backward trunk or the other heads' chains - was tried first and showed
    // a flaky TV term: a few workgroups' horizontal sums one image row too large, 1e-4 ... 5e-4 on the term in one run of
    // two, never in isolation; adding unrelated accumulators to the kernel made it vanish.  Root cause not found - an
    // instruction-level hazard of that kernel under co-residency is the best guess - so the placement that three rounds
    // of parity runs have verified stays.  profiles/r04_streams.md.)
    auto tv = [&](hipStream_t ts) {
        return hbm_profiled(p, HBM_TV, 2.0 * 3 * 4.0 * p->H * p->W, ts, [&] {
            return launch_tv(image, p->H, p->W, p->tv_weight, grad_out, p->red_partials, p->losses + 6, ts, p->tickets + 0);
        });
    };
    // ST_TV_SLOT=1 (diagnostic): the round-4 slot in which the TV term came out flaky - the TAIL of the shallow heads' stream,
    // beside the backward trunk (tests/test_tv_hazard_gpu.py, profiles/r05_tv_hazard.md)
    static Option tv_slot_opt("ST_TV_SLOT", 0);
    const bool tv_tail = tv_slot_opt.get() == 1 && !p->aux_stream;
    if (!tv_tail) {
        static Option lockstep_tv("ST_HEAD_LOCKSTEP", 1);
        const int tvk = (lockstep_tv.get() != 0 && p->net->conv_elem == 1) ? 2 : 0;
        if (!p->aux_stream && ensure_head_stream(p, tvk)) return 1;
        hipStream_t tvs = p->aux_stream ? p->aux_stream : p->head_stream[tvk];
        ST_HIP(hipEventRecord(p->aux_in, s));
        ST_HIP(hipStreamWaitEvent(tvs, p->aux_in, 0));
        if (tv(tvs)) return 1;
        ST_HIP(hipEventRecord(p->tv_done, tvs));
    }
    if (run_forward(p, image, 29, s, /*fork_heads=*/true)) return 1;
    if (p->timeline) ST_HIP(hipEventRecord(p->tl_fwd, s));
    // ContentLossMSE on relu4_2: WRITES that tap's gradient buffer (auxiliary stream, or the tail of relu4_1's head stream;
    // joined before conv4_3's data gradient accumulates into it)
    Node& ct = p->conv[kContentConv];
    ST_HIP(hipEventRecord(p->aux_fwd, s));
    auto content = [&](hipStream_t cstream) {
        ST_HIP(hipStreamWaitEvent(cstream, p->aux_fwd, 0));
        if (hbm_profiled(p, HBM_CONTENT, 3.0 * 4.0 * ct.count(), cstream, [&] {
                return launch_content_mse(ct.y, p->content_target, (long long)ct.count(), p->content_weight, ct.g,
                                          p->red_partials + 4 * kStreamBlocks, p->losses + 0, cstream, p->tickets + 64);
            }))
            return 1;
        ST_HIP(hipEventRecord(p->content_done, cstream));
        return 0;
    };
    if (p->aux_stream && content(p->aux_stream)) return 1;
    // style heads: one side stream each, gated on their tap's event, enqueued in the order the backward pass needs
    // them: relu5_1's chain gates the whole backward, relu1_1's is needed last - the host must not spend ~1 ms
    // enqueueing the other heads before the critical one
    // the three shallow heads in lockstep on relu3_1's stream (style_heads_shallow_lockstep); ST_HEAD_LOCKSTEP=0: one
    // stream and ~62 launches per head, as rounds 1 / 2 (A/B: 512^2 416.5 -> 422.8 it/s, 256^2 650 -> 664, 181^2 646 / 609 ->
    // 656, 128^2 588 ... 619 -> 753 ... 779 - the slow mode of the small scales, where a shallow head shared relu5_1's
    // hardware queue, is gone)
    static Option lockstep_opt("ST_HEAD_LOCKSTEP", 1);
    const bool lockstep = lockstep_opt.get() != 0 && p->net->conv_elem == 1;
    for (int k = 4; k >= (lockstep ? 3 : 0); --k) {
        // (head4_on_caller, the shipped form: relu5_1's head runs on the caller's stream itself - the trunk waits for that
        // head anyway, and in-queue ordering is cheaper than an event across two hardware queues: 256^2 642 -> 664 it/s,
        // 512^2 404 -> 410 against a stream of its own on a third queue; equal to a stream that happens to share the
        // caller's queue.  profiles/r04_streams.md)
        const bool on_caller = k == 4 && p->head4_on_caller;
        if (!on_caller && ensure_head_stream(p, k)) return 1;
        hipStream_t hs = on_caller ? s : p->head_stream[k];
        if (!on_caller) ST_HIP(hipStreamWaitEvent(hs, p->tap_ready[k], 0));
        if (style_head(p, k, hs)) return 1;
        ST_HIP(hipEventRecord(p->head_done[k], hs));
        if (p->timeline) ST_HIP(hipEventRecord(p->tl_head[k], hs));
        if (k == 3 && !p->aux_stream && content(p->head_stream[3])) return 1;
    }
    if (lockstep) {
        // (tried, round 4: relu3_1's head on a stream of its own and only relu2_1 / relu1_1 sharing launches, because at 128^2
        // the three-lane chain ends 0.3 ms after relu5_1's head and the backward trunk waits for it - 128^2 808 -> 794 it/s,
        // 181^2 685.6 -> 679.4, 256^2 670.8 -> 663.2: a third side stream costs more in the queues than the shorter chain gains)
        const int three[3] = {2, 1, 0};
        if (ensure_head_stream(p, 2) || style_heads_shallow_lockstep(p, p->head_stream[2], three, 3)) return 1;
    }
    if (tv_tail) {                                   // (behind the last shallow head: relu1_1's stream without lockstep)
        hipStream_t tvs = p->head_stream[lockstep ? 2 : 0];
        if (tv(tvs)) return 1;
        ST_HIP(hipEventRecord(p->tv_done, tvs));
    }
    if (run_backward(p, grad_out, s)) return 1;      // joins every style head along the way
    if (!p->defer_sum && launch_sum_losses(p->losses, s, losses_out)) return 1;
    if (p->timeline) {
        ST_HIP(hipEventRecord(p->tl_bwd, s));
        if (++p->tl_count % 10 == 0) {
            ST_HIP(hipEventSynchronize(p->tl_bwd));
            float f = 0, b = 0, h[5] = {};
            hipEventElapsedTime(&f, p->tl_start, p->tl_fwd);
            hipEventElapsedTime(&b, p->tl_start, p->tl_bwd);
            for (int i = 0; i < 5; ++i) hipEventElapsedTime(&h[i], p->tl_start, p->tl_head[i]);
            float c4[3] = {};
            for (int i = 0; i < 3; ++i) hipEventElapsedTime(&c4[i], p->tl_start, p->tl_h4[i]);
            fprintf(stderr, "[timeline] relu5_1 head: moments known %.3f | NS forward done %.3f | NS backward done %.3f | gradient written %.3f ms\n",
                    c4[0], c4[1], c4[2], h[4]);
            float c3[3] = {};
            for (int i = 0; i < 3; ++i) hipEventElapsedTime(&c3[i], p->tl_start, p->tl_h3[i]);
            fprintf(stderr, "[timeline] relu4_1 head: moments known %.3f | NS forward done %.3f | NS backward done %.3f | gradient written %.3f ms\n",
                    c3[0], c3[1], c3[2], h[3]);
            fprintf(stderr, "[timeline] forward end %.3f ms | heads done %.3f %.3f %.3f %.3f %.3f | backward end %.3f ms\n",
                    f, h[0], h[1], h[2], h[3], h[4], b);
        }
    }
    return 0;                                  // (losses_out was written by the sum kernel)
}


// ---- strip-sharded closure as a resumable sequence of phases -----------------
st_exchange no_exchange() {
    st_exchange e{};
    e.kind = 3;
    return e;
}
constexpr int kHaloTrailer = 16;            // floats; word 0 = the sender's max |row| (raw bits), the rest unused (64-byte alignment)
st_exchange halo_exchange(st_plan* p, float* halo, int channels, int width) {
    st_exchange e{};
    const size_t row = (size_t)channels * width;
    e.kind = 1;
    e.count = (long long)row + kHaloTrailer;
    e.send_up = p->has_up ? p->send_up : nullptr;                       // [rows | trailer]
    e.send_down = p->has_down ? p->send_down : nullptr;                 // [trailer | rows]
    e.recv_up = p->has_up ? halo - kHaloTrailer : nullptr;              // [trailer | top rows]
    e.recv_down = p->has_down ? halo + row : nullptr;                   // [bottom rows | trailer]
    return e;
}
// a node's boundary rows (masked where `mask` is given) into the two messages, with their bounds
int pack_halo_rows(st_plan* p, const float* src, const float* mask, int channels, int height, int width, hipStream_t s) {
    const size_t row = (size_t)channels * width;
    return launch_pack_rows(src, mask, channels, height, width, p->send_up, p->send_down + kHaloTrailer, s,
                            reinterpret_cast<unsigned int*>(p->send_up + row), reinterpret_cast<unsigned int*>(p->send_down),
                            p->pack_scratch);
}
// a halo block of `floats` payload floats with room for the two trailers around it
int halo_alloc(st_plan* p, float** out, size_t floats) {
    float* base = nullptr;
    if (plan_alloc(p, &base, floats + 2 * kHaloTrailer)) return 1;
    if (hipMemset(base, 0, (floats + 2 * kHaloTrailer) * sizeof(float)) != hipSuccess) { set_error("hipMemset of a halo block failed"); return 1; }
    *out = base + kHaloTrailer;
    return 0;
}
st_exchange allreduce_exchange(float* buffer, long long count) {
    st_exchange e{};
    e.kind = 2;
    e.count = count;
    e.buffer = buffer;
    return e;
}
st_exchange on_stream(st_exchange e, hipStream_t stream, int channel) {
    e.stream = stream;
    e.channel = channel;
    return e;
}
st_exchange rooted_exchange(int kind, float* buffer, long long count, int root) {
    st_exchange e{};
    e.kind = kind;          // 4: reduce (sum) to `root`, 5: broadcast from `root`
    e.count = count;
    e.buffer = buffer;
    e.root = root;
    return e;
}

struct PhaseBuilder {
    st_plan* p;
    std::vector<std::function<int(hipStream_t)>> pending;
    void add(std::function<int(hipStream_t)> f) { pending.push_back(std::move(f)); }
    void flush(st_exchange ex, const float* halo = nullptr) {
        auto steps = std::move(pending);
        pending.clear();
        st_plan::Phase ph;
        ph.run = [steps](hipStream_t s) {
            for (const auto& f : steps)
                if (f(s)) return 1;
            return 0;
        };
        ph.ex = ex;
        ph.halo = halo;
        p->phases.push_back(std::move(ph));
    }
};

// ---- strip plans: device-ordered exchanges -----------------------------------------------------------------------
// A halo exchange is issued on the plan's comm_stream behind the kernel that packed the boundary rows, and the
// convolution that consumes the halo is cut into an interior launch (no halo row needed: runs on the caller's stream
// while the rows are in flight) and a boundary launch behind the exchange (ConvProblem::overlap_part) wherever the
// cost model says the cut costs less than the exchange it hides (conv_pc_overlap_choice).  Transports that are not
// stream-ordered (the single-process lockstep emulation, gloo) perform every exchange synchronously between two
// phases; the event plumbing below is then a no-op and the results are the same.
int ensure_comm_stream(st_plan* p) {
    if (p->pack_done) return 0;
    if (ensure_streams(p)) return 1;               // (compact layout: the probed set provides the communication stream)
    if (!p->comm_stream) ST_HIP(hipStreamCreateWithFlags(&p->comm_stream, hipStreamNonBlocking));
    ST_HIP(hipEventCreateWithFlags(&p->pack_done, hipEventDisableTiming));
    ST_HIP(hipEventCreateWithFlags(&p->halo_landed, hipEventDisableTiming));
    return 0;
}

// Round 6: an exchange whose consumer is not cut has nothing to overlap with - the caller's stream records an event, the
// communication stream waits for it, carries the exchange, records an event, the caller's stream waits for that: two hops
// between hardware queues, ~22 us of idle trunk per exchange with nothing in flight (13 of a closure's 26 exchanges at
// 2896 x 2172 / 8: profiles/r06_strip_breakdown.md).  Those exchanges are issued IN LINE on the caller's stream instead
// (st_exchange::stream = null: "the stream the phase ran on"), between the pack kernel and the consumer, with no event at all.
// Operations of the trunk's communicator stay ordered: an in-line exchange follows the previous consumer's boundary launch
// (which waited for the communication stream), and the next comm_after_pack makes the communication stream wait for the
// caller's.  ST_STRIP_INLINE=0: every halo exchange on the communication stream (the round-4 / 5 form).
bool inline_exchanges() {
    static Option inline_opt("ST_STRIP_INLINE", 1);
    static Option shipped_bound("ST_STRIP_HALO_BOUND", 1);      // (the round-4 bound is built on the communication stream)
    return inline_opt.get() != 0 && shipped_bound.get() != 0;
}
bool halo_is_inline(const st_plan* p, const float* halo) {
    auto it = p->halo_inline.find(halo);
    return it != p->halo_inline.end() && it->second;
}
// after the pack kernel: the exchange (issued by the transport on comm_stream) must start behind it
int comm_after_pack(st_plan* p, hipStream_t s, const float* halo) {
    if (halo_is_inline(p, halo)) return 0;         // (looked up when the phase RUNS: the consumer has been built by then)
    ST_HIP(hipEventRecord(p->pack_done, s));
    ST_HIP(hipStreamWaitEvent(p->comm_stream, p->pack_done, 0));
    return 0;
}
// before the first kernel that reads the halo block: wait for everything enqueued on comm_stream so far
int join_comm(st_plan* p, hipStream_t s, const float* halo) {
    if (halo_is_inline(p, halo)) return 0;
    ST_HIP(hipEventRecord(p->halo_landed, p->comm_stream));
    ST_HIP(hipStreamWaitEvent(s, p->halo_landed, 0));
    return 0;
}

// NS chains under sharding: head k's C x C work (everything between its Gram matrix and (Ssym, b)) is identical on every
// rank, so ONE rank - its owner - runs it and broadcasts the result: the two n = 512 chains land on different GPUs
// (no mutual slowdown, SURVEY.md 8(e) "layers are assigned to ranks") and the other ranks' GPUs stay free for the trunk.
// ST_STRIP_NS_OWNER=0: every rank runs every chain on the all-reduced moments (round-1 / round-2 behaviour).
int head_owner(const st_plan* p, int k) { return (4 - k) % p->world; }
bool heads_owned(const st_plan* p) {
    static Option owner_opt("ST_STRIP_NS_OWNER", 1);
    return p->world > 1 && owner_opt.get() != 0;
}

// fp16x3: the neighbours' halo rows are operands too, so the launch that reads them needs a bound over the operand AND
// its halo rows.  Built on the communication stream behind the exchange (it runs while the interior launch does; on the
// compute stream it was two 6.5 us launches per convolution, 0.3 ms per iteration and rank at 2896 x 2172 / 8), in a
// COPY of the operand's bound: the operand's own word may be being read - by the interior launch, by the tap's Gram
// kernel on a side stream - and must not change under its readers.
// Round 5: the SENDER measures max |row| while it packs the rows and ships the word with them (halo_exchange's trailers);
// the kernels take the maximum of the operand's own bound and the two trailer words - nothing runs between the halo's arrival
// and the boundary launch (ST_STRIP_HALO_BOUND=0: the round-4 form, a copy + an amax launch on the communication stream).
int bound_with_halo(st_plan* p, ConvProblem& c) {
    if (c.elem != 1 || !c.amax_word || !c.in_halo) return 0;
    static Option shipped("ST_STRIP_HALO_BOUND", 1);
    if (shipped.get()) {
        const size_t row = (size_t)c.cin * c.width;
        c.halo_bound_up = c.has_up ? reinterpret_cast<const unsigned int*>(c.in_halo - kHaloTrailer) : nullptr;
        c.halo_bound_down = c.has_down ? reinterpret_cast<const unsigned int*>(c.in_halo + 2 * row) : nullptr;
        c.halo_amax_folded = 1;
        return 0;
    }
    ST_HIP(hipMemcpyAsync(p->halo_bound, c.amax_word, (size_t)kAmaxWordUints * sizeof(unsigned int), hipMemcpyDeviceToDevice,
                          p->comm_stream));
    c.amax_word = p->halo_bound;
    if (fold_halo_amax(c, p->comm_stream)) return 1;
    c.halo_amax_folded = 1;
    return 0;
}

// one strip convolution (forward or data gradient) whose operand halo is in flight on comm_stream
void add_strip_conv(st_plan* p, PhaseBuilder& b, const ConvProblem& whole, std::function<void(ConvProblem&)> late) {
    // `late` fills what is only known when the phase runs (nothing today besides the profile hook's state); the split
    // decision is a pure function of the shapes and is taken here, once
    PcOverlap o{};
    ConvProblem probe = whole;
    const bool split = conv_pc_overlap_choice(probe, &o) && o.pays && whole.in_halo != nullptr;
    if (whole.in_halo) p->halo_inline[whole.in_halo] = !split && inline_exchanges();
    if (split) {
        const double edge = (o.rows_b + o.rows_bottom) / (double)whole.height;      // share of the rows (and FLOPs) in the boundary launch
        b.add([=](hipStream_t s) {
            ConvProblem c = whole;
            late(c);
            c.overlap_part = 1;
            c.in_halo = nullptr; c.has_up = 0; c.has_down = 0;
            return conv_launch_profiled(p, c, s, 1.0 - edge);
        });
        b.add([=](hipStream_t s) {
            ConvProblem c = whole;
            late(c);
            c.overlap_part = 2;
            if (bound_with_halo(p, c)) return 1;
            if (join_comm(p, s, c.in_halo)) return 1;
            return conv_launch_profiled(p, c, s, edge);
        });
    } else {
        b.add([=](hipStream_t s) {
            ConvProblem c = whole;
            late(c);
            if (bound_with_halo(p, c)) return 1;
            if (join_comm(p, s, c.in_halo)) return 1;
            return conv_launch_profiled(p, c, s);
        });
    }
}

// fork_heads (closure only): right after a style tap is produced its local moment sums are computed on the head's side
// stream and reduced there (to the head's owner, or all-reduced), so neither the Gram kernel nor the collective sits on
// the trunk's stream; the owner's chain follows on the same stream and overlaps the remaining forward pass.
void build_forward_phases(st_plan* p, PhaseBuilder& b, const float* image, int last_layer, bool fork_heads = false) {
    const st_net* net = p->net;
    const int W = p->W;
    const bool f16 = net->conv_elem == 1;
    // the image's own boundary rows (conv1_1's replicate pad applies only at the global border; TV too): 35 KB, exchanged
    // on the caller's stream
    b.add([=](hipStream_t s) {
        if (ensure_comm_stream(p)) return 1;
        if (f16)      // fp16x3: Node::y_amax / g_amax of this pass
            ST_HIP(hipMemsetAsync(p->amax_word, 0, (size_t)64 * kAmaxWordUints * sizeof(float), s));
        return pack_halo_rows(p, image, nullptr, 3, p->H, W, s);
    });
    b.flush(halo_exchange(p, p->img_halo, 3, W));
    Node* prev = nullptr;
    for (int i = 0; i < kNumOps; ++i) {
        const OpDesc op = kProgram[i];
        if (op.feat_index > last_layer) break;
        Node* n = (op.kind == 0) ? &p->conv[op.index] : &p->pool[op.index];
        if (op.kind == 0 && op.index == 0) {
            b.add([=](hipStream_t s) {
                return launch_conv_first_fwd(image, net->w_first, net->bias[0], n->y, p->H, W, s, p->img_halo,
                                             p->has_up, p->has_down, f16 ? n->y_amax : nullptr);
            });
        } else if (op.kind == 0) {
            Node* in = prev;
            ConvProblem c{};
            c.in = in->y; c.wgt = net->w_fwd[op.index]; c.bias = net->bias[op.index]; c.out = n->y;
            c.cin = op.cin; c.cout = op.cout; c.height = n->h; c.width = n->w; c.taps = 9; c.relu = 1;
            c.scratch = p->conv_scratch; c.in_halo = in->yhalo; c.has_up = p->has_up; c.has_down = p->has_down;
            conv_arithmetic(net, op.index, false, c);
            c.amax_word = in->y_amax; c.out_amax = f16 ? n->y_amax : nullptr;
            // a following max pool is fused into the epilogue where the tile allows (as in run_forward); with the
            // interior / boundary cut both launches must be able to (PcOverlap::pool)
            const bool pool_next = i + 1 < kNumOps && kProgram[i + 1].kind == 1 && kProgram[i + 1].feat_index <= last_layer &&
                                   net->pooling == 0;
            if (pool_next) c.pool_out = p->pool[kProgram[i + 1].index].y;
            PcOverlap o{};
            const bool split = conv_pc_overlap_choice(c, &o) && o.pays;
            const bool fused = pool_next && c.planes == 2 && c.elem == 1 && c.wgt_split &&
                               (split ? o.pool : conv_pc_fuses_pool(c));
            if (!fused) c.pool_out = nullptr;
            n->pooled_by_conv = fused;
            static Option codes_opt("ST_POOL_CODES", 1);
            n->coded = fork_heads && fused && n->pool_code != nullptr && codes_opt.get() != 0;      // (see run_forward)
            if (n->coded) c.pool_code = n->pool_code;
            add_strip_conv(p, b, c, [](ConvProblem&) {});
        } else {
            Node* in = prev;
            b.add([=](hipStream_t s) {
                if (in->pooled_by_conv) return 0;
                return launch_pool_fwd(in->y, n->y, in->c, in->h, in->w, net->pooling, s);
            });
        }
        prev = n;
        if (fork_heads && op.kind == 0) {
            for (int k = 0; k < 5; ++k) {
                if (kStyleConv[k] != op.index) continue;
                const bool owned = heads_owned(p);
                const int owner = head_owner(p, k);
                const long long nn = (long long)p->style[k].n * p->style[k].n;
                b.add([=](hipStream_t s) {
                    if (ensure_streams(p)) return 1;
                    hipStream_t hs = p->head_stream[k];
                    ST_HIP(hipEventRecord(p->tap_ready[k], s));
                    ST_HIP(hipStreamWaitEvent(hs, p->tap_ready[k], 0));
                    return moment_sums_of_tap(p, k, p->gram_raw[k], hs);
                });
                // (the first phase of a closure has created the streams; before that the handle is null and the
                // descriptor is rebuilt - see st_plan_closure_begin)
                st_exchange ex = owned ? rooted_exchange(4, p->gram_raw[k], nn + p->style[k].n, owner)
                                       : allreduce_exchange(p->gram_raw[k], nn + p->style[k].n);
                b.flush(on_stream(ex, p->head_stream[k], 1));
                b.add([=](hipStream_t) {
                    StyleHead& h = p->style[k];
                    if (owned && p->rank != owner) return 0;               // the owner's result arrives by broadcast
                    // the chain runs on the chain stream (compact layout) behind this head's reduction
                    hipStream_t hs = p->chain_stream ? p->chain_stream : p->head_stream[k];
                    ST_HIP(hipEventRecord(p->moments_ready[k], p->head_stream[k]));
                    ST_HIP(hipStreamWaitEvent(hs, p->moments_ready[k], 0));
                    if (launch_div_by_scalar(p->gram_raw[k], (float)h.npix, h.srm, nn, hs)) return 1;
                    if (launch_div_by_scalar(p->gram_raw[k] + nn, (float)h.npix, h.mean, h.n, hs)) return 1;
                    if (style_head_chain(p, k, hs)) return 1;
                    if (owned) {
                        if (style_head_result_pack(p, k, hs)) return 1;
                        ST_HIP(hipEventRecord(p->chain_done[k], hs));      // (the broadcast on the head's stream waits for it)
                        return 0;
                    }
                    if (style_head_gradient(p, k, hs)) return 1;
                    ST_HIP(hipEventRecord(p->head_done[k], hs));
                    return 0;
                });
            }
        }
        const bool next_is_conv = (i + 1 < kNumOps) && kProgram[i + 1].kind == 0 &&
                                  kProgram[i + 1].feat_index <= last_layer;
        if (next_is_conv && n->yhalo) {
            b.add([=](hipStream_t s) {
                if (pack_halo_rows(p, n->y, nullptr, n->c, n->h, n->w, s)) return 1;
                return comm_after_pack(p, s, n->yhalo);
            });
            b.flush(on_stream(halo_exchange(p, n->yhalo, n->c, n->w), p->comm_stream, 0), n->yhalo);
        }
    }
}

// the phases of a sequence are complete: the exchanges of in-line halo blocks name no stream (= the one the phase ran on)
void finish_phases(st_plan* p) {
    for (st_plan::Phase& ph : p->phases)
        if (ph.halo && ph.ex.kind == 1 && halo_is_inline(p, ph.halo)) ph.ex.stream = nullptr;
}

int build_closure_phases(st_plan* p, const float* image, float* grad_out) {
    p->phases.clear();
    p->halo_inline.clear();
    if (ensure_streams(p) || ensure_comm_stream(p)) return 1;      // the descriptors carry the stream handles
    PhaseBuilder b{p};
    build_forward_phases(p, b, image, 29, /*fork_heads=*/true);
    // TV on the raw image strip (uses the image halo): WRITES grad_out; content MSE on relu4_2
    b.add([=](hipStream_t s) {
        StripInfo si{p->row0, p->Hg, p->has_up, p->has_down, p->img_halo};
        return launch_tv_strip(image, p->H, p->W, si, p->tv_weight, grad_out, p->red_partials, p->lossbuf + 1, s);
    });
    Node* ct = &p->conv[kContentConv];
    b.add([=](hipStream_t s) {
        const long long global_count = (long long)ct->c * ct->hg * ct->w;
        return launch_content_mse_strip(ct->y, p->content_target, (long long)ct->count(), global_count,
                                        p->content_weight, ct->g, p->red_partials + 4 * kStreamBlocks, p->lossbuf, s);
    });
    b.flush(allreduce_exchange(p->lossbuf, 5));
    b.add([=](hipStream_t s) {
        const long long global_count = (long long)ct->c * ct->hg * ct->w;
        if (launch_content_mse_final(p->lossbuf, global_count, p->content_weight, p->losses + 0, s)) return 1;
        return launch_tv_final(p->lossbuf + 1, p->Hg, p->W, p->tv_weight, p->losses + 6, s);
    });
    // The style heads were forked tap by tap during the forward phases.  With owned heads the broadcasts are issued
    // HERE, in the order the backward needs them (4, 3, 2, 1, 0): operations of one communicator execute in issue
    // order, so a broadcast issued at tap time would hold every later head's reduction behind the owner's chain.
    const bool owned = heads_owned(p);
    auto join_head = [&](int conv_index) {
        for (int k = 0; k < 5; ++k) {
            if (kStyleConv[k] != conv_index || !owned || p->style[k].joined_in_build) continue;
            p->style[k].joined_in_build = true;
            const long long cnt = (long long)p->style[k].n * p->style[k].n + p->style[k].n + 1;
            if (p->rank == head_owner(p, k))
                b.add([=](hipStream_t) {
                    ST_HIP(hipStreamWaitEvent(p->head_stream[k], p->chain_done[k], 0));
                    return 0;
                });
            b.flush(on_stream(rooted_exchange(5, p->head_result[k], cnt, head_owner(p, k)), p->head_stream[k], 1));
            b.add([=](hipStream_t) {
                hipStream_t hs = p->head_stream[k];
                if (style_head_result_unpack(p, k, hs)) return 1;
                if (style_head_gradient(p, k, hs)) return 1;
                ST_HIP(hipEventRecord(p->head_done[k], hs));
                return 0;
            });
        }
    };
    for (int k = 0; k < 5; ++k) p->style[k].joined_in_build = false;
    // backward trunk: before each data gradient the masked boundary rows of its operand are exchanged
    const st_net* net = p->net;
    for (int i = kNumOps - 1; i >= 0; --i) {
        const OpDesc op = kProgram[i];
        if (op.kind == 1) {
            Node* n = &p->pool[op.index];
            Node* in = &p->conv[kProgram[i - 1].index];
            b.add([=](hipStream_t s) {
                if (in->coded) return launch_pool_bwd_codes(in->pool_code, n->g, in->g, in->c, in->h, in->w, s);
                return launch_pool_bwd(in->y, n->g, in->g, in->c, in->h, in->w, net->pooling, s);
            });
            continue;
        }
        Node* n = &p->conv[op.index];
        join_head(op.index);
        b.add([=](hipStream_t s) {
            // this conv's output gradient is about to be read: its style head (if any) must be done
            if (join_head_for_conv(p, op.index, s)) return 1;
            // (a coded node's map was not written this pass; its gradient left the pooling backward already masked)
            if (pack_halo_rows(p, n->g, n->coded ? nullptr : n->y, n->c, n->h, n->w, s)) return 1;
            return comm_after_pack(p, s, n->ghalo);
        });
        b.flush(on_stream(halo_exchange(p, n->ghalo, n->c, n->w), p->comm_stream, 0), n->ghalo);
        if (op.index == 0) {
            p->halo_inline[n->ghalo] = inline_exchanges();       // (conv1_1's data gradient is one launch)
            b.add([=](hipStream_t s) {
                if (join_comm(p, s, n->ghalo)) return 1;
                return launch_conv_first_dgrad(n->g, nullptr, net->w_first, grad_out, p->dp_scratch, p->H, p->W, 1, s, n->ghalo,
                                               p->has_up, p->has_down, p->dp_parts);
            });
            continue;
        }
        const OpDesc pop = kProgram[i - 1];
        Node* in = (pop.kind == 0) ? &p->conv[pop.index] : &p->pool[pop.index];
        const int accumulate = (pop.kind == 0 && conv_is_tap(pop.index)) ? 1 : 0;
        if (pop.kind == 0) join_head(pop.index);
        if (pop.kind == 0)
            b.add([=](hipStream_t s) {
                // the launch ACCUMULATES into the input node's gradient: a style tap's head writes that buffer first
                return join_head_for_conv(p, pop.index, s);
            });
        ConvProblem c{};
        c.in = n->g; c.mask = (op.index == kStyleConv[4] && !head5_masks_its_gradient()) ? n->y : nullptr;      // see run_backward
        c.out_mask = (pop.kind == 0) ? in->y : nullptr;
        c.wgt = net->w_bwd[op.index]; c.out = in->g;
        c.cin = op.cout; c.cout = op.cin; c.height = n->h; c.width = n->w; c.taps = 9;
        c.accumulate = accumulate; c.scratch = p->conv_scratch;
        c.in_halo = n->ghalo; c.has_up = p->has_up; c.has_down = p->has_down;
        conv_arithmetic(net, op.index, true, c);
        c.amax_word = n->g_amax; c.out_amax = net->conv_elem == 1 ? in->g_amax : nullptr;
        add_strip_conv(p, b, c, [](ConvProblem&) {});
    }
    b.add([=](hipStream_t s) { return launch_sum_losses(p->losses, s); });      // every head has been joined
    b.flush(no_exchange());
    finish_phases(p);
    return 0;
}

// Eager on first sight of a pointer triple (warm-up: allocations, function attributes), captured on
// the second, replayed afterwards.  Anything that changes baked kernel arguments invalidates the graph.

// ---- activation-aware dynamic-range guard of the fp16x3 mode (round 4) ----------------------------------------------------
// The weights-only heuristic (range_guard) cannot see a DATA-dependent low-energy operand: channels that a particular
// image drives orders of magnitude below the tensor's maximum (dead-ish ReLU channels, which a trained VGG-19 has) sit at
// the bottom of fp16x3's one-scale-per-tensor window.  So the decision is measured on the image itself, once per scale,
// on the cold path: every unflagged trunk convolution is evaluated in bf16x6 (three bf16 planes, 8-bit exponents, no
// scale) on exactly the operand the shipped pass feeds it, and compared with what the shipped arithmetic produced -
// forward on the maps of a plain forward pass, data gradient on the gradients of one closure (same forward, same heads:
// the Newton-Schulz chains would amplify any forward difference, so the two modes never see different inputs).  A layer
// whose results differ by more than 1e-5 rel-L2 over the map, or on any output channel by more than 16 x what the exact-
// fp32 MFMA kernel differs there (range_mismatch), is flagged for that direction: it runs bf16x6 from then on (sticky, per network: merged with st_net_wide_layers; half the
// matrix rate on that layer).  After a flag the pass is repeated, because later layers then see different operands.
// Zero cost in the hot loop; the seeded synthetic weights flag nothing.
__global__ __launch_bounds__(256) void range_diff_kernel(const float* __restrict__ test, const float* __restrict__ exact,
                                                         const float* __restrict__ ref, long long per_channel,
                                                         float* __restrict__ sums) {
    // channel c = blockIdx.y, block b = blockIdx.x: sums[(c gridDim.x + b) 3 + {0, 1, 2}] = this block's sums of (test - ref)^2,
    // (exact - ref)^2, ref^2 - per-block partials that the host adds up in index order (float atomics from up to 64 blocks made
    // a score near 1.0 flip from run to run, and with it the arithmetic of the whole trajectory: ADVICE r4)
    __shared__ float scratch[4];
    const size_t base = (size_t)blockIdx.y * per_channel;
    float d2 = 0.f, x2 = 0.f, r2 = 0.f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < per_channel; i += (long long)gridDim.x * 256) {
        const float r = ref[base + i], d = test[base + i] - r, x = exact[base + i] - r;
        d2 += d * d;
        x2 += x * x;
        r2 += r * r;
    }
    d2 = block_sum_256(d2, scratch);
    x2 = block_sum_256(x2, scratch);
    r2 = block_sum_256(r2, scratch);
    if (threadIdx.x == 0) {
        float* mine = sums + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        mine[0] = d2; mine[1] = x2; mine[2] = r2;
    }
}
constexpr int kRangeBlocks = 64;                           // blocks per channel of range_diff_kernel (at most)
constexpr size_t kRangeSumFloats = (size_t)3 * 512 * kRangeBlocks;

// How far is `test` (the shipped fp16x3 result) from `ref` (bf16x6), measured (a) over the whole map against 1e-5 rel-L2
// and (b) per output channel against what the EXACT-fp32 MFMA kernel's own distance from bf16x6 is on that channel - the
// rounding noise any fp32-class arithmetic carries there, cancellation-dominated channels included.  A channel of tiny
// magnitude matters as much as any other when the next layer's weights for it are large, so (b) has no energy cut-off: a
// channel is off when its fp16x3 deviation exceeds 16 x the fp32 kernel's AND 2e-6 of its own norm.  score > 1: flag.
int range_mismatch(const float* test, const float* exact, const float* ref, int channels, long long per_channel,
                   float* dev_sums, hipStream_t s, double* score, double* map_rel) {
    ST_REQUIRE(channels <= 512, "range guard: more than 512 channels");
    const int bx = (int)std::min<long long>((per_channel + 255) / 256, kRangeBlocks);
    hipLaunchKernelGGL(range_diff_kernel, dim3(bx, channels), dim3(256), 0, s, test, exact, ref, per_channel, dev_sums);
    ST_LAUNCH_CHECK();
    std::vector<float> part((size_t)3 * channels * bx);
    ST_HIP(hipMemcpyAsync(part.data(), dev_sums, part.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    ST_HIP(hipStreamSynchronize(s));
    std::vector<double> h((size_t)3 * channels, 0.0);      // the blocks of a channel in index order: the same verdict every run
    for (int c = 0; c < channels; ++c)
        for (int b = 0; b < bx; ++b)
            for (int k = 0; k < 3; ++k) h[3 * c + k] += part[((size_t)c * bx + b) * 3 + k];
    double d2 = 0, r2 = 0;
    for (int c = 0; c < channels; ++c) { d2 += h[3 * c]; r2 += h[3 * c + 2]; }
    const double whole = r2 > 0 ? std::sqrt(d2 / r2) : 0.0;
    double worst = whole / 1e-5;
    for (int c = 0; c < channels; ++c) {
        const double dt = h[3 * c], dx = h[3 * c + 1], e = h[3 * c + 2];
        if (!(e > 0) || !(dt > 0)) continue;
        const double allowed = 256.0 * dx + 4e-12 * e;             // (16 x the fp32 kernel's deviation)^2 + (2e-6 |ref_c|)^2
        worst = std::max(worst, std::sqrt(dt / allowed));
    }
    *score = worst;
    *map_rel = whole;
    return 0;
}

int ensure_wide_planes(st_net* net, int conv, bool dgrad) {
    void*& slot = dgrad ? net->wsx_bwd[conv] : net->wsx_fwd[conv];
    if (slot) return 0;
    const OpDesc* op = nullptr;
    for (const OpDesc& o : kProgram)
        if (o.kind == 0 && o.index == conv) op = &o;
    ST_REQUIRE(op && net->w_torch[conv], "range guard: the network keeps no source weights for conv %d", conv);
    ST_HIP(hipMalloc(&slot, split_weight_bytes(op->cin, op->cout, 3)));
    if (launch_relayout_split(net->w_torch[conv], slot, op->cin, op->cout, dgrad ? 1 : 0, 3, 0, nullptr)) return 1;
    ST_HIP(hipDeviceSynchronize());
    return 0;
}

int plan_range_guard(st_plan* p, const float* image, hipStream_t s, int* new_fwd, int* new_bwd) {
    st_net* net = const_cast<st_net*>(p->net);                    // (flags and planes are added under the lock below)
    for (int i = 0; i < 13; ++i) new_fwd[i] = new_bwd[i] = 0;
    static Option on_opt("ST_CONV_RANGE_GUARD", 1);
    if (net->conv_elem != 1 || net->conv_planes != 2 || !on_opt.get() || p->strip) return 0;
    static std::mutex guard;
    std::lock_guard<std::mutex> lock(guard);
    if (ensure_grad_alloc(p) || ensure_streams(p, s)) return 1;
    const bool log = option_env("ST_RANGE_LOG") != nullptr;
    // three maps of the largest activation + the partial sums: allocated ONCE per plan (the guard runs several times per scale)
    // and owned by it - freed with the plan on every path, an error in the middle of this function included
    // (keyed on the LAST allocation: a call that failed half-way is repeated from the first missing buffer - advisor, round 5.
    // The three maps stay resident with the plan - 4.8 GB at 2896 x 2172, counted in st_plan_device_bytes.)
    if (!p->guard_sums) {
        size_t biggest = 0;
        for (const Node& n : p->conv) biggest = std::max(biggest, n.count());
        for (int k = 0; k < 3; ++k)
            if (!p->guard_scratch[k] && plan_alloc(p, &p->guard_scratch[k], biggest)) return 1;
        if (plan_alloc(p, &p->guard_sums, kRangeSumFloats)) return 1;
    }
    float *alt = p->guard_scratch[0], *cur = p->guard_scratch[1], *exact = p->guard_scratch[2], *sums = p->guard_sums;
    int rc = 0;
    auto finish = [&](int code) { return code; };

    // ---- forward: the maps of a plain forward pass (every map written: no argmax codes, no fused-pool-only layers) ----
    for (int pass = 0; pass < 13 && rc == 0; ++pass) {
        if (run_forward(p, image, 29, s)) return finish(1);
        bool flagged = false;
        const Node* prev = nullptr;
        for (int i = 0; i < kNumOps && !flagged; ++i) {
            const OpDesc& op = kProgram[i];
            const Node& n = op.kind == 0 ? p->conv[op.index] : p->pool[op.index];
            if (op.kind == 0 && op.index > 0 && !net->wide_fwd[op.index]) {
                if (ensure_wide_planes(net, op.index, false)) return finish(1);
                // (pre-activations: with the ReLU in place a nearly dead channel differs between any two arithmetics in WHICH
                // pixels survive, and the per-channel comparison would flag rounding noise)
                for (int mode = 0; mode < 3; ++mode) {
                    ConvProblem c{};
                    c.in = prev->y; c.wgt = net->w_fwd[op.index]; c.bias = net->bias[op.index];
                    c.out = mode == 1 ? cur : (mode == 2 ? exact : alt);
                    c.cin = op.cin; c.cout = op.cout; c.height = n.h; c.width = n.w; c.taps = 9; c.relu = 0;
                    c.scratch = p->conv_scratch;
                    if (mode == 1) { c.wgt_split = net->ws_fwd[op.index]; c.planes = 2; c.elem = 1; c.amax_word = prev->y_amax; }
                    else if (mode == 0) { c.wgt_split = net->wsx_fwd[op.index]; c.planes = 3; c.elem = 0; }
                    if (launch_conv(c, s)) return finish(1);
                }
                double score = 0, whole = 0;
                if (range_mismatch(cur, exact, alt, n.c, (long long)n.h * n.w, sums, s, &score, &whole)) return finish(1);
                if (log) fprintf(stderr, "[range] forward conv %2d: fp16x3 vs bf16x6 rel-L2 %.2e, score %.2f\n", op.index, whole, score);
                if (score > 1.0) {
                    net->wide_fwd[op.index] = net->guard_fwd[op.index] = new_fwd[op.index] = 1;
                    flagged = true;
                }
            }
            prev = &n;
        }
        if (!flagged) break;
    }
    if (!p->content_set) return finish(0);
    for (int i = 0; i < 5; ++i)
        if (!p->style[i].target_set) return finish(0);

    // ---- data gradients: the gradients of one closure; both arithmetics on the same operand ----
    int lowest_checked = 13;                                       // layers >= this index are settled
    for (int pass = 0; pass < 13; ++pass) {
        if (loss_and_grad(p, image, p->grad_img, nullptr, s)) return finish(1);
        ST_HIP(hipStreamSynchronize(s));
        bool flagged = false;
        for (int i = kNumOps - 1; i >= 0 && !flagged; --i) {
            const OpDesc& op = kProgram[i];
            if (op.kind != 0 || op.index == 0 || op.index >= lowest_checked) continue;
            const Node& n = p->conv[op.index];
            const OpDesc& pop = kProgram[i - 1];
            const Node& in = (pop.kind == 0) ? p->conv[pop.index] : p->pool[pop.index];
            if (net->wide_bwd[op.index]) { lowest_checked = op.index; continue; }
            if (ensure_wide_planes(net, op.index, true)) return finish(1);
            for (int mode = 0; mode < 3; ++mode) {
                ConvProblem c{};
                c.in = n.g; c.wgt = net->w_bwd[op.index]; c.out = mode == 1 ? cur : (mode == 2 ? exact : alt);
                c.cin = op.cout; c.cout = op.cin; c.height = n.h; c.width = n.w; c.taps = 9;
                c.scratch = p->conv_scratch;
                if (mode == 1) { c.wgt_split = net->ws_bwd[op.index]; c.planes = 2; c.elem = 1; c.amax_word = n.g_amax; }
                else if (mode == 0) { c.wgt_split = net->wsx_bwd[op.index]; c.planes = 3; c.elem = 0; }
                if (launch_conv(c, s)) return finish(1);
            }
            double score = 0, whole = 0;
            if (range_mismatch(cur, exact, alt, in.c, (long long)in.h * in.w, sums, s, &score, &whole)) return finish(1);
            if (log) fprintf(stderr, "[range] data gradient of conv %2d: fp16x3 vs bf16x6 rel-L2 %.2e, score %.2f\n", op.index, whole, score);
            if (score > 1.0) {
                net->wide_bwd[op.index] = net->guard_bwd[op.index] = new_bwd[op.index] = 1;
                flagged = true;                                    // shallower layers now see another gradient: again
            }
            lowest_checked = op.index;
        }
        if (!flagged) break;
    }
    return finish(0);
}

int closure_entry(st_plan* p, const float* image, float* grad_out, float* losses_out, hipStream_t s) {
    if (!p->graph_enabled || p->profiling) return loss_and_grad(p, image, grad_out, losses_out, s);
    const bool same = (p->gk_image == image && p->gk_grad == grad_out && p->gk_losses == losses_out);
    if (!same) {
        invalidate_graph(p);
        p->gk_image = image; p->gk_grad = grad_out; p->gk_losses = losses_out;
    }
    if (!p->graph_exec && p->gk_seen == 0) {
        p->gk_seen = 1;
        return loss_and_grad(p, image, grad_out, losses_out, s);
    }
    if (!p->main_stream) ST_HIP(hipStreamCreateWithFlags(&p->main_stream, hipStreamNonBlocking));
    ST_HIP(hipEventRecord(p->bridge_in, s));
    ST_HIP(hipStreamWaitEvent(p->main_stream, p->bridge_in, 0));
    if (!p->graph_exec) {
        ST_HIP(hipStreamBeginCapture(p->main_stream, hipStreamCaptureModeThreadLocal));
        p->capturing = true;
        const int rc = loss_and_grad(p, image, grad_out, losses_out, p->main_stream);
        p->capturing = false;
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(p->main_stream, &g);
        if (rc != 0 || e != hipSuccess || g == nullptr) {
            if (g) hipGraphDestroy(g);
            if (rc == 0) set_error("hipStreamEndCapture failed: %s", hipGetErrorString(e));
            p->graph_enabled = false;          // fall back to eager launches for this plan
            hipGetLastError();
            return rc != 0 ? rc : loss_and_grad(p, image, grad_out, losses_out, s);
        }
        p->graph = g;
        ST_HIP(hipGraphInstantiate(&p->graph_exec, p->graph, nullptr, nullptr, 0));
    }
    ST_HIP(hipGraphLaunch(p->graph_exec, p->main_stream));
    ST_HIP(hipEventRecord(p->bridge_out, p->main_stream));
    ST_HIP(hipStreamWaitEvent(s, p->bridge_out, 0));
    return 0;
}

}  // namespace

// =================================================================================================
extern "C" {

const char* st_last_error(void) { return st::get_error(); }
int st_abi_version(void) { return ST_AMD_ABI_VERSION; }
const char* st_compiled_arch(void) { return "gfx950"; }
int st_has_experiments(void) { return st::kExperiments ? 1 : 0; }
int st_env_switches(const char** names, int capacity) {
    const int n = (int)(sizeof(st::kEnvSwitches) / sizeof(st::kEnvSwitches[0]));
    for (int i = 0; i < n && i < capacity; ++i) names[i] = st::kEnvSwitches[i];
    return n;
}
int st_set_option(const char* name, int value, int clear) {
    ST_REQUIRE(name && name[0] == 'S' && name[1] == 'T' && name[2] == '_', "st_set_option: switch names start with ST_");
    st::option_set(name, value, clear != 0);
    return 0;
}

static int range_guard(st_net* net, int conv, const float* weight_dev, int cin, int cout);
static int net_fill(st_net* net, const float* const* weights, const float* const* biases);

int st_net_create(st_net** out, const float* const* weights, const float* const* biases, int pooling) {
    return st_net_create_ex(out, weights, biases, pooling, 0);
}

int st_net_create_ex(st_net** out, const float* const* weights, const float* const* biases, int pooling,
                     int conv_precision) {
    ST_REQUIRE(out && weights && biases, "st_net_create: null argument");
    ST_REQUIRE(pooling >= 0 && pooling <= 2, "st_net_create: unknown pooling %d", pooling);
    ST_REQUIRE(conv_precision_valid(conv_precision),
               "st_net_create: conv_precision must be 0 (fp32), 2 (bf16x3), 3 (bf16x6) or 4 (fp16x3)");
    st_net* net = new st_net();
    net->pooling = pooling;
    net->conv_planes = conv_precision_planes(conv_precision);
    net->conv_elem = conv_precision_elem(conv_precision);
    if (net_fill(net, weights, biases)) {      // error text already set; release what was allocated so far
        st_net_destroy(net);
        return 1;
    }
    *out = net;
    return 0;
}

// Dynamic-range guard of the fp16x3 mode.  Two fp16 planes under ONE power-of-two scale per tensor keep 22 bits only
// for elements within ~28 binades of the tensor's maximum.  A network may carry feature-map channels that are orders of
// magnitude smaller than their neighbours and are multiplied by correspondingly LARGE weights (any per-channel rescaling
// of a ReLU network is function-preserving, and trained VGG-19s are not normalised): those products matter as much as the
// others but their operands sit at the bottom of the window (tests/test_hot_path_gpu.py,
// test_closure_with_six_decades_of_channel_scales: content term off by 6e-3).  Only the weights are known here, and
// such compensation shows in them: an input channel (forward) / output channel (data gradient) whose largest weight lies
// far above the layer's MEDIAN channel.  Layers flagged that way run bf16x6 (conv_split_kernel, three bf16 planes, half
// the matrix rate of fp16x3, no scale).  ST_CONV_RANGE_GUARD=0 disables the guard, ST_CONV_RANGE_LOG2 (default 8) is the
// max / median ratio, as a power of two, beyond which a layer is flagged.
static int range_guard(st_net* net, int conv, const float* weight_dev, int cin, int cout) {
    static Option guard_opt("ST_CONV_RANGE_GUARD", 1);
    static Option log2_opt("ST_CONV_RANGE_LOG2", 8);
    if (!guard_opt.get()) return 0;
    const size_t wcount = (size_t)cout * cin * 9;
    std::vector<float> w(wcount);
    ST_HIP(hipMemcpy(w.data(), weight_dev, wcount * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<float> by_in(cin, 0.f), by_out(cout, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < 9; ++t) {
                const float a = std::fabs(w[((size_t)co * cin + ci) * 9 + t]);
                by_in[ci] = std::max(by_in[ci], a);
                by_out[co] = std::max(by_out[co], a);
            }
    auto spread = [](std::vector<float> v) {
        std::sort(v.begin(), v.end());
        const float med = v[v.size() / 2];
        return med > 0.f ? v.back() / med : 0.f;
    };
    const float limit = std::ldexp(1.f, log2_opt.get());
    net->wide_fwd[conv] = spread(by_in) > limit;
    net->wide_bwd[conv] = spread(by_out) > limit;
    const size_t bytes = split_weight_bytes(cin, cout, 3);
    if (net->wide_fwd[conv]) {
        ST_HIP(hipMalloc(&net->wsx_fwd[conv], bytes));
        if (launch_relayout_split(weight_dev, net->wsx_fwd[conv], cin, cout, 0, 3, 0, nullptr)) return 1;
    }
    if (net->wide_bwd[conv]) {
        ST_HIP(hipMalloc(&net->wsx_bwd[conv], bytes));
        if (launch_relayout_split(weight_dev, net->wsx_bwd[conv], cin, cout, 1, 3, 0, nullptr)) return 1;
    }
    return 0;
}

static int net_fill(st_net* net, const float* const* weights, const float* const* biases) {
    int conv = 0;
    for (int i = 0; i < kNumOps; ++i) {
        const OpDesc& op = kProgram[i];
        if (op.kind != 0) continue;
        const size_t wcount = (size_t)op.cout * op.cin * 9;
        ST_HIP(hipMalloc(&net->bias[conv], op.cout * sizeof(float)));
        ST_HIP(hipMemcpy(net->bias[conv], biases[conv], op.cout * sizeof(float), hipMemcpyDeviceToDevice));
        if (conv == 0) {
            ST_HIP(hipMalloc(&net->w_first, wcount * sizeof(float)));
            ST_HIP(hipMemcpy(net->w_first, weights[0], wcount * sizeof(float), hipMemcpyDeviceToDevice));
            std::vector<float> hw(wcount), hb(op.cout);
            ST_HIP(hipMemcpy(hw.data(), weights[0], wcount * sizeof(float), hipMemcpyDeviceToHost));
            ST_HIP(hipMemcpy(hb.data(), biases[0], op.cout * sizeof(float), hipMemcpyDeviceToHost));
            for (int co = 0; co < op.cout; ++co) {
                float l1 = 0.f;
                for (int k = 0; k < 27; ++k) l1 += std::fabs(hw[(size_t)co * 27 + k]);
                net->w_first_l1max = std::max(net->w_first_l1max, l1);
                net->b_first_max = std::max(net->b_first_max, std::fabs(hb[co]));
            }
        } else {
            ST_HIP(hipMalloc(&net->w_fwd[conv], wcount * sizeof(float)));
            ST_HIP(hipMalloc(&net->w_bwd[conv], wcount * sizeof(float)));
            if (launch_relayout_fwd(weights[conv], net->w_fwd[conv], op.cin, op.cout, nullptr)) return 1;
            if (launch_relayout_dgrad(weights[conv], net->w_bwd[conv], op.cin, op.cout, nullptr)) return 1;
            if (net->conv_planes > 0) {
                const size_t bytes = split_weight_bytes(op.cin, op.cout, net->conv_planes);
                ST_HIP(hipMalloc(&net->ws_fwd[conv], bytes));
                ST_HIP(hipMalloc(&net->ws_bwd[conv], bytes));
                if (launch_relayout_split(weights[conv], net->ws_fwd[conv], op.cin, op.cout, 0, net->conv_planes,
                                          net->conv_elem, nullptr) ||
                    launch_relayout_split(weights[conv], net->ws_bwd[conv], op.cin, op.cout, 1, net->conv_planes,
                                          net->conv_elem, nullptr))
                    return 1;
                if (net->conv_elem == 1) {
                    ST_HIP(hipMalloc(&net->w_torch[conv], wcount * sizeof(float)));
                    ST_HIP(hipMemcpy(net->w_torch[conv], weights[conv], wcount * sizeof(float), hipMemcpyDeviceToDevice));
                    if (range_guard(net, conv, weights[conv], op.cin, op.cout)) return 1;
                }
            }
        }
        ++conv;
    }
    ST_HIP(hipDeviceSynchronize());
    return 0;
}

int st_net_wide_layers(const st_net* net, int* forward13, int* backward13) {
    ST_REQUIRE(net && forward13 && backward13, "st_net_wide_layers: null argument");
    for (int i = 0; i < 13; ++i) {
        forward13[i] = net->wide_fwd[i];
        backward13[i] = net->wide_bwd[i];
    }
    return 0;
}

int st_net_mark_wide(st_net* net, const int* forward13, const int* backward13) {
    ST_REQUIRE(net && forward13 && backward13, "st_net_mark_wide: null argument");
    if (net->conv_elem != 1 || net->conv_planes != 2) return 0;          // only fp16x3 networks have a second arithmetic
    for (int i = 1; i < 13; ++i) {
        if (forward13[i] && !net->wide_fwd[i]) {
            if (ensure_wide_planes(net, i, false)) return 1;
            net->wide_fwd[i] = net->guard_fwd[i] = 1;
        }
        if (backward13[i] && !net->wide_bwd[i]) {
            if (ensure_wide_planes(net, i, true)) return 1;
            net->wide_bwd[i] = net->guard_bwd[i] = 1;
        }
    }
    return 0;
}

int st_net_destroy(st_net* net) {
    if (!net) return 0;
    hipFree(net->w_first);
    for (int i = 0; i < 13; ++i) {
        hipFree(net->bias[i]);
        hipFree(net->w_fwd[i]);
        hipFree(net->w_bwd[i]);
        hipFree(net->ws_fwd[i]);
        hipFree(net->ws_bwd[i]);
        hipFree(net->wsx_fwd[i]);
        hipFree(net->wsx_bwd[i]);
        hipFree(net->w_torch[i]);
    }
    delete net;
    return 0;
}

static int plan_create_common(st_plan** out, const st_net* net, int local_height, int width, int global_height,
                              int row0, bool strip_mode) {
    st_plan* p = new st_plan();
    p->net = net;
    p->H = local_height;
    p->W = width;
    p->Hg = global_height;
    p->dp_parts = conv_first_dgrad_parts(global_height, width);
    p->row0 = row0;
    p->strip = strip_mode;
    p->has_up = p->strip && row0 > 0;
    p->has_down = p->strip && row0 + local_height < global_height;
    int h = local_height, w = width, hg = global_height;
    for (int i = 0; i < kNumOps; ++i) {
        const OpDesc& op = kProgram[i];
        Node& n = (op.kind == 0) ? p->conv[op.index] : p->pool[op.index];
        if (op.kind == 1) { h /= 2; w /= 2; hg /= 2; }
        n.c = op.cout; n.h = h; n.w = w; n.hg = hg;
        if (plan_alloc(p, &n.y, n.count())) { st_plan_destroy(p); return 1; }
        if (p->strip) {
            const bool feeds_conv = (i + 1 < kNumOps) && kProgram[i + 1].kind == 0;
            if (feeds_conv && halo_alloc(p, &n.yhalo, (size_t)2 * n.c * n.w)) { st_plan_destroy(p); return 1; }
            if (op.kind == 0 && halo_alloc(p, &n.ghalo, (size_t)2 * n.c * n.w)) { st_plan_destroy(p); return 1; }
        }
    }
    // convs whose output only the following max pool consumes (relu1_2, 2_2, 3_4, 4_4): in the closure their epilogue
    // leaves the pooled map + one code byte per window instead of the full-resolution map
    for (int i = 0; i + 1 < kNumOps; ++i) {
        if (kProgram[i].kind != 0 || kProgram[i + 1].kind != 1 || net->pooling != 0) continue;
        Node& n = p->conv[kProgram[i].index];
        if (n.h % 2 != 0 || n.w % 4 != 0) continue;
        float* mem = nullptr;
        if (plan_alloc(p, &mem, ((size_t)n.c * (n.h / 2) * (n.w / 2) + 3) / 4)) { st_plan_destroy(p); return 1; }
        n.pool_code = reinterpret_cast<unsigned char*>(mem);
    }
    for (int i = 0; i < 5; ++i) {
        const Node& tap = p->conv[kStyleConv[i]];
        p->style[i].n = tap.c;
        p->style[i].npix = (long long)tap.hg * tap.w;
        p->style[i].npix_local = (long long)tap.h * tap.w;
    }
    float* ticket_mem = nullptr;
    if (plan_alloc(p, &ticket_mem, 256) || hipMemset(ticket_mem, 0, 256 * sizeof(float)) != hipSuccess) {
        st_plan_destroy(p);
        return 1;
    }
    p->tickets = reinterpret_cast<unsigned int*>(ticket_mem);
    if (plan_alloc(p, &p->losses, 64) || plan_alloc(p, &p->red_partials, 5 * kStreamBlocks) ||
        plan_alloc(p, &p->conv_scratch, kConvScratchFloats) ||
        plan_alloc(p, &p->dp_scratch, (size_t)3 * (local_height + 2) * (width + 2) * p->dp_parts) || plan_alloc(p, &p->amax_word, (size_t)64 * kAmaxWordUints) ||
        plan_alloc(p, &p->content_target, p->conv[kContentConv].count())) {
        st_plan_destroy(p);
        return 1;
    }
    {
        unsigned int* words = reinterpret_cast<unsigned int*>(p->amax_word);
        for (int i = 0; i < kNumOps; ++i) {
            const OpDesc& op = kProgram[i];
            if (op.kind == 0) {
                p->conv[op.index].y_amax = words + (size_t)op.index * kAmaxWordUints;
                p->conv[op.index].g_amax = words + (size_t)(16 + op.index) * kAmaxWordUints;
            } else {
                Node& src = p->conv[kProgram[i - 1].index];          // a pool always follows a conv
                p->pool[op.index].y_amax = src.y_amax;
                p->pool[op.index].g_amax = words + (size_t)(32 + op.index) * kAmaxWordUints;
                src.g_amax = p->pool[op.index].g_amax;
            }
        }
    }
    for (int i = 0; i < 5; ++i)
        p->style[i].s_amax = reinterpret_cast<unsigned int*>(p->amax_word) + (size_t)(48 + i) * kAmaxWordUints;
    if (p->strip) {
        float* hb = nullptr;
        if (plan_alloc(p, &hb, kAmaxWordUints)) { st_plan_destroy(p); return 1; }
        p->halo_bound = reinterpret_cast<unsigned int*>(hb);
        float* ps = nullptr;
        if (halo_alloc(p, &p->img_halo, (size_t)6 * width) || plan_alloc(p, &p->send_up, (size_t)64 * width + kHaloTrailer) ||
            plan_alloc(p, &p->send_down, (size_t)64 * width + kHaloTrailer) || plan_alloc(p, &p->lossbuf, 64) ||
            plan_alloc(p, &ps, kPackScratchUints)) {
            st_plan_destroy(p);
            return 1;
        }
        p->pack_scratch = reinterpret_cast<unsigned int*>(ps);
        if (hipMemset(ps, 0, kPackScratchUints * sizeof(unsigned int)) != hipSuccess ||
            hipMemset(p->send_up, 0, ((size_t)64 * width + kHaloTrailer) * sizeof(float)) != hipSuccess ||
            hipMemset(p->send_down, 0, ((size_t)64 * width + kHaloTrailer) * sizeof(float)) != hipSuccess) {
            set_error("hipMemset of the halo send buffers failed");
            st_plan_destroy(p);
            return 1;
        }
        // raw moment sums of the five heads in ONE block: a single all-reduce per closure
        size_t total = 0;
        for (int i = 0; i < 5; ++i) total += (size_t)p->style[i].n * p->style[i].n + p->style[i].n;
        float* block = nullptr;
        if (plan_alloc(p, &block, total)) { st_plan_destroy(p); return 1; }
        p->gram_total = (long long)total;
        for (int i = 0; i < 5; ++i) {
            p->gram_raw[i] = block;
            block += (size_t)p->style[i].n * p->style[i].n + p->style[i].n;
            if (plan_alloc(p, &p->head_result[i], (size_t)p->style[i].n * p->style[i].n + p->style[i].n + 64)) {
                st_plan_destroy(p);
                return 1;
            }
        }
    }
    *out = p;
    return 0;
}

int st_plan_create(st_plan** out, const st_net* net, int height, int width) {
    ST_REQUIRE(out && net, "st_plan_create: null argument");
    // VGGFeatures.forward size check for taps up to features[29] (style_transfer.py:61-69,81-83)
    ST_REQUIRE(height >= 16 && width >= 16, "Input is %dx%d but must be at least 16x16", height, width);
    ST_REQUIRE((long long)height * width <= (1ll << 25), "image too large (H*W must be <= 2^25)");
    return plan_create_common(out, net, height, width, height, 0, false);
}

int st_plan_create_strip(st_plan** out, const st_net* net, int global_height, int width, int row_begin,
                         int row_end) {
    ST_REQUIRE(out && net, "st_plan_create_strip: null argument");
    ST_REQUIRE(global_height >= 16 && width >= 16, "Input is %dx%d but must be at least 16x16", global_height,
               width);
    ST_REQUIRE(row_begin >= 0 && row_end > row_begin && row_end <= global_height, "strip rows out of range");
    ST_REQUIRE(row_begin % 16 == 0 && (row_end % 16 == 0 || row_end == global_height),
               "strip boundaries must be multiples of 16 rows (all four 2x2 poolings stay strip-local)");
    ST_REQUIRE(row_end - row_begin >= 16, "a strip needs at least 16 rows");
    ST_REQUIRE((long long)(row_end - row_begin) * width <= (1ll << 25), "strip too large");
    return plan_create_common(out, net, row_end - row_begin, width, global_height, row_begin, true);
}

int st_plan_destroy(st_plan* p) {
    if (!p) return 0;
    for (void* a : p->allocations) hipFree(a);
    for (ProfileEvent& e : p->events) {
        hipEventDestroy(e.start);
        hipEventDestroy(e.stop);
    }
    for (HbmEvent& e : p->hbm_events) {
        hipEventDestroy(e.start);
        hipEventDestroy(e.stop);
    }
    invalidate_graph(p);
    if (p->streams_ready) {
        if (p->main_stream) { hipStreamSynchronize(p->main_stream); hipStreamDestroy(p->main_stream); }
        for (hipStream_t j : p->junk_streams) hipStreamDestroy(j);
        hipEventDestroy(p->bridge_in);
        hipEventDestroy(p->bridge_out);
        if (p->aux_stream) { hipStreamSynchronize(p->aux_stream); hipStreamDestroy(p->aux_stream); }
        for (int i = 0; i < 5; ++i)
            if (p->head_stream[i]) {
                hipStreamSynchronize(p->head_stream[i]);
                if (p->head_stream_owned[i]) hipStreamDestroy(p->head_stream[i]);
            }
        for (hipEvent_t e : {p->aux_in, p->aux_fwd, p->tv_done, p->content_done})
            if (e) hipEventDestroy(e);
        for (int i = 0; i < 5; ++i) {
            hipEventDestroy(p->moments_ready[i]);
            hipEventDestroy(p->chain_done[i]);
            hipEventDestroy(p->tap_ready[i]);
            hipEventDestroy(p->head_done[i]);
        }
    }
    if (p->comm_stream) {
        hipStreamSynchronize(p->comm_stream);
        if (!p->comm_stream_borrowed) hipStreamDestroy(p->comm_stream);
        hipEventDestroy(p->pack_done);
        hipEventDestroy(p->halo_landed);
    }
    delete p;
    return 0;
}

long long st_plan_device_bytes(const st_plan* p) { return p ? p->bytes : 0; }

int st_plan_forward(st_plan* p, const float* image, int last_layer, void* stream) {
    ST_REQUIRE(p && image, "st_plan_forward: null argument");
    ST_REQUIRE(last_layer >= 1 && last_layer <= 29, "st_plan_forward: last_layer %d out of range", last_layer);
    return run_forward(p, image, last_layer, static_cast<hipStream_t>(stream));
}

int st_plan_feature(const st_plan* p, int layer, const float** data, int* channels, int* height, int* width) {
    ST_REQUIRE(p && data, "st_plan_feature: null argument");
    const Node* n = feature_node(p, layer);
    ST_REQUIRE(n != nullptr, "st_plan_feature: features[%d] is not a ReLU or pooling output", layer);
    *data = n->y;
    if (channels) *channels = n->c;
    if (height) *height = n->h;
    if (width) *width = n->w;
    return 0;
}

int st_plan_moments(st_plan* p, int layer, float* mean_out, float* srm_out, void* stream) {
    ST_REQUIRE(p && mean_out && srm_out, "st_plan_moments: null argument");
    int idx = -1;
    for (int i = 0; i < 5; ++i)
        if (kStyleFeat[i] == layer) idx = i;
    ST_REQUIRE(idx >= 0, "st_plan_moments: features[%d] is not a style layer", layer);
    if (ensure_style_alloc(p, idx)) return 1;
    return moments_of_tap(p, idx, mean_out, srm_out, static_cast<hipStream_t>(stream));
}

int st_plan_set_content_target(st_plan* p, const float* feat, void* stream) {
    ST_REQUIRE(p && feat, "st_plan_set_content_target: null argument");
    ST_HIP(hipMemcpyAsync(p->content_target, feat, p->conv[kContentConv].count() * sizeof(float),
                          hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    p->content_set = true;     // (buffer contents only: a captured graph stays valid)
    return 0;
}

int st_plan_set_style_target(st_plan* p, int index, const float* mean, const float* srm, void* stream) {
    ST_REQUIRE(p && mean && srm, "st_plan_set_style_target: null argument");
    ST_REQUIRE(index >= 0 && index < 5, "st_plan_set_style_target: index %d out of range", index);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ensure_style_alloc(p, index)) return 1;
    StyleHead& h = p->style[index];
    ST_HIP(hipMemcpyAsync(h.mean_t, mean, h.n * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (launch_cov_from_moments(mean, srm, h.cov_t, h.n, kCovEps, s)) return 1;
    if (ns_sqrt_forward(h.cov_t, h.root_t, h.n, h.ns, s)) return 1;
    h.target_set = true;
    return 0;
}

int st_plan_set_loss_weights(st_plan* p, float content_weight, const float* style_layer_weights,
                             float tv_weight) {
    ST_REQUIRE(p && style_layer_weights, "st_plan_set_loss_weights: null argument");
    p->content_weight = content_weight;
    for (int i = 0; i < 5; ++i) p->style_weight[i] = style_layer_weights[i];
    p->tv_weight = tv_weight;
    invalidate_graph(p);       // the weights are baked into kernel arguments
    p->phases.clear();         // (phase lambdas read the weights at run time, but keep it simple)
    return 0;
}

int st_plan_loss_and_grad(st_plan* p, const float* image, float* grad_out, float* losses_out, void* stream) {
    ST_REQUIRE(p && image && grad_out, "st_plan_loss_and_grad: null argument");
    if (ensure_grad_alloc(p) || ensure_streams(p, static_cast<hipStream_t>(stream))) return 1;
    return closure_entry(p, image, grad_out, losses_out, static_cast<hipStream_t>(stream));
}

int st_plan_step(st_plan* p, float* image, float* exp_avg, float* exp_avg_sq, float* ema_value,
                 long long step, double lr, double beta1, double beta2, double eps, double ema_decay,
                 float* losses_out, void* stream) {
    ST_REQUIRE(p && image && exp_avg && exp_avg_sq && ema_value, "st_plan_step: null argument");
    ST_REQUIRE(step >= 1, "st_plan_step: step must be >= 1");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ensure_grad_alloc(p) || ensure_streams(p, s)) return 1;
    // Round 6: the sum of the loss terms and the clearing of the operand bounds for the next pass were two dependent launches
    // of ~5 us each on the caller's stream, every iteration; they ride in the update kernel (ST_STEP_TAIL=0: launches of their
    // own, as in the closure-only entry points; not under graph replay, whose captured closure ends with the sum)
    // ST_STEP_TAIL=2 (default) goes one further: the update itself is applied by conv1_1's fold kernel - the last kernel of the
    // backward pass, which has each gradient element in a register when it is final - so the iteration ends with ONE launch
    // instead of four (fold, sum, update, memset).  Not while profiling (bench.py's roofline_hbm times the update kernel).
    static Option tail_opt("ST_STEP_TAIL", 2);
    const bool fold_tail = tail_opt.get() != 0 && !p->graph_enabled;
    const bool fold_step = fold_tail && tail_opt.get() >= 2 && !p->profiling;
    AdamTail tail{};
    if (fold_tail) {
        tail.losses8 = p->losses;
        tail.losses_copy = losses_out;
        if (p->net->conv_elem == 1) { tail.zero = reinterpret_cast<unsigned int*>(p->amax_word); tail.zero_count = 64ll * kAmaxWordUints; }
    }
    // host-side scalars exactly as torch computes them (Python doubles; torch/optim/adam.py:476-547)
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    AdamScalars sc{};
    sc.lerp_w = (float)(1.0 - beta1);
    sc.beta2 = (float)beta2;
    sc.one_m_beta2 = (float)(1.0 - beta2);
    sc.step_size = (float)(lr / bc1);
    sc.bc2_sqrt = (float)std::sqrt(bc2);
    sc.eps = (float)eps;
    sc.decay = (float)ema_decay;             // torch.tensor(decay): fp32 buffer (style_transfer.py:243)
    sc.one_m_decay = 1.0f - sc.decay;        // (1 - self.decay) evaluated in fp32 (:253)
    FoldUpdate upd{image, exp_avg, exp_avg_sq, ema_value, sc, tail};
    if (upd.tail.losses_copy == upd.tail.losses8) upd.tail.losses_copy = nullptr;
    p->defer_sum = fold_tail;
    p->fold_update = fold_step ? &upd : nullptr;
    p->fold_updated = false;
    const int closure_rc = closure_entry(p, image, p->grad_img, losses_out, s);
    p->defer_sum = false;
    p->fold_update = nullptr;
    if (closure_rc) return 1;
    if (p->fold_updated) {
        p->fold_updated = false;
        p->amax_clean = tail.zero != nullptr;
        return 0;
    }
    // reads image, gradient, both moments, EMA; writes image, both moments, EMA
    const int rc = hbm_profiled(p, HBM_ADAM, 9.0 * 3 * 4.0 * p->H * p->W, s, [&] {
        return launch_adam_clamp_ema(image, p->grad_img, exp_avg, exp_avg_sq, ema_value, 3ll * p->H * p->W, sc, s, tail);
    });
    p->amax_clean = rc == 0 && tail.zero != nullptr;
    return rc;
}

int st_plan_apply_update(st_plan* p, float* image, const float* grad, float* exp_avg, float* exp_avg_sq,
                         float* ema_value, long long step, double lr, double beta1, double beta2, double eps,
                         double ema_decay, void* stream) {
    ST_REQUIRE(p && image && grad && exp_avg && exp_avg_sq && ema_value, "st_plan_apply_update: null argument");
    ST_REQUIRE(step >= 1, "st_plan_apply_update: step must be >= 1");
    const double bc1 = 1.0 - std::pow(beta1, (double)step);
    const double bc2 = 1.0 - std::pow(beta2, (double)step);
    AdamScalars sc{};
    sc.lerp_w = (float)(1.0 - beta1);
    sc.beta2 = (float)beta2;
    sc.one_m_beta2 = (float)(1.0 - beta2);
    sc.step_size = (float)(lr / bc1);
    sc.bc2_sqrt = (float)std::sqrt(bc2);
    sc.eps = (float)eps;
    sc.decay = (float)ema_decay;
    sc.one_m_decay = 1.0f - sc.decay;
    return launch_adam_clamp_ema(image, grad, exp_avg, exp_avg_sq, ema_value, 3ll * p->H * p->W, sc,
                                 static_cast<hipStream_t>(stream));
}

int st_plan_closure_begin(st_plan* p, const float* image, float* grad_out) {
    ST_REQUIRE(p && image && grad_out, "st_plan_closure_begin: null argument");
    ST_REQUIRE(p->strip, "st_plan_closure_begin: not a strip plan (use st_plan_create_strip)");
    ST_REQUIRE(p->content_set, "content target not set (st_plan_set_content_target)");
    for (int i = 0; i < 5; ++i)
        ST_REQUIRE(p->style[i].target_set, "style target %d not set (st_plan_set_style_target)", i);
    if (ensure_grad_alloc(p)) return 1;
    for (int i = 0; i < 5; ++i)
        if (ensure_style_alloc(p, i)) return 1;
    // (the tile / overlap / ownership decisions of the phase sequence depend on the library's switches)
    if (p->ph_image != image || p->ph_grad != grad_out || p->ph_last_layer != -1 || p->phases.empty() ||
        p->ph_option_gen != option_generation()) {
        if (build_closure_phases(p, image, grad_out)) return 1;
        p->ph_image = image; p->ph_grad = grad_out; p->ph_last_layer = -1;
        p->ph_option_gen = option_generation();
    }
    p->phase_pos = 0;
    return 0;
}

int st_plan_forward_begin(st_plan* p, const float* image, int last_layer) {
    ST_REQUIRE(p && image, "st_plan_forward_begin: null argument");
    ST_REQUIRE(p->strip, "st_plan_forward_begin: not a strip plan");
    ST_REQUIRE(last_layer >= 1 && last_layer <= 29, "st_plan_forward_begin: last_layer %d out of range", last_layer);
    p->phases.clear();
    p->halo_inline.clear();
    if (ensure_comm_stream(p)) return 1;       // the halo descriptors carry its handle
    PhaseBuilder b{p};
    build_forward_phases(p, b, image, last_layer);
    b.flush(no_exchange());
    finish_phases(p);
    p->ph_image = image; p->ph_grad = nullptr; p->ph_last_layer = last_layer;
    p->phase_pos = 0;
    return 0;
}

int st_plan_set_rank(st_plan* p, int rank, int world) {
    ST_REQUIRE(p && p->strip, "st_plan_set_rank: not a strip plan");
    ST_REQUIRE(world >= 1 && rank >= 0 && rank < world, "st_plan_set_rank: rank %d of %d", rank, world);
    p->rank = rank;
    p->world = world;
    p->phases.clear();          // head ownership is baked into the phase sequence
    return 0;
}

int st_plan_closure_next(st_plan* p, st_exchange* ex, void* stream) {
    ST_REQUIRE(p && ex, "st_plan_closure_next: null argument");
    if (p->phase_pos >= p->phases.size()) {
        std::memset(ex, 0, sizeof(*ex));
        return 0;
    }
    st_plan::Phase& ph = p->phases[p->phase_pos++];
    if (ph.run(static_cast<hipStream_t>(stream))) return 1;
    *ex = ph.ex;
    return 0;
}

int st_plan_closure_run(st_plan* p, st_fabric* fabric, void* stream) {
    ST_REQUIRE(p && fabric, "st_plan_closure_run: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (p->phase_pos == 0) {
        // Operations of ONE communicator must not run concurrently: the heads' collectives (channel 1) are ordered by
        // sharing a stream (the compact layout: every head's per-rank work on one stream), the trunk's (channel 0) by the
        // events between the caller's and the communication stream.  ST_STREAMS_COMPACT=0 gives every head a stream of its
        // own - fine for torch.distributed, whose process group serialises on its internal stream, not for this transport.
        void* head_stream = nullptr;
        for (const st_plan::Phase& ph : p->phases) {
            if (ph.ex.channel != 1 || ph.ex.kind == 0 || ph.ex.kind == 3) continue;
            ST_REQUIRE(!head_stream || !ph.ex.stream || ph.ex.stream == head_stream,
                       "st_plan_closure_run: the heads' exchanges name different streams (ST_STREAMS_COMPACT=0?): the in-library "
                       "transport needs the compact stream layout, use the descriptor form (ST_FABRIC_NATIVE=0) otherwise");
            if (ph.ex.stream) head_stream = ph.ex.stream;
        }
    }
    while (p->phase_pos < p->phases.size()) {
        st_plan::Phase& ph = p->phases[p->phase_pos++];
        if (ph.run(s)) return 1;
        if (st::fabric_apply(fabric, ph.ex, s)) return 1;
    }
    return 0;
}

int st_plan_range_guard(st_plan* p, const float* image, int* forward13, int* backward13, void* stream) {
    ST_REQUIRE(p && image && forward13 && backward13, "st_plan_range_guard: null argument");
    return plan_range_guard(p, image, static_cast<hipStream_t>(stream), forward13, backward13);
}

int st_plan_debug_read(st_plan* p, int what, float* out, int count) {
    ST_REQUIRE(p && out && count > 0, "st_plan_debug_read: bad argument");
    // what = 0: the partial sums of the TV kernels' workgroups, 4 floats each (tv_interior_kernel's first, then tv_border_kernel's)
    // what = 1: every thread's (s1, s2, s3, s4, groups visited) of tv_interior_kernel under ST_TV_VARIANT=3
    ST_REQUIRE((what == 0 && count <= 4 * kStreamBlocks) || (what == 1 && count <= kStreamBlocks * 256 * 5 && tv_debug_buffer()),
               "st_plan_debug_read: unknown buffer or count out of range");
    ST_HIP(hipDeviceSynchronize());
    ST_HIP(hipMemcpy(out, what == 0 ? p->red_partials : tv_debug_buffer(), (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}

int st_plan_losses(st_plan* p, float** losses) {
    ST_REQUIRE(p && losses, "st_plan_losses: null argument");
    *losses = p->losses;
    return 0;
}

int st_plan_moment_sums(st_plan* p, int layer, float* sums, void* stream) {
    ST_REQUIRE(p && sums, "st_plan_moment_sums: null argument");
    int idx = -1;
    for (int i = 0; i < 5; ++i)
        if (kStyleFeat[i] == layer) idx = i;
    ST_REQUIRE(idx >= 0, "st_plan_moment_sums: features[%d] is not a style layer", layer);
    if (ensure_style_alloc(p, idx)) return 1;
    return moment_sums_of_tap(p, idx, sums, static_cast<hipStream_t>(stream));
}

int st_plan_set_graph(st_plan* p, int enable) {
    ST_REQUIRE(p, "st_plan_set_graph: null plan");
    p->graph_enabled = enable != 0;
    if (!enable) invalidate_graph(p);
    return 0;
}

int st_plan_profile_enable(st_plan* p, int enable) {
    ST_REQUIRE(p, "st_plan_profile_enable: null plan");
    p->profiling = enable != 0;
    return 0;
}

int st_plan_profile_read(st_plan* p, long long* launches, double* millis, double* flops) {
    ST_REQUIRE(p, "st_plan_profile_read: null plan");
    for (size_t i = 0; i < p->events_used; ++i) {
        ProfileEvent& e = p->events[i];
        ST_HIP(hipEventSynchronize(e.stop));
        float ms = 0.f;
        ST_HIP(hipEventElapsedTime(&ms, e.start, e.stop));
        p->prof_ms += ms;
        p->prof_flops += e.flops;
        p->prof_launches += 1;
    }
    p->events_used = 0;
    p->hbm_used = 0;
    if (launches) *launches = p->prof_launches;
    if (millis) *millis = p->prof_ms;
    if (flops) *flops = p->prof_flops;
    p->prof_launches = 0;
    p->prof_ms = 0;
    p->prof_flops = 0;
    return 0;
}

int st_plan_profile_read_hbm(st_plan* p, int category, long long* launches, double* millis, double* bytes) {
    ST_REQUIRE(p && category >= 0 && category < HBM_CATS, "st_plan_profile_read_hbm: bad argument");
    // call for every category of interest BEFORE st_plan_profile_read / the next profiled step: events are kept until
    // category -1 ... (they are recycled by st_plan_profile_read)
    long long n = 0;
    double ms_sum = 0, b = 0;
    for (size_t i = 0; i < p->hbm_used; ++i) {
        HbmEvent& e = p->hbm_events[i];
        if (e.cat != category) continue;
        ST_HIP(hipEventSynchronize(e.stop));
        float ms = 0.f;
        ST_HIP(hipEventElapsedTime(&ms, e.start, e.stop));
        ms_sum += ms;
        b += e.bytes;
        ++n;
    }
    if (launches) *launches = n;
    if (millis) *millis = ms_sum;
    if (bytes) *bytes = b;
    return 0;
}

// ---- standalone operators for kernel-level tests -------------------------------------------------
int st_op_sqrtm_ns(const float* a, float* root, int n, void* stream) {
    ST_REQUIRE(a && root, "st_op_sqrtm_ns: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* base = nullptr;
    ST_HIP(hipMalloc(&base, ns_workspace_floats(n) * sizeof(float)));
    NSWorkspace ws{};
    ns_workspace_carve(ws, base, n);
    int rc = ns_workspace_reset(ws, s) || ns_sqrt_forward(a, root, n, ws, s);
    hipStreamSynchronize(s);
    if (!rc) rc = ns_chain_check(ws, "st_op_sqrtm_ns");
    hipFree(base);
    return rc;
}

int st_op_sqrtm_ns_backward(const float* root, const float* grad_root, float* grad_a, int n, void* stream) {
    ST_REQUIRE(root && grad_root && grad_a, "st_op_sqrtm_ns_backward: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* base = nullptr;
    ST_HIP(hipMalloc(&base, ns_workspace_floats(n) * sizeof(float)));
    NSWorkspace ws{};
    ns_workspace_carve(ws, base, n);
    const int rc = ns_workspace_reset(ws, s) || ns_sqrt_backward(root, grad_root, nullptr, grad_a, n, ws, s);
    hipStreamSynchronize(s);
    hipFree(base);
    return rc;
}

int st_op_sqrtm_ns_backward_diag(const float* root, float grad_diag, float* grad_a, int n, void* stream) {
    ST_REQUIRE(root && grad_a, "st_op_sqrtm_ns_backward_diag: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float *base = nullptr, *gd = nullptr;
    ST_HIP(hipMalloc(&base, ns_workspace_floats(n) * sizeof(float)));
    int rc = 1;
    if (hipMalloc(&gd, 256) != hipSuccess) {
        set_error("st_op_sqrtm_ns_backward_diag: hipMalloc failed");
    } else if (hipMemcpyAsync(gd, &grad_diag, sizeof(float), hipMemcpyHostToDevice, s) != hipSuccess) {
        set_error("st_op_sqrtm_ns_backward_diag: upload of the gradient scalar failed");
    } else {
        NSWorkspace ws{};
        ns_workspace_carve(ws, base, n);
        rc = ns_workspace_reset(ws, s) || ns_sqrt_backward(root, nullptr, gd, grad_a, n, ws, s);
        hipStreamSynchronize(s);
        if (!rc) rc = ns_chain_check(ws, "st_op_sqrtm_ns_backward_diag");
    }
    hipStreamSynchronize(s);
    hipFree(base);
    hipFree(gd);
    return rc;
}

int st_op_sqrtm_time(int n, int iters, double* fwd_us, double* bwd_us, void* stream) {
    ST_REQUIRE(fwd_us && bwd_us && iters > 0, "st_op_sqrtm_time: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t nn = (size_t)n * n;
    float *base = nullptr, *a = nullptr, *root = nullptr, *g = nullptr, *ga = nullptr;
    ST_HIP(hipMalloc(&base, ns_workspace_floats(n) * 4));
    ST_HIP(hipMalloc(&a, nn * 4)); ST_HIP(hipMalloc(&root, nn * 4));
    ST_HIP(hipMalloc(&g, nn * 4)); ST_HIP(hipMalloc(&ga, nn * 4));
    std::vector<float> h(nn, 0.f);
    unsigned x = 777u;
    for (size_t i = 0; i < nn; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 9) % 2001 - 1000) * 1e-4f; }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) h[(size_t)i * n + j] = h[(size_t)j * n + i];     // symmetric
    for (int i = 0; i < n; ++i) h[(size_t)i * n + i] = 1.0f + 0.1f * (i % 7);          // diagonally dominant
    ST_HIP(hipMemcpy(a, h.data(), nn * 4, hipMemcpyHostToDevice));
    ST_HIP(hipMemcpy(g, h.data(), nn * 4, hipMemcpyHostToDevice));
    NSWorkspace ws{};
    ns_workspace_carve(ws, base, n);
    if (ns_workspace_reset(ws, s)) return 1;
    hipEvent_t e0, e1, e2;
    ST_HIP(hipEventCreate(&e0)); ST_HIP(hipEventCreate(&e1)); ST_HIP(hipEventCreate(&e2));
    // ST_NS_TIME_DIAG=1: time the backward the plan runs (gradient = multiple of I) instead of the general one
    static Option diag_opt("ST_NS_TIME_DIAG", 0);
    const bool diag = diag_opt.get() != 0;
    float* gd = nullptr;
    ST_HIP(hipMalloc(&gd, 256));
    const float gdv = -2.f / n;
    ST_HIP(hipMemcpy(gd, &gdv, sizeof(float), hipMemcpyHostToDevice));
    const float* gfull = diag ? nullptr : g;
    const float* gdiag = diag ? gd : nullptr;
    if (ns_sqrt_forward(a, root, n, ws, s) || ns_sqrt_backward(root, gfull, gdiag, ga, n, ws, s)) return 1;
    ST_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        if (ns_sqrt_forward(a, root, n, ws, s)) return 1;
    ST_HIP(hipEventRecord(e1, s));
    for (int i = 0; i < iters; ++i)
        if (ns_sqrt_backward(root, gfull, gdiag, ga, n, ws, s)) return 1;
    ST_HIP(hipEventRecord(e2, s));
    ST_HIP(hipEventSynchronize(e2));
    float f = 0.f, b = 0.f;
    ST_HIP(hipEventElapsedTime(&f, e0, e1));
    ST_HIP(hipEventElapsedTime(&b, e1, e2));
    *fwd_us = f * 1e3 / iters;
    *bwd_us = b * 1e3 / iters;
    if (ns_chain_check(ws, "st_op_sqrtm_time")) return 1;
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
    hipFree(base); hipFree(a); hipFree(root); hipFree(g); hipFree(ga); hipFree(gd);
    return 0;
}

int st_op_tv_loss(const float* image, int height, int width, float* loss_out, float* grad_out, void* stream) {
    ST_REQUIRE(image && loss_out && grad_out, "st_op_tv_loss: null argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* partials = nullptr;
    ST_HIP(hipMalloc(&partials, 4 * kStreamBlocks * sizeof(float)));
    const int rc = launch_tv(image, height, width, 1.0f, grad_out, partials, loss_out, s);
    hipStreamSynchronize(s);
    hipFree(partials);
    return rc;
}

static int conv_op(const float* in, const float* mask, const float* weight, const float* bias, float* out,
                   int cin, int cout, int height, int width, int relu, int dgrad, int precision, hipStream_t s,
                   const float* halo = nullptr, int has_up = 0, int has_down = 0, int accumulate = 0,
                   const float* out_mask = nullptr, int overlap = 0) {
    // precision 5: fp16x3 in the Winograd F(2x2, 3x3) form wherever that kernel takes the problem (st_conv_wino.hip), else the
    // direct fp16x3 kernels
    const bool wino5 = precision == 5;
    if (wino5) precision = 4;
    ST_REQUIRE(conv_precision_valid(precision), "conv precision must be 0, 2, 3 or 4");
    float* wl = nullptr;
    float* scratch = nullptr;
    void* wsplit = nullptr;
    unsigned int* amax = nullptr;
    ST_HIP(hipMalloc(&wl, (size_t)cin * cout * 9 * sizeof(float)));
    ST_HIP(hipMalloc(&scratch, kConvScratchFloats * sizeof(float)));
    ST_HIP(hipMalloc(&amax, kAmaxWordUints * 4));
    ST_HIP(hipMemsetAsync(amax, 0, kAmaxWordUints * 4, s));
    ConvProblem c{};
    c.scratch = scratch;
    if (precision > 0) {
        c.planes = conv_precision_planes(precision);
        c.elem = conv_precision_elem(precision);
        c.amax_word = amax;
        c.amax_measure = 1;
        ST_HIP(hipMalloc(&wsplit, split_weight_bytes(cin, cout, c.planes)));
        if (launch_relayout_split(weight, wsplit, cin, cout, dgrad, c.planes, c.elem, s)) return 1;
        c.wgt_split = wsplit;
    }
    void* wino = nullptr;
    if (wino5) {
        ST_HIP(hipMalloc(&wino, winograd_weight_bytes(cin, cout)));
        if (launch_winograd_weights(weight, wino, cin, cout, dgrad, s)) return 1;
        c.wgt_wino = wino;
        c.wino = 2;
    }
    if (!dgrad) {
        if (launch_relayout_fwd(weight, wl, cin, cout, s)) return 1;
        c.cin = cin; c.cout = cout;
    } else {
        if (launch_relayout_dgrad(weight, wl, cin, cout, s)) return 1;
        c.cin = cout; c.cout = cin;
    }
    c.in = in; c.mask = mask; c.wgt = wl; c.bias = bias; c.out = out; c.height = height; c.width = width;
    c.taps = 9; c.relu = relu; c.accumulate = accumulate; c.out_mask = out_mask;
    c.in_halo = halo; c.has_up = halo ? has_up : 0; c.has_down = halo ? has_down : 0;
    int rc = 0;
    if (overlap) {          // interior rows first (no halo), then the boundary rows: the strip plans' two-launch form
        PcOverlap o{};
        if (!conv_pc_overlap_choice(c, &o)) {
            set_error("st_op_conv3x3_strip_ex: this problem cannot be cut into interior + boundary launches");
            rc = 1;
        } else {
            ConvProblem part = c;
            part.overlap_part = 1; part.in_halo = nullptr; part.has_up = 0; part.has_down = 0;
            rc = launch_conv(part, s);
            part = c;
            part.overlap_part = 2; part.amax_measure = 0;
            if (!rc) rc = launch_conv(part, s);
        }
    } else {
        rc = launch_conv(c, s);
    }
    hipStreamSynchronize(s);
    hipFree(wl);
    hipFree(scratch);
    hipFree(wsplit);
    hipFree(amax);
    hipFree(wino);
    return rc;
}

int st_op_conv1x1(const float* in, const float* weight, const float* bias, float* out, int cin, int cout,
                  long long npix, int precision, void* stream) {
    ST_REQUIRE(in && weight && out, "st_op_conv1x1: null argument");
    ST_REQUIRE(precision == 0 || precision == 4, "st_op_conv1x1: precision must be 0 (fp32) or 4 (fp16x3)");
    ST_REQUIRE(cin % 32 == 0 && cout % 64 == 0 && npix > 0 && npix < (1ll << 31), "st_op_conv1x1: bad shape");
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned int* amax = nullptr;
    float* scratch = nullptr;
    ST_HIP(hipMalloc(&amax, 2 * kAmaxWordUints * 4));
    ST_HIP(hipMalloc(&scratch, kConvScratchFloats * sizeof(float)));
    ST_HIP(hipMemsetAsync(amax, 0, 2 * kAmaxWordUints * 4, s));
    ConvProblem c{};
    c.in = in; c.wgt = weight; c.bias = bias; c.out = out; c.cin = cin; c.cout = cout;
    c.height = 1; c.width = (int)npix; c.taps = 1; c.scratch = scratch;
    int rc = 0;
    if (precision == 4) {
        c.planes = 2; c.elem = 1; c.amax_word = amax; c.wgt_amax = amax + kAmaxWordUints;
        rc = launch_amax(in, (long long)cin * npix, amax, 0, s) ||
             launch_amax(weight, (long long)cin * cout, amax + kAmaxWordUints, 0, s);
    }
    if (!rc) rc = launch_conv(c, s);
    hipStreamSynchronize(s);
    hipFree(amax);
    hipFree(scratch);
    return rc;
}

int st_op_conv3x3_time(int cin, int cout, int height, int width, int dgrad, int precision, int iters,
                       double* avg_us, void* stream) {
    ST_REQUIRE(avg_us && iters > 0, "st_op_conv3x3_time: bad argument");
    ST_REQUIRE(conv_precision_valid(precision) || precision == 5, "conv precision must be 0, 2, 3, 4 or 5 (fp16x3, Winograd form)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t hw = (size_t)height * width;
    float *in = nullptr, *mask = nullptr, *w = nullptr, *wl = nullptr, *bias = nullptr, *out = nullptr,
          *scratch = nullptr;
    const int kin = dgrad ? cout : cin, kout = dgrad ? cin : cout;
    ST_HIP(hipMalloc(&in, kin * hw * 4));
    ST_HIP(hipMalloc(&mask, kin * hw * 4));
    ST_HIP(hipMalloc(&out, kout * hw * 4));
    ST_HIP(hipMalloc(&w, (size_t)cin * cout * 9 * 4));
    ST_HIP(hipMalloc(&wl, (size_t)cin * cout * 9 * 4));
    ST_HIP(hipMalloc(&bias, kout * 4));
    ST_HIP(hipMalloc(&scratch, kConvScratchFloats * 4));
    // deterministic non-trivial contents (values matter for DVFS: do not time zero-filled operands)
    std::vector<float> host(std::max<size_t>((size_t)cin * cout * 9, kin * hw));
    unsigned x = 12345u;
    for (float& v : host) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 9) % 2001 - 1000) * 1e-3f; }
    ST_HIP(hipMemcpy(in, host.data(), kin * hw * 4, hipMemcpyHostToDevice));
    ST_HIP(hipMemcpy(mask, host.data(), kin * hw * 4, hipMemcpyHostToDevice));
    ST_HIP(hipMemcpy(w, host.data(), (size_t)cin * cout * 9 * 4, hipMemcpyHostToDevice));
    ST_HIP(hipMemcpy(bias, host.data(), kout * 4, hipMemcpyHostToDevice));
    ConvProblem c{};
    if (dgrad) { if (launch_relayout_dgrad(w, wl, cin, cout, s)) return 1; }
    else { if (launch_relayout_fwd(w, wl, cin, cout, s)) return 1; }
    // (ST_CONV_NOMASK=1: time the data gradient as the plan runs it - masked by its producer, no mask stream)
    static Option nomask_opt("ST_CONV_NOMASK", 0);
    c.in = in; c.mask = (dgrad && !nomask_opt.get()) ? mask : nullptr; c.wgt = wl; c.bias = dgrad ? nullptr : bias; c.out = out;
    c.cin = kin; c.cout = kout; c.height = height; c.width = width; c.taps = 9; c.relu = dgrad ? 0 : 1;
    c.scratch = scratch;
    void* wsplit = nullptr;
    unsigned int* amax = nullptr;
    ST_HIP(hipMalloc(&amax, 2 * kAmaxWordUints * 4));
    void* wino = nullptr;
    if (precision == 5) {          // fp16x3, Winograd form wherever it takes the problem
        precision = 4;
        ST_HIP(hipMalloc(&wino, winograd_weight_bytes(cin, cout)));
        if (launch_winograd_weights(w, wino, cin, cout, dgrad, s)) return 1;
        c.wgt_wino = wino;
        c.wino = 2;
        c.mask = nullptr;          // (as the plan runs its data gradients: masked by their producers)
    }
    if (precision > 0) {
        c.planes = conv_precision_planes(precision);
        c.elem = conv_precision_elem(precision);
        c.amax_word = amax;
        ST_HIP(hipMalloc(&wsplit, split_weight_bytes(cin, cout, c.planes)));
        if (launch_relayout_split(w, wsplit, cin, cout, dgrad, c.planes, c.elem, s)) return 1;
        c.wgt_split = wsplit;
    }
    // fp16x3: the operand bound is measured once here; inside a plan it comes for free from the producer's
    // epilogue, so the timed launches (like the plan's) only read the word and fold max |out| into another
    ST_HIP(hipMemsetAsync(amax, 0, 2 * kAmaxWordUints * 4, s));
    c.amax_measure = 1;
    c.out_amax = c.elem == 1 ? amax + kAmaxWordUints : nullptr;
    if (launch_conv(c, s)) return 1;
    c.amax_measure = 0;
    auto launch = [&]() -> int { return launch_conv(c, s); };
    for (int i = 0; i < 3; ++i)
        if (launch()) return 1;
    hipEvent_t e0, e1;
    ST_HIP(hipEventCreate(&e0));
    ST_HIP(hipEventCreate(&e1));
    ST_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i)
        if (launch()) return 1;
    ST_HIP(hipEventRecord(e1, s));
    ST_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    ST_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = ms * 1e3 / iters;
    if (option_env("ST_CONV_PHASES")) {          // s_memtime phase stamps of the producer / consumer kernel (tune bit 32)
        const char* tune_env = option_env("ST_CONV_TUNE");          // ablation bits of the timed launches stay on
        const int keep = tune_env ? atoi(tune_env) : 0;
        c.tune = keep | 32;
        ST_HIP(hipMemsetAsync(scratch, 0, 1 << 20, s));
        for (int i = 0; i < 4; ++i)                     // a few launches back to back: the clock has settled
            if (launch_conv(c, s)) return 1;
        ST_HIP(hipStreamSynchronize(s));
        c.tune = 0;
        {
            std::vector<unsigned long long> st(8 * 4096);
            ST_HIP(hipMemcpy(st.data(), scratch, st.size() * 8, hipMemcpyDeviceToHost));
            double ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int n = 0;
            for (int b = 0; b < 4096; ++b) {
                if (st[8 * b + 6] != 1) continue;
                for (int k = 0; k < 8; ++k) ph[k] += (double)st[8 * b + k];
                ++n;
            }
            if (n)
                fprintf(stderr, "[phases] %d->%d @%d dgrad %d tune %d: %d WGs, ticks avg per WG: consumer MFMA %.0f | consumer barrier "
                        "wait %.0f | epilogue %.0f | producer staging %.0f | producer barrier wait %.0f | whole %.0f; shader "
                        "clock %.0f MHz; %.1f us\n",
                        cin, cout, height, dgrad, keep, n, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n, ph[4] / n, ph[5] / n,
                        ph[7] > 0 ? ph[5] / ph[7] * 100.0 : 0.0, *avg_us);
        }
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    hipFree(in); hipFree(mask); hipFree(out); hipFree(w); hipFree(wl); hipFree(bias); hipFree(scratch); hipFree(wsplit); hipFree(amax);
    hipFree(wino);
    return 0;
}

int st_op_conv3x3(const float* in, const float* weight, const float* bias, float* out, int cin, int cout,
                  int height, int width, int relu, int precision, void* stream) {
    ST_REQUIRE(in && weight && out, "st_op_conv3x3: null argument");
    return conv_op(in, nullptr, weight, bias, out, cin, cout, height, width, relu, 0, precision,
                   static_cast<hipStream_t>(stream));
}

int st_op_conv3x3_dgrad(const float* grad_out, const float* relu_out, const float* weight, float* grad_in,
                        int cin, int cout, int height, int width, int precision, void* stream) {
    ST_REQUIRE(grad_out && weight && grad_in, "st_op_conv3x3_dgrad: null argument");
    return conv_op(grad_out, relu_out, weight, nullptr, grad_in, cin, cout, height, width, 0, 1, precision,
                   static_cast<hipStream_t>(stream));
}

int st_op_conv3x3_strip(const float* in, const float* halo, int has_up, int has_down, const float* weight,
                        const float* bias, float* out, int cin, int cout, int height, int width, int relu, int dgrad,
                        int precision, void* stream) {
    ST_REQUIRE(in && halo && weight && out, "st_op_conv3x3_strip: null argument");
    return conv_op(in, nullptr, weight, dgrad ? nullptr : bias, out, cin, cout, height, width, dgrad ? 0 : relu, dgrad,
                   precision, static_cast<hipStream_t>(stream), halo, has_up != 0, has_down != 0);
}

int st_op_conv3x3_strip_ex(const float* in, const float* halo, int has_up, int has_down, const float* weight,
                           const float* bias, float* out, const float* out_mask, int cin, int cout, int height, int width,
                           int relu, int dgrad, int accumulate, int overlap, int precision, void* stream) {
    ST_REQUIRE(in && weight && out, "st_op_conv3x3_strip_ex: null argument");       // (halo == NULL: a whole image)
    return conv_op(in, nullptr, weight, dgrad ? nullptr : bias, out, cin, cout, height, width, dgrad ? 0 : relu, dgrad,
                   precision, static_cast<hipStream_t>(stream), halo, has_up != 0, has_down != 0, accumulate != 0, out_mask,
                   overlap);
}

}  // extern "C"
