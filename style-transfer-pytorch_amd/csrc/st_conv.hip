// Implicit-GEMM 3x3 / 1x1 convolution on the gfx950 fp32 matrix pipe.
//
// Replaces, on the hot path of the reference:
//   * nn.Conv2d 3x3/s1/p1 + bias + in-place ReLU for vgg19.features[2..28]   (style_transfer.py:35,87)
//   * their autograd data gradients (convolution_backward, input only - weights are frozen, :48-49)
//     with the ReLU threshold_backward folded into the operand staging
//   * the einsum backward of StyleLossW2.get_target (:167): dF = Ssym . F + b, a 1x1 "convolution"
//
// GEMM view:  D[co][pixel] = sum over k=(tap, ci) of  Wt[k][co] * X[k][pixel(+tap offset)]
//   M = output channels  -> MFMA "A" operand, lanes 0..31 = 32 consecutive co          (LDS [k][co])
//   N = pixels           -> MFMA "B" operand, lanes 0..31 = 32 pixels of the tile       (LDS [ci][row][col])
//   K = 2 per v_mfma_f32_32x32x2_f32: lanes 32..63 carry the odd input channel of a pair.
// With N on the lanes the accumulator layout (col = lane & 31) makes every global store a 128-byte
// row segment of one output channel.  The fp32 MFMA issues once per 64 cycles per SIMD, so one
// ds_read_b32 per operand per MFMA is far below LDS bandwidth; the kernel is MFMA-issue bound by
// construction and the staging (plain dword loads, register double buffer) only has to stay ahead.
#include <cstdlib>
#include <type_traits>

#include "st_common.h"

namespace st {

namespace {

// input channels staged per LDS buffer: 8 for the 3x3 kernel (4 MFMA k-steps per tap, 36 per chunk); the 1x1 kernel
// (Gram backward) has a single tap, so it stages 32 channels per chunk - with 8 a chunk was 4 k-steps between two
// barriers and the kernel ran at ~27 TF
constexpr int kChunk3x3 = 8, kChunk1x1 = 32;

template <int TAPS, int TW, int WN, int WGM>
struct Cfg {
    static constexpr int KC = (TAPS == 9) ? kChunk3x3 : kChunk1x1;
    static constexpr int WGN = 4 / WGM;            // waves along the pixel dimension
    static constexpr int TCO = 64 * WGM;           // output channels per workgroup
    static constexpr int NPIX = 32 * WN * WGN;     // pixels per workgroup
    static constexpr int TH = NPIX / TW;
    static constexpr int HALO = (TAPS == 9) ? 1 : 0;
    static constexpr int LH = TH + 2 * HALO;
    static constexpr int LW = TW + 2 * HALO;
    static constexpr int PLANE = LH * LW;
    static constexpr int NE_IN = KC * PLANE;
    static constexpr int IN_FLOATS = (NE_IN + 3) & ~3;
    static constexpr int NI = (NE_IN + 255) / 256;
    static constexpr int NE_W4 = TAPS * KC * TCO / 4;
    static constexpr int NW = (NE_W4 + 255) / 256;
    static constexpr int BUF_FLOATS = IN_FLOATS + TAPS * KC * TCO;
    static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;
    static_assert(NPIX % TW == 0, "tile width must divide the pixel count");
};

// compile-time loop: the body receives std::integral_constant<int, I>, so every array index derived
// from it is a constant by construction (runtime-indexed register arrays are demoted to scratch)
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

__device__ __forceinline__ float buffer_load_f32(__amdgpu_buffer_rsrc_t rsrc, int byte_offset) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_offset, 0, 0));
}

constexpr int kOutOfRange = 0x40000000;   // byte offset beyond any chunk: the buffer load returns 0

template <int TAPS, int TW, int WN, int WGM, bool MASKED, bool HALO>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvProblem p, int tiles_x, int n_co_tiles,
                                                        int ksplit) {
    using C = Cfg<TAPS, TW, WN, WGM>;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave / C::WGN, wn = wave % C::WGN;

    // XCD-aware logical order (see st_conv_split.hip): the Cout tiles / K slices of one pixel tile run on ONE XCD
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int co_tile = bid % n_co_tiles;
    bid /= n_co_tiles;
    const int kslice = bid % ksplit;
    bid /= ksplit;
    const int tile_x = bid % tiles_x, tile_y = bid / tiles_x;
    const int x0 = tile_x * TW, y0 = tile_y * C::TH, co0 = co_tile * C::TCO;
    const int H = p.height, W = p.width;
    const int HW = H * W;

    // ---- per-thread staging maps (fixed for the whole K loop) ----
    int goff[C::NI];
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
        const int e = tid + i * 256;
        const int c = e / C::PLANE, rem = e % C::PLANE;
        const int y = y0 - C::HALO + rem / C::LW, x = x0 - C::HALO + rem % C::LW;
        const bool ok = (e < C::NE_IN) && y >= 0 && y < H && x >= 0 && x < W;
        goff[i] = ok ? (c * HW + y * W + x) * 4 : kOutOfRange;   // byte offset inside the chunk
    }
    // strip sharding: elements of tile rows -1 / H come from the halo block instead (second resource)
    constexpr bool use_halo = HALO;
    int hoff[HALO ? C::NI : 1];
#pragma unroll
    for (int i = 0; i < C::NI; ++i) {
        const int e = tid + i * 256;
        const int c = e / C::PLANE, rem = e % C::PLANE;
        const int y = y0 - C::HALO + rem / C::LW, x = x0 - C::HALO + rem % C::LW;
        const bool xin = (e < C::NE_IN) && x >= 0 && x < W;
        const bool top = use_halo && xin && y == -1 && p.has_up;
        const bool bot = use_halo && xin && y == H && p.has_down;
        hoff[i] = top ? (c * W + x) * 4 : (bot ? ((p.cin + c) * W + x) * 4 : kOutOfRange);

    }
    int woff[C::NW];
#pragma unroll
    for (int i = 0; i < C::NW; ++i) {
        const int f = tid + i * 256;
        const int row = f / (C::TCO / 4), c4 = f % (C::TCO / 4);
        const int tap = row / C::KC, kc = row % C::KC;
        woff[i] = (tap * p.cin + kc) * p.cout + co0 + c4 * 4;
    }

    // Staged operands travel global -> VGPR -> LDS.  Raw buffer loads: the zero padding (and the
    // ragged right/bottom tile edge) is the hardware's out-of-range behaviour, so the loads carry no
    // branches or selects and all stay in flight across the MFMA block of the current chunk.
    float rin[C::NI];
    float rhalo[HALO ? C::NI : 1];
    float rmask[MASKED ? C::NI : 1];
    f32x4 rw[C::NW];
    const int chunk_bytes = C::KC * HW * 4;

    // One staging item of a chunk: items [0, NI) are input-tile dwords (+ mask / halo companions),
    // items [NI, NI + NW) 16-byte weight pieces.  The item index is a compile-time constant.
    constexpr int NITEMS = C::NI + C::NW;
    auto load_item = [&](int ci0, auto IT) __attribute__((always_inline)) {
        constexpr int item = decltype(IT)::value;
        if constexpr (item < C::NI) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.in) + (size_t)ci0 * HW, 0, chunk_bytes, 0x00020000);
            rin[item] = buffer_load_f32(rs, goff[item]);
            if constexpr (MASKED) {
                const __amdgpu_buffer_rsrc_t ms = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.mask) + (size_t)ci0 * HW, 0, chunk_bytes, 0x00020000);
                rmask[item] = buffer_load_f32(ms, goff[item]);
            }
            if constexpr (HALO) {
                const __amdgpu_buffer_rsrc_t hs = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.in_halo) + (size_t)ci0 * W, 0, (p.cin + C::KC) * W * 4, 0x00020000);
                rhalo[item] = buffer_load_f32(hs, hoff[item]);
            }
        } else {
            constexpr int i = item - C::NI;
            const float* wb = p.wgt + (size_t)ci0 * p.cout;
            if (tid + i * 256 < C::NE_W4) rw[i] = *reinterpret_cast<const f32x4*>(wb + woff[i]);
        }
    };
    auto load_chunk = [&](int ci0) __attribute__((always_inline)) { static_for<0, NITEMS>([&](auto IT) __attribute__((always_inline)) { load_item(ci0, IT); }); };
    auto store_item = [&](float* buf, auto IT) __attribute__((always_inline)) {
        constexpr int item = decltype(IT)::value;
        if constexpr (item < C::NI) {
            const int e = tid + item * 256;
            float v = rin[item];
            if constexpr (MASKED) v = (rmask[item] > 0.f) ? v : 0.f;     // threshold_backward
            if constexpr (HALO) v += rhalo[item];   // halo rows (already masked by their owner); 0 elsewhere
            if (e < C::NE_IN) buf[e] = v;
        } else {
            constexpr int i = item - C::NI;
            const int f = tid + i * 256;
            if (f < C::NE_W4) *reinterpret_cast<f32x4*>(buf + C::IN_FLOATS + f * 4) = rw[i];
        }
    };
    auto store_chunk = [&](float* buf) __attribute__((always_inline)) { static_for<0, NITEMS>([&](auto IT) __attribute__((always_inline)) { store_item(buf, IT); }); };

    // ---- MFMA operand addresses ----
    const int a_base = C::IN_FLOATS + half * C::TCO + wm * 64 + l31;
    int b_base[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int pix = (wn * WN + j) * 32 + l31;
        b_base[j] = half * C::PLANE + (pix / TW) * C::LW + (pix % TW);
    }

    f32x16 acc[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A chunk is NSTEP k-steps (tap, channel pair); each k-step is 2*WN MFMAs = 128..256 pipe cycles.
    // Every k-step also carries (a) the LDS operand reads of the k-step PD ahead and (b) its share of
    // the LDS writes that stage the NEXT chunk into the other buffer, so neither the ~100-cycle LDS
    // latency nor the write phase ever leaves the matrix pipe idle (measured before: 62 % MFMA-busy
    // with one wave per SIMD when reads/writes were issued as separate bursts).  sched_barrier(0)
    // pins this interleave; left alone the machine scheduler sinks each read back to its use.
    constexpr int NSTEP = TAPS * C::KC / 2;
    constexpr int PD = (NSTEP >= 8) ? 4 : 2;           // prefetch distance in k-steps
    constexpr int RING = PD + 1;
    auto fetch_step = [&](const float* buf, int st, float (&av)[2], float (&bv)[WN]) {
        const int tap = st / (C::KC / 2), kk = st % (C::KC / 2);
        const int ky = (TAPS == 9) ? tap / 3 : 0, kx = (TAPS == 9) ? tap % 3 : 0;
        av[0] = buf[a_base + (tap * C::KC + 2 * kk) * C::TCO];
        av[1] = buf[a_base + (tap * C::KC + 2 * kk) * C::TCO + 32];
#pragma unroll
        for (int j = 0; j < WN; ++j) bv[j] = buf[b_base[j] + 2 * kk * C::PLANE + ky * C::LW + kx];
    };
    // The LDS writes of the next chunk ride on the last WSPAN k-steps (its global loads are issued as one
    // burst at the top: spreading them over k-steps, or skewing co-resident workgroups by half a chunk,
    // measured no gain).  tune bits 4 / 8 are ABLATIONS for tools/conv_bench.py only (wrong results):
    // 4 = operands fetched from LDS once per chunk, 8 = no global loads / LDS writes inside the loop.
    constexpr int WSPAN = (NSTEP >= 12) ? NSTEP / 2 : NSTEP;
    const bool abl_no_fetch = (p.tune & 4) != 0, abl_no_stage = (p.tune & 8) != 0;
    auto compute = [&](const float* buf, float* next_buf, bool more, int next_ci0) __attribute__((always_inline)) {
        float av[RING][2], bv[RING][WN];
        if (more && !abl_no_stage) load_chunk(next_ci0);
        static_for<0, PD>([&](auto ST) __attribute__((always_inline)) {
            constexpr int st = decltype(ST)::value;
            fetch_step(buf, st, av[st % RING], bv[st % RING]);
        });
        static_for<0, NSTEP>([&](auto ST) __attribute__((always_inline)) {
            constexpr int st = decltype(ST)::value;
            if constexpr (st + PD < NSTEP) {
                if (!abl_no_fetch) fetch_step(buf, st + PD, av[(st + PD) % RING], bv[(st + PD) % RING]);
            }
            if (more && !abl_no_stage) {
                static_for<0, NITEMS>([&](auto IT) __attribute__((always_inline)) {
                    constexpr int item = decltype(IT)::value;
                    if constexpr ((NSTEP - WSPAN) + item * WSPAN / NITEMS == st) store_item(next_buf, IT);
                });
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st % RING][0], bv[st % RING][j], acc[0][j], 0, 0, 0);
                acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st % RING][1], bv[st % RING][j], acc[1][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // ---- K loop: register-staged double buffer, one barrier per chunk ----
    const int nchunks = p.cin / C::KC / ksplit;              // chunks of this K slice
    const int chunk0 = kslice * nchunks;
    load_chunk(chunk0 * C::KC);
    store_chunk(smem);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        float* cur = smem + (c & 1) * C::BUF_FLOATS;
        float* nxt = smem + ((c + 1) & 1) * C::BUF_FLOATS;
        const bool more = (c + 1 < nchunks);
        compute(cur, nxt, more, (chunk0 + c + 1) * C::KC);
        __syncthreads();
    }

    // ---- epilogue: bias, ReLU, optional accumulate; 128-byte row segments per store ----
    // Branch-free: the bias slice goes through LDS (free after the last barrier), out-of-image
    // pixels get an out-of-range buffer offset (loads return 0, stores are dropped by the hardware),
    // and the accumulate reads of a 32x32 tile are issued together before the first use.
    // Split-K slices store raw partial sums into their scratch slab; bias/ReLU/accumulate then happen
    // in conv_splitk_reduce_kernel.
    const bool partial = ksplit > 1;
    float* out_base = partial ? p.scratch + (size_t)kslice * p.cout * HW : p.out;
    float* bias_lds = smem;
    if (tid < C::TCO) bias_lds[tid] = (p.bias && !partial) ? p.bias[co0 + tid] : 0.f;
    __syncthreads();
    const bool accumulate = p.accumulate != 0 && !partial;
    const bool relu = p.relu != 0 && !partial;
    const bool out_mask = p.out_mask != nullptr && !partial;
    unsigned int amax = 0;
    // 16-byte path (W % 4 == 0, aligned bases; see st_conv_split.hip): every wave transposes its 32-channel slab
    // through a private LDS region and moves float4s along the image rows.  For the 1x1 kernel at large images
    // (64 KB of output per 128 MFMAs) the dword path below WAS the kernel.
    const bool vec_ok = (W % 4 == 0) &&
                        (((reinterpret_cast<uintptr_t>(out_base) | reinterpret_cast<uintptr_t>(p.out_mask)) & 15) == 0);
    if (vec_ok) {
        constexpr int TP = WN * 32 + 8;                        // slab pitch: 4 rows apart = 32 banks apart
        const int wave_id = wm * C::WGN + wn;
        float* slab = smem + C::TCO + wave_id * (32 * TP);
        static_assert(C::TCO + 4 * 32 * (WN * 32 + 8) <= 2 * C::BUF_FLOATS, "epilogue slabs must fit the staging LDS");
        const int lane = tid & 63;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co_base = co0 + wm * 64 + i * 32;
            const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(
                out_base + (size_t)co_base * HW, 0, 32 * HW * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t ms = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(out_mask ? p.out_mask : out_base) + (size_t)co_base * HW, 0, 32 * HW * 4, 0x00020000);
            __builtin_amdgcn_wave_barrier();                   // the previous half's reads are done (in-order LDS)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    slab[row * TP + j * 32 + l31] = acc[i][j][r];
                }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 4 * WN; ++t) {
                const int q = lane + 64 * t;
                const int row = q / (WN * 8), px = (q % (WN * 8)) * 4;          // 4 consecutive pixels of one row
                const int pix = wn * WN * 32 + px;
                const int y = y0 + pix / TW, x = x0 + pix % TW;
                const bool inb = (y < H) && (x < W);
                const int off = inb ? (row * HW + y * W + x) * 4 : 0x7FFFFFFF;
                f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * TP + px);
                const float bv = bias_lds[wm * 64 + i * 32 + row];
                f32x4 o, m;
                if (accumulate) o = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(os, off, 0, 0));
                if (out_mask) m = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ms, off, 0, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x_ = v[e] + bv;
                    if (relu) x_ = fmaxf(x_, 0.f);
                    if (accumulate) x_ += o[e];
                    if (out_mask) x_ = (m[e] > 0.f) ? x_ : 0.f;
                    v[e] = x_;
                    amax = max(amax, inb ? abs_bits(x_) : 0u);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), os, off, 0, 0);
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int co_base = co0 + wm * 64 + i * 32;
        const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(
            out_base + (size_t)co_base * HW, 0, 32 * HW * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int pix = (wn * WN + j) * 32 + l31;
            const int y = y0 + pix / TW, x = x0 + pix % TW;
            const bool inb = (y < H) && (x < W);
            const int pix_bytes = inb ? (y * W + x) * 4 : 0x7FFFFFFF;
            float old[16];
            if (accumulate) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    old[r] = buffer_load_f32(os, inb ? row * HW * 4 + pix_bytes : 0x7FFFFFFF);
                }
            }
            float msk[16];
            if (out_mask) {
                const __amdgpu_buffer_rsrc_t ms = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.out_mask) + (size_t)co_base * HW, 0, 32 * HW * 4, 0x00020000);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    msk[r] = buffer_load_f32(ms, inb ? row * HW * 4 + pix_bytes : 0x7FFFFFFF);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[i][j][r] + bias_lds[wm * 64 + i * 32 + row];
                if (relu) v = fmaxf(v, 0.f);
                if (accumulate) v += old[r];
                if (out_mask) v = (msk[r] > 0.f) ? v : 0.f;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), os,
                                                      inb ? row * HW * 4 + pix_bytes : 0x7FFFFFFF, 0, 0);
                amax = max(amax, inb ? abs_bits(v) : 0u);
            }
        }
    }
    if (p.out_amax && !partial) amax_commit(amax, p.out_amax);
}

// out = epilogue(sum over slices in slice order + bias): deterministic, one pass over Cout*H*W.  V = 4: 16-byte
// accesses (hw % 4 == 0, so a vector never straddles two channels); V = 1 for ragged maps.
template <int V>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const float* __restrict__ scratch,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ out, int cout, int hw,
                                                                 int ksplit, int relu, int accumulate,
                                                                 unsigned int* out_amax,
                                                                 const float* __restrict__ out_mask) {
    typedef float vec __attribute__((ext_vector_type(V)));
    const long long total = (long long)cout * hw / V;
    unsigned int amax = 0;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        vec v = reinterpret_cast<const vec*>(scratch)[i];
        for (int k = 1; k < ksplit; ++k) v += reinterpret_cast<const vec*>(scratch)[(size_t)k * total + i];
        if (bias) v += bias[(i * V) / hw];
        vec o, m;
        if (accumulate) o = reinterpret_cast<const vec*>(out)[i];
        if (out_mask) m = reinterpret_cast<const vec*>(out_mask)[i];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float x = v[e];
            if (relu) x = fmaxf(x, 0.f);
            if (accumulate) x += o[e];
            if (out_mask) x = (m[e] > 0.f) ? x : 0.f;
            v[e] = x;
            amax = max(amax, abs_bits(x));
        }
        reinterpret_cast<vec*>(out)[i] = v;
    }
    if (out_amax) amax_commit(amax, out_amax);
}

template <int TAPS, int TW, int WN, int WGM, bool MASKED, bool HALO>
int launch_cfg_m(const ConvProblem& p, int ksplit, hipStream_t stream) {
    using C = Cfg<TAPS, TW, WN, WGM>;
    static bool attr_set = false;
    auto kern = conv_mfma_kernel<TAPS, TW, WN, WGM, MASKED, HALO>;
    if (!attr_set) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
        attr_set = true;
    }
    const int tiles_x = ceil_div(p.width, TW), tiles_y = ceil_div(p.height, C::TH);
    const int n_co_tiles = p.cout / C::TCO;
    const long long blocks = (long long)tiles_x * tiles_y * n_co_tiles * ksplit;
    ST_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv grid out of range");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, stream, p, tiles_x, n_co_tiles,
                       ksplit);
    ST_LAUNCH_CHECK();
    if (ksplit > 1) return launch_conv_splitk_reduce(p, ksplit, stream);
    return 0;
}

template <int TAPS, int TW, int WN, int WGM>
int launch_cfg(const ConvProblem& p, int ksplit, hipStream_t stream) {
    if constexpr (TAPS == 9) {
        if (p.in_halo) {       // strip-sharded plans only
            if (p.mask) return launch_cfg_m<TAPS, TW, WN, WGM, true, true>(p, ksplit, stream);
            return launch_cfg_m<TAPS, TW, WN, WGM, false, true>(p, ksplit, stream);
        }
    }
    if (p.mask) return launch_cfg_m<TAPS, TW, WN, WGM, true, false>(p, ksplit, stream);
    return launch_cfg_m<TAPS, TW, WN, WGM, false, false>(p, ksplit, stream);
}

long long padded_area(int h, int w, int th, int tw) {
    return (long long)ceil_div(h, th) * th * (long long)ceil_div(w, tw) * tw;
}

template <int WN, int WGM>
int launch_3x3(const ConvProblem& p, int ksplit, hipStream_t s) {
    constexpr int NPIX = 32 * WN * (4 / WGM);
    // choose the tile width with the least padded area (ties -> wider rows: longer store segments)
    int best = 32;
    long long best_area = padded_area(p.height, p.width, NPIX / 32, 32);
    for (int tw : {16, 8}) {
        const long long a = padded_area(p.height, p.width, NPIX / tw, tw);
        if (a < best_area) { best_area = a; best = tw; }
    }
    if (best == 32) return launch_cfg<9, 32, WN, WGM>(p, ksplit, s);
    if (best == 16) return launch_cfg<9, 16, WN, WGM>(p, ksplit, s);
    return launch_cfg<9, 8, WN, WGM>(p, ksplit, s);
}

}  // namespace

int launch_conv_splitk_reduce(const ConvProblem& p, int ksplit, hipStream_t stream) {
    const long long total = (long long)p.cout * p.height * p.width;
    const int hw = p.height * p.width;
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.out_mask)) & 15) == 0;
    if (hw % 4 == 0 && aligned) {
        const int rblocks = (int)std::min<long long>((total / 4 + 255) / 256, 4096);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<4>, dim3(rblocks), dim3(256), 0, stream, p.scratch, p.bias, p.out,
                           p.cout, hw, ksplit, p.relu, p.accumulate, p.out_amax, p.out_mask);
    } else {
        const int rblocks = (int)std::min<long long>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel<1>, dim3(rblocks), dim3(256), 0, stream, p.scratch, p.bias, p.out,
                           p.cout, hw, ksplit, p.relu, p.accumulate, p.out_amax, p.out_mask);
    }
    ST_LAUNCH_CHECK();
    return 0;
}

double conv_flops(const ConvProblem& p) {
    return 2.0 * p.taps * (double)p.cin * p.cout * (double)p.height * p.width;
}

// Experiment knobs (microbenchmark only): ST_CONV_TUNE bits -> ConvProblem::tune,
// ST_CONV_SHAPE 0/1/2 forces the tile shape, ST_CONV_KSPLIT forces the split.
static int env_int(const char* name, int dflt) {
    const char* v = option_env(name);
    return v ? atoi(v) : dflt;
}

int launch_conv(const ConvProblem& p_in, hipStream_t stream) {
    ConvProblem p = p_in;
    if (p.tune == 0) p.tune = env_int("ST_CONV_TUNE", 0);
    ST_REQUIRE(p.taps == 9 || p.taps == 1, "conv: taps must be 9 or 1");
    if (p.planes > 0 && p.taps == 9 && p.wgt_split && p.cin % 16 == 0)
        return launch_conv_split(p, stream);
    const int KC = (p.taps == 9) ? kChunk3x3 : kChunk1x1;
    ST_REQUIRE(p.cin % KC == 0 && p.cout % 64 == 0, "conv: Cin %% %d and Cout %% 64 required (got %d, %d)", KC,
               p.cin, p.cout);
    ST_REQUIRE((long long)p.height * p.width * KC * 4 < (1ll << 31), "conv: image too large for 32-bit tile maps");
    const long long pixels = (long long)p.height * p.width;
    // Workgroup tile candidates (co x pixels): A = 64x256, B = 64x128, C = 128x64 (tiny images).
    // Target >= 512 workgroups (two per CU, i.e. two waves per SIMD to cover each other's barriers):
    // A if it gets there alone, else B with the input-channel range split over up to 8 workgroups.
    const int co_tiles = p.cout / 64;
    const long long wg_a = ((pixels + 255) / 256) * co_tiles;
    const long long wg_b = ((pixels + 127) / 128) * co_tiles;
    int shape = (wg_a >= 512) ? 0 : 1;
    if (pixels <= 64 && p.cout % 128 == 0) shape = 2;
    long long wgs = shape == 0 ? wg_a : (shape == 1 ? wg_b : ((pixels + 63) / 64) * (p.cout / 128));
    const int force_shape = env_int("ST_CONV_SHAPE", -1);
    if (force_shape >= 0 && (force_shape != 2 || p.cout % 128 == 0)) {
        shape = force_shape;
        wgs = shape == 0 ? wg_a : (shape == 1 ? wg_b : ((pixels + 63) / 64) * (p.cout / 128));
    }
    int ksplit = 1;
    const int force_ks = env_int("ST_CONV_KSPLIT", -1);
    if (force_ks >= 1 && p.scratch && (p.cin / KC) % force_ks == 0 &&
        (size_t)force_ks * p.cout * pixels <= kConvScratchFloats) {
        ksplit = force_ks;
    } else if (p.scratch && shape != 0) {
        const int nchunks = p.cin / KC;
        const int min_chunks = (p.taps == 9) ? 4 : 1;      // per K slice
        while (wgs * ksplit * 2 <= 640 && nchunks % (ksplit * 2) == 0 && nchunks / (ksplit * 2) >= min_chunks &&
               (size_t)(ksplit * 2) * p.cout * pixels <= kConvScratchFloats)
            ksplit *= 2;
    }
    if (ksplit == 1 && env_int("ST_CONV1X1_FP32", 0) == 0 && conv1x1_split_applies(p)) return launch_conv1x1_split(p, stream);
    if (p.taps == 1) {
        // no spatial structure: treat the image as one row of H*W pixels, tile = NPIX contiguous pixels
        ConvProblem q = p;
        q.height = 1;
        q.width = (int)pixels;
        if (shape == 0) return launch_cfg<1, 256, 2, 1>(q, ksplit, stream);
        if (shape == 1) return launch_cfg<1, 128, 1, 1>(q, ksplit, stream);
        return launch_cfg<1, 64, 1, 2>(q, ksplit, stream);
    }
    if (shape == 0) return launch_3x3<2, 1>(p, ksplit, stream);
    if (shape == 1) return launch_3x3<1, 1>(p, ksplit, stream);
    return launch_3x3<1, 2>(p, ksplit, stream);
}

// ---- boundary-row packing for the strip halo exchange -----------------------------------------
namespace {
// One block row per (channel, up / down): no index arithmetic per element; VEC = 4 where the rows are 16-byte aligned
// (round 4: 7.3 -> ~3 us per launch on a 2896-wide strip, 26 launches per iteration and rank).
// BOUND (round 5): the fp16x3 consumer of these rows needs max |row| for its operand scale; measured here, where the rows
// pass through registers anyway, and shipped with them (a trailer word of the message) - it used to be a device copy + an
// amax launch on the receiver's communication stream between the halo's arrival and the boundary launch.  Every block
// leaves its maximum in scratch[]; the block that draws the last ticket takes the maxima of the two directions (any order:
// max is exact) and writes the two trailer words.  Release / acquire at agent scope as in st_pointwise.hip's last-block kernels.
// Round 6: a workgroup packs `rows_per_block` (channel, direction) rows - a wave per row at a time - instead of one row
// segment: the 2 C x ceil(W / 256) workgroups of the first form each drew a ticket from ONE address (2 048 serialised atomics
// for a 512-channel map of 362 columns: 37 us per launch, 10 launches per iteration of a 2896 x 2172 strip, 6 % of a rank's
// step; profiles/r06_strip_breakdown.md); now <= 256 workgroups per launch.
template <int VEC, bool BOUND>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src,
                                                        const float* __restrict__ mask, int H, int W,
                                                        float* __restrict__ out_up,
                                                        float* __restrict__ out_down, unsigned int* __restrict__ bound_up,
                                                        unsigned int* __restrict__ bound_down, unsigned int* __restrict__ scratch,
                                                        int ticket_at, int rows_total, int rows_per_block) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned int m_up = 0, m_down = 0;
    const int r_end = min(rows_total, ((int)blockIdx.x + 1) * rows_per_block);
    for (int r = blockIdx.x * rows_per_block + wave; r < r_end; r += 4) {
        const int c = r >> 1;
        const bool down = r & 1;
        const size_t row = ((size_t)c * H + (down ? H - 1 : 0)) * W;
        float* __restrict__ dst = (down ? out_down : out_up) + (size_t)c * W;
        unsigned int m = 0;
        for (int x = lane * VEC; x < W; x += 64 * VEC) {
            if constexpr (VEC == 4) {
                f32x4 v = *reinterpret_cast<const f32x4*>(src + row + x);
                if (mask) {
                    const f32x4 mk = *reinterpret_cast<const f32x4*>(mask + row + x);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (mk[e] > 0.f) ? v[e] : 0.f;
                }
                *reinterpret_cast<f32x4*>(dst + x) = v;
                if constexpr (BOUND) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned int a = __float_as_uint(v[e]) & 0x7fffffffu;
                        m = a > m ? a : m;
                    }
                }
            } else {
                float v = src[row + x];
                if (mask) v = (mask[row + x] > 0.f) ? v : 0.f;
                dst[x] = v;
                if constexpr (BOUND) {
                    const unsigned int a = __float_as_uint(v) & 0x7fffffffu;
                    m = a > m ? a : m;
                }
            }
        }
        if (down) m_down = m > m_down ? m : m_down;
        else m_up = m > m_up ? m : m_up;
    }
    if constexpr (BOUND) {
        __shared__ unsigned int wave_max[2][4];
        __shared__ bool last_sh;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned int ou = (unsigned int)__shfl_xor((int)m_up, off), od = (unsigned int)__shfl_xor((int)m_down, off);
            m_up = ou > m_up ? ou : m_up;
            m_down = od > m_down ? od : m_down;
        }
        if (lane == 0) { wave_max[0][wave] = m_up; wave_max[1][wave] = m_down; }
        __syncthreads();
        const int nblocks = gridDim.x;
        if (threadIdx.x == 0) {
            unsigned int u = wave_max[0][0], d = wave_max[1][0];
#pragma unroll
            for (int w = 1; w < 4; ++w) { u = wave_max[0][w] > u ? wave_max[0][w] : u; d = wave_max[1][w] > d ? wave_max[1][w] : d; }
            scratch[2 * blockIdx.x] = u;
            scratch[2 * blockIdx.x + 1] = d;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned int prev = atomicAdd(scratch + ticket_at, 1u);
            last_sh = prev == (unsigned int)nblocks - 1;
            if (last_sh) scratch[ticket_at] = 0u;
        }
        __syncthreads();
        if (!last_sh) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // every reading thread (a CU's L1 is not refreshed by others' stores)
        unsigned int mu = 0, md = 0;
        for (int i = threadIdx.x; i < nblocks; i += 256) {
            const unsigned int vu = __hip_atomic_load(scratch + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int vd = __hip_atomic_load(scratch + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mu = vu > mu ? vu : mu;
            md = vd > md ? vd : md;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned int ou = (unsigned int)__shfl_xor((int)mu, off), od = (unsigned int)__shfl_xor((int)md, off);
            mu = ou > mu ? ou : mu;
            md = od > md ? od : md;
        }
        __shared__ unsigned int fin[2][4];
        if (lane == 0) { fin[0][wave] = mu; fin[1][wave] = md; }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned int u = fin[0][0], d = fin[1][0];
#pragma unroll
            for (int w = 1; w < 4; ++w) { u = fin[0][w] > u ? fin[0][w] : u; d = fin[1][w] > d ? fin[1][w] : d; }
            bound_up[0] = u;
            bound_down[0] = d;
        }
    }
}
}  // namespace

int launch_pack_rows(const float* src, const float* mask, int channels, int height, int width, float* out_up,
                     float* out_down, hipStream_t s, unsigned int* bounds_up, unsigned int* bounds_down, unsigned int* scratch) {
    const bool vec = width % 4 == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(mask) |
                                         reinterpret_cast<uintptr_t>(out_up) | reinterpret_cast<uintptr_t>(out_down)) & 15) == 0;
    const int rows_total = 2 * channels;
    // <= 256 workgroups; at least 4 rows each (one per wave)
    const int rows_per_block = std::max(4, 4 * ((rows_total + 4 * 256 - 1) / (4 * 256)));
    const dim3 grid((rows_total + rows_per_block - 1) / rows_per_block);
    const bool bound = bounds_up && bounds_down && scratch;
    ST_REQUIRE(!bound || 2ll * grid.x < kPackScratchUints - 1, "pack rows: %u blocks exceed the bound scratch", grid.x);
    const int ticket_at = kPackScratchUints - 1;
    if (bound) {
        if (vec) hipLaunchKernelGGL((pack_rows_kernel<4, true>), grid, dim3(256), 0, s, src, mask, height, width, out_up, out_down, bounds_up, bounds_down, scratch, ticket_at, rows_total, rows_per_block);
        else hipLaunchKernelGGL((pack_rows_kernel<1, true>), grid, dim3(256), 0, s, src, mask, height, width, out_up, out_down, bounds_up, bounds_down, scratch, ticket_at, rows_total, rows_per_block);
    } else {
        if (vec) hipLaunchKernelGGL((pack_rows_kernel<4, false>), grid, dim3(256), 0, s, src, mask, height, width, out_up, out_down, nullptr, nullptr, nullptr, 0, rows_total, rows_per_block);
        else hipLaunchKernelGGL((pack_rows_kernel<1, false>), grid, dim3(256), 0, s, src, mask, height, width, out_up, out_down, nullptr, nullptr, nullptr, 0, rows_total, rows_per_block);
    }
    ST_LAUNCH_CHECK();
    return 0;
}

// ---- weight re-layouts (once per network) ------------------------------------------------------
namespace {
__global__ void relayout_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout,
                                int dgrad) {
    const long long total = (long long)cin * cout * 9;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        // i indexes the torch tensor [cout][cin][3][3]
        const int tap = (int)(i % 9);
        const int ci = (int)((i / 9) % cin);
        const int co = (int)(i / (9ll * cin));
        if (!dgrad) {
            out[((size_t)tap * cin + ci) * cout + co] = w[i];          // [tap][ci][co]
        } else {
            out[((size_t)(8 - tap) * cout + co) * cin + ci] = w[i];    // [tap'][co][ci], tap' = 8 - tap
        }
    }
}
}  // namespace

int launch_relayout_fwd(const float* w, float* out, int cin, int cout, hipStream_t stream) {
    hipLaunchKernelGGL(relayout_kernel, dim3(1024), dim3(256), 0, stream, w, out, cin, cout, 0);
    ST_LAUNCH_CHECK();
    return 0;
}
int launch_relayout_dgrad(const float* w, float* out, int cin, int cout, hipStream_t stream) {
    hipLaunchKernelGGL(relayout_kernel, dim3(1024), dim3(256), 0, stream, w, out, cin, cout, 1);
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
