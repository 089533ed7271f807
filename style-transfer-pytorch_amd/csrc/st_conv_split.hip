// 3x3 convolution on the bf16 matrix pipe with SPLIT-PRECISION operands (opt-in).
//
// gfx950 has no TF32-like mode: an fp32 MFMA (v_mfma_f32_32x32x2_f32) issues at 1/16 of the bf16 rate.
// Here every fp32 operand x is written as a sum of bf16 planes, x = x0 + x1 (+ x2), each plane the bf16
// rounding of what the previous ones left over, and the product is assembled from bf16 MFMAs with fp32
// accumulation:
//   P = 2 planes, 3 products  a0 b0 + a0 b1 + a1 b0               (dropped terms <= 2^-16 |a b|)
//   P = 3 planes, 6 products  ... + a0 b2 + a1 b1 + a2 b0         (dropped terms <= 2^-24 |a b|: fp32-class)
// One v_mfma_f32_32x32x16_bf16 covers K = 16 in 32 cycles, so 16 input channels of one tap cost
// 96 / 192 matrix-pipe cycles instead of 512.  Activations stay fp32 in HBM: they are split while being
// staged into LDS; the frozen weights are split once per network.
//
// GEMM orientation, tiling, split-K and epilogue are those of st_conv.hip (M = Cout on the A operand,
// N = pixels on the lanes).  K = 16 input channels per MFMA: lane l supplies k = 8 (l >> 5) .. +7 for row /
// column l & 31, i.e. one 16-byte LDS read per operand per plane.  LDS images are channel-innermost:
//   activations [plane][tile pixel][16 ch]  (32 B per pixel),  weights [plane][tap][co][16 ch].
// The two 16-byte halves of a 32-byte row are swapped for every other group of 8 rows (XOR swizzle), which
// makes the ds_read_b128 of 16 consecutive rows hit 16 distinct 16-byte bank slots.
//
// fp16x3 (E = 1, P = 2): the same kernel on fp16 planes.  fp16 keeps 11 significant bits per plane, so with
// round-to-nearest x = h0 + h1 leaves |x - h0 - h1| <= 2^-24 |x| -- fp32's own rounding -- and the three
// products h0 g0 + h0 g1 + h1 g0 drop only h1 g1 <= 2^-24 |x y|: fp32-class accuracy for HALF the matrix
// work of bf16x6.  The price is fp16's 5-bit exponent: both operands are multiplied by a power of two that
// puts a bound on the tensor's max |x| into [2^13, 2^14) (exact, undone in the epilogue).  The bound is free:
// every kernel that finalises a conv operand folds max |out| into a device word (amax_commit in its epilogue)
// and the consumer reads that word; standalone operators measure the operand with amax_kernel instead.  The
// weights carry their max |w| in the buffer trailer.  Elements below ~2^-28 of the bound lose residual bits
// (absolute error <= 2^-38 of the bound).
#include <cstdlib>
#include <type_traits>

#include "st_common.h"

namespace st {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int E> struct Elem;
template <> struct Elem<0> { using scalar = __bf16; using vec = bf16x8; };
template <> struct Elem<1> { using scalar = _Float16; using vec = f16x8; };

__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

constexpr int SK = 16;                      // input channels per chunk (= K of one MFMA)
constexpr int kOOR = 0x40000000;

template <int B, int E, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        sfor<B + 1, E>(f);
    }
}

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}

template <int TW, int WN, int P>
struct SCfg {
    static constexpr int TCO = 64;
    static constexpr int NPIX = 32 * WN * 4;
    static constexpr int TH = NPIX / TW;
    static constexpr int LH = TH + 2, LW = TW + 2;
    static constexpr int NPX = LH * LW;                    // staged pixels (with the 1-pixel halo)
    static constexpr int ACT_PLANE = NPX * 32;             // bytes
    static constexpr int W_PLANE = 9 * TCO * 32;           // bytes
    static constexpr int LDS_BYTES = P * (ACT_PLANE + W_PLANE);
    static constexpr int NIT = (2 * NPX + 255) / 256;      // (pixel, 8-channel group) items per thread
    static constexpr int NWP = 9 * TCO * 2;                // 16-byte weight pieces per plane
    static constexpr int NWT = (NWP + 255) / 256;
};

// split 8 fp32 values into P 16-bit planes (each plane = round-to-nearest of what the previous ones left)
template <int P, typename V, typename S>
__device__ __forceinline__ void split8(const float (&v)[8], V (&out)[P]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float r = v[e];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const S h = (S)r;
            out[p][e] = h;
            r = r - (float)h;
        }
    }
}

// Two workgroups per CU is what hides this kernel's barrier phases.  The two-plane unmasked variants (what an
// unsharded plan runs since the ReLU mask moved to the producers) sit within a few registers of the 256 limit
// and are held to it; the build's scratch guard proves that costs no spills.  (The halo and three-plane 256-px
// variants would spill 23..137 registers under the same bound and keep one workgroup per CU.)
template <int TW, int WN, int P, int E, bool MASKED, bool HALO>
__global__ __launch_bounds__(256, (!MASKED && !HALO && P == 2) ? 2 : 1) void conv_split_kernel(ConvProblem p, int tiles_x,
                                                                                              int n_co_tiles,
                                                                                              int ksplit) {
    using C = SCfg<TW, WN, P>;
    using V = typename Elem<E>::vec;
    using S = typename Elem<E>::scalar;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* act_lds = smem;                                  // [P][NPX][32 B]
    unsigned char* w_lds = smem + P * C::ACT_PLANE;                 // [P][9][64][32 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);        // 4 waves along the pixel dimension
    const int l31 = lane & 31, half = lane >> 5;

    // Workgroups are dealt round-robin to the 8 XCDs (private L2 each).  Logical order: XCD x works through the
    // contiguous range [x total / 8, (x + 1) total / 8), in which consecutive ids are the Cout tiles (and K
    // slices) of ONE pixel tile - they share its activations through that XCD's L2 instead of 8 L2s fetching
    // them.  (Speed only; tune bit 64 = plain order, for A/B runs.)
    int bid = blockIdx.x;
    if ((gridDim.x & 7) == 0 && !(p.tune & 64)) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int co_tile = bid % n_co_tiles;
    bid /= n_co_tiles;
    const int kslice = bid % ksplit;
    bid /= ksplit;
    const int tile_x = bid % tiles_x, tile_y = bid / tiles_x;
    const int x0 = tile_x * TW, y0 = tile_y * C::TH, co0 = co_tile * C::TCO;
    const int H = p.height, W = p.width, HW = H * W;

    // ---- staging maps ----
    int goff[C::NIT];            // byte offset of channel (8 g) at this item's pixel, or out of range
    int aoff[C::NIT];            // LDS byte offset of the item's 16-byte slot inside a plane
    // (lanes past the end of the last item redo the final element - same address, same value - so that the
    // staging below carries no exec-mask branches)
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) {
        const int t = (tid + i * 256 < 2 * C::NPX) ? tid + i * 256 : 2 * C::NPX - 1;
        const int g = t / C::NPX, q = t % C::NPX;
        const int y = y0 - 1 + q / C::LW, x = x0 - 1 + q % C::LW;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W;
        goff[i] = ok ? (8 * g * HW + y * W + x) * 4 : kOOR;
        aoff[i] = q * 32 + ((g ^ ((q >> 3) & 1)) * 16);
    }
    // strip sharding: tile rows -1 / H come from the neighbours' halo block [2][Cin][W] (second resource)
    int hoff[HALO ? C::NIT : 1];
    if constexpr (HALO) {
#pragma unroll
        for (int i = 0; i < C::NIT; ++i) {
            const int t = (tid + i * 256 < 2 * C::NPX) ? tid + i * 256 : 2 * C::NPX - 1;
            const int g = t / C::NPX, q = t % C::NPX;
            const int y = y0 - 1 + q / C::LW, x = x0 - 1 + q % C::LW;
            const bool xin = x >= 0 && x < W;
            const bool top = xin && y == -1 && p.has_up, bot = xin && y == H && p.has_down;
            hoff[i] = top ? (8 * g * W + x) * 4 : (bot ? ((p.cin + 8 * g) * W + x) * 4 : kOOR);
        }
    }
    const unsigned char* wsplit = static_cast<const unsigned char*>(p.wgt_split);
    const size_t w_plane_stride = (size_t)9 * (p.cin / SK) * p.cout * 32;       // bytes per plane
    // fp16 mode: power-of-two scales (wave-uniform); in_scale multiplies the staged activations, out_scale
    // undoes both operand scales in the epilogue
    float in_scale = 1.f, out_scale_a = 1.f, out_scale_w = 1.f;
    if constexpr (E == 1) {
        const int ea = scale_exp(amax_with_halo(amax_read(p.amax_word), p.halo_bound_up, p.halo_bound_down));
        const int ew = scale_exp(*reinterpret_cast<const unsigned int*>(wsplit + P * w_plane_stride));
        in_scale = pow2f(ea);
        out_scale_a = pow2f(-ea);
        out_scale_w = pow2f(-ew);
    }
    const size_t w_tap_stride = (size_t)(p.cin / SK) * p.cout * 32;

    float ract[C::NIT][8];
    float rhal[HALO ? C::NIT : 1][8];
    float rmsk[MASKED ? C::NIT : 1][8];
    f32x4 rwt[P][C::NWT];
    const int chunk_bytes = SK * HW * 4;

    auto load_chunk = [&](int cc) __attribute__((always_inline)) {     // cc = chunk index (16 channels)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in) + (size_t)cc * SK * HW, 0, chunk_bytes, 0x00020000);
        sfor<0, C::NIT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
#pragma unroll
            for (int c = 0; c < 8; ++c) ract[i][c] = bload(rs, goff[i], c * HW * 4);
        });
        if constexpr (HALO) {
            const __amdgpu_buffer_rsrc_t hs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.in_halo) + (size_t)cc * SK * W, 0, (p.cin + SK) * W * 4, 0x00020000);
            sfor<0, C::NIT>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
#pragma unroll
                for (int c = 0; c < 8; ++c) rhal[i][c] = bload(hs, hoff[i], c * W * 4);
            });
        }
        if constexpr (MASKED) {
            const __amdgpu_buffer_rsrc_t ms = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.mask) + (size_t)cc * SK * HW, 0, chunk_bytes, 0x00020000);
            sfor<0, C::NIT>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
#pragma unroll
                for (int c = 0; c < 8; ++c) rmsk[i][c] = bload(ms, goff[i], c * HW * 4);
            });
        }
        sfor<0, P>([&](auto PL) __attribute__((always_inline)) {
            constexpr int pl = decltype(PL)::value;
            sfor<0, C::NWT>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                const int f = (tid + i * 256 < C::NWP) ? tid + i * 256 : C::NWP - 1;
                const int tap = f / (C::TCO * 2), r = f % (C::TCO * 2);
                rwt[pl][i] = *reinterpret_cast<const f32x4*>(wsplit + pl * w_plane_stride + tap * w_tap_stride +
                                                             ((size_t)cc * p.cout + co0) * 32 + r * 16);
            });
        });
    };
    auto store_chunk = [&]() __attribute__((always_inline)) {
        sfor<0, C::NIT>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                v[c] = ract[i][c];
                if constexpr (MASKED) v[c] = (rmsk[i][c] > 0.f) ? v[c] : 0.f;      // threshold_backward
                if constexpr (HALO) v[c] += rhal[i][c];      // neighbour rows arrive already masked; 0 elsewhere
                if constexpr (E == 1) v[c] *= in_scale;
            }
            V planes[P];
            split8<P, V, S>(v, planes);
#pragma unroll
            for (int pl = 0; pl < P; ++pl) *reinterpret_cast<V*>(act_lds + pl * C::ACT_PLANE + aoff[i]) = planes[pl];
        });
        sfor<0, P>([&](auto PL) __attribute__((always_inline)) {
            constexpr int pl = decltype(PL)::value;
            sfor<0, C::NWT>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                const int f = (tid + i * 256 < C::NWP) ? tid + i * 256 : C::NWP - 1;
                const int row = f >> 1, hsel = f & 1;                // row = tap * 64 + co
                *reinterpret_cast<f32x4*>(w_lds + pl * C::W_PLANE + row * 32 + ((hsel ^ ((row >> 3) & 1)) * 16)) =
                    rwt[pl][i];
            });
        });
    };

    // ---- operand addresses ----
    int a_off[2];                                   // + tap * 64 * 32
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int co = i * 32 + l31;
        a_off[i] = co * 32 + ((half ^ ((co >> 3) & 1)) * 16);
    }
    int b_off[WN][9];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int pix = (wn * WN + j) * 32 + l31;
        const int qb = (pix / TW) * C::LW + (pix % TW);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int q = qb + (tap / 3) * C::LW + (tap % 3);
            b_off[j][tap] = q * 32 + ((half ^ ((q >> 3) & 1)) * 16);
        }
    }

    f32x16 acc[2][WN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto fetch_tap = [&](auto TAP, V (&av)[2][P], V (&bv)[WN][P]) __attribute__((always_inline)) {
        constexpr int tap = decltype(TAP)::value;
#pragma unroll
        for (int pl = 0; pl < P; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                av[i][pl] = *reinterpret_cast<const V*>(w_lds + pl * C::W_PLANE + tap * (C::TCO * 32) + a_off[i]);
#pragma unroll
            for (int j = 0; j < WN; ++j)
                bv[j][pl] = *reinterpret_cast<const V*>(act_lds + pl * C::ACT_PLANE + b_off[j][tap]);
        }
    };
    auto mfma_tap = [&](const V (&av)[2][P], const V (&bv)[WN][P]) __attribute__((always_inline)) {
        // small cross terms first, the dominant a0*b0 last
        sfor<0, P>([&](auto S) __attribute__((always_inline)) {
            constexpr int s = P - 1 - decltype(S)::value;          // s = pa + pb, descending
            sfor<0, s + 1>([&](auto PA) __attribute__((always_inline)) {
                constexpr int pa = decltype(PA)::value, pb = s - pa;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = mfma16(av[i][pa], bv[j][pb], acc[i][j]);
            });
        });
    };
    auto compute = [&]() __attribute__((always_inline)) {
        V a0[2][P], b0[WN][P], a1[2][P], b1[WN][P];
        fetch_tap(std::integral_constant<int, 0>{}, a0, b0);
        sfor<0, 5>([&](auto T2) __attribute__((always_inline)) {
            constexpr int tap = 2 * decltype(T2)::value;
            if constexpr (tap + 1 < 9) fetch_tap(std::integral_constant<int, tap + 1>{}, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_tap(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (tap + 1 < 9) {
                if constexpr (tap + 2 < 9) fetch_tap(std::integral_constant<int, tap + 2>{}, a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tap(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    // ---- K loop: single LDS image, the next chunk waits in registers ----
    const int nchunks = p.cin / SK / ksplit;
    const int chunk0 = kslice * nchunks;
    // tune bit 32 (tools/conv_bench.py, ST_CONV_PHASES=1): s_memtime sums of wave 0 per phase -> p.scratch
    const bool stamp = (p.tune & 32) != 0 && ksplit == 1 && p.scratch != nullptr;
    unsigned long long t_ph[5] = {0, 0, 0, 0, 0}, t_prev = 0, t_begin = 0;
    auto mark = [&](int k) __attribute__((always_inline)) {
        if (stamp) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            t_ph[k] += t - t_prev;
            t_prev = t;
        }
    };
    if (stamp) t_begin = t_prev = __builtin_amdgcn_s_memtime();
    load_chunk(chunk0);
    mark(0);                                 // prologue: maps + first loads issued
    for (int c = 0; c < nchunks; ++c) {
        store_chunk();
        mark(1);                             // wait for the loads + convert + LDS writes
        __syncthreads();
        mark(2);                             // barrier 1
        if (c + 1 < nchunks) load_chunk(chunk0 + c + 1);
        compute();
        mark(3);                             // load issue + operand fetch + MFMAs
        __syncthreads();
        mark(4);                             // barrier 2
    }

    // ---- epilogue (as in st_conv.hip) ----
    const bool partial = ksplit > 1;
    float* out_base = partial ? p.scratch + (size_t)kslice * p.cout * HW : p.out;
    float* bias_lds = reinterpret_cast<float*>(smem);
    if (tid < C::TCO) bias_lds[tid] = (p.bias && !partial) ? p.bias[co0 + tid] : 0.f;
    __syncthreads();
    const bool accumulate = p.accumulate != 0 && !partial;
    const bool relu = p.relu != 0 && !partial;
    const bool out_mask = p.out_mask != nullptr && !partial;
    unsigned int amax = 0;
    // 16-byte path (W % 4 == 0, aligned bases): each wave transposes its 32-channel x (32 WN)-pixel slab through a
    // private LDS region and moves whole float4s along the image rows - 4 WN stores (and accumulate / mask loads)
    // per lane and channel half instead of 16 WN.  The dword path below (ragged widths) issued 64 stores per
    // lane for a 64 x 256 tile and was store-issue bound (~14k cycles per workgroup).
    const bool vec_ok = (W % 4 == 0) &&
                        (((reinterpret_cast<uintptr_t>(out_base) | reinterpret_cast<uintptr_t>(p.out_mask)) & 15) == 0);
    if (vec_ok) {
        constexpr int TP = WN * 32 + 8;                        // slab pitch: 4 rows apart = 32 banks apart
        float* slab = reinterpret_cast<float*>(smem) + 64 + wn * (32 * TP);
        static_assert((64 + 4 * 32 * (WN * 32 + 8)) * 4 <= SCfg<TW, WN, P>::LDS_BYTES, "epilogue slabs must fit the staging LDS");
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co_base = co0 + i * 32;
            const __amdgpu_buffer_rsrc_t os =
                __builtin_amdgcn_make_buffer_rsrc(out_base + (size_t)co_base * HW, 0, 32 * HW * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t ms = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(out_mask ? p.out_mask : out_base) + (size_t)co_base * HW, 0, 32 * HW * 4, 0x00020000);
            __builtin_amdgcn_wave_barrier();                   // the previous half's reads are done (in-order LDS)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = acc[i][j][r];
                    if constexpr (E == 1) v = v * out_scale_a * out_scale_w;
                    slab[row * TP + j * 32 + l31] = v;
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
                const float bv = bias_lds[i * 32 + row];
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
        const int co_base = co0 + i * 32;
        const __amdgpu_buffer_rsrc_t os =
            __builtin_amdgcn_make_buffer_rsrc(out_base + (size_t)co_base * HW, 0, 32 * HW * 4, 0x00020000);
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
                    old[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           os, inb ? row * HW * 4 + pix_bytes : 0x7FFFFFFF, 0, 0));
                }
            }
            float msk[16];
            if (out_mask) {
                const __amdgpu_buffer_rsrc_t ms = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<float*>(p.out_mask) + (size_t)co_base * HW, 0, 32 * HW * 4, 0x00020000);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                    msk[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                           ms, inb ? row * HW * 4 + pix_bytes : 0x7FFFFFFF, 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[i][j][r];
                if constexpr (E == 1) v = v * out_scale_a * out_scale_w;
                v += bias_lds[i * 32 + row];
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
    if (stamp && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.scratch) + (size_t)blockIdx.x * 8;
        for (int k = 0; k < 5; ++k) dst[k] = t_ph[k];
        dst[5] = __builtin_amdgcn_s_memtime() - t_begin;
        dst[6] = 1;
    }
}

// (Round 1 also carried a software-pipelined variant of this kernel - one workgroup per CU, double-buffered LDS images,
// the staging of chunk c + 1 hidden in the MFMA shadow of chunk c: 65 % MFMA-busy in its K loop against 46 % here, but
// its exposed prologue / epilogue lost on every layer at 512^2 (trunk 1762 vs 1621 us) and tied at 1024^2.  The
// persistent producer / consumer kernel of st_conv_pc.hip is what that line of work became; the variant is removed.)

template <int TW, int WN, int P, int E, bool MASKED, bool HALO>
int launch_split_cfg_h(const ConvProblem& p, int ksplit, hipStream_t stream) {
    using C = SCfg<TW, WN, P>;
    static bool attr_set = false;
    auto kern = conv_split_kernel<TW, WN, P, E, MASKED, HALO>;
    if (!attr_set) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   C::LDS_BYTES));
        attr_set = true;
    }
    const int tiles_x = ceil_div(p.width, TW), tiles_y = ceil_div(p.height, C::TH);
    const int n_co_tiles = p.cout / C::TCO;
    const long long blocks = (long long)tiles_x * tiles_y * n_co_tiles * ksplit;
    ST_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv grid out of range");
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), C::LDS_BYTES, stream, p, tiles_x, n_co_tiles, ksplit);
    ST_LAUNCH_CHECK();
    if (ksplit > 1) return launch_conv_splitk_reduce(p, ksplit, stream);
    return 0;
}

template <int TW, int WN, int P, int E, bool MASKED>
int launch_split_cfg(const ConvProblem& p, int ksplit, hipStream_t stream) {
    if (p.in_halo) return launch_split_cfg_h<TW, WN, P, E, MASKED, true>(p, ksplit, stream);
    return launch_split_cfg_h<TW, WN, P, E, MASKED, false>(p, ksplit, stream);
}

template <int WN, int P, int E>
int launch_split_tw(const ConvProblem& p, int ksplit, hipStream_t s) {
    constexpr int NPIX = 32 * WN * 4;
    auto area = [&](int tw) {
        return (long long)ceil_div(p.height, NPIX / tw) * (NPIX / tw) * (long long)ceil_div(p.width, tw) * tw;
    };
    int best = 32;
    long long best_area = area(32);
    for (int tw : {16, 8})
        if (area(tw) < best_area) { best_area = area(tw); best = tw; }
    const bool m = p.mask != nullptr;
    if (best == 32) return m ? launch_split_cfg<32, WN, P, E, true>(p, ksplit, s) : launch_split_cfg<32, WN, P, E, false>(p, ksplit, s);
    if (best == 16) return m ? launch_split_cfg<16, WN, P, E, true>(p, ksplit, s) : launch_split_cfg<16, WN, P, E, false>(p, ksplit, s);
    return m ? launch_split_cfg<8, WN, P, E, true>(p, ksplit, s) : launch_split_cfg<8, WN, P, E, false>(p, ksplit, s);
}

// max |x| of a tensor as raw bits (non-negative floats order like unsigned integers), folded into a slotted
// bound (amax_commit) or, with single != 0, into one plain word (the weight buffers' trailer); the caller
// zeroed the destination.  Streaming read, float4 when the pointer allows it.
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long n, unsigned int* word,
                                                   int single) {
    unsigned int m = 0;
    const long long tid = blockIdx.x * (long long)blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const long long n4 = n / 4;
        const uint4* x4 = reinterpret_cast<const uint4*>(x);
        for (long long i = tid; i < n4; i += nth) {
            const uint4 v = x4[i];
            m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
        }
        for (long long i = n4 * 4 + tid; i < n; i += nth) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    } else {
        for (long long i = tid; i < n; i += nth) m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
    }
    if (!single) {
        amax_commit(m, word);
        return;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, off));
    __shared__ unsigned int wave_max[4];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(word, max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3])));
}

}  // namespace

int launch_amax(const float* x, long long n, unsigned int* word, int single, hipStream_t s) {
    if (n <= 0) return 0;
    const long long want = (n / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(amax_kernel, dim3(blocks), dim3(256), 0, s, x, n, word, single);
    ST_LAUNCH_CHECK();
    return 0;
}

namespace {

// torch [Cout][Cin][3][3] fp32 -> 16-bit planes [P][9][K/16][M][16] (forward: K = Cin, M = Cout; data gradient:
// K = Cout, M = Cin, taps rotated by 180 degrees).  E = 1: values pre-scaled by 2^scale_exp(max |w|), which the
// trailer word (written by amax_kernel just before) holds.
template <int E>
__global__ void relayout_split_kernel(const float* __restrict__ w, typename Elem<E>::scalar* __restrict__ out, int cin,
                                      int cout, int dgrad, int planes) {
    using S = typename Elem<E>::scalar;
    const long long total = (long long)cin * cout * 9;
    const int K = dgrad ? cout : cin, M = dgrad ? cin : cout;
    const size_t plane = (size_t)9 * K * M;
    float scale = 1.f;
    if constexpr (E == 1) scale = pow2f(scale_exp(*reinterpret_cast<const unsigned int*>(out + planes * plane)));
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 9);
        const int ci = (int)((i / 9) % cin);
        const int co = (int)(i / (9ll * cin));
        const int k = dgrad ? co : ci, m = dgrad ? ci : co, t = dgrad ? 8 - tap : tap;
        const size_t idx = (((size_t)t * (K / SK) + k / SK) * M + m) * SK + k % SK;
        float r = w[i] * scale;
        for (int pl = 0; pl < planes; ++pl) {
            const S h = (S)r;
            out[pl * plane + idx] = h;
            r = r - (float)h;
        }
    }
}

}  // namespace

int launch_relayout_split(const float* w, void* out, int cin, int cout, int dgrad, int planes, int elem,
                          hipStream_t s) {
    ST_REQUIRE((planes == 2 || planes == 3) && (elem == 0 || (elem == 1 && planes == 2)),
               "split weights: planes %d / element type %d not supported", planes, elem);
    const size_t wcount = (size_t)cin * cout * 9;
    unsigned int* trailer = reinterpret_cast<unsigned int*>(static_cast<unsigned char*>(out) + wcount * 2 * planes);
    ST_HIP(hipMemsetAsync(trailer, 0, 256, s));
    if (launch_amax(w, (long long)wcount, trailer, 1, s)) return 1;
    if (elem == 1)
        hipLaunchKernelGGL(relayout_split_kernel<1>, dim3(1024), dim3(256), 0, s, w, static_cast<_Float16*>(out), cin,
                           cout, dgrad, planes);
    else
        hipLaunchKernelGGL(relayout_split_kernel<0>, dim3(1024), dim3(256), 0, s, w, static_cast<__bf16*>(out), cin,
                           cout, dgrad, planes);
    ST_LAUNCH_CHECK();
    return 0;
}

// fp16x3: the neighbours' halo rows are operands too - fold max |row| into the operand's bound.  The halo block is
// [2][Cin][W] (top rows, then bottom rows): one launch covers both when the strip has both neighbours.
int fold_halo_amax(const ConvProblem& p, hipStream_t stream) {
    if (!p.in_halo || p.elem != 1 || !p.amax_word) return 0;
    const long long row = (long long)p.cin * p.width;
    if (p.has_up && p.has_down) return launch_amax(p.in_halo, 2 * row, p.amax_word, 0, stream);
    if (p.has_up) return launch_amax(p.in_halo, row, p.amax_word, 0, stream);
    if (p.has_down) return launch_amax(p.in_halo + (size_t)row, row, p.amax_word, 0, stream);
    return 0;
}

int launch_conv_split(const ConvProblem& p, hipStream_t stream) {
    ST_REQUIRE(p.taps == 9 && p.wgt_split != nullptr && (p.planes == 2 || p.planes == 3),
               "split conv: needs 3x3 taps and 2 or 3 weight planes");
    ST_REQUIRE(p.elem == 0 || (p.elem == 1 && p.planes == 2 && p.amax_word != nullptr),
               "split conv: fp16 planes need planes == 2 and an amax word");
    ST_REQUIRE(p.cin % SK == 0 && p.cout % 64 == 0, "split conv: Cin %% 16 and Cout %% 64 required (got %d, %d)",
               p.cin, p.cout);
    ST_REQUIRE((long long)p.height * p.width * SK * 4 < (1ll << 30), "split conv: image too large");
    const long long pixels = (long long)p.height * p.width;
    const int co_tiles = p.cout / 64;
    const long long wg_a = ((pixels + 255) / 256) * co_tiles, wg_b = ((pixels + 127) / 128) * co_tiles;
    // (bf16x6, three planes: the 256-pixel tile does not fit the register budget)
    const bool big = (wg_a >= 512) && p.planes != 3;
    long long wgs = big ? wg_a : wg_b;
    int ksplit = 1;
    if (p.scratch && !big) {
        const int nchunks = p.cin / SK;
        while (wgs * ksplit * 2 <= 640 && nchunks % (ksplit * 2) == 0 && nchunks / (ksplit * 2) >= 2 &&
               (size_t)(ksplit * 2) * p.cout * pixels <= kConvScratchFloats)
            ksplit *= 2;
    }
    if (p.elem == 1) {
        // standalone operators measure max |in| here (the mask only removes elements, the bound stays valid);
        // inside a plan the producer of `in` already left it in the word.  With strip sharding the neighbours'
        // halo rows are operands too.
        if (p.amax_measure && launch_amax(p.in, (long long)p.cin * pixels, p.amax_word, 0, stream)) return 1;
        if (!p.halo_amax_folded && fold_halo_amax(p, stream)) return 1;
        // The producer / consumer kernel (st_conv_pc.hip) takes the layers where it measured faster: see
        // conv_pc_preferred.  ST_CONV_PC=0 disables it, =2 forces it for every eligible problem (A/B runs).
        if (p.wgt_wino && p.wino && (p.wino > 1 ? conv_wino_applies(p) : conv_wino_preferred(p))) return launch_conv_wino(p, stream);
        if (conv_fat_preferred(p)) return launch_conv_fat(p, stream);          // large maps of >= 128 channels (st_conv_fat.hip)
        static Option use_pc_opt("ST_CONV_PC", 1);
        const int use_pc = use_pc_opt.get();
        if (use_pc && (use_pc > 1 ? conv_pc_applies(p) : conv_pc_preferred(p))) return launch_conv_pc(p, stream);
        return big ? launch_split_tw<2, 2, 1>(p, ksplit, stream) : launch_split_tw<1, 2, 1>(p, ksplit, stream);
    }
    if (p.planes == 2) return big ? launch_split_tw<2, 2, 0>(p, ksplit, stream) : launch_split_tw<1, 2, 0>(p, ksplit, stream);
    return big ? launch_split_tw<2, 3, 0>(p, ksplit, stream) : launch_split_tw<1, 3, 0>(p, ksplit, stream);
}

}  // namespace st
