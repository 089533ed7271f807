// PROTOTYPE (round 5; operator level only - st_op_conv3x3 / st_op_conv3x3_time precision code 5; the
// plan does not use it): the 3 x 3 convolution of the trunk (nn.Conv2d 3x3, padding 1, + bias + ReLU: reference
// style_transfer.py:35,87) as Winograd F(2 x 2, 3 x 3) on the fp16x3 planes.
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          per 2 x 2 output tile, 4 x 4 input patch d, 3 x 3 filter g
//
// 16 multiplications per 4 outputs instead of 36: 2.25 x fewer MFMAs.  The filters are transformed once, in double, and
// split into two fp16 planes under one power-of-two scale (wino_weights_kernel); the input patches are transformed in fp32
// (additions only) BEFORE the split; each of the 16 transform positions is a plane GEMM h0 g0 + h0 g1 + h1 g0 with fp32
// accumulation; the output transform runs on the accumulators in registers.  Accuracy by CPU emulation: 1.5 - 2.1e-7 per
// convolution against float64, the class of the direct fp16x3 form (profiles/r05_winograd.md).
//
// Shape of this first kernel - SINGLE-ROLE (no producer / consumer specialisation, single-buffered LDS), one workgroup of four
// waves per 64 co x (16 x 16 px = 64 tiles): per 16-channel chunk
//   (a) the chunk's transformed weights, 64 KB, by LDS-DMA;   (b) the raw 18 x 18 x 16 input patch into LDS (zero padding);
//   (c) every thread transforms (tile, 4 channels) and writes the 16 positions' planes in MFMA B-operand order (64 KB);
//   (d) every wave: 32 co x 32 tiles x 16 positions - 48 v_mfma_f32_32x32x16_f16, 16 accumulators (256 registers).
// The consumer pattern of (d) alone sustains 1990 TF of MFMA work = 1495 TF of direct-convolution-equivalent work per chip
// (tools/winograd_rate.py; the direct tile's pattern: 1890 TF = 630 equivalent): the matrix side has a 2.4 x higher ceiling.
// Whether (a) - (c), serialised with (d) here, leave anything of it is what this prototype measures.
#include "st_common.h"

namespace st {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int kRawPitch = 20;                       // floats per raw patch row (18 used; 80 B keeps 8-byte reads aligned)
constexpr int kRawPlane = 18 * kRawPitch;
constexpr int kLdsA = 0, kLdsB = 65536, kLdsRaw = 131072;
constexpr int kLdsBytes = kLdsRaw + 16 * kRawPlane * 4;          // 154 112 B

// torch [Cout][Cin][3][3] -> U = G g G^T (double), two fp16 planes under 2^e, e from the bound 2.25 max |w|;
// layout [chunk = Cin / 16][position 16][co block = Cout / 32][plane 2][lane 64][8 halves]: a 1 KB block IS the A operand of
// v_mfma_f32_32x32x16_f16 (lane l: row l & 31, k = 8 (l >> 5) + 0..7)
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int cin, int cout,
                                                          const unsigned int* __restrict__ w_amax, int* __restrict__ exp_out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cin * cout) return;
    const int co = idx / cin, ci = idx % cin;
    const float bound = 2.25f * __builtin_bit_cast(float, w_amax[0]);
    const int e = scale_exp(__builtin_bit_cast(unsigned int, bound));
    if (idx == 0) exp_out[0] = e;
    double g[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) g[i][j] = (double)w[((size_t)co * cin + ci) * 9 + i * 3 + j];
    // rows of G: (1, 0, 0), (1/2, 1/2, 1/2), (1/2, -1/2, 1/2), (0, 0, 1)
    double t[4][3];
    for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5 * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5 * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
    const float sc = pow2f(e);
    const int CB = cout / 32, c = ci >> 4, k = ci & 15, cb = co >> 5, lane = (co & 31) + 32 * (k >> 3), j8 = k & 7;
    for (int a = 0; a < 4; ++a) {
        const double u[4] = {t[a][0], 0.5 * (t[a][0] + t[a][1] + t[a][2]), 0.5 * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
        for (int b = 0; b < 4; ++b) {
            const float x = (float)u[b] * sc;
            const _Float16 h0 = (_Float16)x;
            const _Float16 h1 = (_Float16)(x - (float)h0);
            const size_t blk = ((size_t)(c * 16 + a * 4 + b) * CB + cb) * 2;
            out[((blk + 0) * 64 + lane) * 8 + j8] = h0;
            out[((blk + 1) * 64 + lane) * 8 + j8] = h1;
        }
    }
}

struct WinoProblem {
    const float* in;
    const _Float16* wgt;
    const int* wexp;
    const float* bias;
    float* out;
    int cin, cout, H, W, relu;
    const unsigned int* in_amax;
    int tune;                    // ablation bits (ST_WINO_TUNE; wrong results, timing only): 1 no transform, 2 no MFMA, 4 no weight DMA,
                                 // 8 no raw patch
};

template <int TUNE>
__global__ __launch_bounds__(256) void wino_conv_kernel(WinoProblem p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* A = smem + kLdsA;                 // [position 16][co block 2][plane 2][1 KB]
    unsigned char* B = smem + kLdsB;                 // [position 16][plane 2][tile block 2][1 KB]
    float* raw = reinterpret_cast<float*>(smem + kLdsRaw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W;
    const int tiles_x = W >> 4;
    const int y0 = ((int)blockIdx.x / tiles_x) << 4, x0 = ((int)blockIdx.x % tiles_x) << 4;
    const int ct = blockIdx.y;                       // 64 output channels
    const int CB = p.cout >> 5;
    const int ea = p.wexp[0];
    // V = B^T d B grows an entry by at most 4 x max |d|
    const float vbound = 4.f * __builtin_bit_cast(float, amax_read(p.in_amax));
    const int eb = scale_exp(__builtin_bit_cast(unsigned int, vbound));
    const float sc = pow2f(eb);

    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    const int t = tid & 63, tyy = t >> 3, txx = t & 7, cg = tid >> 6;       // transform role: tile t, channels 4 cg .. 4 cg + 3
    const int coh = wave & 1, th = wave >> 1;                              // MFMA role: co half, tile half
    const int nchunks = p.cin >> 4;
    constexpr int kRawPerThread = (16 * 324 + 255) / 256;                  // 21
    float rawv[kRawPerThread];
    int raw_at[kRawPerThread];                                              // LDS float index, -1: nothing to write
    int raw_src[kRawPerThread];                                             // offset inside the chunk's 16 planes, -1: zero padding
#pragma unroll
    for (int i = 0; i < kRawPerThread; ++i) {
        const int e = tid + 256 * i;
        const int ch = e / 324, rem = e - ch * 324, r = rem / 18, col = rem - r * 18;
        const int gy = y0 - 1 + r, gx = x0 - 1 + col;
        raw_at[i] = e < 16 * 324 ? ch * kRawPlane + r * kRawPitch + col : -1;
        raw_src[i] = (e < 16 * 324 && gy >= 0 && gy < H && gx >= 0 && gx < W) ? (ch * H + gy) * W + gx : -1;
    }
    auto fetch_raw = [&](int chunk) __attribute__((always_inline)) {
        const float* plane0 = p.in + (size_t)chunk * 16 * H * W;
#pragma unroll
        for (int i = 0; i < kRawPerThread; ++i) rawv[i] = raw_src[i] >= 0 ? plane0[raw_src[i]] : 0.f;
    };
    auto store_raw = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < kRawPerThread; ++i)
            if (raw_at[i] >= 0) raw[raw_at[i]] = rawv[i];
    };
    for (int c = 0; c < nchunks; ++c) {
        // (a) transformed weights of the chunk: 64 pieces of 1 KB, wave w moves pieces 16 w .. 16 w + 15
        if (!(TUNE & 4))
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int q = wave * 16 + i, pos = q >> 2, cbl = (q >> 1) & 1, pl = q & 1;
            const _Float16* src = p.wgt + ((((size_t)(c * 16 + pos) * CB + (2 * ct + cbl)) * 2 + pl) * 64 + lane) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(A + q * 1024), 16, 0, 0);
        }
        // (b) the raw patch: 16 channels x 18 x 18 around the tile, zeros outside the image - every load of the thread in
        // flight before the first LDS write (the patch of the NEXT chunk is prefetched into these registers during (d))
        if (!(TUNE & 8)) {
            if (c == 0) fetch_raw(0);
            store_raw();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // (c) input transform of (tile t, 4 channels), planes written in B-operand order
        if (!(TUNE & 1)) {
            float V[4][16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* base = raw + (cg * 4 + j) * kRawPlane + (2 * tyy) * kRawPitch + 2 * txx;
                float d[4][4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const f32x2 lo = *reinterpret_cast<const f32x2*>(base + r * kRawPitch);
                    const f32x2 hi = *reinterpret_cast<const f32x2*>(base + r * kRawPitch + 2);
                    d[r][0] = lo[0]; d[r][1] = lo[1]; d[r][2] = hi[0]; d[r][3] = hi[1];
                }
                float u[4][4];                       // B^T d
#pragma unroll
                for (int col = 0; col < 4; ++col) {
                    u[0][col] = d[0][col] - d[2][col];
                    u[1][col] = d[1][col] + d[2][col];
                    u[2][col] = d[2][col] - d[1][col];
                    u[3][col] = d[1][col] - d[3][col];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {        // (B^T d) B
                    V[j][r * 4 + 0] = u[r][0] - u[r][2];
                    V[j][r * 4 + 1] = u[r][1] + u[r][2];
                    V[j][r * 4 + 2] = u[r][2] - u[r][1];
                    V[j][r * 4 + 3] = u[r][1] - u[r][3];
                }
            }
            unsigned char* dst = B + (t >> 5) * 1024 + ((t & 31) + 32 * (cg >> 1)) * 16 + (cg & 1) * 8;
#pragma unroll
            for (int pos = 0; pos < 16; ++pos) {
                f16x4 h0, h1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float x = V[j][pos] * sc;
                    const _Float16 a = (_Float16)x;
                    h0[j] = a;
                    h1[j] = (_Float16)(x - (float)a);
                }
                *reinterpret_cast<f16x4*>(dst + (pos * 2 + 0) * 2048) = h0;
                *reinterpret_cast<f16x4*>(dst + (pos * 2 + 1) * 2048) = h1;
            }
        }
        __syncthreads();
        if (c + 1 < nchunks && !(TUNE & 8)) fetch_raw(c + 1);       // lands during (d); written to LDS at the top of the next chunk
        // (d) 16 positions x (h0 g0 + h0 g1 + h1 g0)
        if (!(TUNE & 2))
#pragma unroll
        for (int pos = 0; pos < 16; ++pos) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(A + ((pos * 2 + coh) * 2 + 0) * 1024 + lane * 16);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(A + ((pos * 2 + coh) * 2 + 1) * 1024 + lane * 16);
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(B + ((pos * 2 + 0) * 2 + th) * 1024 + lane * 16);
            const f16x8 b1 = *reinterpret_cast<const f16x8*>(B + ((pos * 2 + 1) * 2 + th) * 1024 + lane * 16);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[pos], 0, 0, 0);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[pos], 0, 0, 0);
            acc[pos] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[pos], 0, 0, 0);
            if (pos & 1) __builtin_amdgcn_sched_barrier(0);      // (operands of two positions in flight: 32 registers, not 256)
        }
        __syncthreads();
    }

    // output transform on the accumulators: Y = A^T M A, A^T = (1 1 1 0; 0 1 -1 -1); then unscale, bias, ReLU
    const float unscale = pow2f(-(ea + eb));
    const int tile = th * 32 + (lane & 31), oy = y0 + 2 * (tile >> 3), ox = x0 + 2 * (tile & 7);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = ct * 64 + coh * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float s0[4], s1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s0[b] = (acc[0 + b][r] + acc[4 + b][r]) + acc[8 + b][r];
            s1[b] = (acc[4 + b][r] - acc[8 + b][r]) - acc[12 + b][r];
        }
        const float bias = p.bias ? p.bias[co] : 0.f;
        f32x2 row0, row1;
        row0[0] = ((s0[0] + s0[1]) + s0[2]) * unscale + bias;
        row0[1] = ((s0[1] - s0[2]) - s0[3]) * unscale + bias;
        row1[0] = ((s1[0] + s1[1]) + s1[2]) * unscale + bias;
        row1[1] = ((s1[1] - s1[2]) - s1[3]) * unscale + bias;
        if (p.relu) {
            row0[0] = fmaxf(row0[0], 0.f); row0[1] = fmaxf(row0[1], 0.f);
            row1[0] = fmaxf(row1[0], 0.f); row1[1] = fmaxf(row1[1], 0.f);
        }
        float* o = p.out + ((size_t)co * H + oy) * W + ox;
        *reinterpret_cast<f32x2*>(o) = row0;
        *reinterpret_cast<f32x2*>(o + W) = row1;
        __builtin_amdgcn_sched_barrier(0);           // (one row's 16 accumulator reads live at a time)
    }
}

}  // namespace

size_t winograd_weight_bytes(int cin, int cout) { return (size_t)16 * cin * cout * 2 * sizeof(_Float16) + 256; }

bool winograd_applies(int cin, int cout, int height, int width) {
    return cin % 16 == 0 && cout % 64 == 0 && height % 16 == 0 && width % 16 == 0 && height >= 16 && width >= 16;
}

// w: torch [Cout][Cin][3][3]; out: winograd_weight_bytes(); the scale exponent lands in the trailer (int at the end)
int launch_winograd_weights(const float* w, void* out, int cin, int cout, hipStream_t s) {
    unsigned int* trailer = reinterpret_cast<unsigned int*>(static_cast<unsigned char*>(out) + (size_t)16 * cin * cout * 2 * sizeof(_Float16));
    ST_HIP(hipMemsetAsync(trailer, 0, 256, s));
    if (launch_amax(w, (long long)cin * cout * 9, trailer, 1, s)) return 1;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((cin * cout + 255) / 256), dim3(256), 0, s, w, static_cast<_Float16*>(out), cin, cout,
                       trailer, reinterpret_cast<int*>(trailer + 8));
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_conv_winograd(const float* in, const void* wino, const float* bias, float* out, int cin, int cout, int height, int width,
                         int relu, const unsigned int* in_amax, hipStream_t s) {
    ST_REQUIRE(winograd_applies(cin, cout, height, width), "winograd conv: Cin %% 16, Cout %% 64, H %% 16, W %% 16 (got %d %d %d %d)", cin,
               cout, height, width);
    WinoProblem p{};
    p.in = in; p.wgt = static_cast<const _Float16*>(wino); p.bias = bias; p.out = out;
    p.wexp = reinterpret_cast<const int*>(static_cast<const unsigned char*>(wino) + (size_t)16 * cin * cout * 2 * sizeof(_Float16)) + 8;
    p.cin = cin; p.cout = cout; p.H = height; p.W = width; p.relu = relu; p.in_amax = in_amax;
    static Option tune("ST_WINO_TUNE", 0);
    p.tune = tune.get();
    const dim3 grid((height / 16) * (width / 16), cout / 64);
#define ST_WINO_CASE(T)                                                                                                              \
    case T: {                                                                                                                        \
        static bool attr = false;                                                                                                    \
        if (!attr) {                                                                                                                 \
            ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       kLdsBytes));                                                                                  \
            attr = true;                                                                                                             \
        }                                                                                                                            \
        hipLaunchKernelGGL(wino_conv_kernel<T>, grid, dim3(256), kLdsBytes, s, p);                                                   \
        break;                                                                                                                       \
    }
    switch (p.tune) {           // (ablation variants are separate kernels: a run-time switch changed the code of the full kernel)
        ST_WINO_CASE(1) ST_WINO_CASE(2) ST_WINO_CASE(3) ST_WINO_CASE(4) ST_WINO_CASE(8) ST_WINO_CASE(12) ST_WINO_CASE(13)
        ST_WINO_CASE(14) ST_WINO_CASE(15)
        default: ST_WINO_CASE(0)
    }
#undef ST_WINO_CASE
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
