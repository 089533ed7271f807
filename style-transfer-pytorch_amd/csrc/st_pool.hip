// 2x2 / stride-2 pooling (floor mode: a trailing odd row / column is dropped) and its backward.
// Replaces nn.MaxPool2d(2, 2) at vgg19.features[4, 9, 18, 27] and the reference's substitutes
// Scale(AvgPool2d(2), 2.0) / Scale(LPPool2d(2, 2), 0.78) (style_transfer.py:21-22,41-46).
// HBM-bound.  When W % 4 == 0 (and H even) a thread owns two adjacent windows: two 16-byte loads of the
// input rows, one 8-byte access of the pooled map - every input element is read exactly once (the scalar
// kernels, kept for odd sizes, re-read each window per output element in the backward pass).
#include "st_common.h"

namespace st {
namespace {

constexpr float kAvgScale = 2.0f;    // pooling_scales['average']
constexpr float kL2Scale = 0.78f;    // pooling_scales['l2']

__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                       int C, int H, int W, int mode) {
#pragma clang fp contract(off)
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)C * Ho * Wo;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xo = (int)(i % Wo);
        const int yo = (int)((i / Wo) % Ho);
        const int c = (int)(i / ((long long)Wo * Ho));
        const float* p = in + ((size_t)c * H + 2 * yo) * W + 2 * xo;
        const float a = p[0], b = p[1], d = p[W], e = p[W + 1];
        float r;
        if (mode == 0) {
            r = a;                       // first maximum in row-major window order wins
            if (b > r) r = b;
            if (d > r) r = d;
            if (e > r) r = e;
        } else if (mode == 1) {
            r = ((a + b + d + e) / 4.f) * kAvgScale;
        } else {
            // LPPool2d(2): avg_pool(x^2) -> sign * relu(abs) -> * 4 -> ^0.5, then the 0.78 Scale
            const float m = (a * a + b * b + d * d + e * e) / 4.f;
            r = sqrtf(m * 4.f) * kL2Scale;
        }
        out[i] = r;
    }
}

__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ in,
                                                       const float* __restrict__ gout,
                                                       float* __restrict__ gin, int C, int H, int W,
                                                       int mode) {
#pragma clang fp contract(off)
    const int Ho = H / 2, Wo = W / 2;
    const long long total = (long long)C * H * W;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x = (int)(i % W);
        const int y = (int)((i / W) % H);
        const int c = (int)(i / ((long long)W * H));
        const int xo = x >> 1, yo = y >> 1;
        float g = 0.f;
        if (xo < Wo && yo < Ho) {
            const float go = gout[((size_t)c * Ho + yo) * Wo + xo];
            const float* p = in + ((size_t)c * H + 2 * yo) * W + 2 * xo;
            if (mode == 0) {
                const float a = p[0], b = p[1], d = p[W], e = p[W + 1];
                int arg = 0;
                float m = a;
                if (b > m) { m = b; arg = 1; }
                if (d > m) { m = d; arg = 2; }
                if (e > m) { m = e; arg = 3; }
                g = (arg == ((y & 1) * 2 + (x & 1))) ? go : 0.f;
            } else if (mode == 1) {
                g = (go * kAvgScale) / 4.f;
            } else {
                const float a = p[0], b = p[1], d = p[W], e = p[W + 1];
                const float m = (a * a + b * b + d * d + e * e) / 4.f;
                const float root = sqrtf(m * 4.f);
                // d/dx_i [0.78 * sqrt(sum x^2)] = 0.78 * x_i / sqrt(sum x^2); an all-zero window gives
                // 0/0 here exactly as in autograd - the following ReLU mask (select, not multiply)
                // discards it because every x_i in such a window is a zero ReLU output.
                g = (go * kL2Scale) * (in[i] / root);
            }
        }
        // threshold_backward of the convolution that produced `in` (its data-gradient conv then needs no mask;
        // a select, so the L2 variant's 0/0 is discarded here)
        gin[i] = (in[i] > 0.f) ? g : 0.f;
    }
}

// ---- vector paths (W % 4 == 0, H % 2 == 0) ----
__device__ __forceinline__ float pool_window(float a, float b, float d, float e, int mode) {
#pragma clang fp contract(off)
    if (mode == 0) {
        float r = a;                     // first maximum in row-major window order wins
        if (b > r) r = b;
        if (d > r) r = d;
        if (e > r) r = e;
        return r;
    }
    if (mode == 1) return ((a + b + d + e) / 4.f) * kAvgScale;
    const float m = (a * a + b * b + d * d + e * e) / 4.f;
    return sqrtf(m * 4.f) * kL2Scale;
}

__global__ __launch_bounds__(256) void pool_fwd4_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        int C, int H, int W, int mode) {
    const int Ho = H / 2, W4 = W / 4;
    const long long total = (long long)C * Ho * W4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x4 = (int)(i % W4);
        const long long row = i / W4;                       // c * Ho + yo
        const int yo = (int)(row % Ho);
        const long long c = row / Ho;
        const float* p = in + ((size_t)c * H + 2 * yo) * W + 4 * x4;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(p), r1 = *reinterpret_cast<const f32x4*>(p + W);
        f32x2 o;
        o[0] = pool_window(r0[0], r0[1], r1[0], r1[1], mode);
        o[1] = pool_window(r0[2], r0[3], r1[2], r1[3], mode);
        *reinterpret_cast<f32x2*>(out + (size_t)row * (W / 2) + 2 * x4) = o;
    }
}

// gradient of one window to its four inputs (a, b / d, e), already masked by (input > 0)
__device__ __forceinline__ void pool_window_bwd(float a, float b, float d, float e, float go, int mode, float& ga,
                                                float& gb, float& gd, float& ge) {
#pragma clang fp contract(off)
    if (mode == 0) {
        int arg = 0;
        float m = a;
        if (b > m) { m = b; arg = 1; }
        if (d > m) { m = d; arg = 2; }
        if (e > m) { m = e; arg = 3; }
        ga = arg == 0 ? go : 0.f; gb = arg == 1 ? go : 0.f; gd = arg == 2 ? go : 0.f; ge = arg == 3 ? go : 0.f;
    } else if (mode == 1) {
        ga = gb = gd = ge = (go * kAvgScale) / 4.f;
    } else {
        const float m = (a * a + b * b + d * d + e * e) / 4.f;
        const float root = sqrtf(m * 4.f);
        const float s = go * kL2Scale;
        ga = s * (a / root); gb = s * (b / root); gd = s * (d / root); ge = s * (e / root);
    }
    ga = a > 0.f ? ga : 0.f; gb = b > 0.f ? gb : 0.f; gd = d > 0.f ? gd : 0.f; ge = e > 0.f ? ge : 0.f;
}

__global__ __launch_bounds__(256) void pool_bwd4_kernel(const float* __restrict__ in,
                                                        const float* __restrict__ gout, float* __restrict__ gin,
                                                        int C, int H, int W, int mode) {
    const int Ho = H / 2, W4 = W / 4;
    const long long total = (long long)C * Ho * W4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x4 = (int)(i % W4);
        const long long row = i / W4;
        const int yo = (int)(row % Ho);
        const long long c = row / Ho;
        const size_t base = ((size_t)c * H + 2 * yo) * W + 4 * x4;
        const f32x4 r0 = *reinterpret_cast<const f32x4*>(in + base), r1 = *reinterpret_cast<const f32x4*>(in + base + W);
        const f32x2 go = *reinterpret_cast<const f32x2*>(gout + (size_t)row * (W / 2) + 2 * x4);
        f32x4 g0, g1;
        float ga, gb, gd, ge;
        pool_window_bwd(r0[0], r0[1], r1[0], r1[1], go[0], mode, ga, gb, gd, ge);
        g0[0] = ga; g0[1] = gb; g1[0] = gd; g1[1] = ge;
        pool_window_bwd(r0[2], r0[3], r1[2], r1[3], go[1], mode, ga, gb, gd, ge);
        g0[2] = ga; g0[3] = gb; g1[2] = gd; g1[3] = ge;
        *reinterpret_cast<f32x4*>(gin + base) = g0;
        *reinterpret_cast<f32x4*>(gin + base + W) = g1;
    }
}

// Max-pool backward from the argmax codes the producing convolution's epilogue left (ConvProblem::pool_code: bits 0-1 =
// first maximum of the window in row-major order, bit 2 = maximum > 0, i.e. the ReLU mask of the conv that feeds the
// pool): reads one byte + one float per window, writes the full-resolution gradient - the saved map is not read.
// A thread owns two adjacent windows (W % 4 == 0, H even).
__global__ __launch_bounds__(256) void pool_bwd_codes_kernel(const unsigned char* __restrict__ code,
                                                             const float* __restrict__ gout, float* __restrict__ gin,
                                                             int C, int H, int W) {
    const int Ho = H / 2, W4 = W / 4;
    const long long total = (long long)C * Ho * W4;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int x4 = (int)(i % W4);
        const long long row = i / W4;                         // (channel, pooled row)
        const int yo = (int)(row % Ho);
        const long long c = row / Ho;
        const size_t base = ((size_t)c * H + 2 * yo) * W + 4 * x4;
        const size_t pooled = (size_t)row * (W / 2) + 2 * x4;
        const unsigned short two = *reinterpret_cast<const unsigned short*>(code + pooled);
        const f32x2 go = *reinterpret_cast<const f32x2*>(gout + pooled);
        f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const int cd = (two >> (8 * w)) & 0xff;
            const float g = (cd & 4) ? go[w] : 0.f;
            const int at = cd & 3;
            g0[2 * w] = at == 0 ? g : 0.f;
            g0[2 * w + 1] = at == 1 ? g : 0.f;
            g1[2 * w] = at == 2 ? g : 0.f;
            g1[2 * w + 1] = at == 3 ? g : 0.f;
        }
        *reinterpret_cast<f32x4*>(gin + base) = g0;
        *reinterpret_cast<f32x4*>(gin + base + W) = g1;
    }
}

}  // namespace

int launch_pool_bwd_codes(const unsigned char* code, const float* grad_out, float* grad_in, int channels, int height,
                          int width, hipStream_t s) {
    ST_REQUIRE(width % 4 == 0 && height % 2 == 0, "pool backward from codes: W %% 4 == 0 and even H required");
    const long long total4 = (long long)channels * (height / 2) * (width / 4);
    const int blocks4 = (int)std::min<long long>((total4 + 255) / 256, 16384);
    hipLaunchKernelGGL(pool_bwd_codes_kernel, dim3(blocks4), dim3(256), 0, s, code, grad_out, grad_in, channels, height, width);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_pool_fwd(const float* in, float* out, int channels, int height, int width, int mode,
                    hipStream_t s) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    if (width % 4 == 0 && aligned) {
        const long long total4 = (long long)channels * (height / 2) * (width / 4);
        const int blocks4 = (int)std::min<long long>((total4 + 255) / 256, 16384);
        hipLaunchKernelGGL(pool_fwd4_kernel, dim3(blocks4), dim3(256), 0, s, in, out, channels, height, width, mode);
        ST_LAUNCH_CHECK();
        return 0;
    }
    const long long total = (long long)channels * (height / 2) * (width / 2);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(blocks), dim3(256), 0, s, in, out, channels, height, width, mode);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_pool_bwd(const float* in, const float* grad_out, float* grad_in, int channels, int height,
                    int width, int mode, hipStream_t s) {
    const bool aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(grad_out) |
                           reinterpret_cast<uintptr_t>(grad_in)) & 15) == 0;
    if (width % 4 == 0 && height % 2 == 0 && aligned) {
        const long long total4 = (long long)channels * (height / 2) * (width / 4);
        const int blocks4 = (int)std::min<long long>((total4 + 255) / 256, 16384);
        hipLaunchKernelGGL(pool_bwd4_kernel, dim3(blocks4), dim3(256), 0, s, in, grad_out, grad_in, channels, height,
                           width, mode);
        ST_LAUNCH_CHECK();
        return 0;
    }
    const long long total = (long long)channels * height * width;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(blocks), dim3(256), 0, s, in, grad_out, grad_in, channels,
                       height, width, mode);
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
