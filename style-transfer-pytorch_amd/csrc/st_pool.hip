// 2x2 / stride-2 pooling (floor mode: a trailing odd row / column is dropped) and its backward.
// Replaces nn.MaxPool2d(2, 2) at vgg19.features[4, 9, 18, 27] and the reference's substitutes
// Scale(AvgPool2d(2), 2.0) / Scale(LPPool2d(2, 2), 0.78) (style_transfer.py:21-22,41-46).
// HBM-bound; one thread per pooled output, lanes along x.
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

}  // namespace

int launch_pool_fwd(const float* in, float* out, int channels, int height, int width, int mode,
                    hipStream_t s) {
    const long long total = (long long)channels * (height / 2) * (width / 2);
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(pool_fwd_kernel, dim3(blocks), dim3(256), 0, s, in, out, channels, height, width, mode);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_pool_bwd(const float* in, const float* grad_out, float* grad_in, int channels, int height,
                    int width, int mode, hipStream_t s) {
    const long long total = (long long)channels * height * width;
    const int blocks = (int)std::min<long long>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(pool_bwd_kernel, dim3(blocks), dim3(256), 0, s, in, grad_out, grad_in, channels,
                       height, width, mode);
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
