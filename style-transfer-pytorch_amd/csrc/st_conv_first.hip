// conv1_1 (3 -> 64 channels, K = 27): HBM-bound, so a direct VALU kernel - no GEMM reshaping.
//
// Forward replaces transforms.Normalize (style_transfer.py:30-31,85) + the replicate-padded
// nn.Conv2d(3, 64, 3, padding=1, padding_mode='replicate') (:39,52-59) + ReLU (features[1]).
// Backward replaces threshold_backward + convolution_backward(input) + the backward of the
// replicate pad (gradient of the padding ring folds onto the nearest edge pixel) + Normalize's 1/std.
//
// Traffic at H x W: forward reads 3HW, writes 64HW floats; backward reads 2 x 64HW, writes 3HW.
// One thread = one pixel; lanes walk x, so each per-channel store/load is a contiguous 256 B row
// segment per wave.  Weights are wave-uniform and come through the scalar cache.
#include "st_common.h"

namespace st {
namespace {

__constant__ float kMean[3] = {0.485f, 0.456f, 0.406f};
__constant__ float kStd[3] = {0.229f, 0.224f, 0.225f};

// Strip sharding: `halo` = [2][3][W] (neighbour's last row, neighbour's first row) or nullptr;
// without a neighbour the row index is clamped (replicate padding at the GLOBAL border only).
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const float* __restrict__ image,
                                                             const float* __restrict__ w,
                                                             const float* __restrict__ b,
                                                             float* __restrict__ out, int H, int W,
                                                             const float* __restrict__ halo, int has_up,
                                                             int has_down, unsigned int* out_amax) {
    const int HW = H * W;
    // threads past the end redo the last pixel (identical stores) so that whole waves reach amax_commit
    const int pix = min((int)(blockIdx.x * 256 + threadIdx.x), HW - 1);
    const int y = pix / W, x = pix % W;
    unsigned int amax = 0;
    float v[27];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yr = y + ky - 1;
            const int yy = min(max(yr, 0), H - 1);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = min(max(x + kx - 1, 0), W - 1);
                float raw;
                if (yr < 0 && has_up) raw = halo[c * W + xx];
                else if (yr >= H && has_down) raw = halo[(3 + c) * W + xx];
                else raw = image[(size_t)c * HW + yy * W + xx];
                // Normalize first (true division, like the reference), then the replicate pad sees
                // normalised values - identical to padding then normalising.
                v[(c * 3 + ky) * 3 + kx] = (raw - kMean[c]) / kStd[c];
            }
        }
    }
    for (int co = 0; co < 64; ++co) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 27; ++k) acc = fmaf(w[co * 27 + k], v[k], acc);
        acc += b[co];
        acc = fmaxf(acc, 0.f);
        out[(size_t)co * HW + pix] = acc;
        amax = max(amax, abs_bits(acc));
    }
    if (out_amax) amax_commit(amax, out_amax);
}

// Data gradient.  With P = replicate_pad(xhat) and out[o] = sum_k w[k] P[o + k - 1]:
//   dP[p] = sum_k w[k] g[p - k + 1]  (g zero outside the image),  dxhat[y] = sum_{p : clamp(p) = y} dP[p].
// Two kernels: dP on the PADDED domain (columns -1..W, rows -1..H where the strip touches the global border) with
// the same branch-free stencil everywhere, then a fold of the padding ring onto the border pixels (+ 1/std,
// + accumulate).  (One kernel with a per-pixel border path was 4x slower: the runtime-loop ring code made every
// workgroup on the image perimeter ~15x slower than the rest, and the launch waited for them.)
constexpr int FT = 16;   // 16x16 tile of the padded domain per workgroup
constexpr int FC = 8;    // output channels of conv1_1 staged per pass

constexpr int FE = FC * (FT + 2) * (FT + 2);      // staged elements per pass
constexpr int FN = (FE + 255) / 256;               // per thread

__device__ __forceinline__ float first_buffer_load(__amdgpu_buffer_rsrc_t rsrc, int byte_offset) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_offset, 0, 0));
}

__global__ __launch_bounds__(256) void conv_first_dp_kernel(const float* __restrict__ gout,
                                                            const float* __restrict__ yrelu,
                                                            const float* __restrict__ w, float* __restrict__ dp,
                                                            int H, int W, const float* __restrict__ ghalo,
                                                            int has_up, int has_down) {
    // ghalo: [2][64][W] rows -1 / H of the (already masked) gradient from the strip neighbours
    __shared__ float tile[2][FC][FT + 2][FT + 2];
    const int HW = H * W;
    const int py0 = has_up ? 0 : -1, py1 = has_down ? H - 1 : H;          // domain rows (ring only at global borders)
    const int tiles_x = (W + 2 + FT - 1) / FT;
    const int x0 = -1 + (blockIdx.x % tiles_x) * FT, y0 = py0 + (blockIdx.x / tiles_x) * FT;
    const int tx = threadIdx.x % FT, ty = threadIdx.x / FT;
    const int x = x0 + tx, y = y0 + ty;
    const bool active = (x <= W) && (y <= py1);

    // staging map, identical for every pass: byte offset inside an FC-channel slab, or out of range
    // (the buffer load then returns 0 = the zero gradient outside the image)
    int goff[FN], hoff[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int e = threadIdx.x + i * 256;
        const int c = e / ((FT + 2) * (FT + 2)), rem = e % ((FT + 2) * (FT + 2));
        const int yy = y0 - 1 + rem / (FT + 2), xx = x0 - 1 + rem % (FT + 2);
        const bool ok = e < FE && yy >= 0 && yy < H && xx >= 0 && xx < W;
        goff[i] = ok ? (c * HW + yy * W + xx) * 4 : 0x40000000;
        const bool xin = e < FE && xx >= 0 && xx < W && ghalo != nullptr;
        hoff[i] = (xin && yy == -1 && has_up) ? (c * W + xx) * 4
                  : ((xin && yy == H && has_down) ? ((64 + c) * W + xx) * 4 : 0x40000000);
    }
    float rg[FN], ry[FN], rh[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) rh[i] = 0.f;
    auto load_pass = [&](int cb) {
        const __amdgpu_buffer_rsrc_t gs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(gout) + (size_t)cb * HW, 0, FC * HW * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(yrelu ? yrelu : gout) + (size_t)cb * HW, 0, FC * HW * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            rg[i] = first_buffer_load(gs, goff[i]);
            ry[i] = yrelu ? first_buffer_load(ys, goff[i]) : 1.f;     // nullptr: gout is already masked
        }
        if (ghalo != nullptr) {
            const __amdgpu_buffer_rsrc_t hs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(ghalo) + (size_t)cb * W, 0, (64 + FC) * W * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < FN; ++i) rh[i] = first_buffer_load(hs, hoff[i]);
        }
    };
    auto store_pass = [&](int buf) {
        float* t = &tile[buf][0][0][0];
#pragma unroll
        for (int i = 0; i < FN; ++i) {
            const int e = threadIdx.x + i * 256;
            if (e < FE) t[e] = ((ry[i] > 0.f) ? rg[i] : 0.f) + rh[i];  // threshold_backward (+ pre-masked halo)
        }
    };

    float acc[3] = {0.f, 0.f, 0.f};
    load_pass(0);
    store_pass(0);
    __syncthreads();
    for (int cb = 0, pass = 0; cb < 64; cb += FC, ++pass) {
        const int buf = pass & 1;
        const bool more = cb + FC < 64;
        if (more) load_pass(cb + FC);
        // exactly one tap links each of the 9 neighbouring outputs to this padded position
#pragma unroll
        for (int c = 0; c < FC; ++c) {
            const float* wc = w + (cb + c) * 27;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const float g = tile[buf][c][ty + 1 + dy][tx + 1 + dx];
                    const int k = (1 - dy) * 3 + (1 - dx);
                    acc[0] = fmaf(wc[k], g, acc[0]);
                    acc[1] = fmaf(wc[9 + k], g, acc[1]);
                    acc[2] = fmaf(wc[18 + k], g, acc[2]);
                }
        }
        if (more) store_pass(buf ^ 1);
        __syncthreads();
    }
    if (active) {
        const size_t plane = (size_t)(py1 - py0 + 1) * (W + 2);
#pragma unroll
        for (int c = 0; c < 3; ++c) dp[c * plane + (size_t)(y - py0) * (W + 2) + (x + 1)] = acc[c];
    }
}

// grad_image[c][y][x] (+)= (sum of dP over the padded positions that replicate (y, x)) / std[c]
__global__ __launch_bounds__(256) void conv_first_fold_kernel(const float* __restrict__ dp, float* __restrict__ gimg,
                                                              int H, int W, int accumulate, int has_up,
                                                              int has_down) {
    const int py0 = has_up ? 0 : -1, py1 = has_down ? H - 1 : H;
    const size_t plane = (size_t)(py1 - py0 + 1) * (W + 2);
    const int HW = H * W;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= HW) return;
    const int y = pix / W, x = pix % W;
    const int ys = (y == 0 && !has_up) ? -1 : y, ye = (y == H - 1 && !has_down) ? H : y;
    const int xs = (x == 0) ? -1 : x, xe = (x == W - 1) ? W : x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float sum = 0.f;
        for (int yy = ys; yy <= ye; ++yy)
            for (int xx = xs; xx <= xe; ++xx) sum += dp[c * plane + (size_t)(yy - py0) * (W + 2) + (xx + 1)];
        float v = sum / kStd[c];                                   // backward of Normalize
        if (accumulate) v += gimg[(size_t)c * HW + pix];
        gimg[(size_t)c * HW + pix] = v;
    }
}

}  // namespace

int launch_conv_first_fwd(const float* image, const float* w, const float* b, float* out, int height,
                          int width, hipStream_t stream, const float* halo, int has_up, int has_down,
                          unsigned int* out_amax) {
    const int blocks = ceil_div(height * width, 256);
    hipLaunchKernelGGL(conv_first_fwd_kernel, dim3(blocks), dim3(256), 0, stream, image, w, b, out, height,
                       width, halo, has_up, has_down, out_amax);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_conv_first_dgrad(const float* grad_out, const float* relu_out, const float* w, float* grad_image,
                            float* dp_scratch, int height, int width, int accumulate, hipStream_t stream,
                            const float* ghalo, int has_up, int has_down) {
    const int rows = height + (has_up ? 0 : 1) + (has_down ? 0 : 1);
    const int blocks = ceil_div(width + 2, FT) * ceil_div(rows, FT);
    hipLaunchKernelGGL(conv_first_dp_kernel, dim3(blocks), dim3(256), 0, stream, grad_out, relu_out, w, dp_scratch,
                       height, width, ghalo, has_up, has_down);
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv_first_fold_kernel, dim3(ceil_div(height * width, 256)), dim3(256), 0, stream, dp_scratch,
                       grad_image, height, width, accumulate, has_up, has_down);
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
