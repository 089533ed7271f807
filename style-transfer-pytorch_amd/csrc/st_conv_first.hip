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

// Forward in the trunk's fp16x3 arithmetic (the mode in which the VALU kernel above is compute-bound: 1728 FMAs per
// pixel at a quarter of the vector rate, 385 us at 2048^2 against a 160 - 215 us store floor).  Per 32 pixels the
// 64 x 27 contraction is 12 v_mfma_f32_32x32x16_f16 (K padded to 32; two fp16 planes per operand, three products).
//   A (weights, [co][k]):  split once per wave into registers, power-of-two scale from max |w| (wave reduction).
//   B (im2col of the normalised, replicate-padded image): the workgroup stages its patch - 4 rows x 128 columns plus
//     the 1-pixel ring, 3 channels, normalised fp32 - in LDS; a lane gathers its 16 K values of a pixel with 16
//     ds_read_b32 at precomputed patch offsets (K indices >= 27 point at a zero word) and splits them (|x| < 2.7: fixed
//     scale 2^11).
//   A wave owns one image row of the patch: 128 consecutive pixels as FOUR 32-pixel blocks interleaved mod 4 (block j
//   = pixels 4 n + j), so that the four accumulators of a lane are 4 consecutive pixels of one output channel: 16-byte
//   stores, 512 contiguous bytes per half-wave and channel.
typedef _Float16 cf_f16x8 __attribute__((ext_vector_type(8)));
constexpr int MR = 4, MC = 128;                       // patch interior: rows x columns (one wave per row)
constexpr int MPR = MR + 2, MPC = MC + 2;             // with the ring
constexpr int MPLANE = MPR * MPC;                     // floats per channel
constexpr int MZERO = 3 * MPLANE;                     // index of the zero word

__global__ __launch_bounds__(256, 2) void conv_first_fwd_mfma_kernel(const float* __restrict__ image,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ b,
                                                                     float* __restrict__ out, int H, int W,
                                                                     const float* __restrict__ halo, int has_up,
                                                                     int has_down, unsigned int* out_amax) {
    __shared__ float patch[3 * MPLANE + 4];
    __shared__ __attribute__((aligned(16))) float bias_s[64];
    const int HW = H * W;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int tiles_x = (W + MC - 1) / MC;
    const int x0 = (blockIdx.x % tiles_x) * MC, y0 = (blockIdx.x / tiles_x) * MR;

    // ---- stage the normalised patch (replicate padding = clamped indices; strip plans: rows -1 / H from the halo) ----
    for (int e = tid; e < 3 * MPLANE; e += 256) {
        const int c = e / MPLANE, r = (e % MPLANE) / MPC, q = e % MPC;
        const int yr = y0 + r - 1;
        const int yy = min(max(yr, 0), H - 1), xx = min(max(x0 + q - 1, 0), W - 1);
        float raw;
        if (yr < 0 && has_up) raw = halo[c * W + xx];
        else if (yr >= H && has_down) raw = halo[(3 + c) * W + xx];
        else raw = image[(size_t)c * HW + yy * W + xx];
        patch[e] = (raw - kMean[c]) / kStd[c];            // true division, like transforms.Normalize
    }
    if (tid < 4) patch[MZERO + tid] = 0.f;
    if (tid < 64) bias_s[tid] = b[tid];

    // ---- weights: A operands a[m][ks][plane], rows co = 32 m + l31, K = 16 ks + 8 half + e ----
    float wv[2][2][8];
    float wmax = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = 16 * ks + 8 * half + e;
                const float v = kk < 27 ? w[(32 * m + l31) * 27 + kk] : 0.f;
                wv[m][ks][e] = v;
                wmax = fmaxf(wmax, fabsf(v));
            }
    unsigned int wbits = __builtin_bit_cast(unsigned int, wmax);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned int o = (unsigned int)__shfl_xor((int)wbits, off);
        wbits = o > wbits ? o : wbits;
    }
    const int ew = scale_exp(wbits);
    constexpr int ex = 11;                                // |normalised pixel| < 2.7 -> < 2^13 after scaling
    const float sw = pow2f(ew), sx = pow2f(ex), unscale = pow2f(-ew - ex);
    cf_f16x8 a0[2][2], a1[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = wv[m][ks][e] * sw;
                const _Float16 h = (_Float16)v;
                a0[m][ks][e] = h;
                a1[m][ks][e] = (_Float16)(v - (float)h);
            }
    // ---- patch offsets of this lane's 16 K values, relative to its pixel (row r = wave, column 4 l31 + j) ----
    int koff[2][8];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = 16 * ks + 8 * half + e;
            const int c = kk / 9, ky = (kk % 9) / 3, kx = kk % 3;
            koff[ks][e] = kk < 27 ? c * MPLANE + ky * MPC + kx : -1;
        }
    __syncthreads();

    const int y = y0 + wave;
    f32x16 acc[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // pixel (row wave + 1, column 4 l31 + j + 1) of the patch; tap (ky, kx) is at (+ky - 1, +kx - 1)
        const int base = wave * MPC + 4 * l31 + j;
        cf_f16x8 b0[2], b1[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = koff[ks][e] >= 0 ? base + koff[ks][e] : MZERO;
                const float v = patch[idx] * sx;
                const _Float16 h = (_Float16)v;
                b0[ks][e] = h;
                b1[ks][e] = (_Float16)(v - (float)h);
            }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[m][ks], b1[ks], acc[m][j], 0, 0, 0);
                acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[m][ks], b0[ks], acc[m][j], 0, 0, 0);
                acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[m][ks], b0[ks], acc[m][j], 0, 0, 0);
            }
        }
    }
    // ---- epilogue: bias, ReLU, 4 consecutive pixels per store ----
    unsigned int amax = 0;
    const int x = x0 + 4 * l31;
    const bool row_ok = y < H;
    const bool vec_ok = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float bv = bias_s[co];
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc[m][j][r] * unscale + bv, 0.f);
            float* dst = out + (size_t)co * HW + (size_t)y * W + x;
            if (row_ok && vec_ok && x + 3 < W) {
                *reinterpret_cast<f32x4*>(dst) = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) amax = max(amax, abs_bits(v[j]));
            } else if (row_ok) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (x + j < W) {
                        dst[j] = v[j];
                        amax = max(amax, abs_bits(v[j]));
                    }
            }
        }
    if (out_amax) amax_commit(amax, out_amax);
}

// Data gradient.  With P = replicate_pad(xhat) and out[o] = sum_k w[k] P[o + k - 1]:
//   dP[p] = sum_k w[k] g[p - k + 1]  (g zero outside the image),  dxhat[y] = sum_{p : clamp(p) = y} dP[p].
// Two kernels: dP on the PADDED domain (columns -1..W, rows -1..H where the strip touches the global border) with
// the same branch-free stencil everywhere, then a fold of the padding ring onto the border pixels (+ 1/std,
// + accumulate).  (One kernel with a per-pixel border path was 4x slower: the runtime-loop ring code made every
// workgroup on the image perimeter ~15x slower than the rest, and the launch waited for them.)
// Tile: 64 x 16 positions of the padded domain per workgroup, 4 consecutive x per thread: the 3 x 6 window of a
// channel is one 16-byte and one 8-byte LDS read per row for 108 FMAs (one ds_read_b32 per 3 FMAs, the first
// version, was LDS-instruction bound at 92 us).
constexpr int FTX = 64, FTY = 16;
constexpr int FC = 8;                              // output channels of conv1_1 staged per pass
constexpr int FROWS = FTY + 2, FCOLS = FTX + 2;
constexpr int FPITCH = 68;                         // floats per staged row (16-byte aligned rows)
constexpr int FSLAB = FROWS * FCOLS;               // staged elements of one channel
constexpr int FN = (FSLAB + 255) / 256;            // per thread and channel

__device__ __forceinline__ float first_buffer_load(__amdgpu_buffer_rsrc_t rsrc, int byte_offset, int soffset) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_offset, soffset, 0));
}

// MASK: apply (yrelu > 0) while staging (false: gout is already masked by its producer).  HALO: strip sharding,
// rows -1 / H come from ghalo [2][64][W] (already masked by the neighbours).
template <bool MASK, bool HALO>
__global__ __launch_bounds__(256) void conv_first_dp_kernel(const float* __restrict__ gout,
                                                            const float* __restrict__ yrelu,
                                                            const float* __restrict__ w, float* __restrict__ dp,
                                                            int H, int W, const float* __restrict__ ghalo,
                                                            int has_up, int has_down) {
    __shared__ __attribute__((aligned(16))) float tile[2][FC][FROWS][FPITCH];
    const int HW = H * W;
    const int py0 = has_up ? 0 : -1, py1 = has_down ? H - 1 : H;          // domain rows (ring only at global borders)
    const int tiles_x = (W + 2 + FTX - 1) / FTX;
    const int x0 = -1 + (blockIdx.x % tiles_x) * FTX, y0 = py0 + (blockIdx.x / tiles_x) * FTY;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    const int x = x0 + 4 * tx, y = y0 + ty;                               // first of this thread's 4 positions

    // staging map of ONE channel slab (the channel enters through the scalar offset / an immediate): byte offset
    // in the image plane, or out of range (the buffer load then returns 0 = the zero gradient outside the image)
    int goff[FN], hoff[HALO ? FN : 1], loff[FN];
#pragma unroll
    for (int i = 0; i < FN; ++i) {
        const int e0 = threadIdx.x + i * 256;
        const int e = e0 < FSLAB ? e0 : FSLAB - 1;                        // surplus lanes redo the last element
        const int r = e / FCOLS, q = e % FCOLS;
        const int yy = y0 - 1 + r, xx = x0 - 1 + q;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        goff[i] = ok ? (yy * W + xx) * 4 : 0x40000000;
        if constexpr (HALO) {
            const bool xin = xx >= 0 && xx < W;
            hoff[i] = (xin && yy == -1 && has_up) ? xx * 4 : ((xin && yy == H && has_down) ? (64 * W + xx) * 4 : 0x40000000);
        }
        loff[i] = r * FPITCH + q;
    }
    float rg[FC][FN], ry[MASK ? FC : 1][FN], rh[HALO ? FC : 1][FN];
    auto load_pass = [&](int cb) {
        const __amdgpu_buffer_rsrc_t gs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(gout) + (size_t)cb * HW, 0, FC * HW * 4, 0x00020000);
#pragma unroll
        for (int c = 0; c < FC; ++c)
#pragma unroll
            for (int i = 0; i < FN; ++i) rg[c][i] = first_buffer_load(gs, goff[i], c * HW * 4);
        if constexpr (MASK) {
            const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(yrelu) + (size_t)cb * HW, 0, FC * HW * 4, 0x00020000);
#pragma unroll
            for (int c = 0; c < FC; ++c)
#pragma unroll
                for (int i = 0; i < FN; ++i) ry[c][i] = first_buffer_load(ys, goff[i], c * HW * 4);
        }
        if constexpr (HALO) {
            const __amdgpu_buffer_rsrc_t hs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(ghalo) + (size_t)cb * W, 0, (64 + FC) * W * 4, 0x00020000);
#pragma unroll
            for (int c = 0; c < FC; ++c)
#pragma unroll
                for (int i = 0; i < FN; ++i) rh[c][i] = first_buffer_load(hs, hoff[i], c * W * 4);
        }
    };
    auto store_pass = [&](int buf) {
#pragma unroll
        for (int c = 0; c < FC; ++c)
#pragma unroll
            for (int i = 0; i < FN; ++i) {
                float v = rg[c][i];
                if constexpr (MASK) v = (ry[c][i] > 0.f) ? v : 0.f;       // threshold_backward
                if constexpr (HALO) v += rh[c][i];
                (&tile[buf][c][0][0])[loff[i]] = v;
            }
    };

    float acc[4][3];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 0.f;
    load_pass(0);
    store_pass(0);
    __syncthreads();
    for (int cb = 0, pass = 0; cb < 64; cb += FC, ++pass) {
        const int buf = pass & 1;
        const bool more = cb + FC < 64;
        if (more) load_pass(cb + FC);
#pragma unroll
        for (int c = 0; c < FC; ++c) {
            const float* wc = w + (cb + c) * 27;
            // window rows ty .. ty + 2 (gradient rows y - 1 .. y + 1), staged columns 4 tx .. 4 tx + 5 (x - 1 .. x + 4)
            float win[3][6];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float* row = &tile[buf][c][ty + r][4 * tx];
                const f32x4 a = *reinterpret_cast<const f32x4*>(row);
                const f32x2 b = *reinterpret_cast<const f32x2*>(row + 4);
                win[r][0] = a[0]; win[r][1] = a[1]; win[r][2] = a[2]; win[r][3] = a[3]; win[r][4] = b[0]; win[r][5] = b[1];
            }
            // exactly one tap links each of the 9 neighbouring outputs to a padded position
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int k = (1 - dy) * 3 + (1 - dx);
                    const float w0 = wc[k], w1 = wc[9 + k], w2 = wc[18 + k];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float g = win[1 + dy][j + 1 + dx];
                        acc[j][0] = fmaf(w0, g, acc[j][0]);
                        acc[j][1] = fmaf(w1, g, acc[j][1]);
                        acc[j][2] = fmaf(w2, g, acc[j][2]);
                    }
                }
        }
        if (more) store_pass(buf ^ 1);
        __syncthreads();
    }
    if (y <= py1) {
        const size_t plane = (size_t)(py1 - py0 + 1) * (W + 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (x + j > W) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) dp[c * plane + (size_t)(y - py0) * (W + 2) + (x + j + 1)] = acc[j][c];
        }
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
    static Option mfma_opt("ST_CONV_FIRST_MFMA", 1);      // 0: the exact-fp32 VALU kernel also in fp16x3 mode (A/B runs)
    if (out_amax && mfma_opt.get()) {                     // fp16x3 trunk (the plan passes the bound only then)
        const int blocks = ceil_div(width, MC) * ceil_div(height, MR);
        hipLaunchKernelGGL(conv_first_fwd_mfma_kernel, dim3(blocks), dim3(256), 0, stream, image, w, b, out, height, width,
                           halo, has_up, has_down, out_amax);
        ST_LAUNCH_CHECK();
        return 0;
    }
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
    const int blocks = ceil_div(width + 2, FTX) * ceil_div(rows, FTY);
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, stream, grad_out, relu_out, w, dp_scratch, height, width,
                           ghalo, has_up, has_down);
    };
    if (relu_out && ghalo) launch(conv_first_dp_kernel<true, true>);
    else if (relu_out) launch(conv_first_dp_kernel<true, false>);
    else if (ghalo) launch(conv_first_dp_kernel<false, true>);
    else launch(conv_first_dp_kernel<false, false>);
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv_first_fold_kernel, dim3(ceil_div(height * width, 256)), dim3(256), 0, stream, dp_scratch,
                       grad_image, height, width, accumulate, has_up, has_down);
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
