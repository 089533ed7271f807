// 1x1 convolution in the trunk's fp16x3 arithmetic:  out[co][px] = sum_ci W[co][ci] x[ci][px] + b[co].
// This is the style heads' gradient step  dF = Ssym F + b 1^T  (the backward of the einsum + mean in
// StyleLossW2.get_target, style_transfer.py:162-168) on the LARGE taps, where the exact fp32 MFMA kernel
// (st_conv.hip, TAPS = 1) is matrix-pipe bound at ~45 TF and shares that pipe with the backward trunk.  Here
// both operands are scaled by a power of two (bounds: the tap's max |F| from its producing convolution, max |S|
// from style_grad_finish_kernel), split into two fp16 planes WHILE THEY ARE STAGED (the weights change every
// iteration, so there is no pre-split copy), and accumulated as w0 x0 + w0 x1 + w1 x0 with
// v_mfma_f32_32x32x16_f16: fp32-class accuracy at 3/16 of the fp32 matrix-pipe time, which leaves the kernel
// bound by reading F and writing dF once (8 bytes per output element).
//
// Workgroup = 64 output channels x (128 WN) pixels, 4 waves side by side along the pixels, K chunks of 32 input
// channels.  LDS per plane: x as [pixel][32 ci] and W as [co][32 ci], row pitch 40 halfs (80 bytes: the
// ds_read_b128 of 16 consecutive rows hits 16 distinct 16-byte slots).  x is [ci][pixels] in HBM, so the
// transposition happens in registers: a thread loads 4 consecutive pixels (16 bytes) of 8 channels and writes
// 4 x 2 rows of 8 halfs; consecutive lanes take different channel groups so the writes spread over the banks.
// The next chunk's global loads are issued before the current chunk's MFMAs.  Small taps (where the fp32
// launcher would split K) stay on the fp32 kernel: with few pixels the chunk loop is latency-bound.
#include "st_common.h"

namespace st {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int PT = 40;        // LDS row pitch in halfs
constexpr int KC = 32;        // input channels per chunk

__device__ __forceinline__ void split_scaled(const float (&v)[8], float scale, _Float16* p0, _Float16* p1) {
    f16x8 h0, h1;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float x = v[e] * scale;
        const _Float16 a = (_Float16)x;
        h0[e] = a;
        h1[e] = (_Float16)(x - (float)a);
    }
    *reinterpret_cast<f16x8*>(p0) = h0;
    *reinterpret_cast<f16x8*>(p1) = h1;
}

// CM = 32-channel output blocks per wave: the workgroup's tile is (32 CM) output channels x (128 WN) pixels.  CM = 4
// (128 output channels) halves how often F is re-read by the Cout tiles of a pixel tile: for C >= 128 the kernel runs
// at the chip's copy rate on its ACTUAL traffic (F once per Cout tile), so fewer Cout tiles is the lever.
template <int WN, int CM>
__global__ __launch_bounds__(256, 2) void conv1x1_f16_kernel(const float* __restrict__ in, const float* __restrict__ wgt,
                                                          const float* __restrict__ bias, float* __restrict__ out,
                                                          int cin, int cout, long long npix,
                                                          const unsigned int* __restrict__ in_bound,
                                                          const unsigned int* __restrict__ w_bound,
                                                          unsigned int* __restrict__ out_amax, int xcd_remap) {
    constexpr int TPX = 128 * WN;
    __shared__ __attribute__((aligned(16))) _Float16 lds_x[2][TPX * PT];   // [plane][pixel][ci]
    __shared__ __attribute__((aligned(16))) _Float16 lds_w[2][32 * CM * PT];   // [plane][co][ci]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    constexpr int TCO = 32 * CM;
    const int ctiles = cout / TCO;
    const unsigned int lb = logical_block(xcd_remap);    // the co tiles of one pixel tile run together, on one XCD (L2)
    const int ct = (int)(lb % ctiles);
    const long long px0 = (long long)(lb / ctiles) * TPX;
    const int ea = scale_exp(amax_read(in_bound)), ew = scale_exp(amax_read(w_bound));
    const float sa = pow2f(ea), sw = pow2f(ew);
    const bool vec_ok = (npix % 4 == 0);

    // staging maps: activations (4 pixels at 4*pg, channels 8*cg..+7), weights (row tid/4, channels 8*cg..+7)
    const int cg = tid & 3, pg = tid >> 2;
    const bool xact = pg < TPX / 4;
    f32x4 xr[8], wr[CM];                                  // weights: rows pg (and pg + 64), 8 channels

    auto load_chunk = [&](int k0) {
        if (xact) {
            const long long px = px0 + 4 * pg;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float* p = in + (size_t)(k0 + cg * 8 + c) * npix + px;
                if (vec_ok && px + 3 < npix) {
                    xr[c] = *reinterpret_cast<const f32x4*>(p);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[c][e] = (px + e < npix) ? p[e] : 0.f;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < CM / 2; ++h) {
            const float* w = wgt + (size_t)(ct * TCO + h * 64 + pg) * cin + k0 + cg * 8;
            wr[2 * h] = *reinterpret_cast<const f32x4*>(w);
            wr[2 * h + 1] = *reinterpret_cast<const f32x4*>(w + 4);
        }
    };
    auto store_chunk = [&]() {
        if (xact) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = xr[c][j];
                const int off = (4 * pg + j) * PT + cg * 8;
                split_scaled(v, sa, &lds_x[0][off], &lds_x[1][off]);
            }
        }
#pragma unroll
        for (int h = 0; h < CM / 2; ++h) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = wr[2 * h + (e >> 2)][e & 3];
            const int off = (h * 64 + pg) * PT + cg * 8;
            split_scaled(v, sw, &lds_w[0][off], &lds_w[1][off]);
        }
    };

    f32x16 acc[CM][WN];
#pragma unroll
    for (int m = 0; m < CM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    const int nchunks = cin / KC;
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const bool more = ch + 1 < nchunks;
        if (more) load_chunk((ch + 1) * KC);
#pragma unroll
        for (int kb = 0; kb < KC / 16; ++kb) {
            // pixel-block operands stay for the whole k step, the output-channel blocks pass through one register pair
            f16x8 b[WN][2];
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const int row = (wave * WN + n) * 32 + l31;
                b[n][0] = *reinterpret_cast<const f16x8*>(&lds_x[0][row * PT + kb * 16 + 8 * half]);
                b[n][1] = *reinterpret_cast<const f16x8*>(&lds_x[1][row * PT + kb * 16 + 8 * half]);
            }
#pragma unroll
            for (int m = 0; m < CM; ++m) {
                const f16x8 a0 = *reinterpret_cast<const f16x8*>(&lds_w[0][(m * 32 + l31) * PT + kb * 16 + 8 * half]);
                const f16x8 a1 = *reinterpret_cast<const f16x8*>(&lds_w[1][(m * 32 + l31) * PT + kb * 16 + 8 * half]);
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[n][0], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b[n][1], acc[m][n], 0, 0, 0);
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b[n][0], acc[m][n], 0, 0, 0);
                }
            }
        }
        if (more) {
            __syncthreads();
            store_chunk();
            __syncthreads();
        }
    }

    // acc[m][n][r]: co = ct*64 + m*32 + (r&3) + 8*(r>>2) + 4*half,  pixel = px0 + (wave*WN + n)*32 + l31
    const float ua = pow2f(-ea), uw = pow2f(-ew);
    unsigned int amax = 0;
#pragma unroll
    for (int m = 0; m < CM; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = ct * TCO + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float b = bias ? bias[co] : 0.f;
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const long long px = px0 + (wave * WN + n) * 32 + l31;
                const float v = acc[m][n][r] * ua * uw + b;
                if (px < npix) {
                    out[(size_t)co * npix + px] = v;
                    const unsigned int bits = abs_bits(v);
                    amax = bits > amax ? bits : amax;
                }
            }
        }
    }
    if (out_amax) amax_commit(amax, out_amax);
}

}  // namespace

bool conv1x1_split_applies(const ConvProblem& p) {
    return p.taps == 1 && p.planes == 2 && p.elem == 1 && p.amax_word && p.wgt_amax && !p.relu && !p.accumulate &&
           !p.mask && !p.out_mask && p.cin % KC == 0 && p.cout % 64 == 0 &&
           ((reinterpret_cast<uintptr_t>(p.in) | reinterpret_cast<uintptr_t>(p.wgt)) & 15) == 0;
}

int launch_conv1x1_split(const ConvProblem& p, hipStream_t stream) {
    ST_REQUIRE(conv1x1_split_applies(p), "conv1x1 (fp16x3): unsupported problem");
    const long long npix = (long long)p.height * p.width;
    static Option wide_opt("ST_CONV1X1_CO128", 1);        // 0: 64-channel tiles everywhere (A/B runs)
    static Option remap_opt("ST_XCD_REMAP", 1);            // 0: plain workgroup order (A/B runs)
    const int remap = remap_opt.get();
    const bool wide = wide_opt.get() && p.cout % 128 == 0 && ((npix + 255) / 256) * (p.cout / 128) >= 512;
    if (wide) {
        const long long wg = ((npix + 255) / 256) * (p.cout / 128);
        hipLaunchKernelGGL((conv1x1_f16_kernel<2, 4>), dim3((unsigned)wg), dim3(256), 0, stream, p.in, p.wgt, p.bias,
                           p.out, p.cin, p.cout, npix, p.amax_word, p.wgt_amax, p.out_amax, remap);
        ST_LAUNCH_CHECK();
        return 0;
    }
    const int ctiles = p.cout / 64;
    const long long wg2 = ((npix + 255) / 256) * ctiles;
    if (wg2 >= 512) {
        hipLaunchKernelGGL((conv1x1_f16_kernel<2, 2>), dim3((unsigned)wg2), dim3(256), 0, stream, p.in, p.wgt, p.bias,
                           p.out, p.cin, p.cout, npix, p.amax_word, p.wgt_amax, p.out_amax, remap);
    } else {
        const long long wg1 = ((npix + 127) / 128) * ctiles;
        hipLaunchKernelGGL((conv1x1_f16_kernel<1, 2>), dim3((unsigned)wg1), dim3(256), 0, stream, p.in, p.wgt, p.bias,
                           p.out, p.cin, p.cout, npix, p.amax_word, p.wgt_amax, p.out_amax, remap);
    }
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
