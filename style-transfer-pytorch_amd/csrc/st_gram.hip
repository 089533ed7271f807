// Second raw moment and mean of a feature map:  SRM = F F^T / N,  mu = F 1 / N   (F is [C][N]).
// Replaces StyleLossW2.get_target (style_transfer.py:162-168: target.mean([-2,-1]) and
// einsum('chw,dhw->cd') / (h*w)), which dispatches to bmm in the reference.
//
// Split-K over the pixel axis so that even C = 64 (a single 64x64 output tile) fills the chip:
//   pass 1 (MFMA): partial[s] = F[:, K_s] F[:, K_s]^T  and row sums of F over K_s
//   pass 2: fixed-order sum over s (deterministic), divide by N.
// Under spatial sharding pass 1 runs on the local strip and the partials are
// all-reduced before pass 2; nothing else in the kernel changes.
//
// F F^T is symmetric: only the tile pairs ti <= tj are computed, and an off-diagonal workgroup writes its
// tile and the transpose (the MFMA accumulator layout gives each lane 4 consecutive ROWS of one column, i.e.
// 16 contiguous bytes of the transposed tile).  The products commute and the k order is the same, so the
// mirrored entries are bit-identical to what computing them would give.
//
// Two arithmetic variants of pass 1:
//   exact fp32 MFMA (v_mfma_f32_32x32x2_f32), used by the standalone operator and the fp32 / bf16 trunk modes;
//   fp16x3 (the trunk's default arithmetic, see st_conv_split.hip): the tile is scaled by the power of two
//   that the producing convolution's bound on max |F| selects, split into two fp16 planes while it is staged,
//   and accumulated as h0 h0^T + h0 h1^T + h1 h0^T with v_mfma_f32_32x32x16_f16 -- fp32-class accuracy at
//   3/16 of the fp32 matrix-pipe time, which turns the kernel from MFMA-bound into a streaming read of F.
//   The row sums (the mean) are always taken from the fp32 values.
//
// fp32 variant: LDS tile [64 ch][64 px] with a row pitch of 68 floats: ds_read_b128 (4 consecutive pixels of one
// channel per lane) is then conflict-free (16-lane groups hit 16 distinct 16-byte slots), and one
// read feeds four v_mfma_f32_32x32x2_f32 (lanes 0-31 take pixels k..k+3, lanes 32-63 k+4..k+7).
#include "st_common.h"

namespace st {
namespace {

constexpr int GT = 64;        // channels per tile side
constexpr int GK = 64;        // pixels per LDS stage
constexpr int GP = 68;        // LDS row pitch (floats)
constexpr int HK = 32;        // fp16 variant: pixels per LDS stage
constexpr int HP = 40;        // fp16 variant: LDS row pitch in halfs (80 B: b128 reads of 16 rows hit 16 slots)

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// workgroup -> tile pair (ti <= tj) of the upper triangle, row-major
__device__ __forceinline__ void tile_pair(int block, int tiles, int& ti, int& tj) {
    int t = block, row = 0, len = tiles;
    while (t >= len) {
        t -= len;
        --len;
        ++row;
    }
    ti = row;
    tj = row + t;
}

// partial[ti*64 + wi*32 + row][tj*64 + wj*32 + col] and, off the diagonal, its transpose
__device__ __forceinline__ void store_tile_pair(float* __restrict__ out, int C, int ti, int tj, int wi, int wj,
                                                int l31, int half, const f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = ti * GT + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int col = tj * GT + wj * 32 + l31;
        out[(size_t)row * C + col] = acc[r];
    }
    if (ti != tj) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = tj * GT + wj * 32 + l31;
            const int col = ti * GT + wi * 32 + 8 * q + 4 * half;
            const f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *reinterpret_cast<f32x4*>(out + (size_t)row * C + col) = v;
        }
    }
}

__global__ __launch_bounds__(256) void gram_partial_kernel(const float* __restrict__ feat, int C,
                                                           long long N, int splits, long long per_split,
                                                           float* __restrict__ partial,
                                                           float* __restrict__ partial_sum, int xcd_remap) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][GT * GP];   // [buffer][operand][64 x 68]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int tiles = C / GT;
    int ti, tj;
    const unsigned int lb = logical_block(xcd_remap);          // the tile pairs of one split share an XCD's L2
    tile_pair((int)(lb % gridDim.x), tiles, ti, tj);
    const int split = (int)(lb / gridDim.x);
    const long long k_begin = split * per_split;
    const long long k_end = (k_begin + per_split < N) ? k_begin + per_split : N;
    const bool diag = (ti == tj);
    const bool vec_ok = (N % 4 == 0);

    // staging map: thread -> (row = tid/16 + 16 r, 4 consecutive pixels at col 4*(tid%16))
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    f32x4 ra[4], rb[4];
    float rowsum[4] = {0.f, 0.f, 0.f, 0.f};

    auto load_tile = [&](long long k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = srow + 16 * r;
            const long long k = k0 + scol;
            const float* pa = feat + (size_t)(ti * GT + row) * N + k;
            const float* pb = feat + (size_t)(tj * GT + row) * N + k;
            f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
            if (vec_ok && k + 3 < k_end) {
                va = *reinterpret_cast<const f32x4*>(pa);
                if (!diag) vb = *reinterpret_cast<const f32x4*>(pb);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (k + e < k_end) {
                        va[e] = pa[e];
                        if (!diag) vb[e] = pb[e];
                    }
                }
            }
            ra[r] = va;
            rb[r] = vb;
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = srow + 16 * r;
            *reinterpret_cast<f32x4*>(&lds[buf][0][row * GP + scol]) = ra[r];
            if (!diag) *reinterpret_cast<f32x4*>(&lds[buf][1][row * GP + scol]) = rb[r];
            rowsum[r] += (ra[r][0] + ra[r][1]) + (ra[r][2] + ra[r][3]);
        }
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const long long span = k_end - k_begin;
    const int nstages = (int)((span + GK - 1) / GK);
    if (nstages > 0) {
        load_tile(k_begin);
        store_tile(0);
    }
    __syncthreads();
    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nstages;
        if (more) load_tile(k_begin + (long long)(st + 1) * GK);
        const float* la = &lds[buf][0][(wi * 32 + l31) * GP + 4 * half];
        const float* lb = &lds[buf][diag ? 0 : 1][(wj * 32 + l31) * GP + 4 * half];
#pragma unroll
        for (int kb = 0; kb < GK / 8; ++kb) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(la + kb * 8);
            const f32x4 b = *reinterpret_cast<const f32x4*>(lb + kb * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    store_tile_pair(partial + (size_t)split * C * C, C, ti, tj, wi, wj, l31, half, acc);
    // row sums: the 16 threads sharing a staging row are 16 consecutive lanes of one wave
    if (diag) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = rowsum[r];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            v += __shfl_xor(v, 8, 64);
            if ((tid & 15) == 0) partial_sum[(size_t)split * C + ti * GT + srow + 16 * r] = v;
        }
    }
}

// fp16x3 variant (see the header).  Staging map: thread -> (row = tid/4, 8 consecutive pixels at 8*(tid%4)).
__global__ __launch_bounds__(256) void gram_partial_f16_kernel(const float* __restrict__ feat, int C, long long N,
                                                               int splits, long long per_split,
                                                               const unsigned int* __restrict__ bound,
                                                               float* __restrict__ partial,
                                                               float* __restrict__ partial_sum, int xcd_remap) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2][2][2][GT * HP];   // [buffer][operand][plane][64 x 40]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    const int tiles = C / GT;
    int ti, tj;
    const unsigned int lb = logical_block(xcd_remap);          // the tile pairs of one split share an XCD's L2
    tile_pair((int)(lb % gridDim.x), tiles, ti, tj);
    const int split = (int)(lb / gridDim.x);
    const long long k_begin = split * per_split;
    const long long k_end = (k_begin + per_split < N) ? k_begin + per_split : N;
    const bool diag = (ti == tj);
    const bool vec_ok = (N % 4 == 0);
    const int ex = scale_exp(amax_read(bound));
    const float scale = pow2f(ex), unscale = pow2f(-ex);

    const int srow = tid >> 2, scol = (tid & 3) * 8;
    float va[8], vb[8];
    float rowsum = 0.f;

    auto load_row = [&](const float* p, long long k, float (&v)[8]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (vec_ok && k + 4 * q + 3 < k_end) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = t[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = (k + 4 * q + e < k_end) ? p[4 * q + e] : 0.f;
            }
        }
    };
    auto load_tile = [&](long long k0) {
        const long long k = k0 + scol;
        load_row(feat + (size_t)(ti * GT + srow) * N + k, k, va);
        if (!diag) load_row(feat + (size_t)(tj * GT + srow) * N + k, k, vb);
    };
    auto split_store = [&](const float (&v)[8], _Float16* p0, _Float16* p1) {
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
    };
    auto store_tile = [&](int buf) {
        const int off = srow * HP + scol;
        split_store(va, &lds[buf][0][0][off], &lds[buf][0][1][off]);
        if (!diag) split_store(vb, &lds[buf][1][0][off], &lds[buf][1][1][off]);
        rowsum += ((va[0] + va[1]) + (va[2] + va[3])) + ((va[4] + va[5]) + (va[6] + va[7]));
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const long long span = k_end - k_begin;
    const int nstages = span > 0 ? (int)((span + HK - 1) / HK) : 0;
    if (nstages > 0) {
        load_tile(k_begin);
        store_tile(0);
    }
    __syncthreads();
    const int ob = diag ? 0 : 1;
    for (int st = 0; st < nstages; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nstages;
        if (more) load_tile(k_begin + (long long)(st + 1) * HK);
        const int aoff = (wi * 32 + l31) * HP + 8 * half, boff = (wj * 32 + l31) * HP + 8 * half;
#pragma unroll
        for (int kb = 0; kb < HK / 16; ++kb) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(&lds[buf][0][0][aoff + kb * 16]);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(&lds[buf][0][1][aoff + kb * 16]);
            const f16x8 b0 = *reinterpret_cast<const f16x8*>(&lds[buf][ob][0][boff + kb * 16]);
            const f16x8 b1 = *reinterpret_cast<const f16x8*>(&lds[buf][ob][1][boff + kb * 16]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc, 0, 0, 0);
        }
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc[r] * unscale * unscale;
    float* out = partial + (size_t)split * C * C;
    if (diag) {
        // h0 h1^T + h1 h0^T is symmetric as a sum, but entry (i, j) adds the two cross products in the opposite
        // order to entry (j, i): write the upper triangle of the tile and its mirror (through LDS; the main
        // loop ended on a barrier, so the staging buffers are free)
        float* t = reinterpret_cast<float*>(&lds[0][0][0][0]);            // 64 x 65 floats
#pragma unroll
        for (int r = 0; r < 16; ++r)
            t[(wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 65 + wj * 32 + l31] = acc[r];
        __syncthreads();
        for (int idx = tid; idx < GT * GT; idx += 256) {
            const int row = idx >> 6, col = idx & 63;
            out[(size_t)(ti * GT + row) * C + ti * GT + col] = row <= col ? t[row * 65 + col] : t[col * 65 + row];
        }
    } else {
        store_tile_pair(out, C, ti, tj, wi, wj, l31, half, acc);
    }
    // row sums: the 4 threads sharing a staging row are 4 consecutive lanes
    if (diag) {
        float v = rowsum;
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        if ((tid & 3) == 0) partial_sum[(size_t)split * C + ti * GT + srow] = v;
    }
}

// fp16x3 variant with 128 x 128 output tiles (C % 128 == 0: relu2_1 ... relu5_1).  The 64-tile kernel above reads
// F once per tile pair it takes part in (C = 512: 8 x), and at >= 1024^2 it runs at the chip's copy rate on that ACTUAL
// traffic; a 128-tile halves it and gives a wave a 2 x 2 register block (8 LDS operand fetches per 12 MFMAs instead of 4
// per 3).  One LDS buffer (40 KB), the next stage's global loads in flight during the MFMAs, two barriers per stage.
// Symmetry: a wave writes the 32 x 32 blocks on or above the diagonal and their mirrors from the SAME accumulators
// (entry (j, i) is defined as entry (i, j), i <= j), so the result is exactly symmetric without the LDS mirror pass.
__global__ __launch_bounds__(256, 2) void gram_partial_f16_wide_kernel(const float* __restrict__ feat, int C, long long N,
                                                                       int splits, long long per_split,
                                                                       const unsigned int* __restrict__ bound,
                                                                       float* __restrict__ partial,
                                                                       float* __restrict__ partial_sum, int xcd_remap) {
    constexpr int TS = 128;
    __shared__ __attribute__((aligned(16))) _Float16 lds[2][2][TS * HP];        // [operand][plane][128 x 40]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wi = wave >> 1, wj = wave & 1;
    int ti, tj;
    const unsigned int lb = logical_block(xcd_remap);          // the tile pairs of one split share an XCD's L2
    tile_pair((int)(lb % gridDim.x), C / TS, ti, tj);
    const int split = (int)(lb / gridDim.x);
    const long long k_begin = split * per_split;
    const long long k_end = (k_begin + per_split < N) ? k_begin + per_split : N;
    const bool diag = (ti == tj);
    const bool vec_ok = (N % 4 == 0);
    const int ex = scale_exp(amax_read(bound));
    const float scale = pow2f(ex), unscale = pow2f(-ex);

    const int srow = tid >> 2, scol = (tid & 3) * 8;        // rows srow and srow + 64, 8 consecutive pixels
    float va[2][8], vb[2][8];
    float rowsum[2] = {0.f, 0.f};
    auto load_row = [&](const float* p, long long k, float (&v)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (vec_ok && k + 4 * q + 3 < k_end) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(p + 4 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = t[e];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = (k + 4 * q + e < k_end) ? p[4 * q + e] : 0.f;
            }
        }
    };
    auto load_tile = [&](long long k0) __attribute__((always_inline)) {
        const long long k = k0 + scol;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            load_row(feat + (size_t)(ti * TS + h * 64 + srow) * N + k, k, va[h]);
            if (!diag) load_row(feat + (size_t)(tj * TS + h * 64 + srow) * N + k, k, vb[h]);
        }
    };
    auto split_store = [&](const float (&v)[8], _Float16* p0, _Float16* p1) __attribute__((always_inline)) {
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
    };
    auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int off = (h * 64 + srow) * HP + scol;
            split_store(va[h], &lds[0][0][off], &lds[0][1][off]);
            if (!diag) split_store(vb[h], &lds[1][0][off], &lds[1][1][off]);
            rowsum[h] += ((va[h][0] + va[h][1]) + (va[h][2] + va[h][3])) + ((va[h][4] + va[h][5]) + (va[h][6] + va[h][7]));
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const long long span = k_end > k_begin ? k_end - k_begin : 0;
    const int nstages = (int)((span + HK - 1) / HK);
    // on a diagonal tile the wave below the diagonal (wi = 1, wj = 0) would only recompute mirrored blocks
    const bool idle = diag && wi > wj;
    if (nstages > 0) {
        load_tile(k_begin);
        store_tile();
    }
    __syncthreads();
    const int ob = diag ? 0 : 1;
    for (int st = 0; st < nstages; ++st) {
        const bool more = st + 1 < nstages;
        if (more) load_tile(k_begin + (long long)(st + 1) * HK);
        if (!idle) {
#pragma unroll
            for (int kb = 0; kb < HK / 16; ++kb) {
                f16x8 a0[2], a1[2], b0[2], b1[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int aoff = (wi * 64 + i * 32 + l31) * HP + 8 * half + kb * 16;
                    const int boff = (wj * 64 + i * 32 + l31) * HP + 8 * half + kb * 16;
                    a0[i] = *reinterpret_cast<const f16x8*>(&lds[0][0][aoff]);
                    a1[i] = *reinterpret_cast<const f16x8*>(&lds[0][1][aoff]);
                    b0[i] = *reinterpret_cast<const f16x8*>(&lds[ob][0][boff]);
                    b1[i] = *reinterpret_cast<const f16x8*>(&lds[ob][1][boff]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[i], b0[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[i], b1[j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[i], b0[j], acc[i][j], 0, 0, 0);
                    }
            }
        }
        if (more) {
            __syncthreads();
            store_tile();
        }
        __syncthreads();
    }

    float* out = partial + (size_t)split * C * C;
    if (!idle) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int rb = ti * TS + wi * 64 + i * 32, cb = tj * TS + wj * 64 + j * 32;   // block origin
                if (rb > cb) continue;                                 // below the diagonal: written as a mirror
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb + (r & 3) + 8 * (r >> 2) + 4 * half, col = cb + l31;
                    const float v = acc[i][j][r] * unscale * unscale;
                    if (rb < cb || row <= col) {
                        out[(size_t)row * C + col] = v;
                        if (row != col) out[(size_t)col * C + row] = v;
                    }
                }
            }
    }
    if (diag) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float v = rowsum[h];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            if ((tid & 3) == 0) partial_sum[(size_t)split * C + ti * TS + h * 64 + srow] = v;
        }
    }
}

// Fixed-order reduction over the splits.  A workgroup owns 32 consecutive outputs; its 8 thread
// groups take every 8th split (independent loads, several in flight), then combine in a fixed tree.
// cov != nullptr: the workgroups that own second-moment entries also write cov = srm - mean mean^T + eps I
// (StyleLossW2.srm_to_cov + eye_like * eps, style_transfer.py:156,170-177) - they reduce the row sums of their row and of
// their 32 columns in the same order as the workgroups that own `mean` do, so cov is what cov_kernel would compute from
// this kernel's outputs, bit for bit, without being a launch of its own in front of every head's chain.
__global__ __launch_bounds__(256) void gram_finalize_kernel(const float* __restrict__ partial,
                                                            const float* __restrict__ partial_sum, int C,
                                                            long long N, int splits, float* __restrict__ mean,
                                                            float* __restrict__ srm, float* __restrict__ cov, float eps) {
#pragma clang fp contract(off)
    __shared__ float red[8][32];
    const long long total = (long long)C * C;
    const float n = (float)N;
    const int e = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const long long i = (long long)blockIdx.x * 32 + e;
    const bool is_srm = i < total;
    const bool valid = i < total + C;
    // one output's splits, this thread group's share (the caller combines the 8 shares through `red`)
    auto share = [&](const float* src, size_t stride) __attribute__((always_inline)) {
        float s = 0.f;
        int k = grp;
        for (; k + 24 < splits; k += 32) {
            const float v0 = src[(size_t)k * stride], v1 = src[(size_t)(k + 8) * stride];
            const float v2 = src[(size_t)(k + 16) * stride], v3 = src[(size_t)(k + 24) * stride];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; k < splits; k += 8) s += src[(size_t)k * stride];
        return s;
    };
    auto combined = [&](int col) __attribute__((always_inline)) {
        return ((red[0][col] + red[1][col]) + (red[2][col] + red[3][col])) +
               ((red[4][col] + red[5][col]) + (red[6][col] + red[7][col]));
    };
    red[grp][e] = valid ? share(is_srm ? partial + i : partial_sum + (i - total), is_srm ? (size_t)total : (size_t)C) : 0.f;
    __syncthreads();
    float t = 0.f;
    if (valid) t = combined(e) / n;
    if (grp == 0 && valid) {
        if (is_srm) srm[i] = t;
        else mean[i - total] = t;
    }
    if (cov == nullptr || (long long)blockIdx.x * 32 >= total) return;       // (uniform: C % 32 == 0, a block is all srm or all mean)
    const int row = (int)(i / C), col = (int)(i % C);
    __syncthreads();
    red[grp][e] = share(partial_sum + col, (size_t)C);                       // mean of this thread's column
    __syncthreads();
    const float mean_c = combined(e) / n;
    __syncthreads();
    if (e == 0) red[grp][0] = share(partial_sum + row, (size_t)C);           // mean of the block's row
    __syncthreads();
    const float mean_r = combined(0) / n;
    if (grp == 0) {
        const float outer = mean_r * mean_c;
        const float d = t - outer;
        cov[i] = d + ((row == col) ? eps : 0.f);
    }
}

}  // namespace

namespace {
// Measured (tools/gram_bench.py, isolated): 2048^2 relu3_1 (C = 256, 262 144 px) 212 -> 153 us, relu4_1 (C = 512,
// 65 536 px) 194 -> 129 us; but relu5_1 there (16 384 px) 58 -> 63 us, every tap of a 512^2 image 25 % slower (too
// few workgroups), C = 256 with 65 536 px (1024^2) neutral, and C = 128 unchanged or slower (one tile pair, a quarter of
// it idle): C >= 256 and >= 65 536 pixels.
bool gram_wide_tiles(int channels, long long npix) {
    static Option wide_opt("ST_GRAM_WIDE", 1);            // 0: 64 x 64 tiles everywhere, 2: 128 x 128 wherever possible
    const int mode = wide_opt.get();
    if (mode == 0 || channels % 128 != 0) return false;
    return mode == 2 || (channels >= 256 && npix >= 65536);
}
}  // namespace

int gram_choose_splits(int channels, long long npix, int max_splits) {
    const int side = gram_wide_tiles(channels, npix) ? 128 : GT;
    const int tiles = (channels / side) * (channels / side + 1) / 2;    // upper triangle of tile pairs
    long long want = (768 + tiles - 1) / tiles;                  // ~3 workgroups per CU in total
    const long long by_len = (npix + 2 * GK - 1) / (2 * GK);     // at least two LDS stages per split
    if (want > by_len) want = by_len;
    if (want > max_splits) want = max_splits;
    if (want < 1) want = 1;
    return (int)want;
}

int launch_gram_partial(const float* feat, int channels, long long npix, int splits, GramWorkspace ws,
                        hipStream_t s, const unsigned int* bound) {
    ST_REQUIRE(channels % GT == 0, "gram: channel count must be a multiple of 64");
    ST_REQUIRE(splits >= 1 && splits <= ws.max_splits, "gram: bad split count %d", splits);
    long long per_split = (npix + splits - 1) / splits;
    per_split = (per_split + 3) & ~3ll;                          // keep 16-byte alignment of the splits
    static Option remap_opt("ST_XCD_REMAP", 1);            // 0: plain workgroup order (A/B runs)
    const int remap = remap_opt.get();
    if (bound && gram_wide_tiles(channels, npix)) {
        const int wide = (channels / 128) * (channels / 128 + 1) / 2;
        hipLaunchKernelGGL(gram_partial_f16_wide_kernel, dim3(wide, splits), dim3(256), 0, s, feat, channels, npix, splits,
                           per_split, bound, ws.partial, ws.partial_sum, remap);
        ST_LAUNCH_CHECK();
        return 0;
    }
    const int tiles = (channels / GT) * (channels / GT + 1) / 2;
    if (bound)
        hipLaunchKernelGGL(gram_partial_f16_kernel, dim3(tiles, splits), dim3(256), 0, s, feat, channels, npix,
                           splits, per_split, bound, ws.partial, ws.partial_sum, remap);
    else
        hipLaunchKernelGGL(gram_partial_kernel, dim3(tiles, splits), dim3(256), 0, s, feat, channels, npix,
                           splits, per_split, ws.partial, ws.partial_sum, remap);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_gram_finalize(GramWorkspace ws, int channels, long long npix, int splits, float* mean, float* srm,
                         hipStream_t s, float* cov, float cov_eps) {
    const long long total = (long long)channels * channels + channels;
    const int blocks = (int)((total + 31) / 32);
    hipLaunchKernelGGL(gram_finalize_kernel, dim3(blocks), dim3(256), 0, s, ws.partial, ws.partial_sum,
                       channels, npix, splits, mean, srm, cov, cov_eps);
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
