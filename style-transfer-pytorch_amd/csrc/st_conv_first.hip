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
                                                             int has_down, unsigned int* out_amax, int cpt) {
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
    // output channels [blockIdx.y cpt, + cpt): see launch_conv_first_fwd
    for (int co = blockIdx.y * cpt, co_end = co + cpt; co < co_end; ++co) {
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

// The same, four consecutive pixels of a row per thread (W % 4 == 0, 16-byte aligned output): one 16-byte store per
// output channel instead of four 4-byte ones, and the 27 x 64 FMAs of a pixel pair as packed v_pk_fma_f32 (two fp32
// FMAs per lane and cycle: at one FMA per cycle the 14.5 GFLOP of a 2048^2 image are 185 us of VALU time on their own
// - the one-pixel kernel ran 380 us for 1.07 GB written = 2.8 TB/s).  Same FMA order per output (k ascending, then the
// bias): bit-identical to the one-pixel kernel, which remains for the other widths.
__global__ __launch_bounds__(256) void conv_first_fwd4_kernel(const float* __restrict__ image,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ b,
                                                              float* __restrict__ out, int H, int W,
                                                              const float* __restrict__ halo, int has_up,
                                                              int has_down, unsigned int* out_amax, int cpt) {
#pragma clang fp contract(off)
    const int HW = H * W, gpr = W >> 2;
    // threads past the end redo the last group (identical stores) so that whole waves reach amax_commit
    const int grp = min((int)(blockIdx.x * 256 + threadIdx.x), H * gpr - 1);
    const int y = grp / gpr, x0 = (grp - y * gpr) * 4;
    // pair[c][ky][j] = normalised pixels (x0 - 1 + j, x0 + j) of row y + ky - 1: the operand pairs of output pixels
    // (x0, x0 + 1) at tap column kx = j, and of (x0 + 2, x0 + 3) at kx = j - 2
    f32x2 pair[3][3][5];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yr = y + ky - 1;
            const int yy = min(max(yr, 0), H - 1);
            const float* row = image + (size_t)c * HW + (size_t)yy * W;
            if (yr < 0 && has_up) row = halo + c * W;
            else if (yr >= H && has_down) row = halo + (3 + c) * W;
            float v[6];
            const f32x4 mid = *reinterpret_cast<const f32x4*>(row + x0);
            v[0] = row[max(x0 - 1, 0)];
            v[1] = mid[0]; v[2] = mid[1]; v[3] = mid[2]; v[4] = mid[3];
            v[5] = row[min(x0 + 4, W - 1)];
            // Normalize first (true division, like the reference): the replicate pad then sees normalised values
#pragma unroll
            for (int j = 0; j < 6; ++j) v[j] = (v[j] - kMean[c]) / kStd[c];
#pragma unroll
            for (int j = 0; j < 5; ++j) pair[c][ky][j] = f32x2{v[j], v[j + 1]};
        }
    }
    unsigned int amax = 0;
    for (int co = blockIdx.y * cpt, co_end = co + cpt; co < co_end; ++co) {
        f32x2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float wk = w[co * 27 + (c * 3 + ky) * 3 + kx];
                    const f32x2 ww = {wk, wk};
                    lo = __builtin_elementwise_fma(ww, pair[c][ky][kx], lo);
                    hi = __builtin_elementwise_fma(ww, pair[c][ky][kx + 2], hi);
                }
        const float bias = b[co];
        f32x4 r = {lo[0] + bias, lo[1] + bias, hi[0] + bias, hi[1] + bias};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[i] = fmaxf(r[i], 0.f);
            amax = max(amax, abs_bits(r[i]));
        }
        *reinterpret_cast<f32x4*>(out + (size_t)co * HW + (size_t)y * W + x0) = r;
    }
    if (out_amax) amax_commit(amax, out_amax);
}

// ---- conv1_1 forward + relu1_1's Gram matrix and mean in ONE pass (round 4; opt-in: ST_CONV1_GRAM=1) -------------------
// relu1_1 (64 channels at full resolution) is the largest tap: its Gram kernel re-reads 64 HW floats that this kernel has
// just had in registers (2048^2: 365 us isolated, and exposed - a persistent convolution workgroup never overlaps it,
// profiles/r02_side_kernels.md).  Here a workgroup (4 waves) walks through blocks of 256 pixels (64 groups of 4 along a
// row); wave w computes output channels 16 w .. 16 w + 15 of the block exactly as conv_first_fwd4_kernel does (same FMA
// order: the map is bit-identical), and every value is scaled by the BLOCK's own power of two - from the bound
// max_co sum_k |w_co,k| x (the block's largest normalised input) + max |b|, known before the first output exists, so a
// channel's four pixels go to LDS as they are produced (no register copy of the 64 outputs: 2 workgroups per CU) and a
// block keeps fp16x3's 22 bits relative to ITS bound rather than the tensor's - split into two fp16 planes in LDS, and the 64 x 64 partial Gram matrix of the block goes through v_mfma_f32_32x32x16_f16
// (h0 h0^T + h0 h1^T + h1 h0^T, the stand-alone kernel's products) on the otherwise idle matrix pipe: each wave takes a
// quarter of the block's pixels (K) for the three tile pairs (0,0), (0,1), (1,1).  Accumulators persist over the
// workgroup's blocks (rescaled by an exact power of two when a block's scale differs); at the end the four K quarters are
// combined in a fixed order, the diagonal tiles mirrored from their upper triangle and (0,1) written with its transpose -
// exactly symmetric, like the stand-alone kernel - into partial[workgroup] / partial_sum[workgroup] of the tap's Gram
// workspace, which gram_finalize_kernel reduces as if the workgroups were its K splits.
constexpr int FGP = 264;                           // halves per staged channel row: 256 pixels + 8 (528 B: conflict-free b128)
typedef _Float16 fg_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 fg_h4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void conv_first_fwd_gram_kernel(const float* __restrict__ image, const float* __restrict__ w,
                                                                  const float* __restrict__ b, float* __restrict__ out, int H,
                                                                  int W, unsigned int* out_amax, float* __restrict__ partial,
                                                                  float* __restrict__ partial_sum, int nblocks, float w_l1max,
                                                                  float b_max) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) _Float16 planes[2][64][FGP];        // 67 584 B; reused for the final reduction
    __shared__ unsigned int wmax[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int HW = H * W, gpr = W >> 2, ngroups = H * gpr;
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float rs[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) rs[c] = 0.f;
    int e_run = 0;
    bool have = false;
    unsigned int amax_all = 0;
    for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int gi = blk * 64 + lane;
        const bool valid = gi < ngroups;
        const int grp = valid ? gi : ngroups - 1;                 // (lanes past the end redo the last group: identical stores)
        const int y = grp / gpr, x0 = (grp - y * gpr) * 4;
        f32x2 pair[3][3][5];
        float vmax = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int yy = min(max(y + ky - 1, 0), H - 1);
                const float* row = image + (size_t)c * HW + (size_t)yy * W;
                float v[6];
                const f32x4 mid = *reinterpret_cast<const f32x4*>(row + x0);
                v[0] = row[max(x0 - 1, 0)];
                v[1] = mid[0]; v[2] = mid[1]; v[3] = mid[2]; v[4] = mid[3];
                v[5] = row[min(x0 + 4, W - 1)];
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    v[j] = (v[j] - kMean[c]) / kStd[c];
                    vmax = fmaxf(vmax, fabsf(v[j]));
                }
#pragma unroll
                for (int j = 0; j < 5; ++j) pair[c][ky][j] = f32x2{v[j], v[j + 1]};
            }
        }
        // the block's scale from an a-priori bound of its outputs (see the header)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
        if (lane == 0) wmax[wave] = __builtin_bit_cast(unsigned int, vmax);
        __syncthreads();                                          // (also: the previous block's operand reads are done)
        const float vblock = __builtin_bit_cast(float, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
        const int e = scale_exp(__builtin_bit_cast(unsigned int, w_l1max * vblock + b_max));
        if (have && e != e_run) {                                 // exact: a power of two
            const float f = pow2f(2 * (e - e_run));
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] *= f;
        }
        e_run = e;
        have = true;
        const float scale = pow2f(e);
        unsigned int amax = 0;
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            // (unrolled for the register-resident row sums; the barrier keeps the 16 x 28 weight scalars from being hoisted
            // in front of the loop, which spilled 880 SGPRs)
            asm volatile("" ::: "memory");
            const int co = wave * 16 + cc;
            f32x2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float wk = w[co * 27 + (c * 3 + ky) * 3 + kx];
                        const f32x2 ww = {wk, wk};
                        lo = __builtin_elementwise_fma(ww, pair[c][ky][kx], lo);
                        hi = __builtin_elementwise_fma(ww, pair[c][ky][kx + 2], hi);
                    }
            const float bias = b[co];
            f32x4 r = {lo[0] + bias, lo[1] + bias, hi[0] + bias, hi[1] + bias};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r[i] = fmaxf(r[i], 0.f);
                amax = max(amax, abs_bits(r[i]));
            }
            *reinterpret_cast<f32x4*>(out + (size_t)co * HW + (size_t)y * W + x0) = r;
            if (!valid) r = f32x4{0.f, 0.f, 0.f, 0.f};            // a duplicated group must not enter the moments twice
            rs[cc] += (r[0] + r[1]) + (r[2] + r[3]);
            fg_h4 h0, h1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sv = r[i] * scale;
                const _Float16 a = (_Float16)sv;
                h0[i] = a;
                h1[i] = (_Float16)(sv - (float)a);
            }
            *reinterpret_cast<fg_h4*>(&planes[0][co][4 * lane]) = h0;
            *reinterpret_cast<fg_h4*>(&planes[1][co][4 * lane]) = h1;
        }
        amax_all = max(amax_all, amax);
        __syncthreads();
        // this wave's K quarter: pixels [64 wave, 64 wave + 64) of the block, 4 steps of 16
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = wave * 64 + ks * 16 + 8 * half;
            const fg_h8 a0 = *reinterpret_cast<const fg_h8*>(&planes[0][l31][k]);
            const fg_h8 a1 = *reinterpret_cast<const fg_h8*>(&planes[1][l31][k]);
            const fg_h8 c0 = *reinterpret_cast<const fg_h8*>(&planes[0][32 + l31][k]);
            const fg_h8 c1 = *reinterpret_cast<const fg_h8*>(&planes[1][32 + l31][k]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, a0, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, a1, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, a0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, c0, acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, c1, acc[1], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, c0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c0, c0, acc[2], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c0, c1, acc[2], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c1, c0, acc[2], 0, 0, 0);
        }
    }
    if (out_amax) amax_commit(amax_all, out_amax);
    // ---- the workgroup's partial moments ----
    __syncthreads();                                              // the last block's operand reads are done
    float* red = reinterpret_cast<float*>(&planes[0][0][0]);      // [4 waves][3 tiles][16][64] floats = 48 KB
    const float unscale = have ? pow2f(-2 * e_run) : 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave * 3 + t) * 16 + r) * 64 + lane] = acc[t][r] * unscale;
    __syncthreads();
    float* pout = partial + (size_t)blockIdx.x * 4096;
    if (wave < 3) {
        const int t = wave;
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float q0 = red[((0 * 3 + t) * 16 + r) * 64 + lane], q1 = red[((1 * 3 + t) * 16 + r) * 64 + lane];
            const float q2 = red[((2 * 3 + t) * 16 + r) * 64 + lane], q3 = red[((3 * 3 + t) * 16 + r) * 64 + lane];
            v[r] = (q0 + q1) + (q2 + q3);
        }
        if (t == 1) {                                             // rows 0..31 x cols 32..63, and the transpose
#pragma unroll
            for (int r = 0; r < 16; ++r) pout[(size_t)((r & 3) + 8 * (r >> 2) + 4 * half) * 64 + 32 + l31] = v[r];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                *reinterpret_cast<f32x4*>(pout + (size_t)(32 + l31) * 64 + 8 * q + 4 * half) = x;
            }
        }
        // diagonal tiles: h0 h1^T + h1 h0^T adds its two cross products in the opposite order at (i, j) and (j, i) - the
        // upper triangle is written and mirrored.  Scratch behind the reduction buffer: 2 x [32][33] floats.
        float* tri = red + 4 * 3 * 16 * 64 + (t == 2 ? 32 * 33 : 0);
        if (t != 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) tri[((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + l31] = v[r];
        }
        __builtin_amdgcn_wave_barrier();
        if (t != 1) {
            const int o = t == 2 ? 32 : 0;
            for (int idx = lane; idx < 32 * 32; idx += 64) {
                const int row = idx >> 5, col = idx & 31;
                pout[(size_t)(o + row) * 64 + o + col] = row <= col ? tri[row * 33 + col] : tri[col * 33 + row];
            }
        }
    }
    // row sums of this wave's 16 channels
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) {
        float v = rs[cc];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) partial_sum[(size_t)blockIdx.x * 64 + wave * 16 + cc] = v;
    }
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
// (8 per pass - rounds 1 ... 3 - held 40 staged values per thread: 225 registers, two waves per SIMD, nothing to cover the
// LDS and scalar-load latencies of the channel loop.  Round 4, same box, launch + fold from bench.py's roofline_hbm:
// 8 -> 4 -> 2 channels per pass: 512^2 60.7 -> 40.3 -> 38.2 us, 2048^2 ~390 -> 328 -> 312 us, 128^2 29.5 -> 18.6 -> 18.7 us.)
constexpr int FC = 2;                              // output channels of conv1_1 staged per pass
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
                                                            int has_up, int has_down, int cslice) {
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

    // accumulators of pixel pairs (x, x + 1) and (x + 2, x + 3): the 108 FMAs per staged channel become 54 packed ones
    // (v_pk_fma_f32; at one FMA per lane and cycle the 14.5 GFLOP of a 2048^2 image were 185 us of VALU time).  Every
    // accumulator still sees its taps in the same order: bit-identical to the scalar form.
    f32x2 acc2[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) acc2[h][0] = acc2[h][1] = acc2[h][2] = f32x2{0.f, 0.f};
    // this workgroup's slice of conv1_1's output channels: [blockIdx.y cslice, + cslice) -> partial plane blockIdx.y
    const int cb0 = blockIdx.y * cslice, cb1 = cb0 + cslice;
    load_pass(cb0);
    store_pass(0);
    __syncthreads();
    for (int cb = cb0, pass = 0; cb < cb1; cb += FC, ++pass) {
        const int buf = pass & 1;
        const bool more = cb + FC < cb1;
        if (more) load_pass(cb + FC);
#pragma unroll
        for (int c = 0; c < FC; ++c) {
            const float* wc = w + (cb + c) * 27;
            // window rows ty .. ty + 2 (gradient rows y - 1 .. y + 1), staged columns 4 tx .. 4 tx + 5 (x - 1 .. x + 4);
            // pair[r][o] = columns (o, o + 1) of the window row
            f32x2 pair[3][5];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float* row = &tile[buf][c][ty + r][4 * tx];
                const f32x4 a = *reinterpret_cast<const f32x4*>(row);
                const f32x2 b = *reinterpret_cast<const f32x2*>(row + 4);
                pair[r][0] = f32x2{a[0], a[1]}; pair[r][1] = f32x2{a[1], a[2]}; pair[r][2] = f32x2{a[2], a[3]};
                pair[r][3] = f32x2{a[3], b[0]}; pair[r][4] = f32x2{b[0], b[1]};
            }
            // exactly one tap links each of the 9 neighbouring outputs to a padded position
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int k = (1 - dy) * 3 + (1 - dx);
                    const f32x2 w0 = {wc[k], wc[k]}, w1 = {wc[9 + k], wc[9 + k]}, w2 = {wc[18 + k], wc[18 + k]};
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f32x2 g = pair[1 + dy][2 * h + 1 + dx];
                        acc2[h][0] = __builtin_elementwise_fma(w0, g, acc2[h][0]);
                        acc2[h][1] = __builtin_elementwise_fma(w1, g, acc2[h][1]);
                        acc2[h][2] = __builtin_elementwise_fma(w2, g, acc2[h][2]);
                    }
                }
        }
        if (more) store_pass(buf ^ 1);
        __syncthreads();
    }
    if (y <= py1) {
        const size_t plane = (size_t)(py1 - py0 + 1) * (W + 2);
        float* part = dp + (size_t)blockIdx.y * 3 * plane;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (x + j > W) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c) part[c * plane + (size_t)(y - py0) * (W + 2) + (x + j + 1)] = acc2[j >> 1][c][j & 1];
        }
    }
}

// grad_image[c][y][x] (+)= (sum of dP over the padded positions that replicate (y, x)) / std[c]
// UPDATE: ... and st_plan_step's Adam + clamp + EMA update on the element just finished (FoldUpdate), with the update kernel's
// tail (the losses' total, the next pass's operand bounds cleared): the iteration's last three launches in one
template <int PARTS, bool UPDATE>
__global__ __launch_bounds__(256) void conv_first_fold_kernel(const float* __restrict__ dp, float* __restrict__ gimg,
                                                              int H, int W, int accumulate, int has_up,
                                                              int has_down, FoldUpdate upd) {
    if constexpr (UPDATE) adam_tail(upd.tail);
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
            for (int xx = xs; xx <= xe; ++xx) {
                // the channel slices' partial sums, slice 0 first (one slice: the value itself); PARTS is a template
                // argument so that the loads of all slices are in flight together
                const float* at = dp + c * plane + (size_t)(yy - py0) * (W + 2) + (xx + 1);
                float part[PARTS];
#pragma unroll
                for (int s = 0; s < PARTS; ++s) part[s] = at[(size_t)s * 3 * plane];
                float t = part[0];
#pragma unroll
                for (int s = 1; s < PARTS; ++s) t += part[s];
                sum += t;
            }
        float v = sum / kStd[c];                                   // backward of Normalize
        if (accumulate) v += gimg[(size_t)c * HW + pix];
        gimg[(size_t)c * HW + pix] = v;
        if constexpr (UPDATE) {
            const size_t i = (size_t)c * HW + pix;
            float m = upd.exp_avg[i], s2 = upd.exp_avg_sq[i], px = upd.image[i], e = upd.ema[i];
            adam_clamp_ema_element(v, m, s2, px, e, upd.sc);
            upd.exp_avg[i] = m;
            upd.exp_avg_sq[i] = s2;
            upd.image[i] = px;
            upd.ema[i] = e;
        }
    }
}

}  // namespace

int launch_conv_first_fwd(const float* image, const float* w, const float* b, float* out, int height,
                          int width, hipStream_t stream, const float* halo, int has_up, int has_down,
                          unsigned int* out_amax) {
    static Option wide_opt("ST_CONV1_WIDE", 1);          // 0: the one-pixel kernel everywhere (A/B, bit-identity test)
    const bool wide = wide_opt.get() && width % 4 == 0 && width >= 8 &&
                      ((reinterpret_cast<uintptr_t>(image) | reinterpret_cast<uintptr_t>(out) |
                        reinterpret_cast<uintptr_t>(halo)) & 15) == 0;
    // A thread's 64 output channels are one serial chain (27 scalar weight loads + 27 FMAs + a store each): with one
    // workgroup per CU or fewer - every scale of the default run: 16 workgroups at 128^2, 256 at 512^2 - the launch
    // takes that chain's latency whatever the image size (23 / 43 / 38 us at 181^2 / 362^2 / 512^2, round-4 traces).
    // blockIdx.y splits the channels (down to 4 per thread) until the launch has ~4 waves per SIMD; the window is
    // reloaded and renormalised per part, every output keeps its own FMA order (bit-identical).  ST_CONV1_CO_SPLIT=0: off.
    static Option split_opt("ST_CONV1_CO_SPLIT", 1);
    const long long threads = wide ? (long long)height * (width / 4) : (long long)height * width;
    int cpt = 64;
    if (split_opt.get())
        while (cpt > 4 && ((threads + 63) / 64) * (64 / cpt) < 4096) cpt >>= 1;
    const dim3 grid(ceil_div((int)threads, 256), 64 / cpt);
    if (wide) {
        hipLaunchKernelGGL(conv_first_fwd4_kernel, grid, dim3(256), 0, stream, image, w, b, out, height, width, halo, has_up,
                           has_down, out_amax, cpt);
    } else {
        hipLaunchKernelGGL(conv_first_fwd_kernel, grid, dim3(256), 0, stream, image, w, b, out, height, width, halo, has_up,
                           has_down, out_amax, cpt);
    }
    ST_LAUNCH_CHECK();
    return 0;
}


// conv1_1 forward that also leaves relu1_1's partial moments in `partial` / `partial_sum` ([*splits][64][64], [*splits][64])
bool conv_first_gram_applies(int height, int width, const float* image, const float* out, int max_splits) {
    // OFF by default: measured a wash.  2048^2: this launch 519 us against 297 us for conv1_1 alone + 365 us for the Gram
    // kernel - 143 us less kernel time, but all of it now on the trunk's stream, where the stand-alone Gram kernel had run
    // beside conv1_2 on a side stream: 51.0 vs 51.0 it/s (1024^2 180.5 vs 179.8, 512^2 431.5 vs 431.1, 2896 x 2172 32.8 vs
    // 32.9; same box, two rounds).  The kernel is VALU-bound at 2 waves per SIMD (196 registers, 67 KB of LDS).
    static Option on("ST_CONV1_GRAM", 0);
    return on.get() != 0 && width % 4 == 0 && width >= 8 && max_splits >= 8 &&
           ((reinterpret_cast<uintptr_t>(image) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
}

int launch_conv_first_fwd_gram(const float* image, const float* w, const float* b, float* out, int height, int width,
                               hipStream_t stream, unsigned int* out_amax, float* partial, float* partial_sum, int max_splits,
                               int* splits, float w_l1max, float b_max) {
    ST_REQUIRE(conv_first_gram_applies(height, width, image, out, max_splits), "conv1_1 + Gram: unsupported shape");
    const int nblocks = ceil_div(height * (width / 4), 64);
    int grid = nblocks < 512 ? nblocks : 512;                      // two workgroups per CU
    if (grid > max_splits) grid = max_splits;
    hipLaunchKernelGGL(conv_first_fwd_gram_kernel, dim3(grid), dim3(256), 0, stream, image, w, b, out, height, width, out_amax,
                       partial, partial_sum, nblocks, w_l1max, b_max);
    ST_LAUNCH_CHECK();
    *splits = grid;
    return 0;
}

// A workgroup walks through its 64 x 16 positions' 64 gradient channels in 8 dependent passes (stage 8 channels through
// LDS, 432 packed FMAs per thread): ~45 us whatever the image size (45 / 46 / 50 us at 181^2 / 362^2 / 512^2, round-4
// traces), and a 128^2 image has 27 such tiles.  Where a launch would not fill the chip the channels are cut into up to 8
// slices (blockIdx.y), each slice's partial dP goes to a plane of its own and the fold kernel adds the planes in slice
// order - deterministic; the sum over channels is grouped by slices instead of running through all 64 (parity is checked
// against the reference at the 1e-3 gradient bar, like every other kernel's order of summation).  Plans ask with the
// GLOBAL image height, so a strip cuts exactly as the whole image does (the strip tests compare gradients bit for bit).
// ST_CONV1_DGRAD_SPLIT=0: one slice everywhere.
int conv_first_dgrad_parts(int height, int width) {
    static Option split_opt("ST_CONV1_DGRAD_SPLIT", 1);
    if (!split_opt.get()) return 1;
    const long long tiles = (long long)ceil_div(width + 2, FTX) * ceil_div(height + 2, FTY);
    int parts = 1;
    while (parts < 8 && tiles * parts < 512) parts *= 2;
    return parts;
}

int launch_conv_first_dgrad(const float* grad_out, const float* relu_out, const float* w, float* grad_image,
                            float* dp_scratch, int height, int width, int accumulate, hipStream_t stream,
                            const float* ghalo, int has_up, int has_down, int parts, const FoldUpdate* update) {
    const int rows = height + (has_up ? 0 : 1) + (has_down ? 0 : 1);
    const int blocks = ceil_div(width + 2, FTX) * ceil_div(rows, FTY);
    ST_REQUIRE(parts == 1 || parts == 2 || parts == 4 || parts == 8, "conv1_1 data gradient: %d channel slices", parts);
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(blocks, parts), dim3(256), 0, stream, grad_out, relu_out, w, dp_scratch, height, width,
                           ghalo, has_up, has_down, 64 / parts);
    };
    if (relu_out && ghalo) launch(conv_first_dp_kernel<true, true>);
    else if (relu_out) launch(conv_first_dp_kernel<true, false>);
    else if (ghalo) launch(conv_first_dp_kernel<false, true>);
    else launch(conv_first_dp_kernel<false, false>);
    ST_LAUNCH_CHECK();
    auto fold = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(ceil_div(height * width, 256)), dim3(256), 0, stream, dp_scratch, grad_image, height,
                           width, accumulate, has_up, has_down, update ? *update : FoldUpdate{});
    };
    if (update) {
        if (parts == 1) fold(conv_first_fold_kernel<1, true>);
        else if (parts == 2) fold(conv_first_fold_kernel<2, true>);
        else if (parts == 4) fold(conv_first_fold_kernel<4, true>);
        else fold(conv_first_fold_kernel<8, true>);
    } else {
        if (parts == 1) fold(conv_first_fold_kernel<1, false>);
        else if (parts == 2) fold(conv_first_fold_kernel<2, false>);
        else if (parts == 4) fold(conv_first_fold_kernel<4, false>);
        else fold(conv_first_fold_kernel<8, false>);
    }
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
