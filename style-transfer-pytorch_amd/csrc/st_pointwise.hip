// HBM-bound pointwise kernels and small reductions of the hot loop.  All arithmetic here follows
// the reference's operation order with FP contraction disabled, so the only rounding differences
// against the CPU path are in reduction order.
#include "st_common.h"

namespace st {
namespace {

// ------------------------------------------------------------------------------------------------
__global__ void fill_kernel(float* p, long long n, float v) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = v;
}
__global__ void identity_kernel(float* p, int n) {
    const long long nn = (long long)n * n;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nn; i += (long long)gridDim.x * 256)
        p[i] = ((int)(i / n) == (int)(i % n)) ? 1.f : 0.f;
}

// cov = srm - mean mean^T + eps * I      (StyleLossW2.srm_to_cov + eye_like * eps, style_transfer.py:156,170-177)
__global__ void cov_kernel(const float* __restrict__ mean, const float* __restrict__ srm,
                           float* __restrict__ cov, int n, float eps) {
#pragma clang fp contract(off)
    const long long nn = (long long)n * n;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nn; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / n), c = (int)(i % n);
        const float outer = mean[r] * mean[c];
        const float d = srm[i] - outer;
        cov[i] = d + ((r == c) ? eps : 0.f);
    }
}

// two-pass Frobenius norm: partial sums of squares, then fixed-order combine + sqrt
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ a, long long count,
                                                            float* __restrict__ partials) {
    __shared__ float scratch[4];
    float s = 0.f;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const float v = a[i];
        s = fmaf(v, v, s);
    }
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(64) void sqrt_of_sum_kernel(const float* __restrict__ partials, int nparts,
                                                         float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) s += partials[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) out[0] = sqrtf(s);
}

__global__ void div_by_dev_scalar_kernel(const float* __restrict__ a, const float* __restrict__ scalar,
                                         float* __restrict__ y, long long count) {
    const float d = scalar[0];
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (long long)gridDim.x * 256)
        y[i] = a[i] / d;
}
__global__ void scaled_identity_div_kernel(const float* __restrict__ diag_value,
                                           const float* __restrict__ scalar, float* __restrict__ q, int n) {
    const float v = diag_value[0] / scalar[0];
    const long long nn = (long long)n * n;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nn; i += (long long)gridDim.x * 256)
        q[i] = ((int)(i / n) == (int)(i % n)) ? v : 0.f;
}

// The two recurrences' prologues in two launches instead of four (they sit on the critical path of the whole
// iteration: the backward pass waits for relu5_1's chain): sums of squares in <= 256 partials (16-byte loads),
// then every workgroup re-combines the partials in the same fixed order, divides its share of `a` by the norm
// and fills the companion matrix.
__global__ __launch_bounds__(256) void sumsq_partial4_kernel(const float* __restrict__ a, long long count,
                                                             float* __restrict__ partials) {
    __shared__ float scratch[4];
    float s = 0.f;
    const long long n4 = count / 4;          // count = n * n with n % 64 == 0
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = a4[i];
        s = fmaf(v[0], v[0], s); s = fmaf(v[1], v[1], s); s = fmaf(v[2], v[2], s); s = fmaf(v[3], v[3], s);
    }
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
// mode 0: second = I (sqrtm.py:16-20);  1: second = g / norm;  2: second = (gdiag / norm) I (sqrtm.py:38-41);
// 3: second = (3I - a / norm) / 2, the first step's t of sqrtm.py:22 for z = I (see ns_sqrt_forward)
__global__ __launch_bounds__(256) void ns_prepare_kernel(const float* __restrict__ a,
                                                         const float* __restrict__ partials, int nparts,
                                                         float* __restrict__ norm_out, float* __restrict__ a_scaled,
                                                         const float* __restrict__ g, const float* __restrict__ gdiag,
                                                         float* __restrict__ second, int n, int mode, W2LossJob loss) {
    __shared__ float scratch[4];
    __shared__ float norm_sh;
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i];
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) {
        norm_sh = sqrtf(s);
        if (blockIdx.x == 0) norm_out[0] = norm_sh;
    }
    __syncthreads();
    const float d = norm_sh;
    // (a W2 job rides along: the seed gdiag is its own, written by w2_loss_block below for the chain's later launches)
    const float dv = (mode == 2) ? (loss.loss_out ? w2_gdiag(loss) : gdiag[0]) / d : 1.f;
    const long long nn = (long long)n * n;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < nn; i += (long long)gridDim.x * 256) {
        const float y = a[i] / d;
        a_scaled[i] = y;
        const bool diag = (int)(i / n) == (int)(i % n);
        second[i] = (mode == 3) ? ((diag ? 3.f : 0.f) - y) * 0.5f : (mode == 1) ? g[i] / d : (diag ? dv : 0.f);
    }
    if (loss.loss_out && blockIdx.x == 0) w2_loss_block(loss, scratch);
}

// ------------------------------------------------------------------------------------------------
// ContentLossMSE (style_transfer.py:119-126) under Scale(weight): value and gradient in one pass.
// In-kernel final reduction ("last block"): every block writes its partial sums, then takes a ticket; the block
// that draws the last one sums all partials in index order (deterministic for a given grid) and writes the loss.
// Saves the second launch - a single wave that, next to the trunk's persistent convolution workgroups, used to
// wait up to 0.7 ms for a free CU at 2048^2.  Release / acquire at agent scope as MI355X_MICROARCH.md prescribes
// (the asm waitcnt keeps the ticket behind the write-back of the partials).
struct LastBlock {
    unsigned int* ticket;     // device word, zero between launches (the last block resets it); nullptr = two-launch form
};
__device__ __forceinline__ bool last_block_arrives(const LastBlock& lb, bool* shared_flag) {
    if (lb.ticket == nullptr) return false;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int prev = atomicAdd(lb.ticket, 1u);
        const bool last = (prev == gridDim.x - 1);
        if (last) *lb.ticket = 0u;
        *shared_flag = last;
    }
    __syncthreads();
    const bool last = *shared_flag;
    // every thread of the last block reads other blocks' partials: each needs its own agent-scope acquire (a CU's
    // vector L1 is never refreshed by another CU's stores; one lane's fence is not a guarantee for the other waves)
    if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return last;
}

__global__ __launch_bounds__(256) void content_mse_kernel(const float* __restrict__ feat,
                                                          const float* __restrict__ target, long long count,
                                                          float weight, float norm, float* __restrict__ grad,
                                                          float* __restrict__ partials, LastBlock lb,
                                                          float final_count, float* __restrict__ loss_out) {
#pragma clang fp contract(off)
    __shared__ float scratch[4];
    __shared__ bool is_last;
    float s = 0.f;
    if ((count & 3) == 0) {           // 16-byte accesses: three streams of 4 B per element, HBM-bound
        const long long n4 = count >> 2;
        const f32x4* f4 = reinterpret_cast<const f32x4*>(feat);
        const f32x4* t4 = reinterpret_cast<const f32x4*>(target);
        f32x4* g4 = reinterpret_cast<f32x4*>(grad);
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
            const f32x4 f = f4[i], t = t4[i];
            f32x4 g;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float d = f[k] - t[k];
                s += d * d;
                g[k] = (norm * d) * weight;      // mse_loss_backward: (2/numel) * (x - t) * grad_out
            }
            g4[i] = g;
        }
    } else {
        for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
            const float d = feat[i] - target[i];
            s += d * d;
            grad[i] = (norm * d) * weight;
        }
    }
    s = block_sum_256(s, scratch);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
    if (!last_block_arrives(lb, &is_last)) return;
    float t = 0.f;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += partials[i];
    t = block_sum_256(t, scratch);
    if (threadIdx.x == 0) loss_out[0] = (t / final_count) * weight;
}
__global__ __launch_bounds__(64) void content_mse_final_kernel(const float* __restrict__ partials, int nparts,
                                                               float count, float weight,
                                                               float* __restrict__ loss_out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nparts; i += 64) s += partials[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss_out[0] = (s / count) * weight;
}

// W2 head scalars (StyleLossW2.forward, style_transfer.py:178-181) for one layer; single workgroup.
__global__ __launch_bounds__(256) void style_loss_value_kernel(W2LossJob job) {
    __shared__ float scratch[4];
    w2_loss_block(job, scratch);
}

// One workgroup per row c of dcov = g + (weight/n) I:
//   ssym[c][d] = (dcov[c][d] + dcov[d][c]) / npix
//   bvec[c]    = (2 (weight/n) (mu - mu_t)[c] - sum_d (dcov[c][d] + dcov[d][c]) mu[d]) / npix
__global__ __launch_bounds__(256) void style_grad_finish_kernel(const float* __restrict__ g,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ mean_t, int n,
                                                                float weight, float npix,
                                                                float* __restrict__ ssym,
                                                                float* __restrict__ bvec,
                                                                unsigned int* __restrict__ ssym_amax) {
#pragma clang fp contract(off)
    __shared__ float scratch[4];
    const int c = blockIdx.x;
    const float wn = weight / (float)n;
    float dot = 0.f;
    unsigned int amax = 0;
    for (int d = threadIdx.x; d < n; d += 256) {
        float a = g[(size_t)c * n + d], b = g[(size_t)d * n + c];
        if (d == c) { a += wn; b += wn; }
        const float sym = a + b;
        const float sv = sym / npix;
        ssym[(size_t)c * n + d] = sv;
        const unsigned int bits = abs_bits(sv);
        amax = bits > amax ? bits : amax;
        dot += sym * mean[d];
    }
    if (ssym_amax) amax_commit(amax, ssym_amax);      // bound on max |Ssym| for the fp16x3 1x1 convolution
    dot = block_sum_256(dot, scratch);
    if (threadIdx.x == 0) bvec[c] = ((wn * 2.f) * (mean[c] - mean_t[c]) - dot) / npix;
}

// ------------------------------------------------------------------------------------------------
// TVLoss (style_transfer.py:184-195): value and gradient.  P(a, b) is the replicate-padded image in
// padded coordinates a in [-1, H], b in [-1, W].  Differences:
//   D1(a,b) = P(a,b+1) - P(a,b), D2(a,b) = P(a+1,b) - P(a,b)            on [0,H) x [0,W)
//   D3(i,j) = P(i,j) - P(i-1,j-1), D4(i,j) = P(i,j-1) - P(i-1,j)        on [0,H] x [0,W]
//   loss = 2 (mean(D1^2)/3 + mean(D2^2)/3 + mean(D3^2)/12 + mean(D4^2)/12)
// The gradient w.r.t. the padding ring folds back onto the nearest image pixel.
struct TVImage {
    const float* p;        // this channel of the local strip [H][W]
    int H, W;              // local rows, width
    int row0, Hg;          // global row of local row 0, global height
    const float* top;      // neighbour rows (this channel) or nullptr at the global border
    const float* bot;
    __device__ __forceinline__ float at(int a, int b) const {   // a = LOCAL row index, may be -1 / H
        b = min(max(b, 0), W - 1);
        if (a < 0) {
            if (top) return top[b];
            a = 0;
        } else if (a >= H) {
            if (bot) return bot[b];
            a = H - 1;
        }
        return p[(size_t)a * W + b];
    }
};

// gradient w.r.t. the padded image at local row a (global row a + row0), column b
__device__ __forceinline__ float tv_dP(const TVImage& im, int a, int b, float k1, float k3) {
#pragma clang fp contract(off)
    const int H = im.Hg, W = im.W;
    const int ag = a + im.row0;
    const float c = im.at(a, b);
    float g = 0.f;
    const bool row_in = (ag >= 0 && ag < H), col_in = (b >= 0 && b < W);
    if (row_in) {
        if (b >= 1 && b <= W) g += k1 * (c - im.at(a, b - 1));
        if (col_in) g -= k1 * (im.at(a, b + 1) - c);
    }
    if (col_in) {
        if (ag >= 1 && ag <= H) g += k1 * (c - im.at(a - 1, b));
        if (row_in) g -= k1 * (im.at(a + 1, b) - c);
    }
    if (ag >= 0 && b >= 0) g += k3 * (c - im.at(a - 1, b - 1));            // D3(a, b)
    if (ag <= H - 1 && b <= W - 1) g -= k3 * (im.at(a + 1, b + 1) - c);    // D3(a+1, b+1)
    if (ag >= 0 && b <= W - 1) g += k3 * (c - im.at(a - 1, b + 1));        // D4(a, b+1)
    if (ag <= H - 1 && b >= 0) g -= k3 * (im.at(a + 1, b - 1) - c);        // D4(a+1, b)
    return g;
}

// One pixel, any position (global borders: the padding ring's gradient folds onto it; strips: neighbour rows come
// from the halo): gradient + the squared differences this pixel owns.
__device__ __forceinline__ float tv_pixel_generic(const TVImage& im, int y, int x, int yg, int Hg, int W, float k1,
                                                  float k3, float& s1, float& s2, float& s3, float& s4) {
#pragma clang fp contract(off)
    // gradient: this pixel plus the GLOBAL padding-ring positions that replicate it
    float g = 0.f;
    for (int ry = -1; ry <= 1; ++ry) {
        if (ry != 0 && !((ry < 0 && yg == 0) || (ry > 0 && yg == Hg - 1))) continue;
        for (int rx = -1; rx <= 1; ++rx) {
            if (rx != 0 && !((rx < 0 && x == 0) || (rx > 0 && x == W - 1))) continue;
            g += tv_dP(im, y + ry, x + rx, k1, k3);
        }
    }
    // value: every difference is owned by exactly one pixel
    const float c = im.at(y, x);
    const float d1 = im.at(y, x + 1) - c, d2 = im.at(y + 1, x) - c;
    s1 += d1 * d1;
    s2 += d2 * d2;
    for (int iy = y; iy <= ((yg == Hg - 1) ? y + 1 : y); ++iy)
        for (int jx = x; jx <= ((x == W - 1) ? W : x); ++jx) {
            const float d3 = im.at(iy, jx) - im.at(iy - 1, jx - 1);
            const float d4 = im.at(iy, jx - 1) - im.at(iy - 1, jx);
            s3 += d3 * d3;
            s4 += d4 * d4;
        }
    return g;
}

// Two kernels.  tv_interior_kernel: HBM-bound streaming pass - reads the image once (the 3 x 3 neighbourhoods come from
// L1 / L2), writes the gradient once.  A thread owns 4 consecutive pixels of a row: three 16-byte loads + six edge
// scalars, one 16-byte store; rows are dealt to workgroups and a row's groups to the threads, so there is no division
// per element.  Its formula is tv_dP with every range condition true, in the same operation order: identical bits to
// the generic path.  It skips the first / last row and the first / last group of every row; tv_border_kernel takes
// those (and everything when the width is not a multiple of 4) with tv_pixel_generic - global borders fold the
// padding ring, strip borders read the neighbours' halo rows - and its last block finishes the four sums of BOTH
// kernels.  (Round 1: one pixel per thread of a 256-block grid-stride loop, nine tv_dP evaluations and two 64-bit
// divisions per pixel: 991 + 709 us in situ at 2048^2.  One kernel with both paths needed 125 registers: 141 us
// isolated; split: see profiles/r02_side_kernels.md.)
// VAR (ST_TV_VARIANT; profiles/r05_tv_hazard.md - the flaky TV term of round 4, root-caused in round 5):
//   1 = SHIPPED.  The four accumulators are pinned to registers of their own after every update (an empty asm statement), so
//       that the SLP vectoriser cannot pair (d1, d2) / (d3, d4) into packed-FP32 operations.
//   0 = the round-1 ... 4 code, kept as the REPRODUCER: the optimiser turns `d1 = M[j+2] - cc, d2 = D[j+1] - cc` of pixel j = 0
//       into ONE `v_pk_add_f32 vdst, vsrc0, vsrc1 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]` (both lanes subtract the HIGH half of
//       vsrc1).  On MI355X that instruction returned `vsrc0 - 0` in its LOW lane for lanes 48 - 63 of a wave - i.e. d1 = M[2]
//       instead of M[2] - M[1], identified to 1e-8 per thread (tools/tv_hazard_threads.py) - in 55 - 75 % of the FIRST
//       closures of a fresh plan when the kernel ran at the tail of a head stream beside the launch-per-product
//       Newton-Schulz chains, never on a warm plan, never at the start of the iteration.  Neither aligned-only loads, nor a
//       full s_waitcnt vmcnt(0) + 16 idle cycles before the first use, nor another block reduction changed the rate (variants
//       2, 4, 5, 6 of the investigation); the variant WITHOUT that one instruction never failed (0 / 144 closures).  It is the
//       only packed-FP32 instruction with a cross-half op_sel in the whole library; build.py now refuses to build a kernel
//       that contains one (hazard guard), this reproducer and its dumping twin excepted.
//   3 = variant 0 + every thread's four accumulators and its group count written to `dbg` ([blocks][256][5]).
template <int VAR>
__global__ __launch_bounds__(256) void tv_interior_kernel(const float* __restrict__ image, int H, int W, float k1,
                                                          float k3, float* __restrict__ grad,
                                                          float* __restrict__ partials, float* __restrict__ dbg) {
#pragma clang fp contract(off)
    __shared__ float scratch[4];
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    int visited = 0;
    const int gpr = W >> 2;                                  // groups of 4 pixels per row
    int tpr = 256;                                           // threads sharing a row: power of two covering it
    while (tpr > 1 && (tpr >> 1) >= gpr) tpr >>= 1;
    const int rpb = 256 / tpr;
    const int lr = threadIdx.x / tpr, gx = threadIdx.x % tpr;
    const int rows = 3 * H;
    for (int row0 = blockIdx.x * rpb; row0 < rows; row0 += gridDim.x * rpb) {
        const int row = row0 + lr;
        const int ch = row / H;
        const int y = row - ch * H;
        if (row >= rows || y < 1 || y > H - 2) continue;
        const float* rowp = image + (size_t)ch * H * W + (size_t)y * W;
        float* growp = grad + (size_t)ch * H * W + (size_t)y * W;
        for (int g4 = gx; g4 < gpr; g4 += tpr) {
            if (g4 == 0 || g4 == gpr - 1) continue;
            const float* c = rowp + 4 * g4;
            const f32x4 up = *reinterpret_cast<const f32x4*>(c - W);
            const f32x4 mid = *reinterpret_cast<const f32x4*>(c);
            const f32x4 dn = *reinterpret_cast<const f32x4*>(c + W);
            const float U[6] = {c[-W - 1], up[0], up[1], up[2], up[3], c[-W + 4]};
            const float M[6] = {c[-1], mid[0], mid[1], mid[2], mid[3], c[4]};
            const float D[6] = {c[W - 1], dn[0], dn[1], dn[2], dn[3], c[W + 4]};
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float cc = M[j + 1];
                float g = 0.f;
                g += k1 * (cc - M[j]);
                g -= k1 * (M[j + 2] - cc);
                g += k1 * (cc - U[j + 1]);
                g -= k1 * (D[j + 1] - cc);
                g += k3 * (cc - U[j]);               // D3(a, b)
                g -= k3 * (D[j + 2] - cc);           // D3(a+1, b+1)
                g += k3 * (cc - U[j + 2]);           // D4(a, b+1)
                g -= k3 * (D[j] - cc);               // D4(a+1, b)
                o[j] = g;
                const float d1 = M[j + 2] - cc, d2 = D[j + 1] - cc, d3 = cc - U[j], d4 = M[j] - U[j + 1];
                s1 += d1 * d1;
                s2 += d2 * d2;
                s3 += d3 * d3;
                s4 += d4 * d4;
                if (VAR == 1) asm volatile("" : "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4));
            }
            *reinterpret_cast<f32x4*>(growp + 4 * g4) = o;
            if (VAR == 3) ++visited;
        }
    }
    if (VAR == 3) {
        float* d = dbg + ((size_t)blockIdx.x * 256 + threadIdx.x) * 5;
        d[0] = s1; d[1] = s2; d[2] = s3; d[3] = s4; d[4] = (float)visited;
    }
    s1 = block_sum_256(s1, scratch);
    s2 = block_sum_256(s2, scratch);
    s3 = block_sum_256(s3, scratch);
    s4 = block_sum_256(s4, scratch);
    if (threadIdx.x == 0) {
        partials[blockIdx.x * 4 + 0] = s1;
        partials[blockIdx.x * 4 + 1] = s2;
        partials[blockIdx.x * 4 + 2] = s3;
        partials[blockIdx.x * 4 + 3] = s4;
    }
}

// interior != 0: only the groups tv_interior_kernel skipped; partials holds `first` blocks of that kernel already and
// this one appends its own.  The last block sums all of them in index order (lb.ticket != nullptr).
__global__ __launch_bounds__(256) void tv_border_kernel(const float* __restrict__ image, int H, int W, float k1,
                                                        float k3, float* __restrict__ grad,
                                                        float* __restrict__ partials, StripInfo strip, int interior,
                                                        int first, LastBlock lb, float fin_n, float fin_n2,
                                                        float fin_weight, float* __restrict__ loss_out) {
#pragma clang fp contract(off)
    __shared__ float scratch[4];
    __shared__ bool is_last;
    const int Hg = strip.global_height;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;
    auto pixel = [&](int ch, int y, int x) __attribute__((always_inline)) {
        TVImage im{image + (size_t)ch * H * W, H, W, strip.row0, Hg,
                   (strip.halo && strip.has_up) ? strip.halo + (size_t)ch * W : nullptr,
                   (strip.halo && strip.has_down) ? strip.halo + (size_t)(3 + ch) * W : nullptr};
        grad[(size_t)ch * H * W + (size_t)y * W + x] = tv_pixel_generic(im, y, x, y + strip.row0, Hg, W, k1, k3, s1, s2, s3, s4);
    };
    if (interior) {
        const int gpr = W >> 2;
        const int per_ch = 2 * gpr + 2 * (H - 2);              // groups of the first / last row + both ends of the others
        const int total = 3 * per_ch * 4;                      // ... as pixels
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
            const int gi = i >> 2, j = i & 3;
            const int ch = gi / per_ch, k = gi - ch * per_ch;
            int y, g4;
            if (k < gpr) { y = 0; g4 = k; }
            else if (k < 2 * gpr) { y = H - 1; g4 = k - gpr; }
            else { y = 1 + ((k - 2 * gpr) >> 1); g4 = ((k - 2 * gpr) & 1) ? gpr - 1 : 0; }
            pixel(ch, y, 4 * g4 + j);
        }
    } else {
        const int total = 3 * H * W;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
            const int x = i % W, r = i / W;
            pixel(r / H, r % H, x);
        }
    }
    s1 = block_sum_256(s1, scratch);
    s2 = block_sum_256(s2, scratch);
    s3 = block_sum_256(s3, scratch);
    s4 = block_sum_256(s4, scratch);
    if (threadIdx.x == 0) {
        float* mine = partials + (size_t)(first + blockIdx.x) * 4;
        mine[0] = s1; mine[1] = s2; mine[2] = s3; mine[3] = s4;
    }
    if (!last_block_arrives(lb, &is_last)) return;
    const int nparts = first + (int)gridDim.x;
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nparts; i += 256)
        for (int k = 0; k < 4; ++k) t[k] += partials[i * 4 + k];
    for (int k = 0; k < 4; ++k) t[k] = block_sum_256(t[k], scratch);
    if (threadIdx.x == 0) {
        const float d1 = (t[0] / fin_n) / 3.f, d2 = (t[1] / fin_n) / 3.f;
        const float d3 = (t[2] / fin_n2) / 12.f, d4 = (t[3] / fin_n2) / 12.f;
        loss_out[0] = (2.f * (((d1 + d2) + d3) + d4)) * fin_weight;
    }
}
// sums[k] = fixed-order sum over the per-block partials (k = 0..width-1)
__global__ __launch_bounds__(64) void reduce_partials_kernel(const float* __restrict__ partials, int nparts,
                                                             int width, float* __restrict__ sums) {
    for (int k = 0; k < width; ++k) {
        float s = 0.f;
        for (int i = threadIdx.x; i < nparts; i += 64) s += partials[i * width + k];
        s = wave_sum(s);
        if (threadIdx.x == 0) sums[k] = s;
    }
}
__global__ void tv_from_sums_kernel(const float* __restrict__ s, float n, float n2, float weight,
                                    float* __restrict__ loss_out) {
#pragma clang fp contract(off)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float d1 = (s[0] / n) / 3.f, d2 = (s[1] / n) / 3.f;
        const float d3 = (s[2] / n2) / 12.f, d4 = (s[3] / n2) / 12.f;
        loss_out[0] = (2.f * (((d1 + d2) + d3) + d4)) * weight;
    }
}
__global__ void mse_from_sum_kernel(const float* __restrict__ s, float count, float weight,
                                    float* __restrict__ loss_out) {
#pragma clang fp contract(off)
    if (threadIdx.x == 0 && blockIdx.x == 0) loss_out[0] = (s[0] / count) * weight;
}
__global__ void div_by_scalar_kernel(const float* __restrict__ a, float d, float* __restrict__ y,
                                     long long count) {
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (long long)gridDim.x * 256)
        y[i] = a[i] / d;
}
__global__ __launch_bounds__(64) void tv_final_kernel(const float* __restrict__ partials, int nparts, float n,
                                                      float n2, float weight, float* __restrict__ loss_out) {
#pragma clang fp contract(off)
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < nparts; i += 64)
        for (int k = 0; k < 4; ++k) s[k] += partials[i * 4 + k];
    for (int k = 0; k < 4; ++k) s[k] = wave_sum(s[k]);
    if (threadIdx.x == 0) {
        const float d1 = (s[0] / n) / 3.f, d2 = (s[1] / n) / 3.f;
        const float d3 = (s[2] / n2) / 12.f, d4 = (s[3] / n2) / 12.f;
        loss_out[0] = (2.f * (((d1 + d2) + d3) + d4)) * weight;
    }
}

// `copy` (optional): the caller's 8-float result buffer, written by the same launch (a separate 32-byte device copy
// was a 16 us kernel at the very end of every iteration's critical path)
__global__ void sum_losses_kernel(float* l, float* copy) {
#pragma clang fp contract(off)
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 7; ++i) t = t + l[i];        // Python sum(): 0 + l0 + l1 + ... (SumLoss, :208)
        l[7] = t;
        if (copy) {
            for (int i = 0; i < 7; ++i) copy[i] = l[i];
            copy[7] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// torch.optim.Adam single-tensor step (torch/optim/adam.py:414-547, as configured at
// style_transfer.py:458) + image.clamp_(0, 1) (:485) + EMA.update (:250-253), one pass over 3HW.
__global__ __launch_bounds__(256) void adam_clamp_ema_kernel(float* __restrict__ image,
                                                             const float* __restrict__ grad,
                                                             float* __restrict__ exp_avg,
                                                             float* __restrict__ exp_avg_sq,
                                                             float* __restrict__ ema, long long count,
                                                             AdamScalars sc, AdamTail tail) {
#pragma clang fp contract(off)
    adam_tail(tail);
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < count; i += (long long)gridDim.x * 256) {
        const float g = grad[i];
        float m = exp_avg[i], v = exp_avg_sq[i], p = image[i], e = ema[i];
        adam_clamp_ema_element(g, m, v, p, e, sc);
        exp_avg[i] = m;
        exp_avg_sq[i] = v;
        image[i] = p;
        ema[i] = e;
    }
}

int grid_for(long long n, int cap = 4096) {
    long long b = (n + 255) / 256;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

int launch_fill(float* p, long long n, float v, hipStream_t s) {
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, n, v);
    ST_LAUNCH_CHECK();
    return 0;
}
int launch_identity(float* p, int n, hipStream_t s) {
    hipLaunchKernelGGL(identity_kernel, dim3(grid_for((long long)n * n)), dim3(256), 0, s, p, n);
    ST_LAUNCH_CHECK();
    return 0;
}
int launch_cov_from_moments(const float* mean, const float* srm, float* cov, int n, float eps, hipStream_t s) {
    hipLaunchKernelGGL(cov_kernel, dim3(grid_for((long long)n * n)), dim3(256), 0, s, mean, srm, cov, n, eps);
    ST_LAUNCH_CHECK();
    return 0;
}
int launch_frobenius(const float* a, long long count, float* out, hipStream_t s) {
    // out[2..] doubles as the partial buffer: callers pass a scalar slot followed by >= 64 spare floats
    const int blocks = grid_for(count, 32);
    float* partials = out + 2;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, s, a, count, partials);
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(sqrt_of_sum_kernel, dim3(1), dim3(64), 0, s, partials, blocks, out);
    ST_LAUNCH_CHECK();
    return 0;
}
int launch_sumsq_partials(const float* a, long long count, float* partials, int* nparts, hipStream_t s) {
    const int blocks = grid_for(count / 4, 256);
    hipLaunchKernelGGL(sumsq_partial4_kernel, dim3(blocks), dim3(256), 0, s, a, count, partials);
    ST_LAUNCH_CHECK();
    *nparts = blocks;
    return 0;
}
int launch_ns_prepare(const float* a, int n, float* norm_out, float* partials, float* a_scaled, const float* g,
                      const float* gdiag, float* second, hipStream_t s, bool first_t, const W2LossJob* loss,
                      int partials_ready) {
    const long long nn = (long long)n * n;
    int blocks = partials_ready;              // (the product that made `a` left its tiles' sums of squares: no launch here)
    if (blocks <= 0 && launch_sumsq_partials(a, nn, partials, &blocks, s)) return 1;
    const int mode = first_t ? 3 : g ? 1 : (gdiag ? 2 : 0);
    ST_REQUIRE(!loss || mode == 2, "ns prepare: a W2 job needs the diagonal form");
    hipLaunchKernelGGL(ns_prepare_kernel, dim3(grid_for(nn, 1024)), dim3(256), 0, s, a, partials, blocks, norm_out,
                       a_scaled, g, gdiag, second, n, mode, loss ? *loss : W2LossJob{});
    ST_LAUNCH_CHECK();
    return 0;
}
int launch_div_by_dev_scalar(const float* a, const float* scalar, float* y, long long count, hipStream_t s) {
    hipLaunchKernelGGL(div_by_dev_scalar_kernel, dim3(grid_for(count)), dim3(256), 0, s, a, scalar, y, count);
    ST_LAUNCH_CHECK();
    return 0;
}
int launch_scaled_identity_div(const float* diag_value, const float* scalar, float* q, int n, hipStream_t s) {
    hipLaunchKernelGGL(scaled_identity_div_kernel, dim3(grid_for((long long)n * n)), dim3(256), 0, s,
                       diag_value, scalar, q, n);
    ST_LAUNCH_CHECK();
    return 0;
}

// streaming kernels with a two-level sum: up to kStreamBlocks workgroups (8 waves per SIMD chip-wide) of 256 threads
static int stream_blocks(long long work_items) { return grid_for(work_items, kStreamBlocks); }

int launch_content_mse(const float* feat, const float* target, long long count, float weight, float* grad,
                       float* partials, float* loss_out, hipStream_t s, unsigned int* ticket) {
    const int blocks = stream_blocks((count & 3) == 0 ? count / 4 : count);
    const float norm = (float)(2.0 / (double)count);
    hipLaunchKernelGGL(content_mse_kernel, dim3(blocks), dim3(256), 0, s, feat, target, count, weight, norm,
                       grad, partials, LastBlock{ticket}, (float)count, loss_out);
    ST_LAUNCH_CHECK();
    if (ticket) return 0;                 // the last block wrote the loss
    hipLaunchKernelGGL(content_mse_final_kernel, dim3(1), dim3(64), 0, s, partials, blocks, (float)count,
                       weight, loss_out);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_style_loss_value(const float* mean, const float* mean_t, const float* cov, const float* cov_t,
                            const float* root, int n, float weight, float* loss_out, float* gdiag_out,
                            hipStream_t s) {
    hipLaunchKernelGGL(style_loss_value_kernel, dim3(1), dim3(256), 0, s,
                       W2LossJob{mean, mean_t, cov, cov_t, root, n, weight, loss_out, gdiag_out});
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_style_grad_finish(const float* g, const float* mean, const float* mean_t, int n, float weight,
                             long long npix, float* ssym, float* bvec, hipStream_t s, unsigned int* ssym_amax) {
    hipLaunchKernelGGL(style_grad_finish_kernel, dim3(n), dim3(256), 0, s, g, mean, mean_t, n, weight,
                       (float)npix, ssym, bvec, ssym_amax);
    ST_LAUNCH_CHECK();
    return 0;
}

// diagnostic (ST_TV_VARIANT=3): per-thread accumulators of tv_interior_kernel, kStreamBlocks x 256 x 5 floats, allocated once
float* tv_debug_buffer() {
    static float* buf = nullptr;
    if (!buf && hipMalloc(reinterpret_cast<void**>(&buf), (size_t)kStreamBlocks * 256 * 5 * sizeof(float)) != hipSuccess) buf = nullptr;
    return buf;
}

// the two TV launches; returns the number of partial-sum blocks (4 floats each) in `partials`
static int launch_tv_kernels(const float* image, int height, int width, StripInfo strip, float k1, float k3, float* grad,
                             float* partials, LastBlock lb, float n, float n2, float weight, float* loss_out,
                             hipStream_t s, int* nparts) {
    // (width >= 8 and height >= 2: with a single 4-pixel group per row, or a single row, the border kernel's "first and
    // last group / row" coincide and it would visit - and sum - those pixels twice)
    const bool vec = (width & 3) == 0 && width >= 8 && height >= 2 && (((long long)height * width) & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(image) | reinterpret_cast<uintptr_t>(grad)) & 15) == 0;
    int first = 0;
    long long border_threads;
    if (vec) {
        const int gpr = width >> 2;
        int tpr = 256;
        while (tpr > 1 && (tpr >> 1) >= gpr) tpr >>= 1;
        const int rpb = 256 / tpr;
        first = std::min((3 * height + rpb - 1) / rpb, kStreamBlocks - 256);
        static Option variant("ST_TV_VARIANT", 1);              // 1: shipped; 0 / 3: the reproducer (see the kernel)
        float* none = nullptr;
        switch (kExperiments ? variant.get() : 1) {
#if defined(ST_EXPERIMENTS)          // (the reproducer kernels exist only in builds with --experiments)
            case 0: hipLaunchKernelGGL(tv_interior_kernel<0>, dim3(first), dim3(256), 0, s, image, height, width, k1, k3, grad, partials, none); break;
            case 3: hipLaunchKernelGGL(tv_interior_kernel<3>, dim3(first), dim3(256), 0, s, image, height, width, k1, k3, grad, partials, tv_debug_buffer()); break;
#endif
            default: hipLaunchKernelGGL(tv_interior_kernel<1>, dim3(first), dim3(256), 0, s, image, height, width, k1, k3, grad, partials, none); break;
        }
        ST_LAUNCH_CHECK();
        border_threads = 3ll * (2 * gpr + 2 * (height - 2)) * 4;
    } else {
        border_threads = 3ll * height * width;
    }
    const int nb = (int)std::min<long long>((border_threads + 255) / 256, vec ? 256 : kStreamBlocks);
    hipLaunchKernelGGL(tv_border_kernel, dim3(nb), dim3(256), 0, s, image, height, width, k1, k3, grad, partials, strip,
                       vec ? 1 : 0, first, lb, n, n2, weight, loss_out);
    ST_LAUNCH_CHECK();
    *nparts = first + nb;
    return 0;
}

int launch_tv(const float* image, int height, int width, float weight, float* grad, float* partials,
              float* loss_out, hipStream_t s, unsigned int* ticket) {
    const double n = 3.0 * height * width, n2 = 3.0 * (height + 1) * (width + 1);
    // d loss / d D = weight * 2 * (1/3 or 1/12) * (1/n) * 2 D
    const float k1 = (float)(weight * 4.0 / (3.0 * n));
    const float k3 = (float)(weight * 4.0 / (12.0 * n2));
    StripInfo whole{0, height, 0, 0, nullptr};
    int nparts = 0;
    if (launch_tv_kernels(image, height, width, whole, k1, k3, grad, partials, LastBlock{ticket}, (float)n, (float)n2,
                          weight, loss_out, s, &nparts))
        return 1;
    if (ticket) return 0;                 // the border kernel's last block wrote the loss
    hipLaunchKernelGGL(tv_final_kernel, dim3(1), dim3(64), 0, s, partials, nparts, (float)n, (float)n2, weight,
                       loss_out);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_tv_strip(const float* image, int height, int width, StripInfo strip, float weight, float* grad,
                    float* partials, float* sums4, hipStream_t s) {
    const double n = 3.0 * strip.global_height * width, n2 = 3.0 * (strip.global_height + 1) * (width + 1);
    const float k1 = (float)(weight * 4.0 / (3.0 * n));
    const float k3 = (float)(weight * 4.0 / (12.0 * n2));
    int nparts = 0;
    if (launch_tv_kernels(image, height, width, strip, k1, k3, grad, partials, LastBlock{nullptr}, 0.f, 0.f, 0.f,
                          nullptr, s, &nparts))
        return 1;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(64), 0, s, partials, nparts, 4, sums4);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_tv_final(const float* sums4, int global_height, int width, float weight, float* loss_out,
                    hipStream_t s) {
    const double n = 3.0 * global_height * width, n2 = 3.0 * (global_height + 1) * (width + 1);
    hipLaunchKernelGGL(tv_from_sums_kernel, dim3(1), dim3(64), 0, s, sums4, (float)n, (float)n2, weight, loss_out);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_content_mse_strip(const float* feat, const float* target, long long local_count,
                             long long global_count, float weight, float* grad, float* partials,
                             float* sum_out, hipStream_t s) {
    const int blocks = stream_blocks((local_count & 3) == 0 ? local_count / 4 : local_count);
    const float norm = (float)(2.0 / (double)global_count);
    hipLaunchKernelGGL(content_mse_kernel, dim3(blocks), dim3(256), 0, s, feat, target, local_count, weight, norm,
                       grad, partials, LastBlock{nullptr}, 0.f, static_cast<float*>(nullptr));
    ST_LAUNCH_CHECK();
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(64), 0, s, partials, blocks, 1, sum_out);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_content_mse_final(const float* sum, long long global_count, float weight, float* loss_out,
                             hipStream_t s) {
    hipLaunchKernelGGL(mse_from_sum_kernel, dim3(1), dim3(64), 0, s, sum, (float)global_count, weight, loss_out);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_div_by_scalar(const float* a, float d, float* y, long long count, hipStream_t s) {
    hipLaunchKernelGGL(div_by_scalar_kernel, dim3(grid_for(count)), dim3(256), 0, s, a, d, y, count);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_sum_losses(float* losses8, hipStream_t s, float* copy) {
    hipLaunchKernelGGL(sum_losses_kernel, dim3(1), dim3(64), 0, s, losses8, copy == losses8 ? nullptr : copy);
    ST_LAUNCH_CHECK();
    return 0;
}

int launch_adam_clamp_ema(float* image, const float* grad, float* exp_avg, float* exp_avg_sq, float* ema,
                          long long count, AdamScalars sc, hipStream_t s, AdamTail tail) {
    if (tail.losses_copy == tail.losses8) tail.losses_copy = nullptr;
    hipLaunchKernelGGL(adam_clamp_ema_kernel, dim3(grid_for(count)), dim3(256), 0, s, image, grad, exp_avg,
                       exp_avg_sq, ema, count, sc, tail);
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
