// Dense n x n x n products (n = 64..512) for the Wasserstein-2 style loss, and the two fixed
// 12-step recurrences built from them:
//   ns_sqrt_forward   = sqrtm.sqrtm_ns                          (reference sqrtm.py:9-25)
//   ns_sqrt_backward  = _MatrixSquareRootNSLyap.backward        (reference sqrtm.py:36-47)
// Parity requires the recurrences step for step (same normalisation, same 12 iterations, same
// operand order) because NS-12 is NOT converged on the ill-conditioned covariances it sees
//  - so every product below is a separate fp32 GEMM with the elementwise
// ops of the reference applied in the same order in its epilogue.
//
// The chain is latency bound (<= 268 MFLOP per product, ~60 dependent launches per layer, and the backward
// pass of the whole network waits for relu5_1's chain), so the kernel is built for a short critical path, not
// for peak rate: one 32x32 output tile per workgroup, the K range split over WV waves (8 for n >= 256), every
// operand load of a wave issued up front straight into MFMA operand registers, and LDS used only for the
// cross-wave reduction.  Independent products of one recurrence step are batched in one launch (blockIdx.y).
// (Second version staged 32-k slices through wave-private LDS images in 4 rounds with two barriers each:
// 8-9 us per n = 512 launch against ~1.5 us of kernel boundary; this one has a single load -> MFMA -> reduce
// pass.)
#include <cstdint>
#include <cstdlib>

#include "st_common.h"

namespace st {
namespace {

// ---- kernel -----------------------------------------------------------------------------------
// Wave w owns k in [w KW, (w+1) KW), KW = N / WV (32 or 64).  MFMA e (0..3) of 8-block kb takes
// k = 8 kb + 4 (lane >> 5) + e for both operands, row / column lane & 31.  Two memory layouts per operand:
//   "RowK": element (r, k) at base[r N + k] (A, or B^T): the lane's 4 k values are one 16-byte load;
//   "KRow": element (k, r) at base[k N + r] (A^T, or B): one dword load per MFMA, 32 lanes = one 128-byte row.
template <int KW>
struct Operand {
    float v[KW / 8][4];
};

template <int N, int KW>
__device__ __forceinline__ void load_operand(Operand<KW>& o, const float* __restrict__ base, bool rowk, int row0,
                                             int k0, int l31, int half) {
    if (rowk) {
        const float* src = base + (size_t)(row0 + l31) * N + k0 + 4 * half;
#pragma unroll
        for (int kb = 0; kb < KW / 8; ++kb) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(src + kb * 8);
            o.v[kb][0] = t[0]; o.v[kb][1] = t[1]; o.v[kb][2] = t[2]; o.v[kb][3] = t[3];
        }
    } else {
        const float* src = base + (size_t)(k0 + 4 * half) * N + row0 + l31;
#pragma unroll
        for (int kb = 0; kb < KW / 8; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) o.v[kb][e] = src[(size_t)(kb * 8 + e) * N];
    }
}

template <int N, int WV>
__device__ __forceinline__ void gemm_tile(const GemmProblem& pr, int tile, float (*red)[16][64]) {
    constexpr int KW = N / WV;                      // k range of one wave
    constexpr int RPT = 16 / WV;                    // accumulator registers each wave owns in the reduction
    // cross-wave reduction buffer (<= 32 KB: small enough to co-reside with the trunk's conv workgroups, which
    // matters because these kernels run on side streams next to them)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    constexpr int nt = N / 32;
    const int m0 = (tile / nt) * 32, n0 = (tile % nt) * 32;
    const int k0 = wave * KW;
    const bool two = (pr.epilogue == EPI_DIFF);

    // product 1: op(a1) @ op(b1);  product 2 (EPI_DIFF only): a2^T @ (b2 - b2sub)
    Operand<KW> a, b, a2, b2, bs;
    load_operand<N, KW>(a, pr.a1, !pr.ta1, m0, k0, l31, half);
    load_operand<N, KW>(b, pr.b1, pr.tb1 != 0, n0, k0, l31, half);
    if (two) {
        load_operand<N, KW>(a2, pr.a2, false, m0, k0, l31, half);
        load_operand<N, KW>(b2, pr.b2, false, n0, k0, l31, half);
        load_operand<N, KW>(bs, pr.b2sub, false, n0, k0, l31, half);
    }
    f32x16 acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < KW / 8; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[kb][e], b.v[kb][e], acc1, 0, 0, 0);
    if (two) {
#pragma unroll
        for (int kb = 0; kb < KW / 8; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e)      // (a^T q - q a): the difference rounded like the reference's operand
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.v[kb][e], b2.v[kb][e] - bs.v[kb][e], acc2, 0, 0, 0);
    }

    // cross-wave K reduction in a fixed pairwise order; wave w finishes registers [w RPT, (w+1) RPT)
    auto reduce = [&](const f32x16& acc, float (&out)[RPT]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < RPT; ++rr) {
            float part[WV];
#pragma unroll
            for (int w = 0; w < WV; ++w) part[w] = red[w][wave * RPT + rr][lane];
#pragma unroll
            for (int span = 1; span < WV; span *= 2)
#pragma unroll
                for (int w = 0; w < WV; w += 2 * span) part[w] += part[w + span];
            out[rr] = part[0];
        }
    };
    float s1[RPT], s2[RPT];
    reduce(acc1, s1);
    if (two) {
        __syncthreads();
        reduce(acc2, s2);
    }

    float dscale = 1.f;
    if (pr.epilogue == EPI_DEV_SQRT_SCALE) dscale = sqrtf(pr.dev_scalar[0]);
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int r = wave * RPT + rr;
        const int orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ocol = n0 + l31;
        float v;
        if (pr.epilogue == EPI_SCALE) {
            v = s1[rr] * pr.c;
        } else if (pr.epilogue == EPI_IDENT_MINUS) {
            v = ((orow == ocol ? pr.ci : 0.f) - s1[rr]) * pr.c;
        } else if (pr.epilogue == EPI_DIFF) {
            v = (s1[rr] - s2[rr]) * pr.c;
        } else {
            v = s1[rr] * dscale;
        }
        pr.d[(size_t)orow * N + ocol] = v;
    }
}

template <int N, int WV>
__global__ __launch_bounds__(WV * 64) void gemm_batch_kernel(GemmBatch batch) {
    // cross-wave reduction buffer (<= 32 KB: small enough to co-reside with the trunk's conv workgroups, which
    // matters because these kernels run on side streams next to them)
    __shared__ float red[WV][16][64];
    gemm_tile<N, WV>(batch.p[blockIdx.y], blockIdx.x, red);
}

// Problems of DIFFERENT sizes (64 / 128 / 256) in one launch: the recurrence steps of the three shallow style heads in
// lockstep (st_api.hip).  8 waves for every size (k range of a wave: 8 / 16 / 32); blocks beyond a problem's tile count
// leave at once.
__global__ __launch_bounds__(512) void gemm_mixed_kernel(GemmBatch batch) {
    __shared__ float red[8][16][64];
    const GemmProblem& pr = batch.p[blockIdx.y];
    const int n = pr.n ? pr.n : batch.n;
    const int tiles = (n / 32) * (n / 32);
    if ((int)blockIdx.x >= tiles) return;
    if (n == 64) gemm_tile<64, 8>(pr, blockIdx.x, red);
    else if (n == 128) gemm_tile<128, 8>(pr, blockIdx.x, red);
    else gemm_tile<256, 8>(pr, blockIdx.x, red);
}

// ---- n = 512: slices staged through wave-private LDS images --------------------------------------
// (the direct-load kernel above is slower here: 250 / 392 us per forward / backward chain against 243 / 344;
// a 32 x 64-tile variant with 8-byte column-pair loads was slower still, 284 / 489.  PMC: at n = 512 the waves
// spend half their life waiting on L2 / fabric whatever the load pattern, 35 % of the lines miss L2 because every
// XCD re-fetches what the previous launch's other XCDs wrote.  Issuing ALL of a wave's loads up front with only
// wave-level ordering between rounds was also slower, 242 / 368: the one-round-ahead prefetch below stays.)
// One 32x32 output tile per workgroup; wave w owns k in [w N/4, (w+1) N/4) and walks it in rounds of
// RK = min(32, N/4).  Each round the wave copies its A slice [32 x RK] and B slice [RK x 32] into a
// wave-private LDS region with coalesced 16-byte global loads (register-prefetched one round ahead),
// then issues RK/2 MFMAs from LDS.  Two LDS images, chosen by how the operand lies in memory:
//   "RowK": element (r, k) at [r][k], pitch 36 - for operands whose k index is contiguous in memory
//           (A, or B^T); a lane reads 4 consecutive k with one conflict-free ds_read_b128;
//   "KRow": element (k, r) at [k][r], pitch 36 - for operands whose row index is contiguous (A^T, B);
//           a lane reads one dword per MFMA, the 32 lanes of a half-wave hit 32 consecutive banks.
// k order inside an 8-block: MFMA e (0..3) takes k = 8 kb + 4 (lane >> 5) + e for both operands.
constexpr int kPitch = 36;
constexpr int kImage = 32 * kPitch;          // floats per staged operand image (RowK needs 32 rows)

template <int RK>
struct TileRegs {
    f32x4 v[RK / 8];
};

// coalesced global -> register load of one operand slice.  row0/k0 locate the slice; `rowk` = the k
// index is contiguous in memory (element (r,k) at base[(row0+r)*n + k0+k]), else element at base[(k0+k)*n + row0+r]
template <int N, int RK>
__device__ __forceinline__ void load_slice(TileRegs<RK>& t, const float* __restrict__ base, bool rowk,
                                           int row0, int k0, int lane) {
    if (rowk) {
        constexpr int C4 = RK / 4;                  // float4 per row
#pragma unroll
        for (int i = 0; i < RK / 8; ++i) {
            const int r = lane / C4 + (64 / C4) * i, c4 = lane % C4;
            t.v[i] = *reinterpret_cast<const f32x4*>(base + (size_t)(row0 + r) * N + k0 + c4 * 4);
        }
    } else {
#pragma unroll
        for (int i = 0; i < RK / 8; ++i) {
            const int k = lane / 8 + 8 * i, c4 = lane % 8;
            t.v[i] = *reinterpret_cast<const f32x4*>(base + (size_t)(k0 + k) * N + row0 + c4 * 4);
        }
    }
}

template <int RK>
__device__ __forceinline__ void store_slice(const TileRegs<RK>& t, float* __restrict__ img, bool rowk, int lane) {
    if (rowk) {
        constexpr int C4 = RK / 4;
#pragma unroll
        for (int i = 0; i < RK / 8; ++i) {
            const int r = lane / C4 + (64 / C4) * i, c4 = lane % C4;
            *reinterpret_cast<f32x4*>(img + r * kPitch + c4 * 4) = t.v[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < RK / 8; ++i) {
            const int k = lane / 8 + 8 * i, c4 = lane % 8;
            *reinterpret_cast<f32x4*>(img + k * kPitch + c4 * 4) = t.v[i];
        }
    }
}

template <int N, int WV>
__global__ __launch_bounds__(WV * 64) void gemm_staged_kernel(GemmBatch batch) {
    constexpr int KW = N / WV;                      // k range of one wave
    constexpr int RK = KW < 32 ? KW : 32;
    constexpr int NR = KW / RK;                     // rounds per product
    // [wave][operand] images (36 KB: small enough to co-reside with the trunk's conv workgroups, which
    // matters because these kernels run on side streams next to them); the next round waits in registers.
    // The cross-wave reduction reuses the same memory afterwards.
    __shared__ __attribute__((aligned(16))) float lds[WV * 2 * kImage];
    static_assert(WV * 2 * kImage >= 2 * WV * 16 * 64, "reduction buffer must fit the staging images");
    const GemmProblem& pr = batch.p[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    constexpr int nt = N / 32;
    // (round 4: ns_gemm_f16_kernel's XCD-aware tile order - a 4 x 8 block of tiles per XCD, 768 KB instead of 1.1 MB of first
    // touches per L2 - measured here too: forward chain 220.9 -> 224.5 us, full fp32 backward chain 334.7 -> 360.7 us, closure
    // 128^2 / 256^2 / 512^2 within noise.  The misses are latency, not traffic; the row-major order stays.)
    const int m0 = (blockIdx.x / nt) * 32, n0 = (blockIdx.x % nt) * 32;
    const int kbeg = wave * KW;
    const bool two = (pr.epilogue == EPI_DIFF);
    const int rounds = two ? 2 * NR : NR;
    float* my = lds + wave * (2 * kImage);

    f32x16 acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; }

    TileRegs<RK> ra, rb, rs;
    // round r < NR: product 1 (op(a1) @ op(b1)); r >= NR: product 2 (a2^T @ (b2 - b2sub))
    auto a_rowk = [&](int r) { return r < NR ? !pr.ta1 : false; };
    auto b_rowk = [&](int r) { return r < NR ? (pr.tb1 != 0) : false; };
    auto fetch = [&](int r) __attribute__((always_inline)) {
        const bool second = r >= NR;
        const int k0 = kbeg + (second ? r - NR : r) * RK;
        load_slice<N, RK>(ra, second ? pr.a2 : pr.a1, a_rowk(r), m0, k0, lane);
        load_slice<N, RK>(rb, second ? pr.b2 : pr.b1, b_rowk(r), n0, k0, lane);
        if (second) load_slice<N, RK>(rs, pr.b2sub, false, n0, k0, lane);
    };
    auto stash = [&](int r) __attribute__((always_inline)) {
        float* img = my;
        if (r >= NR) {
#pragma unroll
            for (int i = 0; i < RK / 8; ++i) rb.v[i] = rb.v[i] - rs.v[i];   // (a^T q - q a), rounded like the reference
        }
        store_slice<RK>(ra, img, a_rowk(r), lane);
        store_slice<RK>(rb, img + kImage, b_rowk(r), lane);
    };

    fetch(0);
    stash(0);
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        const bool more = r + 1 < rounds;
        if (more) fetch(r + 1);
        const float* ia = my;
        const float* ib = ia + kImage;
        const bool ark = a_rowk(r), brk = b_rowk(r);
        f32x16& acc = (r < NR) ? acc1 : acc2;
#pragma unroll
        for (int kb = 0; kb < RK / 8; ++kb) {
            float a[4], b[4];
            if (ark) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(ia + l31 * kPitch + kb * 8 + 4 * half);
                a[0] = t[0]; a[1] = t[1]; a[2] = t[2]; a[3] = t[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = ia[(kb * 8 + 4 * half + e) * kPitch + l31];
            }
            if (brk) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(ib + l31 * kPitch + kb * 8 + 4 * half);
                b[0] = t[0]; b[1] = t[1]; b[2] = t[2]; b[3] = t[3];
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = ib[(kb * 8 + 4 * half + e) * kPitch + l31];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
        }
        __syncthreads();                 // everyone is done reading this round's images
        if (more) {
            stash(r + 1);
            __syncthreads();
        }
    }

    // cross-wave K reduction through LDS (all staged data is dead after the last barrier), pairwise in a fixed order
    float (*red)[WV][16][64] = reinterpret_cast<float (*)[WV][16][64]>(lds);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[0][wave][r][lane] = acc1[r];
        if (two) red[1][wave][r][lane] = acc2[r];
    }
    __syncthreads();

    float dscale = 1.f;
    if (pr.epilogue == EPI_DEV_SQRT_SCALE) dscale = sqrtf(pr.dev_scalar[0]);
    constexpr int RPT = 16 / WV;                   // accumulator registers each wave finishes
    auto total = [&](int which, int r) __attribute__((always_inline)) {
        float part[WV];
#pragma unroll
        for (int w = 0; w < WV; ++w) part[w] = red[which][w][r][lane];
#pragma unroll
        for (int span = 1; span < WV; span *= 2)
#pragma unroll
            for (int w = 0; w < WV; w += 2 * span) part[w] += part[w + span];
        return part[0];
    };
    float sq = 0.f;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int r = wave * RPT + rr;
        const float s1 = total(0, r);
        const int orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ocol = n0 + l31;
        float v;
        if (pr.epilogue == EPI_SCALE) {
            v = s1 * pr.c;
        } else if (pr.epilogue == EPI_IDENT_MINUS) {
            v = ((orow == ocol ? pr.ci : 0.f) - s1) * pr.c;
        } else if (pr.epilogue == EPI_DIFF) {
            v = (s1 - total(1, r)) * pr.c;
        } else {
            v = s1 * dscale;
        }
        pr.d[(size_t)orow * N + ocol] = v;
        sq = fmaf(v, v, sq);
    }
    // sum of squares of this tile for the consumer's Frobenius norm (ns_prepare_kernel / ns_backward_entry_kernel combine the
    // tiles' partials in index order): wave butterfly, then the WV waves in order - a fixed order, like sumsq_partial4_kernel's
    if (pr.sumsq_partials) {
        __shared__ float sq_wave[WV];
        sq = wave_sum(sq);
        if (lane == 0) sq_wave[wave] = sq;
        __syncthreads();
        if (tid == 0) {
            float t = sq_wave[0];
#pragma unroll
            for (int w = 1; w < WV; ++w) t += sq_wave[w];
            pr.sumsq_partials[blockIdx.x] = t;
        }
    }
}

// ---- dF = Ssym F + b 1^T on a SMALL tap (<= 1024 pixels) in one launch ---------------------------------------------
// The heads' 1x1 gradient step (backward of the einsum + mean of style_transfer.py:162-168).  For such a tap the
// convolution launcher has 64 workgroups' worth of tiles, splits K to fill the chip and reduces in a second launch: 21 + 10 us
// and a kernel boundary on relu5_1's critical path at 512^2.  Here one 32 co x 32 px tile per workgroup, K = C split over 8
// waves, operands straight into MFMA registers as in gemm_tile, the waves' partial tiles combined in a fixed pairwise order;
// bias, the producer-side ReLU mask (relu5_1, see style_head_gradient) and the fp16x3 consumer's bound in the epilogue.
template <int C>
__global__ __launch_bounds__(512) void head_dgrad_small_kernel(const float* __restrict__ ssym, const float* __restrict__ feat,
                                                               const float* __restrict__ bias, const float* __restrict__ mask,
                                                               float* __restrict__ out, int npix, unsigned int* out_amax) {
    constexpr int WV = 8, KW = C / WV, RPT = 16 / WV;
    __shared__ float red[WV][16][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32, k0 = wave * KW;
    Operand<KW> a, b;
    {
        const float* src = ssym + (size_t)(m0 + l31) * C + k0 + 4 * half;            // Ssym[co][ci]: 4 k values = 16 bytes
#pragma unroll
        for (int kb = 0; kb < KW / 8; ++kb) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(src + kb * 8);
            a.v[kb][0] = t[0]; a.v[kb][1] = t[1]; a.v[kb][2] = t[2]; a.v[kb][3] = t[3];
        }
        const float* fs = feat + (size_t)(k0 + 4 * half) * npix + n0 + l31;          // F[ci][px]: 32 lanes = one 128-byte row
#pragma unroll
        for (int kb = 0; kb < KW / 8; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) b.v[kb][e] = fs[(size_t)(kb * 8 + e) * npix];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < KW / 8; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[kb][e], b.v[kb][e], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    unsigned int amax = 0;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
        const int r = wave * RPT + rr;
        float part[WV];
#pragma unroll
        for (int w = 0; w < WV; ++w) part[w] = red[w][r][lane];
#pragma unroll
        for (int span = 1; span < WV; span *= 2)
#pragma unroll
            for (int w = 0; w < WV; w += 2 * span) part[w] += part[w + span];
        const int orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const size_t at = (size_t)orow * npix + n0 + l31;
        float v = part[0] + bias[orow];
        if (mask != nullptr) v = (mask[at] > 0.f) ? v : 0.f;
        out[at] = v;
        amax = max(amax, abs_bits(v));
    }
    if (out_amax) amax_commit(amax, out_amax);
}

GemmProblem plain(const float* a, const float* b, float* d, float c = 1.f, int ta = 0, int tb = 0) {
    GemmProblem p{};
    p.a1 = a; p.b1 = b; p.d = d; p.ta1 = ta; p.tb1 = tb; p.epilogue = EPI_SCALE; p.c = c;
    return p;
}

}  // namespace

int launch_gemm_batch(const GemmBatch& b, hipStream_t s) {
    ST_REQUIRE(b.count >= 1 && b.count <= 6, "gemm: batch count out of range");
    for (int i = 0; i < b.count; ++i) {
        ST_REQUIRE(!(b.p[i].ta1 && b.p[i].tb1), "gemm: A^T @ B^T is not implemented");
        ST_REQUIRE(b.p[i].epilogue != EPI_DIFF || (b.p[i].ta2 == 1), "gemm: second product must be A2^T @ (B2 - B2sub)");
    }
    bool mixed = false;
    int nmax = 0;
    for (int i = 0; i < b.count; ++i) {
        const int ni = b.p[i].n ? b.p[i].n : b.n;
        mixed = mixed || ni != b.n;
        nmax = std::max(nmax, ni);
        ST_REQUIRE(!b.p[i].n || ni == 64 || ni == 128 || ni == 256, "gemm: a problem's own size must be 64, 128 or 256");
    }
    if (mixed) {
        ST_REQUIRE(b.n <= 256, "gemm: mixed batches are for n <= 256");
        for (int i = 0; i < b.count; ++i) ST_REQUIRE(b.p[i].epilogue != EPI_DIFF, "gemm: mixed batches have single products");
        hipLaunchKernelGGL(gemm_mixed_kernel, dim3((nmax / 32) * (nmax / 32), b.count), dim3(512), 0, s, b);
        ST_LAUNCH_CHECK();
        return 0;
    }
    for (int i = 0; i < b.count; ++i)
        ST_REQUIRE(!b.p[i].sumsq_partials || (!mixed && gemm_sumsq_fusable(b.n)), "gemm: fused sums of squares exist for n = 512 only");
    const int nt = b.n / 32;
    const dim3 grid(nt * nt, b.count);
    switch (b.n) {
        case 64: hipLaunchKernelGGL((gemm_batch_kernel<64, 2>), grid, dim3(128), 0, s, b); break;
        case 128: hipLaunchKernelGGL((gemm_batch_kernel<128, 4>), grid, dim3(256), 0, s, b); break;
        case 256: hipLaunchKernelGGL((gemm_batch_kernel<256, 8>), grid, dim3(512), 0, s, b); break;
        // (8 waves / 2 rounds per tile: 218 vs 232 us for the isolated forward chain, no gain in the closure and
        // twice the LDS footprint next to the trunk's conv workgroups - 4 waves stay)
        case 512: hipLaunchKernelGGL((gemm_staged_kernel<512, 4>), grid, dim3(256), 0, s, b); break;
        default: ST_REQUIRE(false, "gemm: n must be 64, 128, 256 or 512 (got %d)", b.n);
    }
    ST_LAUNCH_CHECK();
    return 0;
}

// ST_NS_FUSED_SUMSQ=0: the chains' norms from launches of their own (rounds 1 - 3)
bool gemm_sumsq_fusable(int n) {
    static Option on("ST_NS_FUSED_SUMSQ", 1);
    return on.get() != 0 && n == 512;
}

bool head_dgrad_small_applies(int channels, long long npix) {
    static Option on("ST_HEAD_SMALL_GEMM", 1);            // 0: the convolution launcher's split-K path (rounds 1 - 3)
    return on.get() != 0 && (channels == 64 || channels == 128 || channels == 256 || channels == 512) && npix >= 32 &&
           npix <= 1024 && npix % 32 == 0;
}

int launch_head_dgrad_small(const float* ssym, const float* feat, const float* bias, const float* mask, float* out, int channels,
                            long long npix, unsigned int* out_amax, hipStream_t s) {
    ST_REQUIRE(head_dgrad_small_applies(channels, npix), "head gradient (small tap): unsupported shape %d x %lld", channels, npix);
    ST_REQUIRE((reinterpret_cast<uintptr_t>(ssym) & 15) == 0, "head gradient (small tap): Ssym must be 16-byte aligned");
    const dim3 grid(channels / 32, (unsigned)(npix / 32));
    switch (channels) {
        case 64: hipLaunchKernelGGL(head_dgrad_small_kernel<64>, grid, dim3(512), 0, s, ssym, feat, bias, mask, out, (int)npix, out_amax); break;
        case 128: hipLaunchKernelGGL(head_dgrad_small_kernel<128>, grid, dim3(512), 0, s, ssym, feat, bias, mask, out, (int)npix, out_amax); break;
        case 256: hipLaunchKernelGGL(head_dgrad_small_kernel<256>, grid, dim3(512), 0, s, ssym, feat, bias, mask, out, (int)npix, out_amax); break;
        default: hipLaunchKernelGGL(head_dgrad_small_kernel<512>, grid, dim3(512), 0, s, ssym, feat, bias, mask, out, (int)npix, out_amax); break;
    }
    ST_LAUNCH_CHECK();
    return 0;
}

// 12 matrices + scalars / partials (+ the fp16x3 chains' plane slots: 5 x 2 roles x 2 planes x n*n halves)
// ... + the persistent chain kernel's barrier words (two sets + the error line)
static size_t chain_words() { return (size_t)2 * ns_chain_sync_uints() + 64; }
// ST_NS_CHAIN_L2: 0 the chain kernel's operands from the memory side (sc1 loads), 1 through the L2 behind an acquire per
// barrier, 2 through the L2 with a matrix of its own for every iterate (no acquire; + 132 n^2 floats of workspace)
static int chain_l2_mode() {
    static Option l2_opt("ST_NS_CHAIN_L2", 0);
    return l2_opt.get();
}
static size_t chain_arena_floats(int n) { return (ns_chain_enabled() && chain_l2_mode() == 2) ? (size_t)kNsChainArenaMats * n * n : 0; }
size_t ns_workspace_floats(int n) {
    return (size_t)12 * n * n + 512 + (n >= 256 ? (size_t)10 * n * n : 0) + chain_words() + chain_arena_floats(n);
}

void ns_workspace_carve(NSWorkspace& ws, float* base, int n) {
    const size_t nn = (size_t)n * n;
    float** slots[] = {&ws.y0, &ws.y1, &ws.z0, &ws.z1, &ws.t,  &ws.a0,
                       &ws.a1, &ws.q0, &ws.q1, &ws.e,  &ws.atq, &ws.qa};
    for (int i = 0; i < 12; ++i) *slots[i] = base + i * nn;
    ws.scalars = base + 12 * nn;
    ws.planes = n >= 256 ? reinterpret_cast<_Float16*>(base + 12 * nn + 512) : nullptr;
    ws.chain_sync = reinterpret_cast<unsigned int*>(base + 12 * nn + 512 + (n >= 256 ? 10 * nn : 0));
    ws.chain_launches = 0;
    ws.chain_arena = chain_arena_floats(n) ? reinterpret_cast<float*>(ws.chain_sync + chain_words()) : nullptr;
}

int ns_workspace_reset(NSWorkspace& ws, hipStream_t s) {
    ST_HIP(hipMemsetAsync(ws.chain_sync, 0, chain_words() * sizeof(unsigned int), s));
    ws.chain_launches = 0;
    return 0;
}

NsChainJob ns_chain_job(NSWorkspace& ws, int n) {
    NsChainJob j{};
    j.n = n;
    j.y0 = ws.y0; j.y1 = ws.y1; j.z0 = ws.z0; j.z1 = ws.z1; j.t = ws.t;
    j.yt0 = ws.a0; j.yt1 = ws.a1; j.zt0 = ws.q0; j.zt1 = ws.q1; j.tt = ws.e;       // (the launch-per-product backward's slots)
    // ST_NS_CHAIN_SYM: bit mask over log2(n / 64) - which sizes run on symmetric tile pairs.  Default 8 = n = 512 only:
    // measured (tools/ns_chain_bench.py, CPU emulation in profiles/r05_ns_chain.md), enforcing symmetry costs accuracy on
    // rank-deficient input - 7 x the reference's own fp32-vs-float64 distance at n = 64, 1.8 x at 256, 1.25 x at 512
    static Option sym_mask("ST_NS_CHAIN_SYM", 8);
    const int bit = n == 64 ? 1 : n == 128 ? 2 : n == 256 ? 4 : 8;
    j.symmetric = (sym_mask.get() & bit) ? 1 : 0;
    j.l2_loads = chain_l2_mode() != 0;
    j.arena = chain_l2_mode() == 2 ? ws.chain_arena : nullptr;
    j.l2_loads = j.l2_loads && (chain_l2_mode() == 1 || j.arena);      // (a workspace carved before the option was set: sc1 loads)
    j.scalars = ws.scalars;
    const int parity = ws.chain_launches++ & 1;
    j.sync = ws.chain_sync + (size_t)parity * ns_chain_sync_uints();
    j.sync_next = ws.chain_sync + (size_t)(parity ^ 1) * ns_chain_sync_uints();
    j.error = ws.chain_sync + (size_t)2 * ns_chain_sync_uints();
    return j;
}

int ns_chain_check(NSWorkspace& ws, const char* what) {
    unsigned int err = 0;
    ST_HIP(hipMemcpy(&err, ws.chain_sync + (size_t)2 * ns_chain_sync_uints(), sizeof(err), hipMemcpyDeviceToHost));
    ST_REQUIRE(err == 0, "%s: the persistent Newton-Schulz kernel gave up waiting at a grid barrier (a workgroup did not become "
                         "resident within 50 ms); results are NaN.  ST_NS_CHAIN=0 selects the launch-per-product form", what);
    return 0;
}

// both recurrences of a head in one launch: only in the shipped arithmetic (fp32 forward chain, reduced backward recurrence)
bool ns_chain_combined() {
    static Option full("ST_NS_FULL_BACKWARD", 0);
    static Option f16_fwd("ST_NS_F16_FWD", 0);
    return ns_chain_enabled() && !full.get() && !f16_fwd.get();
}

int ns_sqrt_chain(const float* const* m, float* const* root, float* const* grad_m, const int* n, NSWorkspace* const* ws,
                  const int* m_partials, const W2LossJob* loss, int lanes, hipStream_t s) {
    ST_REQUIRE(lanes >= 1 && lanes <= 3, "ns chain: 1 to 3 heads per launch");
    NsChainLaunch launch{};
    launch.count = lanes;
    for (int l = 0; l < lanes; ++l) {
        NsChainJob& j = launch.job[l];
        j = ns_chain_job(*ws[l], n[l]);
        j.forward = 1; j.backward = 1;
        j.m = m[l]; j.root = root[l]; j.grad_m = grad_m[l];
        if (m_partials && m_partials[l] > 0) { j.m_partials = ws[l]->scalars + 8; j.m_nparts = m_partials[l]; }
        ST_REQUIRE(loss && loss[l].loss_out, "ns chain: the combined form carries the head's W2 scalars");
        j.loss = loss[l];
    }
    return launch_ns_chain(launch, s);
}

static bool ns_skip_identity() {
    static Option opt("ST_NS_SKIP_IDENTITY", 1);
    return opt.get() != 0;
}

int ns_sqrt_forward(const float* m, float* root, int n, NSWorkspace& ws, hipStream_t s, int m_partials, int* root_partials) {
    if (root_partials) *root_partials = 0;
    // The FORWARD chain stays fp32 by default.  Its result enters the loss through a difference of traces, and the
    // non-converged iteration turns rounding noise of the iterates into a systematic shift of tr(root) (every
    // implementation, the reference's fp32 included, sits on the same side of the float64 value).  Two fp16 planes
    // hold 23 bits of an iterate: twice fp32's rounding amplitude, and the measured shift of the relu4_1 / relu5_1
    // terms grows from 1-8e-5 (fp32 chains) to 0.6-2.4e-4 (tools/ns_accuracy.py) - at the edge of the 1e-4-class
    // tolerance for two terms that carry 1.5 % of the loss.  The backward chain only feeds the gradient (1e-3 bar;
    // its result moves by 3e-6): that one runs in fp16x3.  ST_NS_F16_FWD=1 switches the forward chain as well.
    // Round 3: the shift is NOT an operand-precision effect.  A three-plane variant (every fp32 iterate represented
    // exactly, six plane products down to 2^-22 of the result) shifted tr(root) by the SAME amount as two planes
    // (rank-deficient n = 512: +1.45e-5 for both against +6.5e-7 for the fp32 chain, profiles/r03_head_window.md): the
    // bias sits in how the 16-bit matrix instruction accumulates its 16 products, which no operand splitting removes.
    static Option f16_fwd("ST_NS_F16_FWD", 0);
    if (f16_fwd.get() && ns_f16_applies(n) && ws.planes) return ns_sqrt_forward_f16(m, root, n, ws, s);
    // round 5: the whole recurrence as one persistent launch on upper-triangle tile pairs (st_nschain.hip)
    if ((ns_chain_mask() & 8) && ws.chain_sync) {
        NsChainLaunch launch{};
        launch.count = 1;
        NsChainJob& j = launch.job[0];
        j = ns_chain_job(ws, n);
        j.forward = 1;
        j.m = m; j.root = root;
        if (m_partials > 0) { j.m_partials = ws.scalars + 8; j.m_nparts = m_partials; }
        return launch_ns_chain(launch, s);          // (*root_partials stays 0: a separate backward launch sums the root itself)
    }
    // norm_a = a.pow(2).sum().sqrt(); y = a / norm_a; z = I                      (sqrtm.py:16-20)
    // The first step multiplies by z = I twice: z @ y is y and t @ z is t, exactly, in any fp32 GEMM (one non-zero term
    // per sum).  So t_0 = (3I - y_0) / 2 comes out of the prologue kernel, is z_1 as it stands, and the step is the one
    // product y_1 = y_0 @ t_0: one launch and one product instead of two launches and three products, same bits
    // (ST_NS_SKIP_IDENTITY=0: the literal form; test_sqrtm_first_step_shortcut_is_bit_identical).
    const bool shortcut = ns_skip_identity();
    if (launch_ns_prepare(m, n, ws.scalars + 0, ws.scalars + 8, ws.y0, nullptr, nullptr, shortcut ? ws.z1 : ws.z0, s, shortcut,
                          nullptr, m_partials))
        return 1;
    float *y = ws.y0, *yn = ws.y1, *z = ws.z0, *zn = ws.z1;
    if (shortcut) {
        GemmBatch b{};
        b.n = n; b.count = 1;
        b.p[0] = plain(y, zn, yn);                                  // y_1 = y_0 @ t_0         (:23)
        if (launch_gemm_batch(b, s)) return 1;
        std::swap(y, yn);
        std::swap(z, zn);                                           // z_1 = t_0               (:24)
    }
    for (int it = shortcut ? 1 : 0; it < 12; ++it) {
        const bool last = (it == 11);
        GemmBatch b1{};
        b1.n = n; b1.count = 1;                                     // t = (3I - z @ y) / 2   (:22)
        b1.p[0] = plain(z, y, ws.t);
        b1.p[0].epilogue = EPI_IDENT_MINUS; b1.p[0].ci = 3.f; b1.p[0].c = 0.5f;
        if (launch_gemm_batch(b1, s)) return 1;
        GemmBatch b2{};
        b2.n = n;
        if (!last) {
            b2.count = 2;
            b2.p[0] = plain(y, ws.t, yn);                           // y = y @ t              (:23)
            b2.p[1] = plain(ws.t, z, zn);                           // z = t @ z              (:24)
        } else {
            b2.count = 1;                                           // return y * sqrt(norm_a) (:25)
            b2.p[0] = plain(y, ws.t, root);
            b2.p[0].epilogue = EPI_DEV_SQRT_SCALE; b2.p[0].dev_scalar = ws.scalars + 0;
            if (root_partials && gemm_sumsq_fusable(n)) {
                // ||root||_F is the first thing the backward chain needs (sqrtm.py:38): this launch leaves the tiles' sums
                // (the partial buffer is free again: the prologue consumed m's sums eleven steps ago)
                b2.p[0].sumsq_partials = ws.scalars + 8;
                *root_partials = (n / 32) * (n / 32);
            }
        }
        if (launch_gemm_batch(b2, s)) return 1;
        float* tmp = y; y = yn; yn = tmp;
        tmp = z; z = zn; zn = tmp;
    }
    return 0;
}

// ---- up to three fp32 chains of different sizes in lockstep (the shallow style heads: n = 64, 128, 256) -------------
// The same recurrences as ns_sqrt_forward / the reduced ns_sqrt_backward, step for step and product for product; only
// the launches are shared (gemm_mixed_kernel), so a step of all chains is ONE dependent launch instead of one per chain.
namespace {
GemmProblem sized(GemmProblem p, int n) {
    p.n = n;
    return p;
}
int batch_n(const int* n, int lanes) {
    int m = 0;
    for (int l = 0; l < lanes; ++l) m = std::max(m, n[l]);
    return m;
}
}  // namespace

int ns_sqrt_forward_lockstep(const float* const* m, float* const* root, const int* n, NSWorkspace* const* wsp, int lanes,
                             hipStream_t s) {
    ST_REQUIRE(lanes >= 1 && lanes <= 3, "ns forward (lockstep): 1 to 3 chains");
    const bool shortcut = ns_skip_identity();
    float *y[3], *yn[3], *z[3], *zn[3];
    for (int l = 0; l < lanes; ++l) {
        ST_REQUIRE(n[l] == 64 || n[l] == 128 || n[l] == 256, "ns forward (lockstep): n must be 64, 128 or 256");
        NSWorkspace& ws = *wsp[l];
        // norm_a = a.pow(2).sum().sqrt(); y = a / norm_a; z = I                      (sqrtm.py:16-20)
        if (launch_ns_prepare(m[l], n[l], ws.scalars + 0, ws.scalars + 8, ws.y0, nullptr, nullptr, shortcut ? ws.z1 : ws.z0, s,
                              shortcut))
            return 1;
        y[l] = ws.y0; yn[l] = ws.y1; z[l] = ws.z0; zn[l] = ws.z1;
    }
    const int nb = batch_n(n, lanes);
    if (shortcut) {                                                 // first step with z = I: see ns_sqrt_forward
        GemmBatch b{};
        b.n = nb; b.count = lanes;
        for (int l = 0; l < lanes; ++l) b.p[l] = sized(plain(y[l], zn[l], yn[l]), n[l]);
        if (launch_gemm_batch(b, s)) return 1;
        for (int l = 0; l < lanes; ++l) {
            std::swap(y[l], yn[l]);
            std::swap(z[l], zn[l]);
        }
    }
    for (int it = shortcut ? 1 : 0; it < 12; ++it) {
        const bool last = (it == 11);
        GemmBatch b1{};
        b1.n = nb; b1.count = lanes;                                // t = (3I - z @ y) / 2   (:22)
        for (int l = 0; l < lanes; ++l) {
            b1.p[l] = sized(plain(z[l], y[l], wsp[l]->t), n[l]);
            b1.p[l].epilogue = EPI_IDENT_MINUS; b1.p[l].ci = 3.f; b1.p[l].c = 0.5f;
        }
        if (launch_gemm_batch(b1, s)) return 1;
        GemmBatch b2{};
        b2.n = nb;
        for (int l = 0; l < lanes; ++l) {
            if (!last) {
                b2.p[b2.count++] = sized(plain(y[l], wsp[l]->t, yn[l]), n[l]);       // y = y @ t              (:23)
                b2.p[b2.count++] = sized(plain(wsp[l]->t, z[l], zn[l]), n[l]);       // z = t @ z              (:24)
            } else {
                GemmProblem pr = sized(plain(y[l], wsp[l]->t, root[l]), n[l]);       // return y * sqrt(norm_a) (:25)
                pr.epilogue = EPI_DEV_SQRT_SCALE; pr.dev_scalar = wsp[l]->scalars + 0;
                b2.p[b2.count++] = pr;
            }
        }
        if (launch_gemm_batch(b2, s)) return 1;
        for (int l = 0; l < lanes; ++l) {
            std::swap(y[l], yn[l]);
            std::swap(z[l], zn[l]);
        }
    }
    return 0;
}

// _MatrixSquareRootNSLyap.backward for grad_output = gdiag * I in its reduced form (see ns_sqrt_backward)
int ns_sqrt_backward_diag_lockstep(const float* const* root, const float* const* grad_diag, float* const* grad_m, const int* n,
                                   NSWorkspace* const* wsp, int lanes, hipStream_t s, const W2LossJob* loss) {
    ST_REQUIRE(lanes >= 1 && lanes <= 3, "ns backward (lockstep): 1 to 3 chains");
    float *a[3], *an[3], *q[3], *qn[3];
    for (int l = 0; l < lanes; ++l) {
        NSWorkspace& ws = *wsp[l];
        // norm_z = ||z||_F; a = z / norm_z; q = grad / norm_z                        (sqrtm.py:38-41)
        if (launch_ns_prepare(root[l], n[l], ws.scalars + 1, ws.scalars + 8, ws.a0, nullptr, grad_diag[l], ws.q0, s, false,
                              loss ? &loss[l] : nullptr))
            return 1;
        a[l] = ws.a0; an[l] = ws.a1; q[l] = ws.q0; qn[l] = ws.q1;
    }
    const int nb = batch_n(n, lanes);
    for (int it = 0; it < 12; ++it) {
        const bool last = (it == 11);
        GemmBatch b1{};
        b1.n = nb; b1.count = lanes;                                // eye_a_a = 3I - a @ a    (:43)
        for (int l = 0; l < lanes; ++l) {
            b1.p[l] = sized(plain(a[l], a[l], wsp[l]->e), n[l]);
            b1.p[l].epilogue = EPI_IDENT_MINUS; b1.p[l].ci = 3.f; b1.p[l].c = 1.f;
        }
        if (launch_gemm_batch(b1, s)) return 1;
        GemmBatch b2{};
        b2.n = nb;
        for (int l = 0; l < lanes; ++l) {
            // q = q @ eye_a_a / 2 (:44 without the vanishing commutator); the final "/ 2" (:47) folds into the last one
            b2.p[b2.count++] = sized(plain(q[l], wsp[l]->e, last ? grad_m[l] : qn[l], last ? 0.25f : 0.5f), n[l]);
            if (!last) b2.p[b2.count++] = sized(plain(a[l], wsp[l]->e, an[l], 0.5f), n[l]);      // a = a @ eye_a_a / 2 (:46)
        }
        if (launch_gemm_batch(b2, s)) return 1;
        for (int l = 0; l < lanes; ++l) {
            std::swap(a[l], an[l]);
            std::swap(q[l], qn[l]);
        }
    }
    return 0;
}

int ns_sqrt_backward(const float* root, const float* grad_root, const float* grad_diag, float* grad_m, int n,
                     NSWorkspace& ws, hipStream_t s, const W2LossJob* loss, int root_partials) {
    ST_REQUIRE(!loss || (grad_diag && loss->gdiag_out == grad_diag), "ns backward: a W2 job defines the diagonal seed it rides with");
    {
        static Option full_opt("ST_NS_FULL_BACKWARD", 0);
        if (grad_diag && !full_opt.get() && (ns_chain_mask() & 8) && ws.chain_sync) {
            NsChainLaunch launch{};
            launch.count = 1;
            NsChainJob& j = launch.job[0];
            j = ns_chain_job(ws, n);
            j.backward = 1;
            j.root = const_cast<float*>(root); j.grad_m = grad_m;
            j.gdiag_dev = grad_diag;
            if (loss) j.loss = *loss;
            return launch_ns_chain(launch, s);
        }
        if (grad_diag && !full_opt.get() && ns_f16_applies(n) && ws.planes)
            return ns_sqrt_backward_diag_f16(root, grad_diag, grad_m, n, ws, s, loss, root_partials);
    }
    // norm_z = ||z||_F; a = z / norm_z; q = grad / norm_z                        (sqrtm.py:38-41)
    if (launch_ns_prepare(root, n, ws.scalars + 1, ws.scalars + 8, ws.a0, grad_diag ? nullptr : grad_root, grad_diag,
                          ws.q0, s, false, loss, root_partials))
        return 1;
    float *a = ws.a0, *an = ws.a1, *q = ws.q0, *qn = ws.q1;
    // grad_diag: the incoming gradient is a multiple of I (the W2 style loss: d trace(root) = I).  Then q_0
    // commutes with a_0 and every later q, a is a polynomial in a_0, so the commutator a^T q - q a of sqrtm.py:44
    // is identically zero (a is symmetric up to rounding; what the reference accumulates there is O(eps) noise)
    // and the step reduces to q <- q (3I - a a) / 2: three products per step instead of six.  The general
    // operator (grad_root, st_op_sqrtm_ns_backward) keeps the full recurrence.  ST_NS_FULL_BACKWARD=1 forces it.
    static Option force_full_opt("ST_NS_FULL_BACKWARD", 0);
    const bool force_full = force_full_opt.get() != 0;
    const bool reduced = grad_diag != nullptr && !force_full;
    for (int it = 0; it < 12; ++it) {
        const bool last = (it == 11);
        GemmBatch b1{};
        b1.n = n; b1.count = reduced ? 1 : 3;
        b1.p[0] = plain(a, a, ws.e);                                // eye_a_a = 3I - a @ a    (:43)
        b1.p[0].epilogue = EPI_IDENT_MINUS; b1.p[0].ci = 3.f; b1.p[0].c = 1.f;
        if (!reduced) {
            b1.p[1] = plain(a, q, ws.atq, 1.f, /*ta=*/1);           // a^T @ q
            b1.p[2] = plain(q, a, ws.qa);                           // q @ a
        }
        if (launch_gemm_batch(b1, s)) return 1;
        GemmBatch b2{};
        b2.n = n; b2.count = last ? 1 : 2;
        // q = (q @ eye_a_a - a^T @ (a^T @ q - q @ a)) / 2  (:44); the final "/ 2" (:47) folds in
        GemmProblem& pq = b2.p[0];
        pq = plain(q, ws.e, last ? grad_m : qn, last ? 0.25f : 0.5f);
        if (!reduced) {
            pq.epilogue = EPI_DIFF;
            pq.a2 = a; pq.ta2 = 1; pq.b2 = ws.atq; pq.b2sub = ws.qa;
        }
        if (!last) b2.p[1] = plain(a, ws.e, an, 0.5f);              // a = a @ eye_a_a / 2     (:46)
        if (launch_gemm_batch(b2, s)) return 1;
        float* tmp = a; a = an; an = tmp;
        tmp = q; q = qn; qn = tmp;
    }
    return 0;
}

}  // namespace st
