// fp16x3 form of the n = 512 Newton-Schulz chains (sqrtm.py:9-25 forward, :36-47 backward with a gradient that is a
// multiple of I): the same recurrences, step for step, with every n x n x n product evaluated on the 16-bit matrix
// pipe as h0 g0 + h0 g1 + h1 g0 over two fp16 planes per operand (22 significant bits, fp32 accumulation) - the
// arithmetic of the trunk convolutions (st_conv_split.hip), 3/16 of the fp32 matrix-pipe time.
//
// Why: the backward pass of the whole network waits for relu5_1's chain (~76 dependent 512^3 products), and an
// fp32 product costs 7.5 - 16 us per launch (gemm_staged_kernel: 4096 cycles of v_mfma_f32_32x32x2_f32 per tile plus
// an LDS staging round trip).  Here a launch is: 32 coalesced 1 KB loads per wave straight into MFMA operand
// registers, 24 MFMAs, one LDS reduction, one epilogue.
//
// Data layout ("role A" of an n x n matrix X, one array per plane): 1 KB blocks [n/32 row blocks][n/16 k blocks];
// inside a block lane l owns the 8 halves X[32 mb + (l & 31)][16 kb + 8 (l >> 5) + 0..7] - exactly the A operand
// of v_mfma_f32_32x32x16_f16, so a wave's 16-byte-per-lane load of a block IS the operand.  Role B of X = role A of
// X^T.  Every product's epilogue writes its result in the roles its consumers need (both, for the NS iterates),
// already split into planes: an element is converted once, by its producer, instead of by each of its 16 readers.
//
// Scaling: planes hold value * 2^e with e from an A-PRIORI bound of the matrix (the producer cannot know max |x|
// of a tensor it is still writing): y, a <= 1, t in [1, 1.5], E = 3I - a a <= 3 (spectral norms of the exact
// recurrences, entries are bounded by them), z_k <= 1.5^k, q_k <= |q_0| 1.5^k with |q_0| = |gdiag| / ||root||_F read
// from the device scalars.  The bound is placed in [2^13, 2^14) as everywhere else (two spare bits below fp16's
// overflow); a bound that is 2^k loose only raises the underflow floor to 2^(k-38) of the bound.
//
// Accuracy: the products differ from fp32 FMA chains by the plane residual (2^-22 per element) - measured on the
// CPU before this kernel was written (emulated planes inside the oracle's chain, relu4_1 / relu5_1 heads): style
// terms move 1e-5 ... 3e-5 relative, below the reference's own fp32-vs-fp64 floor (3e-5 ... 6e-5); gradients 2e-5.
// n <= 256 chains stay fp32: they are not on the critical path and relu1_1 carries 75 % of the style loss.
#include <utility>

#include "st_common.h"

namespace st {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kTilePitch = 36;      // floats; 144 B rows keep ds_read_b128 aligned and conflict-free
// The residual plane is stored times 2^11: with an a-priori bound the typical entry of an NS iterate sits 2^-10 ...
// 2^-16 below the bound (a 512 x 512 matrix of Frobenius norm 1 has entries ~2^-9, and the bound is a spectral
// one), where an unscaled residual falls into fp16's subnormal range (or is flushed) and the pair keeps only 11 ...
// 18 of its 22 bits - measured as 3 - 5 x the style-term deviation of the fp32 chains.  Scaled, h1 has the
// magnitude of h0's last bit times 2^11 ~ |x|, i.e. full precision wherever h0 is normal (28 binades below the
// bound); the cross products go to their own accumulator and are folded in with the exact factor 2^-11.
constexpr float kResidualScale = 2048.f;

__device__ __forceinline__ int resolve_exp(const NsScale& s) {
    if (s.num == nullptr) return s.exp;
    const float bound = fabsf(s.num[0] / s.den[0]) * s.mult;
    return scale_exp(__builtin_bit_cast(unsigned int, bound));
}

// One 32 x 32 fp32 tile in LDS -> its four 1 KB plane blocks (two k blocks of role A, two of role B); 256 threads,
// threads 0..127 role A, 128..255 role B.  tile[r][c] = X[32 mb + r][32 nb + c].
template <int N>
__device__ __forceinline__ void emit_planes(const float (*tile)[kTilePitch], int mb, int nb, const NsPlanesOut& out,
                                            int exp, int tid) {
    constexpr int KB = N / 16;
    if (tid >= 256) return;
    const int role = tid >> 7, j = (tid >> 6) & 1, lam = tid & 63;
    const int l31 = lam & 31, hi = lam >> 5;
    _Float16* d0 = role == 0 ? out.a0 : out.b0;
    _Float16* d1 = role == 0 ? out.a1 : out.b1;
    if (d0 == nullptr) return;
    float v[8];
    if (role == 0) {
        const f32x4 lo = *reinterpret_cast<const f32x4*>(&tile[l31][16 * j + 8 * hi]);
        const f32x4 up = *reinterpret_cast<const f32x4*>(&tile[l31][16 * j + 8 * hi + 4]);
        v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
        v[4] = up[0]; v[5] = up[1]; v[6] = up[2]; v[7] = up[3];
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = tile[16 * j + 8 * hi + i][l31];
    }
    const float sc = pow2f(exp);
    f16x8 h0, h1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = v[i] * sc;
        const _Float16 a = (_Float16)x;
        h0[i] = a;
        h1[i] = (_Float16)((x - (float)a) * kResidualScale);    // exact: |x - a| <= 2^-11 |x|, so |h1| <= |x|
    }
    const size_t off = role == 0 ? ((size_t)mb * KB + 2 * nb + j) * 512 + lam * 8
                                 : ((size_t)nb * KB + 2 * mb + j) * 512 + lam * 8;
    *reinterpret_cast<f16x8*>(d0 + off) = h0;
    *reinterpret_cast<f16x8*>(d1 + off) = h1;
}

// XCD-aware tile order for n = 512 (16 x 16 tiles): workgroup b runs on XCD b % 8; give each XCD a 4 x 8 block of
// tiles, so that its L2 fetches 4 A panels + 8 B panels (768 KB) instead of 16 + 2 (1.1 MB) from the fabric.
template <int N>
__device__ __forceinline__ void tile_of_block(int b, int& mb, int& nb) {
    constexpr int nt = N / 32;
    if (N == 512) {
        const int x = b & 7, j = b >> 3;
        mb = 4 * (x >> 1) + (j >> 3);
        nb = 8 * (x & 1) + (j & 7);
    } else {
        mb = b / nt;
        nb = b % nt;
    }
}

template <int N, int WV>
__global__ __launch_bounds__(WV * 64) void ns_gemm_f16_kernel(NsGemmBatch batch) {
    constexpr int KB = N / 16;            // 16-wide k blocks
    constexpr int KBW = KB / WV;          // ... per wave
    constexpr int RPT = 16 / WV;
    static_assert(KB % WV == 0 && 16 % WV == 0, "wave count must divide the k blocks and the accumulator");
    __shared__ float red[WV][16][64];
    __shared__ __attribute__((aligned(16))) float tile[32][kTilePitch];
    const NsGemmProblem& pr = batch.p[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    int mb, nb;
    tile_of_block<N>(blockIdx.x, mb, nb);

    // every operand block of this wave, issued up front: 4 KBW loads of 16 bytes per lane
    const size_t abase = ((size_t)mb * KB + wave * KBW) * 512 + lane * 8;
    const size_t bbase = ((size_t)nb * KB + wave * KBW) * 512 + lane * 8;
    f16x8 a0[KBW], a1[KBW], b0[KBW], b1[KBW];
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) {
        a0[kb] = *reinterpret_cast<const f16x8*>(pr.a.p0 + abase + kb * 512);
        b0[kb] = *reinterpret_cast<const f16x8*>(pr.b.p0 + bbase + kb * 512);
        a1[kb] = *reinterpret_cast<const f16x8*>(pr.a.p1 + abase + kb * 512);
        b1[kb] = *reinterpret_cast<const f16x8*>(pr.b.p1 + bbase + kb * 512);
    }
    const int ea = resolve_exp(pr.a.scale), eb = resolve_exp(pr.b.scale);
    const int eo = resolve_exp(pr.out.scale);
    float dscale = 1.f;
    if (pr.epilogue == EPI_DEV_SQRT_SCALE) dscale = sqrtf(pr.dev_scalar[0]);

    f32x16 acc, cross;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; cross[r] = 0.f; }
    // blocks are consumed in the order their loads were issued (the compiler places one vmcnt wait per block);
    // two accumulators: h0 g0, and the cross terms h0 g1' + h1' g0 whose planes carry the extra 2^11
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[kb], b0[kb], acc, 0, 0, 0);
        cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[kb], b1[kb], cross, 0, 0, 0);
        cross = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kb], b0[kb], cross, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += cross[r] * (1.f / kResidualScale);      // (h1 g1 <= 2^-24 |x y| is dropped:
                                                                                       // keeping it changed nothing, profiles/r02_ns_chains.md)

    // cross-wave K reduction in a fixed pairwise order; wave w finishes registers [w RPT, (w+1) RPT)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    const float unscale = pow2f(-(ea + eb));
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
        const float s1 = part[0] * unscale;             // exact: a power of two
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int orow = mb * 32 + trow, ocol = nb * 32 + l31;
        float v;
        if (pr.epilogue == EPI_SCALE) {
            v = s1 * pr.c;
        } else if (pr.epilogue == EPI_IDENT_MINUS) {
            v = ((orow == ocol ? pr.ci : 0.f) - s1) * pr.c;
        } else {
            v = s1 * dscale;
        }
        if (pr.d32) pr.d32[(size_t)orow * N + ocol] = v;
        tile[trow][l31] = v;
    }
    if (pr.out.a0 == nullptr && pr.out.b0 == nullptr) return;
    __syncthreads();
    emit_planes<N>(tile, mb, nb, pr.out, eo, tid);
}

// fp32 row-major matrices -> planes (the chain's entry: y0 / z0, a0 / q0)
template <int N>
__global__ __launch_bounds__(256) void ns_planes_from_f32_kernel(NsToPlanes job) {
    __shared__ __attribute__((aligned(16))) float tile[32][kTilePitch];
    const NsToPlanesItem& it = job.item[blockIdx.y];
    constexpr int nt = N / 32;
    const int mb = blockIdx.x / nt, nb = blockIdx.x % nt;
    const int tid = threadIdx.x;
    const int r = tid >> 3, c4 = (tid & 7) * 4;
    const f32x4 v = *reinterpret_cast<const f32x4*>(it.src + (size_t)(mb * 32 + r) * N + nb * 32 + c4);
    *reinterpret_cast<f32x4*>(&tile[r][c4]) = v;
    __syncthreads();
    emit_planes<N>(tile, mb, nb, it.out, resolve_exp(it.out.scale), tid);
}

// The Lyapunov backward chain's entry for grad_output = gdiag * I in ONE launch (was: sum of squares, ns_prepare_kernel,
// ns_planes_from_f32_kernel - and style_loss_value_kernel in front of them): norm_z = ||z||_F from the per-tile sums of
// squares the forward chain's last product left behind, a = z / norm_z and q = (gdiag / norm_z) I written straight as
// planes (sqrtm.py:38-41), and the head's W2 scalars by one of the workgroups.  grid (tiles, 2): y = 0 -> a, 1 -> q.
struct NsBackwardEntry {
    const float* root;
    const float* partials;
    int nparts;
    float* norm_out;
    const float* gdiag;                    // the seed on the device (not read when a W2 job rides along: it defines it)
    NsPlanesOut a_out, q_out;              // q_out.scale.mult: its bound is |gdiag / norm_z| * mult
    W2LossJob loss;
};
template <int N>
__global__ __launch_bounds__(256) void ns_backward_entry_kernel(NsBackwardEntry job) {
    __shared__ __attribute__((aligned(16))) float tile[32][kTilePitch];
    __shared__ float scratch[4];
    __shared__ float norm_sh;
    const int tid = threadIdx.x;
    float s = 0.f;
    for (int i = tid; i < job.nparts; i += 256) s += job.partials[i];
    s = block_sum_256(s, scratch);
    if (tid == 0) {
        norm_sh = sqrtf(s);
        if (blockIdx.x == 0 && blockIdx.y == 0) job.norm_out[0] = norm_sh;
    }
    __syncthreads();
    const float d = norm_sh;
    constexpr int nt = N / 32;
    const int mb = blockIdx.x / nt, nb = blockIdx.x % nt;
    const int r = tid >> 3, c4 = (tid & 7) * 4;
    if (blockIdx.y == 0) {
        f32x4 v = *reinterpret_cast<const f32x4*>(job.root + (size_t)(mb * 32 + r) * N + nb * 32 + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] / d;
        *reinterpret_cast<f32x4*>(&tile[r][c4]) = v;
        __syncthreads();
        emit_planes<N>(tile, mb, nb, job.a_out, resolve_exp(job.a_out.scale), tid);
    } else {
        const float gd = job.loss.loss_out ? w2_gdiag(job.loss) : job.gdiag[0];
        const float dv = gd / d;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (mb == nb && r == c4 + e) ? dv : 0.f;
        *reinterpret_cast<f32x4*>(&tile[r][c4]) = v;
        __syncthreads();
        const float bound = fabsf(gd / d) * job.q_out.scale.mult;
        emit_planes<N>(tile, mb, nb, job.q_out, scale_exp(__builtin_bit_cast(unsigned int, bound)), tid);
        if (job.loss.loss_out && blockIdx.x == 0) w2_loss_block(job.loss, scratch);
    }
}

}  // namespace

bool ns_f16_applies(int n) {
    static Option on("ST_NS_F16", 1);            // 0: fp32 chains everywhere (A/B runs, parity tests)
    static Option min_n("ST_NS_F16_MIN_N", 512);
    return on.get() != 0 && n >= min_n.get() && (n == 512 || n == 256);
}

int launch_ns_gemm_f16(const NsGemmBatch& b, hipStream_t s) {
    ST_REQUIRE(b.count >= 1 && b.count <= 2, "ns gemm (fp16x3): batch count out of range");
    const int nt = b.n / 32;
    const dim3 grid(nt * nt, b.count);
    // (8 waves per tile instead of 4: 6.1 instead of 6.5 us per launch in isolation, neutral in the iteration -
    // 256^2 638 / 642 it/s, 512^2 410 / 411, profiles/r02_ns_chains.md; not kept)
    switch (b.n) {
        case 512: hipLaunchKernelGGL((ns_gemm_f16_kernel<512, 4>), grid, dim3(256), 0, s, b); break;
        case 256: hipLaunchKernelGGL((ns_gemm_f16_kernel<256, 4>), grid, dim3(256), 0, s, b); break;
        default: ST_REQUIRE(false, "ns gemm (fp16x3): n must be 256 or 512 (got %d)", b.n);
    }
    ST_LAUNCH_CHECK();
    return 0;
}

namespace {
struct Slot {                              // plane arrays of one matrix slot: role A / role B x plane 0 / 1
    _Float16 *a0, *a1, *b0, *b1;
};
Slot slot_of(const NSWorkspace& ws, int n, int index) {
    const size_t nn = (size_t)n * n;
    _Float16* base = ws.planes + (size_t)index * 4 * nn;
    return Slot{base, base + nn, base + 2 * nn, base + 3 * nn};
}
NsScale host_scale(float bound) { return NsScale{scale_exp(__builtin_bit_cast(unsigned int, bound)), nullptr, nullptr, 0.f}; }
NsPlanes role_a(const Slot& m, NsScale sc) { return NsPlanes{m.a0, m.a1, sc}; }
NsPlanes role_b(const Slot& m, NsScale sc) { return NsPlanes{m.b0, m.b1, sc}; }
NsPlanesOut out_both(const Slot& m, NsScale sc) { return NsPlanesOut{m.a0, m.a1, m.b0, m.b1, sc}; }
NsPlanesOut out_a(const Slot& m, NsScale sc) { return NsPlanesOut{m.a0, m.a1, nullptr, nullptr, sc}; }
NsPlanesOut out_b(const Slot& m, NsScale sc) { return NsPlanesOut{nullptr, nullptr, m.b0, m.b1, sc}; }
NsGemmProblem product(NsPlanes a, NsPlanes b, NsPlanesOut out, float* d32, int epilogue, float c, float ci = 0.f) {
    NsGemmProblem p{};
    p.a = a; p.b = b; p.out = out; p.d32 = d32; p.epilogue = epilogue; p.c = c; p.ci = ci;
    return p;
}
float pow15(int k) {
    float v = 1.f;
    for (int i = 0; i < k; ++i) v *= 1.5f;
    return v;
}
}  // namespace

// sqrtm.sqrtm_ns (sqrtm.py:9-25), products in fp16x3.  A-priori bounds (spectral norms of the exact recurrence,
// which bound the entries): y <= 1, t = (3I - z y) / 2 in [1, 1.5], z_k <= 1.5^k; each with a factor 2 of margin.
int ns_sqrt_forward_f16(const float* m, float* root, int n, NSWorkspace& ws, hipStream_t s) {
    // norm_a = a.pow(2).sum().sqrt(); y = a / norm_a; z = I                      (sqrtm.py:16-20)
    if (launch_ns_prepare(m, n, ws.scalars + 0, ws.scalars + 8, ws.y0, nullptr, nullptr, ws.z0, s)) return 1;
    Slot y = slot_of(ws, n, 0), yn = slot_of(ws, n, 1), z = slot_of(ws, n, 2), zn = slot_of(ws, n, 3);
    const Slot t = slot_of(ws, n, 4);
    const NsScale sy = host_scale(2.f), st = host_scale(4.f);
    NsScale sz = host_scale(2.f);
    NsToPlanes tp{};
    tp.count = 2;
    tp.item[0] = NsToPlanesItem{ws.y0, out_both(y, sy)};
    tp.item[1] = NsToPlanesItem{ws.z0, out_both(z, sz)};
    if (launch_ns_planes_from_f32(tp, n, s)) return 1;
    for (int it = 0; it < 12; ++it) {
        const bool last = (it == 11);
        NsGemmBatch b1{};
        b1.n = n; b1.count = 1;                                     // t = (3I - z @ y) / 2   (:22)
        b1.p[0] = product(role_a(z, sz), role_b(y, sy), out_both(t, st), nullptr, EPI_IDENT_MINUS, 0.5f, 3.f);
        if (launch_ns_gemm_f16(b1, s)) return 1;
        NsGemmBatch b2{};
        b2.n = n;
        if (!last) {
            const NsScale szn = host_scale(2.f * pow15(it + 1));
            b2.count = 2;
            b2.p[0] = product(role_a(y, sy), role_b(t, st), out_both(yn, sy), nullptr, EPI_SCALE, 1.f);   // y = y @ t (:23)
            b2.p[1] = product(role_a(t, st), role_b(z, sz), out_both(zn, szn), nullptr, EPI_SCALE, 1.f);  // z = t @ z (:24)
            sz = szn;
        } else {
            b2.count = 1;                                           // return y * sqrt(norm_a) (:25)
            b2.p[0] = product(role_a(y, sy), role_b(t, st), NsPlanesOut{}, root, EPI_DEV_SQRT_SCALE, 1.f);
            b2.p[0].dev_scalar = ws.scalars + 0;
        }
        if (launch_ns_gemm_f16(b2, s)) return 1;
        std::swap(y, yn);
        std::swap(z, zn);
    }
    return 0;
}

// the twelve recurrence steps; the planes of a_0 (slot 0, both roles) and q_0 (slot 2, role A) are in place
static int ns_backward_diag_f16_steps(float* grad_m, int n, NSWorkspace& ws, const float* grad_diag, hipStream_t s) {
    Slot a = slot_of(ws, n, 0), an = slot_of(ws, n, 1), q = slot_of(ws, n, 2), qn = slot_of(ws, n, 3);
    const Slot e = slot_of(ws, n, 4);
    const NsScale sa = host_scale(2.f), se = host_scale(4.f);
    auto q_scale = [&](int k) { return NsScale{0, grad_diag, ws.scalars + 1, 2.f * pow15(k)}; };
    for (int it = 0; it < 12; ++it) {
        const bool last = (it == 11);
        NsGemmBatch b1{};
        b1.n = n; b1.count = 1;                                     // eye_a_a = 3I - a @ a    (:43)
        b1.p[0] = product(role_a(a, sa), role_b(a, sa), out_b(e, se), nullptr, EPI_IDENT_MINUS, 1.f, 3.f);
        if (launch_ns_gemm_f16(b1, s)) return 1;
        NsGemmBatch b2{};
        b2.n = n; b2.count = last ? 1 : 2;
        // q = q @ eye_a_a / 2  (:44 without the vanishing commutator); the final "/ 2" (:47) folds into the last one
        b2.p[0] = last ? product(role_a(q, q_scale(it)), role_b(e, se), NsPlanesOut{}, grad_m, EPI_SCALE, 0.25f)
                       : product(role_a(q, q_scale(it)), role_b(e, se), out_a(qn, q_scale(it + 1)), nullptr, EPI_SCALE, 0.5f);
        if (!last)                                                  // a = a @ eye_a_a / 2     (:46)
            b2.p[1] = product(role_a(a, sa), role_b(e, se), out_both(an, sa), nullptr, EPI_SCALE, 0.5f);
        if (launch_ns_gemm_f16(b2, s)) return 1;
        std::swap(a, an);
        std::swap(q, qn);
    }
    return 0;
}

// _MatrixSquareRootNSLyap.backward (sqrtm.py:36-47) for grad_output = gdiag * I, reduced form (the commutator
// a^T (a^T q - q a) vanishes, see ns_sqrt_backward in st_smallgemm.hip), products in fp16x3.  Bounds: a <= 1,
// E = 3I - a a <= 3, q_k <= |gdiag / ||root||_F| 1.5^k (E / 2 has its spectrum in [1, 1.5]).
int ns_sqrt_backward_diag_f16(const float* root, const float* grad_diag, float* grad_m, int n, NSWorkspace& ws,
                              hipStream_t s, const W2LossJob* loss, int root_partials) {
    const Slot a = slot_of(ws, n, 0), q = slot_of(ws, n, 2);
    const NsScale sa = host_scale(2.f);
    auto q_scale = [&](int k) { return NsScale{0, grad_diag, ws.scalars + 1, 2.f * pow15(k)}; };
    // norm_z = ||z||_F; a = z / norm_z; q = grad / norm_z                        (sqrtm.py:38-41)
    ST_REQUIRE(!loss || loss->gdiag_out == grad_diag, "ns backward (fp16x3): the W2 job must define this chain's seed");
    static Option fused_entry("ST_NS_FUSED_ENTRY", 1);
    if (!fused_entry.get()) {          // the round-2 form: (loss scalars,) sums of squares, a0 / q0 in fp32, planes from them
        if (loss && launch_style_loss_value(loss->mean, loss->mean_t, loss->cov, loss->cov_t, loss->root, loss->n, loss->weight,
                                            loss->loss_out, loss->gdiag_out, s))
            return 1;
        if (launch_ns_prepare(root, n, ws.scalars + 1, ws.scalars + 8, ws.a0, nullptr, grad_diag, ws.q0, s, false, nullptr,
                              root_partials))
            return 1;
        NsToPlanes tp{};
        tp.count = 2;
        tp.item[0] = NsToPlanesItem{ws.a0, out_both(a, sa)};
        tp.item[1] = NsToPlanesItem{ws.q0, out_a(q, q_scale(0))};
        if (launch_ns_planes_from_f32(tp, n, s)) return 1;
        return ns_backward_diag_f16_steps(grad_m, n, ws, grad_diag, s);
    }
    int ready = root_partials;         // (> 0: the forward chain's last product left the tiles' sums of squares)
    if (ready <= 0 && launch_sumsq_partials(root, (long long)n * n, ws.scalars + 8, &ready, s)) return 1;
    NsBackwardEntry job{};
    job.root = root; job.partials = ws.scalars + 8; job.nparts = ready; job.norm_out = ws.scalars + 1;
    job.gdiag = grad_diag;
    job.a_out = out_both(a, sa);
    job.q_out = out_a(q, q_scale(0));
    if (loss) job.loss = *loss;
    {
        const int nt = n / 32;
        const dim3 grid(nt * nt, 2);
        switch (n) {
            case 512: hipLaunchKernelGGL(ns_backward_entry_kernel<512>, grid, dim3(256), 0, s, job); break;
            case 256: hipLaunchKernelGGL(ns_backward_entry_kernel<256>, grid, dim3(256), 0, s, job); break;
            default: ST_REQUIRE(false, "ns backward (fp16x3): n must be 256 or 512 (got %d)", n);
        }
        ST_LAUNCH_CHECK();
    }
    return ns_backward_diag_f16_steps(grad_m, n, ws, grad_diag, s);
}

int launch_ns_planes_from_f32(const NsToPlanes& job, int n, hipStream_t s) {
    ST_REQUIRE(job.count >= 1 && job.count <= 2, "ns planes: item count out of range");
    const int nt = n / 32;
    const dim3 grid(nt * nt, job.count);
    switch (n) {
        case 512: hipLaunchKernelGGL(ns_planes_from_f32_kernel<512>, grid, dim3(256), 0, s, job); break;
        case 256: hipLaunchKernelGGL(ns_planes_from_f32_kernel<256>, grid, dim3(256), 0, s, job); break;
        default: ST_REQUIRE(false, "ns planes: n must be 256 or 512 (got %d)", n);
    }
    ST_LAUNCH_CHECK();
    return 0;
}

}  // namespace st
