// Dense n x n x n products (n = 64..512) for the Wasserstein-2 style loss, and the two fixed
// 12-step recurrences built from them:
//   ns_sqrt_forward   = sqrtm.sqrtm_ns                          (reference sqrtm.py:9-25)
//   ns_sqrt_backward  = _MatrixSquareRootNSLyap.backward        (reference sqrtm.py:36-47)
// Parity requires the recurrences step for step (same normalisation, same 12 iterations, same
// operand order) because NS-12 is NOT converged on the ill-conditioned covariances it sees
//  - so every product below is a separate fp32 GEMM with the elementwise
// ops of the reference applied in the same order in its epilogue.
//
// The chain is latency bound (<= 268 MFLOP per product, ~60 dependent launches per layer), so the
// kernel favours short critical path over peak rate: one 32x32 output tile per workgroup, the K
// range split over the 4 waves (one per SIMD) and combined through LDS, operands read straight
// from L2 (a 64-cycle fp32 MFMA leaves ample time), independent products of one recurrence step
// batched in one launch (blockIdx.y).  K is visited in blocks of 8: lanes 0-31 own k..k+3, lanes
// 32-63 own k+4..k+7; a k-contiguous operand is one 16-byte load, a k-strided one four dwords.
#include "st_common.h"

namespace st {
namespace {

__device__ __forceinline__ void load_a(const float* __restrict__ a, int trans, int n, int m, int k,
                                       float (&v)[4]) {
    if (!trans) {                      // op(A)[m][k] = A[m][k]: k contiguous
        const f32x4 t = *reinterpret_cast<const f32x4*>(a + (size_t)m * n + k);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {                           // op(A)[m][k] = A[k][m]: lanes contiguous along m
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a[(size_t)(k + e) * n + m];
    }
}

__device__ __forceinline__ void load_b(const float* __restrict__ b, int trans, int n, int col, int k,
                                       float (&v)[4]) {
    if (!trans) {                      // op(B)[k][c] = B[k][c]: lanes contiguous along c
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = b[(size_t)(k + e) * n + col];
    } else {                           // op(B)[k][c] = B[c][k]: k contiguous
        const f32x4 t = *reinterpret_cast<const f32x4*>(b + (size_t)col * n + k);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    }
}

// One product, this wave's K range [kbeg, kbeg + N/4): a round of up to 8 k-blocks is loaded in
// one burst (all loads in flight together, one exposed L2 latency per round), then consumed.
template <int N, int TA, int TB, bool SUB>
__device__ __forceinline__ void accumulate_product(f32x16& acc, const float* __restrict__ a,
                                                    const float* __restrict__ b,
                                                    const float* __restrict__ bsub, int row, int col,
                                                    int kbeg, int half) {
    constexpr int NB = N / 32;                   // 8-wide k-blocks per wave
    constexpr int RB = NB < 8 ? NB : 8;          // blocks per round
#pragma unroll 1
    for (int round = 0; round < NB / RB; ++round) {
        float av[RB][4], bv[RB][4], sv[SUB ? RB : 1][4];
#pragma unroll
        for (int u = 0; u < RB; ++u) {
            const int k = kbeg + (round * RB + u) * 8 + 4 * half;
            load_a(a, TA, N, row, k, av[u]);
            load_b(b, TB, N, col, k, bv[u]);
            if constexpr (SUB) load_b(bsub, TB, N, col, k, sv[u]);
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float bb = bv[u][e];
                if constexpr (SUB) bb = bb - sv[u][e];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][e], bb, acc, 0, 0, 0);
            }
        }
    }
}

template <int N>
__global__ __launch_bounds__(256) void gemm_batch_kernel(GemmBatch batch) {
    __shared__ float red[2][4][16][64];
    const GemmProblem& pr = batch.p[blockIdx.y];
    constexpr int n = N;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    constexpr int nt = n / 32;
    const int m0 = (blockIdx.x / nt) * 32, n0 = (blockIdx.x % nt) * 32;
    const int kbeg = wave * (n / 4);
    const bool two = (pr.epilogue == EPI_DIFF);

    f32x16 acc1, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc1[r] = 0.f; acc2[r] = 0.f; }

    const int row = m0 + l31, col = n0 + l31;
    if (!pr.ta1 && !pr.tb1)
        accumulate_product<N, 0, 0, false>(acc1, pr.a1, pr.b1, nullptr, row, col, kbeg, half);
    else if (pr.ta1 && !pr.tb1)
        accumulate_product<N, 1, 0, false>(acc1, pr.a1, pr.b1, nullptr, row, col, kbeg, half);
    else
        accumulate_product<N, 0, 1, false>(acc1, pr.a1, pr.b1, nullptr, row, col, kbeg, half);
    if (two)   // P2 = a2^T @ (b2 - b2sub): the only form the Lyapunov recurrence needs
        accumulate_product<N, 1, 0, true>(acc2, pr.a2, pr.b2, pr.b2sub, row, col, kbeg, half);

#pragma unroll
    for (int r = 0; r < 16; ++r) {
        red[0][wave][r][lane] = acc1[r];
        if (two) red[1][wave][r][lane] = acc2[r];
    }
    __syncthreads();

    float dscale = 1.f;
    if (pr.epilogue == EPI_DEV_SQRT_SCALE) dscale = sqrtf(pr.dev_scalar[0]);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int r = wave * 4 + rr;
        const float s1 = (red[0][0][r][lane] + red[0][1][r][lane]) + (red[0][2][r][lane] + red[0][3][r][lane]);
        const int orow = m0 + rr + 8 * wave + 4 * half;
        const int ocol = n0 + l31;
        float v;
        if (pr.epilogue == EPI_SCALE) {
            v = s1 * pr.c;
        } else if (pr.epilogue == EPI_IDENT_MINUS) {
            v = ((orow == ocol ? pr.ci : 0.f) - s1) * pr.c;
        } else if (pr.epilogue == EPI_DIFF) {
            const float s2 =
                (red[1][0][r][lane] + red[1][1][r][lane]) + (red[1][2][r][lane] + red[1][3][r][lane]);
            v = (s1 - s2) * pr.c;
        } else {
            v = s1 * dscale;
        }
        pr.d[(size_t)orow * n + ocol] = v;
    }
}

GemmProblem plain(const float* a, const float* b, float* d, float c = 1.f, int ta = 0, int tb = 0) {
    GemmProblem p{};
    p.a1 = a; p.b1 = b; p.d = d; p.ta1 = ta; p.tb1 = tb; p.epilogue = EPI_SCALE; p.c = c;
    return p;
}

}  // namespace

int launch_gemm_batch(const GemmBatch& b, hipStream_t s) {
    ST_REQUIRE(b.count >= 1 && b.count <= 3, "gemm: batch count out of range");
    for (int i = 0; i < b.count; ++i) {
        ST_REQUIRE(!(b.p[i].ta1 && b.p[i].tb1), "gemm: A^T @ B^T is not implemented");
        ST_REQUIRE(b.p[i].epilogue != EPI_DIFF || (b.p[i].ta2 == 1), "gemm: second product must be A2^T @ (B2 - B2sub)");
    }
    const int nt = b.n / 32;
    const dim3 grid(nt * nt, b.count), block(256);
    switch (b.n) {
        case 64: hipLaunchKernelGGL(gemm_batch_kernel<64>, grid, block, 0, s, b); break;
        case 128: hipLaunchKernelGGL(gemm_batch_kernel<128>, grid, block, 0, s, b); break;
        case 256: hipLaunchKernelGGL(gemm_batch_kernel<256>, grid, block, 0, s, b); break;
        case 512: hipLaunchKernelGGL(gemm_batch_kernel<512>, grid, block, 0, s, b); break;
        default: ST_REQUIRE(false, "gemm: n must be 64, 128, 256 or 512 (got %d)", b.n);
    }
    ST_LAUNCH_CHECK();
    return 0;
}

size_t ns_workspace_floats(int n) { return (size_t)12 * n * n + 64; }

void ns_workspace_carve(NSWorkspace& ws, float* base, int n) {
    const size_t nn = (size_t)n * n;
    float** slots[] = {&ws.y0, &ws.y1, &ws.z0, &ws.z1, &ws.t,  &ws.a0,
                       &ws.a1, &ws.q0, &ws.q1, &ws.e,  &ws.atq, &ws.qa};
    for (int i = 0; i < 12; ++i) *slots[i] = base + i * nn;
    ws.scalars = base + 12 * nn;
}

int ns_sqrt_forward(const float* m, float* root, int n, NSWorkspace& ws, hipStream_t s) {
    const long long nn = (long long)n * n;
    // norm_a = a.pow(2).sum().sqrt(); y = a / norm_a; z = I                      (sqrtm.py:16-20)
    if (launch_frobenius(m, nn, ws.scalars + 0, s)) return 1;
    if (launch_div_by_dev_scalar(m, ws.scalars + 0, ws.y0, nn, s)) return 1;
    if (launch_identity(ws.z0, n, s)) return 1;
    float *y = ws.y0, *yn = ws.y1, *z = ws.z0, *zn = ws.z1;
    for (int it = 0; it < 12; ++it) {
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
        }
        if (launch_gemm_batch(b2, s)) return 1;
        float* tmp = y; y = yn; yn = tmp;
        tmp = z; z = zn; zn = tmp;
    }
    return 0;
}

int ns_sqrt_backward(const float* root, const float* grad_root, const float* grad_diag, float* grad_m, int n,
                     NSWorkspace& ws, hipStream_t s) {
    const long long nn = (long long)n * n;
    // norm_z = ||z||_F; a = z / norm_z; q = grad / norm_z                        (sqrtm.py:38-41)
    if (launch_frobenius(root, nn, ws.scalars + 1, s)) return 1;
    if (launch_div_by_dev_scalar(root, ws.scalars + 1, ws.a0, nn, s)) return 1;
    if (grad_diag) {
        if (launch_scaled_identity_div(grad_diag, ws.scalars + 1, ws.q0, n, s)) return 1;
    } else {
        if (launch_div_by_dev_scalar(grad_root, ws.scalars + 1, ws.q0, nn, s)) return 1;
    }
    float *a = ws.a0, *an = ws.a1, *q = ws.q0, *qn = ws.q1;
    for (int it = 0; it < 12; ++it) {
        const bool last = (it == 11);
        GemmBatch b1{};
        b1.n = n; b1.count = 3;
        b1.p[0] = plain(a, a, ws.e);                                // eye_a_a = 3I - a @ a    (:43)
        b1.p[0].epilogue = EPI_IDENT_MINUS; b1.p[0].ci = 3.f; b1.p[0].c = 1.f;
        b1.p[1] = plain(a, q, ws.atq, 1.f, /*ta=*/1);               // a^T @ q
        b1.p[2] = plain(q, a, ws.qa);                               // q @ a
        if (launch_gemm_batch(b1, s)) return 1;
        GemmBatch b2{};
        b2.n = n; b2.count = last ? 1 : 2;
        // q = (q @ eye_a_a - a^T @ (a^T @ q - q @ a)) / 2  (:44); the final "/ 2" (:47) folds in
        GemmProblem& pq = b2.p[0];
        pq = plain(q, ws.e, last ? grad_m : qn);
        pq.epilogue = EPI_DIFF; pq.c = last ? 0.25f : 0.5f;
        pq.a2 = a; pq.ta2 = 1; pq.b2 = ws.atq; pq.b2sub = ws.qa;
        if (!last) b2.p[1] = plain(a, ws.e, an, 0.5f);              // a = a @ eye_a_a / 2     (:46)
        if (launch_gemm_batch(b2, s)) return 1;
        float* tmp = a; a = an; an = tmp;
        tmp = q; q = qn; qn = tmp;
    }
    return 0;
}

}  // namespace st
