// The two Newton-Schulz recurrences of a style head (sqrtm.py:9-25 forward, :36-47 backward for a gradient that is a
// multiple of I) as ONE persistent launch per head - or one for the three shallow heads together - instead of ~47 dependent
// launches per head (round 5).
//
// Why.  At 512^2 the trunk idles 0.7 ms between its forward and backward pass while relu5_1's head runs ~50 dependent
// n = 512 launches at 12.5 us each in situ (7.5 isolated); relu4_1's and the shallow heads' chains add another ~130 launches
// to the same window, and what they cost each other is launch traffic through the command processor, not CU time
// (profiles/r03_head_window.md section 6: relu5_1's chains alone run at their isolated speed).  Rounds 3 / 4 priced a
// persistent kernel by a device-wide barrier with release / acquire FENCES - buffer_wbl2 / buffer_inv of an XCD's L2 per
// workgroup: 5 - 7 us over 256 workgroups, what a launch boundary costs - and did not build it.  Round 5 measured the barrier
// WITHOUT cache maintenance (tools/grid_barrier2.py, profiles/r05_ns_chain.md): every iterate is written with agent-scope
// (sc1) stores and read with agent-scope loads, which are coherent at the memory side by themselves, so an arrival is
// "s_waitcnt vmcnt(0), one relaxed atomic" and a release is the poll: 2.0 us per barrier over 136 workgroups (two levels,
// one counter per XCD), 1.4 - 1.8 us over <= 64, no stale read in 300 rounds x 5 sizes.
//
// What.  Every iterate of both recurrences is a polynomial in ONE symmetric matrix, so all of them are symmetric and
// commute in exact arithmetic (what the reference accumulates in their antisymmetric parts is rounding noise).  The kernel
// computes only the tile pairs ti <= tj of every product and writes the mirror image - the scheme of st_gram.hip - i.e.
// 136 instead of 256 tiles of a 512^3 product: -47 % matrix work and operand traffic, iterates exactly symmetric, and both
// operands of every product are read ROW-wise (B = B^T): 16-byte loads, conflict-free ds_read_b128.  One workgroup
// (4 waves, K split four ways, v_mfma_f32_32x32x2_f32 = true fp32 FMA chains, forward AND backward) per tile pair, resident
// for the whole chain; the two products of a step that share an operand (y t, z t / q E, a E) are one step.  A step is:
// all operand loads of the wave issued up front -> wave-private LDS images -> MFMA -> cross-wave reduction -> epilogue
// (the reference's elementwise operations in its order) -> tile + mirror stored write-through -> grid barrier.
// The recurrences are followed step for step (NS-12 is not converged: DESIGN.md section 3): same normalisations, same
// twelve iterations, same products; what differs from the launch-per-product form is the summation order inside a
// product (K split over 4 waves for every n) and the symmetrisation.  Validated like the first-step shortcut and the
// reduced Lyapunov recurrence before it: tests/test_kernels_gpu.py (reference KATs, oracle), tools/ns_accuracy.py against
// float64, the closure goldens at unchanged tolerances.
//
// Safety.  A persistent kernel that spins on other workgroups needs them co-resident: at most 136 workgroups of 256 threads
// and < 64 KB LDS per launch (the chip holds > 1000), nothing they wait for is behind them in a queue, and every poll is
// bounded (50 ms of s_memrealtime): a workgroup that gives up raises the job's error word, every other one leaves at its
// next poll, the results are filled with NaN (a NaN loss is loud) and the launcher's next call reports it.
#include <cmath>
#include <cstdlib>
#include <mutex>

#include "st_common.h"

namespace st {
namespace {

typedef unsigned int u32x4_t __attribute__((__vector_size__(16)));
constexpr int kCoherent = 16;                 // cache policy of every load / store of an iterate: sc1 = agent scope
constexpr int kPitch = 36;                    // floats per staged row: 144 B keeps ds_read_b128 aligned and conflict-free
constexpr int kImg = 32 * kPitch;             // floats per staged operand image (32 rows x <= 32 k)
constexpr int kTilePitch = 33;
constexpr int kSyncLine = 64;                 // unsigned ints per 256-byte line of the barrier words
constexpr int kSyncUints = 24 * kSyncLine;    // line 0: flat / top counter, 1 .. 8: group counters, 16 .. 23: group flags
constexpr unsigned long long kPollLimit = 5000000ull;       // 50 ms of the 100 MHz s_memrealtime

constexpr int kDmaImg = 2 * 1024;              // floats of one DMA image pair: A [32 rows][32 k], then B, 16-byte chunks swizzled
struct Lds {
    // wave-private operand images; the cross-wave reduction reuses the space.  Register-staged form: two padded images
    // (2 x kImg floats) per wave; LDS-DMA form: two image PAIRS per wave (round q fills pair q & 1 while pair (q + 1) & 1 feeds
    // the matrix pipe)
    float stage[4][2][kDmaImg];
    float tile[2][32][kTilePitch];            // finished tiles of the step's (<= 2) products
    float scratch[8];
    unsigned int flag;
    float bcast;
};
static_assert(kDmaImg >= 16 * 64, "a wave's share of the reduction buffer must fit one image pair");

template <int N>
struct Geo {
    static constexpr int KW = N / 4;                      // k range of a wave
    static constexpr int RK = KW < 32 ? KW : 32;          // k per round
    static constexpr int NR = KW / RK;                    // rounds: 4, 2, 1, 1 for n = 512, 256, 128, 64
    static constexpr int LPS = RK / 8;                    // 16-byte loads per lane and slice
    static constexpr int LPRow = RK / 4;                  // lanes per staged row
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t matrix_rsrc(const float* base, int n) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, n * n * 4, 0x00020000);
}
__device__ __forceinline__ f32x4 load16(__amdgpu_buffer_rsrc_t rs, int float_index) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, float_index * 4, 0, kCoherent));
}
// ... through the L2 (job.l2_loads): a plain load; the workgroup has executed an agent-scope acquire (buffer_inv sc1: this
// CU's L1 and this XCD's L2 drop what other XCDs may have rewritten) after the barrier that published the data
__device__ __forceinline__ f32x4 load16_cached(__amdgpu_buffer_rsrc_t rs, int float_index) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, float_index * 4, 0, 0));
}
__device__ __forceinline__ void store16(__amdgpu_buffer_rsrc_t rs, int float_index, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), rs, float_index * 4, 0, kCoherent);
}
__device__ __forceinline__ float load_coherent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_coherent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The k range of one wave of a 32-row panel (rows row0 .. row0 + 31 of a row-major symmetric matrix), every load in flight
// at once: NR x LPS 16-byte loads per lane, 128 contiguous bytes per 8 (or 4) lanes.
template <int N>
struct Panel {
    f32x4 v[Geo<N>::NR][Geo<N>::LPS];
};
template <int N, bool CACHED>
__device__ __forceinline__ void panel_load(Panel<N>& p, const float* base, int row0, int wave, int lane) {
    using G = Geo<N>;
    const __amdgpu_buffer_rsrc_t rs = matrix_rsrc(base, N);
#pragma unroll
    for (int r = 0; r < G::NR; ++r)
#pragma unroll
        for (int i = 0; i < G::LPS; ++i) {
            const int row = lane / G::LPRow + (64 / G::LPRow) * i, c4 = lane % G::LPRow;
            const int at = (row0 + row) * N + wave * G::KW + r * G::RK + c4 * 4;
            p.v[r][i] = CACHED ? load16_cached(rs, at) : load16(rs, at);
        }
}

// acc += A panel x B panel^T over this wave's k range.  `my` = this wave's two LDS images.  MFMA e of 8-block kb takes
// k = 8 kb + 4 (lane >> 5) + e for both operands, row / column lane & 31 (the order of st_smallgemm.hip's kernels).
template <int N>
__device__ __forceinline__ void panel_mfma(f32x16& acc, const Panel<N>& a, const Panel<N>& b, float* my, int lane) {
    using G = Geo<N>;
    const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
    for (int r = 0; r < G::NR; ++r) {
#pragma unroll
        for (int i = 0; i < G::LPS; ++i) {
            const int row = lane / G::LPRow + (64 / G::LPRow) * i, c4 = lane % G::LPRow;
            *reinterpret_cast<f32x4*>(my + row * kPitch + c4 * 4) = a.v[r][i];
            *reinterpret_cast<f32x4*>(my + kImg + row * kPitch + c4 * 4) = b.v[r][i];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kb = 0; kb < G::RK / 8; ++kb) {
            const f32x4 ta = *reinterpret_cast<const f32x4*>(my + l31 * kPitch + kb * 8 + 4 * half);
            const f32x4 tb = *reinterpret_cast<const f32x4*>(my + kImg + l31 * kPitch + kb * 8 + 4 * half);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[e], tb[e], acc, 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- operands by LDS-DMA (operands through the L2 only) ----------------------------------------------------------------------
// One round = 32 k of this wave's k range of both panels: 8 global_load_lds_dwordx4 (64 lanes x 16 bytes = 8 rows of an
// image each), no registers.  The DMA writes base + 16 lane linearly, so the bank swizzle goes on the SOURCE address: the
// 16-byte chunk c of row r lands in slot 8 r + (c ^ ((r >> 1) & 7)) - sixteen consecutive rows of one chunk column cover the
// 64 banks exactly once (ds_read_b128 conflict-free, like the padded pitch of the register-staged form).
template <int N>
__device__ __forceinline__ void dma_round(const float* a, const float* bt, int m0, int n0, int kcol, float* pair, int lane) {
    const int rsub = lane >> 3, pc = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * i + rsub;
        const int c = pc ^ ((row >> 1) & 7);
        const float* sa = a + (size_t)(m0 + row) * N + kcol + 4 * c;
        const float* sb = bt + (size_t)(n0 + row) * N + kcol + 4 * c;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                         (__attribute__((address_space(3))) void*)(pair + i * 256), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                         (__attribute__((address_space(3))) void*)(pair + 1024 + i * 256), 16, 0, 0);
    }
}
// acc += the round's 32 k; MFMA e of 8-block kb takes k = 8 kb + 4 (lane >> 5) + e: the order of panel_mfma, bit for bit
__device__ __forceinline__ void dma_mfma(f32x16& acc, const float* pair, int lane) {
    const int l31 = lane & 31, half = lane >> 5, sw = (l31 >> 1) & 7;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        const int slot = l31 * 8 + ((2 * kb + half) ^ sw);
        const f32x4 ta = *reinterpret_cast<const f32x4*>(pair + slot * 4);
        const f32x4 tb = *reinterpret_cast<const f32x4*>(pair + 1024 + slot * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ta[e], tb[e], acc, 0, 0, 0);
    }
}

// ---- the grid barrier of one job (profiles/r05_ns_chain.md) ---------------------------------------------------------------
struct Grid {
    unsigned int* words;       // this launch's barrier words (zero when the launch starts)
    unsigned int* error;       // the job's error word
    int nwg, wg;
    unsigned int round;
    bool dead;
    bool acquire;              // operands are read through the L2: every barrier ends with an agent-scope acquire
};
// Every thread's coherent stores are acknowledged (vmcnt), then one arrival per workgroup; <= 64 workgroups: one counter;
// more: a counter per group of wg % 8 (an XCD's workgroups under round-robin dispatch), the last arriver of a group
// arrives at the top counter, the last one there raises every group's flag.  All atomics relaxed at agent scope.
__device__ __forceinline__ void grid_sync(Grid& g, Lds& lds) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++g.round;
    if (threadIdx.x == 0 && !g.dead) {
        const unsigned int round = g.round;
        const int groups = g.nwg > 64 ? 8 : 0;
        unsigned int* flag = g.words;
        unsigned int target = round * (unsigned int)g.nwg;
        if (groups) {
            const int grp = g.wg % groups;
            const unsigned int members = (unsigned int)((g.nwg - grp + groups - 1) / groups);
            const unsigned int prev = __hip_atomic_fetch_add(g.words + kSyncLine * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == round * members - 1) {
                const unsigned int top = __hip_atomic_fetch_add(g.words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (top == round * (unsigned int)groups - 1)
                    for (int k = 0; k < groups; ++k)
                        __hip_atomic_store(g.words + kSyncLine * (16 + k), round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            flag = g.words + kSyncLine * (16 + grp);
            target = round;
        } else {
            __hip_atomic_fetch_add(g.words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned int polls = 0;
        bool dead = false;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++polls & 63u) == 0) {
                if (__hip_atomic_load(g.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { dead = true; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > kPollLimit) {
                    __hip_atomic_store(g.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    dead = true;
                    break;
                }
            }
        }
        lds.flag = dead ? 1u : 0u;
        if (g.acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else if (threadIdx.x == 0) {
        lds.flag = 1u;
    }
    __syncthreads();
    g.dead = lds.flag != 0;
}

// ---- tiles ---------------------------------------------------------------------------------------------------------------
// cross-wave K reduction of `np` accumulators in a fixed pairwise order; wave w finishes registers [4 w, 4 w + 4)
// (wave w deposits its accumulator in image pair `pair` of its OWN staging space: the pair its last round has just been read
// from, so that an LDS-DMA round in flight into the other pair is not disturbed)
__device__ __forceinline__ void reduce_waves(const f32x16& acc, Lds& lds, float (&out)[4], int pair, int wave, int lane) {
    __syncthreads();                       // every wave is done with that pair
#pragma unroll
    for (int r = 0; r < 16; ++r) lds.stage[wave][pair][r * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int at = (wave * 4 + rr) * 64 + lane;
        float lo = lds.stage[0][pair][at] + lds.stage[1][pair][at], hi = lds.stage[2][pair][at] + lds.stage[3][pair][at];
        asm volatile("" : "+v"(lo), "+v"(hi));      // (no packed horizontal add: build.py's hazard guard)
        out[rr] = lo + hi;
    }
}
// the (row, column) inside the tile of accumulator register r = 4 wave + rr of this lane
__device__ __forceinline__ int acc_row(int wave, int rr, int lane) {
    const int r = wave * 4 + rr;
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// lds.tile[p] -> the tile (m0, n0) of D and its transpose as the tile (n0, m0) of DT (every iterate is kept in both
// orientations so that both operands of every product are read row-wise).  Symmetric jobs: DT == D, the transpose is the
// mirror image and a diagonal tile is made symmetric from its upper triangle and written once.  Returns this thread's share
// of the sum of squares of D over the tile (symmetric jobs: over tile + mirror).
__device__ __forceinline__ float write_pair(float* D, float* DT, int n, const float (*tile)[kTilePitch], int m0, int n0, bool sym,
                                            int tid) {
    const bool diag = m0 == n0;
    const int row = tid >> 3, c4 = (tid & 7) * 4;
    f32x4 v, w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c4 + e;
        v[e] = (!(sym && diag) || row <= c) ? tile[row][c] : tile[c][row];
        w[e] = tile[c][row];
    }
    store16(matrix_rsrc(D, n), (m0 + row) * n + n0 + c4, v);
    if (DT && !(sym && diag)) store16(matrix_rsrc(DT, n), (n0 + row) * n + m0 + c4, w);
    float sq = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) sq = fmaf(v[e], v[e], sq);
    return (sym && !diag) ? 2.f * sq : sq;
}

// block-wide sum in a fixed order (wave butterflies, then the four waves in order); every thread gets the result
__device__ __forceinline__ float block_total(float v, Lds& lds) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds.scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float lo = lds.scratch[0] + lds.scratch[1], hi = lds.scratch[2] + lds.scratch[3];
    asm volatile("" : "+v"(lo), "+v"(hi));
    return lo + hi;
}

// sum of `count` coherent floats in index order groups (256 strided partial sums, then the block total): the same value in
// every workgroup
__device__ __forceinline__ float sum_partials(const float* p, int count, bool coherent, Lds& lds) {
    float s = 0.f;
    for (int i = threadIdx.x; i < count; i += 256) s += coherent ? load_coherent(p + i) : p[i];
    return block_total(s, lds);
}

__device__ __forceinline__ void tile_of(int id, int nt, bool sym, int& ti, int& tj) {
    if (!sym && nt == 16) {
        // n = 512, every tile: workgroup b runs on XCD b % 8 (round-robin dispatch); an XCD takes a 4 x 8 block of tiles, so that
        // its L2 serves 4 row panels + 8 column panels (768 KB per product) to 32 workgroups
        const int x = id & 7, j = id >> 3;
        ti = 4 * (x >> 1) + (j >> 3);
        tj = 8 * (x & 1) + (j & 7);
        return;
    }
    if (!sym) { ti = id / nt; tj = id % nt; return; }
    int row = 0, rem = id, len = nt;
    while (rem >= len) { rem -= len; ++row; --len; }
    ti = row;
    tj = row + rem;
}

enum StepEpilogue { EP_T, EP_E, EP_SCALE };

// StyleLossW2.forward's scalars (style_transfer.py:178-181) by one workgroup of the chain kernel: w2_loss_block with the
// root read coherently (other workgroups - other XCDs - have just written it)
__device__ __forceinline__ void w2_loss_chain(const W2LossJob& j, Lds& lds) {
#pragma clang fp contract(off)
    float sm = 0.f, sc = 0.f;
    for (int i = threadIdx.x; i < j.n; i += 256) {
        const float d = j.mean[i] - j.mean_t[i];
        sm += d * d;
        const size_t ii = (size_t)i * j.n + i;
        sc += (j.cov_t[ii] + j.cov[ii]) - 2.f * load_coherent(j.root + ii);
    }
    sm = block_total(sm, lds);
    sc = block_total(sc, lds);
    if (threadIdx.x == 0) {
        const float fn = (float)j.n;
        j.loss_out[0] = (sm / fn + sc / fn) * j.weight;
        j.gdiag_out[0] = w2_gdiag(j);
    }
}

struct Product {               // D = c * epilogue(A x B): A row-major, BT = B^T row-major; results in both orientations
    const float* a;
    const float* bt;
    float* d;
    float* dt;
    float c;
};

template <int N, bool CACHED>
__device__ __forceinline__ void chain_body(const NsChainJob& job, int wg, Lds& lds) {
    constexpr int nt = N / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool sym = job.symmetric != 0;
    int ti, tj;
    tile_of(wg, nt, sym, ti, tj);
    const int m0 = ti * 32, n0 = tj * 32;
    const int row = tid >> 3, c4 = (tid & 7) * 4;          // this thread's 4 elements of a tile in the elementwise steps
    float* my = &lds.stage[wave][0][0];
    // Operands through the L2 need EITHER an agent-scope acquire per barrier (buffer_inv sc1: measured slower than the
    // memory-side loads it replaces) OR addresses nobody has read since the launch began: with job.arena every iterate of
    // every step gets a matrix of its own (127 of them for both recurrences), so no L1 / L2 of the chip can hold a line of it
    // from before its write-through stores - the first reader of an XCD fetches it from the memory side, the other 31
    // workgroups of the XCD's 4 x 8 tile block hit that L2 (6 MB instead of 32 MB over the fabric per n = 512 product).
    // What was cached of the arena by the PREVIOUS launch is dropped at the kernel boundary like any other buffer's lines.
    const bool fresh = CACHED && job.arena != nullptr;
    Grid grid{job.sync, job.error, job.tiles, wg, 0u, false, CACHED && !fresh};
    int arena_next = 0;
    auto take = [&](float*& x) { if (fresh) x = job.arena + (size_t)(arena_next++) * N * N; };
    auto take2 = [&](float*& x, float*& xt) { take(x); if (sym) xt = x; else take(xt); };
    // (the OTHER half of the barrier words is this job's next launch's: cleared here, visible at the kernel boundary)
    if (wg == 0)
        for (int i = tid; i < kSyncUints; i += 256) job.sync_next[i] = 0u;

    // one step: one or two products (p1.a == nullptr: one), epilogue, tiles out.  ep applies to p0; a second product is
    // always EP_SCALE.  q1_out: EP_E only - also write (q1_c * E) / 2, the first Lyapunov step's q (see below).
    auto step = [&](const Product& p0, const Product& p1, StepEpilogue ep, float* q1_out, float* q1t_out, float q1_c,
                    float* sumsq_out) __attribute__((always_inline)) {
        const bool two = p1.a != nullptr;
        const int l31 = lane & 31;
        // the epilogue of product p: this wave's four accumulator registers of the reduced tile -> lds.tile
        auto finish = [&](int p, const Product& pr, const float (&out)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int trow = acc_row(wave, rr, lane);
                const bool on_diag = (m0 + trow) == (n0 + l31);
                float v;
                if (p == 0 && ep == EP_T) v = ((on_diag ? 3.f : 0.f) - out[rr]) * 0.5f;              // t = (3I - z y) / 2  (:22)
                else if (p == 0 && ep == EP_E) v = ((on_diag ? 3.f : 0.f) - out[rr]) * 1.f;          // eye_a_a = 3I - a a (:43)
                else v = out[rr] * pr.c;
                lds.tile[p][trow][l31] = v;
                if (p == 0 && q1_out) lds.tile[1][trow][l31] = (q1_c * v) * 0.5f;                  // q_1 = q_0 E / 2, q_0 = q1_c I (:44)
            }
        };
        if constexpr (CACHED && Geo<N>::RK == 32) {
            // Operands by LDS-DMA, no operand registers (the register-staged form below keeps two panels = 128 registers live and
            // leaves a SIMD room for ONE more wave: the other heads' launches then queue behind each other for that slot for as
            // long as this kernel is resident - profiles/r05_ns_chain.md section 5).  The rounds of the step's one or two products
            // form one sequence, two rounds in flight.
            using G = Geo<N>;
            const int rounds = (two ? 2 : 1) * G::NR;
            auto issue = [&](int q) __attribute__((always_inline)) {
                const Product& pr = q < G::NR ? p0 : p1;
                const int r = q < G::NR ? q : q - G::NR;
                dma_round<N>(pr.a, pr.bt, m0, n0, wave * G::KW + r * 32, &lds.stage[wave][q & 1][0], lane);
            };
            issue(0);
            if (rounds > 1) issue(1);
            f32x16 acc;
#pragma unroll 1
            for (int q = 0; q < rounds; ++q) {
                const int r = q < G::NR ? q : q - G::NR;
                if (r == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                }
                if (q + 1 < rounds) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                dma_mfma(acc, &lds.stage[wave][q & 1][0], lane);
                asm volatile("" ::: "memory");
                if (r == G::NR - 1) {
                    float out[4];
                    reduce_waves(acc, lds, out, q & 1, wave, lane);
                    const int p = q < G::NR ? 0 : 1;
                    finish(p, p == 0 ? p0 : p1, out);
                    __syncthreads();           // the tile is complete - and every wave has read the reduction buffer (pair q & 1)
                }
                if (q + 2 < rounds) issue(q + 2);
            }
        } else {
            // (one product at a time - two operand panels = 128 registers and one accumulator live; a second product's loads
            // are issued when the first one's registers are free)
            Panel<N> pa, pb;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (p == 1 && !two) break;
                const Product& pr = p == 0 ? p0 : p1;
                if (p == 0 || p1.bt != p0.bt) panel_load<N, CACHED>(pb, pr.bt, n0, wave, lane);
                panel_load<N, CACHED>(pa, pr.a, m0, wave, lane);
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                panel_mfma<N>(acc, pa, pb, my, lane);
                float out[4];
                reduce_waves(acc, lds, out, 0, wave, lane);
                finish(p, pr, out);
                __syncthreads();               // the tile is complete - and the reduction buffer (= staging space) free again
            }
        }
        float sq = write_pair(p0.d, p0.dt, N, lds.tile[0], m0, n0, sym, tid);
        if (two) write_pair(p1.d, p1.dt, N, lds.tile[1], m0, n0, sym, tid);
        else if (q1_out) write_pair(q1_out, q1t_out, N, lds.tile[1], m0, n0, sym, tid);
        if (sumsq_out) {
            sq = block_total(sq, lds);
            if (tid == 0) store_coherent(sumsq_out + wg, sq);
        }
    };
    const Product none{nullptr, nullptr, nullptr, nullptr, 0.f};
    auto fill_nan = [&](float* d) {
        if (!d) return;
        const float nan = __builtin_nanf("");
#pragma unroll
        for (int e = 0; e < 4; ++e) lds.tile[0][row][c4 + e] = nan;
        __syncthreads();
        write_pair(d, nullptr, N, lds.tile[0], m0, n0, false, tid);
        __syncthreads();
    };
    // this workgroup's tile of a row-major matrix into lds.tile[0] (symmetric jobs: the diagonal tile from its upper triangle)
    auto own_tile = [&](const float* src) {
        const f32x4 v = load16(matrix_rsrc(src, N), (m0 + row) * N + n0 + c4);
        lds.tile[0][row][c4 + 0] = v[0]; lds.tile[0][row][c4 + 1] = v[1];
        lds.tile[0][row][c4 + 2] = v[2]; lds.tile[0][row][c4 + 3] = v[3];
        __syncthreads();
        if (sym && ti == tj) {
            float u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = row <= c4 + e ? lds.tile[0][row][c4 + e] : lds.tile[0][c4 + e][row];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) lds.tile[0][row][c4 + e] = u[e];
            __syncthreads();
        }
    };
    auto own_sumsq = [&]() {
        float sq = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) sq = fmaf(lds.tile[0][row][c4 + e], lds.tile[0][row][c4 + e], sq);
        return block_total((sym && ti != tj) ? 2.f * sq : sq, lds);
    };

    // every iterate in both orientations (X, X^T); symmetric jobs: the same buffer
    float *y = job.y0, *yn = job.y1, *z = job.z1, *zn = job.z0;               // (z_1 = t_0 lands in z1, see below)
    float *yt = sym ? job.y0 : job.yt0, *ytn = sym ? job.y1 : job.yt1;
    float *zt = sym ? job.z1 : job.zt1, *ztn = sym ? job.z0 : job.zt0;
    auto swap2 = [](float*& a, float*& b) { float* t_ = a; a = b; b = t_; };
    float* T = job.t;
    float* TT = sym ? job.t : job.tt;
    // a matrix that is only ever a LEFT operand (q, the results) needs no transpose - but a symmetric job computes only the
    // tile pairs ti <= tj of it, and the other half is the mirror image in the same buffer
    auto mirror = [&](float* x) { return sym ? x : static_cast<float*>(nullptr); };
    float* partials = job.scalars + 8;
    float norm_m = 1.f;

    if (job.forward) {
        // norm_a = a.pow(2).sum().sqrt(); y = a / norm_a; z = I                                                  (sqrtm.py:16-20)
        own_tile(job.m);
        float total;
        if (job.m_nparts > 0) {
            total = sum_partials(job.m_partials, job.m_nparts, false, lds);
        } else {
            const float sq = own_sumsq();
            if (tid == 0) store_coherent(partials + wg, sq);
            grid_sync(grid, lds);
            total = sum_partials(partials, job.tiles, true, lds);
        }
        norm_m = sqrtf(total);
        if (wg == 0 && tid == 0) job.scalars[0] = norm_m;
        // first step with z = I (see ns_sqrt_forward in st_smallgemm.hip): t_0 = (3I - y_0) / 2 is z_1 as it stands
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c4 + e;
            const float y = lds.tile[0][row][c] / norm_m;
            const bool on_diag = (m0 + row) == (n0 + c);
            lds.tile[1][row][c] = ((on_diag ? 3.f : 0.f) - y) * 0.5f;
            lds.tile[0][row][c] = y;
        }
        __syncthreads();
        take2(y, yt); take2(z, zt);
        write_pair(y, yt, N, lds.tile[0], m0, n0, sym, tid);
        write_pair(z, zt, N, lds.tile[1], m0, n0, sym, tid);
        grid_sync(grid, lds);
        take2(yn, ytn);
        step(Product{y, zt, yn, ytn, 1.f}, none, EP_SCALE, nullptr, nullptr, 0.f, nullptr);             // y_1 = y_0 t_0     (:23)
        swap2(y, yn); swap2(yt, ytn);
        grid_sync(grid, lds);
        for (int it = 1; it < 12; ++it) {
            take2(T, TT);
            step(Product{z, yt, T, TT, 0.f}, none, EP_T, nullptr, nullptr, 0.f, nullptr);               // t = (3I - z y) / 2 (:22)
            grid_sync(grid, lds);
            if (it < 11) {
                take2(yn, ytn); take2(zn, ztn);
                step(Product{y, TT, yn, ytn, 1.f},                                                       // y = y t           (:23)
                     Product{T, zt, zn, ztn, 1.f}, EP_SCALE, nullptr, nullptr, 0.f, nullptr);            // z = t z           (:24)
                swap2(y, yn); swap2(yt, ytn); swap2(z, zn); swap2(zt, ztn);
            } else {                                                                                     // y * sqrt(norm_a)   (:25)
                step(Product{y, TT, job.root, mirror(job.root), sqrtf(norm_m)}, none, EP_SCALE, nullptr, nullptr, 0.f, partials);
            }
            grid_sync(grid, lds);
        }
    } else if (job.backward) {
        // the root is an input: its tile, and the tiles' sums of squares for ||root||_F
        own_tile(job.root);
        const float sq = own_sumsq();
        if (tid == 0) store_coherent(partials + wg, sq);
        grid_sync(grid, lds);
    }

    if (job.backward) {
        // (lds.tile[0] still holds this workgroup's tile of the root)
        // norm_z = ||z||_F; a = z / norm_z; q = grad / norm_z                                                  (sqrtm.py:38-41)
        const float norm_r = sqrtf(sum_partials(partials, job.tiles, true, lds));
        if (wg == 0 && tid == 0) job.scalars[1] = norm_r;
        const float gd = job.loss.loss_out ? w2_gdiag(job.loss) : (job.gdiag_dev ? job.gdiag_dev[0] : job.gdiag);
        const float q0 = gd / norm_r;
        if (wg == 0 && job.loss.loss_out) w2_loss_chain(job.loss, lds);
#pragma unroll
        for (int e = 0; e < 4; ++e) lds.tile[1][row][c4 + e] = lds.tile[0][row][c4 + e] / norm_r;
        __syncthreads();
        // a lives in the forward's y slots (a^T in their transposes), q in z's, E / E^T in t's
        float *a = job.y0, *an = job.y1, *at = sym ? job.y0 : job.yt0, *atn = sym ? job.y1 : job.yt1;
        float *q = job.z0, *qn = job.z1;
        take2(a, at); take(q);
        write_pair(a, at, N, lds.tile[1], m0, n0, sym, tid);
        grid_sync(grid, lds);
        for (int it = 0; it < 12; ++it) {
            take2(T, TT);
            if (it < 11) take2(an, atn);
            if (it > 0 && it < 11) take(qn);
            // eye_a_a = 3I - a a (:43): E in t's slot, E^T in its transpose (what the products below read); in the first step
            // q_0 = (gd / norm_z) I, so q_1 = q_0 E / 2 is elementwise
            step(Product{a, at, T, TT, 0.f}, none, EP_E, it == 0 ? q : nullptr, it == 0 ? mirror(q) : nullptr, q0, nullptr);
            grid_sync(grid, lds);
            if (it == 0) {
                step(Product{a, TT, an, atn, 0.5f}, none, EP_SCALE, nullptr, nullptr, 0.f, nullptr);    // a = a E / 2        (:46)
                swap2(a, an); swap2(at, atn);
            } else if (it < 11) {
                // q = q E / 2 (:44 without the commutator, which vanishes for a seed that is a multiple of I), a = a E / 2
                step(Product{q, TT, qn, mirror(qn), 0.5f}, Product{a, TT, an, atn, 0.5f}, EP_SCALE, nullptr, nullptr, 0.f, nullptr);
                swap2(q, qn); swap2(a, an); swap2(at, atn);
            } else {                                                                     // ... and the final / 2 (:47)
                step(Product{q, TT, job.grad_m, mirror(job.grad_m), 0.25f}, none, EP_SCALE, nullptr, nullptr, 0.f, nullptr);
            }
            if (it < 11) grid_sync(grid, lds);
        }
    }
    if (grid.dead || __hip_atomic_load(job.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __syncthreads();
        if (job.forward) fill_nan(job.root);
        if (job.backward) fill_nan(job.grad_m);
        if (wg == 0 && tid == 0 && job.loss.loss_out) job.loss.loss_out[0] = __builtin_nanf("");
    }
}

// One wave per SIMD (~290 registers): ONE workgroup of a chain kernel per CU.  The launcher's callers therefore keep the
// workgroups of all chain kernels that can be in flight at once <= the CU count (st_api.hip: ST_NS_CHAIN's head mask) - two
// persistent kernels that each hold some CUs and wait for the rest would wait for each other.
template <bool CACHED>
__device__ __forceinline__ void chain_kernel_body(const NsChainLaunch& launch, Lds& lds) {
    int j = 0;
    while (j + 1 < launch.count && (int)blockIdx.x >= launch.job[j + 1].tile0) ++j;
    const NsChainJob& job = launch.job[j];
    const int wg = (int)blockIdx.x - job.tile0;
    switch (job.n) {
        case 64: chain_body<64, CACHED>(job, wg, lds); break;
        case 128: chain_body<128, CACHED>(job, wg, lds); break;
        case 256: chain_body<256, CACHED>(job, wg, lds); break;
        default: chain_body<512, CACHED>(job, wg, lds); break;
    }
}
// operands from the memory side (sc1 loads into registers): ~400 registers, one workgroup per CU and little room beside it
__global__ __launch_bounds__(256) void ns_chain_kernel_mem(NsChainLaunch launch) {
    __shared__ __attribute__((aligned(16))) Lds lds;
    chain_kernel_body<false>(launch, lds);
}
// operands through the L2 by LDS-DMA: capped at 128 + 32 registers, so that a SIMD keeps room for two more waves of the
// other heads' kernels (<= 208 and <= 104 registers) while this kernel is resident
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(128))) void ns_chain_kernel_l2(NsChainLaunch launch) {
    __shared__ __attribute__((aligned(16))) Lds lds;
    chain_kernel_body<true>(launch, lds);
}

}  // namespace

int ns_chain_mask() {
    // bit 0: the three shallow heads (one launch), bit 1: relu4_1, bit 2: relu5_1, bit 3: the standalone operators
    static Option on("ST_NS_CHAIN", 0);
    return on.get();
}
bool ns_chain_enabled() {
    static Option on("ST_NS_CHAIN", 0);           // 0 (default): one launch per product (rounds 1 - 4); see profiles/r05_ns_chain.md
    return on.get() != 0;
}

int ns_chain_sync_uints() { return kSyncUints; }

int ns_chain_tiles(int n, bool symmetric) {
    const int nt = n / 32;
    return symmetric ? nt * (nt + 1) / 2 : nt * nt;
}

// (job.sync / sync_next / error come from the workspace: ns_chain_job in st_smallgemm.hip); this fixes the grid
int launch_ns_chain(NsChainLaunch& launch, hipStream_t s) {
    ST_REQUIRE(launch.count >= 1 && launch.count <= 3, "ns chain: 1 to 3 jobs per launch");
    int total = 0;
    for (int i = 0; i < launch.count; ++i) {
        NsChainJob& j = launch.job[i];
        ST_REQUIRE(j.n == 64 || j.n == 128 || j.n == 256 || j.n == 512, "ns chain: n must be 64, 128, 256 or 512 (got %d)", j.n);
        ST_REQUIRE(j.forward || j.backward, "ns chain: nothing to do");
        ST_REQUIRE(j.sync && j.sync_next && j.error && j.scalars, "ns chain: workspace without barrier words");
        j.tile0 = total;
        ST_REQUIRE(j.symmetric || (j.yt0 && j.yt1 && j.zt0 && j.zt1 && j.tt), "ns chain: a full job needs the transposed slots");
        j.tiles = ns_chain_tiles(j.n, j.symmetric != 0);
        total += j.tiles;
    }
    // At most ONE chain kernel in flight per device: a persistent kernel whose workgroups wait for each other needs all of
    // them resident (one per CU), and two such kernels on two streams - two plans of one process, a head of each - could each
    // hold some CUs and wait for the rest.  Every launch waits for the previous launch's completion event.
    static std::mutex guard;
    static hipEvent_t last[16] = {};
    int dev = 0;
    ST_HIP(hipGetDevice(&dev));
    ST_REQUIRE(dev >= 0 && dev < 16, "ns chain: device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(guard);
    if (!last[dev]) ST_HIP(hipEventCreateWithFlags(&last[dev], hipEventDisableTiming));
    else ST_HIP(hipStreamWaitEvent(s, last[dev], 0));
    // (operands through the L2 or from the memory side: one kernel each - both forms inlined into one kernel made it spill)
    for (int i = 0; i < launch.count; ++i)
        ST_REQUIRE(!launch.job[i].arena || launch.job[i].l2_loads, "ns chain: the arena is for operands through the L2");
    if (launch.job[0].l2_loads) hipLaunchKernelGGL(ns_chain_kernel_l2, dim3(total), dim3(256), 0, s, launch);
    else hipLaunchKernelGGL(ns_chain_kernel_mem, dim3(total), dim3(256), 0, s, launch);
    ST_LAUNCH_CHECK();
    ST_HIP(hipEventRecord(last[dev], s));
    return 0;
}

}  // namespace st
