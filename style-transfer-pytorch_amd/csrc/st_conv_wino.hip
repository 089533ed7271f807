// The 3 x 3 convolutions of the trunk (nn.Conv2d 3x3, padding 1, + bias + ReLU: reference style_transfer.py:35,87; their data
// gradients: the same operator on rotated weights with the roles of the channels swapped) as Winograd F(2 x 2, 3 x 3) on the
// fp16x3 planes.
//
//   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A          per 2 x 2 output tile, 4 x 4 input patch d, 3 x 3 filter g
//
// 16 multiplications per 4 outputs instead of 36: 2.25 x fewer MFMAs than the direct form (st_conv_pc.hip), which is bound by
// the matrix pipe at the clock the chip holds under it.  The filters are transformed once, in double, and split into two fp16
// planes under one power-of-two scale (wino_weights_kernel); the input patches are transformed in fp32 (additions only, the
// power-of-two operand scale folded into the first pass) BEFORE the split; each of the 16 transform positions is a plane GEMM
// h0 g0 + h0 g1 + h1 g0 with fp32 accumulation; the output transform runs on the accumulators in registers.  Accuracy: 2 - 4e-7
// per convolution against float64, the class of the direct fp16x3 form (profiles/r05_winograd.md).
//
// Round 5's prototype (single role, raw patch staged through LDS, phases serialised: 0.5 - 0.75 x the direct kernel) is replaced
// by this form (round 6, profiles/r06_winograd.md):
//   * one workgroup of four waves (one per SIMD, 512 registers each) per 64 co x 64 tiles (256 px); a wave multiplies
//     32 co x 32 tiles x 16 positions (16 accumulators = 256 AGPRs) AND transforms (tile, 4 channels) items - the transform's
//     VALU work issues between the wave's own MFMAs (a partner wave's VALU is starved by an MFMA stream, the wave's own is not:
//     profiles/r02_mfma_sustained.md, MI355X_MICROARCH.md "fillers per MFMA gap");
//   * a K chunk of 16 channels runs as TWO stages of 8 positions (V rows 0-1, then rows 2-3): stage buffers of 32 KB for the
//     transformed weights (LDS-DMA, 1 KB pieces in MFMA A-operand order) and 32 KB for the transformed patches, both double
//     buffered: 128 KB of the CU's 160;
//   * every lane loads its own 4 x 4 patches with 8-byte row loads straight from L1 / L2 (no raw staging pass), one chunk of
//     loads in flight across two stages (two register sets); out-of-image rows come back as zeros from the buffer bounds check,
//     out-of-image columns are multiplied away inside the transform's own FMAs;
//   * one s_barrier per stage, counted vmcnt so that the patch loads stay in flight across it.
#include "st_common.h"

namespace st {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int kOOR = 0x40000000;                    // buffer offset beyond every resource: loads return 0
constexpr int kStage = 32768;                       // one stage buffer (8 positions) of either operand
constexpr int kLdsA = 0, kLdsB = 2 * kStage;
constexpr int kLdsBytes = 4 * kStage;               // 128 KB

template <int B, int E, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        sfor<B + 1, E>(f);
    }
}

// torch [Cout][Cin][3][3] -> U = G g G^T (double), two fp16 planes under 2^e, e from the bound 2.25 max |w|;
// layout [chunk = K / 16][position 16][co block = M / 32][plane 2][lane 64][8 halves]: a 1 KB block IS the A operand of
// v_mfma_f32_32x32x16_f16 (lane l: row l & 31, k = 8 (l >> 5) + 0..7).  dgrad: the data gradient's operator - M = torch Cin,
// K = torch Cout, taps rotated by 180 degrees.
__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int kdim, int mdim,
                                                          int dgrad, const unsigned int* __restrict__ w_amax, int* __restrict__ exp_out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= kdim * mdim) return;
    const int co = idx / kdim, ci = idx % kdim;      // GEMM row (output channel of this operator), reduction index
    const float bound = 2.25f * __builtin_bit_cast(float, w_amax[0]);
    const int e = scale_exp(__builtin_bit_cast(unsigned int, bound));
    if (idx == 0) exp_out[0] = e;
    double g[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            g[i][j] = dgrad ? (double)w[((size_t)ci * mdim + co) * 9 + (2 - i) * 3 + (2 - j)] : (double)w[((size_t)co * kdim + ci) * 9 + i * 3 + j];
    // rows of G: (1, 0, 0), (1/2, 1/2, 1/2), (1/2, -1/2, 1/2), (0, 0, 1)
    double t[4][3];
    for (int j = 0; j < 3; ++j) {
        t[0][j] = g[0][j];
        t[1][j] = 0.5 * (g[0][j] + g[1][j] + g[2][j]);
        t[2][j] = 0.5 * (g[0][j] - g[1][j] + g[2][j]);
        t[3][j] = g[2][j];
    }
    const float sc = pow2f(e);
    const int CB = mdim / 32, c = ci >> 4, k = ci & 15, cb = co >> 5, lane = (co & 31) + 32 * (k >> 3), j8 = k & 7;
    for (int a = 0; a < 4; ++a) {
        const double u[4] = {t[a][0], 0.5 * (t[a][0] + t[a][1] + t[a][2]), 0.5 * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
        for (int b = 0; b < 4; ++b) {
            const float x = (float)u[b] * sc;
            const _Float16 h0 = (_Float16)x;
            const _Float16 h1 = (_Float16)(x - (float)h0);
            const size_t blk = ((size_t)(c * 16 + a * 4 + b) * CB + cb) * 2;
            out[((blk + 0) * 64 + lane) * 8 + j8] = h0;
            out[((blk + 1) * 64 + lane) * 8 + j8] = h1;
        }
    }
}

__device__ __forceinline__ f32x2 bload2(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}
__device__ __forceinline__ void bstore2(f32x2 v, __amdgpu_buffer_rsrc_t rs, int voff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), rs, voff, 0, 0);
}

// s_waitcnt immediates (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]); through the builtin, so that the
// compiler's own wait-count bookkeeping sees them
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }

// TX = tiles per row of the workgroup's pixel tile: 8 (16 x 16 px), 16 (32 x 8 px) or 32 (64 x 4 px)
// RMW: the epilogue reads the tensor it writes (accumulate) and / or a ReLU mask (out_mask) - data gradients
// TUNE: ablation bits (ST_WINO_TUNE, timing only, wrong results): 1 no transform pieces, 2 no MFMA / operand fetch, 4 no weight
// DMA, 8 no patch loads, 16 no barriers
template <int TX, bool RMW, int TUNE = 0>
__global__ __launch_bounds__(256) void wino_conv_kernel(ConvProblem p, int tiles_x, int n_co_tiles, int ksplit, int nchunks, int total) {
    constexpr int TY = 64 / TX;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.height, W = p.width, HW = H * W;

    // XCD-aware order (as conv_pc_kernel): XCD x = blockIdx & 7 takes a contiguous range of logical ids, in which consecutive
    // ids are the Cout tiles (and K slices) of ONE pixel tile: they share its patches through that XCD's L2
    int bid = blockIdx.x;
    if ((total & 7) == 0) bid = (bid & 7) * (total >> 3) + (bid >> 3);
    const int ct = bid % n_co_tiles;
    bid /= n_co_tiles;
    const int ks = bid % ksplit;
    bid /= ksplit;
    const int x0 = (bid % tiles_x) * (2 * TX), y0 = (bid / tiles_x) * (2 * TY);
    const int chunk0 = ks * nchunks;

    const _Float16* wgt = static_cast<const _Float16*>(p.wgt_wino);
    const int CB = p.cout >> 5;
    const int ea = *reinterpret_cast<const int*>(reinterpret_cast<const unsigned char*>(wgt) + (size_t)16 * p.cin * p.cout * 4 + 32);
    // V = B^T d B grows an entry by at most 4 x max |d|
    const float vbound = 4.f * __builtin_bit_cast(float, amax_read(p.amax_word));
    const int eb = scale_exp(__builtin_bit_cast(unsigned int, vbound));
    const float sc = pow2f(eb);
    const f32x2 sc2 = {sc, sc}, nsc2 = {-sc, -sc};

    // ---- transform role: lane <-> (tile, channel quad); lanes 2k, 2k + 1 are the two quads of one kgroup of tile k: their
    // 8-byte writes into a B block (lane (tile & 31) + 32 kgroup, 16 bytes each) are contiguous
    const int ttile = (wave >> 1) * 32 + (lane >> 1);
    const int cq = (wave & 1) * 2 + (lane & 1);
    const int toy = y0 + 2 * (ttile / TX), tox = x0 + 2 * (ttile % TX);
    const bool tile_in = tox < W && toy < H;
    const bool left = tox == 0;
    // the patch's columns are tox - 1 .. tox + 2 as two 8-byte loads (L, R).  At the image's left edge L would start at x = -1,
    // for row 0 of channel 0 in front of the tensor: L is taken one column further right instead (x = 0, 1), its first element
    // is moved into the place of the second one and the first one multiplied away (m0); x = W at the right edge reads the next
    // row's first element (or zero past the end of the resource) and is multiplied away as well (nm3)
    int voffL[4], voffR[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = toy - 1 + r;
        const bool ok = tile_in && y >= 0 && y < H;
        const int base = ((cq * 4 * H + y) * W + tox - 1) * 4;
        voffL[r] = ok ? base + (left ? 4 : 0) : kOOR;
        voffR[r] = ok ? base + 8 : kOOR;
    }
    const float m0 = left ? 0.f : 1.f, nm3 = (tox + 2 < W) ? -1.f : 0.f;
    const int bw_off = lane * 8 + (wave & 1) * 512 + (wave >> 1) * 1024;       // inside a (position, plane) pair of 1 KB blocks: [tb][1 KB]

    // ---- MFMA role: 32 co x 32 tiles x 16 positions
    const int cb = wave & 1, tb = wave >> 1;
    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;

    f32x2 raw[4][4][2];                                 // [channel of the quad][patch row][L / R]: the chunk to be transformed
    f32x2 tv[4][4][2];                                  // B^T d (scaled): [channel][row][L / R]; rows 2, 3 live across a stage
    float vrow[4][4];                                   // one row of V: [column][channel]
    const int hw4 = HW * 4;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16*>(wgt), 0, (int)((size_t)16 * p.cin * p.cout * 4), 0x00020000);
    const int lane16 = lane * 16;

    // patch rows [R0, R1) of channel J of the quad
    auto load_raw = [&](const __amdgpu_buffer_rsrc_t rs, auto J, auto R0, auto R1) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value;
        if constexpr (TUNE & 8) return;
        sfor<decltype(R0)::value, decltype(R1)::value>([&](auto R) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            if constexpr (TUNE & 32) {                  // (experiment: half the loads)
                raw[j][r][0] = bload2(rs, voffL[r], j * hw4);
                raw[j][r][1] = raw[j][r][0];
            } else if constexpr (TUNE & 64) {           // (experiment: one 16-byte load per patch row)
                const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voffL[r], j * hw4, 0));
                raw[j][r][0] = f32x2{q[0], q[1]};
                raw[j][r][1] = f32x2{q[2], q[3]};
            } else {
                raw[j][r][0] = bload2(rs, voffL[r], j * hw4);
                raw[j][r][1] = bload2(rs, voffR[r], j * hw4);
            }
        });
    };
    // transformed weights of stage (chunk, half): 32 pieces of 1 KB, wave w moves pieces 8 w .. 8 w + 7 (piece q = (position,
    // co block, plane); the source of a position is 4 KB apart from the next one's only when CB == 2)
    auto dma_weights = [&](int chunk, int half, unsigned char* dst) __attribute__((always_inline)) {
        if constexpr (TUNE & 4) return;
        const int base = (((chunk0 + chunk) * 16 + half * 8) * CB + 2 * ct) * 2048;
        sfor<0, 8>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int q = wave * 8 + i, pos = q >> 2, rest = q & 3;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, lane16,
                                                     base + (pos * CB * 2 + rest) * 1024, 0, 0);
        });
    };
    // split a position's four channel values into the two planes (h0 = fp16(v), h1 = fp16(v - h0): the subtraction is exact,
    // v_fma_mix rounds once) and write them in B-operand order.  Six VALU instructions for four values; as inline assembly
    // because the compiler's own lowering of the same expressions takes 13.
    auto write_pos = [&](unsigned char* dst, int pos, const float (&v)[4]) __attribute__((always_inline)) {
        u32x2 h0, h1;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h0[0]) : "v"(v[0]), "v"(v[1]));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h0[1]) : "v"(v[2]), "v"(v[3]));
        asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(h1[0]) : "v"(h0[0]), "v"(v[0]));
        asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(h1[0]) : "v"(h0[0]), "v"(v[1]));
        asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(h1[1]) : "v"(h0[1]), "v"(v[2]));
        asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(h1[1]) : "v"(h0[1]), "v"(v[3]));
        *reinterpret_cast<u32x2*>(dst + (pos * 2 + 0) * 2048 + bw_off) = h0;
        *reinterpret_cast<u32x2*>(dst + (pos * 2 + 1) * 2048 + bw_off) = h1;
    };
    // B^T d of one column pair of one channel, scaled: rows 0 .. 3 of tv
    auto col_pass = [&](auto J, auto SIDE) __attribute__((always_inline)) {
        constexpr int j = decltype(J)::value, sd = decltype(SIDE)::value;
        f32x2 d0 = raw[j][0][sd], d1 = raw[j][1][sd], d2 = raw[j][2][sd], d3 = raw[j][3][sd];
        if constexpr (sd == 0) {                        // image's left edge: see voffL
            d0[1] = left ? d0[0] : d0[1];
            d1[1] = left ? d1[0] : d1[1];
            d2[1] = left ? d2[0] : d2[1];
            d3[1] = left ? d3[0] : d3[1];
        }
        const f32x2 s1 = d1 * sc2, s2 = d2 * sc2;
        tv[j][0][sd] = __builtin_elementwise_fma(d0, sc2, -s2);
        tv[j][1][sd] = s1 + s2;
        tv[j][2][sd] = s2 - s1;
        tv[j][3][sd] = __builtin_elementwise_fma(d3, nsc2, s1);
    };
    // (B^T d) B for row R of channels J0, J0 + 1 -> vrow; edge columns multiplied away
    auto row_pass = [&](auto R, auto J0) __attribute__((always_inline)) {
        constexpr int r = decltype(R)::value;
        sfor<decltype(J0)::value, decltype(J0)::value + 2>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = decltype(J)::value;
            const f32x2 tl = tv[j][r][0], tr = tv[j][r][1];
            vrow[0][j] = __builtin_fmaf(tl[0], m0, -tr[0]);
            vrow[1][j] = tl[1] + tr[0];
            vrow[2][j] = tr[0] - tl[1];
            vrow[3][j] = __builtin_fmaf(tr[1], nm3, tl[1]);
        });
    };
    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 1> I1{};
    constexpr std::integral_constant<int, 2> I2{};
    constexpr std::integral_constant<int, 3> I3{};
    constexpr std::integral_constant<int, 4> I4{};
    // The work beside the MFMAs of a stage, cut into pieces of 6 - 10 VALU instructions; piece K runs in MFMA slot K.
    // FIRST half of a chunk's transform (runs in stage 1 of the chunk before): B^T d of the raw patches, V rows 0 and 1 written,
    // rows 2 and 3 of B^T d kept;  SECOND half (stage 0 of the chunk itself): V rows 2 and 3 written.
    auto piece_first = [&](auto K, unsigned char* dst) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        if constexpr (TUNE & 1) return;
        if constexpr (k < 8) {
            col_pass(std::integral_constant<int, k / 2>{}, std::integral_constant<int, k % 2>{});
        } else if constexpr (k == 8) {
            row_pass(I0, I0);
        } else if constexpr (k == 9) {
            row_pass(I0, I2);
        } else if constexpr (k < 14) {
            write_pos(dst, k - 10, vrow[k - 10]);
        } else if constexpr (k == 14) {
            row_pass(I1, I0);
        } else if constexpr (k == 15) {
            row_pass(I1, I2);
        } else if constexpr (k < 20) {
            write_pos(dst, 4 + k - 16, vrow[k - 16]);
        }
    };
    auto piece_second = [&](auto K, unsigned char* dst) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value;
        if constexpr (TUNE & 1) return;
        if constexpr (k == 0) {
            row_pass(I2, I0);
        } else if constexpr (k == 1) {
            row_pass(I2, I2);
        } else if constexpr (k < 6) {
            write_pos(dst, k - 2, vrow[k - 2]);
        } else if constexpr (k == 6) {
            row_pass(I3, I0);
        } else if constexpr (k == 7) {
            row_pass(I3, I2);
        } else if constexpr (k < 12) {
            write_pos(dst, 4 + k - 8, vrow[k - 8]);
        }
    };

    // A stage: 8 positions x (h0 g1 + h1 g0 + h0 g0) = 24 MFMA slots, two positions at a time (consecutive MFMAs go to
    // different accumulators); the operands of the next position pair are fetched and one piece of the transform / of the
    // patch loads runs beside every MFMA.  One scheduling region per slot: the interleave is the source order.
    f16x8 opa[2][2][2], opb[2][2][2];                   // [buffer][position of the pair][plane]
    auto stage = [&](auto HALF, const unsigned char* A, const unsigned char* B, unsigned char* bdst, int load_chunk)
                     __attribute__((always_inline)) {
        constexpr int half = decltype(HALF)::value;
        const unsigned char* a_rd = A + cb * 2048 + lane16;
        const unsigned char* b_rd = B + tb * 1024 + lane16;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.in) + (size_t)(chunk0 + load_chunk) * 16 * HW, 0, 16 * hw4, 0x00020000);
        auto fetch = [&](auto BUF, auto E, int pos, auto WHICH) __attribute__((always_inline)) {
            constexpr int bf = decltype(BUF)::value, e = decltype(E)::value;
            if constexpr (TUNE & 2) return;
            if constexpr (decltype(WHICH)::value == 0) {
                opa[bf][e][0] = *reinterpret_cast<const f16x8*>(a_rd + pos * 4096);
                opa[bf][e][1] = *reinterpret_cast<const f16x8*>(a_rd + pos * 4096 + 1024);
            } else {
                opb[bf][e][0] = *reinterpret_cast<const f16x8*>(b_rd + pos * 4096);
                opb[bf][e][1] = *reinterpret_cast<const f16x8*>(b_rd + pos * 4096 + 2048);
            }
        };
        fetch(I0, I0, 0, I0);
        fetch(I0, I0, 0, I1);
        fetch(I0, I1, 1, I0);
        fetch(I0, I1, 1, I1);
        sfor<0, 24>([&](auto K) __attribute__((always_inline)) {
            constexpr int k = decltype(K)::value, g = k / 6, s = k % 6, cur = g & 1, nxt = cur ^ 1;
            if constexpr (g < 3 && s < 4)
                fetch(std::integral_constant<int, nxt>{}, std::integral_constant<int, s / 2>{}, 2 * (g + 1) + s / 2,
                      std::integral_constant<int, s % 2>{});
            if constexpr (half == 0) {
                // stage 0 of chunk c: V rows 2 - 3 of chunk c
                if constexpr (k % 2 == 0) piece_second(std::integral_constant<int, k / 2>{}, bdst);
            } else {
                // stage 1 of chunk c: B^T d and V rows 0 - 1 of chunk c + 1; once its raw patches are consumed (slot 7), the
                // patches of chunk c + 2 start (8 slots x 4 loads): in flight for the rest of this stage and all of the next one
                if constexpr (k < 20) piece_first(K, bdst);
                if constexpr (k >= 8 && k < 16) {
                    if constexpr (k % 2 == 0) load_raw(rs, std::integral_constant<int, (k - 8) / 2>{}, I0, I2);
                    else load_raw(rs, std::integral_constant<int, (k - 8) / 2>{}, I2, I4);
                }
            }
            constexpr int e = s & 1, prod = s >> 1, q = half * 8 + 2 * g + e;
            if constexpr (TUNE & 2) {
            } else if constexpr (prod == 0) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(opa[cur][e][0], opb[cur][e][1], acc[q], 0, 0, 0);
            else if constexpr (prod == 1) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(opa[cur][e][1], opb[cur][e][0], acc[q], 0, 0, 0);
            else acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(opa[cur][e][0], opb[cur][e][0], acc[q], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    unsigned char* A0 = smem + kLdsA;
    unsigned char* A1 = smem + kLdsA + kStage;
    unsigned char* B0 = smem + kLdsB;
    unsigned char* B1 = smem + kLdsB + kStage;
    const int last = nchunks - 1;

    // ---- prologue: stage (0, 0) complete, chunk 1's patches in flight
    dma_weights(0, 0, A0);
    auto load_chunk = [&](int chunk) __attribute__((always_inline)) {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + (size_t)(chunk0 + chunk) * 16 * HW, 0, 16 * hw4, 0x00020000);
        load_raw(rs, I0, I0, I4);
        load_raw(rs, I1, I0, I4);
        load_raw(rs, I2, I0, I4);
        load_raw(rs, I3, I0, I4);
    };
    load_chunk(0);
    sfor<0, 20>([&](auto K) __attribute__((always_inline)) { piece_first(K, B0); });
    __builtin_amdgcn_sched_barrier(0);
    load_chunk(last < 1 ? last : 1);
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(32, 0));           // the weights have landed; the patches stay in flight
    __builtin_amdgcn_s_barrier();

    // one chunk = stage 0 (positions 0 - 7) + stage 1 (positions 8 - 15).  Past the last chunk the indices are clamped: the same
    // straight-line code runs, its results are never read.
    for (int c = 0; c < nchunks; ++c) {
        const int c1 = c + 1 < nchunks ? c + 1 : last, c2 = c + 2 < nchunks ? c + 2 : last;
        // stage (c, 0): multiplies V rows 0 - 1; writes rows 2 - 3 (from the kept half of B^T d)
        dma_weights(c, 1, A1);
        __builtin_amdgcn_sched_barrier(0);
        stage(I0, A0, B0, B1, c2);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));        // weights of stage (c, 1) and chunk c + 1's patches have landed
        __builtin_amdgcn_s_barrier();
        // stage (c, 1): multiplies rows 2 - 3; transforms chunk c + 1 (B^T d, rows 0 - 1 written, rows 2 - 3 kept); loads chunk c + 2
        dma_weights(c1, 0, A0);
        __builtin_amdgcn_sched_barrier(0);
        stage(I1, A1, B1, B0, c2);
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(32, 0));       // weights of stage (c + 1, 0) landed; the patches stay in flight
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));

    // ---- output transform on the accumulators: Y = A^T M A, A^T = (1 1 1 0; 0 1 -1 -1); then unscale, bias, ReLU, ...
    // Absent streams (bias, accumulate, mask) read through zero-sized buffer resources - zeros, no traffic, no branches.
    const bool partial = ksplit > 1;
    const bool accumulate = RMW && p.accumulate != 0 && !partial;
    const bool out_mask = RMW && p.out_mask != nullptr && !partial;
    const bool has_bias = p.bias != nullptr && !partial;
    const float relu_floor = (p.relu != 0 && !partial) ? 0.f : -__builtin_inff();
    const float mask_thr = out_mask ? 0.f : -1.f;
    const float unscale = pow2f(-(ea + eb));
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));                     // (epilogue addresses are derived here, not carried across the K loop)
    const int tile = tb * 32 + (lane_e & 31), oy = y0 + 2 * (tile / TX), ox = x0 + 2 * (tile % TX);
    const bool in0 = ox < W && oy < H, in1 = ox < W && oy + 1 < H;
    float* out_base = partial ? p.scratch + (size_t)ks * p.cout * HW : p.out;
    const int co_base = ct * 64 + cb * 32;
    const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(out_base + (size_t)co_base * HW, 0, 32 * hw4, 0x00020000);
    const __amdgpu_buffer_rsrc_t os_ld =
        __builtin_amdgcn_make_buffer_rsrc(out_base + (size_t)co_base * HW, 0, accumulate ? 32 * hw4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ms_ld = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(out_mask ? p.out_mask : out_base) + (size_t)co_base * HW, 0, out_mask ? 32 * hw4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t bs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(has_bias ? p.bias : out_base) + (has_bias ? co_base : 0), 0, has_bias ? 32 * 4 : 0, 0x00020000);
    const int PH = H >> 1, PW = W >> 1;
    const bool pool = !RMW && p.pool_out != nullptr && !partial;
    const bool coded = pool && p.pool_code != nullptr;
    const __amdgpu_buffer_rsrc_t ps = __builtin_amdgcn_make_buffer_rsrc((pool ? p.pool_out : out_base) + (size_t)co_base * (PH * PW), 0,
                                                                         pool ? 32 * PH * PW * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t cs = __builtin_amdgcn_make_buffer_rsrc(
        coded ? p.pool_code + (size_t)co_base * (PH * PW) : reinterpret_cast<unsigned char*>(out_base), 0, coded ? 32 * PH * PW : 0, 0x00020000);
    const int row0 = 4 * (lane_e >> 5);                                 // channel row of accumulator register r: (r & 3) + 8 (r >> 2) + row0
    const int off00 = in0 ? ((row0 * H + oy) * W + ox) * 4 : kOOR;
    const int off10 = in1 ? ((row0 * H + oy + 1) * W + ox) * 4 : kOOR;
    const int poff0 = in1 ? ((row0 * PH + (oy >> 1)) * PW + (ox >> 1)) * 4 : kOOR;
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        bias_r[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bs, (row0 + (r & 3) + 8 * (r >> 2)) * 4, 0, 0));
    unsigned int amax = 0;
    f32x2 nx_a0 = {0.f, 0.f}, nx_a1 = {0.f, 0.f}, nx_k0 = {0.f, 0.f}, nx_k1 = {0.f, 0.f};
    if constexpr (RMW) {
        nx_a0 = bload2(os_ld, off00, 0); nx_a1 = bload2(os_ld, off10, 0);
        nx_k0 = bload2(ms_ld, off00, 0); nx_k1 = bload2(ms_ld, off10, 0);
    }
    sfor<0, 16>([&](auto R) __attribute__((always_inline)) {
        constexpr int r = decltype(R)::value;
        constexpr int rr = (r & 3) + 8 * (r >> 2);
        const f32x2 a0 = nx_a0, a1 = nx_a1, k0 = nx_k0, k1 = nx_k1;
        if constexpr (RMW && r + 1 < 16) {               // the next register's streams, before this one's stores
            constexpr int rn = ((r + 1) & 3) + 8 * ((r + 1) >> 2);
            const int n0 = in0 ? off00 + rn * hw4 : kOOR, n1 = in1 ? off10 + rn * hw4 : kOOR;
            nx_a0 = bload2(os_ld, n0, 0); nx_a1 = bload2(os_ld, n1, 0);
            nx_k0 = bload2(ms_ld, n0, 0); nx_k1 = bload2(ms_ld, n1, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        float s0[4], s1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s0[b] = (acc[0 + b][r] + acc[4 + b][r]) + acc[8 + b][r];
            s1[b] = (acc[4 + b][r] - acc[8 + b][r]) - acc[12 + b][r];
        }
        const float bias = bias_r[r];
        f32x2 y0v, y1v;
        y0v[0] = fmaxf(((s0[0] + s0[1]) + s0[2]) * unscale + bias, relu_floor);
        y0v[1] = fmaxf(((s0[1] - s0[2]) - s0[3]) * unscale + bias, relu_floor);
        y1v[0] = fmaxf(((s1[0] + s1[1]) + s1[2]) * unscale + bias, relu_floor);
        y1v[1] = fmaxf(((s1[1] - s1[2]) - s1[3]) * unscale + bias, relu_floor);
        if constexpr (RMW) {
            y0v += a0;
            y1v += a1;
            y0v[0] = k0[0] > mask_thr ? y0v[0] : 0.f; y0v[1] = k0[1] > mask_thr ? y0v[1] : 0.f;
            y1v[0] = k1[0] > mask_thr ? y1v[0] : 0.f; y1v[1] = k1[1] > mask_thr ? y1v[1] : 0.f;
        }
        amax = max(amax, in0 ? max(abs_bits(y0v[0]), abs_bits(y0v[1])) : 0u);
        amax = max(amax, in1 ? max(abs_bits(y1v[0]), abs_bits(y1v[1])) : 0u);
        const int o0 = in0 ? off00 + rr * hw4 : kOOR, o1 = in1 ? off10 + rr * hw4 : kOOR;
        if (RMW || !coded) {
            bstore2(y0v, os, o0);
            bstore2(y1v, os, o1);
        }
        if constexpr (!RMW) {
            if (pool) {
                // MaxPool2d(2) of the tile = one window; the first maximum in row-major order (a later element wins only if
                // strictly greater); bit 2: the maximum is > 0 (the ReLU mask)
                float m = y0v[0];
                int at = 0;
                if (y0v[1] > m) { m = y0v[1]; at = 1; }
                if (y1v[0] > m) { m = y1v[0]; at = 2; }
                if (y1v[1] > m) { m = y1v[1]; at = 3; }
                const int po = in1 ? poff0 + rr * (PH * PW * 4) : kOOR;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, m), ps, po, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)(at | (m > 0.f ? 4 : 0)), cs, in1 ? (po >> 2) : kOOR, 0, 0);
            }
        }
    });
    if (p.out_amax && !partial) amax_commit(amax, p.out_amax);
}

int wino_force_tx() {
    static Option tx_opt("ST_WINO_TX", 0);
    return tx_opt.get();
}

}  // namespace

size_t winograd_weight_bytes(int cin, int cout) { return (size_t)16 * cin * cout * 2 * sizeof(_Float16) + 256; }

// The shapes this kernel takes (the launcher decides whether it is also the faster one: conv_wino_preferred)
bool conv_wino_applies(const ConvProblem& p) {
    return p.taps == 9 && p.planes == 2 && p.elem == 1 && p.wgt_wino && p.amax_word && !p.mask && !p.in_halo && p.cin % 16 == 0 &&
           p.cout % 64 == 0 && p.width % 2 == 0 && p.width >= 2 && p.height >= 1 && p.row_begin == 0 && p.row_end == 0 &&
           p.row_skip_len == 0 && p.overlap_part == 0 && (long long)16 * p.height * p.width * 4 < (long long)kOOR &&
           (long long)p.height * p.width * 32 * 4 < (1ll << 31);
}

bool conv_wino_preferred(const ConvProblem& p) {
    static Option use_opt("ST_CONV_WINO", 1);
    return use_opt.get() != 0 && conv_wino_applies(p);
}

// w: torch [Cout][Cin][3][3]; out: winograd_weight_bytes(); the trailer holds max |w| (word 0) and the scale exponent (word 8).
// dgrad: the planes of the data-gradient operator (Cout -> Cin channels, rotated taps)
int launch_winograd_weights(const float* w, void* out, int cin, int cout, int dgrad, hipStream_t s) {
    unsigned int* trailer = reinterpret_cast<unsigned int*>(static_cast<unsigned char*>(out) + (size_t)16 * cin * cout * 2 * sizeof(_Float16));
    ST_HIP(hipMemsetAsync(trailer, 0, 256, s));
    if (launch_amax(w, (long long)cin * cout * 9, trailer, 1, s)) return 1;
    const int kdim = dgrad ? cout : cin, mdim = dgrad ? cin : cout;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((cin * cout + 255) / 256), dim3(256), 0, s, w, static_cast<_Float16*>(out), kdim, mdim,
                       dgrad, trailer, reinterpret_cast<int*>(trailer + 8));
    ST_LAUNCH_CHECK();
    return 0;
}

namespace {
template <int TX, bool RMW, int TUNE = 0>
int launch_wino_cfg(const ConvProblem& p, int ksplit, hipStream_t s) {
    constexpr int TY = 64 / TX;
    auto kern = wino_conv_kernel<TX, RMW, TUNE>;
    static bool attr = false;
    if (!attr) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
        attr = true;
    }
    const int tiles_x = ceil_div(p.width, 2 * TX), tiles_y = ceil_div(p.height, 2 * TY);
    const int n_co = p.cout / 64;
    const long long total = (long long)tiles_x * tiles_y * n_co * ksplit;
    ST_REQUIRE(total > 0 && total < (1ll << 30), "winograd conv: grid out of range");
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), kLdsBytes, s, p, tiles_x, n_co, ksplit, p.cin / 16 / ksplit, (int)total);
    ST_LAUNCH_CHECK();
    if (ksplit > 1) return launch_conv_splitk_reduce(p, ksplit, s);
    return 0;
}
template <bool RMW>
int launch_wino_tx(const ConvProblem& p, int tx, int ksplit, hipStream_t s) {
    if constexpr (!RMW) {
        static Option tune_opt("ST_WINO_TUNE", 0);
        switch (tune_opt.get()) {       // ablation variants (timing only), 64 x 4 px tile rows
            case 1: return launch_wino_cfg<32, false, 1>(p, ksplit, s);
            case 2: return launch_wino_cfg<32, false, 2>(p, ksplit, s);
            case 3: return launch_wino_cfg<32, false, 3>(p, ksplit, s);
            case 4: return launch_wino_cfg<32, false, 4>(p, ksplit, s);
            case 8: return launch_wino_cfg<32, false, 8>(p, ksplit, s);
            case 12: return launch_wino_cfg<32, false, 12>(p, ksplit, s);
            case 13: return launch_wino_cfg<32, false, 13>(p, ksplit, s);
            case 14: return launch_wino_cfg<32, false, 14>(p, ksplit, s);
            case 15: return launch_wino_cfg<32, false, 15>(p, ksplit, s);
            case 31: return launch_wino_cfg<32, false, 31>(p, ksplit, s);
            case 32: return launch_wino_cfg<32, false, 32>(p, ksplit, s);
            case 64: return launch_wino_cfg<32, false, 64>(p, ksplit, s);
            case 36: return launch_wino_cfg<32, false, 36>(p, ksplit, s);
            case 68: return launch_wino_cfg<32, false, 68>(p, ksplit, s);
            default: break;
        }
    }
    if (tx == 32) return launch_wino_cfg<32, RMW>(p, ksplit, s);
    if (tx == 16) return launch_wino_cfg<16, RMW>(p, ksplit, s);
    return launch_wino_cfg<8, RMW>(p, ksplit, s);
}
}  // namespace

int launch_conv_wino(const ConvProblem& p, hipStream_t s) {
    ST_REQUIRE(conv_wino_applies(p), "winograd conv: unsupported problem (Cin %% 16, Cout %% 64, even width, fp16x3 planes, no halo)");
    // pixel tile: the widest of 64 x 4 / 32 x 8 / 16 x 16 px with the least padded area
    int tx = 32;
    {
        long long best = -1;
        for (int t : {32, 16, 8}) {
            const long long a = (long long)ceil_div(p.width, 2 * t) * 2 * t * (long long)ceil_div(p.height, 128 / t) * (128 / t);
            if (best < 0 || a < best) { best = a; tx = t; }
        }
        if (wino_force_tx() == 8 || wino_force_tx() == 16 || wino_force_tx() == 32) tx = wino_force_tx();
    }
    // K split: fill the chip when the layer has few tiles (the partial sums go through the direct kernel's reduce pass)
    int ksplit = 1;
    {
        static Option ks_opt("ST_WINO_KSPLIT", 0);
        const int nchunks = p.cin / 16;
        const long long pixels = (long long)p.height * p.width;
        const long long wgs = (long long)ceil_div(p.width, 2 * tx) * ceil_div(p.height, 128 / tx) * (p.cout / 64);
        if (ks_opt.get() > 0) {
            ksplit = ks_opt.get();
            ST_REQUIRE(ksplit == 1 || (p.scratch && nchunks % ksplit == 0 && (size_t)ksplit * p.cout * pixels <= kConvScratchFloats),
                       "ST_WINO_KSPLIT=%d does not fit this problem", ksplit);
        } else if (p.scratch) {
            while (wgs * ksplit < 192 && nchunks % (ksplit * 2) == 0 && nchunks / (ksplit * 2) >= 2 &&
                   (size_t)(ksplit * 2) * p.cout * pixels <= kConvScratchFloats)
                ksplit *= 2;
        }
    }
    if (p.accumulate || p.out_mask) return launch_wino_tx<true>(p, tx, ksplit, s);
    return launch_wino_tx<false>(p, tx, ksplit, s);
}

}  // namespace st
