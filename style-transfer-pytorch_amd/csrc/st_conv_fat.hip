// fp16x3 3x3 convolution, "fat" single-role form (round 6): the trunk's forward and data-gradient convolutions on large maps.
//
// Same arithmetic, LDS images, operand swizzle and K order as conv_pc_kernel (st_conv_pc.hip) - results are bit-identical -
// but no producer waves: ONE workgroup of four waves (one per SIMD, 512 registers each) per (32 CB) output channels x 512
// pixels; a wave owns CB x 4 accumulator tiles of 32 x 32 (CB = 4: 128 co x 128 px, 256 AGPRs) and stages the next chunk's
// activations ITSELF, between its own MFMAs.
//
// Why (profiles/r06_fat_conv.md): the producer / consumer tile is bound by its consumer pattern's LDS operand stream -
// 8 ds_read_b128 per 12 MFMAs sustain 77 % of the matrix peak before a byte of staging moves
// (profiles/r02_mfma_sustained.md), and the kernel sits 7 % below that.  A 128 x 128 register tile needs 16 fetches per 48
// MFMAs (half the LDS traffic per MFMA: the measured ceiling of that pattern is 85 %); and the staging work that starves in
// partner waves beside an MFMA stream (one VALU instruction per ~43 cycles) is nearly free INSIDE the stream that multiplies
// (MI355X_MICROARCH.md: up to five single-issue instructions hide behind each 8-pass MFMA of the issuing wave) - per MFMA
// this kernel has 0.33 operand fetches, 0.25 VALU instructions, 0.1 global loads.
// Measured: MFMAs + operand fetches alone run at 100 % of the matrix pipe (1.57 x conv_pc_kernel), loads and splits add 9 % -
// and the LDS WRITES of the staged planes another 33 %, whatever their width, addresses or place in the stream: the kernel
// ends 4 - 8 % ahead of conv_pc_kernel where a launch is whole rounds of long tiles, behind it elsewhere (conv_fat_preferred).
//
// A K chunk of 16 channels = two stages (taps 0 - 4, taps 5 - 8), one s_barrier each:
//   activations: two LDS images (chunk c multiplies out of image c & 1 while chunk c + 1 is written into the other one);
//                the patch of chunk c + 1 is loaded into registers during stage A of chunk c (one item = 4 pixels x 4 channels,
//                see FCfg) and split + written during stage B;
//   weights:     ONE image (32 CB co x 9 taps x 2 planes); the LDS-DMA of the next chunk's taps 0 - 4 is issued when stage A
//                of this chunk has ended, that of this chunk's taps 5 - 8 ... one stage ahead of their use, each.
#include "st_common.h"

namespace st {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr int SK = 16;
constexpr int kOOR = 0x40000000;

template <int B, int E, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        sfor<B + 1, E>(f);
    }
}

__device__ __forceinline__ float bload(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
constexpr int waitcnt_imm(int vm, int lgkm) { return (vm & 15) | ((vm >> 4) << 14) | (7 << 4) | ((lgkm & 15) << 8); }

template <int CB>
struct FCfg {
    static constexpr int TCO = 32 * CB, TW = 32, TH = 16, LW = TW + 2, LH = TH + 2;
    static constexpr int NPX = LH * LW;                     // 612 staged pixels
    static constexpr int ACT_PLANE = NPX * 32;
    static constexpr int PS = 39 * 512;                     // plane stride (19 968): the 384 bytes behind each plane absorb the LDS
                                                            // writes of staged pixels that are not part of the patch (see aoff);
                                                            // a multiple of 512 so that ONE ds_write2st64_b64 writes both planes
    static constexpr int ACT_BUF = 2 * PS;                  // two planes
    static constexpr int W_TAP = TCO * 32;                  // one plane of one tap
    static constexpr int W_PLANE = 9 * W_TAP;
    static constexpr int W_OFF = 2 * ACT_BUF;
    static constexpr int LDS = W_OFF + 2 * W_PLANE;         // CB = 4: 153 088 B, CB = 2: 116 224 B
    // staging items: 4 pixels (one 16-byte load per channel) x 4 channels.  A patch row is the ten 16-byte groups that cover
    // x0 - 4 .. x0 + 35 (the halo columns x0 - 1 and x0 + 32 are the last / first pixel of the edge groups): 18 rows x 10
    // groups x 4 channel quads = 720 items, three per thread (the vector-memory INSTRUCTION is what costs beside an MFMA
    // stream - ~150 cycles of the wave's issue each, measured - so the patch is fetched in as few of them as the layout allows)
    static constexpr int GROUPS = 10, ITEMS = LH * GROUPS * 4;
    static constexpr int NIT = (ITEMS + 255) / 256;         // 3
};

// CB = 32-channel output blocks per wave; RMW: the epilogue reads the tensor it writes (accumulate) / a ReLU mask (out_mask)
// TUNE: ablation bits (ST_CONV_FAT_TUNE, timing only, wrong results): 1 no activation staging (loads, split, LDS writes),
// 2 no MFMA / operand fetches, 4 no weight DMA, 8 no barriers
template <int CB, bool RMW, int TUNE = 0>
__global__ __launch_bounds__(256) void conv_fat_kernel(ConvProblem p, int tiles_x, int n_co_tiles, int ksplit, int nchunks, int total) {
    using C = FCfg<CB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int H = p.height, W = p.width, HW = H * W;

    int bid = blockIdx.x;
    if ((total & 7) == 0) bid = (bid & 7) * (total >> 3) + (bid >> 3);          // consecutive ids (the Cout tiles of a pixel tile) on one XCD
    const int ct = bid % n_co_tiles;
    bid /= n_co_tiles;
    const int ks = bid % ksplit;
    bid /= ksplit;
    const int x0 = (bid % tiles_x) * C::TW, y0 = (bid / tiles_x) * C::TH;
    const int co0 = ct * C::TCO, chunk0 = ks * nchunks;

    const unsigned char* wsplit = static_cast<const unsigned char*>(p.wgt_split);
    const int w_plane_stride = 9 * (p.cin / SK) * p.cout * 32;                  // bytes per plane
    const int w_tap_stride = (p.cin / SK) * p.cout * 32;
    const int ea = scale_exp(amax_read(p.amax_word));
    const int ew = scale_exp(*reinterpret_cast<const unsigned int*>(wsplit + (size_t)2 * w_plane_stride));
    const float in_scale = pow2f(ea);
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wsplit), 0, 2 * w_plane_stride, 0x00020000);

    // ---- staging role: NIT items (patch row r, 16-byte group gx, channel quad cq) per thread ----
    int goff[C::NIT], aoff[C::NIT][4];
#pragma unroll
    for (int i = 0; i < C::NIT; ++i) {
        const int it = tid + i * 256;
        const int cq = it / (C::LH * C::GROUPS), rem = it % (C::LH * C::GROUPS), r = rem / C::GROUPS, gx = rem % C::GROUPS;
        const int y = y0 - 1 + r, xg = x0 - 4 + 4 * gx;
        const bool ok = it < C::ITEMS && y >= 0 && y < H && xg >= 0 && xg + 3 < W;      // (W % 4 == 0: a group is inside or outside as a whole)
        goff[i] = ok ? (4 * cq * HW + y * W + xg) * 4 : kOOR;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = 4 * gx + j - 3, q = r * C::LW + col;
            aoff[i][j] = (it < C::ITEMS && col >= 0 && col < C::LW) ? q * 32 + (((cq >> 1) ^ ((q >> 3) & 1)) * 16) + (cq & 1) * 8
                                                                    : C::ACT_PLANE + (tid & 15) * 8;      // the gap behind the plane
        }
    }
    f32x4 raw[C::NIT][4];                                  // [item][channel of the quad]: 4 pixels each
    const int hw4 = HW * 4;
    auto load_one = [&](const __amdgpu_buffer_rsrc_t rs, auto I, auto CH) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, c = decltype(CH)::value;
        if constexpr (TUNE & 1) return;
        raw[i][c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, goff[i], c * hw4, 0));
    };
    // One pixel of an item: split its four channel values into the two planes (h0 = fp16(v), h1 = fp16(v - h0): exact
    // subtraction, one rounding) - 4 + 6 VALU instructions, the split as inline assembly (the compiler's lowering of the C
    // expressions takes twice as many).  The planes stay in registers (cvt) until the write burst behind the next barrier.
    u32x2 cvt[C::NIT][4][2];                               // [item][pixel][plane]
    auto convert_px = [&](auto I, auto PX) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, j = decltype(PX)::value;
        if constexpr (TUNE & 1) return;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float a = raw[i][2 * e][j] * in_scale, b = raw[i][2 * e + 1][j] * in_scale;
            unsigned int lo, hi;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
            asm("v_fma_mixlo_f16 %0, -%1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(lo), "v"(a));
            asm("v_fma_mixhi_f16 %0, -%1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(hi) : "v"(lo), "v"(b));
            cvt[i][j][0][e] = lo;
            cvt[i][j][1][e] = hi;
        }
    };
    // ... both planes of a pixel in ONE LDS instruction (PS is a multiple of 512 bytes).  A pixel outside the patch - the three
    // spare pixels of an edge group, the items past the end - lands in the gaps behind the two planes.
    auto write_px = [&](unsigned char* act, auto I, auto PX) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, j = decltype(PX)::value;
        if constexpr (!(TUNE & 1)) {
            const unsigned int addr = (unsigned int)(uintptr_t)(act - smem) + (unsigned int)aoff[i][j];
            const u32x2 p0 = cvt[i][j][0], p1 = cvt[i][j][1];
            asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:39" :: "v"(addr), "v"(p0), "v"(p1) : "memory");
        }
    };
    // weights of taps [T0, T1) of a chunk: 1 KB pieces (32 rows of one plane and tap), dealt round the four waves.  The
    // 16-byte halves of a row are swapped for every other group of 8 rows on the SOURCE side (the DMA writes base + 16 lane)
    const int dma_lane = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) * 16);
    auto dma_weights = [&](int chunk, auto T0, auto T1) __attribute__((always_inline)) {
        constexpr int t0 = decltype(T0)::value, t1 = decltype(T1)::value;
        constexpr int NP = 2 * (t1 - t0) * CB;                                   // pieces
        if constexpr (TUNE & 4) return;
        const int base = ((chunk0 + chunk) * p.cout + co0) * 32;
        sfor<0, (NP + 3) / 4>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int j = i * 4 + wave;                                          // (plane, tap, 32-row block), wave-uniform
            static_assert(NP % 4 == 0, "the pieces of a stage are dealt evenly to the four waves (no branch in the stream)");
            const int pl = j / ((t1 - t0) * CB), rest = j % ((t1 - t0) * CB), tap = t0 + rest / CB, blk = rest % CB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                wrs, (__attribute__((address_space(3))) void*)(smem + C::W_OFF + pl * C::W_PLANE + tap * C::W_TAP + blk * 1024), 16,
                dma_lane, pl * w_plane_stride + tap * w_tap_stride + base + blk * 1024, 0, 0);
        });
    };

    // ---- MFMA role: wave w owns rows 4 w .. 4 w + 3 of the 16 x 32 pixel tile (pixel block j = one row) x CB channel blocks ----
    const int a_lane = C::W_OFF + l31 * 32 + ((half ^ ((l31 >> 3) & 1)) * 16);
    const int b_q0 = 4 * wave * C::LW + l31;               // the lane's pixel in block 0, tap 0; the swizzled offset of (block, tap)
                                                           // is recomputed where it is used (4 VALU instructions, hidden)
    auto b_off = [&](int j, int tap) __attribute__((always_inline)) {
        const int q = b_q0 + (j + tap / 3) * C::LW + tap % 3;
        return q * 32 + ((half ^ ((q >> 3) & 1)) * 16);
    };
    f32x16 acc[CB][4];
#pragma unroll
    for (int i = 0; i < CB; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr std::integral_constant<int, 0> I0{};
    constexpr std::integral_constant<int, 5> I5{};
    constexpr std::integral_constant<int, 9> I9{};
    f16x8 av[2][CB][2], bv[2][2];                          // weight operands of a tap and pixel-block operands, both double buffered

    // One stage: taps [T0, T1) of the chunk in image `act`.  Per tap CB x 4 tiles x 3 products, ordered by pixel block (the
    // block's operands are fetched while the block before it multiplies, the next tap's weights behind the last block).  The
    // staging work of the stage is cut into pieces, one per (tap, pixel block) slot; a slot is one scheduling region, so the
    // interleave is the source order.
    auto stage = [&](auto T0, auto T1, const unsigned char* act, auto&& piece) __attribute__((always_inline)) {
        constexpr int t0 = decltype(T0)::value, t1 = decltype(T1)::value;
        auto fetch_a = [&](auto TAP) __attribute__((always_inline)) {
            constexpr int tap = decltype(TAP)::value, ab = tap & 1;
            if constexpr (TUNE & 2) return;
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                av[ab][i][0] = *reinterpret_cast<const f16x8*>(smem + a_lane + tap * C::W_TAP + i * 1024);
                av[ab][i][1] = *reinterpret_cast<const f16x8*>(smem + a_lane + C::W_PLANE + tap * C::W_TAP + i * 1024);
            }
        };
        auto fetch_b = [&](auto BUF, auto TAP, auto J) __attribute__((always_inline)) {
            constexpr int bf = decltype(BUF)::value, tap = decltype(TAP)::value, j = decltype(J)::value;
            if constexpr (TUNE & 2) return;
            const int off = b_off(j, tap);
            bv[bf][0] = *reinterpret_cast<const f16x8*>(act + off);
            bv[bf][1] = *reinterpret_cast<const f16x8*>(act + C::PS + off);
        };
        fetch_a(T0);
        fetch_b(I0, T0, I0);
        sfor<t0, t1>([&](auto TAP) __attribute__((always_inline)) {
            constexpr int tap = decltype(TAP)::value;
            sfor<0, 4>([&](auto J) __attribute__((always_inline)) {
                constexpr int j = decltype(J)::value, slot = (tap - t0) * 4 + j, cur = slot & 1, nxt = cur ^ 1, ab = tap & 1;
                if constexpr (j < 3) fetch_b(std::integral_constant<int, nxt>{}, TAP, std::integral_constant<int, j + 1>{});
                else if constexpr (tap + 1 < t1) fetch_b(std::integral_constant<int, nxt>{}, std::integral_constant<int, tap + 1>{}, I0);
                if constexpr (j == 1 && tap + 1 < t1) fetch_a(std::integral_constant<int, tap + 1>{});      // two blocks ahead of its use
                piece(std::integral_constant<int, slot>{});
                // (the piece stays in ONE gap of the MFMA stream: a VALU instruction between two back-to-back MFMAs costs ~43
                // cycles for the first one of a gap and ~6 for every further one - scattered over the slot's 12 gaps the 120
                // VALU instructions of a chunk cost 5 600 cycles, measured; MI355X_MICROARCH.md "one extra issue slot")
                __builtin_amdgcn_sched_barrier(0);
                // cross terms first, the dominant a0 * b0 last (the order of conv_split_kernel / conv_pc_kernel: bit-identical sums)
                if constexpr (!(TUNE & 2)) {
#pragma unroll
                for (int i = 0; i < CB; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ab][i][0], bv[cur][1], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CB; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ab][i][1], bv[cur][0], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < CB; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[ab][i][0], bv[cur][0], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };

    unsigned char* act0 = smem;
    unsigned char* act1 = smem + C::ACT_BUF;
    const int last = nchunks - 1;
    auto chunk_rsrc = [&](int chunk) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.in) + (size_t)(chunk0 + chunk) * SK * HW, 0, SK * hw4, 0x00020000);
    };

    // ---- prologue: chunk 0 staged, its weights landed ----
    dma_weights(0, I0, I5);
    {
        const __amdgpu_buffer_rsrc_t rs = chunk_rsrc(0);
        sfor<0, 4 * C::NIT>([&](auto L) __attribute__((always_inline)) {
            load_one(rs, std::integral_constant<int, decltype(L)::value / 4>{}, std::integral_constant<int, decltype(L)::value % 4>{});
        });
    }
    sfor<0, 4 * C::NIT>([&](auto L) __attribute__((always_inline)) {
        convert_px(std::integral_constant<int, decltype(L)::value / 4>{}, std::integral_constant<int, decltype(L)::value % 4>{});
        write_px(act0, std::integral_constant<int, decltype(L)::value / 4>{}, std::integral_constant<int, decltype(L)::value % 4>{});
    });
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
    if constexpr (!(TUNE & 8)) __builtin_amdgcn_s_barrier();

    // Where the staging work sits in the MFMA stream (measured on conv4_2 at 2048^2, 64 chunks per CU, profiles/r06_fat_conv.md):
    // MFMAs + operand fetches alone run the chunk in its 13 824 cycles of MFMA issue (100 % busy); the patch loads add 0.4 us
    // per chunk, the splits (VALU) 0.3 us - and the LDS WRITES of the planes 2.3 us, whether as 24 ds_write_b64, 12
    // ds_write2st64_b64 or to conflict-free addresses, scattered over the slots, gathered into one gap per item, or issued as a
    // burst behind the barrier (worse: then nothing overlaps them): a store's data goes from the VGPRs to the LDS through a
    // path that the MFMAs' operand traffic keeps busy, and the wave that waits for it is the wave that multiplies.  One item
    // (4 loads / 4 pixels) per gap is what is kept.
    for (int c = 0; c < nchunks; ++c) {
        const int c1 = c + 1 < nchunks ? c + 1 : last;               // (past the end: the same straight-line code, results unused)
        unsigned char* act = (c & 1) ? act1 : act0;
        unsigned char* nxt = (c & 1) ? act0 : act1;
        // stage A: taps 0 - 4; this chunk's taps 5 - 8 start (every wave is past stage B of the chunk before); the patch of
        // chunk c + 1 is loaded
        dma_weights(c, I5, I9);
        __builtin_amdgcn_sched_barrier(0);
        {
            const __amdgpu_buffer_rsrc_t rs = chunk_rsrc(c1);
            stage(I0, I5, act, [&](auto S) __attribute__((always_inline)) {
                constexpr int s = decltype(S)::value;
                if constexpr (s == 2 || s == 8 || s == 14)
                    sfor<0, 4>([&](auto CH) __attribute__((always_inline)) { load_one(rs, std::integral_constant<int, (s - 2) / 6>{}, CH); });
            });
        }
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(4 * C::NIT, 0));     // this chunk's taps 5 - 8 have landed; the patch loads stay in flight
        if constexpr (!(TUNE & 8)) __builtin_amdgcn_s_barrier();
        // stage B: taps 5 - 8; the next chunk's taps 0 - 4 start (every wave is past stage A); chunk c + 1 is split and written
        dma_weights(c1, I0, I5);
        __builtin_amdgcn_sched_barrier(0);
        stage(I5, I9, act, [&](auto S) __attribute__((always_inline)) {
            constexpr int s = decltype(S)::value;
            if constexpr (s == 3 || s == 8 || s == 13)
                sfor<0, 4>([&](auto PX) __attribute__((always_inline)) {
                    convert_px(std::integral_constant<int, (s - 3) / 5>{}, PX);
                    write_px(nxt, std::integral_constant<int, (s - 3) / 5>{}, PX);
                });
        });
        __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));              // the next chunk's taps 0 - 4 have landed; LDS writes done
        if constexpr (!(TUNE & 8)) __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(0, 0));
    if constexpr (!(TUNE & 8)) __builtin_amdgcn_s_barrier();                                   // (the images become the epilogue's slabs)

    // ---- epilogue (as conv_pc_kernel's 16-byte path): a channel block's 32 x 128 slab is transposed through LDS, whole
    // float4s move along the image rows.  Absent streams (bias, accumulate, mask) read through zero-sized resources.
    const bool partial = ksplit > 1;
    const bool accumulate = RMW && p.accumulate != 0 && !partial;
    const bool out_mask = RMW && p.out_mask != nullptr && !partial;
    const bool has_bias = p.bias != nullptr && !partial;
    const float relu_floor = (p.relu != 0 && !partial) ? 0.f : -__builtin_inff();
    const float mask_thr = out_mask ? 0.f : -1.f;
    const float unscale = pow2f(-(ea + ew));
    constexpr int TP = 128 + 8;                                                 // slab pitch: 4 rows apart = 32 banks apart
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * TP);
    float* out_base = partial ? p.scratch + (size_t)ks * p.cout * HW : p.out;
    const int PH = H >> 1, PW = W >> 1;
    const bool pool = !RMW && p.pool_out != nullptr && !partial;
    const bool coded = pool && p.pool_code != nullptr;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int lrow = lane_e >> 5, k = lane_e & 31;                              // channel row of a row group; 4 pixels at 4 k of the 128
    const int jrow = k >> 3, px = (k & 7) * 4;
    const int y = y0 + 4 * wave + jrow, x = x0 + px;
    const bool inb = y < H && x < W;
    const bool pin = inb && (jrow & 1) == 0 && y + 1 < H;                       // the lane of a 2 x 2 window's first row stores the pooled pair
    unsigned int amax_e = 0;
    sfor<0, CB>([&](auto IB) __attribute__((always_inline)) {
        constexpr int i = decltype(IB)::value;
        const int co_base = co0 + i * 32;
        const __amdgpu_buffer_rsrc_t os = __builtin_amdgcn_make_buffer_rsrc(out_base + (size_t)co_base * HW, 0, 32 * hw4, 0x00020000);
        const __amdgpu_buffer_rsrc_t os_ld =
            __builtin_amdgcn_make_buffer_rsrc(out_base + (size_t)co_base * HW, 0, accumulate ? 32 * hw4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t ms_ld = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(out_mask ? p.out_mask : out_base) + (size_t)co_base * HW, 0, out_mask ? 32 * hw4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t bs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(has_bias ? p.bias : out_base) + (has_bias ? co_base : 0), 0, has_bias ? 32 * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t ps = __builtin_amdgcn_make_buffer_rsrc((pool ? p.pool_out : out_base) + (size_t)co_base * (PH * PW), 0,
                                                                             pool ? 32 * PH * PW * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t cs = __builtin_amdgcn_make_buffer_rsrc(
            coded ? p.pool_code + (size_t)co_base * (PH * PW) : reinterpret_cast<unsigned char*>(out_base), 0, coded ? 32 * PH * PW : 0, 0x00020000);
        __builtin_amdgcn_wave_barrier();                   // the previous block's slab reads are done (in-order LDS)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[((r & 3) + 8 * (r >> 2) + 4 * half) * TP + j * 32 + l31] = acc[i][j][r] * unscale;
        __builtin_amdgcn_wave_barrier();
        const unsigned off0 = inb ? (unsigned)((lrow * H + y) * W + x) * 4u : 0x7FFFFFFFu;
        const unsigned step = (unsigned)(2 * HW) * 4u;                          // two channel rows per row group
        const unsigned poff0 = pin ? (unsigned)((lrow * PH + (y >> 1)) * PW + (x >> 1)) * 4u : 0x7FFFFFFFu;
        const unsigned pstep = (unsigned)(2 * PH * PW) * 4u;
        const float* slab_rd = slab + lrow * TP + k * 4;
        f32x4 o_next = {0.f, 0.f, 0.f, 0.f}, m_next = {0.f, 0.f, 0.f, 0.f};
        if constexpr (RMW) {
            o_next = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(os_ld, (int)off0, 0, 0));
            m_next = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ms_ld, (int)off0, 0, 0));
        }
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int off = (int)(off0 + (unsigned)q * step);
            f32x4 v = *reinterpret_cast<const f32x4*>(slab_rd + q * (2 * TP));
            const float bvv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(bs, (2 * q + lrow) * 4, 0, 0));
            const f32x4 o = o_next, m = m_next;
            if constexpr (RMW) {
                const int off1 = (q + 1 < 16) ? (int)(off0 + (unsigned)(q + 1) * step) : 0x7FFFFFFF;
                o_next = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(os_ld, off1, 0, 0));
                m_next = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ms_ld, off1, 0, 0));
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x_ = fmaxf(v[e] + bvv, relu_floor);
                if constexpr (RMW) {
                    x_ += o[e];
                    x_ = (m[e] > mask_thr) ? x_ : 0.f;
                }
                v[e] = x_;
            }
            if (RMW || !coded) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), os, off, 0, 0);
            if constexpr (!RMW) {
                if (pool) {
                    // fused MaxPool2d(2): lanes k and k + 8 hold the same 4 columns of the two image rows of a window (DPP row
                    // rotate by 8); the first maximum in row-major order, bit 2 = the maximum is > 0 (as conv_pc_kernel)
                    float m0 = fmaxf(v[0], v[1]), m1 = fmaxf(v[2], v[3]);
                    const int mine = (v[1] > v[0] ? 1 : 0) | (v[3] > v[2] ? 2 : 0);
                    const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m0), 0x128, 0xf, 0xf, false));
                    const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m1), 0x128, 0xf, 0xf, false));
                    const int theirs = __builtin_amdgcn_update_dpp(0, mine, 0x128, 0xf, 0xf, false);
                    const int at0 = n0 > m0 ? 2 + (theirs & 1) : (mine & 1);
                    const int at1 = n1 > m1 ? 2 + ((theirs >> 1) & 1) : ((mine >> 1) & 1);
                    m0 = fmaxf(m0, n0);
                    m1 = fmaxf(m1, n1);
                    const unsigned poff = poff0 + (unsigned)q * pstep;
                    const f32x2 pv = {m0, m1};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, pv), ps, (int)poff, 0, 0);
                    const int two = (at0 | (m0 > 0.f ? 4 : 0)) | ((at1 | (m1 > 0.f ? 4 : 0)) << 8);
                    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)two, cs, pin ? (int)(poff >> 2) : 0x7FFFFFFF, 0, 0);
                }
            }
            amax_e = max(amax_e, inb ? max(max(abs_bits(v[0]), abs_bits(v[1])), max(abs_bits(v[2]), abs_bits(v[3]))) : 0u);
        }
    });
    if (p.out_amax && !partial) amax_commit(amax_e, p.out_amax);
}

template <int CB, bool RMW, int TUNE = 0>
int launch_fat_cfg(const ConvProblem& p, int ksplit, hipStream_t s) {
    using C = FCfg<CB>;
    auto kern = conv_fat_kernel<CB, RMW, TUNE>;
    static bool attr = false;
    if (!attr) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        attr = true;
    }
    const int tiles_x = ceil_div(p.width, C::TW), tiles_y = ceil_div(p.height, C::TH);
    const int n_co = p.cout / C::TCO;
    const long long total = (long long)tiles_x * tiles_y * n_co * ksplit;
    ST_REQUIRE(total > 0 && total < (1ll << 30), "conv (fat tile): grid out of range");
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), C::LDS, s, p, tiles_x, n_co, ksplit, p.cin / SK / ksplit, (int)total);
    ST_LAUNCH_CHECK();
    if (ksplit > 1) return launch_conv_splitk_reduce(p, ksplit, s);
    return 0;
}

}  // namespace

bool conv_fat_applies(const ConvProblem& p) {
    return p.taps == 9 && p.planes == 2 && p.elem == 1 && p.wgt_split && p.amax_word && !p.mask && !p.in_halo && p.cin % SK == 0 &&
           p.cout % 64 == 0 && p.width % 4 == 0 && p.row_begin == 0 && p.row_end == 0 && p.row_skip_len == 0 && p.overlap_part == 0 &&
           ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.out_mask) | reinterpret_cast<uintptr_t>(p.pool_out)) & 15) == 0 &&
           (long long)SK * p.height * p.width * 4 < (long long)kOOR && (long long)p.height * p.width * 32 * 4 < (1ll << 31) &&
           (long long)9 * (p.cin / SK) * p.cout * 32 * 2 < (1ll << 31);
}

// Where this form is the faster one (profiles/r06_fat_conv.md): layers of >= 128 channels on both sides whose (128 co x 512 px)
// tiles run in at least two WHOLE rounds of the chip (one workgroup per CU at a time, no persistent loop: a partial last round
// idles CUs, and a single round exposes the prologue) and whose K loop has >= 16 chunks - conv3_x, conv4_x at 2048^2: 1.04 -
// 1.08 x the producer / consumer kernel per launch, +1.6 % on the iteration; measured 0.3 - 0.9 x where a launch has fewer than
// 256 workgroups or 64 output channels, and in the iteration -1.7 % at 1024^2 (conv2_2's 8-chunk tiles beside the heads'
// kernels) and -3.5 % at 2896 x 2172 (partial rounds) under looser rules.  ST_CONV_FAT: 0 never, 1 this rule (default),
// 2 wherever the kernel applies (A/B runs).
bool conv_fat_preferred(const ConvProblem& p) {
    static Option fat_opt("ST_CONV_FAT", 1);
    const int mode = fat_opt.get();
    if (mode == 0 || !conv_fat_applies(p)) return false;
    if (mode == 2) return true;
    if (p.cout % 128 != 0 || p.cin < 256) return false;      // (>= 16 chunks per tile: a short K loop exposes the tile's prologue and epilogue)
    const long long wgs = (long long)ceil_div(p.width, 32) * ceil_div(p.height, 16) * (p.cout / 128);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
    static Option rounds_opt("ST_CONV_FAT_ROUNDS", 1);     // whole rounds of workgroups a launch must have (tools/fat_rounds_ab.sh)
    return wgs >= (long long)rounds_opt.get() * cus && wgs % cus == 0;
}

int launch_conv_fat(const ConvProblem& p, hipStream_t s) {
    ST_REQUIRE(conv_fat_applies(p), "conv (fat tile): unsupported problem");
    const bool rmw = p.accumulate || p.out_mask;
    static Option cb_opt("ST_CONV_FAT_CB", 0);               // 2 / 4: force the channel blocks per wave (A/B runs)
    const int cb = (cb_opt.get() == 2 || p.cout % 128 != 0) ? 2 : 4;
    static Option tune_opt("ST_CONV_FAT_TUNE", 0);
    if (cb == 4 && !rmw)
        switch (tune_opt.get()) {               // ablation variants (timing only)
            case 1: return launch_fat_cfg<4, false, 1>(p, 1, s);
            case 2: return launch_fat_cfg<4, false, 2>(p, 1, s);
            case 4: return launch_fat_cfg<4, false, 4>(p, 1, s);
            case 5: return launch_fat_cfg<4, false, 5>(p, 1, s);
            case 7: return launch_fat_cfg<4, false, 7>(p, 1, s);
            case 15: return launch_fat_cfg<4, false, 15>(p, 1, s);
            default: break;
        }
    if (cb == 4) return rmw ? launch_fat_cfg<4, true>(p, 1, s) : launch_fat_cfg<4, false>(p, 1, s);
    return rmw ? launch_fat_cfg<2, true>(p, 1, s) : launch_fat_cfg<2, false>(p, 1, s);
}

}  // namespace st
