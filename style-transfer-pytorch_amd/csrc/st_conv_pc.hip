// fp16x3 3x3 convolution, producer / consumer form (the trunk's forward and data-gradient convs, unsharded and
// strip-sharded).
//
// Same arithmetic, LDS images, operand swizzle and epilogue as conv_split_kernel<TW, WN, 2, 1, false, HALO>
// (st_conv_split.hip) - results are bit-identical - but the two phases of a K chunk no longer alternate inside
// every wave.  ONE persistent workgroup per CU walks through its tiles (XCD-aware order); its waves have fixed roles:
//   consumers (waves 0 .. CW-1): ds_read_b128 + v_mfma_f32_32x32x16_f16 on LDS image c & 1, nothing else in the
//                                MFMA stream; before a chunk's MFMAs they issue the LDS-DMA (global_load_lds_dwordx4)
//                                of the NEXT chunk's pre-split weights into image (c + 1) & 1;
//   producers (the other waves): global loads of the activations of the chunks ahead -> registers; convert / split
//                                chunk c + 1 into two fp16 planes and write it into image (c + 1) & 1;
// one s_barrier per chunk (the barrier's fence retires the DMA; the producers' plain loads stay in flight across
// it).  The producers are already staging the next tile while the consumers write the finished one out.
//
// Why: in the single-role kernel every workgroup goes load-issue -> MFMA -> barrier -> wait for loads -> convert ->
// ds_write -> barrier, the matrix pipe idles through the second half, and the second workgroup of the CU does not
// fill the hole because both start together and stay in phase (s_memtime stamps: MFMA phase 3944 cycles per chunk
// with two workgroups per CU, 2993 alone, 1728 of pure MFMA issue; PMC: matrix pipe 43 % busy).  Here the SIMD's
// matrix pipe belongs to waves that never leave their MFMA stream, the staging VALU / VMEM / ds_write work of the
// partner waves issues beside it, and the global-load latency has one to two chunk periods to hide in.
//
// Tile shapes (PCfg): 64co x 512px "XL" (8 consumers + 4 producers; 125 instead of 81 FLOP per staged byte) wherever
// a layer has >= 256 such tiles; 64co x 256px (4 + 4) and 64co x 128px (4 + 8) for the small deep layers; split-K
// when that pays.  choose_pc_tile() picks shape, width and K split by a fitted cost model.
#include <mutex>
#include <type_traits>

#include "st_common.h"

#ifndef ST_PC_WDMA_ALL
#define ST_PC_WDMA_ALL 1
#endif

namespace st {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int SK = 16;                      // input channels per chunk (= K of one MFMA)
constexpr int kOOR = 0x40000000;            // buffer offset beyond every resource: loads return 0

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

// WN = 32-pixel blocks per consumer wave, CW = consumer waves: the tile is 64 output channels x (32 WN CW) pixels
// amax_commit (st_common.h) with ds_swizzle butterflies instead of __shfl_xor: the bpermute lane addresses of
// __shfl_xor are shared with amax_read at the top of the kernel by CSE and then stay live (and get spilled) across
// the whole K loop; the swizzle patterns are immediates.
__device__ __forceinline__ void amax_commit_lean(unsigned int m, unsigned int* bound) {
    sfor<0, 5>([&](auto K) __attribute__((always_inline)) {
        constexpr int k = 1 << decltype(K)::value;                     // lane ^ k within each half of the wave
        const unsigned int o = (unsigned int)__builtin_amdgcn_ds_swizzle((int)m, (k << 10) | 0x1f);
        m = o > m ? o : m;
    });
    const unsigned int o = (unsigned int)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63) ^ 32) << 2), (int)m);
    m = o > m ? o : m;
    unsigned int* slot = bound + (blockIdx.x % kAmaxSlots) * kAmaxSlotStride;
    if ((threadIdx.x & 63) == 0 && (m >> 23) > (*slot >> 23)) atomicMax(slot, m);
}

template <int TW, int WN, int CW>
struct PCfg {
    // producer threads beside the consumers.  64co x 128px tile: 4 + 8 waves (3 per SIMD, 168 registers each);
    // 64co x 256px tile: its consumers hold two 32-pixel blocks (~200 registers), so 4 + 4 waves (2 per SIMD, 256).
    // 64co x 512px tile (CW = 8, WN = 2: 125 instead of 81 FLOP per staged byte): 8 + 4 waves (168 registers); its two
    // images fill the CU's LDS, so the epilogue slabs live in the image the finished tile's last chunk was read
    // from and the producers wait one extra barrier at every tile boundary before refilling it (XL).
    static constexpr bool XL = CW == 8;
    // the consumers stage the weights by LDS-DMA (see dma_weights); false: the producers stage them through registers
    static constexpr bool WDMA = ST_PC_WDMA_ALL ? true : XL;
    static constexpr int PT = (WN == 1 && !XL) ? 512 : 256;
    static constexpr int THREADS = 64 * CW + PT;
    // producer register sets = chunks of global loads in flight (the XL tile's chunk period covers the load latency
    // and two of its 50-load sets would overflow the 6-bit vmcnt)
    static constexpr int SETS = XL ? 1 : 2;
    static constexpr int TCO = 64;
    static constexpr int NPIX = 32 * WN * CW;
    static constexpr int TH = NPIX / TW;
    static constexpr int LH = TH + 2, LW = TW + 2;
    static constexpr int NPX = LH * LW;                    // staged pixels (with the 1-pixel halo)
    static constexpr int ACT_PLANE = NPX * 32;             // bytes
    static constexpr int W_PLANE = 9 * TCO * 32;           // bytes
    static constexpr int BUF = 2 * (ACT_PLANE + W_PLANE);  // one LDS image: 2 act planes, then 2 weight planes
    static constexpr int W_OFF = 2 * ACT_PLANE;
    static constexpr int NIT = (2 * NPX + PT - 1) / PT;    // (pixel, 8-channel group) items per producer thread
    static constexpr int NWP = 9 * TCO * 2;                // 16-byte weight pieces per plane
    static constexpr int NWT = (NWP + PT - 1) / PT;
};

template <int TW, int WN, int CW, bool HALO>
__global__ __launch_bounds__((CW == 8 || WN == 1 ? 768 : 512), 1) void conv_pc_kernel(ConvProblem p, int tiles_x, int n_co_tiles, int ksplit,
                                                         int nchunks, int total) {
    using C = PCfg<TW, WN, CW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // 2 images, then the epilogue slabs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Roles by wave index: waves 0 .. CW-1 multiply, the others stage.  (Tried: the staging waves as the workgroup's
    // FIRST waves - neutral, 512^2 1470 / 1476 us, 2048^2 21.7 / 21.7 ms for the 12 forward convs: a wave's VALU issue
    // beside saturated MFMA streams does not depend on its age or priority, profiles/r02_mfma_sustained.md.)
    const bool producer = wave >= CW;
    const int cwave = wave;                                         // consumer index
    const int wn = cwave;                                           // consumer: position along the pixel dimension
    const int ptid = tid - 64 * CW;                                 // producer: staging thread index
    const int l31 = lane & 31, half = lane >> 5;
    const int H = p.height, W = p.width, HW = H * W;
    const int Y0 = p.row_begin, Y1 = p.row_end ? p.row_end : H;    // output rows of this launch

    // Persistent workgroups: this one works through the tiles v = blockIdx.x, + gridDim.x, ... (gridDim.x is a
    // multiple of 8 or equals `total`).  XCD-aware order as in conv_split_kernel: XCD x (= v & 7) takes the contiguous
    // range [x total / 8, (x + 1) total / 8) of logical ids, in which consecutive ids are the Cout tiles (and K
    // slices) of ONE pixel tile, so they share its activations through that XCD's L2.
    const int my_tiles = (total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int gtot = my_tiles * nchunks;                            // chunks this workgroup multiplies
    struct Tile { int x0, y0, co0, kslice; };
    auto tile_of = [&](int ordinal) __attribute__((always_inline)) {
        const int v = blockIdx.x + ordinal * gridDim.x;
        int bid = ((total & 7) == 0) ? (v & 7) * (total >> 3) + (v >> 3) : v;
        Tile t;
        if (p.xcd_co_groups > 1) {
            // Weight-heavy layers (launcher: xcd_co_groups = G in {2, 4, 8}; total % 8 == 0, (Cout tiles x K slices) % G == 0,
            // pixel tiles % (8 / G) == 0): XCD x = v & 7 takes weight block x % G (a range of (K slice, Cout tile) pairs) of
            // pixel-tile block x / G instead of every weight tile of its pixel tiles, so an XCD's L2 fetches 1 / G of the
            // layer's weights (and its activations G times) instead of all of them: conv4_2 at 512^2 moves 9.4 MB of weights
            // and 8.4 MB of activations - 8 x 9.4 + 8.4 MB through the fabric with G = 1, 2 x 9.4 + 4 x 8.4 with G = 4.
            const int G = p.xcd_co_groups, x = v & 7, L = v >> 3;
            const int nw = n_co_tiles * ksplit, wpg = nw / G, w_local = L % wpg, pl = L / wpg;
            const int ppg = (total >> 3) / wpg;                    // pixel tiles per pixel-tile block
            const int wt = (x % G) * wpg + w_local;
            t.co0 = (wt % n_co_tiles) * C::TCO;
            t.kslice = wt / n_co_tiles;
            bid = (x / G) * ppg + pl;
        } else {
            t.co0 = (bid % n_co_tiles) * C::TCO;
            bid /= n_co_tiles;
            t.kslice = bid % ksplit;
            bid /= ksplit;
        }
        t.x0 = (bid % tiles_x) * TW;
        t.y0 = Y0 + (bid / tiles_x) * C::TH;
        if (p.row_skip_len != 0 && t.y0 >= p.row_skip_begin) t.y0 += p.row_skip_len;
        return t;
    };

    const unsigned char* wsplit = static_cast<const unsigned char*>(p.wgt_split);
    const size_t w_plane_stride = (size_t)9 * (p.cin / SK) * p.cout * 32;       // bytes per plane
    const size_t w_tap_stride = (size_t)(p.cin / SK) * p.cout * 32;
    const int ea = scale_exp(amax_with_halo(amax_read(p.amax_word), p.halo_bound_up, p.halo_bound_down));
    const int ew = scale_exp(*reinterpret_cast<const unsigned int*>(wsplit + 2 * w_plane_stride));
    const float in_scale = pow2f(ea), out_scale_a = pow2f(-ea), out_scale_w = pow2f(-ew);

    // tune bit 32 (ST_CONV_PHASES=1, tools/conv_bench.py): s_memtime sums of wave 0 (consumer) and the first producer wave
    // -> p.scratch[blockIdx.x][8]: {consumer MFMA, consumer barrier wait, consumer epilogue, producer staging,
    // producer barrier wait, whole, 1}
    const bool stamp = (p.tune & 32) != 0 && ksplit == 1 && p.scratch != nullptr;
    unsigned long long t_a = 0, t_b = 0, t_c = 0, t_prev = 0, t_begin = 0, r_begin = 0;
    if (stamp) {
        t_begin = t_prev = __builtin_amdgcn_s_memtime();
        r_begin = __builtin_amdgcn_s_memrealtime();         // 100 MHz: ticks / this = the shader clock the kernel held
    }
    auto mark = [&](unsigned long long& bucket) __attribute__((always_inline)) {
        if (stamp) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            bucket += t - t_prev;
            t_prev = t;
        }
    };

    if (producer) {
        // the producer waves are the youngest on their SIMDs and lose every issue arbitration against the MFMA
        // streams; in the XL tile they are the critical path of a chunk period and get static priority (+1.3 % there;
        // neutral to slightly negative for the smaller tiles; tune bit 128: off)
        if (C::XL && !(p.tune & 128)) __builtin_amdgcn_s_setprio(3);
        // ---- load cursor: the tile / chunk whose global loads are issued next ----
        int goff[C::NIT], aoff[C::NIT];
        // strip sharding: patch rows -1 / H come from the neighbours' halo block [2][Cin][W] (second resource).  Only
        // the waves of a tile that touches the strip's first / last row have such items; the others skip the loads.
        int hoff[HALO ? C::NIT : 1];
        bool halo_tile = false;                            // this wave has halo items in the load cursor's tile
        // ... and had them when the set's chunk was loaded - one flag per HALF of the item list: the XL tile pipelines the
        // halves against each other (store A, load A', store B, load B'), so when "store B" of a tile's last chunk runs,
        // half A of the register set already belongs to the NEXT tile, which may differ in whether it touches the strip's
        // first / last row.  (A single flag per set - rounds 1 / 2 - dropped or added halo rows in the second half of the
        // last chunk of such tiles: wrong results at every strip seam once a workgroup walks through more than one XL
        // tile, i.e. on images of >= 1024 rows; found by tests/test_large_strips_gpu.py.)
        bool set_halo[C::SETS][2] = {};
        int l_tile = 0, l_chunk = 0, l_chunk0 = 0, l_co0 = 0;
        auto point_at_tile = [&](int ordinal) __attribute__((always_inline)) {
            const Tile t = tile_of(ordinal);
            l_chunk0 = t.kslice * nchunks;
            l_co0 = t.co0;
            bool any_halo = false;
            // (lanes past the end of an item list redo the last item: no exec-mask branches in the staging)
#pragma unroll
            for (int i = 0; i < C::NIT; ++i) {
                const int it = (ptid + i * C::PT < 2 * C::NPX) ? ptid + i * C::PT : 2 * C::NPX - 1;
                const int g = it / C::NPX, q = it % C::NPX;
                const int y = t.y0 - 1 + q / C::LW, x = t.x0 - 1 + q % C::LW;
                const bool ok = y >= 0 && y < H && x >= 0 && x < W;
                goff[i] = ok ? (8 * g * HW + y * W + x) * 4 : kOOR;
                aoff[i] = q * 32 + ((g ^ ((q >> 3) & 1)) * 16);
                if constexpr (HALO) {
                    const bool xin = x >= 0 && x < W;
                    const bool top = xin && y == -1 && p.has_up, bot = xin && y == H && p.has_down;
                    hoff[i] = top ? (8 * g * W + x) * 4 : (bot ? ((p.cin + 8 * g) * W + x) * 4 : kOOR);
                    any_halo = any_halo || top || bot;
                }
            }
            if constexpr (HALO) halo_tile = __builtin_amdgcn_ballot_w64(any_halo) != 0;
        };
        // two register sets: the loads of chunks g + 2 and g + 3 are in flight while chunk g is multiplied (one
        // chunk period does not cover the global-load latency of a fully loaded chip)
        float ract[C::SETS][C::NIT][8];
        float rhal[HALO ? C::SETS : 1][HALO ? C::NIT : 1][8];
        f32x4 rwt[C::SETS][2][C::NWT];
        const int chunk_bytes = SK * HW * 4;
        int loaded = 0;                                    // chunks whose loads have been issued
        // PART 0 / 1: the two halves of a chunk's staging work (activation items [0, NIT/2) + weight plane 0, the rest
        // + plane 1), PART 2: all of it.  With a single register set the halves are software-pipelined against each
        // other (store A, load A', store B, load B'), so every load has half a staging period in flight before it is
        // needed instead of none.
        constexpr int NA = C::NIT / 2;
        // ablation (tools/conv_bench.py, ST_CONV_TUNE bit 256): the producers skip loads, conversion and LDS writes and
        // only keep the barrier protocol - the consumers' own chunk period (wrong results; profiles/r02_conv_xl_ablation.md)
        const bool ablate_prod = (p.tune & 256) != 0;
        // finer: bit 1024 = no activation loads (conversion + LDS writes of stale registers stay), bit 2048 = loads only
        // (no conversion, no LDS writes)
        const bool ablate_loads = (p.tune & 1024) != 0, ablate_stores = (p.tune & 2048) != 0;
        auto load_part = [&](auto SET, auto PART, auto WITHW) __attribute__((always_inline)) {
            constexpr int st = decltype(SET)::value, part = decltype(PART)::value;
            constexpr bool withw = decltype(WITHW)::value;
            if (ablate_prod || ablate_loads) return;
            const int cc = l_chunk0 + l_chunk;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.in) + (size_t)cc * SK * HW, 0, chunk_bytes, 0x00020000);
            sfor<(part == 1 ? NA : 0), (part == 0 ? NA : C::NIT)>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
#pragma unroll
                for (int c = 0; c < 8; ++c) ract[st][i][c] = bload(rs, goff[i], c * HW * 4);
            });
            if constexpr (HALO) {
                // (the cursor advances after the last part: halo_tile is this chunk's tile for both halves)
                if constexpr (part != 1) set_halo[st][0] = halo_tile;
                if constexpr (part != 0) set_halo[st][1] = halo_tile;
                if (halo_tile) {
                    const __amdgpu_buffer_rsrc_t hs = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<float*>(p.in_halo) + (size_t)cc * SK * W, 0, (p.cin + SK) * W * 4, 0x00020000);
                    sfor<(part == 1 ? NA : 0), (part == 0 ? NA : C::NIT)>([&](auto I) __attribute__((always_inline)) {
                        constexpr int i = decltype(I)::value;
#pragma unroll
                        for (int c = 0; c < 8; ++c) rhal[st][i][c] = bload(hs, hoff[i], c * W * 4);
                    });
                }
            }
            sfor<(part == 1 ? 1 : 0), (!withw ? 0 : part == 0 ? 1 : 2)>([&](auto PL) __attribute__((always_inline)) {
                constexpr int pl = decltype(PL)::value;
                sfor<0, C::NWT>([&](auto I) __attribute__((always_inline)) {
                    constexpr int i = decltype(I)::value;
                    const int f = (ptid + i * C::PT < C::NWP) ? ptid + i * C::PT : C::NWP - 1;
                    const int tap = f / (C::TCO * 2), r = f % (C::TCO * 2);
                    rwt[st][pl][i] = *reinterpret_cast<const f32x4*>(wsplit + pl * w_plane_stride + tap * w_tap_stride +
                                                                     ((size_t)cc * p.cout + l_co0) * 32 + r * 16);
                });
            });
        };
        auto advance = [&]() __attribute__((always_inline)) {      // the addresses were consumed at issue
            if (++l_chunk == nchunks) {
                l_chunk = 0;
                if (++l_tile < my_tiles) point_at_tile(l_tile);
            }
        };
        constexpr std::integral_constant<int, 0> PA{};
        constexpr std::integral_constant<int, 1> PB{};
        constexpr std::integral_constant<int, 2> PALL{};
        constexpr std::integral_constant<bool, !C::WDMA> WL{};  // weights staged by the producers (else: consumer DMA)
        constexpr std::integral_constant<bool, !C::WDMA> WW{};
        auto load_next = [&](auto SET, auto WITHW) __attribute__((always_inline)) {
            if (loaded >= gtot) return;
            ++loaded;
            load_part(SET, PALL, WITHW);
            advance();
        };
        auto store_part = [&](auto SET, auto PART, unsigned char* buf, auto WITHW) __attribute__((always_inline)) {
            constexpr int st = decltype(SET)::value, part = decltype(PART)::value;
            constexpr bool withw = decltype(WITHW)::value;
            if (ablate_prod || ablate_stores) return;
            sfor<(part == 1 ? NA : 0), (part == 0 ? NA : C::NIT)>([&](auto I) __attribute__((always_inline)) {
                constexpr int i = decltype(I)::value;
                f16x8 h0, h1;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float v = ract[st][i][c];
                    if constexpr (HALO) {
                        if (set_halo[st][i < NA ? 0 : 1]) v += rhal[st][i][c];        // neighbour rows; 0 elsewhere
                    }
                    v *= in_scale;
                    const _Float16 a = (_Float16)v;
                    h0[c] = a;
                    h1[c] = (_Float16)(v - (float)a);
                }
                *reinterpret_cast<f16x8*>(buf + aoff[i]) = h0;
                *reinterpret_cast<f16x8*>(buf + C::ACT_PLANE + aoff[i]) = h1;
            });
            sfor<(part == 1 ? 1 : 0), (!withw ? 0 : part == 0 ? 1 : 2)>([&](auto PL) __attribute__((always_inline)) {
                constexpr int pl = decltype(PL)::value;
                sfor<0, C::NWT>([&](auto I) __attribute__((always_inline)) {
                    constexpr int i = decltype(I)::value;
                    const int f = (ptid + i * C::PT < C::NWP) ? ptid + i * C::PT : C::NWP - 1;
                    const int row = f >> 1, hsel = f & 1;                // row = tap * 64 + co
                    *reinterpret_cast<f32x4*>(buf + C::W_OFF + pl * C::W_PLANE + row * 32 +
                                              ((hsel ^ ((row >> 3) & 1)) * 16)) = rwt[st][pl][i];
                });
            });
        };
        auto store_chunk = [&](auto SET, unsigned char* buf, auto WITHW) __attribute__((always_inline)) {
            store_part(SET, PALL, buf, WITHW);
        };
        constexpr std::integral_constant<int, 0> S0{};
        constexpr std::integral_constant<int, C::SETS - 1> S1{};
        point_at_tile(0);
        load_next(S0, WW);                                 // chunk 0 (with its weights)
        store_chunk(S0, smem, WW);
        load_next(S0, WL);                                 // chunk 1
        if constexpr (C::SETS == 2) load_next(S1, WL);     // chunk 2
        mark(t_a);
        __syncthreads();                                   // image 0 complete
        mark(t_b);
        if constexpr (C::SETS == 2) {
            for (int g = 0; g < gtot; g += 2) {
                if (g + 1 < gtot) {
                    store_chunk(S0, smem + C::BUF, WL);    // chunk g + 1 (odd) -> image 1
                    load_next(S0, WL);                     // chunk g + 3
                }
                mark(t_a);
                __syncthreads();                           // image 1 complete, image 0 free
                mark(t_b);
                if (g + 1 >= gtot) break;
                if (g + 2 < gtot) {
                    store_chunk(S1, smem, WL);             // chunk g + 2 (even) -> image 0
                    load_next(S1, WL);                     // chunk g + 4
                }
                mark(t_a);
                __syncthreads();                           // image 0 complete, image 1 free
                mark(t_b);
            }
        } else {
            int left = nchunks;                            // chunks left in the tile the consumers are multiplying
            for (int g = 0; g < gtot; ++g) {
                if (g + 1 < gtot) {
                    unsigned char* img = smem + ((g + 1) & 1) * C::BUF;
                    const bool more = g + 2 < gtot;
                    store_part(S0, PA, img, WL);           // chunk g + 1, first half
                    if (more) load_part(S0, PA, WL);       // chunk g + 2, first half
                    store_part(S0, PB, img, WL);
                    if (more) {
                        load_part(S0, PB, WL);
                        advance();
                    }
                }
                mark(t_a);
                __syncthreads();                           // image (g + 1) & 1 complete, image g & 1 free ...
                if (--left == 0) {
                    left = nchunks;
                    if (C::XL && g + 1 < gtot) __syncthreads();       // ... after the consumers' epilogue used it
                }
                mark(t_b);
            }
        }
        if (stamp && ptid == 0) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.scratch) + (size_t)blockIdx.x * 8;
            dst[3] = t_a;
            dst[4] = t_b;
        }
        return;
    }

    // ---------------------------------------- consumers ----------------------------------------
    int a_off[2];                                          // operand addresses relative to an image
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int co = i * 32 + l31;
        a_off[i] = C::W_OFF + co * 32 + ((half ^ ((co >> 3) & 1)) * 16);
    }
    // XL (168 registers): only the block's base pixel is kept and the tap's swizzled offset is recomputed (4 VALU)
    int b_off[WN][C::XL ? 1 : 9];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int pix = (wn * WN + j) * 32 + l31;
        const int qb = (pix / TW) * C::LW + (pix % TW);
        if constexpr (C::XL) {
            b_off[j][0] = qb;
        } else {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int q = qb + (tap / 3) * C::LW + (tap % 3);
                b_off[j][tap] = q * 32 + ((half ^ ((q >> 3) & 1)) * 16);
            }
        }
    }
    f32x16 acc[2][WN];
    auto fetch_tap = [&](const unsigned char* buf, auto TAP, f16x8 (&av)[2][2], f16x8 (&bv)[WN][2])
                         __attribute__((always_inline)) {
        constexpr int tap = decltype(TAP)::value;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                av[i][pl] = *reinterpret_cast<const f16x8*>(buf + pl * C::W_PLANE + tap * (C::TCO * 32) + a_off[i]);
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                int off;
                if constexpr (C::XL) {
                    const int q = b_off[j][0] + (tap / 3) * C::LW + (tap % 3);
                    off = q * 32 + ((half ^ ((q >> 3) & 1)) * 16);
                } else {
                    off = b_off[j][tap];
                }
                bv[j][pl] = *reinterpret_cast<const f16x8*>(buf + pl * C::ACT_PLANE + off);
            }
        }
    };
    // cross terms first, the dominant a0*b0 last (the order of conv_split_kernel: bit-identical sums)
    auto mfma_tap = [&](const f16x8 (&av)[2][2], const f16x8 (&bv)[WN][2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i][0], bv[j][1], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i][1], bv[j][0], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i][0], bv[j][0], acc[i][j], 0, 0, 0);
    };

    const bool partial = ksplit > 1;
    const bool accumulate = p.accumulate != 0 && !partial;
    const bool relu = p.relu != 0 && !partial;
    const bool out_mask = p.out_mask != nullptr && !partial;
    const bool has_bias = p.bias != nullptr && !partial;
    constexpr int TP = WN * 32 + 8;                            // slab pitch: 4 rows apart = 32 banks apart
    // wave-private epilogue slab: behind the images, or (XL) inside the image the tile's last chunk was read from
    float* slab = reinterpret_cast<float*>(smem + 2 * C::BUF) + wn * (32 * TP + 64);
    float* bias_w = C::XL ? reinterpret_cast<float*>(smem + 2 * C::BUF) + wn * 64 : slab + 32 * TP;   // 64 bias values
    unsigned int amax = 0;
    int g = 0;
    // XL: the consumers wait ~half of every chunk period for the 4 producer waves, so they stage the weights
    // themselves: 16-byte LDS-DMA (global_load_lds_dwordx4: no registers, no ds_write pass) into the image the
    // producers are filling, issued before a chunk's MFMAs and retired by the barrier's fence.  One wave-instruction
    // = 64 pieces = 32 rows (half a tap) of one plane, so everything but the lane's (row, half) offset is
    // wave-uniform; the DMA writes base + 16 lane linearly, so the 16-byte-half swizzle goes on the SOURCE address.
    auto dma_weights = [&](const Tile& tl, int chunk, int image) __attribute__((always_inline)) {
        const size_t base = ((size_t)(tl.kslice * nchunks + chunk) * p.cout + tl.co0) * 32;
        unsigned char* wimg = smem + image * C::BUF + C::W_OFF;
        constexpr int NJ = 2 * C::NWP / 64;                              // 36 wave-instructions per chunk
        const int lane_off = (lane >> 1) * 32 + (((lane & 1) ^ ((lane >> 4) & 1)) * 16);
        sfor<0, (NJ + CW - 1) / CW>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = decltype(I)::value;
            const int j = i * CW + cwave;                                // (plane, tap, co half), wave-uniform
            if (j < NJ) {
                const int pl = j / 18, tap = (j % 18) >> 1, hf = j & 1;
                const unsigned char* src = wsplit + pl * w_plane_stride + tap * w_tap_stride + base + hf * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane_off),
                                                 (__attribute__((address_space(3))) void*)(wimg + j * 1024), 16, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);                           // one address live at a time (168 registers)
        });
    };
    // (Tried for the XL tile: the five pieces of a wave issued one by one between the taps, in the shadow of the MFMAs
    // just issued, instead of this burst at the start of the chunk while the matrix pipe is empty - 3 - 4 % SLOWER:
    // 1024^2 4.74 -> 4.93 ms, 2048^2 17.6 -> 18.1 ms for the 23 trunk convs.)
    if constexpr (C::WDMA) dma_weights(tile_of(0), 0, 0);
    __syncthreads();                                       // image 0 complete
    mark(t_b);
    for (int k = 0; k < my_tiles; ++k) {
        const Tile t = tile_of(k);
        const float bias_v = has_bias ? p.bias[t.co0 + lane] : 0.f;    // lands during the K loop
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int c = 0; c < nchunks; ++c, ++g) {
            const unsigned char* buf = smem + (g & 1) * C::BUF;
            if constexpr (C::WDMA) {
                if (g + 1 < gtot && !(p.tune & 4096)) {    // next chunk's weights, retired by this chunk's barrier
                                                           // (ablation bit 4096: no weight DMA)
                    const bool same = c + 1 < nchunks;
                    dma_weights(same ? t : tile_of(k + 1), same ? c + 1 : 0, (g + 1) & 1);
                }
            }
            if (p.tune & 512) {
                // ablation (ST_CONV_TUNE bit 512): no operand fetch, no MFMA - the producers' own chunk period
            } else if constexpr (C::XL) {
                // single operand set (168 registers per wave): the SIMD's other consumer wave covers the LDS latency
                f16x8 a0[2][2], b0[WN][2];
                sfor<0, 9>([&](auto T) __attribute__((always_inline)) {
                    fetch_tap(buf, T, a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_tap(a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                });
            } else {
            f16x8 a0[2][2], b0[WN][2], a1[2][2], b1[WN][2];
            fetch_tap(buf, std::integral_constant<int, 0>{}, a0, b0);
            sfor<0, 5>([&](auto T2) __attribute__((always_inline)) {
                constexpr int tap = 2 * decltype(T2)::value;
                if constexpr (tap + 1 < 9) fetch_tap(buf, std::integral_constant<int, tap + 1>{}, a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tap(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (tap + 1 < 9) {
                    if constexpr (tap + 2 < 9) fetch_tap(buf, std::integral_constant<int, tap + 2>{}, a0, b0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_tap(a1, b1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            }
            mark(t_a);
            __syncthreads();                               // image (g + 1) & 1 complete, image g & 1 free
            mark(t_b);
        }

        // ---- epilogue of tile k (as in conv_split_kernel; the producers are already staging the next tile) ----
        // The lane-derived offsets of the epilogue are computed from an opaque copy of the lane id, i.e. HERE: hoisted
        // above the K loop (they are loop invariants) they would have to be carried across it, and the XL tile has no
        // register for that.
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int l31_e = lane_e & 31, half_e = lane_e >> 5;
        if constexpr (C::XL) slab = reinterpret_cast<float*>(smem + ((g - 1) & 1) * C::BUF) + wn * (32 * TP);
        bias_w[lane_e] = bias_v;                             // (the wave barriers below order it before the reads)
        float* out_base = partial ? p.scratch + (size_t)t.kslice * p.cout * HW : p.out;
        const int PH = H >> 1, PW = W >> 1;                 // MaxPool2d(2) output (floor)
        const bool pool = (WN == 2 && TW == 32) && p.pool_out != nullptr && !partial;
        const bool coded = pool && p.pool_code != nullptr;      // argmax codes INSTEAD of the full-resolution map
        const bool vec_ok = (W % 4 == 0) &&
                            (((reinterpret_cast<uintptr_t>(out_base) | reinterpret_cast<uintptr_t>(p.out_mask)) & 15) == 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co_base = t.co0 + i * 32;
            const __amdgpu_buffer_rsrc_t os =
                __builtin_amdgcn_make_buffer_rsrc(out_base + (size_t)co_base * HW, 0, 32 * HW * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t ps = __builtin_amdgcn_make_buffer_rsrc(
                (pool ? p.pool_out : out_base) + (size_t)co_base * (PH * PW), 0, 32 * PH * PW * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t cs = __builtin_amdgcn_make_buffer_rsrc(
                coded ? p.pool_code + (size_t)co_base * (PH * PW) : reinterpret_cast<unsigned char*>(out_base), 0, 32 * PH * PW,
                0x00020000);
            // Two forms of the store loops, chosen per launch: WITHOUT read-modify-write streams (forward, split-K
            // partials) the loop holds no vector-memory wait at all, so a wave's stores leave back to back; WITH them
            // (data gradients: the ReLU mask of the tensor being written and / or the tap gradient it accumulates into) the
            // loads of row group q + 1 are issued BEFORE the store of group q.  gfx950 counts loads and stores in one
            // in-order counter (vmcnt): a load that follows a store can only be waited for together with that store, and
            // the first form of this epilogue - load, wait, store, load, ... under launch-uniform branches, for which the
            // compiler placed a vmcnt(0) in EVERY iteration, loads or not - paid one store round trip per 16-byte row
            // group: 14 000 cycles per 64 x 512 tile in every layer (s_memtime), 8 - 24 % of a launch.
            // An absent stream reads through a zero-sized buffer resource (hardware returns 0, no traffic).
            const __amdgpu_buffer_rsrc_t os_ld = __builtin_amdgcn_make_buffer_rsrc(
                out_base + (size_t)co_base * HW, 0, accumulate ? 32 * HW * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t ms_ld = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(out_mask ? p.out_mask : out_base) + (size_t)co_base * HW, 0, out_mask ? 32 * HW * 4 : 0, 0x00020000);
            auto store_rows = [&](auto LOADS) __attribute__((always_inline)) {
            constexpr bool loads = decltype(LOADS)::value;
            if (vec_ok) {
                // 16-byte path: the wave transposes its 32-channel x (32 WN)-pixel slab through LDS and moves whole
                // float4s along the image rows
                __builtin_amdgcn_wave_barrier();               // the previous half_e's reads are done (in-order LDS)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half_e;
                        slab[row * TP + j * 32 + l31_e] = acc[i][j][r] * out_scale_a * out_scale_w;
                    }
                __builtin_amdgcn_wave_barrier();
                // A lane keeps its 4 columns through the loop; row group q4 is the channel rows RG q4 + lane / (8 WN).  Every
                // address below is "lane constant + q4 x launch constant" - spelled out, because the compiler re-derived the
                // whole chain (signed divisions, two 32-bit multiplies, a 64-bit multiply-add) in every iteration, and with
                // two consumer waves per SIMD those ~45 VALU instructions per 16-byte store were most of the epilogue.
                constexpr int RG = 64 / (WN * 8);                  // channel rows per row group
                const unsigned lrow = (unsigned)lane_e / (WN * 8), px = ((unsigned)lane_e % (WN * 8)) * 4;
                const int pix = wn * WN * 32 + (int)px;
                const int y = t.y0 + pix / TW, x = t.x0 + pix % TW;
                const bool inb = (y < Y1) && (x < W);
                // (unsigned: an out-of-range lane stays out of range under "+ q4 step": 0x7FFFFFFF + 7 x 32 HW bytes < 2^32)
                const unsigned off0 = inb ? (unsigned)((int)lrow * HW + y * W + x) * 4u : 0x7FFFFFFFu;
                const unsigned step = (unsigned)(RG * HW) * 4u;
                const float* slab_rd = slab + lrow * TP + px;
                const float* bias_rd = bias_w + i * 32 + lrow;
                const bool pin = (px < 32) && (y + 1 < Y1) && (x < W);     // (Y1 <= H: a tile that overshoots this launch's rows writes no window of the next launch's)
                const unsigned poff0 = pin ? (unsigned)((int)lrow * (PH * PW) + (y >> 1) * PW + (x >> 1)) * 4u : 0x7FFFFFFFu;
                const unsigned pstep = (unsigned)(RG * PH * PW) * 4u;
                f32x4 o_next = {0.f, 0.f, 0.f, 0.f}, m_next = {0.f, 0.f, 0.f, 0.f};
                if constexpr (loads) {
                    o_next = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(os_ld, (int)off0, 0, 0));
                    m_next = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ms_ld, (int)off0, 0, 0));
                    // an out-of-range (dropped) store: the loop is entered with the same history of memory operations -
                    // loads, then a store - as its back edge, so the wait counts inside it need not assume the shorter one
                    const f32x4 nothing = {0.f, 0.f, 0.f, 0.f};
                    __builtin_amdgcn_raw_buffer_store_b128(
                        __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, nothing), os, 0x7FFFFFFF, 0, 0);
                }
#pragma unroll(C::XL ? 2 : 4 * WN)
                for (int q4 = 0; q4 < 4 * WN; ++q4) {
                    const int off = (int)(off0 + (unsigned)q4 * step);
                    f32x4 v = *reinterpret_cast<const f32x4*>(slab_rd + q4 * (RG * TP));
                    const float bv = bias_rd[q4 * RG];
                    const f32x4 o = o_next, m = m_next;
                    if constexpr (loads) {
                        // (unconditional - past the last group an out-of-range offset, which the hardware answers with 0 -
                        // so that o_next / m_next are plain double buffers: a conditional refill makes them merge points,
                        // whose register copies wait for the load before the store below has even been issued)
                        const int off1 = (q4 + 1 < 4 * WN) ? (int)(off0 + (unsigned)(q4 + 1) * step) : 0x7FFFFFFF;
                        o_next = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(os_ld, off1, 0, 0));
                        m_next = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ms_ld, off1, 0, 0));
                        __builtin_amdgcn_sched_barrier(0);          // (the scheduler must not sink them below this group's store)
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x_ = v[e] + bv;
                        if (relu) x_ = fmaxf(x_, 0.f);
                        if constexpr (loads) {
                            if (accumulate) x_ += o[e];
                            if (out_mask) x_ = (m[e] > 0.f) ? x_ : 0.f;
                        }
                        v[e] = x_;
                        amax = max(amax, inb ? abs_bits(x_) : 0u);
                    }
                    // (coded launches - pooled map + argmax codes instead of the full-resolution map - are forward launches;
                    // in the read-modify-write form the store must be unconditional or the wait counts above it merge two
                    // histories and fall back to waiting for the previous store)
                    if (loads || !coded)
                        __builtin_amdgcn_raw_buffer_store_b128(
                            __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), os, off, 0, 0);
                    if constexpr (WN == 2 && TW == 32 && !loads) {
                        // fused MaxPool2d(2): lanes l and l + 8 of a 16-lane_e row hold the same 4 columns of the two image
                        // rows of a wave's block, so a 2x2 window is two adjacent elements here and the same two in the
                        // partner lane_e (DPP row rotate by 8); the lane_e of the even row writes the two pooled values
                        if (pool) {
                            float m0 = fmaxf(v[0], v[1]), m1 = fmaxf(v[2], v[3]);
                            // position of the first maximum inside this lane's half of each window (a later element wins
                            // only if strictly greater), bit 0 = window 0, bit 1 = window 1
                            const int mine = (v[1] > v[0] ? 1 : 0) | (v[3] > v[2] ? 2 : 0);
                            const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m0), 0x128, 0xf, 0xf, false));
                            const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, m1), 0x128, 0xf, 0xf, false));
                            const int theirs = __builtin_amdgcn_update_dpp(0, mine, 0x128, 0xf, 0xf, false);
                            // (seen from the lane of the window's first row, the only one that stores: the second row wins
                            // only if strictly greater)
                            const int at0 = n0 > m0 ? 2 + (theirs & 1) : (mine & 1);
                            const int at1 = n1 > m1 ? 2 + ((theirs >> 1) & 1) : ((mine >> 1) & 1);
                            m0 = fmaxf(m0, n0);
                            m1 = fmaxf(m1, n1);
                            const unsigned poff = poff0 + (unsigned)q4 * pstep;
                            typedef float f32x2 __attribute__((ext_vector_type(2)));
                            f32x2 pv = {m0, m1};
                            __builtin_amdgcn_raw_buffer_store_b64(
                                __builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int, pv), ps, (int)poff, 0, 0);
                            if (coded) {
                                const int two = (at0 | (m0 > 0.f ? 4 : 0)) | ((at1 | (m1 > 0.f ? 4 : 0)) << 8);
                                __builtin_amdgcn_raw_buffer_store_b16((unsigned short)two, cs, pin ? (int)(poff >> 2) : 0x7FFFFFFF, 0, 0);
                            }
                        }
                    }
                }
            } else {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const int pix = (wn * WN + j) * 32 + l31_e;
                    const int y = t.y0 + pix / TW, x = t.x0 + pix % TW;
                    const bool inb = (y < Y1) && (x < W);
                    const int pix_bytes = inb ? (y * W + x) * 4 : 0x7FFFFFFF;
                    auto offset_r = [&](int r) __attribute__((always_inline)) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half_e;
                        return inb ? row * HW * 4 + pix_bytes : 0x7FFFFFFF;
                    };
                    float o_next = 0.f, m_next = 0.f;
                    if constexpr (loads) {
                        o_next = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(os_ld, offset_r(0), 0, 0));
                        m_next = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ms_ld, offset_r(0), 0, 0));
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half_e;
                        const int off = offset_r(r);
                        const float o = o_next, mk = m_next;
                        if constexpr (loads) {
                            const int off1 = (r + 1 < 16) ? offset_r(r + 1) : 0x7FFFFFFF;
                            o_next = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(os_ld, off1, 0, 0));
                            m_next = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ms_ld, off1, 0, 0));
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        float v = acc[i][j][r] * out_scale_a * out_scale_w;
                        v += bias_w[i * 32 + row];
                        if (relu) v = fmaxf(v, 0.f);
                        if constexpr (loads) {
                            if (accumulate) v += o;
                            if (out_mask) v = (mk > 0.f) ? v : 0.f;
                        }
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), os, off, 0, 0);
                        amax = max(amax, inb ? abs_bits(v) : 0u);
                    }
                }
            }
            };
            if (accumulate || out_mask) store_rows(std::true_type{});
            else store_rows(std::false_type{});
        }
        if (C::XL && k + 1 < my_tiles) __syncthreads();    // the producers may refill the slab image
        mark(t_c);                                          // stores issued (not drained)
    }
    if (p.out_amax && !partial) amax_commit_lean(amax, p.out_amax);
    if (stamp && cwave == 0 && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        mark(t_c);                                          // drain of the last tile's stores
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(p.scratch) + (size_t)blockIdx.x * 8;
        dst[0] = t_a;
        dst[1] = t_b;
        dst[2] = t_c;
        dst[5] = __builtin_amdgcn_s_memtime() - t_begin;
        dst[7] = __builtin_amdgcn_s_memrealtime() - r_begin;
        dst[6] = 1;
    }
}

inline int ceil_div_i(int a, int b) { return (a + b - 1) / b; }

template <int TW, int WN, int CW, bool HALO>
int launch_pc_cfg_h(const ConvProblem& p, int ksplit, hipStream_t stream) {
    using C = PCfg<TW, WN, CW>;
    constexpr int LDS = 2 * C::BUF + (C::XL ? CW * 64 : CW * (32 * (WN * 32 + 8) + 64)) * 4;
    static_assert(!C::XL || CW * 32 * (WN * 32 + 8) * 4 <= C::BUF, "XL epilogue slabs must fit one image");       // two images + the consumers' epilogue slabs
    static_assert(LDS <= 160 * 1024, "LDS budget of one CU");
    static bool attr_set = false;
    static int n_cu = 256;
    auto kern = conv_pc_kernel<TW, WN, CW, HALO>;
    if (!attr_set) {
        ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8)
            n_cu = prop.multiProcessorCount & ~7;          // a multiple of 8 keeps a workgroup's tiles on one XCD
        attr_set = true;
    }
    const int rows = (p.row_end ? p.row_end : p.height) - p.row_begin - p.row_skip_len;
    ST_REQUIRE(p.row_skip_len == 0 || (p.row_skip_begin - p.row_begin) % C::TH == 0,
               "conv (producer/consumer): the skipped row range must start on a tile row");
    const int tiles_x = ceil_div_i(p.width, TW), tiles_y = ceil_div_i(rows, C::TH);
    const int n_co_tiles = p.cout / C::TCO;
    const long long total = (long long)tiles_x * tiles_y * n_co_tiles * ksplit;
    ST_REQUIRE(total > 0 && total < (1ll << 30), "conv grid out of range");
    // Strip plans, interior launch of a convolution whose halo is in flight (overlap_part == 1): RCCL's point-to-point
    // kernel needs a CU too, and a persistent workgroup per CU leaves it none - the exchange then starts only when this
    // launch has finished, i.e. nothing overlaps (measured over the real transport on one GPU, tools/fabric_host_time.py:
    // +1.5 ms per iteration on a 2896 x 272 strip whatever the overlap mode).  ST_STRIP_SPARE_CUS CUs (a multiple of 8: one
    // per XCD) are left free for it.
    static Option spare_opt("ST_STRIP_SPARE_CUS", 0);
    int cus = n_cu;
    if (p.overlap_part == 1 && spare_opt.get() > 0 && n_cu - spare_opt.get() >= 64) cus = (n_cu - spare_opt.get()) & ~7;
    const int grid = total <= cus ? (int)total : cus;      // one persistent 8-wave workgroup per CU
    // XCD <-> tile mapping: by default an XCD walks through ALL Cout tiles of its pixel tiles (its L2 holds the pixel tile's
    // activations once and every XCD fetches the whole layer's weights).  Where the weights outweigh the activations the
    // Cout tiles are dealt to groups of XCDs instead (tile_of): G minimises weights x 8 / G + activations x G.
    // ST_CONV_XCD_COGROUPS: 0 (default) the model, 1 never, 2 / 4 / 8 forced where the shape allows.
    ConvProblem q = p;
    q.xcd_co_groups = 1;
    {
        static Option cg_opt("ST_CONV_XCD_COGROUPS", 0);
        const long long px_tiles = (long long)tiles_x * tiles_y;
        if ((total & 7) == 0 && grid == n_cu && cus == n_cu && n_cu % 8 == 0 && cg_opt.get() != 1) {
            const double wbytes = 9.0 * p.cin * p.cout * 4.0, abytes = (double)p.cin * rows * p.width * 4.0;
            double best = wbytes * 8.0 + abytes;
            for (int G = 2; G <= 8; G *= 2) {
                if ((n_co_tiles * ksplit) % G != 0 || px_tiles % (8 / G) != 0) continue;
                const double cost = wbytes * 8.0 / G + abytes * G;
                if (cg_opt.get() == G || (cg_opt.get() == 0 && cost < 0.9 * best)) {
                    best = cost;
                    q.xcd_co_groups = G;
                }
            }
        }
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::THREADS), LDS, stream, q, tiles_x, n_co_tiles, ksplit,
                       p.cin / SK / ksplit, (int)total);
    ST_LAUNCH_CHECK();
    if (ksplit > 1) {
        ST_REQUIRE(p.row_begin == 0 && p.row_end == 0, "conv (producer/consumer): K split needs the whole image");
        return launch_conv_splitk_reduce(p, ksplit, stream);
    }
    return 0;
}

template <int TW, int WN, int CW>
int launch_pc_cfg(const ConvProblem& p, int ksplit, hipStream_t stream) {
    if (p.in_halo) return launch_pc_cfg_h<TW, WN, CW, true>(p, ksplit, stream);
    return launch_pc_cfg_h<TW, WN, CW, false>(p, ksplit, stream);
}

template <int WN, int CW>
int launch_pc_tw(const ConvProblem& p, int ksplit, hipStream_t s, int tw) {
    if (tw == 32) return launch_pc_cfg<32, WN, CW>(p, ksplit, s);
    if (tw == 16) return launch_pc_cfg<16, WN, CW>(p, ksplit, s);
    return launch_pc_cfg<8, WN, CW>(p, ksplit, s);
}

}  // namespace

bool conv_pc_applies(const ConvProblem& p) {
    static Option halo_ok("ST_CONV_PC_HALO", 1);     // A/B knob
    return p.taps == 9 && p.planes == 2 && p.elem == 1 && p.wgt_split && p.amax_word && !p.mask &&
           (!p.in_halo || halo_ok.get()) && p.cin % SK == 0 && p.cout % 64 == 0;
}

namespace {
// ST_CONV_PC_XL=0 keeps the 64co x 512px tile off (A/B runs)
bool xl_tile_pays(const ConvProblem& p) {
    static Option xl_opt("ST_CONV_PC_XL", 1);
    const int xl = xl_opt.get();
    // (32-wide tiles only: the 16- and 8-wide XL variants are 1-3 registers over the 168 budget)
    const long long tiles = (long long)ceil_div_i(p.width, 32) * ceil_div_i(p.height, 16) * (p.cout / 64);
    return xl && p.cin >= 64 && tiles >= 256;
}
}  // namespace

// Where this form beats conv_split_kernel.  Round 1 measured it per layer with the tile rule of that time (XL when
// >= 256 tiles, else by workgroup count) and kept shallow layers (Cin < 256) with few tiles on the single-role kernel.
// With the tile shape chosen by the cost model below it wins or ties on every trunk layer from 181^2 up (23 convs:
// 181^2 636 -> 504 us, 256^2 658 -> 574, 362^2 1217 -> 1131, 512^2 1474 -> 1316, 1024^2 5481 -> 4851) and loses 2-6 %
// only on the 8 x 8 ... 16 x 16 pixel layers of the 128^2 scale (388 vs 386 us in total): it takes every problem
// it applies to.  ST_CONV_PC_MODEL=0 restores the round-1 rules (A/B runs and the bit-identity test).
bool conv_pc_preferred(const ConvProblem& p) {
    if (!conv_pc_applies(p)) return false;
    static Option model_opt("ST_CONV_PC_MODEL", 1);
    if (model_opt.get()) return true;
    const long long pixels = (long long)p.height * p.width;
    const long long wg_a = ((pixels + 255) / 256) * (p.cout / 64);
    return xl_tile_pays(p) || (p.cin >= 256 && wg_a < 512);
}

// Tile choice.  One persistent workgroup per CU walks through ceil(tiles / CUs) tiles, so the time of a shape is
//     launch + rounds x (chunks per tile x chunk period + per-tile overhead)  [+ the split-K reduce pass],
// and a layer whose tile count is just above a multiple of the CU count pays a whole extra round (a 362^2 image has
// 276 XL tiles of conv1_2: 2 rounds for 1.08 rounds of work; with the round-1 rule "XL whenever >= 256 tiles" the
// 362^2 scale ran slower than the 512^2 one).  Candidates: the XL tile (32 wide only), the 256- and 128-pixel tiles
// in their three widths, K split in 1 ... 16.  The constants (microseconds) are a least-squares fit to 1809 timed
// (layer, size, direction, shape) points of tools/conv_shapes.py, 128^2 ... 1024^2 (profiles/r02_conv_tile_choice.md:
// rms error 10 %; choosing by the model loses <= 1 % against the best forced shape of every layer, and gains 1.5 %
// (512^2) ... 17 % (181^2) on the 23 trunk convolutions against the round-1 rule).  Near-ties go to the wider tile
// (less halo to stage).
namespace {
struct PcChoice { int shape, tw, ksplit; double cost; int split_row, shape2, tw2; };

constexpr double kPcLaunch = 2.31;
constexpr double kPcChunk[4] = {0.0, 3.972, 2.607, 1.706};     // per chunk of 16 input channels and round
constexpr double kPcRound[4] = {0.0, 13.369, 8.402, 4.379};    // prologue + epilogue + tile hand-over per round
constexpr double kPcReduce0 = 2.995, kPcReduce1 = 1.163e-6;    // reduce pass: launch + per float of partials
constexpr int kPcPix[4] = {0, 512, 256, 128};

// best single launch over rows [0, rows) of the problem (ksplit only when it covers the whole image)
PcChoice choose_pc_single(const ConvProblem& p, int rows, bool allow_ksplit, int n_cu) {
    const int nchunks = p.cin / SK, co_tiles = p.cout / 64;
    const long long pixels = (long long)p.height * p.width;
    PcChoice best{0, 32, 1, 1e30, 0, 0, 0};
    for (int shape = 1; shape <= 3; ++shape) {
        for (int tw : {32, 16, 8}) {
            if (shape == 1 && tw != 32) continue;           // narrower XL variants are over the register budget
            const int th = kPcPix[shape] / tw;
            const long long tiles = (long long)ceil_div_i(p.width, tw) * ceil_div_i(rows, th) * co_tiles;
            for (int ks = 1; ks <= 16; ks *= 2) {
                if (ks > 1 && (!allow_ksplit || !p.scratch || shape == 1 || nchunks % ks != 0 || nchunks / ks < 2 ||
                               (size_t)ks * p.cout * pixels > kConvScratchFloats))
                    continue;
                const long long rounds = (tiles * ks + n_cu - 1) / n_cu;
                double cost = kPcLaunch + (double)rounds * ((double)(nchunks / ks) * kPcChunk[shape] + kPcRound[shape]);
                // (the reduce pass is a DEPENDENT launch on the trunk's stream: in the iteration it costs ~4 us more than in the
                // isolated per-layer timings the constants were fitted to - charged where a layer has the alternative of a
                // smaller unsplit tile: the 512-channel layers from 4096 pixels (conv4_x at 512^2); ST_CONV_PC_REDUCE_US / ST_CONV_PC_REDUCE_PIXELS: experiments)
                static Option reduce_pen("ST_CONV_PC_REDUCE_US", 4);
                static Option reduce_px("ST_CONV_PC_REDUCE_PIXELS", 4096);
                if (ks > 1)
                    cost += kPcReduce0 + ((pixels >= reduce_px.get() && nchunks >= 32) ? reduce_pen.get() : 0) +
                            kPcReduce1 * (double)ks * (double)p.cout * (double)pixels;
                if (cost < 0.97 * best.cost) best = PcChoice{shape, tw, ks, cost, 0, 0, 0};
            }
        }
    }
    return best;
}

PcChoice choose_pc_tile_uncached(const ConvProblem& p, int n_cu);

// The choice depends on the problem's shape, on whether a split-K workspace exists and on the library's switches - not on
// the pointers.  The fuse-pool question, the launcher and the strip plans' overlap decision each ask for it, for the 23
// trunk convolutions of every step (~250 candidates per call: tens of microseconds of host time per launch on the
// launch-bound 128^2 ... 256^2 scales, advisor finding of round 2): remembered per (shape, workspace, switch generation).
PcChoice choose_pc_tile(const ConvProblem& p, int n_cu) {
    struct Entry { int height, width, cin, cout, scratch, n_cu; unsigned gen; PcChoice choice; };
    static Entry cache[256];
    static std::mutex guard;
    const unsigned gen = option_generation();
    const int has_scratch = p.scratch != nullptr;
    const unsigned long long h = (unsigned long long)(unsigned)p.height * 0x9E3779B97F4A7C15ull ^
                                 (unsigned long long)(unsigned)p.width * 0xC2B2AE3D27D4EB4Full ^
                                 (unsigned long long)(unsigned)(p.cin * 1024 + p.cout) * 0x165667B19E3779F9ull ^ (unsigned)has_scratch;
    const size_t slot = (size_t)(h >> 56);
    auto matches = [&](const Entry& e) {
        return e.choice.shape != 0 && e.gen == gen && e.height == p.height && e.width == p.width && e.cin == p.cin &&
               e.cout == p.cout && e.scratch == has_scratch && e.n_cu == n_cu;
    };
    {
        std::lock_guard<std::mutex> lock(guard);
        if (matches(cache[slot])) return cache[slot].choice;
    }
    const PcChoice c = choose_pc_tile_uncached(p, n_cu);
    std::lock_guard<std::mutex> lock(guard);
    cache[slot] = Entry{p.height, p.width, p.cin, p.cout, has_scratch, n_cu, gen, c};
    return c;
}

PcChoice choose_pc_tile_uncached(const ConvProblem& p, int n_cu) {
    const int nchunks = p.cin / SK, co_tiles = p.cout / 64;
    PcChoice best = choose_pc_single(p, p.height, true, n_cu);
    // Two launches: the first rows with a large tile in WHOLE rounds, the remaining rows with whatever tile suits
    // them (usually a small one that fits one short round) - instead of a last round that is nearly empty.
    static Option split_opt("ST_CONV_PC_SPLIT", 1);
    if (!split_opt.get()) return best;
    for (int shape = 1; shape <= 2; ++shape) {
        for (int tw : {32, 16, 8}) {
            if (shape == 1 && tw != 32) continue;
            const int th = kPcPix[shape] / tw;
            const int tiles_y = ceil_div_i(p.height, th);
            const long long per_row = (long long)ceil_div_i(p.width, tw) * co_tiles;
            const long long total = per_row * tiles_y;
            const long long full_rounds = total / n_cu;
            if (full_rounds < 1 || total % n_cu == 0) continue;
            const int rows_a = (int)((full_rounds * n_cu) / per_row);        // tile rows of the first launch
            if (rows_a < 1 || rows_a >= tiles_y) continue;
            const long long rounds_a = (per_row * rows_a + n_cu - 1) / n_cu;
            const double cost_a = kPcLaunch + (double)rounds_a * ((double)nchunks * kPcChunk[shape] + kPcRound[shape]);
            const PcChoice rest = choose_pc_single(p, p.height - rows_a * th, false, n_cu);
            const double cost = cost_a + rest.cost;
            if (cost < 0.95 * best.cost) best = PcChoice{shape, tw, 1, cost, rows_a * th, rest.shape, rest.tw};
        }
    }
    return best;
}
}  // namespace

namespace {
int pc_n_cu() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount >= 8)
            n_cu = prop.multiProcessorCount & ~7;
    }
    return n_cu;
}
}  // namespace

// Will launch_conv(p) write p.pool_out?  Only the tiles in which one consumer wave owns whole 2x2 windows (32-wide, two
// 32-pixel blocks per wave: the XL and the 256-pixel tile) on the 16-byte store path, no K split, the shipped tile
// choice (no forced shapes) - the same deterministic decision launch_conv_pc takes.
bool conv_pc_fuses_pool(const ConvProblem& p) {
    static Option fuse_opt("ST_CONV_POOL_FUSE", 1);
    // (the fat kernel - st_conv_fat.hip, taken by launch_conv_split where conv_fat_preferred() says so - writes the pooled map and
    // the argmax codes in its own epilogue: every tile holds whole 2 x 2 windows)
    if (fuse_opt.get() && p.pool_out && !p.mask && !p.accumulate && !p.out_mask && conv_fat_preferred(p)) return true;
    static Option shape_opt("ST_CONV_PC_SHAPE", 0);
    static Option model_opt("ST_CONV_PC_MODEL", 1);
    static Option use_pc_opt("ST_CONV_PC", 1);
    if (!fuse_opt.get() || shape_opt.get() || !model_opt.get() || use_pc_opt.get() != 1) return false;
    if (!p.pool_out || p.mask || p.accumulate || p.out_mask || !conv_pc_applies(p)) return false;
    if (p.width % 4 != 0 || ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.pool_out)) & 15) != 0) return false;
    const PcChoice c = choose_pc_tile(p, pc_n_cu());
    auto ok = [](int shape, int tw) { return (shape == 1 || shape == 2) && tw == 32; };
    if (c.ksplit != 1 || !ok(c.shape, c.tw)) return false;
    return c.split_row == 0 || ok(c.shape2, c.tw2);
}

// ---- strip plans: interior + boundary launches (ConvProblem::overlap_part) -------------------------------------------
// The boundary launch covers one tile row at the top of the strip (tile height b in {4, 8, 16, 32}) and 1 ... 8 tile
// rows at its bottom, the interior launch the rows between; both tiles and the bottom row count are chosen to minimise
// the summed cost-model time (extra bottom rows let the interior launch end on a whole round of tiles).  kPcExchangeUs is what the
// split is allowed to cost: the latency of one neighbour exchange (pack kernel + RCCL send / recv of <= 741 KB over
// xGMI + event hand-over) that would otherwise sit between two convolutions - an estimate, the transport has never
// been timed on hardware (ST_STRIP_OVERLAP_US overrides it; ST_STRIP_OVERLAP=0 never splits, =2 splits whenever the
// kernel can).
constexpr double kPcExchangeUs = 25.0;

bool conv_pc_overlap_choice(const ConvProblem& p_in, PcOverlap* out) {
    ConvProblem p = p_in;
    p.overlap_part = 0;
    p.row_begin = p.row_end = p.row_skip_begin = p.row_skip_len = 0;
    static Option mode_opt("ST_STRIP_OVERLAP", 1);
    static Option us_opt("ST_STRIP_OVERLAP_US", (int)kPcExchangeUs);
    static Option use_pc_opt("ST_CONV_PC", 1);
    ConvProblem q = p;
    if (!q.in_halo) q.in_halo = q.in;           // (applies() only asks whether a halo block exists)
    if (!use_pc_opt.get() || !mode_opt.get() || !conv_pc_applies(q) || !conv_pc_preferred(q)) return false;
    const int n_cu = pc_n_cu();
    const int nchunks = p.cin / SK, co_tiles = p.cout / 64;
    // (an odd strip height: the last rows' tile row would start on an odd row and cut through the 2 x 2 windows)
    static Option fuse_opt("ST_CONV_POOL_FUSE", 1);      // (the same switches conv_pc_fuses_pool() honours: the A/B knobs
    static Option shape_opt("ST_CONV_PC_SHAPE", 0);      //  turn the fused pool off on split strip convolutions too)
    static Option model_opt("ST_CONV_PC_MODEL", 1);
    const bool want_pool = fuse_opt.get() && !shape_opt.get() && model_opt.get() && use_pc_opt.get() == 1 &&
                           p.pool_out != nullptr && p.width % 4 == 0 && p.height % 2 == 0 && !p.mask && !p.accumulate && !p.out_mask &&
                           ((reinterpret_cast<uintptr_t>(p.out) | reinterpret_cast<uintptr_t>(p.pool_out)) & 15) == 0;
    auto pool_tile = [](int shape, int tw) { return (shape == 1 || shape == 2) && tw == 32; };
    // (asked once when the phases are built and once per overlap launch: remembered like the tile choice)
    struct Entry { int height, width, cin, cout, pool, scratch, n_cu; unsigned gen; bool valid; PcOverlap choice; };
    static Entry cache[64];
    static std::mutex guard;
    const unsigned gen = option_generation();
    const size_t slot = (size_t)(((unsigned long long)(unsigned)p.height * 0x9E3779B97F4A7C15ull ^
                                  (unsigned long long)(unsigned)p.width * 0xC2B2AE3D27D4EB4Full ^
                                  (unsigned long long)(unsigned)(p.cin * 1024 + p.cout) * 0x165667B19E3779F9ull ^ (unsigned)want_pool) >> 58);
    {
        std::lock_guard<std::mutex> lock(guard);
        const Entry& e = cache[slot];
        if (e.valid && e.gen == gen && e.height == p.height && e.width == p.width && e.cin == p.cin && e.cout == p.cout &&
            e.pool == (int)want_pool && e.scratch == (int)(p.scratch != nullptr) && e.n_cu == n_cu) {
            *out = e.choice;
            return true;
        }
    }
    PcOverlap best{};
    best.cost_split = 1e30;
    for (int shape_b = 1; shape_b <= 3; ++shape_b) {
        for (int tw_b : {32, 16, 8}) {
            if (shape_b == 1 && tw_b != 32) continue;
            if (want_pool && !pool_tile(shape_b, tw_b)) continue;
            const int th_b = kPcPix[shape_b] / tw_b;
            // one tile row on top, k_bot tile rows at the bottom: the extra bottom rows let the interior launch end on
            // a whole round of its (usually larger) tile instead of a nearly empty last one
            for (int k_bot = 1; k_bot <= 8; ++k_bot) {
                const int rows_b = (1 + k_bot) * th_b;
                if (p.height < rows_b + 2 || k_bot * th_b > p.height / 2) break;
                const long long tiles_b = (long long)ceil_div_i(p.width, tw_b) * (1 + k_bot) * co_tiles;
                const long long rounds_b = (tiles_b + n_cu - 1) / n_cu;
                const double cost_b = kPcLaunch + (double)rounds_b * ((double)nchunks * kPcChunk[shape_b] + kPcRound[shape_b]);
                // interior: best single launch over the remaining rows (no K split: a row range has no reduce pass)
                PcChoice in_best{0, 32, 1, 1e30, 0, 0, 0};
                for (int shape = 1; shape <= 3; ++shape)
                    for (int tw : {32, 16, 8}) {
                        if (shape == 1 && tw != 32) continue;
                        if (want_pool && !pool_tile(shape, tw)) continue;
                        const int th = kPcPix[shape] / tw;
                        const long long tiles = (long long)ceil_div_i(p.width, tw) * ceil_div_i(p.height - rows_b, th) * co_tiles;
                        const long long rounds = (tiles + n_cu - 1) / n_cu;
                        const double cost = kPcLaunch + (double)rounds * ((double)nchunks * kPcChunk[shape] + kPcRound[shape]);
                        if (cost < 0.97 * in_best.cost) in_best = PcChoice{shape, tw, 1, cost, 0, 0, 0};
                    }
                if (in_best.shape == 0) continue;
                const double cost = cost_b + in_best.cost;
                if (cost < best.cost_split) {
                    best.rows_b = th_b;
                    best.rows_bottom = k_bot * th_b;
                    best.shape_i = in_best.shape; best.tw_i = in_best.tw;
                    best.shape_b = shape_b; best.tw_b = tw_b;
                    best.cost_split = cost;
                }
            }
        }
    }
    if (best.cost_split >= 1e30) return false;
    ConvProblem w = p;
    if (!want_pool) w.pool_out = nullptr;
    best.cost_whole = choose_pc_tile(w, n_cu).cost;
    best.pool = want_pool;
    const int mode = mode_opt.get();
    best.pays = mode == 2 || (mode == 1 && best.cost_split <= best.cost_whole + (double)us_opt.get());
    *out = best;
    std::lock_guard<std::mutex> lock(guard);
    cache[slot] = Entry{p.height, p.width, p.cin, p.cout, (int)want_pool, (int)(p.scratch != nullptr), n_cu, gen, true, best};
    return true;
}

namespace {
int launch_pc_shape(const ConvProblem& q, int shape, int tw, hipStream_t stream) {
    if (shape == 1) return launch_pc_cfg<32, 2, 8>(q, 1, stream);
    if (shape == 2) return launch_pc_tw<2, 4>(q, 1, stream, tw);
    return launch_pc_tw<1, 4>(q, 1, stream, tw);
}
}  // namespace

// The caller (launch_conv_split) has validated the problem and measured / folded the operand bound.
int launch_conv_pc(const ConvProblem& p, hipStream_t stream) {
    ST_REQUIRE(conv_pc_applies(p) || (p.overlap_part == 1 && !p.in_halo), "conv (producer/consumer): unsupported problem");
    if (p.overlap_part != 0) {
        PcOverlap o{};
        ST_REQUIRE(conv_pc_overlap_choice(p, &o), "conv (producer/consumer): this problem cannot be split for overlap");
        ConvProblem q = p;
        if (!o.pool) q.pool_out = nullptr;
        if (p.overlap_part == 1) {
            ST_REQUIRE(p.in_halo == nullptr, "conv interior launch must not read the halo block");
            q.row_begin = o.rows_b;
            q.row_end = p.height - o.rows_bottom;
            return launch_pc_shape(q, o.shape_i, o.tw_i, stream);
        }
        ST_REQUIRE(p.in_halo != nullptr, "conv boundary launch needs the halo block");
        q.row_begin = 0;
        q.row_end = p.height;
        q.row_skip_begin = o.rows_b;
        q.row_skip_len = p.height - o.rows_b - o.rows_bottom;
        return launch_pc_shape(q, o.shape_b, o.tw_b, stream);
    }
    if (p.pool_out && !conv_pc_fuses_pool(p)) {         // the caller runs the pool kernel: do not write half of it here
        ConvProblem q = p;
        q.pool_out = nullptr;
        q.pool_code = nullptr;
        return launch_conv_pc(q, stream);
    }
    static Option shape_opt("ST_CONV_PC_SHAPE", 0);     // experiment knobs: 1 XL / 2 256-pixel / 3 128-pixel tile,
    static Option tw_opt("ST_CONV_PC_TW", 0);           // tile width 32 / 16 / 8,
    static Option ks_opt("ST_CONV_PC_KSPLIT", 0);       // K split 1 / 2 / 4,
    static Option model_opt("ST_CONV_PC_MODEL", 1);     // 0: the round-1 rule (XL when >= 256 tiles, else by count)
    const int f_shape = shape_opt.get(), f_tw = tw_opt.get(), f_ks = ks_opt.get();
    if (f_shape || !model_opt.get()) {
        if (f_shape == 1 || (!f_shape && xl_tile_pays(p))) return launch_pc_cfg<32, 2, 8>(p, 1, stream);
        const long long pixels = (long long)p.height * p.width;
        const int co_tiles = p.cout / 64;
        const long long wg_a = ((pixels + 255) / 256) * co_tiles, wg_b = ((pixels + 127) / 128) * co_tiles;
        const bool big = f_shape ? f_shape == 2 : wg_a >= 256;
        long long wgs = big ? wg_a : wg_b;
        int ksplit = 1;
        if (f_ks) {
            ksplit = f_ks;
            ST_REQUIRE(ksplit == 1 || (p.scratch && (p.cin / SK) % ksplit == 0 &&
                                       (size_t)ksplit * p.cout * pixels <= kConvScratchFloats),
                       "ST_CONV_PC_KSPLIT=%d does not fit this problem", ksplit);
        } else if (p.scratch && !big) {
            const int nchunks = p.cin / SK;
            while (wgs * ksplit < 256 && nchunks % (ksplit * 2) == 0 && nchunks / (ksplit * 2) >= 2 &&
                   (size_t)(ksplit * 2) * p.cout * pixels <= kConvScratchFloats)
                ksplit *= 2;
        }
        int tw = f_tw;
        if (!tw) {                                       // least padded area
            const int npix = big ? 256 : 128;
            auto area = [&](int w) {
                return (long long)ceil_div_i(p.height, npix / w) * (npix / w) * (long long)ceil_div_i(p.width, w) * w;
            };
            tw = 32;
            for (int w : {16, 8})
                if (area(w) < area(tw)) tw = w;
        }
        return big ? launch_pc_tw<2, 4>(p, ksplit, stream, tw) : launch_pc_tw<1, 4>(p, ksplit, stream, tw);
    }
    const PcChoice c = choose_pc_tile(p, pc_n_cu());
    auto launch_shape = [&](const ConvProblem& q, int shape, int tw, int ks) -> int {
        if (shape == 1) return launch_pc_cfg<32, 2, 8>(q, 1, stream);
        if (shape == 2) return launch_pc_tw<2, 4>(q, ks, stream, tw);
        return launch_pc_tw<1, 4>(q, ks, stream, tw);
    };
    if (c.split_row == 0) return launch_shape(p, c.shape, c.tw, c.ksplit);
    ConvProblem top = p, rest = p;
    top.row_begin = 0;
    top.row_end = c.split_row;
    rest.row_begin = c.split_row;
    rest.row_end = p.height;
    if (launch_shape(top, c.shape, c.tw, 1)) return 1;
    return launch_shape(rest, c.shape2, c.tw2, 1);
}

}  // namespace st
