// Measurement aid, not on the hot path: what rate does the 16-bit matrix pipe SUSTAIN on this chip, alone and next to
// the LDS operand stream an LDS-fed tile needs?  The convolution roofline in DESIGN.md / bench.py is quoted against the
// nominal 2.5 PFLOP/s (2.4 GHz); the XL convolution tile runs at ~1.7 GHz under load.  This kernel
// separates "the matrix pipe's own power draw" from "what the tile adds": one persistent workgroup per CU, every wave
// repeats the consumer pattern of conv_pc_kernel's XL tile - 12 v_mfma_f32_32x32x16_f16 (3 plane products x 4 output
// blocks) per step - with 0, 4 or 8 ds_read_b128 operand fetches per step (the XL tile: 8).
// Reports FLOP/s from HIP events and the shader clock from s_memtime against the 100 MHz s_memrealtime.
#include "st_common.h"

namespace st {
namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int READS>
__global__ __launch_bounds__(768, 1) void mfma_rate_kernel(int steps, float* sink, unsigned long long* clocks, int mfma_waves,
                                                           int valu_steps, int valu_prio) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1e-3f * (float)(i & 63);
    __syncthreads();
    const int n_valu = (int)(blockDim.x >> 6) - mfma_waves;
    // valu_prio bit 1: s_setprio 3; bit 2: the VALU waves are the workgroup's FIRST (oldest) waves instead of its last
    const bool is_valu = (valu_prio & 2) ? (tid >> 6) < n_valu : (tid >> 6) >= mfma_waves;
    if (is_valu) {
        // "producer-like" wave: nothing but independent fp32 VALU work (32 v_fma_f32 per step on 8 chains), to see how
        // many VALU instructions a wave gets issued beside the SIMD's MFMA streams
        if (valu_prio & 1) __builtin_amdgcn_s_setprio(3);
        float c[8];
        for (int i = 0; i < 8; ++i) c[i] = 1.0f + 0.001f * (float)(lane + i);
        const float m = reinterpret_cast<const float*>(smem)[lane], a = reinterpret_cast<const float*>(smem)[lane + 64];
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int s = 0; s < valu_steps; ++s) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_fmaf(c[i], m, a);
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        float v = 0.f;
        for (int i = 0; i < 8; ++i) v += c[i];
        if (v == 12345.678f) sink[1] = v;
        if (blockIdx.x == 0 && (tid & 63) == 0 && (tid >> 6) == ((valu_prio & 2) ? 0 : mfma_waves)) clocks[2] = t1 - t0;
        return;
    }
    f16x8 op[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) op[i] = *reinterpret_cast<const f16x8*>(smem + lane * 16 + i * 1024);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned off = (unsigned)lane * 16u + (unsigned)(tid >> 6) * 1024u;
    // the NEXT step's operand fetches are issued before this step's MFMAs (software pipelined like the tile's consumer
    // loop, two register sets in ping-pong): READS fetches of 1 KB per wave, addresses moving through a 64 KB window
    f16x8 alt[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) alt[r] = op[r];
    auto fetch = [&](f16x8(&dst)[8]) __attribute__((always_inline)) {
        if constexpr (READS > 0) {
#pragma unroll
            for (int r = 0; r < READS; ++r) {
                const unsigned a = (off + (unsigned)r * 4096u) & 0xffffu;
                dst[r & 7] = *reinterpret_cast<const f16x8*>(smem + a);
            }
            off += 32768u;                                  // the other half of the window: no fetch repeats the previous step's
        }
    };
    // a0 a1 (2 co blocks each), b0 b1 (2 px blocks each): 4 blocks x (a0 b0 + a0 b1 + a1 b0)
    auto multiply = [&](const f16x8(&o)[8]) __attribute__((always_inline)) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int ca = blk >> 1, pb = blk & 1;
            acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o[ca], o[4 + pb], acc[blk], 0, 0, 0);
            acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o[ca], o[6 + pb], acc[blk], 0, 0, 0);
            acc[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o[2 + ca], o[4 + pb], acc[blk], 0, 0, 0);
        }
    };
    for (int s = 0; s < steps; s += 2) {
        fetch(alt);
        multiply(op);
        fetch(op);
        multiply(alt);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float v = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 16; ++j) v += acc[i][j];
    if (v == 12345.678f) sink[0] = v;                              // keeps the chain alive
    if (blockIdx.x == 0 && tid == ((valu_prio & 2) ? 64 * n_valu : 0)) { clocks[0] = t1 - t0; clocks[1] = r1 - r0; }
}

template <int READS>
int run_rate(int waves, int steps, int launches, int valu_waves, int valu_steps, int valu_prio, double* tflops, double* mhz,
             double* valu_cycles, hipStream_t s) {
    float* sink = nullptr;
    unsigned long long* clocks = nullptr;
    ST_HIP(hipMalloc(&sink, 256));
    ST_HIP(hipMalloc(&clocks, 256));
    int dev = 0, cus = 0;
    ST_HIP(hipGetDevice(&dev));
    ST_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t lds = 96 * 1024;                                  // one workgroup per CU
    ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_rate_kernel<READS>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    ST_HIP(hipEventCreate(&e0));
    ST_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) mfma_rate_kernel<READS><<<cus, 64 * (waves + valu_waves), lds, s>>>(steps, sink, clocks, waves, valu_steps, valu_prio);
    ST_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < launches; ++i) mfma_rate_kernel<READS><<<cus, 64 * (waves + valu_waves), lds, s>>>(steps, sink, clocks, waves, valu_steps, valu_prio);
    ST_HIP(hipEventRecord(e1, s));
    ST_HIP(hipEventSynchronize(e1));
    ST_LAUNCH_CHECK();
    float ms = 0.f;
    ST_HIP(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[3] = {0, 0, 0};
    ST_HIP(hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost));
    const double flops = (double)launches * cus * waves * (double)steps * 12.0 * 32768.0;
    *tflops = flops / (ms * 1e-3) / 1e12;
    *mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
    // shader cycles per VALU instruction of the VALU-only wave, and (via *tflops) the MFMA rate beside it; the MFMA
    // waves' own loop took h[0] cycles for steps x 12 MFMAs
    if (valu_cycles) {
        valu_cycles[0] = (valu_waves && valu_steps) ? (double)h[2] / ((double)valu_steps * 32.0) : 0.0;
        valu_cycles[1] = (double)h[0] / ((double)steps * 12.0);
    }
    ST_HIP(hipEventDestroy(e0));
    ST_HIP(hipEventDestroy(e1));
    ST_HIP(hipFree(sink));
    ST_HIP(hipFree(clocks));
    return 0;
}

}  // namespace
}  // namespace st

extern "C" int st_op_mfma_valu_rate(int lds_reads, int waves, int steps, int launches, int valu_waves, int valu_steps,
                                    int valu_prio, double* tflops, double* mhz, double* cycles, void* stream);

extern "C" int st_op_mfma_rate(int lds_reads, int waves, int steps, int launches, double* tflops, double* mhz,
                               void* stream) {
    return st_op_mfma_valu_rate(lds_reads, waves, steps, launches, 0, 0, 0, tflops, mhz, nullptr, stream);
}

extern "C" int st_op_mfma_valu_rate(int lds_reads, int waves, int steps, int launches, int valu_waves, int valu_steps,
                                    int valu_prio, double* tflops, double* mhz, double* cycles, void* stream) {
    using namespace st;
    ST_REQUIRE(tflops && mhz && steps > 0 && launches > 0 && waves >= 1 && valu_waves >= 0 && waves + valu_waves <= 12,
               "st_op_mfma_valu_rate: bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (lds_reads) {
        case 0: return run_rate<0>(waves, steps, launches, valu_waves, valu_steps, valu_prio, tflops, mhz, cycles, s);
        case 4: return run_rate<4>(waves, steps, launches, valu_waves, valu_steps, valu_prio, tflops, mhz, cycles, s);
        case 8: return run_rate<8>(waves, steps, launches, valu_waves, valu_steps, valu_prio, tflops, mhz, cycles, s);
        default: ST_REQUIRE(false, "st_op_mfma_rate: lds_reads must be 0, 4 or 8");
    }
    return 1;
}
