// Measurement aid, not on the hot path: what rate does the 16-bit matrix pipe SUSTAIN on this chip, alone and next to
// the LDS operand stream an LDS-fed tile needs?  The convolution roofline in DESIGN.md / bench.py is quoted against the
// nominal 2.5 PFLOP/s (2.4 GHz); the XL convolution tile runs at ~1.7 GHz under load.  This kernel
// separates "the matrix pipe's own power draw" from "what the tile adds": one persistent workgroup per CU, every wave
// repeats the consumer pattern of conv_pc_kernel's XL tile - 12 v_mfma_f32_32x32x16_f16 (3 plane products x 4 output
// blocks) per step - with 0, 4 or 8 ds_read_b128 operand fetches per step (the XL tile: 8).
// Reports FLOP/s from HIP events and the shader clock from s_memtime against the 100 MHz s_memrealtime.
#include <chrono>

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


#if defined(ST_EXPERIMENTS)
// ---- the consumer pattern of a Winograd F(2x2, 3x3) fp16x3 tile (profiles/r05_winograd.md) ------------------------------------
// Per transform position a wave needs FOUR fresh operands (two planes of the weights' block, two planes of the transformed
// input's block) for THREE MFMAs (h0 g0 + h0 g1 + h1 g0) - 0.75 MFMAs per ds_read_b128 against 1.5 in the shipped direct
// tile, which re-uses one operand over several output blocks and taps.  What does the matrix pipe sustain in that pattern?
// One persistent workgroup of `waves` waves per CU; a step = 4 positions = 16 ds_read_b128 + 12 v_mfma_f32_32x32x16_f16,
// next step's reads issued before this step's MFMAs, 16 accumulators (positions) per wave as in the real tile.
__global__ __launch_bounds__(256, 1) void wino_rate_kernel(int steps, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 32768; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1e-3f * (float)(i & 63);
    __syncthreads();
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    f16x8 cur[16], nxt[16];
    unsigned off = (unsigned)lane * 16u + (unsigned)(tid >> 6) * 1024u;
    auto fetch = [&](f16x8(&dst)[16]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r] = *reinterpret_cast<const f16x8*>(smem + ((off + (unsigned)r * 4096u) & 0x1ffffu));
        off += 65536u;
    };
    auto multiply = [&](const f16x8(&o)[16], int base) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            acc[base + p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o[4 * p], o[4 * p + 2], acc[base + p], 0, 0, 0);
            acc[base + p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o[4 * p], o[4 * p + 3], acc[base + p], 0, 0, 0);
            acc[base + p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o[4 * p + 1], o[4 * p + 2], acc[base + p], 0, 0, 0);
        }
    };
    fetch(cur);
    for (int s = 0; s < steps; s += 8) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                         // 4 steps x 4 positions = the 16 positions of a chunk, twice
            fetch(nxt);
            multiply(cur, 4 * q);
            fetch(cur);
            multiply(nxt, 4 * q);
        }
    }
    float v = 0.f;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) v += acc[i][j];
    if (v == 12345.678f) sink[0] = v;
}

int run_wino_rate(int steps, int launches, double* tflops, hipStream_t s) {
    float* sink = nullptr;
    ST_HIP(hipMalloc(&sink, 256));
    int dev = 0, cus = 0;
    ST_HIP(hipGetDevice(&dev));
    ST_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const size_t lds = 128 * 1024;
    ST_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_rate_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    ST_HIP(hipEventCreate(&e0));
    ST_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) wino_rate_kernel<<<cus, 256, lds, s>>>(steps, sink);
    ST_HIP(hipEventRecord(e0, s));
    for (int i = 0; i < launches; ++i) wino_rate_kernel<<<cus, 256, lds, s>>>(steps, sink);
    ST_HIP(hipEventRecord(e1, s));
    ST_HIP(hipEventSynchronize(e1));
    ST_LAUNCH_CHECK();
    float ms = 0.f;
    ST_HIP(hipEventElapsedTime(&ms, e0, e1));
    *tflops = (double)launches * cus * 4.0 * (double)steps * 12.0 * 32768.0 / (ms * 1e-3) / 1e12;
    ST_HIP(hipEventDestroy(e0));
    ST_HIP(hipEventDestroy(e1));
    ST_HIP(hipFree(sink));
    return 0;
}
#else
int run_wino_rate(int, int, double*, hipStream_t) {
    set_error("st_op_winograd_consumer_rate needs a library built with build.py --experiments");
    return 1;
}
#endif

// ---- which streams share a hardware queue with a given stream?  (probe_queue_sharing, used by st_api.hip) ----------------
// ROCm deals HIP streams to GPU_MAX_HW_QUEUES (default 4) hardware queues and streams on one queue run in submission order.
// A kernel that spins on a host-mapped flag is put on `ref`, then TWO one-store marker kernels on every candidate.  The
// first marker of a stream carries no dependency and runs beside the spinner even on a shared queue; the second depends on
// the first, which the runtime expresses with the packet's barrier bit - and on a shared hardware queue that bit makes it
// wait for EVERY earlier packet of the queue, the spinner included (the very mechanism by which streams on one queue end up
// running in submission order).  A candidate whose second marker has not landed after ~2 ms shares ref's hardware queue.  The spinner gives up by
// itself after ~20 ms of wall time (s_memrealtime, 100 MHz), so a lost host write cannot hang the device.
__global__ void queue_probe_spin_kernel(volatile int* flag) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(const_cast<int*>(flag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull) break;
        __builtin_amdgcn_s_sleep(32);
    }
}
__global__ void queue_probe_mark_kernel(int* mark) {
    __hip_atomic_store(mark, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- what does a device-wide barrier cost inside one launch?  (st_op_grid_barrier_time) ---------------------------------
// `wgs` co-resident workgroups (<= one per CU) repeat: write `payload` floats into the workgroup's slot, barrier, read the
// slot of another workgroup (normally on another XCD) and check it.  The barrier is a monotonic counter: release fence
// (agent scope: L2 write-back towards the other XCDs), one atomic add per workgroup, spin on an agent-scope load, acquire
// fence (invalidate).  The persistent Newton-Schulz chain kernel (st_nschain.hip) is built on exactly this barrier; this
// aid measured what a recurrence level costs before that kernel existed (profiles/r04_grid_barrier.md).
// Barrier word layout (unsigned ints, 64 per 256-byte line): line 0 = the flat / top counter, lines 1 .. G = group
// counters, lines 33 .. 32 + G = group release flags, line 72 = the second barrier's counter (payload mode).
// groups == 0: every workgroup adds to ONE counter and polls it (256 same-address atomics: ~28 ns each, serialised at the
// memory side).  groups == G: workgroup w arrives at group counter w % G (under round-robin dispatch, G = 8 puts an XCD's
// workgroups on one counter); the last arriver of a group arrives at the top counter; the last arriver there raises
// every group's release flag; a workgroup polls only its group's flag.
// `cfg` = groups + 100 * per_wg_flags + 1000 * sleep: per_wg_flags = 1: the last arriver's workgroup raises one flag PER
// WORKGROUP (256 lanes, one store each, every poller on its own line: words + 64 * (80 + wg)) instead of one per group;
// sleep = s_sleep argument between polls.  All atomics relaxed: a workgroup's data is at the memory side before its arrival
// is issued (release fence first), every relay acts on a returned value, and the acquire fence follows the last poll.
__device__ __forceinline__ void grid_barrier_wait(unsigned int* words, int cfg, unsigned int round, int wg, int nwg, int tid,
                                                  unsigned int* lds_flag) {
    const int groups = cfg % 100, per_wg = (cfg / 100) % 10, nap = (cfg / 1000) % 10;
    // mode 1 (cfg / 10000 % 10): NO cache maintenance - the payload travels in agent-scope (sc1) stores and loads, which are
    // coherent at the memory side by themselves; the arrival only has to wait until the workgroup's stores are acknowledged
    // (s_waitcnt vmcnt(0) by every lane, done by the caller before its __syncthreads)
    const bool fenced = (cfg / 10000) % 10 == 0;
    unsigned int* flag = per_wg ? words + 64 * (80 + wg) : (groups ? words + 64 * (33 + wg % groups) : words);
    if (tid == 0) {
        if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        bool last = false;
        if (groups == 0) {
            const unsigned int prev = __hip_atomic_fetch_add(words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = prev == round * (unsigned int)nwg - 1;
        } else {
            const int g = wg % groups;
            const unsigned int members = (unsigned int)((nwg - g + groups - 1) / groups);
            const unsigned int prev = __hip_atomic_fetch_add(words + 64 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == round * members - 1) {
                const unsigned int top = __hip_atomic_fetch_add(words + 64 * 70, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = top == round * (unsigned int)groups - 1;
            }
        }
        *lds_flag = last ? 1u : 0u;
    }
    if (per_wg || groups) {
        __syncthreads();
        if (*lds_flag) {                                      // the last arriver's workgroup releases everybody
            const int n = per_wg ? nwg : groups;
            for (int k = tid; k < n; k += blockDim.x)
                __hip_atomic_store(per_wg ? words + 64 * (80 + k) : words + 64 * (33 + k), round, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0) {
        const unsigned int target = (groups == 0 && !per_wg) ? round * (unsigned int)nwg : round;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (nap <= 1) __builtin_amdgcn_s_sleep(1);
            else if (nap <= 4) __builtin_amdgcn_s_sleep(4);
            else __builtin_amdgcn_s_sleep(16);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) break;      // 0.2 s: a lost workgroup must not hang the device
        }
        if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
}

typedef unsigned int u32x4_t __attribute__((__vector_size__(16)));

__global__ __launch_bounds__(256) void grid_barrier_kernel(unsigned int* counter, float* slots, int payload, int rounds,
                                                           unsigned int* errors, unsigned long long* clocks, int groups) {
    const int wg = blockIdx.x, nwg = gridDim.x, tid = threadIdx.x;
    __shared__ unsigned int lds_flag;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned int bad = 0;
    const bool wt = (groups / 10000) % 10 == 1;               // write-through payload (see grid_barrier_wait)
    const int reads = groups / 100000;                        // each workgroup reads `reads` other slots per round (>= 1)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slots, 0, nwg * payload * 4, 0x00020000);
    for (int r = 1; r <= rounds; ++r) {
        if (wt) {
            for (int i = tid * 4; i < payload; i += blockDim.x * 4) {
                const float v = (float)(r * 1024 + wg);
                const f32x4 q = {v, v, v, v};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, q), rs, (wg * payload + i) * 4, 0, 16);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (int i = tid; i < payload; i += blockDim.x) slots[(size_t)wg * payload + i] = (float)(r * 1024 + wg);
        }
        __syncthreads();
        grid_barrier_wait(counter, groups, (unsigned int)r, wg, nwg, tid, &lds_flag);
        __syncthreads();
        for (int k = 0; k < (reads > 0 ? reads : 1); ++k) {
            const int other = (wg + 37 + 8 * k + k) % nwg;    // 37 is odd: another XCD under round-robin dispatch
            if (wt) {
                for (int i = tid * 4; i < payload; i += blockDim.x * 4) {
                    const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (other * payload + i) * 4, 0, 16));
                    const float want = (float)(r * 1024 + other);
                    bad += (q[0] != want) + (q[1] != want) + (q[2] != want) + (q[3] != want);
                }
            } else {
                for (int i = tid; i < payload; i += blockDim.x)
                    if (slots[(size_t)other * payload + i] != (float)(r * 1024 + other)) ++bad;
            }
        }
        // the slot is overwritten next round: nobody may still be reading it -> second barrier only when there is a payload
        if (payload > 0) {
            if (wt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (a workgroup-scope barrier does not wait for loads)
            __syncthreads();
            grid_barrier_wait(counter + 64 * 512, groups, (unsigned int)r, wg, nwg, tid, &lds_flag);
            __syncthreads();
        }
    }
    if (bad) atomicAdd(errors, bad);
    if (wg == 0 && tid == 0) clocks[0] = __builtin_amdgcn_s_memtime() - t0;
}

int run_grid_barrier(int wgs, int rounds, int payload, int groups, double* us_per_round, int* errors_out, hipStream_t s) {
    unsigned int *counter = nullptr, *errors = nullptr;
    float* slots = nullptr;
    unsigned long long* clocks = nullptr;
    ST_HIP(hipMalloc(&counter, 1 << 19));
    ST_HIP(hipMalloc(&errors, 256));
    ST_HIP(hipMalloc(&clocks, 256));
    ST_HIP(hipMalloc(&slots, (size_t)wgs * (payload > 0 ? payload : 1) * sizeof(float)));
    hipEvent_t e0, e1;
    ST_HIP(hipEventCreate(&e0));
    ST_HIP(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        ST_HIP(hipMemsetAsync(counter, 0, 1 << 19, s));
        ST_HIP(hipMemsetAsync(errors, 0, 256, s));
        ST_HIP(hipEventRecord(e0, s));
        hipLaunchKernelGGL(grid_barrier_kernel, dim3(wgs), dim3(256), 0, s, counter, slots, payload, rounds, errors, clocks, groups);
        ST_HIP(hipEventRecord(e1, s));
        ST_HIP(hipEventSynchronize(e1));
        ST_LAUNCH_CHECK();
        float ms = 0.f;
        ST_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    unsigned int herr = 0;
    ST_HIP(hipMemcpy(&herr, errors, sizeof(herr), hipMemcpyDeviceToHost));
    *us_per_round = (double)best * 1e3 / rounds;
    *errors_out = (int)herr;
    ST_HIP(hipEventDestroy(e0));
    ST_HIP(hipEventDestroy(e1));
    ST_HIP(hipFree(counter)); ST_HIP(hipFree(errors)); ST_HIP(hipFree(clocks)); ST_HIP(hipFree(slots));
    return 0;
}

}  // namespace

// shares[i] = 1 if candidate i did not run while a kernel on `ref` was executing (same hardware queue), 0 if it did, and
// returns 0; returns 1 (shares untouched) if the probe could not be carried out.
int probe_queue_sharing(hipStream_t ref, const hipStream_t* candidates, int count, int* shares) {
    if (count < 1 || count > 7) return 1;
    int* host = nullptr;
    if (hipHostMalloc(reinterpret_cast<void**>(&host), 16 * sizeof(int), hipHostMallocMapped) != hipSuccess) {
        hipGetLastError();
        return 1;
    }
    for (int i = 0; i < 16; ++i) host[i] = 0;
    int* dev = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0) != hipSuccess) {
        hipGetLastError();
        hipHostFree(host);
        return 1;
    }
    hipLaunchKernelGGL(queue_probe_spin_kernel, dim3(1), dim3(64), 0, ref, dev);
    for (int i = 0; i < count; ++i) {
        hipLaunchKernelGGL(queue_probe_mark_kernel, dim3(1), dim3(64), 0, candidates[i], dev + 8 + i);
        hipLaunchKernelGGL(queue_probe_mark_kernel, dim3(1), dim3(64), 0, candidates[i], dev + 1 + i);
    }
    volatile int* v = host;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        int landed = 0;
        for (int i = 0; i < count; ++i) landed += v[1 + i] != 0;
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (landed == count || us > 2000.0) break;
    }
    for (int i = 0; i < count; ++i) shares[i] = v[1 + i] == 0;
    __atomic_store_n(host, 1, __ATOMIC_SEQ_CST);                  // release the spinner
    bool ok = hipStreamSynchronize(ref) == hipSuccess;
    for (int i = 0; i < count; ++i) ok = (hipStreamSynchronize(candidates[i]) == hipSuccess) && ok;
    hipHostFree(host);
    if (!ok) hipGetLastError();
    return ok ? 0 : 1;
}

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


extern "C" int st_op_winograd_consumer_rate(int steps, int launches, double* tflops, void* stream) {
    using namespace st;
    ST_REQUIRE(tflops && steps >= 8 && steps % 8 == 0 && launches > 0, "st_op_winograd_consumer_rate: bad argument");
    return run_wino_rate(steps, launches, tflops, static_cast<hipStream_t>(stream));
}

extern "C" int st_op_grid_barrier_time(int workgroups, int rounds, int payload_floats, int groups, double* us_per_round,
                                       int* errors, void* stream) {
    using namespace st;
    ST_REQUIRE(us_per_round && errors && workgroups >= 1 && workgroups <= 256 && rounds >= 1 && payload_floats >= 0 &&
               payload_floats <= (1 << 20) && payload_floats % 4 == 0 && groups >= 0 && groups % 100 <= 32, "st_op_grid_barrier_time: bad argument");
    return run_grid_barrier(workgroups, rounds, payload_floats, groups, us_per_round, errors, static_cast<hipStream_t>(stream));
}
