// Streams confined to a subset of the chip's 8 XCDs (hipExtStreamCreateWithCUMask).
//
// Why: the two n = 512 Newton-Schulz chains of the W2 style loss (relu4_1, relu5_1; reference sqrtm.py:9-47) run in the
// same window - between the end of the forward trunk and the start of the backward - as ~110 dependent 512^3 products.
// Every product's operands were written by the previous launch on ALL XCDs, the per-XCD L2s are not coherent with each
// other, so every XCD re-fetches all of them over the fabric; two chains side by side are fabric-throughput bound and
// slow each other 1.6x ("critical path at 512^2").  Giving each chain its own XCDs halves the number
// of L2s that fetch a chain's operands and keeps the chains out of each other's L2s and CUs.
//
// The bit -> CU mapping of a queue's CU mask is not documented for multi-XCC parts, so it is PROBED: a kernel records
// the XCC id (s_getreg_b32 HW_REG_XCC_ID) of its workgroups on a stream built with a candidate mask; the first candidate
// layout whose workgroups land exactly on the requested XCDs is used, and if none does the caller gets an ordinary
// stream (the partition is a speed measure, never a correctness requirement).
#include <vector>

#include "st_common.h"

namespace st {
namespace {

__global__ void xcc_probe_kernel(unsigned int* hist) {
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // some work, so that the launch spreads over every CU the queue may use instead of draining through the first few
    float v = (float)threadIdx.x;
    for (int i = 0; i < 2000; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
    if (v == 123.456f) hist[15] = 1;
    if (threadIdx.x == 0) atomicAdd(&hist[xcc & 7], 1u);
}

// layout 0: bit i of the mask belongs to XCC i % 8 (the KFD spreads a queue's mask round-robin over the XCCs);
// layout 1: bits [n i, n (i + 1)) belong to XCC i (n = CUs per XCC)
std::vector<uint32_t> candidate_mask(unsigned xcc_set, int layout, int cus) {
    std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
    const int per_xcc = cus / 8;
    for (int i = 0; i < cus; ++i) {
        const int xcc = layout == 0 ? i % 8 : i / per_xcc;
        if (xcc_set >> xcc & 1u) mask[(size_t)i / 32] |= 1u << (i % 32);
    }
    return mask;
}

int probe(hipStream_t s, unsigned int* hist_dev, unsigned* seen) {
    ST_HIP(hipMemsetAsync(hist_dev, 0, 16 * sizeof(unsigned int), s));
    hipLaunchKernelGGL(xcc_probe_kernel, dim3(2048), dim3(64), 0, s, hist_dev);
    ST_LAUNCH_CHECK();
    unsigned int h[16];
    ST_HIP(hipMemcpyAsync(h, hist_dev, sizeof(h), hipMemcpyDeviceToHost, s));
    ST_HIP(hipStreamSynchronize(s));
    *seen = 0;
    for (int i = 0; i < 8; ++i)
        if (h[i]) *seen |= 1u << i;
    return 0;
}

}  // namespace

// A stream whose kernels run only on the XCDs in `xcc_set` (bit i = XCD i).  *confined = 1 when the partition was
// verified by the probe, 0 when an ordinary non-blocking stream was returned instead.  NOTE: HIP creates CU-mask streams
// as BLOCKING streams (they synchronise implicitly with the legacy null stream); st_api.hip therefore moves the closure
// off the null stream when it uses them.
int create_xcc_stream(hipStream_t* out, unsigned xcc_set, int* confined) {
    *confined = 0;
    int dev = 0, cus = 0;
    ST_HIP(hipGetDevice(&dev));
    ST_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    xcc_set &= 0xffu;
    static int good_layout = -2;          // -2: not probed yet; -1: no layout works on this system
    if (xcc_set != 0 && xcc_set != 0xffu && cus >= 8 && cus % 8 == 0 && good_layout != -1) {
        unsigned int* hist = nullptr;
        ST_HIP(hipMalloc(&hist, 16 * sizeof(unsigned int)));
        for (int layout = (good_layout >= 0 ? good_layout : 0); layout <= (good_layout >= 0 ? good_layout : 1); ++layout) {
            const std::vector<uint32_t> mask = candidate_mask(xcc_set, layout, cus);
            hipStream_t s = nullptr;
            if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
                hipGetLastError();
                continue;
            }
            unsigned seen = 0;
            if (probe(s, hist, &seen) == 0 && seen == xcc_set) {
                good_layout = layout;
                *out = s;
                *confined = 1;
                hipFree(hist);
                return 0;
            }
            hipStreamDestroy(s);
        }
        hipFree(hist);
        if (good_layout == -2) good_layout = -1;
    }
    ST_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return 0;
}

}  // namespace st

// Diagnostic (include/st_amd.h, "measurement aids"): which XCDs do workgroups of a stream confined to `xcc_set` run on?
// seen_out = bit set of the XCC ids observed, layout_out = 0 (mask bits interleaved over the XCCs), 1 (blocked), -1 (none
// of the candidate layouts gave the requested set; the stream is unconfined).
extern "C" int st_op_xcc_stream_probe(unsigned int xcc_set, unsigned int* seen_out, int* confined_out) {
    using namespace st;
    ST_REQUIRE(seen_out && confined_out, "st_op_xcc_stream_probe: null argument");
    hipStream_t s = nullptr;
    if (create_xcc_stream(&s, xcc_set, confined_out)) return 1;
    unsigned int* hist = nullptr;
    ST_HIP(hipMalloc(&hist, 16 * sizeof(unsigned int)));
    const int rc = probe(s, hist, seen_out);
    hipFree(hist);
    hipStreamDestroy(s);
    return rc;
}
