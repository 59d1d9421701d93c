// mfp_probe.hip — attainable HBM bandwidth of the device, measured the way the roofline figures are priced
// against: a device-to-device copy with 16 bytes per lane and access (the shape MI355X_MICROARCH.md quotes
// 6.29 TB/s for), grid-stride over a buffer far larger than the Infinity Cache.  Measurement aid of bench.py
// (cmfp_hbm_copy_probe); no counterpart in the reference.
#include <hip/hip_runtime.h>

#include "../../include/cerberus_mfp.h"

namespace probe {

typedef double v2d __attribute__((ext_vector_type(2)));  // a native 16-byte vector (the nontemporal builtins take no struct)

template <int UNROLL>
__global__ __launch_bounds__(256) void k_copy16(const v2d* __restrict__ src, v2d* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        v2d v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

}  // namespace probe
using probe::v2d;
using probe::k_copy16;

extern "C" int cmfp_hbm_copy_probe(int device, void* stream, long long nbytes, int reps, double* gbs) {
    if (!gbs || nbytes < (1 << 20) || reps < 1) return CMFP_ERR_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return CMFP_ERR_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return CMFP_ERR_HIP;
    hipStream_t st = (hipStream_t)stream;
    const long n = nbytes / 16;
    v2d *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = CMFP_ERR_HIP;
    double best = 0.0;
    if (hipMalloc((void**)&a, (size_t)n * 16) != hipSuccess) goto done;
    if (hipMalloc((void**)&b, (size_t)n * 16) != hipSuccess) goto done;
    if (hipMemsetAsync(a, 0, (size_t)n * 16, st) != hipSuccess) goto done;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) goto done;
    {
        hipDeviceProp_t p;
        int cus = 256;
        if (hipGetDeviceProperties(&p, device) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
        // three grid sizes (workgroups per CU): the best is the attainable figure
        const int per_cu[3] = {8, 16, 32};
        for (int g = 0; g < 3; ++g) {
            const dim3 grid((unsigned)(cus * per_cu[g])), block(256);
            hipLaunchKernelGGL(k_copy16<4>, grid, block, 0, st, a, b, n);  // warm-up
            if (hipEventRecord(e0, st) != hipSuccess) goto done;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_copy16<4>, grid, block, 0, st, a, b, n);
            if (hipEventRecord(e1, st) != hipSuccess) goto done;
            if (hipEventSynchronize(e1) != hipSuccess) goto done;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f) goto done;
            const double v = 2.0 * (double)n * 16.0 * reps / (ms * 1e-3) / 1e9;
            if (v > best) best = v;
        }
    }
    *gbs = best;
    rc = CMFP_OK;
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    return rc;
}
