// mfp_peer.hip — kernels of the device-side halo / dt transport (mfp_peer.h).
#include "mfp_peer.h"

namespace cmfp_peer {

static constexpr int TPB = 256;

__device__ __forceinline__ void sys_store(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ double sys_load(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_SYSTEM));
}

__device__ __forceinline__ bool decode(const IBox& b, long t, int& i, int& j, int& k) {
    const long nx = b.hi[0] - b.lo[0] + 1, ny = b.hi[1] - b.lo[1] + 1, nz = b.hi[2] - b.lo[2] + 1;
    if (t >= nx * ny * nz) return false;
    i = b.lo[0] + (int)(t % nx);
    j = b.lo[1] + (int)((t / nx) % ny);
    k = b.lo[2] + (int)(t / (nx * ny));
    return true;
}

// the ngs layers next to a face over the full ghosted extent of the other dimensions (the layout of k_pack,
// mfp_kernels.hip): component m of slab point t at [t + m * npts]
template <bool UNPACK>
__global__ __launch_bounds__(TPB) void k_peer_copy(Geo g, IBox slab, int nc, int d, int first, double* __restrict__ A, double* __restrict__ buf) {
    const long t = (long)blockIdx.x * TPB + threadIdx.x;
    int c3[3];
    if (!decode(slab, t, c3[0], c3[1], c3[2])) return;
    const long np = slab.npts();
    c3[d] = first + (c3[d] - slab.lo[d]);
    const long c = g.at(c3[0], c3[1], c3[2]);
    if (UNPACK) {
        double v[10];
#pragma unroll
        for (int m = 0; m < 10; ++m)
            if (m < nc) v[m] = sys_load(buf + t + m * np);
#pragma unroll
        for (int m = 0; m < 10; ++m)
            if (m < nc) A[c + m * g.sc] = v[m];
    } else {
        double v[10];
#pragma unroll
        for (int m = 0; m < 10; ++m)
            if (m < nc) v[m] = A[c + m * g.sc];
#pragma unroll
        for (int m = 0; m < 10; ++m)
            if (m < nc) sys_store(buf + t + m * np, v[m]);
    }
}

static IBox slab_of(const Geo& g, int d, int ngs) {
    IBox s;
    for (int t = 0; t < 3; ++t) {
        s.lo[t] = -g.ng[t];
        s.hi[t] = g.n[t] - 1 + g.ng[t];
    }
    s.lo[d] = 1;
    s.hi[d] = ngs;
    return s;
}

void pack(const Geo& g, int nc, int ngs, const double* A, int d, int side, double* remote_slot, hipStream_t st) {
    const IBox s = slab_of(g, d, ngs);
    const int first = side == 0 ? 0 : g.n[d] - ngs;  // valid layers next to the face
    const long n = s.npts();
    hipLaunchKernelGGL(k_peer_copy<false>, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, st, g, s, nc, d, first, const_cast<double*>(A),
                       remote_slot);
}
void unpack(const Geo& g, int nc, int ngs, double* A, int d, int side, const double* local_slot, hipStream_t st) {
    const IBox s = slab_of(g, d, ngs);
    const int first = side == 0 ? -ngs : g.n[d];  // ghost layers
    const long n = s.npts();
    hipLaunchKernelGGL(k_peer_copy<true>, dim3((unsigned)((n + TPB - 1) / TPB)), dim3(TPB), 0, st, g, s, nc, d, first, A,
                       const_cast<double*>(local_slot));
}

// everything this stream wrote before (the pack launches: complete, their stores written through at system scope)
// is ordered before the words
__global__ void k_peer_signal(Flags2 f) {
    const int t = threadIdx.x;
    if (t >= f.n) return;
    __threadfence_system();
    __hip_atomic_store(f.flag[t], f.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bounded spin on words in this rank's own arena
__device__ __forceinline__ bool spin_until(const unsigned long long* flag, unsigned long long value, long long ticks) {
    const long long t0 = wall_clock64();
    for (;;) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) >= value) return true;
        if (wall_clock64() - t0 > ticks) return false;
        __builtin_amdgcn_s_sleep(8);
    }
}
__global__ void k_peer_wait(Flags2 f, int* err, long long ticks) {
    const int t = threadIdx.x;
    if (t >= f.n) return;
    if (!spin_until(f.flag[t], f.value, ticks)) atomicOr(err, ERR_TIMEOUT_BIT);
}

void signal(const Flags2& f, hipStream_t st) {
    if (f.n > 0) hipLaunchKernelGGL(k_peer_signal, dim3(1), dim3(64), 0, st, f);
}

static long long ticks_of(double seconds) {
    static int khz = 0;
    if (!khz) {
        int dev = 0, v = 0;
        khz = 100000;  // 100 MHz: the constant-rate counter of CDNA3 / CDNA4
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeWallClockRate, dev) == hipSuccess && v > 0) khz = v;
    }
    return (long long)(seconds * 1e3 * (double)khz);
}

void wait(const Flags2& f, int* err, double timeout_s, hipStream_t st) {
    if (f.n > 0) hipLaunchKernelGGL(k_peer_wait, dim3(1), dim3(64), 0, st, f, err, ticks_of(timeout_s));
}

// this rank's dt into slot [rank][parity] of every other rank, then their word dtready[rank]
__global__ void k_peer_dt_put(DtPeers p, const double* dt) {
    const int t = threadIdx.x;
    if (t >= p.world || t == p.rank) return;
    const double v = dt[0];
    double* slot = reinterpret_cast<double*>(p.arena[t] + p.dtslot) + 2 * p.rank + (int)(p.value & 1ull);
    sys_store(slot, v);
    __threadfence_system();
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p.arena[t] + p.dtready) + p.rank, p.value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// dt = min over every rank's value (one workgroup of MAX_RANKS threads)
__global__ void k_peer_dt_min(DtPeers p, double* dt, int* err, long long ticks) {
    __shared__ double sh[MAX_RANKS];
    const int t = threadIdx.x;
    double v = dt[0];
    if (t < p.world && t != p.rank) {
        char* mine = p.arena[p.rank];
        if (spin_until(reinterpret_cast<unsigned long long*>(mine + p.dtready) + t, p.value, ticks))
            v = sys_load(reinterpret_cast<double*>(mine + p.dtslot) + 2 * t + (int)(p.value & 1ull));
        else
            atomicOr(err, ERR_TIMEOUT_BIT);
    }
    sh[t] = v;
    __syncthreads();
    if (t == 0) {
        double m = sh[0];
        for (int q = 1; q < MAX_RANKS; ++q) m = sh[q] < m ? sh[q] : m;  // std::min(m, sh[q])
        dt[0] = m;
    }
}

void dt_put(const DtPeers& p, const double* dt, hipStream_t st) { hipLaunchKernelGGL(k_peer_dt_put, dim3(1), dim3(MAX_RANKS), 0, st, p, dt); }
void dt_min(const DtPeers& p, double* dt, int* err, double timeout_s, hipStream_t st) {
    hipLaunchKernelGGL(k_peer_dt_min, dim3(1), dim3(MAX_RANKS), 0, st, p, dt, err, ticks_of(timeout_s));
}

}  // namespace cmfp_peer
