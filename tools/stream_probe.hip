// Bandwidth ceiling of the source kernel's access pattern: read NR component planes, write NW of them,
// no arithmetic.  Variants: grid-stride vs full grid, 1 or 2 cells per lane (8- or 16-byte accesses).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int NR = 20, NW = 13;
template <int V> struct Vec;
template <> struct Vec<1> { typedef double T; };
template <> struct Vec<2> { typedef double2 T; };
template <int V>
__global__ __launch_bounds__(256) void k(const double* __restrict__ in, double* __restrict__ out, long n, long sc) {
    typedef typename Vec<V>::T T;
    const long nv = n / V;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < nv; t += (long)gridDim.x * 256) {
        T v[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) v[c] = reinterpret_cast<const T*>(in + c * sc)[t];
#pragma unroll
        for (int c = 0; c < NW; ++c) {
            T w = v[c];
            if constexpr (V == 1) w = w + v[(c + 7) % NR];
            else { w.x += v[(c + 7) % NR].x; w.y += v[(c + 7) % NR].y; }
            reinterpret_cast<T*>(out + c * sc)[t] = w;
        }
    }
}
int main(int argc, char** argv) {
    const long n = 4104L * 4104L, sc = n;
    double *in, *out;
    hipMalloc(&in, NR * sc * 8); hipMalloc(&out, NW * sc * 8);
    hipMemset(in, 0, NR * sc * 8); hipMemset(out, 0, NW * sc * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)(NR + NW) * n * 8;
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s %.3f ms  %.2f TB/s\n", name, ms / 10, bytes / (ms / 10 * 1e-3) / 1e12);
    };
    for (int blocks : {1024, 2048, 4096, 8192}) {
        char nm[64]; snprintf(nm, 64, "1 cell/lane, grid-stride %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, in, out, n, sc); });
        snprintf(nm, 64, "2 cells/lane, grid-stride %d blocks", blocks);
        run(nm, [&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, in, out, n, sc); });
    }
    run("1 cell/lane, full grid", [&] { hipLaunchKernelGGL(k<1>, dim3((n + 255) / 256), dim3(256), 0, 0, in, out, n, sc); });
    run("2 cells/lane, full grid", [&] { hipLaunchKernelGGL(k<2>, dim3((n / 2 + 255) / 256), dim3(256), 0, 0, in, out, n, sc); });
    return 0;
}
