// Accuracy probe for the FAST-mode division / square root (v_rcp_f64 / v_rsq_f64 seed + Newton steps).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__device__ double rcp_n(double x, int it) {
    double r = __builtin_amdgcn_rcp(x);
    for (int k = 0; k < it; ++k) { double e = fma(-x, r, 1.0); r = fma(e, r, r); }
    return r;
}
__device__ double sqrt_n(double x, int it) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    for (int k = 0; k < it; ++k) { double r = fma(-h, g, 0.5); g = fma(g, r, g); h = fma(h, r, h); }
    double d = fma(-g, g, x);
    return fma(d, h, g);
}
__global__ void probe(const double* x, double* out, int n, int it) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = rcp_n(x[i], it);
    out[n + i] = sqrt_n(x[i], it);
}
int main() {
    const int n = 1 << 22;
    std::mt19937_64 gen(7);
    std::uniform_real_distribution<double> ex(-40, 40), man(1.0, 2.0);
    std::vector<double> x(n), o(2 * n);
    for (auto& v : x) v = std::ldexp(man(gen), (int)ex(gen));
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, 2 * n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    for (int it = 0; it <= 3; ++it) {
        hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dx, dout, n, it);
        hipMemcpy(o.data(), dout, 2 * n * 8, hipMemcpyDeviceToHost);
        double er = 0, es = 0;
        for (int i = 0; i < n; ++i) {
            er = std::fmax(er, std::fabs(o[i] * x[i] - 1.0));
            long double s = sqrtl((long double)x[i]);
            es = std::fmax(es, (double)fabsl((o[n + i] - s) / s));
        }
        printf("newton steps %d: max rel err rcp %.3e  sqrt(+1 final corr) %.3e\n", it, er, es);
    }
    return 0;
}
