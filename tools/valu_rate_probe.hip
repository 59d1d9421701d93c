// valu_rate_probe.hip — issue cost of the fp64 instructions the marching kernels are made of (diagnostic, not part
// of the library): cycles per wave-instruction on one SIMD with 1 and 2 waves resident, for 8 / 16 / 32 independent chains of
// v_fma_f64, v_mul_f64, v_add_f64, v_rcp_f64, v_rsq_f64, v_rcp_f32 (+ conversions), v_cndmask pairs, v_mov_dpp pairs.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/bin/valu_rate_probe && tools/bin/valu_rate_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHK(x)                                                                 \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                     \
            return 1;                                                          \
        }                                                                      \
    } while (0)

constexpr int ITER = 4096;

template <int OP, int NACC>
__global__ __launch_bounds__(64) void k_probe(double* out, double seed) {
    double a[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) a[q] = seed + 0.001 * (threadIdx.x + 64 * q);
    const double c1 = 1.0000001, c2 = 1e-9;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) {
            if (OP == 0) a[q] = fma(a[q], c1, c2);
            if (OP == 1) a[q] = a[q] * c1;
            if (OP == 2) a[q] = a[q] + c2;
            if (OP == 3) a[q] = __builtin_amdgcn_rcp(a[q]);
            if (OP == 4) a[q] = __builtin_amdgcn_rsq(a[q]);
            if (OP == 5) a[q] = (double)__builtin_amdgcn_rcpf((float)a[q]);
            if (OP == 6) a[q] = (a[q] > 1.5) ? c1 : a[q];                                      // compare + select pair
            if (OP == 7) {                                                                       // wave shift of a double
                const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a[q]), 0x138, 0xf, 0xf, true);
                const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a[q]), 0x138, 0xf, 0xf, true);
                a[q] = __hiloint2double(hi, lo);
            }
            if (OP == 8) a[q] = fmax(a[q], c2);
            if (OP == 9) a[q] = sqrt(a[q]);   // the compiler's full-precision sequence
            if (OP == 10) a[q] = 1.0 / a[q];
        }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += a[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int OP, int NACC = 8>
int run(const char* name, double instr_per_op, double* d_out) {
    int dev = 0;
    hipDeviceProp_t p;
    CHK(hipGetDevice(&dev));
    CHK(hipGetDeviceProperties(&p, dev));
    const int simds = p.multiProcessorCount * 4;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int waves_per_simd = 1; waves_per_simd <= 2; ++waves_per_simd) {
        const int grid = simds * waves_per_simd;
        hipLaunchKernelGGL((k_probe<OP, NACC>), dim3(grid), dim3(64), 0, 0, d_out, 1.25);
        CHK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_probe<OP, NACC>), dim3(grid), dim3(64), 0, 0, d_out, 1.25);
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double clk = (double)p.clockRate * 1e3;  // Hz
        const double ops_per_simd = (double)ITER * NACC * waves_per_simd;
        const double cyc = ms * 1e-3 * clk / ops_per_simd;
        printf("%-28s chains %2d waves/SIMD %d: %7.2f cycles per op (%4.1f instr) at %.0f MHz, %.3f ms\n", name, NACC, waves_per_simd, cyc, instr_per_op, clk / 1e6, ms);
    }
    return 0;
}

int main() {
    double* d_out = nullptr;
    CHK(hipMalloc((void**)&d_out, sizeof(double) * 64 * 8192));
    run<0>("v_fma_f64", 1, d_out);
    run<0, 16>("v_fma_f64", 1, d_out);
    run<0, 32>("v_fma_f64", 1, d_out);
    run<1, 16>("v_mul_f64", 1, d_out);
    run<2, 32>("v_add_f64", 1, d_out);
    run<3, 16>("v_rcp_f64", 1, d_out);
    run<7, 16>("2 v_mov_b32 dpp wave_shr", 2, d_out);
    run<6, 16>("v_cmp_f64 + 2 v_cndmask", 3, d_out);
    run<1>("v_mul_f64", 1, d_out);
    run<2>("v_add_f64", 1, d_out);
    run<8>("v_max_f64", 1, d_out);
    run<3>("v_rcp_f64", 1, d_out);
    run<4>("v_rsq_f64", 1, d_out);
    run<5>("cvt + v_rcp_f32 + cvt", 3, d_out);
    run<6>("v_cmp_f64 + 2 v_cndmask", 3, d_out);
    run<7>("2 v_mov_b32 dpp wave_shr", 2, d_out);
    run<9>("sqrt() full sequence", 20, d_out);
    run<10>("1.0 / x full sequence", 9, d_out);
    printf("done\n");
    return 0;
}
