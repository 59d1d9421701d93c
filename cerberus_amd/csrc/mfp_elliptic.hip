// mfp_elliptic.hip — divergence projection of the Maxwell state on a periodic uniform block:
// Elliptic::project_divergence_field + solve_divergence (actions/MFP_elliptic.cpp:636-782, 784-964).
//
// The reference hands the solve to AMReX: a cell-centred vector V is projected with the *nodal* operators of
// MLNodeLaplacian [AMReX, absent from the reference tree: restated from its published discretisation]:
//     rhs  = D V + avg(S)          nodal divergence of the cell-centred field (compRHS), S cell-centred
//     L phi = rhs                  bilinear / trilinear finite-element Laplacian on the nodes
//     V   += -G phi                cell-centred gradient of the nodal phi (getFluxes), sigma = 1
//     phi_cc = average of the cell's nodes (average_node_to_cellcenter)
// D and G are (difference along d) x (average across the other directions); L is the FE stencil
// ([1 1 1; 1 -8 1; 1 1 1] / 3h^2 in 2-D for dx = dy), not D G: the projection is approximate to O(h^2), as the
// reference's own comment says (MFP_elliptic.cpp:806).  On a periodic block L is diagonal in Fourier space, so the
// solve is exact here (the reference iterates MLMG to reltol 1e-8): parity with the reference is unpinned.
// Power-of-two cells per periodic dimension (radix-2 FFT in LDS, one line per workgroup).
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "mfp_elliptic.h"

namespace cmfp_ell {
using cmfp::Geo;

__device__ __forceinline__ int wrap(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// S_cc = -cd_scale * (sum_s rho_s q_s / m_s) / ep on the valid cells (MFP_elliptic.cpp:700-765:
// HydroState::calc_current_and_charge, divide by ep, mult(-cd_scale))
__global__ void k_ell_charge(Geo g, ChargeArgs a, double* __restrict__ S) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long ncell = (long)g.n[0] * g.n[1] * g.n[2];
    if (t >= ncell) return;
    const int i = (int)(t % g.n[0]), j = (int)((t / g.n[0]) % g.n[1]), k = (int)(t / ((long)g.n[0] * g.n[1]));
    const long c = g.at(i, j, k);
    double cd = 0.0;
    for (int s = 0; s < a.ns; ++s) {
        const double rho = a.rho[s][c];
        double r = a.r[s];
        const cmfp::SpeciesMix& M = a.mix[s];
        if (M.nt) {  // get_charge_from_cons / get_mass_from_cons (MFP_hydro_gas.cpp:72-96, 123-150)
            double S_aq = 0.0, S_am = 0.0, S_a = 0.0;
            for (int i = 0; i < 2; ++i)
                if (i < M.nt) {
                    double al = a.rho[s][c + (5 + i) * a.sc] / rho;
                    al = (al < 0.0) ? 0.0 : ((1.0 < al) ? 1.0 : al);
                    S_aq += al * M.q[i] / M.m[i];
                    S_am += al / M.m[i];
                    S_a += al;
                }
            S_aq += (1.0 - S_a) * M.q[M.nt] / M.m[M.nt];
            S_am += (1.0 - S_a) / M.m[M.nt];
            const double m = (M.flags & 1) ? 1.0 / S_am : M.m[0];
            const double q = (M.flags & 2) ? S_aq / S_am : M.q[0];
            r = q / m;
        }
        cd += rho * r;
    }
    cd = cd / a.ep[c];
    S[t] = cd * (-a.cd_scale);
}

// rhs at node (i, j, k) = lower corner of cell (i, j, k), periodic: compRHS
__global__ void k_ell_rhs(Geo g, const double* __restrict__ V, const double* __restrict__ S, Cplx* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const long nn = (long)n0 * n1 * n2;
    if (t >= nn) return;
    const int i = (int)(t % n0), j = (int)((t / n0) % n1), k = (int)(t / ((long)n0 * n1));
    const int im = wrap(i - 1, n0), jm = wrap(j - 1, n1), km = g.dim == 3 ? wrap(k - 1, n2) : 0;
    const double* vx = V;
    const double* vy = V + g.sc;
    const double* vz = V + 2 * g.sc;
    double r;
    if (g.dim == 2) {
        const double fx = 0.5 / g.dx[0], fy = 0.5 / g.dx[1];
        r = fx * (-vx[g.at(im, jm, 0)] + vx[g.at(i, jm, 0)] - vx[g.at(im, j, 0)] + vx[g.at(i, j, 0)]) +
            fy * (-vy[g.at(im, jm, 0)] - vy[g.at(i, jm, 0)] + vy[g.at(im, j, 0)] + vy[g.at(i, j, 0)]);
        if (S) r += 0.25 * (S[(long)jm * n0 + im] + S[(long)jm * n0 + i] + S[(long)j * n0 + im] + S[(long)j * n0 + i]);
    } else {
        const double fx = 0.25 / g.dx[0], fy = 0.25 / g.dx[1], fz = 0.25 / g.dx[2];
        double sx = 0, sy = 0, sz = 0, ss = 0;
        const int I[2] = {im, i}, J[2] = {jm, j}, K[2] = {km, k};
        for (int c = 0; c < 2; ++c)
            for (int b = 0; b < 2; ++b)
                for (int a = 0; a < 2; ++a) {
                    const long q = g.at(I[a], J[b], K[c]);
                    sx += (a ? 1.0 : -1.0) * vx[q];
                    sy += (b ? 1.0 : -1.0) * vy[q];
                    sz += (c ? 1.0 : -1.0) * vz[q];
                    if (S) ss += S[((long)K[c] * n1 + J[b]) * n0 + I[a]];
                }
        r = fx * sx + fy * sy + fz * sz;
        if (S) r += 0.125 * ss;
    }
    out[t].re = r;
    out[t].im = 0.0;
}

// radix-2 FFT of one line of N = 2^lg complex values per workgroup, in LDS (N <= 4096: 64 KB)
__global__ __launch_bounds__(256) void k_ell_fft(Cplx* __restrict__ data, int N, int lg, long elem_stride, int na, long stride_a,
                                                  long stride_b, const Cplx* __restrict__ tw, int inverse) {
    extern __shared__ double lds[];
    Cplx* x = reinterpret_cast<Cplx*>(lds);
    const int line = blockIdx.x;
    const long base = (long)(line % na) * stride_a + (long)(line / na) * stride_b;
    for (int t = threadIdx.x; t < N; t += blockDim.x) {
        const unsigned rev = __brev((unsigned)t) >> (32 - lg);
        x[rev] = data[base + (long)t * elem_stride];
    }
    __syncthreads();
    for (int s = 1; s <= lg; ++s) {
        const int half = 1 << (s - 1);
        for (int b = threadIdx.x; b < N / 2; b += blockDim.x) {
            const int grp = b >> (s - 1), jj = b & (half - 1);
            const int p = (grp << s) + jj, q = p + half;
            Cplx w = tw[(long)jj << (lg - s)];  // exp(-2 pi i jj / 2^s)
            if (inverse) w.im = -w.im;
            const Cplx u = x[p], v = x[q];
            const double tr = v.re * w.re - v.im * w.im, ti = v.re * w.im + v.im * w.re;
            x[p].re = u.re + tr;
            x[p].im = u.im + ti;
            x[q].re = u.re - tr;
            x[q].im = u.im - ti;
        }
        __syncthreads();
    }
    for (int t = threadIdx.x; t < N; t += blockDim.x) data[base + (long)t * elem_stride] = x[t];
}

// phi_hat = rhs_hat / symbol(L) / N_total (the unnormalised inverse transform follows); the mean mode is set to 0
__global__ void k_ell_solve(Geo g, Cplx* __restrict__ hat) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const long nn = (long)n0 * n1 * n2;
    if (t >= nn) return;
    const int i = (int)(t % n0), j = (int)((t / n0) % n1), k = (int)(t / ((long)n0 * n1));
    const double PI2 = 6.283185307179586;
    const double cx = cos(PI2 * i / n0), cy = cos(PI2 * j / n1), cz = g.dim == 3 ? cos(PI2 * k / n2) : 1.0;
    // 1-D stiffness (2 - 2 cos) / h^2 and mass (4 + 2 cos) / 6 symbols of the linear elements
    const double Kx = (2.0 - 2.0 * cx) / (g.dx[0] * g.dx[0]), Ky = (2.0 - 2.0 * cy) / (g.dx[1] * g.dx[1]);
    const double Mx = (4.0 + 2.0 * cx) / 6.0, My = (4.0 + 2.0 * cy) / 6.0;
    double sym;
    if (g.dim == 2) {
        sym = -(Kx * My + Mx * Ky);
    } else {
        const double Kz = (2.0 - 2.0 * cz) / (g.dx[2] * g.dx[2]), Mz = (4.0 + 2.0 * cz) / 6.0;
        sym = -(Kx * My * Mz + Mx * Ky * Mz + Mx * My * Kz);
    }
    const double f = (i == 0 && j == 0 && k == 0) ? 0.0 : 1.0 / (sym * (double)nn);
    hat[t].re *= f;
    hat[t].im *= f;
}

// V -= G phi at the cell centres; phi_cc = average of the cell's nodes
__global__ void k_ell_apply(Geo g, double* __restrict__ V, double* __restrict__ phi_cc, const Cplx* __restrict__ phi) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const long nn = (long)n0 * n1 * n2;
    if (t >= nn) return;
    const int i = (int)(t % n0), j = (int)((t / n0) % n1), k = (int)(t / ((long)n0 * n1));
    const int ip = wrap(i + 1, n0), jp = wrap(j + 1, n1), kp = g.dim == 3 ? wrap(k + 1, n2) : 0;
    const long c = g.at(i, j, k);
    auto P = [&](int a, int b, int cc) { return phi[((long)cc * n1 + b) * n0 + a].re; };
    if (g.dim == 2) {
        const double p00 = P(i, j, 0), p10 = P(ip, j, 0), p01 = P(i, jp, 0), p11 = P(ip, jp, 0);
        const double gx = (0.5 / g.dx[0]) * (-p00 + p10 - p01 + p11);
        const double gy = (0.5 / g.dx[1]) * (-p00 - p10 + p01 + p11);
        V[c] += -gx;
        V[c + g.sc] += -gy;
        phi_cc[c] = 0.25 * (p00 + p10 + p01 + p11);
    } else {
        const int I[2] = {i, ip}, J[2] = {j, jp}, K[2] = {k, kp};
        double gx = 0, gy = 0, gz = 0, av = 0;
        for (int cc = 0; cc < 2; ++cc)
            for (int b = 0; b < 2; ++b)
                for (int a = 0; a < 2; ++a) {
                    const double p = P(I[a], J[b], K[cc]);
                    gx += (a ? 1.0 : -1.0) * p;
                    gy += (b ? 1.0 : -1.0) * p;
                    gz += (cc ? 1.0 : -1.0) * p;
                    av += p;
                }
        V[c] += -(0.25 / g.dx[0]) * gx;
        V[c + g.sc] += -(0.25 / g.dx[1]) * gy;
        V[c + 2 * g.sc] += -(0.25 / g.dx[2]) * gz;
        phi_cc[c] = 0.125 * av;
    }
}

static bool pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
static int ilog2(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}

const char* check(const Geo& g, const int periodic[3], const int split[3]) {
    if (g.dim < 2) return "the nodal projection is not defined in 1-D (MFP_elliptic.cpp:640)";
    for (int d = 0; d < g.dim; ++d) {
        if (!periodic[d]) return "divergence projection: periodic dimensions only (wall / outflow faces need the AMReX solver's boundary stencils)";
        if (split[d]) return "divergence projection: the block must span the domain (single rank)";
        if (!pow2(g.n[d]) || g.n[d] > 4096 || g.n[d] < 2) return "divergence projection: cells per dimension must be a power of two in [2, 4096]";
    }
    return nullptr;
}

hipError_t setup(const Geo& g, Work& w) {
    const long nn = (long)g.n[0] * g.n[1] * g.n[2];
    hipError_t e;
    if (!w.nodes && (e = hipMalloc((void**)&w.nodes, sizeof(Cplx) * nn)) != hipSuccess) return e;
    if (!w.S && (e = hipMalloc((void**)&w.S, sizeof(double) * nn)) != hipSuccess) return e;
    for (int d = 0; d < g.dim; ++d) {
        if (w.tw[d]) continue;
        const int N = g.n[d];
        std::vector<Cplx> h(N / 2 > 0 ? N / 2 : 1);
        for (int q = 0; q < N / 2; ++q) {
            const double a = -6.283185307179586476925286766559 * q / N;
            h[q].re = std::cos(a);
            h[q].im = std::sin(a);
        }
        if ((e = hipMalloc((void**)&w.tw[d], sizeof(Cplx) * h.size())) != hipSuccess) return e;
        if ((e = hipMemcpy(w.tw[d], h.data(), sizeof(Cplx) * h.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
    }
    return hipSuccess;
}

void release(Work& w) {
    if (w.nodes) (void)hipFree(w.nodes);
    if (w.S) (void)hipFree(w.S);
    for (int d = 0; d < 3; ++d)
        if (w.tw[d]) (void)hipFree(w.tw[d]);
    w = Work();
}

static void fft_all(const Geo& g, Work& w, int inverse, hipStream_t st) {
    const long str[3] = {1, g.n[0], (long)g.n[0] * g.n[1]};
    for (int d = 0; d < g.dim; ++d) {
        const int N = g.n[d];
        const int a = (d + 1) % 3, b = (d + 2) % 3;
        const int na = g.n[a], nb = g.n[b];
        const int threads = N / 2 < 256 ? (N / 2 < 64 ? 64 : N / 2) : 256;
        hipLaunchKernelGGL(k_ell_fft, dim3(na * nb), dim3(threads), sizeof(Cplx) * N, st, w.nodes, N, ilog2(N), str[d], na, str[a], str[b],
                           w.tw[d], inverse);
    }
}

// one solve_divergence (MFP_elliptic.cpp:784-964): V = the three components starting at V (component stride g.sc),
// phi_cc = the cleaning potential's component, charge = nullptr for the B field
void project(const Geo& g, Work& w, double* V, double* phi_cc, const ChargeArgs* charge, hipStream_t st) {
    const long nn = (long)g.n[0] * g.n[1] * g.n[2];
    const dim3 grid((unsigned)((nn + 255) / 256)), block(256);
    if (charge) hipLaunchKernelGGL(k_ell_charge, grid, block, 0, st, g, *charge, w.S);
    hipLaunchKernelGGL(k_ell_rhs, grid, block, 0, st, g, V, charge ? w.S : nullptr, w.nodes);
    fft_all(g, w, 0, st);
    hipLaunchKernelGGL(k_ell_solve, grid, block, 0, st, g, w.nodes);
    fft_all(g, w, 1, st);
    hipLaunchKernelGGL(k_ell_apply, grid, block, 0, st, g, V, phi_cc, w.nodes);
}

}  // namespace cmfp_ell
