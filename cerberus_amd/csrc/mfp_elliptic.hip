// mfp_elliptic.hip — divergence projection of the Maxwell state on a uniform block:
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
// reference's own comment says (MFP_elliptic.cpp:806).
//
// Domain faces (set_field_bcs, MFP_elliptic.cpp:83-122: outflow / symmetry -> Neumann, inflow -> homogeneous
// Dirichlet).  AMReX mirrors the ghost nodes of a Neumann face (the boundary row is the doubled one-sided stencil),
// keeps phi = 0 on a Dirichlet face, and in compRHS lets the cells beyond a Neumann face contribute their normal
// component only (mlndlap_divu) and doubles the node's value per Neumann face (mlndlap_impose_neumann_bc).  With
// those rules the operator on the n + 1 nodes of a direction equals the periodic operator on the mirrored
// extension of that direction (2 n points for two faces of one kind, 4 n for one of each; negative mirror image
// about a Dirichlet face), which is diagonal in Fourier space: the right-hand side is evaluated on the true nodes
// with the boundary rules, copied onto the extension with the mirror signs, and the solve is the periodic one.
// Exact to rounding (the reference iterates MLMG to reltol 1e-8): parity with the reference is unpinned.
// Radix-2 FFT in LDS (one line per workgroup) where the extension length is a power of two, else a direct
// O(P^2) transform of the line; at most 8192 points per line.
#include <hip/hip_runtime.h>

#include <cmath>
#include <vector>

#include "../../include/cerberus_mfp.h"
#include "mfp_elliptic.h"

namespace cmfp_ell {
using cmfp::Geo;

__device__ __forceinline__ int wrap(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// point e of a direction's periodic extension -> true node (0 .. n; 0 .. n - 1 where periodic) and mirror sign
__device__ __forceinline__ void ext_map(int e, int n, int lo, int hi, int& node, double& sgn) {
    sgn = 1.0;
    if (lo == PERIODIC) {
        node = e;
        return;
    }
    if (lo == hi) {  // period 2 n
        if (e <= n) {
            node = e;
        } else {
            node = 2 * n - e;
            if (lo == DIRICHLET) sgn = -1.0;
        }
        return;
    }
    // one face of each kind: period 4 n; mirror about node 0 first, then about node n
    int e1 = e;
    if (e > 2 * n) {
        e1 = 4 * n - e;
        if (lo == DIRICHLET) sgn = -1.0;
    }
    if (e1 <= n) {
        node = e1;
    } else {
        node = 2 * n - e1;
        if (hi == DIRICHLET) sgn = -sgn;
    }
}

// S_cc = -cd_scale * (sum_s rho_s q_s / m_s) / ep on the valid cells (MFP_elliptic.cpp:722-766:
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

// compRHS on the points of the extension: the value of the true node (its cells: lower corner of cell (i, j, k)
// is node (i, j, k)) times the mirror sign.  V carries one filled ghost layer where the domain has faces.
__global__ void k_ell_rhs(Geo g, Grid E, const double* __restrict__ V, const double* __restrict__ S, Cplx* __restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E.total()) return;
    int e[3] = {(int)(t % E.P[0]), (int)((t / E.P[0]) % E.P[1]), (int)(t / ((long)E.P[0] * E.P[1]))};
    int node[3] = {0, 0, 0};
    double sgn = 1.0, twice = 1.0;
    bool masked = false;
    for (int d = 0; d < E.dim; ++d) {
        double s;
        ext_map(e[d], E.n[d], E.bc[d][0], E.bc[d][1], node[d], s);
        sgn *= s;
        if (E.bc[d][0] != PERIODIC) {
            if (node[d] == 0) {
                if (E.bc[d][0] == DIRICHLET) masked = true;
                else twice *= 2.0;  // mlndlap_impose_neumann_bc
            } else if (node[d] == E.n[d]) {
                if (E.bc[d][1] == DIRICHLET) masked = true;
                else twice *= 2.0;
            }
        }
    }
    double r = 0.0;
    if (!masked) {
        const double fac = E.dim == 2 ? 0.5 : 0.25;
        double sum[3] = {0.0, 0.0, 0.0}, ss = 0.0;
        const int ncorner = 1 << E.dim;
        for (int q = 0; q < ncorner; ++q) {
            int c[3] = {0, 0, 0};
            bool outside[3] = {false, false, false};
            for (int d = 0; d < E.dim; ++d) {
                const int a = (q >> d) & 1;
                c[d] = node[d] - 1 + a;
                if (E.bc[d][0] == PERIODIC) c[d] = wrap(c[d], E.n[d]);
                else outside[d] = c[d] < 0 || c[d] >= E.n[d];  // a ghost cell beyond a Neumann face
            }
            const long cell = g.at(c[0], c[1], c[2]);
            bool any_out = false;
            for (int d = 0; d < E.dim; ++d) any_out = any_out || outside[d];
            for (int d = 0; d < E.dim; ++d) {
                // the nodal divergence does not see the tangential components of a cell beyond the face (mlndlap_divu)
                bool other_out = false;
                for (int f = 0; f < E.dim; ++f)
                    if (f != d && outside[f]) other_out = true;
                if (other_out) continue;
                const double v = V[cell + d * g.sc];
                sum[d] += ((q >> d) & 1) ? v : -v;
            }
            if (S && !any_out) ss += S[((long)c[2] * g.n[1] + c[1]) * g.n[0] + c[0]];  // the source is zero beyond the domain
        }
        for (int d = 0; d < E.dim; ++d) r += (fac / E.dx[d]) * sum[d];
        if (S) r += (E.dim == 2 ? 0.25 : 0.125) * ss;
        r *= twice * sgn;
    }
    out[t].re = r;
    out[t].im = 0.0;
}

// radix-2 FFT of one line of N = 2^lg complex values per workgroup, in LDS
__global__ __launch_bounds__(256) void k_ell_fft(Cplx* __restrict__ data, int N, int lg, long elem_stride, int na, long stride_a,
                                                  long stride_b, const Cplx* __restrict__ tw, int inverse) {
    extern __shared__ double lds[];
    Cplx* x = reinterpret_cast<Cplx*>(lds);
    const int line = blockIdx.x;
    const long base = (long)(line % na) * stride_a + (long)(line / na) * stride_b;
    for (int t = threadIdx.x; t < N; t += blockDim.x) {
        const unsigned rev = lg ? __brev((unsigned)t) >> (32 - lg) : 0u;
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

// direct transform of one line of any length N (tw: exp(-2 pi i m / N), m < N): O(N^2) per line
__global__ __launch_bounds__(256) void k_ell_dft(Cplx* __restrict__ data, int N, long elem_stride, int na, long stride_a, long stride_b,
                                                  const Cplx* __restrict__ tw, int inverse) {
    extern __shared__ double lds[];
    Cplx* x = reinterpret_cast<Cplx*>(lds);
    const int line = blockIdx.x;
    const long base = (long)(line % na) * stride_a + (long)(line / na) * stride_b;
    for (int t = threadIdx.x; t < N; t += blockDim.x) x[t] = data[base + (long)t * elem_stride];
    __syncthreads();
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        double ar = 0.0, ai = 0.0;
        int m = 0;  // (j k) mod N
        for (int j = 0; j < N; ++j) {
            Cplx w = tw[m];
            if (inverse) w.im = -w.im;
            const Cplx v = x[j];
            ar += v.re * w.re - v.im * w.im;
            ai += v.re * w.im + v.im * w.re;
            m += k;
            if (m >= N) m -= N;
        }
        data[base + (long)k * elem_stride].re = ar;  // the line was copied to LDS: nobody reads `data` any more
        data[base + (long)k * elem_stride].im = ai;
    }
}

// phi_hat = rhs_hat / symbol(L) / N_total (the unnormalised inverse transform follows); the mean mode is set to 0
__global__ void k_ell_solve(Grid E, Cplx* __restrict__ hat) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nn = E.total();
    if (t >= nn) return;
    const int i = (int)(t % E.P[0]), j = (int)((t / E.P[0]) % E.P[1]), k = (int)(t / ((long)E.P[0] * E.P[1]));
    const double PI2 = 6.283185307179586;
    const double cx = cos(PI2 * i / E.P[0]), cy = cos(PI2 * j / E.P[1]), cz = E.dim == 3 ? cos(PI2 * k / E.P[2]) : 1.0;
    // 1-D stiffness (2 - 2 cos) / h^2 and mass (4 + 2 cos) / 6 symbols of the linear elements
    const double Kx = (2.0 - 2.0 * cx) / (E.dx[0] * E.dx[0]), Ky = (2.0 - 2.0 * cy) / (E.dx[1] * E.dx[1]);
    const double Mx = (4.0 + 2.0 * cx) / 6.0, My = (4.0 + 2.0 * cy) / 6.0;
    double sym;
    if (E.dim == 2) {
        sym = -(Kx * My + Mx * Ky);
    } else {
        const double Kz = (2.0 - 2.0 * cz) / (E.dx[2] * E.dx[2]), Mz = (4.0 + 2.0 * cz) / 6.0;
        sym = -(Kx * My * Mz + Mx * Ky * Mz + Mx * My * Kz);
    }
    const double f = (i == 0 && j == 0 && k == 0) ? 0.0 : 1.0 / (sym * (double)nn);
    hat[t].re *= f;
    hat[t].im *= f;
}

// V -= G phi at the cell centres; phi_cc = average of the cell's nodes
__global__ void k_ell_apply(Geo g, Grid E, double* __restrict__ V, double* __restrict__ phi_cc, const Cplx* __restrict__ phi) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const long nc = (long)n0 * n1 * n2;
    if (t >= nc) return;
    const int c3[3] = {(int)(t % n0), (int)((t / n0) % n1), (int)(t / ((long)n0 * n1))};
    const long c = g.at(c3[0], c3[1], c3[2]);
    double gsum[3] = {0.0, 0.0, 0.0}, av = 0.0;
    const int ncorner = 1 << E.dim;
    for (int q = 0; q < ncorner; ++q) {
        int nd[3] = {0, 0, 0};
        for (int d = 0; d < E.dim; ++d) {
            nd[d] = c3[d] + ((q >> d) & 1);
            if (E.bc[d][0] == PERIODIC) nd[d] = wrap(nd[d], E.n[d]);  // else node n is point n of the extension
        }
        const double p = phi[((long)nd[2] * E.P[1] + nd[1]) * E.P[0] + nd[0]].re;
        av += p;
        for (int d = 0; d < E.dim; ++d) gsum[d] += ((q >> d) & 1) ? p : -p;
    }
    const double fac = E.dim == 2 ? 0.5 : 0.25;
    for (int d = 0; d < E.dim; ++d) V[c + d * g.sc] += -(fac / E.dx[d]) * gsum[d];
    phi_cc[c] = (E.dim == 2 ? 0.25 : 0.125) * av;
}

static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
static int ilog2(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}
static constexpr int MAX_LINE = 8192;  // points per transformed line: 128 KB of LDS

const char* make_grid(const Geo& g, const int periodic[3], const int split[3], const int phys_bc[3][2], Grid& E) {
    E = Grid();
    E.dim = g.dim;
    for (int d = 0; d < 3; ++d) {
        E.n[d] = d < g.dim ? g.n[d] : 1;
        E.P[d] = 1;
        E.pow2[d] = 1;
        E.dx[d] = g.dx[d];
        E.bc[d][0] = E.bc[d][1] = PERIODIC;
    }
    if (g.dim < 2) return "the nodal projection is not defined in 1-D (MFP_elliptic.cpp:640)";
    for (int d = 0; d < g.dim; ++d) {
        if (split[d]) return "divergence projection: the block must span the domain (single rank)";
        if (g.n[d] < 2) return "divergence projection: at least two cells per dimension";
        if (periodic[d]) {
            E.P[d] = g.n[d];
        } else {
            if (g.ng[d] < 1) return "divergence projection: the field needs a ghost layer on a domain with physical faces";
            for (int side = 0; side < 2; ++side) {
                switch (phys_bc[d][side]) {  // set_field_bcs, MFP_elliptic.cpp:93-119
                case CMFP_PHYSBC_OUTFLOW:
                case CMFP_PHYSBC_SYMMETRY: E.bc[d][side] = NEUMANN; break;
                case CMFP_PHYSBC_INFLOW: E.bc[d][side] = DIRICHLET; break;
                case CMFP_PHYSBC_ASYMMETRY:
                    return "divergence projection: an 'asymmetry' face maps to LinOpBCType::reflect_odd in the reference "
                           "(MFP_elliptic.cpp:102-104 through the value 4 of MFP_field.cpp:40), which AMReX's nodal solver does not define";
                default: return "divergence projection: a non-periodic face needs one of inflow / outflow / symmetry";
                }
            }
            E.P[d] = (E.bc[d][0] == E.bc[d][1] ? 2 : 4) * g.n[d];
        }
        E.pow2[d] = is_pow2(E.P[d]);
        if (E.P[d] > MAX_LINE)
            return "divergence projection: more than 8192 points per transformed line (cells of a periodic direction, twice the "
                   "cells of a direction between two faces of one kind, four times between a Neumann and a Dirichlet face)";
    }
    return nullptr;
}

hipError_t setup(const Grid& E, const Geo& g, Work& w) {
    const long nn = E.total();
    hipError_t e;
    if (w.nodes_cap < nn) {
        if (w.nodes) (void)hipFree(w.nodes);
        w.nodes = nullptr;
        w.nodes_cap = 0;
        if ((e = hipMalloc((void**)&w.nodes, sizeof(Cplx) * nn)) != hipSuccess) return e;
        w.nodes_cap = nn;
    }
    const long ncell = (long)g.n[0] * g.n[1] * g.n[2];
    if (!w.S && (e = hipMalloc((void**)&w.S, sizeof(double) * ncell)) != hipSuccess) return e;
    for (int d = 0; d < E.dim; ++d) {
        const int N = E.P[d];
        const int len = E.pow2[d] ? (N / 2 > 0 ? N / 2 : 1) : N;  // radix-2: exp(-2 pi i q / N), q < N / 2; direct: q < N
        if (w.tw[d] && w.tw_len[d] == (E.pow2[d] ? -N : N)) continue;
        if (w.tw[d]) (void)hipFree(w.tw[d]);
        w.tw[d] = nullptr;
        std::vector<Cplx> h((size_t)len);
        for (int q = 0; q < len; ++q) {
            const double a = -6.283185307179586476925286766559 * q / N;
            h[q].re = std::cos(a);
            h[q].im = std::sin(a);
        }
        if ((e = hipMalloc((void**)&w.tw[d], sizeof(Cplx) * h.size())) != hipSuccess) return e;
        if ((e = hipMemcpy(w.tw[d], h.data(), sizeof(Cplx) * h.size(), hipMemcpyHostToDevice)) != hipSuccess) return e;
        w.tw_len[d] = E.pow2[d] ? -N : N;
    }
    // lines of more than 4096 points need more than the default 64 KB of dynamic LDS
    static bool attr = false;
    if (!attr) {
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ell_fft), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(sizeof(Cplx) * MAX_LINE))) != hipSuccess)
            return e;
        if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ell_dft), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)(sizeof(Cplx) * MAX_LINE))) != hipSuccess)
            return e;
        attr = true;
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

static void fft_all(const Grid& E, Work& w, int inverse, hipStream_t st) {
    const long str[3] = {1, E.P[0], (long)E.P[0] * E.P[1]};
    for (int d = 0; d < E.dim; ++d) {
        const int N = E.P[d];
        const int a = (d + 1) % 3, b = (d + 2) % 3;
        const int na = E.P[a], nb = E.P[b];
        if (E.pow2[d]) {
            const int threads = N / 2 < 256 ? (N / 2 < 64 ? 64 : N / 2) : 256;
            hipLaunchKernelGGL(k_ell_fft, dim3(na * nb), dim3(threads), sizeof(Cplx) * N, st, w.nodes, N, ilog2(N), str[d], na, str[a], str[b],
                               w.tw[d], inverse);
        } else {
            hipLaunchKernelGGL(k_ell_dft, dim3(na * nb), dim3(256), sizeof(Cplx) * N, st, w.nodes, N, str[d], na, str[a], str[b], w.tw[d],
                               inverse);
        }
    }
}

// one solve_divergence (MFP_elliptic.cpp:784-964): V = the three components starting at V (component stride g.sc),
// phi_cc = the cleaning potential's component, charge = nullptr for the B field
void project(const Geo& g, const Grid& E, Work& w, double* V, double* phi_cc, const ChargeArgs* charge, hipStream_t st) {
    const long nn = E.total(), nc = (long)g.n[0] * g.n[1] * g.n[2];
    const dim3 block(256), grid_n((unsigned)((nn + 255) / 256)), grid_c((unsigned)((nc + 255) / 256));
    if (charge) hipLaunchKernelGGL(k_ell_charge, grid_c, block, 0, st, g, *charge, w.S);
    hipLaunchKernelGGL(k_ell_rhs, grid_n, block, 0, st, g, E, V, charge ? w.S : nullptr, w.nodes);
    fft_all(E, w, 0, st);
    hipLaunchKernelGGL(k_ell_solve, grid_n, block, 0, st, E, w.nodes);
    fft_all(E, w, 1, st);
    hipLaunchKernelGGL(k_ell_apply, grid_c, block, 0, st, g, E, V, phi_cc, w.nodes);
}

}  // namespace cmfp_ell
