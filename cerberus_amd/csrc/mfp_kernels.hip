// mfp_kernels.hip — stage-by-stage HIP kernels (gfx950) of the hyperbolic
// update, one per FAB-level function of the reference, plus sources, dt
// reductions, ghost fill / halo pack and the batched leaf kernels.
//
// Compiled twice: -DCMFP_NS=strict -ffp-contract=off and -DCMFP_NS=fast
// -ffp-contract=fast.  All arrays are fp64 SoA (i fastest, component slowest)
// with the state's ghost width on every array, so one thread per cell gives
// fully coalesced 8-byte lane accesses on every component plane.
#include "mfp_device.h"
#include "mfp_source.h"

#include <algorithm>
#include <cfloat>

namespace CMFP_NS {

using namespace cmfp;

static constexpr int TPB = 256;
static inline unsigned nblocks(long n) { return (unsigned)((n + TPB - 1) / TPB); }

__device__ __forceinline__ bool decode(const IBox& b, long t, int& i, int& j, int& k) {
    const int l0 = b.hi[0] - b.lo[0] + 1, l1 = b.hi[1] - b.lo[1] + 1, l2 = b.hi[2] - b.lo[2] + 1;
    if (t >= (long)l0 * l1 * l2) return false;
    i = b.lo[0] + (int)(t % l0);
    t /= l0;
    j = b.lo[1] + (int)(t % l1);
    k = b.lo[2] + (int)(t / l1);
    return true;
}

static IBox valid_box(const Geo& g) {
    IBox b;
    for (int d = 0; d < 3; ++d) {
        b.lo[d] = 0;
        b.hi[d] = g.n[d] - 1;
    }
    return b;
}
static IBox grown_box(const Geo& g, int n) {
    IBox b = valid_box(g);
    for (int d = 0; d < g.dim; ++d) {
        b.lo[d] -= n;
        b.hi[d] += n;
    }
    return b;
}

// ------------------------------------------------------------ calc_primitives
// EulerianState::calc_primitives MFP_eulerian.cpp:313-433 over grow(box, num_grow).
template <bool HYDRO>
__global__ void k_prim(Geo g, IBox box, HydroParams hp, const double* __restrict__ U, double* __restrict__ Q, int* err) {
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long t = g.at(i, j, k);
    if (HYDRO) {
        double u[5], q[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) u[n] = U[t + n * g.sc];
        GasP G = hp.gas;
        if (hp.mix) {  // get_gamma_from_cons / get_mass_from_cons: alpha_i = tracer_i / rho
            double al[2] = {0.0, 0.0};
            for (int n = 0; n < hp.nt; ++n) al[n] = f_div(U[t + (5 + n) * g.sc], u[0]);
            G = mix_gas(hp, al);
            Q[t + (5 + hp.nt) * g.sc] = G.gamma;  // Q[Gamma], MFP_thermally_perfect_gas.cpp:210
        }
        if (!hydro_cons2prim(G, u, q)) *err = 1;
#pragma unroll
        for (int n = 0; n < 5; ++n) Q[t + n * g.sc] = q[n];
        if (hp.nt) {  // alpha = tracer * rhoinv (MFP_thermally_perfect_gas.cpp:212-214)
            const double rhoinv = f_rcp(u[0]);
            for (int n = 5; n < 5 + hp.nt; ++n) Q[t + n * g.sc] = U[t + n * g.sc] * rhoinv;
        }
    } else {
#pragma unroll
        for (int n = 0; n < 10; ++n) Q[t + n * g.sc] = U[t + n * g.sc];
    }
}

// ------------------------------------------ calc_reconstruction + time average
// HydroState::calc_reconstruction MFP_hydro.cpp:780-1038 /
// EulerianState::calc_reconstruction MFP_eulerian.cpp:513-607, then (do_ctu)
// EulerianState::calc_time_averaged_faces MFP_eulerian.cpp:609-676, on grow(box,1).
template <bool HYDRO, int RECON>
__global__ void k_recon(Geo g, IBox box, HydroParams hp, FieldParams fp, const double* __restrict__ Q, Ptr3 rlo,
                        Ptr3 rhi, int do_ctu, double dt) {
    constexpr int SL = ReconTraits<RECON>::SL;
    constexpr int off = SL / 2;
    constexpr int NQ = HYDRO ? 5 : 10;
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long c0 = g.at(i, j, k);
    const long str[3] = {1, g.sj, g.sk};

    double centre[NQ];
#pragma unroll
    for (int n = 0; n < NQ; ++n) centre[n] = Q[c0 + n * g.sc];
    double speed[3];
    if (HYDRO) {
        // the reference stores eps = p/(gamma-1) in place during reconstruction and
        // restores p = eps*(gamma-1) afterwards (MFP_hydro.cpp:847-862,1025-1035):
        // the time-average step sees that round-tripped pressure
        GasP Gc = hp.gas;
        if (hp.mix) {  // get_gamma_from_prim(Q): from the cell's alpha (:857, MFP_thermally_perfect_gas.cpp:323)
            double al[2] = {0.0, 0.0};
            for (int n = 0; n < hp.nt; ++n) al[n] = Q[c0 + (5 + n) * g.sc];
            Gc = mix_gas(hp, al);
        }
        const double gm1 = Gc.gamma - 1.0;
        centre[H_P] = (centre[H_P] / gm1) * gm1;
        hydro_speed_from_prim(Gc, centre, speed);
    } else {
        speed[0] = speed[1] = speed[2] = fp.fastest;
    }

    for (int d = 0; d < g.dim; ++d) {
        double lo[NQ], hi[NQ];
        double gam_lo = 0.0, gam_hi = 0.0;  // mixtures: gamma of the two face states (:997-998)
        if (HYDRO) {
            double st[SL][5];
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                const long c = c0 + (long)(s - off) * str[d];
#pragma unroll
                for (int n = 0; n < 5; ++n) st[s][n] = Q[c + n * g.sc];
                double gm1 = hp.gas.gamma - 1.0;
                if (hp.mix) {  // every cell of the stencil with its own gamma (:857-859)
                    double al[2] = {0.0, 0.0};
                    for (int n = 0; n < hp.nt; ++n) al[n] = Q[c + (5 + n) * g.sc];
                    gm1 = mix_gas(hp, al).gamma - 1.0;
                }
                st[s][H_P] = st[s][H_P] / gm1;
            }
            if (!hp.mix) {
                hydro_cell_faces<RECON>(hp.gas.gamma, st, lo, hi);
            } else {
                // gamma = 2 leaves eps -+ phi_eps / 2 in the pressure slots ((2 - 1) x is exact); the faces' own
                // gamma comes from their alpha (:982-1002)
                hydro_cell_faces<RECON>(2.0, st, lo, hi);
                double alo[2] = {0.0, 0.0}, ahi[2] = {0.0, 0.0};
                for (int n = 0; n < hp.nt; ++n) {
                    double sa[SL];
#pragma unroll
                    for (int s = 0; s < SL; ++s) sa[s] = Q[c0 + (long)(s - off) * str[d] + (5 + n) * g.sc];
                    const double phi_alpha = recon_slope<RECON>(sa);
                    alo[n] = sa[off] - 0.5 * phi_alpha;
                    ahi[n] = sa[off] + 0.5 * phi_alpha;
                }
                gam_lo = mix_gas(hp, alo).gamma;
                gam_hi = mix_gas(hp, ahi).gamma;
                lo[H_P] = lo[H_P] * (gam_lo - 1.0);
                hi[H_P] = hi[H_P] * (gam_hi - 1.0);
            }
        } else {
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                double st[SL];
#pragma unroll
                for (int s = 0; s < SL; ++s) st[s] = Q[c0 + (long)(s - off) * str[d] + n * g.sc];
                recon_faces<RECON>(st, lo[n], hi[n]);
            }
        }
        if (do_ctu) {
            const double dt_2dx = 0.5 * dt / g.dx[d];
#pragma unroll
            for (int n = 0; n < NQ; ++n) trace_faces(speed[d], dt_2dx, centre[n], lo[n], hi[n]);
        }
        double* __restrict__ plo = rlo.p[d];
        double* __restrict__ phi = rhi.p[d];
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            plo[c0 + n * g.sc] = lo[n];
            phi[c0 + n * g.sc] = hi[n];
        }
        if (HYDRO) {
            // passive scalars: alpha -+ slope / 2 (MFP_hydro.cpp:982-995), traced like every primitive
            for (int n = 5; n < 5 + hp.nt; ++n) {
                double st[SL];
#pragma unroll
                for (int s = 0; s < SL; ++s) st[s] = Q[c0 + (long)(s - off) * str[d] + n * g.sc];
                const double phi_alpha = recon_slope<RECON>(st);
                double alo = st[off] - 0.5 * phi_alpha;
                double ahi = st[off] + 0.5 * phi_alpha;
                if (do_ctu) trace_faces(speed[d], 0.5 * dt / g.dx[d], st[off], alo, ahi);
                plo[c0 + n * g.sc] = alo;
                phi[c0 + n * g.sc] = ahi;
            }
            if (hp.mix) {  // the Gamma entry of the face states, traced like every primitive (:1013-1014)
                const int GI = 5 + hp.nt;
                if (do_ctu) trace_faces(speed[d], 0.5 * dt / g.dx[d], Q[c0 + GI * g.sc], gam_lo, gam_hi);
                plo[c0 + GI * g.sc] = gam_lo;
                phi[c0 + GI * g.sc] = gam_hi;
            }
        }
    }
}

// ---------------------------------------------------------------- face_bc
// EulerianState::update_face_prim / face_bc, do_all branch
// (MFP_eulerian.cpp:882-955, 794-836) for one (d, side) domain face.
__global__ void k_face_bc(Geo g, IBox plane, int d, int side, int nq, FillRules rules, double* __restrict__ rlo,
                          double* __restrict__ rhi) {
    int i, j, k;
    if (!decode(plane, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long str[3] = {1, g.sj, g.sk};
    const long c = g.at(i, j, k);  // the cell whose low face is the plane
    const long cm = c - str[d];
    for (int n = 0; n < nq; ++n) {
        const int r = rules.rule[n];
        if (r == FILL_SKIP) continue;
        if (side == 0) {
            const double s = rlo[c + n * g.sc];
            rhi[cm + n * g.sc] = (r == FILL_ODD) ? -s : s;
        } else {
            const double s = rhi[cm + n * g.sc];
            rlo[c + n * g.sc] = (r == FILL_ODD) ? -s : s;
        }
    }
}

// ---------------------------------------------------------------- calc_fluxes
// EulerianState::calc_fluxes MFP_eulerian.cpp:957-1045 for direction d.
template <bool HYDRO, int FLUX>
__global__ void k_fluxes(Geo g, IBox fbox, int d, HydroParams hp, FieldParams fp, const double* __restrict__ rlo,
                         const double* __restrict__ rhi, double* __restrict__ flux) {
    constexpr int NQ = HYDRO ? 5 : 10;
    constexpr int NC = HYDRO ? 5 : 10;
    int i, j, k;
    if (!decode(fbox, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long str[3] = {1, g.sj, g.sk};
    const long c = g.at(i, j, k);
    const long cm = c - str[d];
    double L[NQ], R[NQ], F[NC];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        L[n] = rhi[cm + n * g.sc];
        R[n] = rlo[c + n * g.sc];
    }
    if (HYDRO && hp.nt > 0) {
        // hydro state with passive scalars: the same solve plus the tracer fluxes
        if constexpr (HYDRO && FLUX != 2) {
            double Lt[5 + 2], Rt[5 + 2], Ft[5 + 2];
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                Lt[n] = L[n];
                Rt[n] = R[n];
            }
            for (int n = 5; n < 5 + hp.nt; ++n) {
                Lt[n] = rhi[cm + n * g.sc];
                Rt[n] = rlo[c + n * g.sc];
            }
            rot_g2l(Lt, H_U, d);
            rot_g2l(Rt, H_U, d);
            double gamL = hp.gas.gamma, gamR = hp.gas.gamma;
            if (hp.mix) {  // L[Gamma], R[Gamma] (MFP_hllc.cpp:38,56)
                gamL = rhi[cm + (5 + hp.nt) * g.sc];
                gamR = rlo[c + (5 + hp.nt) * g.sc];
            }
            hydro_riemann_tracers<FLUX>(Lt, Rt, gamL, gamR, hp.nt, Ft);
            rot_l2g(Ft, 1, d);
            for (int n = 0; n < 5 + hp.nt; ++n) flux[c + n * g.sc] = 0.0 + Ft[n];
        }
        return;
    }
    if (HYDRO) {
        HydroPhys<1, FLUX>::riemann(hp, d, L, R, F);
    } else {
        FieldPhys<1>::riemann(fp, d, L, R, F);
    }
#pragma unroll
    for (int n = 0; n < NC; ++n) flux[c + n * g.sc] = 0.0 + F[n];  // flux.setVal(0); flux += F (:987,1035)
}

// ---------------------------------------------------------- correct_face_prim
// EulerianState::correct_face_prim MFP_eulerian.cpp:1047-1188 for face
// direction d1: one thread per face of box + e_d1; the (DIM-1) transverse
// corrections are applied one after the other as in the reference.
template <bool HYDRO>
__global__ void k_correct(Geo g, IBox faces, int d1, HydroParams hp, double* __restrict__ rlo, double* __restrict__ rhi,
                          Ptr3 flux, double dt, int* err) {
    constexpr int NQ = HYDRO ? 5 : 10;
    constexpr int NC = HYDRO ? 5 : 10;
    int i, j, k;
    if (!decode(faces, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long str[3] = {1, g.sj, g.sk};
    const long c = g.at(i, j, k);
    const long cm = c - str[d1];
    double Lp[NQ], Rp[NQ], Lc[NC], Rc[NC];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        Lp[n] = rhi[cm + n * g.sc];
        Rp[n] = rlo[c + n * g.sc];
    }
    const int nt = HYDRO ? hp.nt : 0;
    double La[2] = {0, 0}, Ra[2] = {0, 0};  // alpha of the two face states
    for (int t = 0; t < nt; ++t) {
        La[t] = rhi[cm + (5 + t) * g.sc];
        Ra[t] = rlo[c + (5 + t) * g.sc];
    }
    const bool mix = HYDRO && hp.mix;
    double Lg = 0.0, Rg = 0.0;  // mixtures: the Gamma entry of the two face states
    if (mix) {
        Lg = rhi[cm + (5 + nt) * g.sc];
        Rg = rlo[c + (5 + nt) * g.sc];
    }
    for (int dd = 1; dd < g.dim; ++dd) {
        const int d2 = (d1 + dd) % g.dim;
        const double* __restrict__ f = flux.p[d2];
        // passive scalars: tracer = alpha * rho, += the transverse flux difference, alpha = tracer * rhoinv
        // with the corrected density (prim2cons / cons2prim, MFP_thermally_perfect_gas.cpp:212-214,243-245)
        double Lt[2], Rt[2];
        for (int t = 0; t < nt; ++t) {
            const int n = 5 + t;
            Lt[t] = La[t] * Lp[H_RHO];
            Rt[t] = Ra[t] * Rp[H_RHO];
            Lt[t] += 0.5 * dt / g.dx[d2] * (f[cm + n * g.sc] - f[cm + str[d2] + n * g.sc]);
            Rt[t] += 0.5 * dt / g.dx[d2] * (f[c + n * g.sc] - f[c + str[d2] + n * g.sc]);
        }
        if (HYDRO) {
            GasP GL = hp.gas, GR = hp.gas;
            if (hp.mix) {  // get_gamma_from_prim(Q): the face state's alpha
                GL = mix_gas(hp, La);
                GR = mix_gas(hp, Ra);
            }
            hydro_prim2cons(GL, Lp, Lc);
            hydro_prim2cons(GR, Rp, Rc);
        } else {
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                Lc[n] = Lp[n];
                Rc[n] = Rp[n];
            }
        }
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            double loF = f[cm + n * g.sc];
            double hiF = f[cm + str[d2] + n * g.sc];
            Lc[n] += 0.5 * dt / g.dx[d2] * (loF - hiF);
        }
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            double loF = f[c + n * g.sc];
            double hiF = f[c + str[d2] + n * g.sc];
            Rc[n] += 0.5 * dt / g.dx[d2] * (loF - hiF);
        }
        if (HYDRO) {
            GasP GL = hp.gas, GR = hp.gas;
            if (hp.mix) {  // get_gamma_from_cons / get_mass_from_cons of the corrected state
                double al[2] = {0.0, 0.0}, ar[2] = {0.0, 0.0};
                for (int t = 0; t < nt; ++t) {
                    al[t] = f_div(Lt[t], Lc[0]);
                    ar[t] = f_div(Rt[t], Rc[0]);
                }
                GL = mix_gas(hp, al);
                GR = mix_gas(hp, ar);
                Lg = GL.gamma;
                Rg = GR.gamma;
            }
            if (!hydro_cons2prim(GL, Lc, Lp)) *err = 1;
            if (!hydro_cons2prim(GR, Rc, Rp)) *err = 1;
            if (nt) {
                const double rinvL = f_rcp(Lc[0]), rinvR = f_rcp(Rc[0]);
                for (int t = 0; t < nt; ++t) {
                    La[t] = Lt[t] * rinvL;
                    Ra[t] = Rt[t] * rinvR;
                }
            }
        } else {
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                Lp[n] = Lc[n];
                Rp[n] = Rc[n];
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
        rhi[cm + n * g.sc] = Lp[n];
        rlo[c + n * g.sc] = Rp[n];
    }
    for (int t = 0; t < nt; ++t) {
        rhi[cm + (5 + t) * g.sc] = La[t];
        rlo[c + (5 + t) * g.sc] = Ra[t];
    }
    if (mix) {
        rhi[cm + (5 + nt) * g.sc] = Lg;
        rlo[c + (5 + nt) * g.sc] = Rg;
    }
}

// Dirichlet tables (EulerianState::update_boundary_cells MFP_eulerian.cpp:499-505, face_bc :866-872)
__global__ void k_table_store(Geo g, IBox region, IBox tbox, const double* __restrict__ table, double* __restrict__ dst,
                              long shift) {
    int i, j, k;
    if (!decode(region, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long lx = tbox.hi[0] - tbox.lo[0] + 1, ly = tbox.hi[1] - tbox.lo[1] + 1;
    const long t = ((long)(k - tbox.lo[2]) * ly + (j - tbox.lo[1])) * lx + (i - tbox.lo[0]);
    dst[g.at(i, j, k) + shift] = table[t];
}

// ------------------------------------------------------------ calc_divergence
// EulerianState::calc_divergence MFP_eulerian.cpp:1614-1662
__global__ void k_divergence(Geo g, IBox box, int nc, Ptr3 flux, double* __restrict__ dU, double dt) {
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long str[3] = {1, g.sj, g.sk};
    const long c = g.at(i, j, k);
    for (int n = 0; n < nc; ++n) {
        double du = 0.0;
        for (int d = 0; d < g.dim; ++d) {
            const double dxinv = dt / g.dx[d];
            const double flux_lo = flux.p[d][c + n * g.sc];
            const double flux_hi = flux.p[d][c + str[d] + n * g.sc];
            du += dxinv * (flux_lo - flux_hi);
        }
        dU[c + n * g.sc] = du;
    }
}

// ------------------------------------------------------- MultiFab-level helpers
// MultiFab::Copy / setVal / Add / LinComb on the valid cells (ng = 0), as used by
// Action::get_data (MFP_action.cpp:79,98) and advance_* (MFP_advance.cpp).
__global__ void k_copy(Geo g, IBox box, int nc, double* __restrict__ dst, const double* __restrict__ src) {
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long c = g.at(i, j, k);
    for (int n = 0; n < nc; ++n) dst[c + n * g.sc] = src[c + n * g.sc];
}
__global__ void k_zero(Geo g, IBox box, int nc, double* __restrict__ dst) {
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long c = g.at(i, j, k);
    for (int n = 0; n < nc; ++n) dst[c + n * g.sc] = 0.0;
}
__global__ void k_add(Geo g, IBox box, int nc, double* __restrict__ dst, const double* __restrict__ src) {
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long c = g.at(i, j, k);
    for (int n = 0; n < nc; ++n) dst[c + n * g.sc] += src[c + n * g.sc];
}
__global__ void k_lincomb(Geo g, IBox box, int nc, double* dst, double a, const double* x, double b, const double* y) {
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long c = g.at(i, j, k);
    for (int n = 0; n < nc; ++n) dst[c + n * g.sc] = a * x[c + n * g.sc] + b * y[c + n * g.sc];
}

// -------------------------------------------------------------------- sources
// CollisionsRambo::calc_update (MFP_collisions_rambo.cpp:68-126) over the species of one cell
template <int NS>
CMFP_DEV void collisions_rambo_pairs(const CollArgs& ca, const double* rho, const double* m, const double* q, const double* T,
                                     const double* q2, const double* nd, const double (*vel)[3], double (*delta)[5]) {
    const double lnC = 10.0;
#pragma unroll
    for (int a = 0; a < NS; ++a) {
#pragma unroll
        for (int b = a + 1; b < NS; ++b) {
            const double m_ab = f_div(m[a] * m[b], m[a] + m[b]);
            double du2 = 0.0, du[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                du[i] = vel[b][i] - vel[a][i];
                du2 += du[i] * du[i];
            }
            double nu;
            if ((q[a] != 0) && (q[b] != 0)) {
                const double coeff_1 = ca.c1 * f_div(q2[a] * q2[b] * lnC, m_ab * m[a]);
                const double x = ca.c2 * du2 + f_div(T[a], m[a]) + f_div(T[b], m[b]);
#ifdef CMFP_FASTMATH
                const double p15 = f_rsqrt(x) * f_rcp(x);  // x^-1.5
#else
                const double p15 = pow(x, -1.5);
#endif
                nu = nd[b] * coeff_1 * p15;
            } else {
                const double coeff_2 = f_div(m[b] * ca.ccs[a][b], m[a] + m[b]);
                nu = nd[b] * coeff_2 * ca.c3 * f_sqrt(f_div(T[a], m[a]) + f_div(T[b], m[b]));
            }
            const double S = 0.0;
            double Qx = m_ab * nd[a] * nu * du2 + f_div(3 * rho[a] * nu, m[a] + m[b]) * (T[b] - T[a]);
            double R[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                R[i] = rho[a] * nu * du[i];
                Qx += R[i] * vel[a][i];
            }
            delta[a][0] += S;
            delta[b][0] -= S;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                delta[a][1 + i] += R[i];
                delta[b][1 + i] -= R[i];
            }
            delta[a][4] += Qx;
            delta[b][4] -= Qx;
        }
    }
}

// One cell of Collisions::calc_time_derivative (MFP_collisions.cpp:158-246) with
// CollisionsRambo::calc_update (MFP_collisions_rambo.cpp:68-126), single-species states
// (alpha = 1).  out[s][c] is what the reference adds to dU: (U + dt * delta) - U.
template <int NS>
CMFP_DEV void collisions_rambo_cell(const CollArgs& ca, double dt, const double (*sU)[5], double (*out)[5]) {
    double rho[NS], m[NS], q[NS], T[NS], q2[NS], nd[NS], vel[NS][3], delta[NS][5];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        double Tn = 0.0;
        if (!(sU[n][0] < ca.ez[n])) {  // gas->cons2prim, MFP_collisions.cpp:166-171
            double Q[5];
            GasP G{ca.m[n], ca.q[n], ca.gamma[n]};
            bool ok = hydro_cons2prim(G, sU[n], Q);
            const double rhoinv = f_rcp(sU[n][0]);
            Tn = Q[H_P] * rhoinv * ca.m[n];  // MFP_thermally_perfect_gas.cpp:207
            if (!ok) *ca.err = 1;
        }
#pragma unroll
        for (int c = 0; c < 5; ++c) delta[n][c] = 0.0;
        if (sU[n][0] > ca.ez[n]) {
            rho[n] = 1.0 * sU[n][0];
            m[n] = ca.m[n];
            q[n] = ca.q[n];
            T[n] = Tn;
            q2[n] = q[n] * q[n];
            nd[n] = f_div(rho[n], m[n]);
            const double i_rho = f_rcp(rho[n]);
            vel[n][0] = 1.0 * sU[n][1] * i_rho;
            vel[n][1] = 1.0 * sU[n][2] * i_rho;
            vel[n][2] = 1.0 * sU[n][3] * i_rho;
        } else {
            rho[n] = m[n] = q[n] = T[n] = q2[n] = nd[n] = 0.0;
            vel[n][0] = vel[n][1] = vel[n][2] = 0.0;
        }
    }
    collisions_rambo_pairs<NS>(ca, rho, m, q, T, q2, nd, vel, delta);
#pragma unroll
    for (int n = 0; n < NS; ++n)
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            double acc = sU[n][c] * 1.0;
            acc += dt * delta[n][c];
            out[n][c] = (0.0 + acc) - sU[n][c];
        }
}

template <int NS>
__global__ void k_collisions(Geo g, IBox box, SourceArgs a, double dt) {
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long c = g.at(i, j, k);
    double sU[NS][5], out[NS][5];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int n = 0; n < 5; ++n) sU[s][n] = a.sp_U[s][c + n * g.sc];
    collisions_rambo_cell<NS>(a.coll, dt, sU, out);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int n = 0; n < 5; ++n) a.sp_dU[s][c + n * g.sc] += out[s][n];
}

// The same where a state has sub-components (tracers: passive scalars or a true mixture): the species of the action is
// sub-component CollMix::comp of its state (SpeciesInfo::alpha_idx) with the share alpha of the state's density, mass
// and charge of that sub-component and the state's temperature (MFP_collisions.cpp:160-209); every sub-component's
// share of U is accounted for, summed again and the original subtracted, the tracers follow from the density shares
// (:218-257).  Mass fractions as get_alpha_fractions_from_prim gives them (MFP_hydro_gas.cpp:13-27: no clamp).
template <int NS>
__global__ void k_collisions_mix(Geo g, IBox box, SourceArgs a, CollMix cm, double dt) {
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long c = g.at(i, j, k);
    const CollArgs& ca = a.coll;
    double sU[NS][7], alpha[NS][3];
    double rho[NS], m[NS], q[NS], T[NS], q2[NS], nd[NS], vel[NS][3], delta[NS][5];
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const HydroParams& hp = cm.hp[n];
        const int nt = hp.nt;
#pragma unroll
        for (int l = 0; l < 7; ++l) sU[n][l] = l < 5 + nt ? a.sp_U[n][c + l * g.sc] : 0.0;
        double Tn = 0.0;
        alpha[n][0] = alpha[n][1] = alpha[n][2] = 0.0;  // (below effective_zero the reference keeps stale fractions)
        if (!(sU[n][0] < ca.ez[n])) {  // gas->cons2prim + get_alpha_fractions_from_prim, MFP_collisions.cpp:166-171
            double Q[5];
            GasP G = hp.gas;
            if (hp.mix) {  // get_gamma_from_cons / get_mass_from_cons: alpha_i = tracer_i / rho
                double al[2] = {0.0, 0.0};
                for (int l = 0; l < nt; ++l) al[l] = f_div(sU[n][5 + l], sU[n][0]);
                G = mix_gas(hp, al);
            }
            bool ok = hydro_cons2prim(G, sU[n], Q);
            const double rhoinv = f_rcp(sU[n][0]);
            Tn = Q[H_P] * rhoinv * G.mass;  // MFP_thermally_perfect_gas.cpp:207
            if (!ok) *ca.err = 1;
            double sum_alpha = 0.0;
            for (int l = 0; l < nt; ++l) {
                alpha[n][l] = sU[n][5 + l] * rhoinv;  // Q[tracer], :212-214
                sum_alpha += alpha[n][l];
            }
            alpha[n][nt] = 1.0 - sum_alpha;
        }
#pragma unroll
        for (int l = 0; l < 5; ++l) delta[n][l] = 0.0;
        const double al = alpha[n][cm.comp[n]];
        if (sU[n][0] > ca.ez[n]) {
            rho[n] = al * sU[n][0];
            m[n] = cm.m[n];
            q[n] = cm.q[n];
            T[n] = Tn;
            q2[n] = q[n] * q[n];
            nd[n] = f_div(rho[n], m[n]);
            const double i_rho = f_rcp(rho[n]);
            vel[n][0] = al * sU[n][1] * i_rho;
            vel[n][1] = al * sU[n][2] * i_rho;
            vel[n][2] = al * sU[n][3] * i_rho;
        } else {
            rho[n] = m[n] = q[n] = T[n] = q2[n] = nd[n] = 0.0;
            vel[n][0] = vel[n][1] = vel[n][2] = 0.0;
        }
    }
    collisions_rambo_pairs<NS>(ca, rho, m, q, T, q2, nd, vel, delta);
#pragma unroll
    for (int n = 0; n < NS; ++n) {
        const int nt = cm.hp[n].nt;
        double rho_share[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            double sum = 0.0;
            for (int ai = 0; ai <= nt; ++ai) {
                double acc = sU[n][l] * alpha[n][ai];
                if (ai == cm.comp[n]) acc += dt * delta[n][l];
                if (l == 0) rho_share[ai] = acc;
                sum += acc;
            }
            a.sp_dU[n][c + l * g.sc] += sum - sU[n][l];
        }
        for (int ti = 0; ti < nt; ++ti) a.sp_dU[n][c + (5 + ti) * g.sc] += rho_share[ti] - sU[n][5 + ti];
    }
}

// One cell of Plasma5::implicit_solve (MFP_plasma5.cpp:324-448).  The
// reference factors the dense 3(Ns+1) system with Eigen's ColPivHouseholderQR;
// here the same linear system
//     (I + mc_n [B]x) m_n + tB_n E = m_n^0 ,   E + sum_n tC_n m_n = E^0
// is solved in registers by block elimination: (I + [a]x)^-1 =
// (I + a a^T - [a]x) / (1 + |a|^2) in closed form per species, then one 3x3
// Schur system for E.  Same solution to rounding (tested to 1e-12).
template <int MAXS>
CMFP_DEV void plasma5_implicit_cell(const SourceArgs& a, const SourceCoef& kf, double dt, const double* fU,
                                    const double (*sU)[5], double (*sdU)[4], double* fdU, const double* rc = nullptr) {
    const double Larmor = a.Larmor, Debye = a.Debye, lightspeed = a.lightspeed;
    const double ep = fU[F_EP];
    const double B[3] = {fU[F_BX], fU[F_BY], fU[F_BZ]};
    const double E0[3] = {fU[F_DX] / ep, fU[F_DY] / ep, fU[F_DZ] / ep};
    double charge_density = 0.0;
    double S[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double rhs[3] = {E0[0], E0[1], E0[2]};
    double Minv[MAXS][3][3];
    double tB[MAXS];
#pragma unroll
    for (int n = 0; n < MAXS; ++n) {
        if (n < a.n_species) {
            const double rho = sU[n][0];
            const double r = rc ? rc[n] : a.r[n];
            charge_density += rho * r;
            charge_density += rho * r;  // MFP_plasma5.cpp:370-372 counts it twice (quirk Q3)
            const double trace_B = -dt * r * rho * lightspeed / Larmor;
            const double trace_C = dt * r * Larmor / (lightspeed * Debye * Debye);
            const double mc = dt * r / Larmor;
            tB[n] = trace_B;
            const double av[3] = {mc * B[0], mc * B[1], mc * B[2]};
            const double den = 1.0 + (av[0] * av[0] + av[1] * av[1] + av[2] * av[2]);
            // (I + a a^T - [a]x) / den ;  [a]x = {{0,-az,ay},{az,0,-ax},{-ay,ax,0}}
            const double ax[3][3] = {{0, -av[2], av[1]}, {av[2], 0, -av[0]}, {-av[1], av[0], 0}};
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q) Minv[n][p][q] = ((p == q ? 1.0 : 0.0) + av[p] * av[q] - ax[p][q]) / den;
            const double m0[3] = {sU[n][1], sU[n][2], sU[n][3]};
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                double mm = Minv[n][p][0] * m0[0] + Minv[n][p][1] * m0[1] + Minv[n][p][2] * m0[2];
                rhs[p] -= trace_C * mm;
#pragma unroll
                for (int q = 0; q < 3; ++q) S[p][q] -= trace_C * trace_B * Minv[n][p][q];
            }
        }
    }
    // 3x3 solve by cofactors
    const double c00 = S[1][1] * S[2][2] - S[1][2] * S[2][1];
    const double c01 = S[1][2] * S[2][0] - S[1][0] * S[2][2];
    const double c02 = S[1][0] * S[2][1] - S[1][1] * S[2][0];
    const double det = S[0][0] * c00 + S[0][1] * c01 + S[0][2] * c02;
    const double c10 = S[0][2] * S[2][1] - S[0][1] * S[2][2];
    const double c11 = S[0][0] * S[2][2] - S[0][2] * S[2][0];
    const double c12 = S[0][1] * S[2][0] - S[0][0] * S[2][1];
    const double c20 = S[0][1] * S[1][2] - S[0][2] * S[1][1];
    const double c21 = S[0][2] * S[1][0] - S[0][0] * S[1][2];
    const double c22 = S[0][0] * S[1][1] - S[0][1] * S[1][0];
    double E[3];
    E[0] = (c00 * rhs[0] + c10 * rhs[1] + c20 * rhs[2]) / det;
    E[1] = (c01 * rhs[0] + c11 * rhs[1] + c21 * rhs[2]) / det;
    E[2] = (c02 * rhs[0] + c12 * rhs[1] + c22 * rhs[2]) / det;

    fdU[0] = E[0] * ep - fU[F_DX];
    fdU[1] = E[1] * ep - fU[F_DY];
    fdU[2] = E[2] * ep - fU[F_DZ];
#pragma unroll
    for (int n = 0; n < MAXS; ++n) {
        if (n < a.n_species) {
            const double m0[3] = {sU[n][1], sU[n][2], sU[n][3]};
            const double b[3] = {m0[0] - tB[n] * E[0], m0[1] - tB[n] * E[1], m0[2] - tB[n] * E[2]};
            double m[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) m[p] = Minv[n][p][0] * b[0] + Minv[n][p][1] * b[1] + Minv[n][p][2] * b[2];
            sdU[n][0] = m[0] - m0[0];
            sdU[n][1] = m[1] - m0[1];
            sdU[n][2] = m[2] - m0[2];
            sdU[n][3] = dt * ((rc ? rc[n] : a.r[n]) * lightspeed / Larmor) * (E[0] * m[0] + E[1] * m[1] + E[2] * m[2]);
        }
    }
    fdU[3] = -fU[F_PHI] * kf.f3 + kf.f2 * charge_density;
    fdU[4] = -fU[F_PSI] * kf.f3;
}

// staged form: dU += S(U) on the valid cells (Plasma5::calc_time_derivative
// MFP_plasma5.cpp:73-85, Lorentz::calc_time_derivative MFP_lorentz.cpp:60-198)
template <int NS, bool IMPL>
__global__ void k_source(Geo g, IBox box, SourceArgs a, SourceCoef kf, double dt) {
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long c = g.at(i, j, k);
    double fU[10], sU[NS][5], sdU[NS][4], fdU[5];
#pragma unroll
    for (int n = 0; n < 10; ++n) fU[n] = a.f_U[c + n * g.sc];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int n = 0; n < 5; ++n) sU[s][n] = a.sp_U[s][c + n * g.sc];
    double rc[NS];  // q / m of the cell's species (mixtures: from the mass fractions)
#pragma unroll
    for (int s = 0; s < NS; ++s) rc[s] = species_r(g, a, s, c, sU[s][0]);
    if (IMPL)
        plasma5_implicit_cell<NS>(a, kf, dt, fU, sU, sdU, fdU, rc);
    else
        plasma5_explicit_cell<NS>(a, kf, dt, fU, sU, sdU, fdU, rc);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int n = 0; n < 4; ++n) a.sp_dU[s][c + (n + 1) * g.sc] += sdU[s][n];
    if (!a.lorentz_only) {
        a.f_dU[c + F_DX * g.sc] += fdU[0];
        a.f_dU[c + F_DY * g.sc] += fdU[1];
        a.f_dU[c + F_DZ * g.sc] += fdU[2];
        a.f_dU[c + F_PHI * g.sc] += fdU[3];
        a.f_dU[c + F_PSI * g.sc] += fdU[4];
    }
}

// ------------------------------------------------------------- dt reductions
// Block partials are written with plain stores (no contended atomics: 64 Ki
// single-address atomics cost ~0.75 ms on MI355X); a one-block kernel folds them.
static constexpr int RED_BLOCKS = 2048;

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
    return v;
}
template <bool IS_MAX> __device__ __forceinline__ void block_reduce_store(double v, double* part) {
    __shared__ double sh[TPB / 64];
    v = IS_MAX ? wave_max(v) : wave_min(v);
    __syncthreads();  // sh may be reused by a previous call
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) v = IS_MAX ? fmax(v, sh[w]) : fmin(v, sh[w]);
        part[blockIdx.x] = v;
    }
}
// folds nb block partials into *out (the bit pattern of the double)
__global__ void k_reduce_final(const double* __restrict__ part, int nb, int is_max, unsigned long long* out) {
    double v = is_max ? DBL_MIN : DBL_MAX;
    for (int t = threadIdx.x; t < nb; t += TPB) v = is_max ? fmax(v, part[t]) : fmin(v, part[t]);
    __shared__ double sh[TPB / 64];
    v = is_max ? wave_max(v) : wave_min(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) v = is_max ? fmax(v, sh[w]) : fmin(v, sh[w]);
        *out = (unsigned long long)__double_as_longlong(v);
    }
}

// fold the 256 atomic slots of job `job` of a RedJobs and reset them; every thread of a 256-thread workgroup calls it
__device__ __forceinline__ void reduce_job(const RedJobs& J, int job, double* sh) {
    unsigned long long* const slots = J.slots[job];
    const int is_max = J.is_max[job];
    const unsigned long long identity = is_max ? 0ull : ~0ull;
    if (J.mode[job] == 0) {
        double v = __longlong_as_double((long long)slots[threadIdx.x]);
        v = is_max ? wave_max(v) : wave_min(v);  // fmax / fmin: the all-ones NaN of an untouched slot is ignored
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) v = is_max ? fmax(v, sh[w]) : fmin(v, sh[w]);
            // as k_reduce_final over the same slots: DBL_MIN / DBL_MAX where no lane contributed
            v = is_max ? fmax(v, DBL_MIN) : fmin(v, DBL_MAX);
            *J.out[job] = (unsigned long long)__double_as_longlong(v);
        }
    }
    slots[threadIdx.x] = identity;
}
// one workgroup per job of a RedJobs
__global__ __launch_bounds__(256) void k_reduce_many(RedJobs J) {
    __shared__ double sh[4];
    reduce_job(J, blockIdx.x, sh);
}
// post-mortem of a failed step: the first valid cell (in memory order) of a hydro state without valid primitives
__global__ void k_first_invalid(Geo g, IBox box, HydroParams hp, const double* __restrict__ U, unsigned long long* out) {
    int i, j, k;
    if (!decode(box, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long t = g.at(i, j, k);
    double u[5], q[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) u[n] = U[t + n * g.sc];
    GasP G = hp.gas;
    if (hp.mix) {
        double al[2] = {0.0, 0.0};
        for (int n = 0; n < hp.nt; ++n) al[n] = f_div(U[t + (5 + n) * g.sc], u[0]);
        G = mix_gas(hp, al);
    }
    const double T = (u[4] - 0.5 * (u[1] * u[1] + u[2] * u[2] + u[3] * u[3]) / u[0]) * (G.gamma - 1) / u[0] * G.mass;
    if (!hydro_cons2prim(G, u, q) || !(T > 0.0))  // NaNs count as invalid here
        atomicMin(out, (unsigned long long)(((long)k * g.n[1] + j) * g.n[0] + i));
}
static void l_first_invalid(const Geo& g, const StatePhys& sp, const double* U, unsigned long long* out, hipStream_t st) {
    IBox b = valid_box(g);
    const long np = b.npts();
    hipLaunchKernelGGL(k_first_invalid, dim3((unsigned)((np + TPB - 1) / TPB)), dim3(TPB), 0, st, g, b, sp.hp, U, out);
}
static void l_reduce_many(const RedJobs& J, hipStream_t st) {
    if (J.n > 0) hipLaunchKernelGGL(k_reduce_many, dim3(J.n), dim3(256), 0, st, J);
}

// fused form of one Strang source stage with the RK2 inner integrator
// (MFP_advance.cpp:146-197 / 235-281), in place on the new-time data:
//   U1 = U0 + S(U0) ; U2 = U1 + S(U1) ; new = 0.5 U0 + 0.5 U2
// Components no source touches are left as they are (the reference rewrites
// them with 0.5 x + 0.5 (x + 0 + 0) = x).  With REDUCE the kernel also leaves
// the block partials of the CFL quantities of the values it just wrote
// (max a+|u_d| per species, min plasma/cyclotron dt), so the dt estimate of the
// next step costs no extra pass over the state.
// plane-relative access: uniform base pointer (scalar registers) + 32-bit per-lane byte offset
__device__ __forceinline__ double ld_plane(const double* base, unsigned off) {
    return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ void st_plane(double* base, unsigned off, double v) {
    *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off) = v;
}

template <int NS, bool IMPL, bool REDUCE, bool COLL>
__global__ __launch_bounds__(TPB) void k_source_rk2(Geo g, IBox box, SourceArgs a, SourceCoef kf, double dt) {
    if (g.dtp) {  // device-resident time step (cmfp_advance_steps): dt is the factor 1/2 of the half step
        dt = launch_dt(g, dt);
        kf = source_coef(a, dt);
    }
    const long npts = box.npts();
    double red_speed[NS];
    double red_dt = DBL_MAX;
#pragma unroll
    for (int s = 0; s < NS; ++s) red_speed[s] = DBL_MIN;
    // grid-stride walk with the (i, j, k) index advanced incrementally: the 64-bit divisions of a
    // per-cell decode cost more VALU work than the source terms themselves
    const int l0 = box.hi[0] - box.lo[0] + 1, l1 = box.hi[1] - box.lo[1] + 1;
    const long stride = (long)gridDim.x * TPB;
    const int st0 = (int)(stride % l0), st1 = (int)((stride / l0) % l1), st2 = (int)(stride / l0 / l1);
    long t = (long)blockIdx.x * TPB + threadIdx.x;
    int ii = (int)(t % l0), jj = (int)((t / l0) % l1), kk = (int)(t / l0 / l1);
    for (; t < npts; t += stride) {
        // byte offset inside one component plane, 32-bit (cmfp_create bounds the plane at 4 GiB): with the
        // uniform plane base in scalar registers every access is base + offset, no 64-bit address per comp
        const unsigned c = (unsigned)g.at(box.lo[0] + ii, box.lo[1] + jj, box.lo[2] + kk) * 8u;
        ii += st0;
        if (ii >= l0) {
            ii -= l0;
            ++jj;
        }
        jj += st1;
        if (jj >= l1) {
            jj -= l1;
            ++kk;
        }
        kk += st2;
        double f0[10], s0[NS][5], f1[10], s1[NS][5], sdU[NS][4], fdU[5];
#pragma unroll
        for (int n = 0; n < 10; ++n) f0[n] = ld_plane(a.f_U + n * g.sc, c);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int n = 0; n < 5; ++n) s0[s][n] = ld_plane(a.sp_U[s] + n * g.sc, c);
        // stage 1
        if (IMPL)
            plasma5_implicit_cell<NS>(a, kf, dt, f0, s0, sdU, fdU);
        else
            plasma5_explicit_cell<NS>(a, kf, dt, f0, s0, sdU, fdU);
        if constexpr (COLL) {  // a collisions_rambo action over the same species adds to the same dU
            double cd[NS][5];
            collisions_rambo_cell<NS>(a.coll, dt, s0, cd);
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int n = 0; n < 4; ++n) sdU[s][n] += cd[s][n + 1];
        }
#pragma unroll
        for (int n = 0; n < 10; ++n) f1[n] = f0[n];
        if (!a.lorentz_only) {
            f1[F_DX] = f0[F_DX] + fdU[0];
            f1[F_DY] = f0[F_DY] + fdU[1];
            f1[F_DZ] = f0[F_DZ] + fdU[2];
            f1[F_PHI] = f0[F_PHI] + fdU[3];
            f1[F_PSI] = f0[F_PSI] + fdU[4];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            s1[s][0] = s0[s][0];
#pragma unroll
            for (int n = 0; n < 4; ++n) s1[s][n + 1] = s0[s][n + 1] + sdU[s][n];
        }
        // stage 2
        if (IMPL)
            plasma5_implicit_cell<NS>(a, kf, dt, f1, s1, sdU, fdU);
        else
            plasma5_explicit_cell<NS>(a, kf, dt, f1, s1, sdU, fdU);
        if constexpr (COLL) {
            double cd[NS][5];
            collisions_rambo_cell<NS>(a.coll, dt, s1, cd);
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int n = 0; n < 4; ++n) sdU[s][n] += cd[s][n + 1];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            double Un[5];
            Un[0] = s0[s][0];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const double U2 = s1[s][n + 1] + sdU[s][n];
                Un[n + 1] = 0.5 * s0[s][n + 1] + 0.5 * U2;
                st_plane(a.sp_U[s] + (n + 1) * g.sc, c, Un[n + 1]);
            }
            if (REDUCE) {
                GasP G{1.0, 0.0, a.gamma[s]};
                red_speed[s] = cell_max_speed(G, Un, g.dim, red_speed[s]);
                if (a.limits_dt) red_dt = cell_plasma_dt(a.Larmor, a.Debye, f0[F_BX], f0[F_BY], f0[F_BZ], Un[0], a.r[s], red_dt);
            }
        }
        if (!a.lorentz_only) {
            st_plane(a.f_U + F_DX * g.sc, c, 0.5 * f0[F_DX] + 0.5 * (f1[F_DX] + fdU[0]));
            st_plane(a.f_U + F_DY * g.sc, c, 0.5 * f0[F_DY] + 0.5 * (f1[F_DY] + fdU[1]));
            st_plane(a.f_U + F_DZ * g.sc, c, 0.5 * f0[F_DZ] + 0.5 * (f1[F_DZ] + fdU[2]));
            st_plane(a.f_U + F_PHI * g.sc, c, 0.5 * f0[F_PHI] + 0.5 * (f1[F_PHI] + fdU[3]));
            st_plane(a.f_U + F_PSI * g.sc, c, 0.5 * f0[F_PSI] + 0.5 * (f1[F_PSI] + fdU[4]));
#ifndef CMFP_FASTMATH
            // The components no source term touches also go through the reference's 0.5 x + 0.5 (x + 0 + 0)
            // (plasma5 marks the whole field state as changed, MFP_plasma5.cpp:96).  That is x for every
            // normal number; for a subnormal x with an odd last bit 0.5 x rounds, and the sum is x +- 1 ulp
            // (seen at the foot of a wave front running into a field that is exactly zero).  STRICT follows the
            // reference there too; the store happens only for such values.
#pragma unroll
            for (int n : {F_BX, F_BY, F_BZ, F_MU, F_EP}) {
                const double y = 0.5 * f0[n] + 0.5 * f0[n];
                if (y != f0[n]) st_plane(a.f_U + n * g.sc, c, y);
            }
#endif
        }
    }
    if (REDUCE) {
#pragma unroll
        for (int s = 0; s < NS; ++s) block_reduce_store<true>(red_speed[s], a.part_speed[s]);
        if (a.limits_dt) block_reduce_store<false>(red_dt, a.part_dt);
    }
}

// ------------------------------------------------------------------ viscous hydro fluxes
// Cell quantities the viscous terms need, from the conserved data with the arithmetic of
// ThermallyPerfectGas::cons2prim (so u, v, w, T equal the reference's primitive array), and
// HydroViscous::get_coeffs (MFP_hydro_viscous.cpp:145-160, 207-220, 266-279).
struct ViscCell {
    double u, v, w, T, mu, kappa, rho, gamma;
};
__device__ __forceinline__ double visc_mu(const ViscP& V, double T) {
    if (V.kind == 1) {  // CMFP_VISC_SUTHERLAND
        const double T_ = T / V.T0;
        return V.mu0 * T_ * sqrt(T_) * (V.T0 + V.S) / (T + V.S);
    }
    if (V.kind == 2) return  /* CMFP_VISC_POWERLAW */ V.mu0 * pow(T / V.T0, V.n);
    return V.mu0;
}
__device__ __forceinline__ ViscCell visc_cell(const HydroParams& hp, const ViscP& V, const double* __restrict__ U, long c, long sc) {
    ViscCell q;
    const double rho = U[c], mx = U[c + sc], my = U[c + 2 * sc], mz = U[c + 3 * sc], ed = U[c + 4 * sc];
    const double rhoinv = 1 / rho;
    GasP G = hp.gas;
    double cp_mix = 0.0;
    if (hp.mix) {  // gamma, mass from the conserved alpha (tracer / rho), cp from the primitive one (tracer * rhoinv)
        double ac[2] = {0.0, 0.0}, ap[2] = {0.0, 0.0};
        for (int n = 0; n < hp.nt; ++n) {
            ac[n] = f_div(U[c + (5 + n) * sc], rho);
            ap[n] = U[c + (5 + n) * sc] * rhoinv;
        }
        G = mix_gas(hp, ac);
        cp_mix = mix_cp(hp, ap);
    }
    q.gamma = G.gamma;
    q.rho = rho;
    q.u = mx * rhoinv;
    q.v = my * rhoinv;
    q.w = mz * rhoinv;
    const double ke = 0.5 * rho * (q.u * q.u + q.v * q.v + q.w * q.w);
    const double p = (ed - ke) * (G.gamma - 1);
    q.T = p * rhoinv * G.mass;
    const double cp = hp.mix ? cp_mix : G.gamma / (G.mass * (G.gamma - 1));  // MFP_thermally_perfect_gas.cpp:127
    q.mu = visc_mu(V, q.T);
    q.kappa = q.mu * cp / V.Pr;
    return q;
}

// HydroState::calc_viscous_fluxes MFP_hydro.cpp:1085-1271, one thread per face of direction D.
// `two_thirds = 2 / 3` is integer division in the reference (:1123): the divergence term is absent.
template <int D>
__global__ void k_viscous(Geo g, IBox faces, HydroParams G, ViscP V, const double* __restrict__ U, double* __restrict__ flux) {
    int i, j, k;
    if (!decode(faces, (long)blockIdx.x * TPB + threadIdx.x, i, j, k)) return;
    const long sc = g.sc;
    const long c = g.at(i, j, k);
    const long sx = 1, sy = g.sj;
    const double dxinv0 = 1 / g.dx[0], dxinv1 = 1 / g.dx[1];
    const double two_thirds = 2 / 3;
    const long cm = c - (D == 0 ? sx : sy);
    const ViscCell a = visc_cell(G, V, U, c, sc), b = visc_cell(G, V, U, cm, sc);
    double dudz = 0, dwdz = 0, dvdz = 0;
    if (D == 0) {
        const double dTdx = (a.T - b.T) * dxinv0;
        const double dudx = (a.u - b.u) * dxinv0;
        const double dvdx = (a.v - b.v) * dxinv0;
        const double dwdx = (a.w - b.w) * dxinv0;
        double dudy = 0, dvdy = 0;
        if (g.dim >= 2) {
            const ViscCell pa = visc_cell(G, V, U, c + sy, sc), pb = visc_cell(G, V, U, cm + sy, sc);
            const ViscCell ma = visc_cell(G, V, U, c - sy, sc), mb = visc_cell(G, V, U, cm - sy, sc);
            dudy = (pa.u + pb.u - ma.u - mb.u) * (0.25 * dxinv1);
            dvdy = (pa.v + pb.v - ma.v - mb.v) * (0.25 * dxinv1);
        }
        const double divu = dudx + dvdy + dwdz;
        const double muf = 0.5 * (a.mu + b.mu);
        const double tauxx = muf * (2 * dudx - two_thirds * divu);
        const double tauxy = muf * (dudy + dvdx);
        const double tauxz = muf * (dudz + dwdx);
        flux[c + 1 * sc] -= tauxx;
        flux[c + 2 * sc] -= tauxy;
        flux[c + 3 * sc] -= tauxz;
        flux[c + 4 * sc] -= 0.5 * ((a.u + b.u) * tauxx + (a.v + b.v) * tauxy + (a.w + b.w) * tauxz + (a.kappa + b.kappa) * dTdx);
    } else {
        const double dTdy = (a.T - b.T) * dxinv1;
        const double dudy = (a.u - b.u) * dxinv1;
        const double dvdy = (a.v - b.v) * dxinv1;
        const double dwdy = (a.w - b.w) * dxinv1;
        const ViscCell pa = visc_cell(G, V, U, c + sx, sc), pb = visc_cell(G, V, U, cm + sx, sc);
        const ViscCell ma = visc_cell(G, V, U, c - sx, sc), mb = visc_cell(G, V, U, cm - sx, sc);
        const double dudx = (pa.u + pb.u - ma.u - mb.u) * (0.25 * dxinv0);
        const double dvdx = (pa.v + pb.v - ma.v - mb.v) * (0.25 * dxinv0);
        const double divu = dudx + dvdy + dwdz;
        const double muf = 0.5 * (a.mu + b.mu);
        const double tauyy = muf * (2 * dvdy - two_thirds * divu);
        const double tauxy = muf * (dudy + dvdx);
        const double tauyz = muf * (dwdy + dvdz);
        flux[c + 1 * sc] -= tauxy;
        flux[c + 2 * sc] -= tauyy;
        flux[c + 3 * sc] -= tauyz;
        flux[c + 4 * sc] -= 0.5 * ((a.u + b.u) * tauxy + (a.v + b.v) * tauyy + (a.w + b.w) * tauyz + (a.kappa + b.kappa) * dTdy);
    }
}

// HydroViscous::get_min_dt MFP_hydro_viscous.cpp:34-97, cell loop: max of calc_stability_condition
// (:162-176, 222-236, 281-292), starting from 0
__global__ void k_visc_stability(Geo g, IBox box, HydroParams G, ViscP V, const double* __restrict__ U, double* part) {
    const long npts = box.npts();
    double m = 0.0;
    for (long t = (long)blockIdx.x * TPB + threadIdx.x; t < npts; t += (long)gridDim.x * TPB) {
        int i, j, k;
        decode(box, t, i, j, k);
        const ViscCell q = visc_cell(G, V, U, g.at(i, j, k), g.sc);
        const double s = (4 * q.mu * q.gamma / (V.Pr * q.rho));
        m = (m < s) ? s : m;
    }
    block_reduce_store<true>(m, part);
}

// EulerianState::get_allowed_time_step MFP_eulerian.cpp:255-300: max over valid
// cells and over d < DIM of get_speed_from_cons.
__global__ void k_max_speed(Geo g, IBox box, HydroParams hp, const double* __restrict__ U, double* part) {
    const long npts = box.npts();
    double m = DBL_MIN;  // std::numeric_limits<Real>::min()  (:265)
    for (long t = (long)blockIdx.x * TPB + threadIdx.x; t < npts; t += (long)gridDim.x * TPB) {
        int i, j, k;
        decode(box, t, i, j, k);
        const long c = g.at(i, j, k);
        double u[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) u[n] = U[c + n * g.sc];
        GasP G = hp.gas;
        if (hp.mix) {  // get_gamma_from_cons (MFP_thermally_perfect_gas.cpp:309)
            double al[2] = {0.0, 0.0};
            for (int n = 0; n < hp.nt; ++n) al[n] = f_div(U[c + (5 + n) * g.sc], u[0]);
            G = mix_gas(hp, al);
        }
        m = cell_max_speed(G, u, g.dim, m);
    }
    block_reduce_store<true>(m, part);
}

// Plasma5::get_allowed_time_step MFP_plasma5.cpp:522-557 (cell loop part)
__global__ void k_plasma_dt(Geo g, IBox box, SourceArgs a, double* part) {
    const long npts = box.npts();
    double dt = DBL_MAX;
    for (long t = (long)blockIdx.x * TPB + threadIdx.x; t < npts; t += (long)gridDim.x * TPB) {
        int i, j, k;
        decode(box, t, i, j, k);
        const long c = g.at(i, j, k);
        const double Bx = a.f_U[c + F_BX * g.sc], By = a.f_U[c + F_BY * g.sc], Bz = a.f_U[c + F_BZ * g.sc];
        for (int n = 0; n < a.n_species; ++n)
            dt = cell_plasma_dt(a.Larmor, a.Debye, Bx, By, Bz, a.sp_U[n][c], species_r(g, a, n, c, a.sp_U[n][c]), dt);
    }
    block_reduce_store<false>(dt, part);
}

// ------------------------------------------------------------------ ghosts
// One (dim, side) sweep of the ghost fill: fab_filcc semantics (dimension by
// dimension over the full extent of the other dimensions) + periodic wrap.
__device__ __forceinline__ void fill_dim_cell(const Geo& g, const IBox& slab, long t, int nc, int d, int side, const FillRules& rules,
                                              double* __restrict__ A);
__global__ void k_fill_dim(Geo g, IBox slab, int nc, int ngs, int d, int side, FillRules rules, double* __restrict__ A) {
    fill_dim_cell(g, slab, (long)blockIdx.x * TPB + threadIdx.x, nc, d, side, rules, A);
}
// blockIdx.y = 2 * job + side
__global__ void k_fill_multi(Geo g, int d, FillJobs J) {
    const int job = (int)blockIdx.y >> 1, side = (int)blockIdx.y & 1;
    const FillJob& f = J.j[job];
    if (!f.on[side]) return;
    IBox slab;
    for (int t = 0; t < 3; ++t) {
        slab.lo[t] = -g.ng[t];
        slab.hi[t] = g.n[t] - 1 + g.ng[t];
    }
    slab.lo[d] = 1;
    slab.hi[d] = f.ngs;
    fill_dim_cell(g, slab, (long)blockIdx.x * TPB + threadIdx.x, f.nc, d, side, f.rules[side], f.A);
}
__device__ __forceinline__ void fill_dim_cell(const Geo& g, const IBox& slab, long t, int nc, int d, int side, const FillRules& rules,
                                              double* __restrict__ A) {
    int c3[3];
    if (!decode(slab, t, c3[0], c3[1], c3[2])) return;
    // slab coordinate along d is the layer 1..ngs
    const int layer = c3[d];
    const int n = g.n[d];
    const int idst = side == 0 ? -layer : n - 1 + layer;
    int s3[3] = {c3[0], c3[1], c3[2]};
    c3[d] = idst;
    const long cd = g.at(c3[0], c3[1], c3[2]);
    // all loads first, then all stores: source and destination are the same array, so a load after a store
    // would wait for it (ten dependent round trips to memory per thread instead of one)
    double v[10];
#pragma unroll
    for (int m = 0; m < 10; ++m) {
        const int r = m < nc ? rules.rule[m] : FILL_SKIP;
        int isrc;
        if (r == FILL_WRAP)
            isrc = side == 0 ? n - layer : layer - 1;
        else if (r == FILL_FOEXTRAP)
            isrc = side == 0 ? 0 : n - 1;
        else
            isrc = side == 0 ? layer - 1 : n - layer;
        s3[d] = isrc;
        v[m] = (r == FILL_SKIP) ? 0.0 : A[g.at(s3[0], s3[1], s3[2]) + m * g.sc];
    }
#pragma unroll
    for (int m = 0; m < 10; ++m) {
        const int r = m < nc ? rules.rule[m] : FILL_SKIP;
        if (r != FILL_SKIP) A[cd + m * g.sc] = (r == FILL_ODD) ? -v[m] : v[m];
    }
}

// halo pack / unpack of the ngs layers next to a face; the slab spans the full
// ghosted extent of the other dimensions.
__global__ void k_pack(Geo g, IBox slab, int nc, int d, int first, double* __restrict__ A, double* __restrict__ buf, int unpack) {
    const long t = (long)blockIdx.x * TPB + threadIdx.x;
    int c3[3];
    if (!decode(slab, t, c3[0], c3[1], c3[2])) return;
    const long np = slab.npts();
    c3[d] = first + (c3[d] - slab.lo[d]);
    const long c = g.at(c3[0], c3[1], c3[2]);
    for (int m = 0; m < nc; ++m) {
        if (unpack)
            A[c + m * g.sc] = buf[t + m * np];
        else
            buf[t + m * np] = A[c + m * g.sc];
    }
}

// ------------------------------------------------------------- leaf kernels
template <int RECON> __global__ void k_leaf_recon(long n, const double* st, double* slope, double* lo, double* hi) {
    const long t = (long)blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    constexpr int SL = ReconTraits<RECON>::SL;
    double s[SL];
#pragma unroll
    for (int q = 0; q < SL; ++q) s[q] = st[q * n + t];
    slope[t] = recon_slope<RECON>(s);
    double l = 0, h = 0;
    recon_faces<RECON>(s, l, h);
    lo[t] = l;
    hi[t] = h;
}
template <int FLUX> __global__ void k_leaf_riemann_hydro(long n, const double* L, const double* R, double* F, double thr) {
    const long t = (long)blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    double l[5], r[5], f[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        l[q] = L[q * n + t];
        r[q] = R[q * n + t];
    }
    const double gL = L[6 * n + t], gR = R[6 * n + t];
    if (FLUX == 0)
        hllc_solve(l, r, gL, gR, f);
    else if (FLUX == 1)
        hlle_solve(l, r, gL, gR, f);
    else
        hybrid_solve(l, r, gL, gR, shock_detect(l, r, thr), f);
#pragma unroll
    for (int q = 0; q < 5; ++q) F[q * n + t] = f[q];
}
__global__ void k_leaf_riemann_field(long n, const double* L, const double* R, double* F, double c0, double ch) {
    const long t = (long)blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    double l[10], r[10], f[10];
#pragma unroll
    for (int q = 0; q < 10; ++q) {
        l[q] = L[q * n + t];
        r[q] = R[q * n + t];
    }
    field_rh_solve(c0, ch, l, r, f);
#pragma unroll
    for (int q = 0; q < 10; ++q) F[q * n + t] = f[q];
}
__global__ void k_leaf_cons2prim(long n, GasP G, const double* U, double* Q, double* sp) {
    const long t = (long)blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    double u[5], q[5], s[3];
#pragma unroll
    for (int c = 0; c < 5; ++c) u[c] = U[c * n + t];
    hydro_cons2prim(G, u, q);
    hydro_speed_from_cons(G, u, s);
#pragma unroll
    for (int c = 0; c < 5; ++c) Q[c * n + t] = q[c];
    // T, gamma, cp as ThermallyPerfectGas::cons2prim reports them (:202,210-211)
    Q[5 * n + t] = q[H_P] * (1 / u[0]) * G.mass;
    Q[6 * n + t] = G.gamma;
    Q[7 * n + t] = G.gamma / (G.mass * (G.gamma - 1));
#pragma unroll
    for (int c = 0; c < 3; ++c) sp[c * n + t] = s[c];
}
__global__ void k_leaf_prim2cons(long n, GasP G, const double* Q, double* U) {
    const long t = (long)blockIdx.x * TPB + threadIdx.x;
    if (t >= n) return;
    double u[5], q[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) q[c] = Q[c * n + t];
    hydro_prim2cons(G, q, u);
#pragma unroll
    for (int c = 0; c < 5; ++c) U[c * n + t] = u[c];
}

// ===========================================================================
// host launchers
#define LAUNCH(kern, npts, stream, ...) \
    hipLaunchKernelGGL(kern, dim3(nblocks(npts)), dim3(TPB), 0, stream, __VA_ARGS__)

static void l_prim(const Geo& g, const StatePhys& sp, const double* U, double* Q, int* err, hipStream_t st) {
    IBox b = grown_box(g, sp.ng);
    if (sp.kind == 0)
        LAUNCH(k_prim<true>, b.npts(), st, g, b, sp.hp, U, Q, err);
    else
        LAUNCH(k_prim<false>, b.npts(), st, g, b, sp.hp, U, Q, err);
}

template <bool HYDRO>
static void l_recon_t(const Geo& g, const StatePhys& sp, const double* Q, Ptr3 rlo, Ptr3 rhi, int do_ctu, double dt,
                      hipStream_t st) {
    IBox b = grown_box(g, 1);
    const long n = b.npts();
    switch (sp.recon) {
    case 0: LAUNCH((k_recon<HYDRO, 0>), n, st, g, b, sp.hp, sp.fp, Q, rlo, rhi, do_ctu, dt); break;
    case 1: LAUNCH((k_recon<HYDRO, 1>), n, st, g, b, sp.hp, sp.fp, Q, rlo, rhi, do_ctu, dt); break;
    case 2: LAUNCH((k_recon<HYDRO, 2>), n, st, g, b, sp.hp, sp.fp, Q, rlo, rhi, do_ctu, dt); break;
    case 3: LAUNCH((k_recon<HYDRO, 3>), n, st, g, b, sp.hp, sp.fp, Q, rlo, rhi, do_ctu, dt); break;
    case 4: LAUNCH((k_recon<HYDRO, 4>), n, st, g, b, sp.hp, sp.fp, Q, rlo, rhi, do_ctu, dt); break;
    case 5: LAUNCH((k_recon<HYDRO, 5>), n, st, g, b, sp.hp, sp.fp, Q, rlo, rhi, do_ctu, dt); break;
    case 6: LAUNCH((k_recon<HYDRO, 6>), n, st, g, b, sp.hp, sp.fp, Q, rlo, rhi, do_ctu, dt); break;
    }
}
static void l_recon(const Geo& g, const StatePhys& sp, const double* Q, Ptr3 rlo, Ptr3 rhi, int do_ctu, double dt,
                    hipStream_t st) {
    if (sp.kind == 0)
        l_recon_t<true>(g, sp, Q, rlo, rhi, do_ctu, dt, st);
    else
        l_recon_t<false>(g, sp, Q, rlo, rhi, do_ctu, dt, st);
}

static FillRules face_rules(const StatePhys& sp, int d, int side) {
    // MFP_eulerian.cpp:796-835: foextrap / hoextrap / reflect_even copy, reflect_odd negates
    FillRules r;
    for (int n = 0; n < 10; ++n) {
        int bc = (n < sp.nc) ? sp.bc[n][d][side] : -666;
        // hydro with tracers: fill_bc[5], fill_bc[6] (the tracer entries) index the primitives T and
        // gamma in the reference (MFP_hydro.cpp:448-463 sets the first n_cons of n_prim entries), which
        // never reach the conserved result with a single-species gas; alpha (primitive 8+) keeps its
        // reconstructed face value.  The carried slots 5+ are alpha: skip them.
        if (sp.kind == 0 && n >= 5) bc = -666;
        // mixtures carry Gamma (primitive 6) in slot 5 + nt: with two tracers it takes fill_bc[6], else bogus
        if (sp.kind == 0 && sp.hp.mix && n == 5 + sp.hp.nt && sp.hp.nt == 2) bc = sp.bc[6][d][side];
        r.rule[n] = (bc == 2 || bc == 4 || bc == 1) ? FILL_EVEN : (bc == -1 ? FILL_ODD : FILL_SKIP);
    }
    return r;
}
static void l_face_bc(const Geo& g, const StatePhys& sp, int d, int side, double* rlo, double* rhi, hipStream_t st) {
    IBox pl = valid_box(g);
    pl.lo[d] = pl.hi[d] = (side == 0) ? 0 : g.n[d];
    LAUNCH(k_face_bc, pl.npts(), st, g, pl, d, side, sp.nq, face_rules(sp, d, side), rlo, rhi);
}

static void l_table_store(const Geo& g, const IBox& region, const IBox& tbox, const double* table, double* dst, long shift,
                          hipStream_t st) {
    if (region.npts() <= 0) return;
    LAUNCH(k_table_store, region.npts(), st, g, region, tbox, table, dst, shift);
}

static void l_fluxes(const Geo& g, const StatePhys& sp, int d, const double* rlo, const double* rhi, double* flux,
                     int grown, hipStream_t st) {
    IBox fb = valid_box(g);
    fb.hi[d] += 1;  // surroundingNodes(box, d)
    if (grown)      // MFP_CTU_hydro.cpp:256-262
        for (int t = 0; t < g.dim; ++t)
            if (t != d) {
                fb.lo[t] -= 1;
                fb.hi[t] += 1;
            }
    const long n = fb.npts();
    if (sp.kind == 0) {
        switch (sp.flux) {
        case 0: LAUNCH((k_fluxes<true, 0>), n, st, g, fb, d, sp.hp, sp.fp, rlo, rhi, flux); break;
        case 1: LAUNCH((k_fluxes<true, 1>), n, st, g, fb, d, sp.hp, sp.fp, rlo, rhi, flux); break;
        default: LAUNCH((k_fluxes<true, 2>), n, st, g, fb, d, sp.hp, sp.fp, rlo, rhi, flux); break;
        }
    } else {
        LAUNCH((k_fluxes<false, 0>), n, st, g, fb, d, sp.hp, sp.fp, rlo, rhi, flux);
    }
}

static void l_correct(const Geo& g, const StatePhys& sp, int d1, double* rlo, double* rhi, Ptr3 flux, double dt,
                      int* err, hipStream_t st) {
    IBox fb = valid_box(g);
    fb.hi[d1] += 1;  // MFP_eulerian.cpp:1100-1103
    if (sp.kind == 0)
        LAUNCH(k_correct<true>, fb.npts(), st, g, fb, d1, sp.hp, rlo, rhi, flux, dt, err);
    else
        LAUNCH(k_correct<false>, fb.npts(), st, g, fb, d1, sp.hp, rlo, rhi, flux, dt, err);
}

static void l_divergence(const Geo& g, int nc, Ptr3 flux, double* dU, double dt, hipStream_t st) {
    IBox b = valid_box(g);
    LAUNCH(k_divergence, b.npts(), st, g, b, nc, flux, dU, dt);
}

static void l_copy(const Geo& g, int nc, double* dst, const double* src, int with_ghosts, hipStream_t st) {
    IBox b = with_ghosts ? grown_box(g, g.ng[0]) : valid_box(g);
    LAUNCH(k_copy, b.npts(), st, g, b, nc, dst, src);
}
static void l_zero(const Geo& g, int nc, double* dst, hipStream_t st) {
    IBox b = valid_box(g);
    LAUNCH(k_zero, b.npts(), st, g, b, nc, dst);
}
static void l_add(const Geo& g, int nc, double* dst, const double* src, hipStream_t st) {
    IBox b = valid_box(g);
    LAUNCH(k_add, b.npts(), st, g, b, nc, dst, src);
}
static void l_lincomb(const Geo& g, int nc, double* dst, double a, const double* x, double bb, const double* y,
                      hipStream_t st) {
    IBox b = valid_box(g);
    LAUNCH(k_lincomb, b.npts(), st, g, b, nc, dst, a, x, bb, y);
}

template <int NS> static void l_source_n(const Geo& g, const SourceArgs& a, double dt, hipStream_t st) {
    IBox b = valid_box(g);
    if (a.implicit)
        LAUNCH((k_source<NS, true>), b.npts(), st, g, b, a, source_coef(a, dt), dt);
    else
        LAUNCH((k_source<NS, false>), b.npts(), st, g, b, a, source_coef(a, dt), dt);
}
static void l_source(const Geo& g, const SourceArgs& a, double dt, hipStream_t st) {
    switch (a.n_species) {
    case 1: l_source_n<1>(g, a, dt, st); break;
    case 2: l_source_n<2>(g, a, dt, st); break;
    case 3: l_source_n<3>(g, a, dt, st); break;
    case 4: l_source_n<4>(g, a, dt, st); break;
    }
}
static void l_collisions_mix(const Geo& g, const SourceArgs& a, const CollMix& cm, double dt, hipStream_t st) {
    const IBox b = valid_box(g);
    switch (a.n_species) {
    case 1: LAUNCH((k_collisions_mix<1>), b.npts(), st, g, b, a, cm, dt); break;
    case 2: LAUNCH((k_collisions_mix<2>), b.npts(), st, g, b, a, cm, dt); break;
    case 3: LAUNCH((k_collisions_mix<3>), b.npts(), st, g, b, a, cm, dt); break;
    case 4: LAUNCH((k_collisions_mix<4>), b.npts(), st, g, b, a, cm, dt); break;
    default: break;
    }
}
static void l_collisions(const Geo& g, const SourceArgs& a, double dt, hipStream_t st) {
    IBox b = valid_box(g);
    switch (a.n_species) {
    case 1: LAUNCH((k_collisions<1>), b.npts(), st, g, b, a, dt); break;
    case 2: LAUNCH((k_collisions<2>), b.npts(), st, g, b, a, dt); break;
    case 3: LAUNCH((k_collisions<3>), b.npts(), st, g, b, a, dt); break;
    case 4: LAUNCH((k_collisions<4>), b.npts(), st, g, b, a, dt); break;
    }
}
static unsigned red_blocks(long npts) { return (unsigned)std::min<long>(nblocks(npts), RED_BLOCKS); }

template <int NS> static void l_source_rk2_n(const Geo& g, const SourceArgs& a, double dt, int reduce, hipStream_t st,
                                             const IBox* sub = nullptr) {
    IBox b = sub ? *sub : valid_box(g);
    const unsigned nb = red_blocks(b.npts());
    const SourceCoef kf = source_coef(a, dt);
#define RK2(I, R, C) hipLaunchKernelGGL((k_source_rk2<NS, I, R, C>), dim3(nb), dim3(TPB), 0, st, g, b, a, kf, dt)
#define RK2C(I, R) \
    do {           \
        if (a.coll.on) RK2(I, R, true); else RK2(I, R, false); \
    } while (0)
    if (a.implicit) {
        if (reduce) RK2C(true, true); else RK2C(true, false);
    } else {
        if (reduce) RK2C(false, true); else RK2C(false, false);
    }
#undef RK2C
#undef RK2
    if (reduce) {
        for (int s = 0; s < NS; ++s)
            hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(TPB), 0, st, a.part_speed[s], (int)nb, 1, a.out_speed[s]);
        if (a.limits_dt) hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(TPB), 0, st, a.part_dt, (int)nb, 0, a.out_dt);
    }
}
static void l_source_coef(const SourceArgs& a, double dt, double out[3]) {
    const SourceCoef k = source_coef(a, dt);
    out[0] = k.f1;
    out[1] = k.f2;
    out[2] = k.f3;
}
// the same stage on a sub-box of the valid cells (no reductions): stage A split into the frame the ghost
// fills read and the interior, so that the interior overlaps the halo exchange
static int l_source_fused_box(const Geo& g, const SourceArgs& a, double dt, const IBox& box, hipStream_t st) {
    if (box.npts() <= 0) return 1;
    switch (a.n_species) {
    case 1: l_source_rk2_n<1>(g, a, dt, 0, st, &box); break;
    case 2: l_source_rk2_n<2>(g, a, dt, 0, st, &box); break;
    case 3: l_source_rk2_n<3>(g, a, dt, 0, st, &box); break;
    case 4: l_source_rk2_n<4>(g, a, dt, 0, st, &box); break;
    default: return 0;
    }
    return 1;
}
static int l_source_fused(const Geo& g, const SourceArgs& a, double dt, int reduce, hipStream_t st) {
    switch (a.n_species) {
    case 1: l_source_rk2_n<1>(g, a, dt, reduce, st); break;
    case 2: l_source_rk2_n<2>(g, a, dt, reduce, st); break;
    case 3: l_source_rk2_n<3>(g, a, dt, reduce, st); break;
    case 4: l_source_rk2_n<4>(g, a, dt, reduce, st); break;
    default: return 0;
    }
    return 1;
}

static void l_max_speed(const Geo& g, const StatePhys& sp, const double* U, double* part, unsigned long long* out,
                        hipStream_t st) {
    IBox b = valid_box(g);
    const unsigned nb = red_blocks(b.npts());
    hipLaunchKernelGGL(k_max_speed, dim3(nb), dim3(TPB), 0, st, g, b, sp.hp, U, part);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(TPB), 0, st, part, (int)nb, 1, out);
}
static void l_viscous(const Geo& g, const StatePhys& sp, int d, const double* U, double* flux, hipStream_t st) {
    IBox f = valid_box(g);
    f.hi[d] += 1;  // faces lo.d .. hi.d + 1 of the box (MFP_hydro.cpp:1131,1208)
    if (d == 0)
        LAUNCH((k_viscous<0>), f.npts(), st, g, f, sp.hp, sp.visc, U, flux);
    else
        LAUNCH((k_viscous<1>), f.npts(), st, g, f, sp.hp, sp.visc, U, flux);
}
static void l_visc_stability(const Geo& g, const StatePhys& sp, const double* U, double* part, unsigned long long* out,
                             hipStream_t st) {
    IBox b = valid_box(g);
    const unsigned nb = red_blocks(b.npts());
    hipLaunchKernelGGL(k_visc_stability, dim3(nb), dim3(TPB), 0, st, g, b, sp.hp, sp.visc, U, part);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(TPB), 0, st, part, (int)nb, 1, out);
}
static void l_plasma_dt(const Geo& g, const SourceArgs& a, double* part, unsigned long long* out, hipStream_t st) {
    IBox b = valid_box(g);
    const unsigned nb = red_blocks(b.npts());
    hipLaunchKernelGGL(k_plasma_dt, dim3(nb), dim3(TPB), 0, st, g, b, a, part);
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(TPB), 0, st, part, (int)nb, 0, out);
}

static IBox ghost_slab(const Geo& g, int d, int ngs, int lo_other_full) {
    // other dims: full ghosted extent; dim d: ngs layers (coordinates 1..ngs / 0..ngs-1 set by caller)
    IBox s;
    for (int t = 0; t < 3; ++t) {
        s.lo[t] = -g.ng[t];
        s.hi[t] = g.n[t] - 1 + g.ng[t];
    }
    (void)lo_other_full;
    s.lo[d] = 1;
    s.hi[d] = ngs;
    return s;
}
static void l_fill_dim(const Geo& g, int nc, int ngs, double* A, int d, int side, FillRules rules, hipStream_t st) {
    IBox s = ghost_slab(g, d, ngs, 1);
    LAUNCH(k_fill_dim, s.npts(), st, g, s, nc, ngs, d, side, rules, A);
}
static void l_fill_multi(const Geo& g, int d, const FillJobs& J, hipStream_t st) {
    long npts = 0;
    for (int k = 0; k < J.n; ++k) npts = std::max(npts, ghost_slab(g, d, J.j[k].ngs, 1).npts());
    if (J.n <= 0 || npts <= 0) return;
    hipLaunchKernelGGL(k_fill_multi, dim3(nblocks(npts), 2 * J.n), dim3(TPB), 0, st, g, d, J);
}
static void l_pack(const Geo& g, int nc, int ngs, double* A, int d, int side, double* buf, int unpack, hipStream_t st) {
    IBox s = ghost_slab(g, d, ngs, 1);
    int first;
    if (!unpack)
        first = side == 0 ? 0 : g.n[d] - ngs;  // valid layers next to the face
    else
        first = side == 0 ? -ngs : g.n[d];     // ghost layers
    LAUNCH(k_pack, s.npts(), st, g, s, nc, d, first, A, buf, unpack);
}

static int l_leaf_reconstruct(int recon, long n, const double* s, double* slope, double* lo, double* hi, hipStream_t st) {
    switch (recon) {
    case 0: LAUNCH(k_leaf_recon<0>, n, st, n, s, slope, lo, hi); break;
    case 1: LAUNCH(k_leaf_recon<1>, n, st, n, s, slope, lo, hi); break;
    case 2: LAUNCH(k_leaf_recon<2>, n, st, n, s, slope, lo, hi); break;
    case 3: LAUNCH(k_leaf_recon<3>, n, st, n, s, slope, lo, hi); break;
    case 4: LAUNCH(k_leaf_recon<4>, n, st, n, s, slope, lo, hi); break;
    case 5: LAUNCH(k_leaf_recon<5>, n, st, n, s, slope, lo, hi); break;
    case 6: LAUNCH(k_leaf_recon<6>, n, st, n, s, slope, lo, hi); break;
    default: return 0;
    }
    return 1;
}
static int l_leaf_riemann(int flux, long n, const double* L, const double* R, double* F, double c0, double ch, double thr,
                          hipStream_t st) {
    switch (flux) {
    case 0: LAUNCH(k_leaf_riemann_hydro<0>, n, st, n, L, R, F, thr); break;
    case 1: LAUNCH(k_leaf_riemann_hydro<1>, n, st, n, L, R, F, thr); break;
    case 2: LAUNCH(k_leaf_riemann_hydro<2>, n, st, n, L, R, F, thr); break;
    case 10: LAUNCH(k_leaf_riemann_field, n, st, n, L, R, F, c0, ch); break;
    default: return 0;
    }
    return 1;
}
static int l_leaf_cons2prim(long n, double mass, double gamma, const double* U, double* Q, double* sp, hipStream_t st) {
    GasP G{mass, 0.0, gamma};
    LAUNCH(k_leaf_cons2prim, n, st, n, G, U, Q, sp);
    return 1;
}
static int l_leaf_prim2cons(long n, double mass, double gamma, const double* Q, double* U, hipStream_t st) {
    GasP G{mass, 0.0, gamma};
    LAUNCH(k_leaf_prim2cons, n, st, n, G, Q, U);
    return 1;
}

// defined in mfp_fused.hip
int ctu_fused(const Geo&, const StatePhys&, const double* Uin, double* Uout, int do_ctu, double dt,
              const int touches[3][2], int* err, unsigned long long* red, int part, hipStream_t);
#define CMFP_GATHER_DECL(D)                                                                                          \
    int ctu_gather_##D(const Geo&, const StatePhys&, const double* Uin, double* Uout, const Ptr3& F1, const Ptr3& F2, \
                       int do_ctu, double dt, const int touches[3][2], int* err, unsigned long long* red, hipStream_t);
CMFP_GATHER_DECL(1) CMFP_GATHER_DECL(2) CMFP_GATHER_DECL(3)
#undef CMFP_GATHER_DECL
// defined in mfp_field3d.hip
int ctu_field_src(const Geo&, const StatePhys&, const double* Uin, double* Uout, int do_ctu, double dt, const int touches[3][2],
                  const SourceEpi& epi, int part, hipStream_t);
int ctu_field3d(const Geo&, const StatePhys&, const double* Uin, double* Uout, const Ptr3& F1, double* DU, int do_ctu, double dt,
                const int touches[3][2], const SourceEpi* epi, hipStream_t);
// defined in mfp_march_field.hip
int ctu_march_field(const Geo&, const StatePhys&, const double* Uin, double* Uout, int do_ctu, double dt, const int touches[3][2],
                    const SourceEpi* epi, int part, const SourcePro* pro, hipStream_t);
void field_vacuum_check(const Geo&, const double* U, int* flag, hipStream_t);
int ctu_interior(const Geo&, const StatePhys&, int do_ctu, int flags, IBox* box);
// defined in mfp_march.hip
int ctu_march2(const Geo&, const StatePhys* const sp[2], const double* const Uin[2], double* const Uout[2], int do_ctu, double dt,
               const int touches[3][2], int* err, unsigned long long* const red[2], int part, const SourcePro* pro, hipStream_t);
int ctu_march_tab(const Geo&, const StatePhys&, const double* Uin, double* Uout, int do_ctu, double dt, const int touches[3][2],
                  int* err, unsigned long long* red, int part, const MarchTab& tab, hipStream_t);
static int l_ctu_gather(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, const Ptr3& F1, const Ptr3& F2,
                        int do_ctu, double dt, const int touches[3][2], int* err, unsigned long long* red, hipStream_t st) {
    if (sp.kind == 0 && sp.hp.nt) return 0;  // passive scalars: the 2-D marching kernel or stage by stage
    if (g.dim == 1) return ctu_gather_1(g, sp, Uin, Uout, F1, F2, do_ctu, dt, touches, err, red, st);
    if (g.dim == 2) return ctu_gather_2(g, sp, Uin, Uout, F1, F2, do_ctu, dt, touches, err, red, st);
    return ctu_gather_3(g, sp, Uin, Uout, F1, F2, do_ctu, dt, touches, err, red, st);
}
// MFP::estTimeStep (MFP_timestep.cpp:4-21: cfl * min over states and actions) and the growth cap of
// MFP::computeNewDt (:86-93) on the device; the same expressions as the host path of
// cmfp_est_time_step / cmfp_compute_new_dt
__device__ __forceinline__ void new_dt_body(const DtPlan& p, const unsigned long long* red, double* dt, double* log, int idx) {
    double est = DBL_MAX;
    for (int n = 0; n < p.n; ++n) {
        double d;
        if (p.kind[n] == 0) {  // EulerianState::get_allowed_time_step MFP_eulerian.cpp:302-303
            const double sp = __longlong_as_double((long long)red[p.word[n]]);
            d = p.dx[0] / sp;
            for (int i = 1; i < p.dim; ++i) d = std_min(d, p.dx[i] / sp);
        } else if (p.kind[n] == 1) {
            d = p.cst[n];
        } else {  // Plasma5::get_allowed_time_step MFP_plasma5.cpp:457-567
            d = std_min(p.cst[n], __longlong_as_double((long long)red[p.word[n]]));
        }
        est = std_min(est, d);
    }
    est *= p.cfl;
    if (p.force_dt > 0.0) est = p.force_dt;  // MFP_timestep.cpp:8
    const double old = dt[0];
    log[idx] = old;
    dt[0] = std_min(est, 1.1 * old);
}
__global__ void k_new_dt(DtPlan p, const unsigned long long* red, double* dt, double* log, int idx) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    new_dt_body(p, red, dt, log, idx);
}
// the step's fold of the reduction words (k_reduce_many) and the dt update in one launch: one workgroup folds the
// jobs one after the other, then its first thread — which wrote the folded words — applies estTimeStep / computeNewDt
__global__ __launch_bounds__(256) void k_fold_new_dt(RedJobs J, DtPlan p, const unsigned long long* red, double* dt, double* log, int idx) {
    __shared__ double sh[4];
    for (int job = 0; job < J.n; ++job) {
        reduce_job(J, job, sh);
        __syncthreads();
    }
    if (threadIdx.x == 0) new_dt_body(p, red, dt, log, idx);
}
static void l_fold_new_dt(const RedJobs& J, const DtPlan& p, const unsigned long long* red, double* dt, double* log, int idx, hipStream_t st) {
    hipLaunchKernelGGL(k_fold_new_dt, dim3(1), dim3(256), 0, st, J, p, red, dt, log, idx);
}
static void l_new_dt(const DtPlan& p, const unsigned long long* red, double* dt, double* log, int idx, hipStream_t st) {
    hipLaunchKernelGGL(k_new_dt, dim3(1), dim3(64), 0, st, p, red, dt, log, idx);
}

static void l_reduce_final(const double* part, int n, int is_max, unsigned long long* out, hipStream_t st) {
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(TPB), 0, st, part, n, is_max, out);
}

const Launchers* launchers() {
    static const Launchers L = {l_prim,      l_recon,         l_face_bc,        l_fluxes,          l_correct,
                                l_divergence, l_copy,         l_zero,           l_add,             l_lincomb,
                                l_source,    l_max_speed,     l_plasma_dt,      l_viscous,   l_visc_stability, l_table_store, l_fill_dim,        l_pack, l_fill_multi,
                                ctu_fused,   l_source_fused,  l_collisions, l_ctu_gather, ctu_field3d, ctu_field_src, ctu_interior, ctu_march2, ctu_march_field, field_vacuum_check, l_source_coef, l_source_fused_box, l_leaf_reconstruct, l_leaf_riemann,  l_leaf_cons2prim,
                                l_leaf_prim2cons, l_reduce_final, l_reduce_many, l_first_invalid, l_new_dt, ctu_march_tab, l_fold_new_dt, l_collisions_mix};
    return &L;
}

}  // namespace CMFP_NS
