// mfp_source.h — per-cell source terms shared by the pointwise source kernels (mfp_kernels.hip) and
// the source epilogue of the fused field kernel (mfp_fused.hip).
#pragma once
#include <cfloat>

#include "mfp_device.h"

namespace CMFP_NS {
using namespace cmfp;

struct SourceCoef {
    double f1, f2, f3;
};
__host__ __device__ static inline SourceCoef source_coef(const SourceArgs& a, double dt) {
    // MFP_plasma5.cpp:123-129
    const double c_h = a.c_h;
    const double c_d = a.div_damp != 0.0 ? c_h / a.div_damp : 0.0;
    SourceCoef k;
    k.f1 = dt * a.Larmor / (a.Debye * a.Debye * a.lightspeed);
    k.f2 = dt * c_h * c_h * k.f1 / a.lightspeed;
    k.f3 = c_d != 0.0 ? dt * c_h * c_h / (c_d * c_d) : 0.0;
    return k;
}

// One cell of Plasma5::explicit_solve (MFP_plasma5.cpp:161-236) /
// Lorentz::calc_time_derivative (MFP_lorentz.cpp:129-189).  fU: field comps,
// sU[n]: species comps {rho,mx,my,mz,E}.  Outputs are the increments.
template <int MAXS>
CMFP_DEV void plasma5_explicit_cell(const SourceArgs& a, const SourceCoef& kf, double dt, const double* fU,
                                    const double (*sU)[5], double (*sdU)[4], double* fdU /*Dx,Dy,Dz,phi,psi*/) {
    const double Larmor = a.Larmor, lightspeed = a.lightspeed;
    const double ep = fU[F_EP];
    const double Bx = fU[F_BX], By = fU[F_BY], Bz = fU[F_BZ];
    const double Ex = fU[F_DX] / ep, Ey = fU[F_DY] / ep, Ez = fU[F_DZ] / ep;
    double charge_density = 0.0, current_x = 0.0, current_y = 0.0, current_z = 0.0;
#pragma unroll
    for (int n = 0; n < MAXS; ++n) {
        if (n < a.n_species) {
            const double rho = sU[n][0], mx = sU[n][1], my = sU[n][2], mz = sU[n][3];
            const double r = a.r[n];
            charge_density += rho * r;
            current_x += r * mx;
            current_y += r * my;
            current_z += r * mz;
            const double u = mx / rho, v = my / rho, w = mz / rho;
            sdU[n][0] = dt * (rho * r / Larmor) * (lightspeed * Ex + v * Bz - w * By);
            sdU[n][1] = dt * (rho * r / Larmor) * (lightspeed * Ey + w * Bx - u * Bz);
            sdU[n][2] = dt * (rho * r / Larmor) * (lightspeed * Ez + u * By - v * Bx);
            sdU[n][3] = dt * (rho * r * lightspeed) / Larmor * (u * Ex + v * Ey + w * Ez);
        }
    }
    fdU[0] = -kf.f1 * current_x;
    fdU[1] = -kf.f1 * current_y;
    fdU[2] = -kf.f1 * current_z;
    fdU[3] = -fU[F_PHI] * kf.f3 + kf.f2 * charge_density;
    fdU[4] = -fU[F_PSI] * kf.f3;
}

// per-cell contributions
// EulerianState::get_allowed_time_step MFP_eulerian.cpp:290-296
CMFP_DEV double cell_max_speed(const GasP& G, const double* U, int dim, double m) {
    double s[3];
    hydro_speed_from_cons(G, U, s);
    for (int d = 0; d < dim; ++d) m = std_max(m, s[d]);
    return m;
}
// Plasma5::get_allowed_time_step MFP_plasma5.cpp:535-555
CMFP_DEV double cell_plasma_dt(double Larmor, double Debye, double Bx, double By, double Bz, double rho, double r, double dt) {
    const double PI = 3.141592653589793;  // acos(-1)
    const double D2 = Debye * Debye;
    const double B = sqrt(Bx * Bx + By * By + Bz * Bz);
    const double omega_p = 10 * sqrt(rho * r * r / D2) / (2 * PI);
    if (omega_p > 0) dt = std_min(dt, 1 / omega_p);
    const double omega_c = 10 * (fabs(r) * B) / (Larmor * 2 * PI);
    if (omega_c > 0) dt = std_min(dt, 1 / omega_c);
    return dt;
}


// One Strang source stage with its inner RK2 (MFP_advance.cpp:146-197 / 235-281) for one cell,
// explicit plasma5 / Lorentz, the arithmetic of k_source_rk2 in the same order:
//   U1 = U0 + S(U0) ; U2 = U1 + S(U1) ; new = 0.5 U0 + 0.5 U2
// f0: field comps, s0: species comps; fnew = {Dx,Dy,Dz,phi,psi}, snew[s] = {mx,my,mz,E} after the stage.
template <int NS>
CMFP_DEV void source_rk2_explicit_cell(const SourceArgs& a, const SourceCoef& kf, double dt, const double* f0,
                                       const double (*s0)[5], double* fnew, double (*snew)[4]) {
    double f1[10], s1[NS][5], sdU[NS][4], fdU[5];
    plasma5_explicit_cell<NS>(a, kf, dt, f0, s0, sdU, fdU);
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
    plasma5_explicit_cell<NS>(a, kf, dt, f1, s1, sdU, fdU);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const double U2 = s1[s][n + 1] + sdU[s][n];
            snew[s][n] = 0.5 * s0[s][n + 1] + 0.5 * U2;
        }
    fnew[0] = 0.5 * f0[F_DX] + 0.5 * (f1[F_DX] + fdU[0]);
    fnew[1] = 0.5 * f0[F_DY] + 0.5 * (f1[F_DY] + fdU[1]);
    fnew[2] = 0.5 * f0[F_DZ] + 0.5 * (f1[F_DZ] + fdU[2]);
    fnew[3] = 0.5 * f0[F_PHI] + 0.5 * (f1[F_PHI] + fdU[3]);
    fnew[4] = 0.5 * f0[F_PSI] + 0.5 * (f1[F_PSI] + fdU[4]);
}

// kernel argument of the source epilogue (SRC species; nothing without one)
struct NoEpi {};
template <int SRC> struct EpiArg { typedef SourceEpi type; };
template <> struct EpiArg<0> { typedef NoEpi type; };

// Stage C on one updated cell of the field's CTU kernel: Unew = the field after its CTU update (in/out),
// the species are read from and written to their new-time arrays at offset c; the CFL quantities of what
// was written are folded into src_speed / src_dt (k_source_rk2<REDUCE>).
// the species' new-time values at cell c, as the epilogue reads them (a kernel may request them ahead of time)
template <int NS> CMFP_DEV void source_epilogue_load(const Geo& g, const SourceEpi& epi, long c, double (*s0)[5]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int n = 0; n < 5; ++n) s0[s][n] = epi.a.sp_U[s][c + n * g.sc];
}
template <int NS>
CMFP_DEV void source_epilogue_cell_pre(const Geo& g, const SourceEpi& epi, double dt, long c, double* Unew, const double (*s0)[5],
                                       double* src_speed, double& src_dt);
template <int NS>
CMFP_DEV void source_epilogue_cell(const Geo& g, const SourceEpi& epi, double dt, long c, double* Unew, double* src_speed,
                                   double& src_dt) {
    double s0[NS][5];
    source_epilogue_load<NS>(g, epi, c, s0);
    source_epilogue_cell_pre<NS>(g, epi, dt, c, Unew, s0, src_speed, src_dt);
}
template <int NS>
CMFP_DEV void source_epilogue_cell_pre(const Geo& g, const SourceEpi& epi, double dt, long c, double* Unew, const double (*s0)[5],
                                       double* src_speed, double& src_dt) {
    const SourceArgs& a = epi.a;
    const double hdt = 0.5 * dt;
    SourceCoef kf{epi.kf[0], epi.kf[1], epi.kf[2]};
    if (g.dtp) kf = source_coef(a, hdt);
    double snew[NS][4], fnew[5];
    source_rk2_explicit_cell<NS>(a, kf, hdt, Unew, s0, fnew, snew);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        double Un[5];
        Un[0] = s0[s][0];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            Un[n + 1] = snew[s][n];
            a.sp_U[s][c + (n + 1) * g.sc] = snew[s][n];
        }
        GasP G{1.0, 0.0, a.gamma[s]};
        src_speed[s] = cell_max_speed(G, Un, g.dim, src_speed[s]);
        if (a.limits_dt) src_dt = cell_plasma_dt(a.Larmor, a.Debye, Unew[F_BX], Unew[F_BY], Unew[F_BZ], Un[0], a.r[s], src_dt);
    }
    if (!a.lorentz_only) {
        Unew[F_DX] = fnew[0];
        Unew[F_DY] = fnew[1];
        Unew[F_DZ] = fnew[2];
        Unew[F_PHI] = fnew[3];
        Unew[F_PSI] = fnew[4];
#ifndef CMFP_FASTMATH
        // components no source term touches: the reference's 0.5 x + 0.5 x, which is x except for subnormal
        // values with an odd last bit (see k_source_rk2)
#pragma unroll
        for (int n : {F_BX, F_BY, F_BZ, F_MU, F_EP}) Unew[n] = 0.5 * Unew[n] + 0.5 * Unew[n];
#endif
    }
}

// workgroup reduction of the epilogue's CFL quantities and one integer atomic per word on slot `slot`;
// shr: >= 4 doubles per wave of LDS nobody else uses any more; every thread of the workgroup calls it
template <int NS>
CMFP_DEV void source_epilogue_reduce(const SourceEpi& epi, double* src_speed, double src_dt, double* shr, int tid, int nwaves,
                                     int slot) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) src_speed[s] = fmax(src_speed[s], __shfl_xor(src_speed[s], o, 64));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) src_dt = fmin(src_dt, __shfl_xor(src_dt, o, 64));
    if ((tid & 63) == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) shr[(tid >> 6) * 4 + s] = src_speed[s];
        shr[(tid >> 6) * 4 + 3] = src_dt;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w2 = 1; w2 < nwaves; ++w2) {
#pragma unroll
            for (int s = 0; s < NS; ++s) src_speed[s] = fmax(src_speed[s], shr[w2 * 4 + s]);
            src_dt = fmin(src_dt, shr[w2 * 4 + 3]);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) atomicMax(&epi.red_speed[s][slot], (unsigned long long)__double_as_longlong(src_speed[s]));
        if (epi.a.limits_dt) atomicMin(&epi.red_dt[slot], (unsigned long long)__double_as_longlong(src_dt));
    }
}

}  // namespace CMFP_NS
