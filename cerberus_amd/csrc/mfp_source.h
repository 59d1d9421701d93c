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
// (a: SourceArgs, or any type with n_species, r[], Larmor, lightspeed — ProArgs below)
// rc: charge / mass of this cell's species where a species is a mixture (species_r below); nullptr: a.r
template <int MAXS, class Args>
CMFP_DEV void plasma5_explicit_cell(const Args& a, const SourceCoef& kf, double dt, const double* fU,
                                    const double (*sU)[5], double (*sdU)[4], double* fdU /*Dx,Dy,Dz,phi,psi*/,
                                    const double* rc = nullptr) {
    // FAST: no contraction left to the compiler here — the fused multiply-adds are written out, so that every
    // inlined copy of this function (the marching kernels hold several) rounds alike and a cell's result does not
    // depend on which launch, segment or row parity it falls in.  (STRICT builds never contract.)
#pragma clang fp contract(off)
    const double Larmor = a.Larmor, lightspeed = a.lightspeed;
    const double ep = fU[F_EP];
    const double Bx = fU[F_BX], By = fU[F_BY], Bz = fU[F_BZ];
    const double Ex = fU[F_DX] / ep, Ey = fU[F_DY] / ep, Ez = fU[F_DZ] / ep;
    double charge_density = 0.0, current_x = 0.0, current_y = 0.0, current_z = 0.0;
#pragma unroll
    for (int n = 0; n < MAXS; ++n) {
        if (n < a.n_species) {
            const double rho = sU[n][0], mx = sU[n][1], my = sU[n][2], mz = sU[n][3];
            const double r = rc ? rc[n] : a.r[n];
#ifdef CMFP_FASTMATH
            charge_density = fma(rho, r, charge_density);
            current_x = fma(r, mx, current_x);
            current_y = fma(r, my, current_y);
            current_z = fma(r, mz, current_z);
            const double rhoinv = f_rcp(rho);
            const double u = mx * rhoinv, v = my * rhoinv, w = mz * rhoinv;
            const double k = dt * (rho * r / Larmor);
            sdU[n][0] = k * fma(v, Bz, fma(-w, By, lightspeed * Ex));
            sdU[n][1] = k * fma(w, Bx, fma(-u, Bz, lightspeed * Ey));
            sdU[n][2] = k * fma(u, By, fma(-v, Bx, lightspeed * Ez));
            sdU[n][3] = dt * (rho * r * lightspeed) / Larmor * fma(w, Ez, fma(v, Ey, u * Ex));
#else
            charge_density += rho * r;
            current_x += r * mx;
            current_y += r * my;
            current_z += r * mz;
            const double u = mx / rho, v = my / rho, w = mz / rho;
            sdU[n][0] = dt * (rho * r / Larmor) * (lightspeed * Ex + v * Bz - w * By);
            sdU[n][1] = dt * (rho * r / Larmor) * (lightspeed * Ey + w * Bx - u * Bz);
            sdU[n][2] = dt * (rho * r / Larmor) * (lightspeed * Ez + u * By - v * Bx);
            sdU[n][3] = dt * (rho * r * lightspeed) / Larmor * (u * Ex + v * Ey + w * Ez);
#endif
        }
    }
#ifdef CMFP_FASTMATH
    fdU[0] = -kf.f1 * current_x;
    fdU[1] = -kf.f1 * current_y;
    fdU[2] = -kf.f1 * current_z;
    fdU[3] = fma(kf.f2, charge_density, -fU[F_PHI] * kf.f3);
    fdU[4] = -fU[F_PSI] * kf.f3;
#else
    fdU[0] = -kf.f1 * current_x;
    fdU[1] = -kf.f1 * current_y;
    fdU[2] = -kf.f1 * current_z;
    fdU[3] = -fU[F_PHI] * kf.f3 + kf.f2 * charge_density;
    fdU[4] = -fU[F_PSI] * kf.f3;
#endif
}

// r = q / m of species s at the cell with plane offset c (MFP_plasma5.cpp:199-202): the constants, or for a mixture
// get_charge_from_cons / get_mass_from_cons (MFP_hydro_gas.cpp:72-96,123-150) from alpha_i = tracer_i / rho
CMFP_DEV double species_r(const Geo& g, const SourceArgs& a, int s, long c, double rho) {
    const cmfp::SpeciesMix& M = a.mix[s];
    if (M.nt == 0) return a.r[s];
    double S_aq = 0.0, S_am = 0.0, S_a = 0.0;
    for (int i = 0; i < 2; ++i)
        if (i < M.nt) {
            double al = f_div(a.sp_U[s][c + (5 + i) * g.sc], rho);
            al = (al < 0.0) ? 0.0 : ((1.0 < al) ? 1.0 : al);
            S_aq += f_div(al * M.q[i], M.m[i]);
            S_am += f_div(al, M.m[i]);
            S_a += al;
        }
    S_aq += f_div((1.0 - S_a) * M.q[M.nt], M.m[M.nt]);
    S_am += f_div(1.0 - S_a, M.m[M.nt]);
    const double m = (M.flags & 1) ? f_div(1.0, S_am) : M.m[0];
    const double q = (M.flags & 2) ? f_div(S_aq, S_am) : M.q[0];
    return q / m;
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

// The plasma / cyclotron dt limit of the cells one thread has written in a source epilogue.  STRICT: the running
// minimum of cell_plasma_dt.  FAST: that limit decreases monotonically with rho_s (plasma frequency) and with |B|^2
// (cyclotron frequency), so the thread keeps the maxima of those and evaluates cell_plasma_dt once at the end —
// the two roots and four divisions per cell were a sixteenth of the field marching kernel's instructions.
struct SrcDt {
#ifdef CMFP_FASTMATH
    double rho_max[2] = {0.0, 0.0};
    double b2_max = 0.0;
#else
    double dt = DBL_MAX;
#endif
};
template <int NS> CMFP_DEV double src_dt_value(const SourceArgs& a, const SrcDt& acc) {
#ifdef CMFP_FASTMATH
    double dt = DBL_MAX;
    const double b = sqrt(acc.b2_max);
#pragma unroll
    for (int s = 0; s < NS; ++s) dt = cell_plasma_dt(a.Larmor, a.Debye, b, 0.0, 0.0, acc.rho_max[s], a.r[s], dt);
    return dt;
#else
    return acc.dt;
#endif
}

// One Strang source stage with its inner RK2 (MFP_advance.cpp:146-197 / 235-281) for one cell,
// explicit plasma5 / Lorentz, the arithmetic of k_source_rk2 in the same order:
//   U1 = U0 + S(U0) ; U2 = U1 + S(U1) ; new = 0.5 U0 + 0.5 U2
// f0: field comps, s0: species comps; fnew = {Dx,Dy,Dz,phi,psi}, snew[s] = {mx,my,mz,E} after the stage.
template <int NS, class Args>
CMFP_DEV void source_rk2_explicit_cell(const Args& a, const SourceCoef& kf, double dt, const double* f0,
                                       const double (*s0)[5], double* fnew, double (*snew)[4]) {
#pragma clang fp contract(off)  // as in plasma5_explicit_cell
#ifdef CMFP_FASTMATH
    auto half_sum = [](double x, double y) { return fma(0.5, x, 0.5 * y); };
#else
    auto half_sum = [](double x, double y) { return 0.5 * x + 0.5 * y; };
#endif
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
            snew[s][n] = half_sum(s0[s][n + 1], U2);
        }
    fnew[0] = half_sum(f0[F_DX], f1[F_DX] + fdU[0]);
    fnew[1] = half_sum(f0[F_DY], f1[F_DY] + fdU[1]);
    fnew[2] = half_sum(f0[F_DZ], f1[F_DZ] + fdU[2]);
    fnew[3] = half_sum(f0[F_PHI], f1[F_PHI] + fdU[3]);
    fnew[4] = half_sum(f0[F_PSI], f1[F_PSI] + fdU[4]);
}

// ---- Strang source stage A as the prologue of the CTU kernels (2-D, explicit plasma5 / Lorentz over two species).  Stage A (MFP_advance.cpp:146-197) is pointwise, so a
// CTU kernel can apply it to every cell it loads — ghost cells included: a periodic or neighbour copy of the raw
// values gives the same result as the copy of the result — instead of a launch of its own that reads and
// rewrites all states.  Each kernel evaluates what its state needs of the coupled RK2 (a species: its own
// momentum and energy; the field: D, phi, psi) and the compiler drops the rest.
// (kernel argument: SourcePro, mfp_types.h)
struct ProArgs {  // what plasma5_explicit_cell / source_rk2_explicit_cell / source_coef read
    int n_species;
    double r[2];
    double Larmor, Debye, lightspeed, c_h, div_damp;
    int lorentz_only;
};
__host__ __device__ static inline SourceCoef source_coef(const ProArgs& a, double dt) {
    const double c_h = a.c_h;
    const double c_d = a.div_damp != 0.0 ? c_h / a.div_damp : 0.0;
    SourceCoef k;
    k.f1 = dt * a.Larmor / (a.Debye * a.Debye * a.lightspeed);
    k.f2 = dt * c_h * c_h * k.f1 / a.lightspeed;
    k.f3 = c_d != 0.0 ? dt * c_h * c_h / (c_d * c_d) : 0.0;
    return k;
}
// species `first` is entry 0 of the argument's lists (a hydro wave lists its own species first)
CMFP_DEV ProArgs pro_args(const SourcePro& p, int first) {
    ProArgs a;
    a.n_species = 2;
    a.r[0] = p.r[first];
    a.r[1] = p.r[1 - first];
    a.Larmor = p.Larmor, a.Debye = p.Debye, a.lightspeed = p.lightspeed, a.c_h = p.c_h, a.div_damp = p.div_damp;
    a.lorentz_only = p.lorentz_only;
    return a;
}
// Sign masks of the cell (i, row) at the physical faces of the block (SourcePro::odd_*): which components of
// the hydro (.x) and field (.y) states a ghost fill has negated there; 0 for a valid cell
struct ProMask {
    unsigned h, f;
};
CMFP_DEV ProMask pro_mask(const SourcePro& p, int i, int row, int n0, int n1) {
    ProMask m{0u, 0u};
    if (p.touch[0][0] && i < 0) m.h ^= p.odd_h[0][0], m.f ^= p.odd_f[0][0];
    if (p.touch[0][1] && i >= n0) m.h ^= p.odd_h[0][1], m.f ^= p.odd_f[0][1];
    if (p.touch[1][0] && row < 0) m.h ^= p.odd_h[1][0], m.f ^= p.odd_f[1][0];
    if (p.touch[1][1] && row >= n1) m.h ^= p.odd_h[1][1], m.f ^= p.odd_f[1][1];
    return m;
}
// v with its sign flipped where bit `bit` of mask is set
CMFP_DEV double pro_flip(double v, unsigned mask, int bit) {
    const int hi = __double2hiint(v) ^ (int)(((mask >> bit) & 1u) << 31);
    return __hiloint2double(hi, __double2loint(v));
}

// A hydro wave: own = {rho, mx, my, mz, E} before stage A (in) / after it (out); oth = {mx, my, mz} of the other
// species (its current enters E after the first RK2 stage); f = {Dx, Dy, Dz, Bx, By, Bz}, vacuum (ep = 1).
CMFP_DEV void source_prologue_species(const ProArgs& a, const SourceCoef& kf, double hdt, double* own, const double* oth,
                                      const double* f) {
    const double f0[10] = {f[0], f[1], f[2], f[3], f[4], f[5], 0.0, 0.0, 1.0, 1.0};
    const double s0[2][5] = {{own[0], own[1], own[2], own[3], own[4]}, {1.0, oth[0], oth[1], oth[2], 0.0}};
    double fnew[5], snew[2][4];
    source_rk2_explicit_cell<2>(a, kf, hdt, f0, s0, fnew, snew);
#pragma unroll
    for (int n = 0; n < 4; ++n) own[n + 1] = snew[0][n];
#ifndef CMFP_FASTMATH
    // the density, which no source term touches: the reference's LinComb 0.5 x + 0.5 x, which is x except for
    // subnormal values with an odd last bit (see k_source_rk2)
    own[0] = 0.5 * own[0] + 0.5 * own[0];
#endif
}
// A field wave: f = {Dx, Dy, Dz, Bx, By, Bz, phi, psi} (in/out); sa, sb = {rho, mx, my, mz} of the two species
CMFP_DEV void source_prologue_field(const ProArgs& a, const SourceCoef& kf, double hdt, double* f, const double* sa,
                                    const double* sb) {
    if (a.lorentz_only) return;
    const double f0[10] = {f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], 1.0, 1.0};
    const double s0[2][5] = {{sa[0], sa[1], sa[2], sa[3], 0.0}, {sb[0], sb[1], sb[2], sb[3], 0.0}};
    double fnew[5], snew[2][4];
    source_rk2_explicit_cell<2>(a, kf, hdt, f0, s0, fnew, snew);
    f[F_DX] = fnew[0], f[F_DY] = fnew[1], f[F_DZ] = fnew[2], f[F_PHI] = fnew[3], f[F_PSI] = fnew[4];
#ifndef CMFP_FASTMATH
#pragma unroll
    for (int n : {F_BX, F_BY, F_BZ}) f[n] = 0.5 * f[n] + 0.5 * f[n];  // as above (mu = ep = 1 on a vacuum block: exact)
#endif
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
                                       double* src_speed, SrcDt& src_dt);
template <int NS>
CMFP_DEV void source_epilogue_cell(const Geo& g, const SourceEpi& epi, double dt, long c, double* Unew, double* src_speed,
                                   SrcDt& src_dt) {
    double s0[NS][5];
    source_epilogue_load<NS>(g, epi, c, s0);
    source_epilogue_cell_pre<NS>(g, epi, dt, c, Unew, s0, src_speed, src_dt);
}
template <int NS>
CMFP_DEV void source_epilogue_cell_pre(const Geo& g, const SourceEpi& epi, double dt, long c, double* Unew, const double (*s0)[5],
                                       double* src_speed, SrcDt& src_dt) {
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
#ifdef CMFP_FASTMATH
        if (a.limits_dt) src_dt.rho_max[s] = fmax(src_dt.rho_max[s], Un[0]);
#else
        if (a.limits_dt) src_dt.dt = cell_plasma_dt(a.Larmor, a.Debye, Unew[F_BX], Unew[F_BY], Unew[F_BZ], Un[0], a.r[s], src_dt.dt);
#endif
    }
#ifdef CMFP_FASTMATH
    if (a.limits_dt) src_dt.b2_max = fmax(src_dt.b2_max, Unew[F_BX] * Unew[F_BX] + Unew[F_BY] * Unew[F_BY] + Unew[F_BZ] * Unew[F_BZ]);
#endif
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
CMFP_DEV void source_epilogue_reduce(const SourceEpi& epi, double* src_speed, const SrcDt& acc, double* shr, int tid, int nwaves,
                                     int slot) {
    double src_dt = src_dt_value<NS>(epi.a, acc);
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
