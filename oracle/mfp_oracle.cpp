// oracle/mfp_oracle.cpp
//
// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// CPU restatement of the PlasmaSimUQ/cerberus (MFP) per-patch hyperbolic update,
// written from a reading of the reference sources; every function cites the
// reference file:line it follows and keeps that code's expression order and
// loop structure (staged FAB-by-FAB pipeline, separate prim / face / flux
// arrays).  It is compiled by oracle/Makefile with g++ -O3 -ffp-contract=off.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library.  Nothing under cerberus_amd/ includes, links or calls it.
//
// PINNING: the reference cannot be built here (its AMReX submodule is an empty
// directory and no stand-in headers are written), and it ships no unit tests or
// golden vectors.  The leaf numerics are pinned by the known-answer vectors the
// survey recorded from the reference's own leaf sources (SURVEY.md §8c,
// tests/golden/survey_kat.json).  The FAB-level orchestration, ghost fill
// (AMReX fab_filcc, source absent) and the time integrator glue have no
// reference fixture: for those, PARITY IS UNPINNED by the reference.
//
// Restrictions: mixtures (species of different mass / charge / gamma) not in collisions / viscous terms, no EB,
// no Lua inflow / dynamic functions, no viscous terms, single AMR level.

#include "../include/cerberus_mfp.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

typedef double Real;

// ---------------------------------------------------------------------------
// index enums: hydro/MFP_hydro_defs.H:9-28, field/MFP_field_defs.H:7
enum HC { Density = 0, Xmom, Ymom, Zmom, Eden, HC_NUM };
enum HP { PDensity = 0, Xvel, Yvel, Zvel, Prs, Temp, Gamma, SpHeat, HP_NUM };
enum FC { Dx = 0, Dy, Dz, Bx, By, Bz, phi, psi, mu, ep, FC_NUM };

static int g_error = 0;  // set when the reference would amrex::Abort
static void flag_abort() {
#pragma omp atomic write
    g_error = 1;
}

// ---------------------------------------------------------------------------
// MFP_utility.H:149-158, 289-298
static inline int sgn(Real val) { return (Real(0) < val) - (val < Real(0)); }
static inline Real sign(Real A, Real B) { return (B >= Real(0)) ? std::abs(A) : -std::abs(A); }
static inline Real tanh_approx(Real x) {
    if (x > 1.92033) return 0.96016;
    if (x > 0) return 0.96016 - 0.26037 * (x - 1.92033) * (x - 1.92033);
    if (x <= -1.92033) return -0.96016;
    return 0.26037 * (x + 1.92033) * (x + 1.92033) - 0.96016;
}

// ---------------------------------------------------------------------------
// Reconstruction family — states/Eulerian/MFP_reconstruction.cpp
static int recon_stencil_length(int r) {
    switch (r) {
    case CMFP_RECON_CONSTANT: return 1;  // :59
    case CMFP_RECON_MINMOD:              // :81
    case CMFP_RECON_VANLEER:             // :123
    case CMFP_RECON_MC:                  // :163
    case CMFP_RECON_CENTRE: return 3;    // :216
    case CMFP_RECON_O6:                  // :247
    case CMFP_RECON_WENO: return 7;      // :283
    }
    return 0;
}
static int recon_num_grow(int r) {
    switch (r) {
    case CMFP_RECON_CONSTANT: return 1;  // :60
    case CMFP_RECON_MINMOD:
    case CMFP_RECON_VANLEER:
    case CMFP_RECON_MC:
    case CMFP_RECON_CENTRE: return 2;    // :82,124,164,217
    case CMFP_RECON_O6:
    case CMFP_RECON_WENO: return 4;      // :248,284
    }
    return 0;
}

// WENOReconstruction::WENOSYMBO  :297-357
static Real wenosymbo(const Real* stencil, int upwind) {
    const int i = 7 / 2;
    const int r = 3;
    const Real C[4] = {0.094647545896, 0.428074212384, 0.408289331408, 0.068988910311};
    const Real akl[4][3] = {{2 / 6.0, -7 / 6.0, 11 / 6.0},
                            {-1 / 6.0, 5 / 6.0, 2 / 6.0},
                            {2 / 6.0, 5 / 6.0, -1 / 6.0},
                            {11 / 6.0, -7 / 6.0, 2 / 6.0}};
    const Real b2 = std::sqrt(13 / 12.0);
    const Real dkml[4][2][3] = {{{1 / 2.0, -4 / 2.0, 3 / 2.0}, {b2, -2 * b2, b2}},
                                {{-1 / 2.0, 0, 1 / 2.0}, {b2, -2 * b2, b2}},
                                {{-3 / 2.0, 4 / 2.0, -1 / 2.0}, {b2, -2 * b2, b2}},
                                {{-5 / 2.0, 8 / 2.0, -3 / 2.0}, {b2, -2 * b2, b2}}};
    const Real epsilon = 1e-10;

    Real IS[4] = {0, 0, 0, 0};
    for (int k = 0; k <= r; ++k) {
        for (int m = 0; m < r - 1; ++m) {
            Real is = 0.0;
            for (int l = 0; l < r; ++l) { is += dkml[k][m][l] * stencil[i - upwind * (r - 1 - k - l)]; }
            IS[k] += is * is;
        }
    }
    if (upwind != 0) {
        Real max_IS = IS[0];
        for (int k = 1; k <= r; ++k) { max_IS = std::max(IS[k], max_IS); }
        IS[r] = max_IS;
    }
    Real alpha[4];
    for (int k = 0; k <= r; ++k) { alpha[k] = C[k] / (epsilon + IS[k]); }
    Real sum_alpha = alpha[0] + alpha[1] + alpha[2];
    Real omega[4];
    for (int k = 0; k <= r; ++k) { omega[k] = alpha[k] / sum_alpha; }
    Real face_value = 0.0;
    Real q;
    for (int k = 0; k < r; ++k) {
        q = 0.0;
        for (int l = 0; l < r; ++l) { q += akl[k][l] * stencil[i - upwind * (r - 1 - k - l)]; }
        face_value += omega[k] * q;
    }
    return face_value;
}

static void recon_face_values(int r, const Real* stencil, Real& lo, Real& hi);

// get_slope of each class (:85,127,167,220,251,287; base class :17 returns 0)
static Real recon_slope(int r, const Real* stencil) {
    switch (r) {
    case CMFP_RECON_MINMOD: {
        Real a = stencil[1] - stencil[0];
        Real b = stencil[2] - stencil[1];
        Real g = 0.0;
        if ((a >= 0.0) && (b >= 0.0)) {
            g = std::min(a, b);
        } else if ((a <= 0.0) && (b <= 0.0)) {
            g = std::max(a, b);
        }
        return g;
    }
    case CMFP_RECON_VANLEER: {
        Real a = stencil[1] - stencil[0];
        Real b = stencil[2] - stencil[1];
        Real sa = sgn(a);
        Real sb = sgn(b);
        Real aa = std::abs(a);
        Real bb = std::abs(b);
        Real g = (sa + sb) * ((aa * bb) / (aa + bb + 1e-20));
        return g;
    }
    case CMFP_RECON_MC: {
        Real x1 = stencil[0];
        Real x2 = stencil[1];
        Real x3 = stencil[2];
        Real dqm = (x2 - x1);
        Real dqp = (x3 - x2);
        Real dqc = 0.5 * (x3 - x1);
        Real s = dqm * dqp;
        if (s <= 0.0) {
            s = 0.0;
        } else {
            if ((std::abs(dqm) < std::abs(dqp)) && (std::abs(dqm) < std::abs(dqc))) {
                s = dqm;
            } else if (std::abs(dqp) < std::abs(dqc)) {
                s = dqp;
            } else {
                s = dqc;
            }
        }
        return s;
    }
    case CMFP_RECON_CENTRE: return 0.5 * (stencil[2] - stencil[0]);
    case CMFP_RECON_O6:
    case CMFP_RECON_WENO: {
        Real lo, hi;
        recon_face_values(r, stencil, lo, hi);
        return hi - lo;
    }
    default: return 0;  // Reconstruction::get_slope, ConstantReconstruction inherits it
    }
}

// get_face_values of each class (:63,103,143,196,227,261,359)
static void recon_face_values(int r, const Real* stencil, Real& lo, Real& hi) {
    switch (r) {
    case CMFP_RECON_CONSTANT:
        lo = stencil[0];
        hi = stencil[0];
        return;
    case CMFP_RECON_MINMOD:
    case CMFP_RECON_VANLEER:
    case CMFP_RECON_MC:
    case CMFP_RECON_CENTRE: {
        Real g = recon_slope(r, stencil);
        lo = stencil[1] - 0.5 * g;
        hi = stencil[1] + 0.5 * g;
        return;
    }
    case CMFP_RECON_O6:
        hi = (37.0 / 60.0) * (stencil[3] + stencil[4]) - (2.0 / 15.0) * (stencil[2] + stencil[5]) +
             (1.0 / 60.0) * (stencil[1] + stencil[6]);
        lo = (37.0 / 60.0) * (stencil[3] + stencil[2]) - (2.0 / 15.0) * (stencil[4] + stencil[1]) +
             (1.0 / 60.0) * (stencil[5] + stencil[0]);
        return;
    case CMFP_RECON_WENO:
        hi = wenosymbo(stencil, 1);
        lo = wenosymbo(stencil, -1);
        return;
    }
}

// ---------------------------------------------------------------------------
// Gas model — hydro/gas/MFP_thermally_perfect_gas.cpp, MFP_hydro_gas.cpp.  A gas of nt + 1 species carries
// nt mass fractions alpha; where every species has the same mass / gamma the mass_const / gamma_const
// short-circuits (:55,91,127,155, MFP_hydro_gas.cpp:54,78) return the constants of species 0.
static inline Real clamp01(Real a) { return (a < 0.0) ? 0.0 : ((1.0 < a) ? 1.0 : a); }  // std::clamp(a, 0.0, 1.0)
struct Gas {
    Real mass, charge, gamma;  // species 0
    int nt = 0;                // n_tracers = n_species - 1
    bool mass_const = true, gamma_const = true, charge_const = true;
    Real sm[CMFP_MAX_TRACERS + 1] = {0}, sg[CMFP_MAX_TRACERS + 1] = {0}, sq[CMFP_MAX_TRACERS + 1] = {0};  // per species
    Real cp() const { return gamma / (mass * (gamma - 1)); }  // :127
    // get_gamma_from_prim :51-83 (alpha: the nt mass fractions)
    Real gamma_from_alpha(const Real* alpha) const {
        if (gamma_const) return gamma;
        Real S_alphaicpi = 0.0, S_alphaicvi = 0.0, S_alphai = 0.0, cvi, cpi, alphai;
        for (int i = 0; i < nt; ++i) {
            alphai = clamp01(alpha[i]);
            S_alphai += alphai;
            cpi = sg[i] / (sm[i] * (sg[i] - 1.0));
            cvi = 1.0 / (sm[i] * (sg[i] - 1.0));
            S_alphaicpi += alphai * cpi;
            S_alphaicvi += alphai * cvi;
        }
        cpi = sg[nt] / (sm[nt] * (sg[nt] - 1.0));
        cvi = 1.0 / (sm[nt] * (sg[nt] - 1.0));
        S_alphaicpi += (1.0 - S_alphai) * cpi;
        S_alphaicvi += (1.0 - S_alphai) * cvi;
        return S_alphaicpi / S_alphaicvi;
    }
    // get_cp_from_prim :123-147
    Real cp_from_alpha(const Real* alpha) const {
        if (gamma_const && mass_const) return cp();
        Real S_alphai = 0.0, S_alphaicpi = 0.0, cpi, alphai;
        for (int i = 0; i < nt; ++i) {
            alphai = clamp01(alpha[i]);
            S_alphai += alphai;
            cpi = sg[i] / (sm[i] * (sg[i] - 1.0));
            S_alphaicpi += alphai * cpi;
        }
        cpi = sg[nt] / (sm[nt] * (sg[nt] - 1.0));
        S_alphaicpi += (1.0 - S_alphai) * cpi;
        return S_alphaicpi;
    }
    // HydroGas::get_mass_from_prim MFP_hydro_gas.cpp:50-70
    Real mass_from_alpha(const Real* alpha) const {
        if (mass_const) return mass;
        Real S_alphai_mi = 0.0, S_alphas = 0.0, alphai;
        for (int i = 0; i < nt; ++i) {
            alphai = clamp01(alpha[i]);
            S_alphas += alphai;
            S_alphai_mi += alphai / sm[i];
        }
        S_alphai_mi += (1.0 - S_alphas) / sm[nt];
        return 1.0 / S_alphai_mi;
    }
    // HydroGas::get_charge_from_prim MFP_hydro_gas.cpp:98-121
    Real charge_from_alpha(const Real* alpha) const {
        if (charge_const) return charge;
        Real S_alphaiqi_mi = 0, S_alphai_mi = 0, S_alphas = 0, alphai;
        for (int i = 0; i < nt; ++i) {
            alphai = clamp01(alpha[i]);
            S_alphaiqi_mi += alphai * sq[i] / sm[i];
            S_alphai_mi += alphai / sm[i];
            S_alphas += alphai;
        }
        S_alphaiqi_mi += (1.0 - S_alphas) * sq[nt] / sm[nt];
        S_alphai_mi += (1.0 - S_alphas) / sm[nt];
        return S_alphaiqi_mi / S_alphai_mi;
    }
    // get_mass_from_cons / get_charge_from_cons of the species' conserved values at (i, j, k)
    template <class F> void mass_charge_from_cons(const F& U4, int i, int j, int k, Real& m, Real& q) const {
        Real alpha[CMFP_MAX_TRACERS + 1];
        for (int n = 0; n < nt; ++n) alpha[n] = U4(i, j, k, 5 + n) / U4(i, j, k, 0);
        m = mass_from_alpha(alpha);
        q = charge_from_alpha(alpha);
    }
    // the *_from_cons forms (:85-121,149-177, MFP_hydro_gas.cpp:72-96): alpha_i = U[tracer_i] / rho
    void alpha_from_cons(const Real* U, Real* alpha) const {
        for (int i = 0; i < nt; ++i) alpha[i] = U[5 + i] / U[0];
    }
};

// HydroGas::prim_valid  MFP_hydro_gas.cpp:152-164
static bool prim_valid(const Real* Q) {
    if ((Q[PDensity] <= 0.0) || (Q[Prs] <= 0.0) || (Q[Temp] <= 0.0)) {
        flag_abort();
        return false;
    }
    return true;
}

// ThermallyPerfectGas::cons2prim :180-218
static bool gas_cons2prim(const Gas& G, const Real* U, Real* Q) {
    Real rho = U[Density];
    Real mx = U[Xmom];
    Real my = U[Ymom];
    Real mz = U[Zmom];
    Real ed = U[Eden];

    Real rhoinv = 1 / rho;
    Real u = mx * rhoinv;
    Real v = my * rhoinv;
    Real w = mz * rhoinv;
    Real ke = 0.5 * rho * (u * u + v * v + w * w);
    Real alpha[CMFP_MAX_TRACERS + 1];
    G.alpha_from_cons(U, alpha);
    Real m = G.mass_from_alpha(alpha);    // get_mass_from_cons
    Real g = G.gamma_from_alpha(alpha);   // get_gamma_from_cons
    Real cp = G.cp_from_alpha(alpha);     // get_cp_from_cons
    Real p = (ed - ke) * (g - 1);
    Real T = p * rhoinv * m;

    Q[PDensity] = rho;
    Q[Xvel] = u;
    Q[Yvel] = v;
    Q[Zvel] = w;
    Q[Prs] = p;
    Q[Temp] = T;
    Q[Gamma] = g;
    Q[SpHeat] = cp;
    for (int i = 0; i < G.nt; ++i) { Q[HP_NUM + i] = U[HC_NUM + i] * rhoinv; }  // :212-214
    return prim_valid(Q);
}

// ThermallyPerfectGas::prim2cons :220-246
static void gas_prim2cons(const Gas& G, const Real* Q, Real* U) {
    Real rho = Q[PDensity];
    Real u = Q[Xvel];
    Real v = Q[Yvel];
    Real w = Q[Zvel];
    Real p = Q[Prs];

    Real mx = u * rho;
    Real my = v * rho;
    Real mz = w * rho;
    Real ke = 0.5 * rho * (u * u + v * v + w * w);
    Real g = G.gamma_from_alpha(Q + HP_NUM);  // get_gamma_from_prim(Q)
    Real ed = p / (g - 1) + ke;

    U[Density] = rho;
    U[Xmom] = mx;
    U[Ymom] = my;
    U[Zmom] = mz;
    U[Eden] = ed;
    for (int i = 0; i < G.nt; ++i) { U[HC_NUM + i] = Q[HP_NUM + i] * rho; }  // :243-245
}

// ThermallyPerfectGas::get_speed_from_cons :294-317 (all three components;
// callers use the first AMREX_SPACEDIM)
static void gas_speed_from_cons(const Gas& G, const Real* U, Real* s) {
    Real rho = U[Density];
    Real mx = U[Xmom];
    Real my = U[Ymom];
    Real mz = U[Zmom];
    Real ed = U[Eden];

    Real rhoinv = 1 / rho;
    Real u = mx * rhoinv;
    Real v = my * rhoinv;
    Real w = mz * rhoinv;
    Real ke = 0.5 * rho * (u * u + v * v + w * w);
    Real alpha[CMFP_MAX_TRACERS + 1];
    G.alpha_from_cons(U, alpha);
    Real g = G.gamma_from_alpha(alpha);  // get_gamma_from_cons
    Real p = (ed - ke) * (g - 1);
    Real a = std::sqrt(g * p * rhoinv);
    s[0] = a + std::abs(u);
    s[1] = a + std::abs(v);
    s[2] = a + std::abs(w);
}

// ThermallyPerfectGas::get_speed_from_prim :319-332
static void gas_speed_from_prim(const Gas& G, const Real* Q, Real* s) {
    Real g = G.gamma_from_alpha(Q + HP_NUM);  // get_gamma_from_prim(Q)
    Real a = std::sqrt(g * Q[Prs] / Q[PDensity]);
    s[0] = a + std::abs(Q[Xvel]);
    s[1] = a + std::abs(Q[Yvel]);
    s[2] = a + std::abs(Q[Zvel]);
}

// ---------------------------------------------------------------------------
// Riemann solvers
// HydroHLLC::solve  hydro/riemann/MFP_hllc.cpp:30-185; nt = n_alpha tracers behind the 8 primitives
static void hllc_solve(const Real* L, const Real* R, Real* F, int nt = 0) {
    Real rhoL = L[PDensity];
    Real uL = L[Xvel];
    Real vL = L[Yvel];
    Real wL = L[Zvel];
    Real pL = L[Prs];
    Real gamL = L[Gamma];
    Real nrgL = pL / (gamL - 1) + 0.5 * rhoL * (uL * uL + vL * vL + wL * wL);
    Real aL = sqrt(gamL * pL / rhoL);
    Real trL[CMFP_MAX_TRACERS], trR[CMFP_MAX_TRACERS];
    for (int i = 0; i < nt; ++i) trL[i] = L[HP_NUM + i] * rhoL;  // :46-48

    Real rhoR = R[PDensity];
    Real uR = R[Xvel];
    Real vR = R[Yvel];
    Real wR = R[Zvel];
    Real pR = R[Prs];
    Real gamR = R[Gamma];
    Real nrgR = pR / (gamR - 1) + 0.5 * rhoR * (uR * uR + vR * vR + wR * wR);
    Real aR = sqrt(gamR * pR / rhoR);
    for (int i = 0; i < nt; ++i) trR[i] = R[HP_NUM + i] * rhoR;  // :62-64

    Real rho_bar = 0.5 * (rhoL + rhoR);
    Real a_bar = 0.5 * (aL + aR);

    Real p_star = 0.5 * (pL + pR) - 0.5 * (uR - uL) * rho_bar * a_bar;
    Real S_star = 0.5 * (uL + uR) - 0.5 * (pR - pL) / (rho_bar * a_bar);

    Real qL;
    if (p_star <= pL) {
        qL = 1.0;
    } else {
        qL = std::sqrt(1.0 + ((gamL + 1.0) / (2 * gamL)) * (p_star / pL - 1.0));
    }
    Real S_L = uL - aL * qL;

    Real qR;
    if (p_star <= pR) {
        qR = 1.0;
    } else {
        qR = std::sqrt(1.0 + ((gamR + 1.0) / (2 * gamR)) * (p_star / pR - 1.0));
    }
    Real S_R = uR + aR * qR;

    Real fv[5], sv[5], svs[5];
    if (S_L >= 0.0) {
        F[Density] = rhoL * uL;
        F[Xmom] = rhoL * uL * uL + pL;
        F[Ymom] = rhoL * uL * vL;
        F[Zmom] = rhoL * uL * wL;
        F[Eden] = uL * (nrgL + pL);
        for (int i = 0; i < nt; ++i) { F[HC_NUM + i] = uL * trL[i]; }  // :100
        return;
    } else if ((S_L <= 0.0) && (0.0 <= S_star)) {
        fv[Density] = rhoL * uL;
        fv[Xmom] = rhoL * uL * uL + pL;
        fv[Ymom] = rhoL * uL * vL;
        fv[Zmom] = rhoL * uL * wL;
        fv[Eden] = uL * (nrgL + pL);

        sv[Density] = rhoL;
        sv[Xmom] = rhoL * uL;
        sv[Ymom] = rhoL * vL;
        sv[Zmom] = rhoL * wL;
        sv[Eden] = nrgL;

        Real coeff = rhoL * ((S_L - uL) / (S_L - S_star));

        svs[Density] = coeff;
        svs[Xmom] = coeff * S_star;
        svs[Ymom] = coeff * vL;
        svs[Zmom] = coeff * wL;
        svs[Eden] = coeff * (nrgL / rhoL + (S_star - uL) * (S_star + pL / (rhoL * (S_L - uL))));

        for (int i = 0; i < 5; ++i) { F[i] = fv[i] + S_L * (svs[i] - sv[i]); }
        for (int i = 0; i < nt; ++i) {  // :111,120,131-137
            Real fvt = uL * trL[i], svt = trL[i], svst = trL[i] * ((S_L - uL) / (S_L - S_star));
            F[HC_NUM + i] = fvt + S_L * (svst - svt);
        }
    } else if ((S_star <= 0.0) && (0.0 <= S_R)) {
        fv[Density] = rhoR * uR;
        fv[Xmom] = rhoR * uR * uR + pR;
        fv[Ymom] = rhoR * uR * vR;
        fv[Zmom] = rhoR * uR * wR;
        fv[Eden] = uR * (nrgR + pR);

        sv[Density] = rhoR;
        sv[Xmom] = rhoR * uR;
        sv[Ymom] = rhoR * vR;
        sv[Zmom] = rhoR * wR;
        sv[Eden] = nrgR;

        Real coeff = rhoR * ((S_R - uR) / (S_R - S_star));

        svs[Density] = coeff;
        svs[Xmom] = coeff * S_star;
        svs[Ymom] = coeff * vR;
        svs[Zmom] = coeff * wR;
        svs[Eden] = coeff * (nrgR / rhoR + (S_star - uR) * (S_star + pR / (rhoR * (S_R - uR))));

        for (int i = 0; i < 5; ++i) { F[i] = fv[i] + S_R * (svs[i] - sv[i]); }
        for (int i = 0; i < nt; ++i) {  // :147,156,167-173
            Real fvt = uR * trR[i], svt = trR[i], svst = trR[i] * ((S_R - uR) / (S_R - S_star));
            F[HC_NUM + i] = fvt + S_R * (svst - svt);
        }
    } else {
        F[Density] = rhoR * uR;
        F[Xmom] = rhoR * uR * uR + pR;
        F[Ymom] = rhoR * uR * vR;
        F[Zmom] = rhoR * uR * wR;
        F[Eden] = uR * (nrgR + pR);
        for (int i = 0; i < nt; ++i) { F[HC_NUM + i] = trR[i] * uR; }  // :183
    }
}

// HydroHLLE::solve  hydro/riemann/MFP_hydro_hlle.cpp:30-120; nt = n_alpha tracers
static void hlle_solve(const Real* L, const Real* R, Real* F, int nt = 0) {
    Real rhoL = L[PDensity];
    Real uL = L[Xvel];
    Real vL = L[Yvel];
    Real wL = L[Zvel];
    Real pL = L[Prs];
    Real gamL = L[Gamma];
    Real nrgL = pL / (gamL - 1) + 0.5 * rhoL * (uL * uL + vL * vL + wL * wL);
    Real aL = sqrt(gamL * pL / rhoL);
    Real trL[CMFP_MAX_TRACERS], trR[CMFP_MAX_TRACERS];
    for (int i = 0; i < nt; ++i) trL[i] = L[HP_NUM + i] * rhoL;  // :46-48

    Real rhoR = R[PDensity];
    Real uR = R[Xvel];
    Real vR = R[Yvel];
    Real wR = R[Zvel];
    Real pR = R[Prs];
    Real gamR = R[Gamma];
    Real nrgR = pR / (gamR - 1) + 0.5 * rhoR * (uR * uR + vR * vR + wR * wR);
    Real aR = sqrt(gamR * pR / rhoR);
    for (int i = 0; i < nt; ++i) trR[i] = R[HP_NUM + i] * rhoR;  // :60-62

    Real sL = std::min(uL - aL, uR - aR);
    Real sR = std::max(uL + aL, uR + aR);

    if (sL >= 0.0) {
        F[Density] = rhoL * uL;
        F[Xmom] = rhoL * uL * uL + pL;
        F[Ymom] = rhoL * uL * vL;
        F[Zmom] = rhoL * uL * wL;
        F[Eden] = uL * (nrgL + pL);
        for (int i = 0; i < nt; ++i) { F[HC_NUM + i] = uL * trL[i]; }  // :75
    } else if ((sL <= 0.0) && (sR >= 0.0)) {
        Real fvL[5], fvR[5], svL[5], svR[5];
        fvL[Density] = rhoL * uL;
        fvL[Xmom] = rhoL * uL * uL + pL;
        fvL[Ymom] = rhoL * uL * vL;
        fvL[Zmom] = rhoL * uL * wL;
        fvL[Eden] = uL * (nrgL + pL);

        fvR[Density] = rhoR * uR;
        fvR[Xmom] = rhoR * uR * uR + pR;
        fvR[Ymom] = rhoR * uR * vR;
        fvR[Zmom] = rhoR * uR * wR;
        fvR[Eden] = uR * (nrgR + pR);

        svL[Density] = rhoL;
        svL[Xmom] = rhoL * uL;
        svL[Ymom] = rhoL * vL;
        svL[Zmom] = rhoL * wL;
        svL[Eden] = nrgL;

        svR[Density] = rhoR;
        svR[Xmom] = rhoR * uR;
        svR[Ymom] = rhoR * vR;
        svR[Zmom] = rhoR * wR;
        svR[Eden] = nrgR;

        for (int i = 0; i < 5; ++i) {
            F[i] = (sR * fvL[i] - sL * fvR[i] + sL * sR * (svR[i] - svL[i])) / (sR - sL);
        }
        for (int i = 0; i < nt; ++i) {  // :83,91,99,107,109-111
            Real fl = uL * trL[i], fr = uR * trR[i];
            F[HC_NUM + i] = (sR * fl - sL * fr + sL * sR * (trR[i] - trL[i])) / (sR - sL);
        }
    } else {
        F[Density] = rhoR * uR;
        F[Xmom] = rhoR * uR * uR + pR;
        F[Ymom] = rhoR * uR * vR;
        F[Zmom] = rhoR * uR * wR;
        F[Eden] = uR * (nrgR + pR);
        for (int i = 0; i < nt; ++i) { F[HC_NUM + i] = uR * trR[i]; }  // :118
    }
}

// PressureJumpShockDetector::solve  MFP_hydro_shockdetector.cpp:35-44
static Real shock_detector(const Real* L, const Real* R, Real shock_threshold) {
    Real pL = L[Prs];
    Real pR = R[Prs];
    Real varphi = std::abs(pR - pL) / (pR + pL);
    return 0.5 + 0.5 * tanh_approx(5 * (varphi - 0.75 * shock_threshold) / shock_threshold);
}

// HydroHybridHLL::solve  MFP_hybrid_hll.cpp:30-47
static void hybrid_solve(const Real* L, const Real* R, Real* F, Real shk) {
    const Real eps = 1e-14;
    if (shk < eps) {
        hllc_solve(L, R, F);
    } else if ((1.0 - shk) < eps) {
        hlle_solve(L, R, F);
    } else {
        Real F_hllc[5], F_hlle[5];
        hllc_solve(L, R, F_hllc);
        hlle_solve(L, R, F_hlle);
        for (int i = 0; i < 5; ++i) { F[i] = (1.0 - shk) * F_hllc[i] + shk * F_hlle[i]; }
    }
}

// FieldRH::solve  field/riemann/MFP_field_RH.cpp:29-106
static void field_rh_solve(Real c0, Real ch, const Real* L, const Real* R, Real* F) {
    const Real ch2 = ch * ch;  // :26
    for (int n = 0; n < FC_NUM; ++n) F[n] = 0;

    Real Dr, Dl, Br, Bl, Pl, Pr, fl, fr, epr, epl, mur, mul, cr, cl;

    Dl = L[Dy];
    Dr = R[Dy];
    epr = R[ep];
    epl = L[ep];
    Bl = L[Bz];
    Br = R[Bz];
    mur = R[mu];
    mul = L[mu];

    cr = 1 / std::sqrt(mur * epr);
    cl = 1 / std::sqrt(mul * epl);

    F[Dy] = c0 * ((Bl * cl + Br * cr) + (Dl / epl - Dr / epr)) / (cl * mul + cr * mur);
    F[Bz] = c0 * ((Dl * cl + Dr * cr) + (Bl / mul - Br / mur)) / (cl * epl + cr * epr);

    if (ch > 0) {
        Pl = L[phi];
        Pr = R[phi];
        fl = L[Dx];
        fr = R[Dx];
        F[Dx] = 0.5 * c0 * (Pr + Pl) - 0.5 * ch * (fr - fl);
        F[phi] = 0.5 * ch2 / c0 * (fr + fl) - 0.5 * ch * (Pr - Pl);
    }

    Dl = L[Dz];
    Dr = R[Dz];
    epr = R[ep];
    epl = L[ep];
    Bl = L[By];
    Br = R[By];

    F[Dz] = c0 * (-(Bl * cl + Br * cr) + (Dl / epl - Dr / epr)) / (cl * mul + cr * mur);
    F[By] = c0 * (-(Dl * cl + Dr * cr) + (Bl / mul - Br / mur)) / (cl * epl + cr * epr);

    if (ch > 0) {
        Pl = L[psi];
        Pr = R[psi];
        fl = L[Bx];
        fr = R[Bx];
        F[Bx] = 0.5 * c0 * (Pr + Pl) - 0.5 * ch * (fr - fl);
        F[psi] = 0.5 * ch2 / c0 * (fr + fl) - 0.5 * ch * (Pr - Pl);
    }
}

// transform_global2local / local2global  modules/MFP_transforms.H:11-55
static void global2local(Real* U, int idir, const int* vi, int nv) {
    if (idir == 0) return;
    int in = (idir + 0) % 3, it1 = (idir + 1) % 3, it2 = (idir + 2) % 3;
    for (int a = 0; a < nv; ++a) {
        int v = vi[a];
        Real n = U[v + in], t1 = U[v + it1], t2 = U[v + it2];
        U[v + 0] = n;
        U[v + 1] = t1;
        U[v + 2] = t2;
    }
}
static void local2global(Real* U, int idir, const int* vi, int nv) {
    if (idir == 0) return;
    int in = (idir + 0) % 3, it1 = (idir + 1) % 3, it2 = (idir + 2) % 3;
    for (int a = 0; a < nv; ++a) {
        int v = vi[a];
        Real n = U[v + 0], t1 = U[v + 1], t2 = U[v + 2];
        U[v + in] = n;
        U[v + it1] = t1;
        U[v + it2] = t2;
    }
}

// ---------------------------------------------------------------------------
// Box / Fab: the minimum of amrex::Box / FArrayBox (Fortran order, component
// slowest) the FAB-level functions need.
struct Box {
    int lo[3], hi[3];
    int len(int d) const { return hi[d] - lo[d] + 1; }
    long npts() const { return (long)len(0) * len(1) * len(2); }
};
static Box grow(Box b, int n, int DIM) {
    for (int d = 0; d < DIM; ++d) {
        b.lo[d] -= n;
        b.hi[d] += n;
    }
    return b;
}
struct Fab {
    Box b;
    int nc = 0;
    std::vector<Real> d;
    void resize(const Box& bx, int n) {
        b = bx;
        nc = n;
        d.resize((size_t)bx.npts() * n);
    }
    void setVal(Real v) { std::fill(d.begin(), d.end(), v); }
    inline Real& operator()(int i, int j, int k, int n) {
        return d[(((size_t)n * b.len(2) + (k - b.lo[2])) * b.len(1) + (j - b.lo[1])) * b.len(0) + (i - b.lo[0])];
    }
    inline const Real& operator()(int i, int j, int k, int n) const {
        return d[(((size_t)n * b.len(2) + (k - b.lo[2])) * b.len(1) + (j - b.lo[1])) * b.len(0) + (i - b.lo[0])];
    }
};

// ---------------------------------------------------------------------------
struct StateDef {
    cmfp_state_desc desc;
    Gas gas;
    int n_cons, n_prim, num_grow;
    Real div_speed, fastest_speed;  // field  (MFP_field.cpp:382-388)
    int vec_prim[2], vec_cons[2], n_vec;
    bool hydro() const { return desc.kind == CMFP_STATE_HYDRO; }
    // Dirichlet data of the domain faces: the reference's per-face Lua values / functions of
    // (x, y, z, t) (BoundaryState, modules/boundary/MFP_fillbc.cpp:8-88), tabulated by the caller.
    // [dim][side][primitive]: values at the centres of the ghost cells beyond the face (box dbox:
    // tng layers along dim, the other dimensions grown by tng) and at the face centres (box fbox).
    struct FaceTab {
        std::vector<Real> cell, face;
    };
    FaceTab tab[3][2][CMFP_MAX_COMP];
    int tng = 0;
    // 'dynamic' components (EulerianState::dynamic_functions, MFP_eulerian.H:54; read for field states only,
    // MFP_field.cpp:175-221): the function values over grow(domain, dyn_ng), i fastest; empty: none
    std::vector<Real> dyn[CMFP_MAX_COMP];
    int dyn_ng = 0;
    bool inflow(int d, int side) const {  // where_is_inflow == ext_dir  (:72-87)
        for (int n = 0; n < CMFP_MAX_COMP; ++n)
            if (!tab[d][side][n].cell.empty()) return true;
        return false;
    }
};
// boxes of the Dirichlet tables of face (d, side); linear index with i fastest
static Box tab_box(const Box& domain, int DIM, int tng, int d, int side, bool face) {
    Box b = grow(domain, tng, DIM);
    if (face) {
        b.lo[d] = b.hi[d] = side == 0 ? domain.lo[d] : domain.hi[d] + 1;
    } else if (side == 0) {
        b.hi[d] = domain.lo[d] - 1;
    } else {
        b.lo[d] = domain.hi[d] + 1;
    }
    return b;
}
static inline long tab_at(const Box& b, int i, int j, int k) {
    return ((long)(k - b.lo[2]) * b.len(1) + (j - b.lo[1])) * b.len(0) + (i - b.lo[0]);
}

// HydroViscous family, hydro/viscous/MFP_hydro_viscous.cpp: get_coeffs (:145-160 Sutherland,
// :207-220 PowerLaw, :266-279 UserDefined) and calc_stability_condition (:162-176, :222-236, :281-292)
static void visc_coeffs(const StateDef& S, const Real* Q, Real& T, Real& mu, Real& kappa) {
    const cmfp_state_desc& d = S.desc;
    T = Q[Temp];
    Real cp = S.gas.cp_from_alpha(Q + HP_NUM);  // get_cp_from_prim(Q)
    if (d.viscosity == CMFP_VISC_SUTHERLAND) {
        Real T_ = T / d.visc_T0;
        mu = d.visc_mu0 * T_ * sqrt(T_) * (d.visc_T0 + d.visc_S) / (T + d.visc_S);
    } else if (d.viscosity == CMFP_VISC_POWERLAW) {
        mu = d.visc_mu0 * pow(T / d.visc_T0, d.visc_n);
    } else {
        mu = d.visc_mu0;
    }
    kappa = mu * cp / d.visc_Pr;
}
static Real visc_stability(const StateDef& S, const Real* U) {
    const cmfp_state_desc& d = S.desc;
    Real Q[HP_NUM];
    // get_temperature_from_cons (MFP_thermally_perfect_gas.cpp:271-292): the arithmetic of cons2prim
    Real rho = U[Density];
    Real rhoinv = 1 / rho;
    Real u = U[Xmom] * rhoinv, v = U[Ymom] * rhoinv, w = U[Zmom] * rhoinv;
    Real ke = 0.5 * rho * (u * u + v * v + w * w);
    Real alpha[CMFP_MAX_TRACERS + 1];
    S.gas.alpha_from_cons(U, alpha);
    const Real gam = S.gas.gamma_from_alpha(alpha);  // get_gamma_from_cons
    Real p = (U[Eden] - ke) * (gam - 1);
    Real T = p * rhoinv * S.gas.mass_from_alpha(alpha);
    (void)Q;
    Real mu;
    if (d.viscosity == CMFP_VISC_SUTHERLAND) {
        const Real T_ = T / d.visc_T0;
        mu = d.visc_mu0 * T_ * sqrt(T_) * (d.visc_T0 + d.visc_S) / (T + d.visc_S);
    } else if (d.viscosity == CMFP_VISC_POWERLAW) {
        mu = d.visc_mu0 * pow(T / d.visc_T0, d.visc_n);
    } else {
        mu = d.visc_mu0;
    }
    return (4 * mu * gam / (d.visc_Pr * rho));
}

// HydroState::calc_viscous_fluxes MFP_hydro.cpp:1085-1329 (regular cells; 1-D and 2-D: the 3-D
// branch of the reference names undeclared variables).  `two_thirds = 2 / 3` is integer division in
// the reference (:1123), i.e. 0: the divergence term is absent.  Reproduced.
static void calc_viscous_fluxes(const StateDef& S, int DIM, const Box& box, Fab* fluxes, const Fab& prim, const Real* dx) {
    Real dxinv[3];
    for (int d = 0; d < DIM; ++d) dxinv[d] = 1 / dx[d];
    const Real two_thirds = 2 / 3;
    auto coef = [&](int i, int j, int k, Real& T, Real& mu, Real& kappa) {
        Real Q[CMFP_MAX_COMP];
        for (int n = 0; n < S.n_prim; ++n) Q[n] = prim(i, j, k, n);
        visc_coeffs(S, Q, T, mu, kappa);
    };
    Real dudx = 0, dudy = 0, dudz = 0, dvdx = 0, dvdy = 0, dwdx = 0, dwdz = 0, divu = 0;
    Real tauxx, tauxy, tauxz, dTdx, muf;
    Fab& fluxX = fluxes[0];
    for (int k = box.lo[2]; k <= box.hi[2]; ++k)
        for (int j = box.lo[1]; j <= box.hi[1]; ++j)
            for (int i = box.lo[0]; i <= box.hi[0] + 1; ++i) {
                Real T1, mu1, ka1, T0, mu0, ka0;
                coef(i, j, k, T1, mu1, ka1);
                coef(i - 1, j, k, T0, mu0, ka0);
                dTdx = (T1 - T0) * dxinv[0];
                dudx = (prim(i, j, k, Xvel) - prim(i - 1, j, k, Xvel)) * dxinv[0];
                dvdx = (prim(i, j, k, Yvel) - prim(i - 1, j, k, Yvel)) * dxinv[0];
                dwdx = (prim(i, j, k, Zvel) - prim(i - 1, j, k, Zvel)) * dxinv[0];
                if (DIM >= 2) {
                    dudy = (prim(i, j + 1, k, Xvel) + prim(i - 1, j + 1, k, Xvel) - prim(i, j - 1, k, Xvel) -
                            prim(i - 1, j - 1, k, Xvel)) *
                           (0.25 * dxinv[1]);
                    dvdy = (prim(i, j + 1, k, Yvel) + prim(i - 1, j + 1, k, Yvel) - prim(i, j - 1, k, Yvel) -
                            prim(i - 1, j - 1, k, Yvel)) *
                           (0.25 * dxinv[1]);
                }
                divu = dudx + dvdy + dwdz;
                muf = 0.5 * (mu1 + mu0);
                tauxx = muf * (2 * dudx - two_thirds * divu);
                tauxy = muf * (dudy + dvdx);
                tauxz = muf * (dudz + dwdx);
                fluxX(i, j, k, Xmom) -= tauxx;
                fluxX(i, j, k, Ymom) -= tauxy;
                fluxX(i, j, k, Zmom) -= tauxz;
                fluxX(i, j, k, Eden) -= 0.5 * ((prim(i, j, k, Xvel) + prim(i - 1, j, k, Xvel)) * tauxx +
                                               (prim(i, j, k, Yvel) + prim(i - 1, j, k, Yvel)) * tauxy +
                                               (prim(i, j, k, Zvel) + prim(i - 1, j, k, Zvel)) * tauxz +
                                               (ka1 + ka0) * dTdx);
            }
    if (DIM >= 2) {
        Real tauyy, tauyz, dTdy;
        Real dvdz = 0, dwdy = 0;
        Fab& fluxY = fluxes[1];
        for (int k = box.lo[2]; k <= box.hi[2]; ++k)
            for (int j = box.lo[1]; j <= box.hi[1] + 1; ++j)
                for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
                    Real T1, mu1, ka1, T0, mu0, ka0;
                    coef(i, j, k, T1, mu1, ka1);
                    coef(i, j - 1, k, T0, mu0, ka0);
                    dTdy = (T1 - T0) * dxinv[1];
                    dudy = (prim(i, j, k, Xvel) - prim(i, j - 1, k, Xvel)) * dxinv[1];
                    dvdy = (prim(i, j, k, Yvel) - prim(i, j - 1, k, Yvel)) * dxinv[1];
                    dwdy = (prim(i, j, k, Zvel) - prim(i, j - 1, k, Zvel)) * dxinv[1];
                    dudx = (prim(i + 1, j, k, Xvel) + prim(i + 1, j - 1, k, Xvel) - prim(i - 1, j, k, Xvel) -
                            prim(i - 1, j - 1, k, Xvel)) *
                           (0.25 * dxinv[0]);
                    dvdx = (prim(i + 1, j, k, Yvel) + prim(i + 1, j - 1, k, Yvel) - prim(i - 1, j, k, Yvel) -
                            prim(i - 1, j - 1, k, Yvel)) *
                           (0.25 * dxinv[0]);
                    divu = dudx + dvdy + dwdz;
                    muf = 0.5 * (mu1 + mu0);
                    tauyy = muf * (2 * dvdy - two_thirds * divu);
                    tauxy = muf * (dudy + dvdx);
                    tauyz = muf * (dwdy + dvdz);
                    fluxY(i, j, k, Xmom) -= tauxy;
                    fluxY(i, j, k, Ymom) -= tauyy;
                    fluxY(i, j, k, Zmom) -= tauyz;
                    fluxY(i, j, k, Eden) -= 0.5 * ((prim(i, j, k, Xvel) + prim(i, j - 1, k, Xvel)) * tauxy +
                                                   (prim(i, j, k, Yvel) + prim(i, j - 1, k, Yvel)) * tauyy +
                                                   (prim(i, j, k, Zvel) + prim(i, j - 1, k, Zvel)) * tauyz +
                                                   (ka1 + ka0) * dTdy);
                }
    }
}

struct Sim;

// per-state virtuals of EulerianState
static bool st_cons2prim(const StateDef& S, const Real* U, Real* Q) {
    if (S.hydro()) return gas_cons2prim(S.gas, U, Q);
    for (int n = 0; n < FC_NUM; ++n) Q[n] = U[n];  // FieldState::cons2prim MFP_field.cpp:460-466
    return false;
}
static void st_prim2cons(const StateDef& S, const Real* Q, Real* U) {
    if (S.hydro()) {
        gas_prim2cons(S.gas, Q, U);
        return;
    }
    for (int n = 0; n < FC_NUM; ++n) U[n] = Q[n];  // :467-472
}
static void st_speed_from_prim(const StateDef& S, const Real* Q, Real* s) {
    if (S.hydro()) {
        gas_speed_from_prim(S.gas, Q, s);
        return;
    }
    s[0] = s[1] = s[2] = S.fastest_speed;  // MFP_field.H:76-79
}
static void st_speed_from_cons(const StateDef& S, const Real* U, Real* s) {
    if (S.hydro()) {
        gas_speed_from_cons(S.gas, U, s);
        return;
    }
    s[0] = s[1] = s[2] = S.fastest_speed;  // MFP_field.H:72-75
}

// EulerianState::calc_primitives  MFP_eulerian.cpp:313-433 (no EB; the dynamic functions: apply_dynamic below)
static void calc_primitives(const StateDef& S, const Box& box, const Fab& cons, Fab& prim) {
    Real U[CMFP_MAX_COMP], Q[CMFP_MAX_COMP];
    prim.resize(box, S.n_prim);
    for (int k = box.lo[2]; k <= box.hi[2]; ++k)
        for (int j = box.lo[1]; j <= box.hi[1]; ++j)
            for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
                for (int n = 0; n < S.n_cons; ++n) U[n] = cons(i, j, k, n);
                st_cons2prim(S, U, Q);
                for (int n = 0; n < S.n_prim; ++n) prim(i, j, k, n) = Q[n];
            }
}

// The 'dynamic' part of calc_primitives (MFP_eulerian.cpp:403-423): the listed primitives take the value of their
// initialisation function at (x, y, z, t), and prim2cons writes the result back into the conserved copy the action
// works on (UpdateData::U, shared with the actions that follow in the same step).  Field states: prim = cons.
static void apply_dynamic(const StateDef& S, int DIM, const Box& domain, const Box& box, Fab& cons, Fab& prim) {
    if (S.hydro()) return;  // (HydroState::set_udf does not read 'dynamic')
    const Box tb = grow(domain, S.dyn_ng, DIM);
    for (int n = 0; n < S.n_cons; ++n) {
        if (S.dyn[n].empty()) continue;
        for (int k = box.lo[2]; k <= box.hi[2]; ++k)
            for (int j = box.lo[1]; j <= box.hi[1]; ++j)
                for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
                    const Real v = S.dyn[n][(size_t)tab_at(tb, i, j, k)];
                    prim(i, j, k, n) = v;
                    cons(i, j, k, n) = v;
                }
    }
}

// EulerianState::update_boundary_cells  MFP_eulerian.cpp:435-511 with State::get_bc_limits
// (MFP_state.cpp:17-116): in the ghost cells beyond a domain face that carries Dirichlet data, every
// primitive that has a value there takes it (values tabulated at the ghost-cell centres)
static void update_boundary_cells(const StateDef& S, int DIM, const Box& domain, const int* periodic, const Box& box, Fab& prim) {
    bool all_periodic = true;
    for (int d = 0; d < DIM; ++d) all_periodic = all_periodic && periodic[d];
    if (all_periodic) return;  // :23
    for (int d = 0; d < DIM; ++d) {
        if (periodic[d]) continue;
        for (int side = 0; side < 2; ++side) {
            Box L = box;  // ghost slab of this box beyond the face  (:45-63)
            if (side == 0) {
                if (!(box.lo[d] < domain.lo[d])) continue;
                L.hi[d] = domain.lo[d] - 1;
            } else {
                if (!(box.hi[d] > domain.hi[d])) continue;
                L.lo[d] = domain.hi[d] + 1;
            }
            if (!S.inflow(d, side)) continue;  // :470-471
            const Box tb = tab_box(domain, DIM, S.tng, d, side, false);
            for (int n = 0; n < S.n_prim; ++n) {
                const std::vector<Real>& t = S.tab[d][side][n].cell;
                if (t.empty()) continue;  // f.is_valid()  :504
                for (int k = L.lo[2]; k <= L.hi[2]; ++k)
                    for (int j = L.lo[1]; j <= L.hi[1]; ++j)
                        for (int i = L.lo[0]; i <= L.hi[0]; ++i) prim(i, j, k, n) = t[tab_at(tb, i, j, k)];
            }
        }
    }
}

// EulerianState::calc_reconstruction  MFP_eulerian.cpp:513-607 (used by the field)
static void calc_reconstruction_generic(const StateDef& S, int DIM, const Box& box, const Fab& prim,
                                        Fab* rlo, Fab* rhi) {
    const int r = S.desc.reconstruction;
    const int sl = recon_stencil_length(r);
    const int offset = sl / 2;
    Real stencil[7];
    int si[3];
    Real lo_face, hi_face;
    for (int d = 0; d < DIM; ++d) {
        rlo[d].resize(box, S.n_prim);
        rhi[d].resize(box, S.n_prim);
        for (int n = 0; n < S.n_cons; ++n) {  // :567 (n_cons, quirk Q2)
            for (int k = box.lo[2]; k <= box.hi[2]; ++k)
                for (int j = box.lo[1]; j <= box.hi[1]; ++j)
                    for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
                        si[0] = si[1] = si[2] = 0;
                        for (int s = 0; s < sl; ++s) {
                            si[d] = s - offset;
                            stencil[s] = prim(i + si[0], j + si[1], k + si[2], n);
                        }
                        recon_face_values(r, stencil, lo_face, hi_face);
                        rlo[d](i, j, k, n) = lo_face;
                        rhi[d](i, j, k, n) = hi_face;
                    }
        }
    }
}

// HydroState::calc_reconstruction  hydro/MFP_hydro.cpp:780-1038 (no EB)
static void calc_reconstruction_hydro(const StateDef& S, int DIM, const Box& box, Fab& prim,
                                      Fab* rlo, Fab* rhi) {
    const Box& pbox = prim.b;
    Fab gamma_minus_one;
    gamma_minus_one.resize(pbox, 1);
    const int r = S.desc.reconstruction;
    const int sl = recon_stencil_length(r);
    const int offset = sl / 2;
    Real stencil[7];
    int si[3];
    Real Q[CMFP_MAX_COMP], cell_slope[CMFP_MAX_COMP];
    Real rho_lo, rho_hi, abs_phi, phi_scale, coeff_eps, gam_lo, gam_hi;
    Real alphas_lo[CMFP_MAX_TRACERS + 1], alphas_hi[CMFP_MAX_TRACERS + 1];

    for (int d = 0; d < DIM; ++d) {
        rlo[d].resize(box, S.n_prim);
        rhi[d].resize(box, S.n_prim);
    }

    // change pressure to internal energy  :847-862
    for (int k = pbox.lo[2]; k <= pbox.hi[2]; ++k)
        for (int j = pbox.lo[1]; j <= pbox.hi[1]; ++j)
            for (int i = pbox.lo[0]; i <= pbox.hi[0]; ++i) {
                Real al[CMFP_MAX_TRACERS + 1];
                for (int n = 0; n < S.gas.nt; ++n) al[n] = prim(i, j, k, HP_NUM + n);
                gamma_minus_one(i, j, k, 0) = S.gas.gamma_from_alpha(al) - 1.0;  // get_gamma_from_prim(Q) :857
                prim(i, j, k, Prs) /= gamma_minus_one(i, j, k, 0);
            }

    for (int d = 0; d < DIM; ++d) {
        Fab& lo4 = rlo[d];
        Fab& hi4 = rhi[d];
        for (int k = box.lo[2]; k <= box.hi[2]; ++k)
            for (int j = box.lo[1]; j <= box.hi[1]; ++j)
                for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
                    for (int n = 0; n < S.n_prim; ++n) {
                        si[0] = si[1] = si[2] = 0;
                        for (int s = 0; s < sl; ++s) {
                            si[d] = s - offset;
                            stencil[s] = prim(i + si[0], j + si[1], k + si[2], n);
                        }
                        cell_slope[n] = recon_slope(r, stencil);
                        Q[n] = stencil[offset];
                    }

                    Real& rho = Q[PDensity];
                    Real& phi_rho = cell_slope[PDensity];
                    Real& u = Q[Xvel];
                    Real& phi_u = cell_slope[Xvel];
                    Real& v = Q[Yvel];
                    Real& phi_v = cell_slope[Yvel];
                    Real& w = Q[Zvel];
                    Real& phi_w = cell_slope[Zvel];
                    Real& eps = Q[Prs];
                    Real& phi_eps = cell_slope[Prs];

                    // correct density slope  :934 (quirk Q1)
                    if (std::abs(phi_rho) > 2 * rho) { phi_rho = 2 * sign(phi_rho, 0.0) * rho; }

                    rho_lo = rho - 0.5 * phi_rho;
                    rho_hi = rho + 0.5 * phi_rho;

                    abs_phi = phi_u * phi_u + phi_v * phi_v + phi_w * phi_w;

                    Real eps_face = eps - 0.5 * std::abs(phi_eps);

                    if (eps_face <= 0.0) {
                        phi_u = 0.0;
                        phi_v = 0.0;
                        phi_w = 0.0;
                        phi_eps = 0.0;
                    } else {
                        coeff_eps = (rho / (rho_lo * rho_hi)) * eps_face;
                        if ((0.125 * abs_phi) > coeff_eps) {
                            phi_scale = sqrt(abs_phi);
                            coeff_eps = sqrt(8 * coeff_eps);
                            phi_u = (phi_u / phi_scale) * coeff_eps;
                            phi_v = (phi_v / phi_scale) * coeff_eps;
                            phi_w = (phi_w / phi_scale) * coeff_eps;
                        }
                        abs_phi = phi_u * phi_u + phi_v * phi_v + phi_w * phi_w;
                        eps -= (rho_lo * rho_hi / rho) * 0.125 * abs_phi;
                    }

                    lo4(i, j, k, PDensity) = rho_lo;
                    hi4(i, j, k, PDensity) = rho_hi;

                    lo4(i, j, k, Xvel) = u - 0.5 * (rho_hi / rho) * phi_u;
                    hi4(i, j, k, Xvel) = u + 0.5 * (rho_lo / rho) * phi_u;

                    lo4(i, j, k, Yvel) = v - 0.5 * (rho_hi / rho) * phi_v;
                    hi4(i, j, k, Yvel) = v + 0.5 * (rho_lo / rho) * phi_v;

                    lo4(i, j, k, Zvel) = w - 0.5 * (rho_hi / rho) * phi_w;
                    hi4(i, j, k, Zvel) = w + 0.5 * (rho_lo / rho) * phi_w;

                    for (int n = 0; n < S.gas.nt; ++n) {  // :982-995
                        Real& alpha = Q[HP_NUM + n];
                        Real& phi_alpha = cell_slope[HP_NUM + n];
                        Real alpha_lo = alpha - 0.5 * phi_alpha;
                        Real alpha_hi = alpha + 0.5 * phi_alpha;
                        lo4(i, j, k, HP_NUM + n) = alpha_lo;
                        hi4(i, j, k, HP_NUM + n) = alpha_hi;
                        alphas_lo[n] = alpha_lo;
                        alphas_hi[n] = alpha_hi;
                    }

                    gam_lo = S.gas.gamma_from_alpha(alphas_lo);  // get_gamma_from_prim(alphas_lo, 0) :997
                    gam_hi = S.gas.gamma_from_alpha(alphas_hi);

                    lo4(i, j, k, Prs) = (eps - 0.5 * phi_eps) * (gam_lo - 1.0);
                    hi4(i, j, k, Prs) = (eps + 0.5 * phi_eps) * (gam_hi - 1.0);

                    lo4(i, j, k, Temp) = lo4(i, j, k, Prs) / (rho_lo / S.gas.mass_from_alpha(alphas_lo));
                    hi4(i, j, k, Temp) = hi4(i, j, k, Prs) / (rho_hi / S.gas.mass_from_alpha(alphas_hi));

                    lo4(i, j, k, Gamma) = gam_lo;
                    hi4(i, j, k, Gamma) = gam_hi;

                    lo4(i, j, k, SpHeat) = S.gas.cp_from_alpha(alphas_lo);
                    hi4(i, j, k, SpHeat) = S.gas.cp_from_alpha(alphas_hi);
                }
    }

    // convert back to pressure  :1025-1035
    for (int k = pbox.lo[2]; k <= pbox.hi[2]; ++k)
        for (int j = pbox.lo[1]; j <= pbox.hi[1]; ++j)
            for (int i = pbox.lo[0]; i <= pbox.hi[0]; ++i) { prim(i, j, k, Prs) *= gamma_minus_one(i, j, k, 0); }
}

// EulerianState::calc_time_averaged_faces  MFP_eulerian.cpp:609-676
static void calc_time_averaged_faces(const StateDef& S, int DIM, const Box& box, const Fab& prim,
                                     Fab* rlo, Fab* rhi, const Real* dx, Real dt) {
    Real Q[CMFP_MAX_COMP];
    Real lo_face, centre, hi_face, a6, dt_2dx;
    const Real ft = 4.0 / 3.0;
    Real local_c[3];
    for (int k = box.lo[2]; k <= box.hi[2]; ++k)
        for (int j = box.lo[1]; j <= box.hi[1]; ++j)
            for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
                for (int n = 0; n < S.n_prim; ++n) Q[n] = prim(i, j, k, n);
                st_speed_from_prim(S, Q, local_c);
                for (int d = 0; d < DIM; ++d) {
                    Fab& lo4 = rlo[d];
                    Fab& hi4 = rhi[d];
                    dt_2dx = 0.5 * dt / dx[d];
                    for (int n = 0; n < S.n_prim; ++n) {
                        lo_face = lo4(i, j, k, n);
                        hi_face = hi4(i, j, k, n);
                        centre = prim(i, j, k, n);
                        a6 = 6.0 * centre - 3 * (lo_face + hi_face);
                        lo4(i, j, k, n) += local_c[d] * dt_2dx * (hi_face - lo_face + (1 - local_c[d] * ft * dt_2dx) * a6);
                        hi4(i, j, k, n) -= local_c[d] * dt_2dx * (hi_face - lo_face - (1 - local_c[d] * ft * dt_2dx) * a6);
                    }
                }
            }
}

struct Geom {
    int DIM;
    Box domain;
    Real dx[3], prob_lo[3];
    int periodic[3];
};

// EulerianState::update_face_prim + face_bc  MFP_eulerian.cpp:882-955, 683-880 (Dirichlet data
// as tabulated values at the face centres).
// At the low domain face the "L" state of the face plane (r_hi of the ghost
// cell, shifted) is overwritten from the "R" state (r_lo of the first cell);
// at the high face R (r_lo of the ghost cell) from L (r_hi of the last cell).
static void update_face_prim(const StateDef& S, const Geom& G, const Box& box, Fab* r_lo, Fab* r_hi,
                             bool do_all) {
    for (int d = 0; d < G.DIM; ++d) {
        if (G.periodic[d]) continue;  // :899
        int e[3] = {0, 0, 0};
        e[d] = 1;
        for (int side = 0; side < 2; ++side) {
            Box pl = box;  // face plane (face index = cell index of the hi-side cell)
            if (side == 0) {
                if (!(box.lo[d] <= G.domain.lo[d])) continue;  // :901
                pl.lo[d] = pl.hi[d] = G.domain.lo[d];
            } else {
                if (!(box.hi[d] >= G.domain.hi[d])) continue;  // :928
                pl.lo[d] = pl.hi[d] = G.domain.hi[d] + 1;
            }
            for (int n = 0; n < S.n_prim && do_all; ++n) {
                // fill_bc has n_prim entries but only the first n_cons are set
                // (MFP_hydro.cpp:448-463); the rest stay BCType::bogus
                int bc = (n < S.n_cons) ? S.desc.bc[n][d][side] : CMFP_BC_BOGUS;
                Real sgnv;
                if (bc == CMFP_BC_FOEXTRAP || bc == CMFP_BC_HOEXTRAP || bc == CMFP_BC_REFLECT_EVEN) {
                    sgnv = 1.0;
                } else if (bc == CMFP_BC_REFLECT_ODD) {
                    sgnv = -1.0;
                } else {
                    continue;
                }
                for (int k = pl.lo[2]; k <= pl.hi[2]; ++k)
                    for (int j = pl.lo[1]; j <= pl.hi[1]; ++j)
                        for (int i = pl.lo[0]; i <= pl.hi[0]; ++i) {
                            if (side == 0) {
                                // dest = L = r_hi(cell - e), src = R = r_lo(cell)
                                Real s = r_lo[d](i, j, k, n);
                                r_hi[d](i - e[0], j - e[1], k - e[2], n) = (sgnv > 0) ? s : -s;
                            } else {
                                // dest = R = r_lo(cell), src = L = r_hi(cell - e)
                                Real s = r_hi[d](i - e[0], j - e[1], k - e[2], n);
                                r_lo[d](i, j, k, n) = (sgnv > 0) ? s : -s;
                            }
                        }
            }
            if (!S.inflow(d, side)) continue;  // now do any dirichlet boundaries  :840-877
            const Box tb = tab_box(G.domain, G.DIM, S.tng, d, side, true);
            for (int n = 0; n < S.n_prim; ++n) {
                const std::vector<Real>& t = S.tab[d][side][n].face;
                if (t.empty()) continue;
                for (int k = pl.lo[2]; k <= pl.hi[2]; ++k)
                    for (int j = pl.lo[1]; j <= pl.hi[1]; ++j)
                        for (int i = pl.lo[0]; i <= pl.hi[0]; ++i) {
                            const Real v = t[tab_at(tb, i, j, k)];
                            if (side == 0)
                                r_hi[d](i - e[0], j - e[1], k - e[2], n) = v;  // dest = L
                            else
                                r_lo[d](i, j, k, n) = v;  // dest = R
                        }
            }
        }
    }
}

static void st_riemann(const StateDef& S, Real c0, Real* L, Real* R, Real* F) {
    if (S.hydro()) {
        Real shk = 0.0;
        if (S.desc.shock_threshold > 0) shk = shock_detector(L, R, S.desc.shock_threshold);
        switch (S.desc.flux) {
        case CMFP_FLUX_HLLC: hllc_solve(L, R, F, S.gas.nt); break;
        case CMFP_FLUX_HLLE: hlle_solve(L, R, F, S.gas.nt); break;
        case CMFP_FLUX_HYBRID_HLL: hybrid_solve(L, R, F, shk); break;
        }
    } else {
        field_rh_solve(c0, S.div_speed, L, R, F);
    }
}

// EulerianState::calc_fluxes  MFP_eulerian.cpp:957-1045
static void calc_fluxes(const StateDef& S, int DIM, Real c0, Fab* r_lo, Fab* r_hi, Fab* fluxes) {
    Real L[CMFP_MAX_COMP], R[CMFP_MAX_COMP], F[CMFP_MAX_COMP];
    for (int d = 0; d < DIM; ++d) {
        Fab& flux = fluxes[d];
        flux.setVal(0.0);
        const Box& fb = flux.b;
        int e[3] = {0, 0, 0};
        e[d] = 1;
        for (int k = fb.lo[2]; k <= fb.hi[2]; ++k)
            for (int j = fb.lo[1]; j <= fb.hi[1]; ++j)
                for (int i = fb.lo[0]; i <= fb.hi[0]; ++i) {
                    for (int n = 0; n < S.n_prim; ++n) {
                        L[n] = r_hi[d](i - e[0], j - e[1], k - e[2], n);
                        R[n] = r_lo[d](i, j, k, n);
                    }
                    global2local(L, d, S.vec_prim, S.n_vec);
                    global2local(R, d, S.vec_prim, S.n_vec);
                    st_riemann(S, c0, L, R, F);
                    local2global(F, d, S.vec_cons, S.n_vec);
                    for (int n = 0; n < S.n_cons; ++n) { flux(i, j, k, n) += F[n]; }
                }
    }
}

// EulerianState::correct_face_prim  MFP_eulerian.cpp:1047-1188
static void correct_face_prim(const StateDef& S, int DIM, const Box& box, Fab* r_lo, Fab* r_hi,
                              const Fab* fluxes, const Real* dx, Real dt) {
    Real L_prim[CMFP_MAX_COMP], R_prim[CMFP_MAX_COMP], L_cons[CMFP_MAX_COMP], R_cons[CMFP_MAX_COMP];
    Real hiF, loF;
    for (int d1 = 0; d1 < DIM; ++d1) {
        int idx1[3] = {0, 0, 0};
        idx1[d1] = 1;
        Fab& lo4 = r_lo[d1];
        Fab& hi4 = r_hi[d1];
        for (int dd = 1; dd < DIM; ++dd) {
            int d2 = (d1 + dd) % DIM;
            int idx2[3] = {0, 0, 0};
            idx2[d2] = 1;
            const Fab& flux4 = fluxes[d2];
            for (int k = box.lo[2]; k <= box.hi[2] + idx1[2]; ++k)
                for (int j = box.lo[1]; j <= box.hi[1] + idx1[1]; ++j)
                    for (int i = box.lo[0]; i <= box.hi[0] + idx1[0]; ++i) {
                        for (int n = 0; n < S.n_prim; ++n) {
                            L_prim[n] = hi4(i - idx1[0], j - idx1[1], k - idx1[2], n);
                            R_prim[n] = lo4(i, j, k, n);
                        }
                        st_prim2cons(S, L_prim, L_cons);
                        st_prim2cons(S, R_prim, R_cons);
                        for (int n = 0; n < S.n_cons; ++n) {
                            loF = flux4(i - idx1[0], j - idx1[1], k - idx1[2], n);
                            hiF = flux4(i - idx1[0] + idx2[0], j - idx1[1] + idx2[1], k - idx1[2] + idx2[2], n);
                            L_cons[n] += 0.5 * dt / dx[d2] * (loF - hiF);
                        }
                        for (int n = 0; n < S.n_cons; ++n) {
                            loF = flux4(i, j, k, n);
                            hiF = flux4(i + idx2[0], j + idx2[1], k + idx2[2], n);
                            R_cons[n] += 0.5 * dt / dx[d2] * (loF - hiF);
                        }
                        st_cons2prim(S, L_cons, L_prim);
                        st_cons2prim(S, R_cons, R_prim);
                        for (int n = 0; n < S.n_prim; ++n) {
                            hi4(i - idx1[0], j - idx1[1], k - idx1[2], n) = L_prim[n];
                            lo4(i, j, k, n) = R_prim[n];
                        }
                    }
        }
    }
}

// EulerianState::calc_divergence  MFP_eulerian.cpp:1614-1662
static void calc_divergence(int DIM, const Box& box, const Fab* fluxes, Fab& du, const Real* dx, Real dt) {
    du.setVal(0.0);
    int N = du.nc;
    int index[3] = {0, 0, 0};
    Real dxinv;
    for (int n = 0; n < N; ++n)
        for (int k = box.lo[2]; k <= box.hi[2]; ++k)
            for (int j = box.lo[1]; j <= box.hi[1]; ++j)
                for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
                    for (int d = 0; d < DIM; ++d) {
                        index[d] = 1;
                        dxinv = dt / dx[d];
                        const Real& flux_lo = fluxes[d](i, j, k, n);
                        const Real& flux_hi = fluxes[d](i + index[0], j + index[1], k + index[2], n);
                        du(i, j, k, n) += dxinv * (flux_lo - flux_hi);
                        index[d] = 0;
                    }
                }
}

// ---------------------------------------------------------------------------
// Implicit plasma source: dense solve of the 3(Ns+1) system.  The reference
// uses Eigen::ColPivHouseholderQR (MFP_plasma5.cpp:272,404-405; Eigen 3.3.7 is
// vendored by the reference, not by this repository).  Restated here from the
// published algorithm (Businger & Golub 1965: Householder QR with column
// pivoting on the largest remaining column norm), c = P R^-1 Q^T b.
static void colpiv_householder_solve(int N, Real* A /*row-major N*N, destroyed*/, Real* b, Real* x) {
    std::vector<int> perm(N);
    std::vector<Real> cn(N);
    for (int j = 0; j < N; ++j) {
        perm[j] = j;
        Real s = 0;
        for (int i = 0; i < N; ++i) s += A[i * N + j] * A[i * N + j];
        cn[j] = s;
    }
    for (int k = 0; k < N; ++k) {
        int p = k;
        Real best = -1;
        for (int j = k; j < N; ++j) {
            Real s = 0;
            for (int i = k; i < N; ++i) s += A[i * N + j] * A[i * N + j];
            cn[j] = s;
            if (s > best) {
                best = s;
                p = j;
            }
        }
        if (p != k) {
            for (int i = 0; i < N; ++i) std::swap(A[i * N + k], A[i * N + p]);
            std::swap(perm[k], perm[p]);
        }
        // Householder vector for column k
        Real norm = 0;
        for (int i = k; i < N; ++i) norm += A[i * N + k] * A[i * N + k];
        norm = std::sqrt(norm);
        if (norm == 0.0) continue;
        Real alpha = (A[k * N + k] > 0) ? -norm : norm;
        std::vector<Real> v(N, 0.0);
        for (int i = k; i < N; ++i) v[i] = A[i * N + k];
        v[k] -= alpha;
        Real vtv = 0;
        for (int i = k; i < N; ++i) vtv += v[i] * v[i];
        if (vtv == 0.0) continue;
        for (int j = k; j < N; ++j) {
            Real s = 0;
            for (int i = k; i < N; ++i) s += v[i] * A[i * N + j];
            s = 2 * s / vtv;
            for (int i = k; i < N; ++i) A[i * N + j] -= s * v[i];
        }
        Real s = 0;
        for (int i = k; i < N; ++i) s += v[i] * b[i];
        s = 2 * s / vtv;
        for (int i = k; i < N; ++i) b[i] -= s * v[i];
    }
    std::vector<Real> y(N);
    for (int i = N - 1; i >= 0; --i) {
        Real s = b[i];
        for (int j = i + 1; j < N; ++j) s -= A[i * N + j] * y[j];
        y[i] = s / A[i * N + i];
    }
    for (int j = 0; j < N; ++j) x[perm[j]] = y[j];
}

// ---------------------------------------------------------------------------
struct ActionDef {
    cmfp_action_desc desc;
};

enum Status { Inactive = 0, Local, Filled, Expired, Changed, Zero };  // MFP_action.H:17-24
struct UpdateData {
    Status dU_status = Inactive, U_status = Inactive;
    std::vector<Fab> U, dU;  // per box
    int U_ng = -1;
};

struct Sim {
    Geom G;
    int max_grid;
    cmfp_params P;
    std::vector<StateDef> states;
    std::vector<ActionDef> actions;
    std::vector<Box> boxes;
    std::vector<std::vector<Fab>> old_data, new_data;  // [state][box], no ghosts
    int nsteps;

    // --- FillPatch (MFP_action.cpp:59): same-level copy + periodic wrap +
    // FillBC -> fab_filcc (MFP_fillbc.cpp:147-172) [AMReX semantics from
    // memory: dimension by dimension over the full extent of the lower
    // dimensions' ghost cells].
    void fill_patch(int s, const std::vector<Fab>& src, std::vector<Fab>& U, int ng) const {
        const StateDef& S = states[s];
        const int DIM = G.DIM;
        Fab glob;
        Box gb = grow(G.domain, ng, DIM);
        glob.resize(gb, S.n_cons);
        glob.setVal(std::numeric_limits<Real>::quiet_NaN());
        // (the copies of this function are threaded over the boxes since round 3: three FillPatch calls per state
        // and step were the serial part that held the 16-thread baseline of bench.py to 4.4 x one core)
#pragma omp parallel for schedule(static)
        for (long b = 0; b < (long)boxes.size(); ++b) {
            const Box& bx = boxes[b];
            for (int n = 0; n < S.n_cons; ++n)
                for (int k = bx.lo[2]; k <= bx.hi[2]; ++k)
                    for (int j = bx.lo[1]; j <= bx.hi[1]; ++j)
                        for (int i = bx.lo[0]; i <= bx.hi[0]; ++i) glob(i, j, k, n) = src[b](i, j, k, n);
        }
        for (int d = 0; d < DIM; ++d) {
            // region: full extent in dims < d (ghosts included), valid+ghost in dims > d too
            // (values there are overwritten by the later sweeps)
            Box reg = gb;
            const int dlo = G.domain.lo[d], dhi = G.domain.hi[d];
            for (int n = 0; n < S.n_cons; ++n) {
                for (int side = 0; side < 2; ++side) {
                    int bc = G.periodic[d] ? CMFP_BC_INT_DIR : S.desc.bc[n][d][side];
                    for (int g = 1; g <= ng; ++g) {
                        int idst = side == 0 ? dlo - g : dhi + g;
                        int isrc;
                        Real sg = 1.0;
                        if (G.periodic[d]) {
                            isrc = side == 0 ? dhi + 1 - g : dlo - 1 + g;
                        } else if (bc == CMFP_BC_FOEXTRAP || bc == CMFP_BC_HOEXTRAP || bc == CMFP_BC_EXT_DIR ||
                                   bc == CMFP_BC_INT_DIR) {
                            // hoextrap / ext_dir / int_dir at a physical face are not
                            // used by the path's BC tables (MFP_bc.cpp); treated as foextrap
                            isrc = side == 0 ? dlo : dhi;
                        } else if (bc == CMFP_BC_REFLECT_EVEN) {
                            isrc = side == 0 ? dlo + g - 1 : dhi - g + 1;
                        } else {  // reflect_odd
                            isrc = side == 0 ? dlo + g - 1 : dhi - g + 1;
                            sg = -1.0;
                        }
                        int c[3], cs[3];
                        Box pl = reg;
                        pl.lo[d] = pl.hi[d] = idst;
                        for (c[2] = pl.lo[2]; c[2] <= pl.hi[2]; ++c[2])
                            for (c[1] = pl.lo[1]; c[1] <= pl.hi[1]; ++c[1])
                                for (c[0] = pl.lo[0]; c[0] <= pl.hi[0]; ++c[0]) {
                                    cs[0] = c[0];
                                    cs[1] = c[1];
                                    cs[2] = c[2];
                                    cs[d] = isrc;
                                    Real v = glob(cs[0], cs[1], cs[2], n);
                                    glob(c[0], c[1], c[2], n) = (sg > 0) ? v : -v;
                                }
                    }
                }
            }
        }
        U.resize(boxes.size());
#pragma omp parallel for schedule(static)
        for (long b = 0; b < (long)boxes.size(); ++b) {
            Box ub = grow(boxes[b], ng, DIM);
            U[b].resize(ub, S.n_cons);
            for (int n = 0; n < S.n_cons; ++n)
                for (int k = ub.lo[2]; k <= ub.hi[2]; ++k)
                    for (int j = ub.lo[1]; j <= ub.hi[1]; ++j)
                        for (int i = ub.lo[0]; i <= ub.hi[0]; ++i) U[b](i, j, k, n) = glob(i, j, k, n);
        }
    }

    // Action::get_data  MFP_action.cpp:15-102.  `use_new`: which time level
    // FillPatch / get_data(time) read (old at `time`, new at `time+dt`; they
    // hold identical values when read at `time`, MFP_advance.cpp:19-25).
    void get_data(const std::vector<std::pair<int, int>>& options, std::vector<UpdateData>& update) {
        for (auto& opt : options) {
            int s = opt.first;
            bool fill = opt.second;
            const StateDef& S = states[s];
            UpdateData& ud = update[s];
            int ns = S.n_cons;
            if (fill) {
                if (ud.U_status != Filled) {
                    int ng = S.num_grow;
                    fill_patch(s, new_data[s], ud.U, ng);
                    ud.U_ng = ng;
                    ud.U_status = Filled;
                }
            } else {
                if ((ud.U_status != Local) && (ud.U_status != Filled)) {
                    if (ud.U_status == Inactive) {
                        ud.U.resize(boxes.size());
                        for (size_t b = 0; b < boxes.size(); ++b) ud.U[b].resize(boxes[b], ns);
                        ud.U_ng = 0;
                    }
#pragma omp parallel for schedule(static)
                    for (long b = 0; b < (long)boxes.size(); ++b) {
                        const Box& bx = boxes[b];
                        for (int n = 0; n < ns; ++n)
                            for (int k = bx.lo[2]; k <= bx.hi[2]; ++k)
                                for (int j = bx.lo[1]; j <= bx.hi[1]; ++j)
                                    for (int i = bx.lo[0]; i <= bx.hi[0]; ++i)
                                        ud.U[b](i, j, k, n) = new_data[s][b](i, j, k, n);
                    }
                    ud.U_status = Local;
                }
            }
            if (ud.dU_status == Inactive) {
                ud.dU.resize(boxes.size());
                for (size_t b = 0; b < boxes.size(); ++b) ud.dU[b].resize(boxes[b], ns);
            }
            if ((ud.dU_status == Expired) || (ud.dU_status == Inactive)) {
                for (auto& f : ud.dU) f.setVal(0.0);
                ud.dU_status = Zero;
            }
        }
    }

    void action_get_data(int a, std::vector<UpdateData>& update) {
        const cmfp_action_desc& A = actions[a].desc;
        std::vector<std::pair<int, int>> options;
        if (A.kind == CMFP_ACTION_CTU) {
            // HydroCTU::get_data  MFP_CTU_hydro.cpp:43-52
            for (int i = 0; i < A.n_states; ++i) options.push_back({A.states[i], 1});
        } else if (A.kind == CMFP_ACTION_COLLISIONS_RAMBO) {
            // Collisions::get_data MFP_collisions.cpp:83-93
            for (int i = 0; i < A.n_states; ++i) options.push_back({A.states[i], 0});
        } else {
            // Plasma5::get_data MFP_plasma5.cpp:60-71 / Lorentz::get_data MFP_lorentz.cpp:47-58
            options.push_back({A.states[A.n_states - 1], 0});
            for (int i = 0; i < A.n_states - 1; ++i) options.push_back({A.states[i], 0});
        }
        get_data(options, update);
    }

    // HydroCTU::calc_spatial_derivative  MFP_CTU_hydro.cpp:54-532 (no EB, no
    // flux registers, no viscous terms)
    void ctu_spatial_derivative(int a, std::vector<UpdateData>& update, Real dt) {
        const cmfp_action_desc& A = actions[a].desc;
        const int DIM = G.DIM;
        const bool do_CTU = A.corner_transport != 0;
        for (int i = 0; i < A.n_states; ++i) update[A.states[i]].dU_status = Changed;  // :88
#pragma omp parallel for schedule(dynamic)
        for (long b = 0; b < (long)boxes.size(); ++b) {
            const Box& box = boxes[b];
            const Box rbox = grow(box, 1, DIM);
            for (int idx = 0; idx < A.n_states; ++idx) {
                const int s = A.states[idx];
                const StateDef& S = states[s];
                Fab& cons = update[s].U[b];
                Fab prim;
                Fab R_lo[3], R_hi[3], fluxes[3];
                const Box pbox = grow(box, S.num_grow, DIM);
                calc_primitives(S, pbox, cons, prim);
                apply_dynamic(S, DIM, G.domain, pbox, cons, prim);                // MFP_eulerian.cpp:403-423
                update_boundary_cells(S, DIM, G.domain, G.periodic, pbox, prim);  // MFP_CTU_hydro.cpp:197
                if (S.hydro())
                    calc_reconstruction_hydro(S, DIM, rbox, prim, R_lo, R_hi);
                else
                    calc_reconstruction_generic(S, DIM, rbox, prim, R_lo, R_hi);
                if (do_CTU) calc_time_averaged_faces(S, DIM, rbox, prim, R_lo, R_hi, G.dx, dt);
                for (int d = 0; d < DIM; ++d) {
                    Box fbox = box;  // surroundingNodes(box, d)
                    fbox.hi[d] += 1;
                    if (do_CTU) {
                        for (int t = 0; t < DIM; ++t)
                            if (t != d) {
                                fbox.lo[t] -= 1;
                                fbox.hi[t] += 1;
                            }
                    }
                    fluxes[d].resize(fbox, S.n_cons);
                }
                update_face_prim(S, G, box, R_lo, R_hi, false);
                calc_fluxes(S, DIM, P.lightspeed, R_lo, R_hi, fluxes);
                if (DIM > 1 && do_CTU) {
                    correct_face_prim(S, DIM, box, R_lo, R_hi, fluxes, G.dx, dt);
                    update_face_prim(S, G, box, R_lo, R_hi, true);
                    calc_fluxes(S, DIM, P.lightspeed, R_lo, R_hi, fluxes);
                }
                if (S.hydro() && S.desc.viscosity) calc_viscous_fluxes(S, DIM, box, fluxes, prim, G.dx);  // MFP_CTU_hydro.cpp:370-380
                calc_divergence(DIM, box, fluxes, update[s].dU[b], G.dx, dt);
            }
        }
    }

    // Plasma5::explicit_solve  MFP_plasma5.cpp:87-244 ; Lorentz::calc_time_derivative
    // MFP_lorentz.cpp:60-198 (lorentz_only = species part only)
    void plasma5_explicit(int a, std::vector<UpdateData>& update, Real dt, bool lorentz_only) {
        const cmfp_action_desc& A = actions[a].desc;
        const int n_species = A.n_states - 1;
        const int fs = A.states[A.n_states - 1];
        const StateDef& field = states[fs];
        if (!lorentz_only) update[fs].dU_status = Changed;
        for (int n = 0; n < n_species; ++n) update[A.states[n]].dU_status = Changed;

        const Real Larmor = P.Larmor, Debye = P.Debye, lightspeed = P.lightspeed;
        const Real c_h = field.div_speed;
        const Real div_damp = field.desc.div_damping;
        const Real c_d = div_damp != 0.0 ? c_h / div_damp : 0.0;
        const Real f1 = dt * Larmor / (Debye * Debye * lightspeed);
        const Real f2 = dt * c_h * c_h * f1 / lightspeed;
        const Real f3 = c_d != 0.0 ? dt * c_h * c_h / (c_d * c_d) : 0.0;

#pragma omp parallel for schedule(dynamic)
        for (long b = 0; b < (long)boxes.size(); ++b) {
            const Box& box = boxes[b];
            Fab& field4 = update[fs].U[b];
            Fab& field_dU4 = update[fs].dU[b];
            Real Bx_, By_, Bz_, Ex, Ey, Ez, ep_, q, m, r, rho, mx, my, mz, u, v, w;
            Real charge_density, current_x, current_y, current_z;
            for (int k = box.lo[2]; k <= box.hi[2]; ++k)
                for (int j = box.lo[1]; j <= box.hi[1]; ++j)
                    for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
                        ep_ = field4(i, j, k, ep);
                        Bx_ = field4(i, j, k, Bx);
                        By_ = field4(i, j, k, By);
                        Bz_ = field4(i, j, k, Bz);
                        Ex = field4(i, j, k, Dx) / ep_;
                        Ey = field4(i, j, k, Dy) / ep_;
                        Ez = field4(i, j, k, Dz) / ep_;
                        charge_density = 0.0;
                        current_x = 0.0;
                        current_y = 0.0;
                        current_z = 0.0;
                        for (int n = 0; n < n_species; ++n) {
                            const int s = A.states[n];
                            Fab& sp4 = update[s].U[b];
                            Fab& dU4 = update[s].dU[b];
                            rho = sp4(i, j, k, Density);
                            mx = sp4(i, j, k, Xmom);
                            my = sp4(i, j, k, Ymom);
                            mz = sp4(i, j, k, Zmom);
                            states[s].gas.mass_charge_from_cons(sp4, i, j, k, m, q);  // get_mass_from_cons, get_charge_from_cons
                            r = q / m;
                            charge_density += rho * r;
                            current_x += r * mx;
                            current_y += r * my;
                            current_z += r * mz;
                            u = mx / rho;
                            v = my / rho;
                            w = mz / rho;
                            dU4(i, j, k, Xmom) += dt * (rho * r / Larmor) * (lightspeed * Ex + v * Bz_ - w * By_);
                            dU4(i, j, k, Ymom) += dt * (rho * r / Larmor) * (lightspeed * Ey + w * Bx_ - u * Bz_);
                            dU4(i, j, k, Zmom) += dt * (rho * r / Larmor) * (lightspeed * Ez + u * By_ - v * Bx_);
                            dU4(i, j, k, Eden) += dt * (rho * r * lightspeed) / Larmor * (u * Ex + v * Ey + w * Ez);
                        }
                        if (lorentz_only) continue;
                        field_dU4(i, j, k, Dx) += -f1 * current_x;
                        field_dU4(i, j, k, Dy) += -f1 * current_y;
                        field_dU4(i, j, k, Dz) += -f1 * current_z;
                        field_dU4(i, j, k, phi) += -field4(i, j, k, phi) * f3 + f2 * charge_density;
                        field_dU4(i, j, k, psi) += -field4(i, j, k, psi) * f3;
                    }
        }
    }

    // Plasma5::implicit_solve  MFP_plasma5.cpp:246-455
    void plasma5_implicit(int a, std::vector<UpdateData>& update, Real dt) {
        const cmfp_action_desc& A = actions[a].desc;
        const int n_species = A.n_states - 1;
        const int fs = A.states[A.n_states - 1];
        const StateDef& field = states[fs];
        update[fs].dU_status = Changed;
        for (int n = 0; n < n_species; ++n) update[A.states[n]].dU_status = Changed;

        const int N = 3 * (n_species + 1);
        const Real Larmor = P.Larmor, Debye = P.Debye, lightspeed = P.lightspeed;
        const Real c_h = field.div_speed;
        const Real div_damp = field.desc.div_damping;
        const Real c_d = div_damp != 0.0 ? c_h / div_damp : 0.0;
        const Real f1 = dt * Larmor / (Debye * Debye * lightspeed);
        const Real f2 = dt * c_h * c_h * f1 / lightspeed;
        const Real f3 = c_d != 0.0 ? dt * c_h * c_h / (c_d * c_d) : 0.0;

#pragma omp parallel for schedule(dynamic)
        for (long b = 0; b < (long)boxes.size(); ++b) {
            const Box& box = boxes[b];
            Fab& field4 = update[fs].U[b];
            Fab& field_dU4 = update[fs].dU[b];
            std::vector<Real> Amat(N * N), Awork(N * N), bvec(N), bwork(N), c(N), Rr(n_species);
            // A = Identity once (:269); entries below are overwritten per cell
            std::fill(Amat.begin(), Amat.end(), 0.0);
            for (int i = 0; i < N; ++i) Amat[i * N + i] = 1.0;
            Real trace_B, trace_C, mc, Bx_, By_, Bz_, Ex, Ey, Ez, ep_, charge_density, q, m, r, rho, mx, my, mz;
            for (int k = box.lo[2]; k <= box.hi[2]; ++k)
                for (int j = box.lo[1]; j <= box.hi[1]; ++j)
                    for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
                        ep_ = field4(i, j, k, ep);
                        Bx_ = field4(i, j, k, Bx);
                        By_ = field4(i, j, k, By);
                        Bz_ = field4(i, j, k, Bz);
                        Ex = field4(i, j, k, Dx) / ep_;
                        Ey = field4(i, j, k, Dy) / ep_;
                        Ez = field4(i, j, k, Dz) / ep_;
                        bvec[n_species * 3 + 0] = Ex;
                        bvec[n_species * 3 + 1] = Ey;
                        bvec[n_species * 3 + 2] = Ez;
                        charge_density = 0.0;
                        for (int n = 0; n < n_species; ++n) {
                            const int s = A.states[n];
                            Fab& sp4 = update[s].U[b];
                            rho = sp4(i, j, k, Density);
                            mx = sp4(i, j, k, Xmom);
                            my = sp4(i, j, k, Ymom);
                            mz = sp4(i, j, k, Zmom);
                            states[s].gas.mass_charge_from_cons(sp4, i, j, k, m, q);  // get_mass_from_cons, get_charge_from_cons
                            r = q / m;
                            Rr[n] = r;
                            charge_density += rho * r;
                            charge_density += rho * r;  // :370-372 (quirk Q3: counted twice)
                            trace_B = -dt * r * rho * lightspeed / Larmor;
                            trace_C = dt * r * Larmor / (lightspeed * Debye * Debye);
                            mc = dt * r / Larmor;
                            Amat[(3 * n + 0) * N + 3 * n + 1] = -Bz_ * mc;
                            Amat[(3 * n + 0) * N + 3 * n + 2] = By_ * mc;
                            Amat[(3 * n + 1) * N + 3 * n + 0] = Bz_ * mc;
                            Amat[(3 * n + 1) * N + 3 * n + 2] = -Bx_ * mc;
                            Amat[(3 * n + 2) * N + 3 * n + 0] = -By_ * mc;
                            Amat[(3 * n + 2) * N + 3 * n + 1] = Bx_ * mc;
                            Amat[(3 * n + 0) * N + 3 * n_species + 0] = trace_B;
                            Amat[(3 * n + 1) * N + 3 * n_species + 1] = trace_B;
                            Amat[(3 * n + 2) * N + 3 * n_species + 2] = trace_B;
                            Amat[(3 * n_species + 0) * N + 3 * n + 0] = trace_C;
                            Amat[(3 * n_species + 1) * N + 3 * n + 1] = trace_C;
                            Amat[(3 * n_species + 2) * N + 3 * n + 2] = trace_C;
                            bvec[n * 3 + 0] = mx;
                            bvec[n * 3 + 1] = my;
                            bvec[n * 3 + 2] = mz;
                        }
                        Awork = Amat;
                        bwork = bvec;
                        colpiv_householder_solve(N, Awork.data(), bwork.data(), c.data());

                        field_dU4(i, j, k, Dx) += c[n_species * 3 + 0] * ep_ - field4(i, j, k, Dx);
                        field_dU4(i, j, k, Dy) += c[n_species * 3 + 1] * ep_ - field4(i, j, k, Dy);
                        field_dU4(i, j, k, Dz) += c[n_species * 3 + 2] * ep_ - field4(i, j, k, Dz);
                        Ex = c[n_species * 3 + 0];
                        Ey = c[n_species * 3 + 1];
                        Ez = c[n_species * 3 + 2];
                        for (int n = 0; n < n_species; ++n) {
                            const int s = A.states[n];
                            Fab& sp4 = update[s].U[b];
                            Fab& dU4 = update[s].dU[b];
                            mx = c[n * 3 + 0];
                            my = c[n * 3 + 1];
                            mz = c[n * 3 + 2];
                            dU4(i, j, k, Xmom) += mx - sp4(i, j, k, Xmom);
                            dU4(i, j, k, Ymom) += my - sp4(i, j, k, Ymom);
                            dU4(i, j, k, Zmom) += mz - sp4(i, j, k, Zmom);
                            dU4(i, j, k, Eden) += dt * (Rr[n] * lightspeed / Larmor) * (Ex * mx + Ey * my + Ez * mz);
                        }
                        field_dU4(i, j, k, phi) += -field4(i, j, k, phi) * f3 + f2 * charge_density;
                        field_dU4(i, j, k, psi) += -field4(i, j, k, psi) * f3;
                    }
        }
    }

    // Collisions::calc_time_derivative MFP_collisions.cpp:95-267 with CollisionsRambo::calc_update
    // MFP_collisions_rambo.cpp:68-126 (no EB).  A state of several sub-components (tracers: passive or a mixture)
    // takes part with the one named in the action's `species` list (SpeciesInfo::alpha_idx, :47-57 =
    // cmfp_action_desc::species_component).  The reference reads U[n], Q[n], alpha[n] and states[n] with the
    // species index (:181-190), which is the species' own state only where species n lives in state n: that is the
    // configuration restated here (one species per state, in list order).
    void collisions_rambo(int a, std::vector<UpdateData>& update, Real dt) {
        const cmfp_action_desc& A = actions[a].desc;
        const int n_species = A.n_states;
        for (int n = 0; n < n_species; ++n) update[A.states[n]].dU_status = Changed;
        // coefficients, MFP_collisions_rambo.cpp:57-64
        const Real Debye = P.Debye, n0 = P.n0;
        const Real c1 = 0.02116454531141366 / (n0 * pow(Debye, 4));
        const Real c2 = 0.4135669939329334;
        const Real c3 = 2.127692162140974 * n0;
        const Real lnC = 10.0;
        struct Info {
            Real rho, m, q, T, q2, n, vel[3], delta[5];
        };
#pragma omp parallel for schedule(dynamic)
        for (long b = 0; b < (long)boxes.size(); ++b) {
            const Box& box = boxes[b];
            Info sp[CMFP_MAX_ACTION_STATES];
            Real U[CMFP_MAX_ACTION_STATES][HC_NUM + CMFP_MAX_TRACERS], Q[CMFP_MAX_ACTION_STATES][HP_NUM + CMFP_MAX_TRACERS];
            Real alpha[CMFP_MAX_ACTION_STATES][CMFP_MAX_TRACERS + 1];
            Real accounting[CMFP_MAX_TRACERS + 1][5];
            for (int si = 0; si < n_species; ++si)
                for (int l = 0; l <= CMFP_MAX_TRACERS; ++l) alpha[si][l] = 0.0;
            for (int k = box.lo[2]; k <= box.hi[2]; ++k)
                for (int j = box.lo[1]; j <= box.hi[1]; ++j)
                    for (int i = box.lo[0]; i <= box.hi[0]; ++i) {
                        for (int si = 0; si < n_species; ++si) {
                            const StateDef& S = states[A.states[si]];
                            for (int l = 0; l < S.n_cons; ++l) U[si][l] = update[A.states[si]].U[b](i, j, k, l);
                            if (U[si][Density] < S.desc.effective_zero) {
                                for (int l = 0; l < HP_NUM + S.gas.nt; ++l) Q[si][l] = 0.0;
                                // (the reference leaves alpha at the previous cell's values here, :166-171)
                            } else {
                                gas_cons2prim(S.gas, U[si], Q[si]);
                                // get_alpha_fractions_from_prim MFP_hydro_gas.cpp:13-27 (no clamp)
                                Real sum_alpha = 0.0;
                                for (int l = 0; l < S.gas.nt; ++l) {
                                    alpha[si][l] = Q[si][HP_NUM + l];
                                    sum_alpha += alpha[si][l];
                                }
                                alpha[si][S.gas.nt] = 1.0 - sum_alpha;
                            }
                        }
                        for (int n = 0; n < n_species; ++n) {
                            const StateDef& S = states[A.states[n]];
                            const int ai = A.species_component[n];
                            Info& info = sp[n];
                            for (int l = 0; l < 5; ++l) info.delta[l] = 0.0;
                            const Real al = alpha[n][ai];
                            if (U[n][Density] > S.desc.effective_zero) {
                                info.rho = al * U[n][Density];
                                info.m = S.gas.sm[ai];
                                info.q = S.gas.sq[ai];
                                info.T = Q[n][Temp];
                                info.q2 = info.q * info.q;
                                info.n = info.rho / info.m;
                                const Real i_rho = 1 / info.rho;
                                info.vel[0] = al * U[n][Xmom] * i_rho;
                                info.vel[1] = al * U[n][Ymom] * i_rho;
                                info.vel[2] = al * U[n][Zmom] * i_rho;
                            } else {
                                info.rho = info.m = info.q = info.T = info.q2 = info.n = 0.0;
                                info.vel[0] = info.vel[1] = info.vel[2] = 0.0;
                            }
                        }
                        // calc_update
                        for (int ia = 0; ia < n_species; ++ia) {
                            Info& sda = sp[ia];
                            for (int ib = ia + 1; ib < n_species; ++ib) {
                                Info& sdb = sp[ib];
                                const Real m_ab = (sda.m * sdb.m) / (sda.m + sdb.m);
                                Real du2 = 0.0;
                                Real du[3];
                                for (int d = 0; d < 3; ++d) {
                                    du[d] = sdb.vel[d] - sda.vel[d];
                                    du2 += du[d] * du[d];
                                }
                                Real nu;
                                if ((sda.q != 0) && (sdb.q != 0)) {
                                    const Real coeff_1 = c1 * ((sda.q2 * sdb.q2 * lnC) / (m_ab * sda.m));
                                    nu = sdb.n * coeff_1 * pow(c2 * du2 + sda.T / sda.m + sdb.T / sdb.m, -1.5);
                                } else {
                                    const Real coeff_2 = (sdb.m * A.cross_sections[ia][ib]) / (sda.m + sdb.m);
                                    nu = sdb.n * coeff_2 * c3 * sqrt(sda.T / sda.m + sdb.T / sdb.m);
                                }
                                const Real S_ = 0.0;
                                Real Qx = m_ab * sda.n * nu * du2 + 3 * sda.rho * nu / (sda.m + sdb.m) * (sdb.T - sda.T);
                                Real R[3];
                                for (int d = 0; d < 3; ++d) {
                                    R[d] = sda.rho * nu * du[d];
                                    Qx += R[d] * sda.vel[d];
                                }
                                sda.delta[Density] += S_;
                                sdb.delta[Density] -= S_;
                                for (int d = 0; d < 3; ++d) {
                                    sda.delta[Xmom + d] += R[d];
                                    sdb.delta[Xmom + d] -= R[d];
                                }
                                sda.delta[Eden] += Qx;
                                sdb.delta[Eden] -= Qx;
                            }
                        }
                        // accounting (:218-257): every sub-component's share alpha_ai U, the participating one plus
                        // dt delta, summed again, minus the original; tracers from the density share of their component
                        for (int si = 0; si < n_species; ++si) {
                            const StateDef& S = states[A.states[si]];
                            Fab& U4 = update[A.states[si]].U[b];
                            Fab& dU4 = update[A.states[si]].dU[b];
                            for (int ai = 0; ai <= S.gas.nt; ++ai)
                                for (int ci = 0; ci < 5; ++ci) accounting[ai][ci] = U[si][ci] * alpha[si][ai];
                            for (int ci = 0; ci < 5; ++ci) accounting[A.species_component[si]][ci] += dt * sp[si].delta[ci];
                            for (int ci = 0; ci < 5; ++ci) {
                                Real sum = 0.0;
                                for (int ai = 0; ai <= S.gas.nt; ++ai) sum += accounting[ai][ci];
                                dU4(i, j, k, ci) += sum - U4(i, j, k, ci);
                            }
                            for (int ti = 0; ti < S.gas.nt; ++ti)
                                dU4(i, j, k, HC_NUM + ti) += accounting[ti][Density] - U4(i, j, k, HC_NUM + ti);
                        }
                    }
        }
    }

    void action_calc_time_derivative(int a, std::vector<UpdateData>& update, Real dt) {
        const cmfp_action_desc& A = actions[a].desc;
        if (A.kind == CMFP_ACTION_COLLISIONS_RAMBO) {
            collisions_rambo(a, update, dt);
        } else if (A.kind == CMFP_ACTION_PLASMA5) {
            if (A.solver == CMFP_SOLVER_EXPLICIT)
                plasma5_explicit(a, update, dt, false);
            else
                plasma5_implicit(a, update, dt);
        } else if (A.kind == CMFP_ACTION_LORENTZ) {
            plasma5_explicit(a, update, dt, true);
        }
    }
    void action_calc_spatial_derivative(int a, std::vector<UpdateData>& update, Real dt) {
        if (actions[a].desc.kind == CMFP_ACTION_CTU) ctu_spatial_derivative(a, update, dt);
    }

    // MultiFab helpers on valid cells
    void mf_add(std::vector<Fab>& dst, const std::vector<Fab>& src) {  // dst += src (MultiFab::Add, ng = 0)
#pragma omp parallel for
        for (long b = 0; b < (long)boxes.size(); ++b) {
            const Box& bx = boxes[b];
            for (int n = 0; n < src[b].nc; ++n)
                for (int k = bx.lo[2]; k <= bx.hi[2]; ++k)
                    for (int j = bx.lo[1]; j <= bx.hi[1]; ++j)
                        for (int i = bx.lo[0]; i <= bx.hi[0]; ++i) dst[b](i, j, k, n) += src[b](i, j, k, n);
        }
    }
    void mf_lincomb(std::vector<Fab>& dst, Real a, const std::vector<Fab>& x, Real bb, const std::vector<Fab>& y) {
#pragma omp parallel for
        for (long b = 0; b < (long)boxes.size(); ++b) {
            const Box& bx = boxes[b];
            for (int n = 0; n < dst[b].nc; ++n)
                for (int k = bx.lo[2]; k <= bx.hi[2]; ++k)
                    for (int j = bx.lo[1]; j <= bx.hi[1]; ++j)
                        for (int i = bx.lo[0]; i <= bx.hi[0]; ++i)
                            dst[b](i, j, k, n) = a * x[b](i, j, k, n) + bb * y[b](i, j, k, n);  // MultiFab::LinComb
        }
    }

    // MFP::advance_RK  MFP_advance.cpp:58-135
    void advance_RK(Real dt) {
        const int ns = (int)states.size();
        std::vector<UpdateData> RK_step(ns);
        for (size_t a = 0; a < actions.size(); ++a) {
            action_get_data(a, RK_step);
            action_calc_time_derivative(a, RK_step, dt);
            action_calc_spatial_derivative(a, RK_step, dt);
        }
        for (int s = 0; s < ns; ++s) {
            if (RK_step[s].dU_status != Changed) continue;
            mf_add(new_data[s], RK_step[s].dU);
            RK_step[s].U_status = Expired;
            RK_step[s].dU_status = Expired;
        }
        if (nsteps >= 2) {
            for (size_t a = 0; a < actions.size(); ++a) {
                action_get_data(a, RK_step);
                action_calc_time_derivative(a, RK_step, dt);
                action_calc_spatial_derivative(a, RK_step, dt);
            }
            for (int s = 0; s < ns; ++s) {
                if (RK_step[s].dU_status != Changed) continue;
                mf_add(RK_step[s].U, RK_step[s].dU);
                mf_lincomb(new_data[s], 0.5, old_data[s], 0.5, RK_step[s].U);
            }
        }
    }

    // MFP::advance_strang  MFP_advance.cpp:137-289
    void advance_strang(Real dt) {
        const int ns = (int)states.size();
        std::vector<UpdateData> RK_step(ns);

        for (size_t a = 0; a < actions.size(); ++a) {
            action_get_data(a, RK_step);
            action_calc_time_derivative(a, RK_step, 0.5 * dt);
        }
        for (int s = 0; s < ns; ++s) {
            if (RK_step[s].dU_status != Changed) continue;
            if (nsteps >= 2) {
                mf_add(RK_step[s].U, RK_step[s].dU);
                for (auto& f : RK_step[s].dU) f.setVal(0.0);
                RK_step[s].dU_status = Expired;
            } else {
                mf_add(new_data[s], RK_step[s].dU);
            }
        }
        if (nsteps >= 2) {
            for (size_t a = 0; a < actions.size(); ++a) action_calc_time_derivative(a, RK_step, 0.5 * dt);
            for (int s = 0; s < ns; ++s) {
                if (RK_step[s].dU_status != Changed) continue;
                mf_add(RK_step[s].U, RK_step[s].dU);
                for (auto& f : RK_step[s].dU) f.setVal(0.0);
                mf_lincomb(new_data[s], 0.5, old_data[s], 0.5, RK_step[s].U);
                RK_step[s].U_status = Expired;
                RK_step[s].dU_status = Expired;
            }
        }

        // fluxes
        for (size_t a = 0; a < actions.size(); ++a) {
            action_get_data(a, RK_step);
            action_calc_spatial_derivative(a, RK_step, dt);
        }
        for (int s = 0; s < ns; ++s) {
            if (RK_step[s].dU_status != Changed) continue;
            mf_add(new_data[s], RK_step[s].dU);
            RK_step[s].U_status = Expired;
            RK_step[s].dU_status = Expired;
        }

        // sources
        for (size_t a = 0; a < actions.size(); ++a) {
            action_get_data(a, RK_step);
            action_calc_time_derivative(a, RK_step, 0.5 * dt);
        }
        for (int s = 0; s < ns; ++s) {
            if (RK_step[s].dU_status != Changed) continue;
            if (nsteps >= 2) {
                mf_add(RK_step[s].U, RK_step[s].dU);
                for (auto& f : RK_step[s].dU) f.setVal(0.0);
                RK_step[s].dU_status = Inactive;
            } else {
                mf_add(new_data[s], RK_step[s].dU);
            }
        }
        if (nsteps >= 2) {
            for (size_t a = 0; a < actions.size(); ++a) action_calc_time_derivative(a, RK_step, 0.5 * dt);
            for (int s = 0; s < ns; ++s) {
                if (RK_step[s].dU_status != Changed) continue;
                mf_add(RK_step[s].U, RK_step[s].dU);
                for (auto& f : RK_step[s].dU) f.setVal(0.0);
                mf_lincomb(new_data[s], 0.5, new_data[s], 0.5, RK_step[s].U);
            }
        }
    }

    // MFP::advance  MFP_advance.cpp:6-56
    void advance(Real dt) {
        // swapTimeLevels + copy old into new: old := previous new, new := old
        for (size_t s = 0; s < states.size(); ++s) old_data[s] = new_data[s];
        if (P.time_integrator == CMFP_TI_STRANG)
            advance_strang(dt);
        else
            advance_RK(dt);
    }

    // EulerianState::get_allowed_time_step  MFP_eulerian.cpp:255-311 ;
    // FieldState::get_allowed_time_step  MFP_field.cpp:57-67
    Real state_allowed_dt(int s) const {
        const StateDef& S = states[s];
        const int DIM = G.DIM;
        if (!S.hydro()) {
            Real dt = G.dx[0] / S.fastest_speed;
            for (int i = 1; i < DIM; ++i) dt = std::min(dt, G.dx[i] / S.fastest_speed);
            return dt;
        }
        Real max_speed = std::numeric_limits<Real>::min();
        // (a maximum / minimum is the same whatever the order: threaded over the boxes)
#pragma omp parallel for schedule(static) reduction(max : max_speed)
        for (long b = 0; b < (long)boxes.size(); ++b) {
            Real U[CMFP_MAX_COMP], sp[3];
            const Box& bx = boxes[b];
            const Fab& data = new_data[s][b];
            for (int k = bx.lo[2]; k <= bx.hi[2]; ++k)
                for (int j = bx.lo[1]; j <= bx.hi[1]; ++j)
                    for (int i = bx.lo[0]; i <= bx.hi[0]; ++i) {
                        for (int n = 0; n < S.n_cons; ++n) U[n] = data(i, j, k, n);
                        st_speed_from_cons(S, U, sp);
                        for (int d = 0; d < DIM; ++d) max_speed = std::max(max_speed, sp[d]);
                    }
        }
        Real dt = G.dx[0] / max_speed;
        for (int i = 1; i < DIM; ++i) dt = std::min(dt, G.dx[i] / max_speed);
        if (S.desc.viscosity) {
            // HydroState::get_allowed_time_step MFP_hydro.cpp:726-736 with HydroViscous::get_min_dt
            // (MFP_hydro_viscous.cpp:34-97): max over cells starts from 0
            Real vmax = 0;
            Real U[CMFP_MAX_COMP];
            for (size_t b = 0; b < boxes.size(); ++b) {
                const Box& bx = boxes[b];
                const Fab& data = new_data[s][b];
                for (int k = bx.lo[2]; k <= bx.hi[2]; ++k)
                    for (int j = bx.lo[1]; j <= bx.hi[1]; ++j)
                        for (int i = bx.lo[0]; i <= bx.hi[0]; ++i) {
                            for (int n = 0; n < S.n_cons; ++n) U[n] = data(i, j, k, n);
                            vmax = std::max(vmax, visc_stability(S, U));
                        }
            }
            Real min_dt = std::numeric_limits<Real>::max();
            for (int d = 0; d < DIM; ++d) min_dt = std::min(min_dt, G.dx[d] * G.dx[d] / vmax);
            dt = std::min(dt, min_dt);
        }
        return dt;
    }

    // Plasma5::get_allowed_time_step  MFP_plasma5.cpp:457-567
    Real action_allowed_dt(int a) const {
        const cmfp_action_desc& A = actions[a].desc;
        if (A.kind != CMFP_ACTION_PLASMA5) return std::numeric_limits<Real>::max();
        if (A.solver != CMFP_SOLVER_EXPLICIT) return std::numeric_limits<Real>::max();
        const Real PI = acos(-1);
        const int DIM = G.DIM;
        const int n_species = A.n_states - 1;
        const int fs = A.states[A.n_states - 1];
        const StateDef& field = states[fs];
        const Real Larmor = P.Larmor, Debye = P.Debye;
        Real D2 = Debye * Debye;
        Real max_speed = P.lightspeed;
        max_speed = std::max(max_speed, field.div_speed);
        Real dt = G.dx[0] / max_speed;
        for (int d = 1; d < DIM; ++d) dt = std::min(dt, G.dx[d] / max_speed);
#pragma omp parallel for schedule(static) reduction(min : dt)
        for (long b = 0; b < (long)boxes.size(); ++b) {
            const Box& bx = boxes[b];
            const Fab& field4 = new_data[fs][b];
            for (int k = bx.lo[2]; k <= bx.hi[2]; ++k)
                for (int j = bx.lo[1]; j <= bx.hi[1]; ++j)
                    for (int i = bx.lo[0]; i <= bx.hi[0]; ++i) {
                        Real Bx_ = field4(i, j, k, Bx), By_ = field4(i, j, k, By), Bz_ = field4(i, j, k, Bz);
                        const Real B = std::sqrt(Bx_ * Bx_ + By_ * By_ + Bz_ * Bz_);
                        for (int n = 0; n < n_species; ++n) {
                            const int s = A.states[n];
                            Real rho = new_data[s][b](i, j, k, Density);
                            Real m, q;
                            states[s].gas.mass_charge_from_cons(new_data[s][b], i, j, k, m, q);
                            Real r = q / m;
                            const Real omega_p = 10 * std::sqrt(rho * r * r / D2) / (2 * PI);
                            if (omega_p > 0) dt = std::min(dt, 1 / omega_p);
                            const Real omega_c = 10 * (std::abs(r) * B) / (Larmor * 2 * PI);
                            if (omega_c > 0) dt = std::min(dt, 1 / omega_c);
                        }
                    }
        }
        return dt;
    }

    // MFP::estTimeStep  MFP_timestep.cpp:4-21
    Real est_time_step() const {
        if (P.force_dt > 0.0) { return P.force_dt; }  // :8
        Real estdt = std::numeric_limits<Real>::max();
        for (size_t s = 0; s < states.size(); ++s) estdt = std::min(estdt, state_allowed_dt((int)s));
        for (size_t a = 0; a < actions.size(); ++a) estdt = std::min(estdt, action_allowed_dt((int)a));
        estdt *= P.cfl;
        return estdt;
    }
};

static void setup_state(StateDef& S, const cmfp_state_desc& d, const cmfp_params& P) {
    S.desc = d;
    S.num_grow = recon_num_grow(d.reconstruction);
    if (d.kind == CMFP_STATE_HYDRO && d.viscosity) S.num_grow = std::max(S.num_grow, 2);  // MFP_hydro.cpp:108-110
    if (d.kind == CMFP_STATE_HYDRO) {
        S.gas = Gas{d.mass, d.charge, d.gamma, d.n_tracers};
        for (int i = 0; i <= d.n_tracers; ++i) {  // mass_const / gamma_const MFP_thermally_perfect_gas.cpp:32-34
            S.gas.sm[i] = d.mixture ? d.species_mass[i] : d.mass;
            S.gas.sg[i] = d.mixture ? d.species_gamma[i] : d.gamma;
            S.gas.sq[i] = d.mixture ? d.species_charge[i] : d.charge;
            if (S.gas.sq[i] != S.gas.sq[0]) S.gas.charge_const = false;
            if (S.gas.sm[i] != S.gas.sm[0]) S.gas.mass_const = false;
            if (S.gas.sg[i] != S.gas.sg[0]) S.gas.gamma_const = false;
        }
        S.n_cons = HC_NUM + d.n_tracers;  // MFP_hydro.H n_cons() / n_prim()
        S.n_prim = HP_NUM + d.n_tracers;
        S.vec_prim[0] = Xvel;  // MFP_hydro.cpp:32-33
        S.vec_cons[0] = Xmom;
        S.n_vec = 1;
        S.div_speed = 0;
        S.fastest_speed = 0;
    } else {
        S.gas = Gas{1, 0, 1};
        S.n_cons = S.n_prim = FC_NUM;
        S.vec_prim[0] = S.vec_cons[0] = Bx;  // MFP_field.cpp:33-34
        S.vec_prim[1] = S.vec_cons[1] = Dx;
        S.n_vec = 2;
        S.div_speed = d.div_transport * P.lightspeed;  // MFP_field.cpp:382-386
        S.fastest_speed = std::max(P.lightspeed, S.div_speed);
    }
}

}  // namespace orc

// ===========================================================================
// C interface for the tests (ctypes)
using namespace orc;

extern "C" {

int orc_error_flag() { return g_error; }
void orc_clear_error() { g_error = 0; }
int orc_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// --- leaves
int orc_recon_stencil_length(int r) { return recon_stencil_length(r); }
int orc_recon_num_grow(int r) { return recon_num_grow(r); }
void orc_reconstruct(int r, long n, const double* stencil, double* slope, double* lo, double* hi) {
    const int sl = recon_stencil_length(r);
    double st[7];
    for (long i = 0; i < n; ++i) {
        for (int s = 0; s < sl; ++s) st[s] = stencil[s * n + i];
        slope[i] = recon_slope(r, st);
        double l = 0, h = 0;
        recon_face_values(r, st, l, h);
        lo[i] = l;
        hi[i] = h;
    }
}
void orc_riemann(int flux, long n, const double* L, const double* R, double* F, double c0, double ch,
                 double shock_threshold) {
    const int np = (flux == CMFP_FLUX_FIELD_RH) ? 10 : 8;
    const int nf = (flux == CMFP_FLUX_FIELD_RH) ? 10 : 5;
    double l[10], r[10], f[10];
    for (long i = 0; i < n; ++i) {
        for (int c = 0; c < np; ++c) {
            l[c] = L[c * n + i];
            r[c] = R[c * n + i];
        }
        switch (flux) {
        case CMFP_FLUX_HLLC: hllc_solve(l, r, f); break;
        case CMFP_FLUX_HLLE: hlle_solve(l, r, f); break;
        case CMFP_FLUX_HYBRID_HLL: hybrid_solve(l, r, f, shock_detector(l, r, shock_threshold)); break;
        case CMFP_FLUX_FIELD_RH: field_rh_solve(c0, ch, l, r, f); break;
        }
        for (int c = 0; c < nf; ++c) F[c * n + i] = f[c];
    }
}
void orc_cons2prim(long n, double mass, double gamma, const double* U, double* Q, double* speeds) {
    Gas G{mass, 0, gamma};
    double u[5], q[8], s[3];
    for (long i = 0; i < n; ++i) {
        for (int c = 0; c < 5; ++c) u[c] = U[c * n + i];
        gas_cons2prim(G, u, q);
        gas_speed_from_cons(G, u, s);
        for (int c = 0; c < 8; ++c) Q[c * n + i] = q[c];
        for (int c = 0; c < 3; ++c) speeds[c * n + i] = s[c];
    }
}
void orc_prim2cons(long n, double mass, double gamma, const double* Q, double* U) {
    Gas G{mass, 0, gamma};
    double u[5], q[8];
    for (long i = 0; i < n; ++i) {
        for (int c = 0; c < 8; ++c) q[c] = Q[c * n + i];
        gas_prim2cons(G, q, u);
        for (int c = 0; c < 5; ++c) U[c * n + i] = u[c];
    }
}
// dense solve used by the implicit source (row-major A)
void orc_colpiv_solve(int N, const double* A, const double* b, double* x) {
    std::vector<double> a(A, A + N * N), bb(b, b + N);
    colpiv_householder_solve(N, a.data(), bb.data(), x);
}

// --- level
void* orc_create(const cmfp_geometry* g, const cmfp_params* p, int n_states, const cmfp_state_desc* sd,
                 int n_actions, const cmfp_action_desc* ad, int max_grid_size) {
    Sim* S = new Sim();
    S->G.DIM = g->dim;
    for (int d = 0; d < 3; ++d) {
        bool on = d < g->dim;
        S->G.domain.lo[d] = 0;
        S->G.domain.hi[d] = on ? g->n_cell[d] - 1 : 0;
        S->G.dx[d] = on ? g->dx[d] : 1.0;
        S->G.prob_lo[d] = on ? g->prob_lo[d] : 0.0;
        S->G.periodic[d] = on ? g->is_periodic[d] : 0;
    }
    S->P = *p;
    S->nsteps = (p->time_integrator == CMFP_TI_EULER) ? 1 : 2;  // MFP_config.cpp:73-81
    S->max_grid = max_grid_size > 0 ? max_grid_size : 1 << 30;
    S->states.resize(n_states);
    for (int s = 0; s < n_states; ++s) setup_state(S->states[s], sd[s], *p);
    S->actions.resize(n_actions);
    for (int a = 0; a < n_actions; ++a) S->actions[a].desc = ad[a];
    // mixtures are restated for the transport, the viscous fluxes and the plasma5 / Lorentz sources: collisions
    // keep the constants of species 0
    for (int a = 0; a < n_actions; ++a)
        if (ad[a].kind == CMFP_ACTION_COLLISIONS_RAMBO)
            for (int n = 0; n < ad[a].n_states; ++n)
                if (sd[ad[a].states[n]].kind == CMFP_STATE_HYDRO && sd[ad[a].states[n]].mixture) flag_abort();
    // chop the domain into boxes of at most max_grid cells per side
    int nb[3], bs[3];
    for (int d = 0; d < 3; ++d) {
        int n = S->G.domain.len(d);
        nb[d] = (n + S->max_grid - 1) / S->max_grid;
        bs[d] = (n + nb[d] - 1) / nb[d];
    }
    for (int kb = 0; kb < nb[2]; ++kb)
        for (int jb = 0; jb < nb[1]; ++jb)
            for (int ib = 0; ib < nb[0]; ++ib) {
                Box b;
                int bi[3] = {ib, jb, kb};
                for (int d = 0; d < 3; ++d) {
                    b.lo[d] = bi[d] * bs[d];
                    b.hi[d] = std::min((bi[d] + 1) * bs[d] - 1, S->G.domain.hi[d]);
                }
                S->boxes.push_back(b);
            }
    S->old_data.resize(n_states);
    S->new_data.resize(n_states);
    for (int s = 0; s < n_states; ++s) {
        S->new_data[s].resize(S->boxes.size());
        for (size_t b = 0; b < S->boxes.size(); ++b) {
            S->new_data[s][b].resize(S->boxes[b], S->states[s].n_cons);
            S->new_data[s][b].setVal(0.0);
        }
        S->old_data[s] = S->new_data[s];
    }
    return S;
}
void orc_destroy(void* h) { delete (Sim*)h; }
// Dirichlet data of face (d, side) for primitive `prim` of state s (index into the reference's
// primitive vector: hydro rho,x_vel,y_vel,z_vel,p,T,gamma,cp,alpha_i; field = conserved): values at the
// ghost-cell centres (tab_box(.., false), tng ghost layers) and at the face centres (tab_box(.., true))
void orc_set_face_values(void* h, int s, int d, int side, int prim, int tng, const double* cell, const double* face) {
    Sim* S = (Sim*)h;
    StateDef& st = S->states[s];
    st.tng = tng;
    const Box cb = tab_box(S->G.domain, S->G.DIM, tng, d, side, false), fb = tab_box(S->G.domain, S->G.DIM, tng, d, side, true);
    st.tab[d][side][prim].cell.assign(cell, cell + cb.npts());
    st.tab[d][side][prim].face.assign(face, face + fb.npts());
}
// 'dynamic' component `comp` of state s: values over grow(domain, ng), i fastest (n = 0 removes the entry)
void orc_set_dynamic(void* h, int s, int comp, int ng, const double* values, long n) {
    Sim* S = (Sim*)h;
    StateDef& st = S->states[s];
    st.dyn_ng = ng;
    if (values && n > 0)
        st.dyn[comp].assign(values, values + n);
    else
        st.dyn[comp].clear();
}
int orc_num_boxes(void* h) { return (int)((Sim*)h)->boxes.size(); }

// host layout [comp][k][j][i] over the whole domain
void orc_set_state(void* h, int s, const double* data) {
    Sim* S = (Sim*)h;
    const Box& D = S->G.domain;
    long nx = D.len(0), ny = D.len(1), nz = D.len(2);
    for (size_t b = 0; b < S->boxes.size(); ++b) {
        const Box& bx = S->boxes[b];
        for (int n = 0; n < S->states[s].n_cons; ++n)
            for (int k = bx.lo[2]; k <= bx.hi[2]; ++k)
                for (int j = bx.lo[1]; j <= bx.hi[1]; ++j)
                    for (int i = bx.lo[0]; i <= bx.hi[0]; ++i)
                        S->new_data[s][b](i, j, k, n) = data[((n * nz + k) * ny + j) * nx + i];
    }
}
static void gather(const Sim* S, const std::vector<Fab>& mf, int nc, double* data) {
    const Box& D = S->G.domain;
    long nx = D.len(0), ny = D.len(1), nz = D.len(2);
    for (size_t b = 0; b < S->boxes.size(); ++b) {
        const Box& bx = S->boxes[b];
        for (int n = 0; n < nc; ++n)
            for (int k = bx.lo[2]; k <= bx.hi[2]; ++k)
                for (int j = bx.lo[1]; j <= bx.hi[1]; ++j)
                    for (int i = bx.lo[0]; i <= bx.hi[0]; ++i) data[((n * nz + k) * ny + j) * nx + i] = mf[b](i, j, k, n);
    }
}
void orc_get_state(void* h, int s, double* data) {
    Sim* S = (Sim*)h;
    gather(S, S->new_data[s], S->states[s].n_cons, data);
}
void orc_advance(void* h, double dt) { ((Sim*)h)->advance(dt); }
double orc_est_time_step(void* h) { return ((Sim*)h)->est_time_step(); }
double orc_state_allowed_dt(void* h, int s) { return ((Sim*)h)->state_allowed_dt(s); }
double orc_action_allowed_dt(void* h, int a) { return ((Sim*)h)->action_allowed_dt(a); }

// dU of one action evaluated on the current new-time data (valid cells of the
// requested state): get_data + calc_time_derivative + calc_spatial_derivative
void orc_action_dU(void* h, int a, double dt, int s, double* out) {
    Sim* S = (Sim*)h;
    std::vector<UpdateData> ud(S->states.size());
    S->action_get_data(a, ud);
    S->action_calc_time_derivative(a, ud, dt);
    S->action_calc_spatial_derivative(a, ud, dt);
    if (ud[s].dU.empty()) return;
    gather(S, ud[s].dU, S->states[s].n_cons, out);
}
// ghosted copy of the new-time data of state s after FillPatch, on the domain
// grown by num_grow: [comp][k][j][i]
void orc_fill_patch_global(void* h, int s, double* out) {
    Sim* S = (Sim*)h;
    const StateDef& St = S->states[s];
    std::vector<Box> saved = S->boxes;
    std::vector<Fab> U;
    S->fill_patch(s, S->new_data[s], U, St.num_grow);
    // assemble from the per-box ghosted copies (ghost values agree wherever boxes overlap)
    Box gb = grow(S->G.domain, St.num_grow, S->G.DIM);
    long nx = gb.len(0), ny = gb.len(1), nz = gb.len(2);
    for (size_t b = 0; b < S->boxes.size(); ++b) {
        const Box& ub = U[b].b;
        for (int n = 0; n < St.n_cons; ++n)
            for (int k = ub.lo[2]; k <= ub.hi[2]; ++k)
                for (int j = ub.lo[1]; j <= ub.hi[1]; ++j)
                    for (int i = ub.lo[0]; i <= ub.hi[0]; ++i)
                        out[((n * nz + (k - gb.lo[2])) * ny + (j - gb.lo[1])) * nx + (i - gb.lo[0])] = U[b](i, j, k, n);
    }
}

}  // extern "C"
