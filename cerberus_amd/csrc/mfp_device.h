// mfp_device.h — per-cell / per-face device numerics of the hyperbolic update.
//
// Each function states the reference routine whose arithmetic it reproduces
// (file:line in PlasmaSimUQ/cerberus Source/).  Expression order follows the
// reference so that the STRICT build (-ffp-contract=off) is bit-comparable with
// a host build of the same formulas; the FAST build differs only by FMA
// contraction.  Hydro face/cell primitive vectors carry {rho,u,v,w,p}: for the
// single-species gas T, gamma and cp never feed back into conserved results
// (gamma is the state's constant, MFP_thermally_perfect_gas.cpp:55).
#pragma once
#include <hip/hip_runtime.h>

#include "mfp_types.h"

#define CMFP_DEV __device__ __forceinline__

namespace CMFP_NS {

enum { H_RHO = 0, H_U = 1, H_V = 2, H_W = 3, H_P = 4, H_N = 5 };
enum { F_DX = 0, F_DY, F_DZ, F_BX, F_BY, F_BZ, F_PHI, F_PSI, F_MU, F_EP, F_N };

// ---------------------------------------------------------------- utilities
// sgn / sign: MFP_utility.H:149-158
CMFP_DEV int sgn_i(double v) { return (0.0 < v) - (v < 0.0); }
CMFP_DEV double sign_ab(double A, double B) { return (B >= 0.0) ? fabs(A) : -fabs(A); }
// tanh_approx: MFP_utility.H:289-298
CMFP_DEV double tanh_approx(double x) {
    if (x > 1.92033) return 0.96016;
    if (x > 0) return 0.96016 - 0.26037 * (x - 1.92033) * (x - 1.92033);
    if (x <= -1.92033) return -0.96016;
    return 0.26037 * (x + 1.92033) * (x + 1.92033) - 0.96016;
}
// Division / square root.  STRICT: the IEEE operations, as the reference's host build.  FAST
// (-DCMFP_FASTMATH): hardware seed (v_rcp_f64 / v_rsq_f64, 2^-24 relative) + two Newton steps
// for the reciprocal, one coupled step + a residual correction for the root: measured <= 1.2e-16
// relative on 4 M random arguments (tools/fastmath_probe.hip); 5 and 8 instructions instead of
// the compiler's 9 and ~20 (range scaling for denormal arguments that this path never sees).
#ifdef CMFP_FASTMATH
CMFP_DEV double f_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(e, r, r);
    e = fma(-x, r, 1.0);
    return fma(e, r, r);
}
CMFP_DEV double f_div(double a, double b) { return a * f_rcp(b); }
CMFP_DEV double f_sqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    return (x == 0.0) ? 0.0 : g;
}
// 1 / sqrt(x): seed + one step with the quadratic term (e = 1 - x y^2 ~ 2^-23, e^3 is below an ulp)
CMFP_DEV double f_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-(x * y), y, 1.0);
    double t = fma(0.375, e, 0.5);
    return fma(y * e, t, y);
}
#else
CMFP_DEV double f_rsqrt(double x) { return 1 / sqrt(x); }
CMFP_DEV double f_rcp(double x) { return 1 / x; }
CMFP_DEV double f_div(double a, double b) { return a / b; }
CMFP_DEV double f_sqrt(double x) { return sqrt(x); }
#endif
// Time step of a launch: the `dt` argument, or, with a device-resident time step (Geo::dtp,
// cmfp_advance_steps), that argument as a factor of *dtp.  The product is wave-uniform; it is moved
// to scalar registers so that it costs no vector registers (the tiled field kernel sits exactly at
// its 168-VGPR budget).
// a wave-uniform value moved to scalar registers
CMFP_DEV double uni(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
CMFP_DEV double launch_dt(const cmfp::Geo& g, double dt) {
    if (!g.dtp) return dt;
    return uni(dt * *g.dtp);
}

// std::min / std::max semantics (return the first argument on ties / NaN)
CMFP_DEV double std_min(double a, double b) { return (b < a) ? b : a; }
CMFP_DEV double std_max(double a, double b) { return (a < b) ? b : a; }

// ------------------------------------------------------------ reconstruction
// Reconstruction family, MFP_reconstruction.cpp.  RECON is a cmfp_recon value.
template <int RECON> struct ReconTraits;
template <> struct ReconTraits<0> { static constexpr int SL = 1, NG = 1; };  // constant :59-60
template <> struct ReconTraits<1> { static constexpr int SL = 3, NG = 2; };  // minmod   :81-82
template <> struct ReconTraits<2> { static constexpr int SL = 3, NG = 2; };  // vanLeer  :123-124
template <> struct ReconTraits<3> { static constexpr int SL = 3, NG = 2; };  // MC       :163-164
template <> struct ReconTraits<4> { static constexpr int SL = 3, NG = 2; };  // centre   :216-217
template <> struct ReconTraits<5> { static constexpr int SL = 7, NG = 4; };  // O6       :247-248
template <> struct ReconTraits<6> { static constexpr int SL = 7, NG = 4; };  // WENO     :283-284

// WENOReconstruction::WENOSYMBO :297-357.  UP = +1 / -1.
template <int UP> CMFP_DEV double wenosymbo(const double* st) {
    constexpr int i = 3, r = 3;
    const double C[4] = {0.094647545896, 0.428074212384, 0.408289331408, 0.068988910311};
    const double akl[4][3] = {{2 / 6.0, -7 / 6.0, 11 / 6.0},
                              {-1 / 6.0, 5 / 6.0, 2 / 6.0},
                              {2 / 6.0, 5 / 6.0, -1 / 6.0},
                              {11 / 6.0, -7 / 6.0, 2 / 6.0}};
    const double b2 = 1.0408329997330663;  // std::sqrt(13 / 12.0), correctly rounded
    const double dkml[4][2][3] = {{{1 / 2.0, -4 / 2.0, 3 / 2.0}, {b2, -2 * b2, b2}},
                                  {{-1 / 2.0, 0, 1 / 2.0}, {b2, -2 * b2, b2}},
                                  {{-3 / 2.0, 4 / 2.0, -1 / 2.0}, {b2, -2 * b2, b2}},
                                  {{-5 / 2.0, 8 / 2.0, -3 / 2.0}, {b2, -2 * b2, b2}}};
    const double epsilon = 1e-10;
    double IS[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k <= r; ++k) {
#pragma unroll
        for (int m = 0; m < r - 1; ++m) {
            double is = 0.0;
#pragma unroll
            for (int l = 0; l < r; ++l) { is += dkml[k][m][l] * st[i - UP * (r - 1 - k - l)]; }
            IS[k] += is * is;
        }
    }
    double max_IS = IS[0];
#pragma unroll
    for (int k = 1; k <= r; ++k) { max_IS = std_max(IS[k], max_IS); }
    IS[r] = max_IS;
    double alpha[4];
#pragma unroll
    for (int k = 0; k <= r; ++k) { alpha[k] = C[k] / (epsilon + IS[k]); }
    double sum_alpha = alpha[0] + alpha[1] + alpha[2];
    double omega[4];
#pragma unroll
    for (int k = 0; k <= r; ++k) { omega[k] = alpha[k] / sum_alpha; }
    double face_value = 0.0;
#pragma unroll
    for (int k = 0; k < r; ++k) {
        double q = 0.0;
#pragma unroll
        for (int l = 0; l < r; ++l) { q += akl[k][l] * st[i - UP * (r - 1 - k - l)]; }
        face_value += omega[k] * q;
    }
    return face_value;
}

// get_face_values of each class (:63,103,143,196,227,261,359) and get_slope
// (:85,127,167,220,251,287; base :17).
template <int RECON> CMFP_DEV double recon_slope(const double* st);
template <int RECON> CMFP_DEV void recon_faces(const double* st, double& lo, double& hi);

template <> CMFP_DEV double recon_slope<0>(const double*) { return 0; }
template <> CMFP_DEV double recon_slope<1>(const double* st) {
    double a = st[1] - st[0];
    double b = st[2] - st[1];
#ifdef CMFP_FASTMATH
    // the same function without the reference's comparisons and selects: max(0, min(a, b)) + min(0, max(a, b))
    // (only the sign of a zero result can differ)
    return fmax(0.0, fmin(a, b)) + fmin(0.0, fmax(a, b));
#endif
    double g = 0.0;
    if ((a >= 0.0) && (b >= 0.0)) {
        g = std_min(a, b);
    } else if ((a <= 0.0) && (b <= 0.0)) {
        g = std_max(a, b);
    }
    return g;
}
template <> CMFP_DEV double recon_slope<2>(const double* st) {
    double a = st[1] - st[0];
    double b = st[2] - st[1];
    double sa = sgn_i(a);
    double sb = sgn_i(b);
    double aa = fabs(a);
    double bb = fabs(b);
    return (sa + sb) * ((aa * bb) / (aa + bb + 1e-20));
}
template <> CMFP_DEV double recon_slope<3>(const double* st) {
    double x1 = st[0], x2 = st[1], x3 = st[2];
    double dqm = (x2 - x1);
    double dqp = (x3 - x2);
    double dqc = 0.5 * (x3 - x1);
    double s = dqm * dqp;
    if (s <= 0.0) {
        s = 0.0;
    } else {
        if ((fabs(dqm) < fabs(dqp)) && (fabs(dqm) < fabs(dqc))) {
            s = dqm;
        } else if (fabs(dqp) < fabs(dqc)) {
            s = dqp;
        } else {
            s = dqc;
        }
    }
    return s;
}
template <> CMFP_DEV double recon_slope<4>(const double* st) { return 0.5 * (st[2] - st[0]); }

template <> CMFP_DEV void recon_faces<0>(const double* st, double& lo, double& hi) {
    lo = st[0];
    hi = st[0];
}
#define CMFP_SLOPE_FACES(R)                                                              \
    template <> CMFP_DEV void recon_faces<R>(const double* st, double& lo, double& hi) { \
        double g = recon_slope<R>(st);                                                   \
        lo = st[1] - 0.5 * g;                                                            \
        hi = st[1] + 0.5 * g;                                                            \
    }
CMFP_SLOPE_FACES(1)
CMFP_SLOPE_FACES(2)
CMFP_SLOPE_FACES(3)
CMFP_SLOPE_FACES(4)
#undef CMFP_SLOPE_FACES
template <> CMFP_DEV void recon_faces<5>(const double* st, double& lo, double& hi) {
    hi = (37.0 / 60.0) * (st[3] + st[4]) - (2.0 / 15.0) * (st[2] + st[5]) + (1.0 / 60.0) * (st[1] + st[6]);
    lo = (37.0 / 60.0) * (st[3] + st[2]) - (2.0 / 15.0) * (st[4] + st[1]) + (1.0 / 60.0) * (st[5] + st[0]);
}
template <> CMFP_DEV void recon_faces<6>(const double* st, double& lo, double& hi) {
    hi = wenosymbo<1>(st);
    lo = wenosymbo<-1>(st);
}
template <> CMFP_DEV double recon_slope<5>(const double* st) {
    double lo, hi;
    recon_faces<5>(st, lo, hi);
    return hi - lo;
}
template <> CMFP_DEV double recon_slope<6>(const double* st) {
    double lo, hi;
    recon_faces<6>(st, lo, hi);
    return hi - lo;
}

// ------------------------------------------------------------------- gas
using cmfp::GasP;

// ThermallyPerfectGas::cons2prim MFP_thermally_perfect_gas.cpp:180-218 and
// HydroGas::prim_valid MFP_hydro_gas.cpp:152-164 (returns false where the
// reference aborts).
CMFP_DEV bool hydro_cons2prim(const GasP& G, const double* U, double* Q) {
    double rho = U[0], mx = U[1], my = U[2], mz = U[3], ed = U[4];
    double rhoinv = f_rcp(rho);
    double u = mx * rhoinv;
    double v = my * rhoinv;
    double w = mz * rhoinv;
    double ke = 0.5 * rho * (u * u + v * v + w * w);
    double g = G.gamma;
    double p = (ed - ke) * (g - 1);
    Q[H_RHO] = rho;
    Q[H_U] = u;
    Q[H_V] = v;
    Q[H_W] = w;
    Q[H_P] = p;
#ifdef CMFP_FASTMATH
    return !((rho <= 0.0) || (p <= 0.0));  // T = p / rho * mass: its sign test adds nothing for mass > 0
#else
    double T = p * rhoinv * G.mass;
    return !((rho <= 0.0) || (p <= 0.0) || (T <= 0.0));
#endif
}
// ThermallyPerfectGas::prim2cons :220-246
CMFP_DEV void hydro_prim2cons(const GasP& G, const double* Q, double* U) {
    double rho = Q[H_RHO], u = Q[H_U], v = Q[H_V], w = Q[H_W], p = Q[H_P];
    double mx = u * rho;
    double my = v * rho;
    double mz = w * rho;
    double ke = 0.5 * rho * (u * u + v * v + w * w);
    double g = G.gamma;
    double ed = f_div(p, g - 1) + ke;
    U[0] = rho;
    U[1] = mx;
    U[2] = my;
    U[3] = mz;
    U[4] = ed;
}
// ThermallyPerfectGas::get_speed_from_cons :294-317
CMFP_DEV void hydro_speed_from_cons(const GasP& G, const double* U, double* s) {
    double rho = U[0], mx = U[1], my = U[2], mz = U[3], ed = U[4];
    double rhoinv = f_rcp(rho);
    double u = mx * rhoinv;
    double v = my * rhoinv;
    double w = mz * rhoinv;
    double ke = 0.5 * rho * (u * u + v * v + w * w);
    double g = G.gamma;
    double p = (ed - ke) * (g - 1);
    double a = f_sqrt(g * p * rhoinv);
    s[0] = a + fabs(u);
    s[1] = a + fabs(v);
    s[2] = a + fabs(w);
}
// ThermallyPerfectGas::get_speed_from_prim :319-332
CMFP_DEV void hydro_speed_from_prim(const GasP& G, const double* Q, double* s) {
#ifdef CMFP_FASTMATH
    const double a2 = f_div(G.gamma * Q[H_P], Q[H_RHO]);
    double a = a2 * f_rsqrt(a2);  // sqrt(x) = x rsqrt(x): valid states have x > 0
#else
    double a = f_sqrt(f_div(G.gamma * Q[H_P], Q[H_RHO]));
#endif
    s[0] = a + fabs(Q[H_U]);
    s[1] = a + fabs(Q[H_V]);
    s[2] = a + fabs(Q[H_W]);
}

// Mixture of nt + 1 species (HydroParams::mix): the gas constants of a cell from its nt mass fractions, the last
// species taking 1 - sum(alpha).  get_gamma_from_prim MFP_thermally_perfect_gas.cpp:51-83 (gamma_const returns
// gamma[0]), get_mass_from_prim MFP_hydro_gas.cpp:50-70 (mass_const returns mass[0]); alpha clamped to [0, 1].
CMFP_DEV double mix_clamp01(double a) { return (a < 0.0) ? 0.0 : ((1.0 < a) ? 1.0 : a); }
CMFP_DEV GasP mix_gas(const cmfp::HydroParams& hp, const double* alpha) {
    GasP G = hp.gas;
    double a0 = 0.0, a1 = 0.0;
    if (hp.nt > 0) a0 = mix_clamp01(alpha[0]);
    if (hp.nt > 1) a1 = mix_clamp01(alpha[1]);
    if (hp.mix & 1) {
        double S_a = 0.0, S_cp = 0.0, S_cv = 0.0;
        if (hp.nt > 0) {
            S_a += a0;
            S_cp += a0 * hp.sp_cp[0];
            S_cv += a0 * hp.sp_cv[0];
        }
        if (hp.nt > 1) {
            S_a += a1;
            S_cp += a1 * hp.sp_cp[1];
            S_cv += a1 * hp.sp_cv[1];
        }
        const double cpl = hp.nt > 1 ? hp.sp_cp[2] : hp.sp_cp[1], cvl = hp.nt > 1 ? hp.sp_cv[2] : hp.sp_cv[1];
        S_cp += (1.0 - S_a) * cpl;
        S_cv += (1.0 - S_a) * cvl;
        G.gamma = f_div(S_cp, S_cv);
    }
    if (hp.mix & 2) {
        double S_a = 0.0, S_m = 0.0;
        if (hp.nt > 0) {
            S_a += a0;
            S_m += f_div(a0, hp.sp_mass[0]);
        }
        if (hp.nt > 1) {
            S_a += a1;
            S_m += f_div(a1, hp.sp_mass[1]);
        }
        S_m += f_div(1.0 - S_a, hp.nt > 1 ? hp.sp_mass[2] : hp.sp_mass[1]);
        G.mass = f_div(1.0, S_m);
    }
    return G;
}

// get_cp_from_prim MFP_thermally_perfect_gas.cpp:123-147 (hp.mix != 0: the species differ in gamma or mass)
CMFP_DEV double mix_cp(const cmfp::HydroParams& hp, const double* alpha) {
    double S_a = 0.0, S_cp = 0.0;
    if (hp.nt > 0) {
        const double a0 = mix_clamp01(alpha[0]);
        S_a += a0;
        S_cp += a0 * hp.sp_cp[0];
    }
    if (hp.nt > 1) {
        const double a1 = mix_clamp01(alpha[1]);
        S_a += a1;
        S_cp += a1 * hp.sp_cp[1];
    }
    S_cp += (1.0 - S_a) * (hp.nt > 1 ? hp.sp_cp[2] : hp.sp_cp[1]);
    return S_cp;
}

// ------------------------------------------------------------ Riemann: hydro
// What the passive-scalar (tracer) fluxes need from a hydro Riemann solve: HLLC (MFP_hllc.cpp:100,
// 111-137,147-173,183): side K the flux was taken from, its normal velocity and density, and in a
// star region S_K and (S_K - u_K) / (S_K - S_star); HLLE (MFP_hydro_hlle.cpp:75-118): the branch,
// sL, sR and the normal velocities and densities of both sides.
struct HllAux {
    int branch;      // HLLC: 0 pure state, 1 star state.  HLLE: 0 left, 1 middle, 2 right
    bool useL;       // HLLC: the state comes from the left side
    double S_K, ratio;
    double sL, sR;
};

// HydroHLLC::solve MFP_hllc.cpp:30-185.  L, R in the face-normal frame.
CMFP_DEV void hllc_solve(const double* L, const double* R, double gamL, double gamR, double* F, HllAux* aux = nullptr) {
    double rhoL = L[H_RHO], uL = L[H_U], vL = L[H_V], wL = L[H_W], pL = L[H_P];
    double rhoR = R[H_RHO], uR = R[H_U], vR = R[H_V], wR = R[H_W], pR = R[H_P];
#ifdef CMFP_FASTMATH
    // one reciprocal root per side: y = 1 / sqrt(gamma p rho) gives a = gamma p y, 1 / rho = a y, a^2 = a a
    // (valid states have gamma p rho > 0); 1 / rho is reused for nrg / rho of the chosen side below
    const double gpL = gamL * pL, gpR = gamR * pR;
    const double yL = f_rsqrt(gpL * rhoL), yR = f_rsqrt(gpR * rhoR);
    double aL = gpL * yL, aR = gpR * yR;
    const double rinvL = aL * yL, rinvR = aR * yR;
    const double a2L = aL * aL, a2R = aR * aR;
#else
    double aL = f_sqrt(f_div(gamL * pL, rhoL));
    double aR = f_sqrt(f_div(gamR * pR, rhoR));
#endif

    double rho_bar = 0.5 * (rhoL + rhoR);
    double a_bar = 0.5 * (aL + aR);
    double p_star = 0.5 * (pL + pR) - 0.5 * (uR - uL) * rho_bar * a_bar;
    double S_star = 0.5 * (uL + uR) - f_div(0.5 * (pR - pL), rho_bar * a_bar);

#ifdef CMFP_FASTMATH
    // a_K q_K of :69-87 as one root without the division by p_K and without a branch:
    //   a_K^2 q_K^2 = a_K^2 + (gamma_K + 1) / (2 rho_K) * max(p* - p_K, 0)      (a_K^2 / p_K = gamma_K / rho_K)
    const double aq2L = fma(0.5 * (gamL + 1.0) * rinvL, fmax(p_star - pL, 0.0), a2L);
    const double aq2R = fma(0.5 * (gamR + 1.0) * rinvR, fmax(p_star - pR, 0.0), a2R);
    double S_L = uL - aq2L * f_rsqrt(aq2L);
    double S_R = uR + aq2R * f_rsqrt(aq2R);
#else
    double qL;
    if (p_star <= pL) {
        qL = 1.0;
    } else {
        qL = f_sqrt(1.0 + f_div(gamL + 1.0, 2 * gamL) * (f_div(p_star, pL) - 1.0));
    }
    double S_L = uL - aL * qL;
    double qR;
    if (p_star <= pR) {
        qR = 1.0;
    } else {
        qR = f_sqrt(1.0 + f_div(gamR + 1.0, 2 * gamR) * (f_div(p_star, pR) - 1.0));
    }
    double S_R = uR + aR * qR;
#endif

    // The reference's four-way branch (MFP_hllc.cpp:92-184) written as a side
    // selection: every lane evaluates the same expressions on the state of the
    // side the branch would have used, so results are identical while a wave
    // whose lanes fall in different branches runs one body instead of four.
    const bool pureL = (S_L >= 0.0);
    const bool starL = !pureL && (S_L <= 0.0) && (0.0 <= S_star);
    const bool starR = !pureL && !starL && (S_star <= 0.0) && (0.0 <= S_R);
    const bool useL = pureL || starL;
    const double rho = useL ? rhoL : rhoR, u = useL ? uL : uR, v = useL ? vL : vR, w = useL ? wL : wR;
    const double p = useL ? pL : pR, S_K = useL ? S_L : S_R;
    // nrg_K of :44,60 for the side in use only: the same expression on the same operands as the reference's
    const double nrg = f_div(p, (useL ? gamL : gamR) - 1) + 0.5 * rho * (u * u + v * v + w * w);
    const double fv0 = rho * u;
    const double fv1 = rho * u * u + p;
    const double fv2 = rho * u * v;
    const double fv3 = rho * u * w;
    const double fv4 = u * (nrg + p);
    if (starL || starR) {
        const double sv0 = rho, sv1 = rho * u, sv2 = rho * v, sv3 = rho * w, sv4 = nrg;
#ifdef CMFP_FASTMATH
        // 1 / (S_K - S*) and 1 / (rho (S_K - u)) from one reciprocal of their product
        const double dSs = S_K - S_star, rdu = rho * (S_K - u);
        const double rprod = f_rcp(dSs * rdu);
        const double ratio = (S_K - u) * (rprod * rdu);
        const double p_rdu = p * (rprod * dSs);
#else
        const double ratio = f_div(S_K - u, S_K - S_star);
#endif
        const double coeff = rho * ratio;
        if (aux) {
            aux->branch = 1;
            aux->useL = useL;
            aux->S_K = S_K;
            aux->ratio = ratio;
        }
        const double ss0 = coeff;
        const double ss1 = coeff * S_star;
        const double ss2 = coeff * v;
        const double ss3 = coeff * w;
#ifdef CMFP_FASTMATH
        const double nrg_rho = nrg * (useL ? rinvL : rinvR);
#else
        const double nrg_rho = f_div(nrg, rho);
#endif
#ifdef CMFP_FASTMATH
        const double ss4 = coeff * (nrg_rho + (S_star - u) * (S_star + p_rdu));
#else
        const double ss4 = coeff * (nrg_rho + (S_star - u) * (S_star + f_div(p, rho * (S_K - u))));
#endif
        F[0] = fv0 + S_K * (ss0 - sv0);
        F[1] = fv1 + S_K * (ss1 - sv1);
#ifdef CMFP_FASTMATH
        // the tangential momentum fluxes are the mass flux times the tangential velocity of the side in use:
        // rho u v + S_K (coeff v - rho v) = v (rho u + S_K (coeff - rho))
        F[2] = F[0] * v;
        F[3] = F[0] * w;
#else
        F[2] = fv2 + S_K * (ss2 - sv2);
        F[3] = fv3 + S_K * (ss3 - sv3);
#endif
        F[4] = fv4 + S_K * (ss4 - sv4);
    } else {
        if (aux) {
            aux->branch = 0;
            aux->useL = useL;
        }
        F[0] = fv0;
        F[1] = fv1;
        F[2] = fv2;
        F[3] = fv3;
        F[4] = fv4;
    }
}
// tracer flux of HydroHLLC::solve: aL, aR = alpha on the two sides (face-normal frame states L, R)
CMFP_DEV double hllc_tracer_flux(const HllAux& x, const double* L, const double* R, double aL, double aR) {
    const double rho = x.useL ? L[H_RHO] : R[H_RHO], u = x.useL ? L[H_U] : R[H_U];
    const double tr = (x.useL ? aL : aR) * rho;  // trK = alpha_K * rho_K  :46-48,62-64
    const double fv = u * tr;
    if (x.branch == 0) return fv;
    return fv + x.S_K * (tr * x.ratio - tr);
}

// HydroHLLE::solve MFP_hydro_hlle.cpp:30-120
CMFP_DEV void hlle_solve(const double* L, const double* R, double gamL, double gamR, double* F, HllAux* aux = nullptr) {
    double rhoL = L[H_RHO], uL = L[H_U], vL = L[H_V], wL = L[H_W], pL = L[H_P];
    double nrgL = f_div(pL, gamL - 1) + 0.5 * rhoL * (uL * uL + vL * vL + wL * wL);
    double aL = f_sqrt(f_div(gamL * pL, rhoL));
    double rhoR = R[H_RHO], uR = R[H_U], vR = R[H_V], wR = R[H_W], pR = R[H_P];
    double nrgR = f_div(pR, gamR - 1) + 0.5 * rhoR * (uR * uR + vR * vR + wR * wR);
    double aR = f_sqrt(f_div(gamR * pR, rhoR));

    double sL = std_min(uL - aL, uR - aR);
    double sR = std_max(uL + aL, uR + aR);
    if (aux) {
        aux->sL = sL;
        aux->sR = sR;
        aux->branch = (sL >= 0.0) ? 0 : ((sL <= 0.0) && (sR >= 0.0)) ? 1 : 2;
    }

    if (sL >= 0.0) {
        F[0] = rhoL * uL;
        F[1] = rhoL * uL * uL + pL;
        F[2] = rhoL * uL * vL;
        F[3] = rhoL * uL * wL;
        F[4] = uL * (nrgL + pL);
    } else if ((sL <= 0.0) && (sR >= 0.0)) {
        double fL[5], fR[5], sLv[5], sRv[5];
        fL[0] = rhoL * uL;
        fL[1] = rhoL * uL * uL + pL;
        fL[2] = rhoL * uL * vL;
        fL[3] = rhoL * uL * wL;
        fL[4] = uL * (nrgL + pL);
        fR[0] = rhoR * uR;
        fR[1] = rhoR * uR * uR + pR;
        fR[2] = rhoR * uR * vR;
        fR[3] = rhoR * uR * wR;
        fR[4] = uR * (nrgR + pR);
        sLv[0] = rhoL;
        sLv[1] = rhoL * uL;
        sLv[2] = rhoL * vL;
        sLv[3] = rhoL * wL;
        sLv[4] = nrgL;
        sRv[0] = rhoR;
        sRv[1] = rhoR * uR;
        sRv[2] = rhoR * vR;
        sRv[3] = rhoR * wR;
        sRv[4] = nrgR;
#pragma unroll
        for (int i = 0; i < 5; ++i) { F[i] = f_div(sR * fL[i] - sL * fR[i] + sL * sR * (sRv[i] - sLv[i]), sR - sL); }
    } else {
        F[0] = rhoR * uR;
        F[1] = rhoR * uR * uR + pR;
        F[2] = rhoR * uR * vR;
        F[3] = rhoR * uR * wR;
        F[4] = uR * (nrgR + pR);
    }
}

// tracer flux of HydroHLLE::solve
CMFP_DEV double hlle_tracer_flux(const HllAux& x, const double* L, const double* R, double aL, double aR) {
    const double trL = aL * L[H_RHO], trR = aR * R[H_RHO];
    if (x.branch == 0) return L[H_U] * trL;
    if (x.branch == 2) return R[H_U] * trR;
    const double fl = L[H_U] * trL, fr = R[H_U] * trR;
    return f_div(x.sR * fl - x.sL * fr + x.sL * x.sR * (trR - trL), x.sR - x.sL);
}

// PressureJumpShockDetector::solve MFP_hydro_shockdetector.cpp:35-44
CMFP_DEV double shock_detect(const double* L, const double* R, double thr) {
    double pL = L[H_P], pR = R[H_P];
    double varphi = fabs(pR - pL) / (pR + pL);
    return 0.5 + 0.5 * tanh_approx(5 * (varphi - 0.75 * thr) / thr);
}

// HydroHybridHLL::solve MFP_hybrid_hll.cpp:30-47
CMFP_DEV void hybrid_solve(const double* L, const double* R, double gamL, double gamR, double shk, double* F) {
    const double eps = 1e-14;
    if (shk < eps) {
        hllc_solve(L, R, gamL, gamR, F);
    } else if ((1.0 - shk) < eps) {
        hlle_solve(L, R, gamL, gamR, F);
    } else {
        double Fc[5], Fe[5];
        hllc_solve(L, R, gamL, gamR, Fc);
        hlle_solve(L, R, gamL, gamR, Fe);
#pragma unroll
        for (int i = 0; i < 5; ++i) { F[i] = (1.0 - shk) * Fc[i] + shk * Fe[i]; }
    }
}

// FLUX: cmfp_flux (0 HLLC, 1 HLLE, 2 hybrid).  shock_threshold <= 0: shk = 0
// (EulerianState::calc_fluxes MFP_eulerian.cpp:1024-1028).
template <int FLUX> CMFP_DEV void hydro_riemann(const double* L, const double* R, double gam, double thr, double* F) {
    if (FLUX == 0) {
        hllc_solve(L, R, gam, gam, F);
    } else if (FLUX == 1) {
        hlle_solve(L, R, gam, gam, F);
    } else {
        double shk = 0.0;
        if (thr > 0) shk = shock_detect(L, R, thr);
        hybrid_solve(L, R, gam, gam, shk, F);
    }
}

// hydro face flux incl. nt passive scalars: L, R carry alpha behind the five primitives, F the tracer
// fluxes behind the five conserved fluxes (HLLC / HLLE; the hybrid solver leaves tracer fluxes
// undefined in the reference, quirk Q5, and is refused with tracers)
template <int FLUX>
CMFP_DEV void hydro_riemann_tracers(const double* L, const double* R, double gamL, double gamR, int nt, double* F) {
    HllAux x;
    if (FLUX == 0) {
        hllc_solve(L, R, gamL, gamR, F, &x);
        for (int t = 0; t < nt; ++t) F[5 + t] = hllc_tracer_flux(x, L, R, L[5 + t], R[5 + t]);
    } else {
        hlle_solve(L, R, gamL, gamR, F, &x);
        for (int t = 0; t < nt; ++t) F[5 + t] = hlle_tracer_flux(x, L, R, L[5 + t], R[5 + t]);
    }
}

// ------------------------------------------------------------ Riemann: field
// FieldRH::solve MFP_field_RH.cpp:29-106 (face-normal frame; c0 = lightspeed,
// ch = div_speed).
CMFP_DEV void field_rh_solve(double c0, double ch, const double* L, const double* R, double* F) {
    const double ch2 = ch * ch;
#pragma unroll
    for (int n = 0; n < F_N; ++n) F[n] = 0;
    double Dl = L[F_DY], Dr = R[F_DY];
    double epr = R[F_EP], epl = L[F_EP];
    double Bl = L[F_BZ], Br = R[F_BZ];
    double mur = R[F_MU], mul = L[F_MU];
    if (mul == 1.0 && epl == 1.0 && mur == 1.0 && epr == 1.0) {
        // vacuum on both sides: the general expressions below evaluate to exactly these values
        // (c = 1/sqrt(1) = 1, x/1 = x, division by 2 is exact), with no division or root
        const double hc = 0.5 * c0;
        F[F_DY] = (c0 * ((Bl + Br) + (Dl - Dr))) * 0.5;
        F[F_BZ] = (c0 * ((Dl + Dr) + (Bl - Br))) * 0.5;
        Dl = L[F_DZ];
        Dr = R[F_DZ];
        Bl = L[F_BY];
        Br = R[F_BY];
        F[F_DZ] = (c0 * (-(Bl + Br) + (Dl - Dr))) * 0.5;
        F[F_BY] = (c0 * (-(Dl + Dr) + (Bl - Br))) * 0.5;
        if (ch > 0) {
            double Pl = L[F_PHI], Pr = R[F_PHI], fl = L[F_DX], fr = R[F_DX];
            F[F_DX] = hc * (Pr + Pl) - 0.5 * ch * (fr - fl);
            F[F_PHI] = 0.5 * ch2 / c0 * (fr + fl) - 0.5 * ch * (Pr - Pl);
            Pl = L[F_PSI], Pr = R[F_PSI], fl = L[F_BX], fr = R[F_BX];
            F[F_BX] = hc * (Pr + Pl) - 0.5 * ch * (fr - fl);
            F[F_PSI] = 0.5 * ch2 / c0 * (fr + fl) - 0.5 * ch * (Pr - Pl);
        }
        return;
    }
    double cr = f_rsqrt(mur * epr);
    double cl = f_rsqrt(mul * epl);
    F[F_DY] = f_div(c0 * ((Bl * cl + Br * cr) + (f_div(Dl, epl) - f_div(Dr, epr))), cl * mul + cr * mur);
    F[F_BZ] = f_div(c0 * ((Dl * cl + Dr * cr) + (f_div(Bl, mul) - f_div(Br, mur))), cl * epl + cr * epr);
    if (ch > 0) {
        double Pl = L[F_PHI], Pr = R[F_PHI], fl = L[F_DX], fr = R[F_DX];
        F[F_DX] = 0.5 * c0 * (Pr + Pl) - 0.5 * ch * (fr - fl);
        F[F_PHI] = 0.5 * ch2 / c0 * (fr + fl) - 0.5 * ch * (Pr - Pl);
    }
    Dl = L[F_DZ];
    Dr = R[F_DZ];
    Bl = L[F_BY];
    Br = R[F_BY];
    F[F_DZ] = f_div(c0 * (-(Bl * cl + Br * cr) + (f_div(Dl, epl) - f_div(Dr, epr))), cl * mul + cr * mur);
    F[F_BY] = f_div(c0 * (-(Dl * cl + Dr * cr) + (f_div(Bl, mul) - f_div(Br, mur))), cl * epl + cr * epr);
    if (ch > 0) {
        double Pl = L[F_PSI], Pr = R[F_PSI], fl = L[F_BX], fr = R[F_BX];
        F[F_BX] = 0.5 * c0 * (Pr + Pl) - 0.5 * ch * (fr - fl);
        F[F_PSI] = 0.5 * ch2 / c0 * (fr + fl) - 0.5 * ch * (Pr - Pl);
    }
}

// ------------------------------------------------------------- frame rotation
// transform_global2local / local2global, modules/MFP_transforms.H:11-55:
// cyclic permutation of the 3-vector starting at v.
CMFP_DEV void rot_g2l(double* U, int v, int d) {
    if (d == 0) return;
    double a = U[v], b = U[v + 1], c = U[v + 2];
    if (d == 1) {  // n = y, t1 = z, t2 = x
        U[v] = b;
        U[v + 1] = c;
        U[v + 2] = a;
    } else {  // n = z, t1 = x, t2 = y
        U[v] = c;
        U[v + 1] = a;
        U[v + 2] = b;
    }
}
CMFP_DEV void rot_l2g(double* U, int v, int d) {
    if (d == 0) return;
    double n = U[v], t1 = U[v + 1], t2 = U[v + 2];
    if (d == 1) {  // U[y] = n, U[z] = t1, U[x] = t2
        U[v + 1] = n;
        U[v + 2] = t1;
        U[v] = t2;
    } else {  // U[z] = n, U[x] = t1, U[y] = t2
        U[v + 2] = n;
        U[v] = t1;
        U[v + 1] = t2;
    }
}

// ------------------------------------------------------------------ physics
// Compile-time description of a state kind as the update kernels see it.
using cmfp::FieldParams;
using cmfp::HydroParams;

template <int RECON_, int FLUX_> struct HydroPhys {
    static constexpr int NC = 5;  // conserved comps
    static constexpr int NQ = 5;  // primitive comps carried
    static constexpr int NF = 5;  // flux comps that can be non-zero
    static constexpr int RECON = RECON_;
    static constexpr int FLUX = FLUX_;
    static constexpr bool IS_HYDRO = true;
    typedef HydroParams Params;

    static CMFP_DEV bool cons2prim(const Params& P, const double* U, double* Q) {
        return hydro_cons2prim(P.gas, U, Q);
    }
    static CMFP_DEV void prim2cons(const Params& P, const double* Q, double* U) { hydro_prim2cons(P.gas, Q, U); }
    static CMFP_DEV void speed_from_prim(const Params& P, const double* Q, double* s) {
        hydro_speed_from_prim(P.gas, Q, s);
    }
    // face flux in direction d from global-frame face states
    static CMFP_DEV void riemann(const Params& P, int d, const double* Lg, const double* Rg, double* F) {
        double L[5], R[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            L[n] = Lg[n];
            R[n] = Rg[n];
        }
        rot_g2l(L, H_U, d);
        rot_g2l(R, H_U, d);
        hydro_riemann<FLUX_>(L, R, P.gas.gamma, P.shock_threshold, F);
        rot_l2g(F, 1, d);
    }
    // the same with NT passive scalars: aL, aR the mass fractions of the two face states, Ft their fluxes (the hybrid
    // solver leaves them undefined in the reference, quirk Q5: refused with tracers by cmfp_create)
    template <int NT>
    static CMFP_DEV void riemann_tracers(const Params& P, int d, const double* Lg, const double* Rg, const double* aL, const double* aR,
                                         double* F, double* Ft) {
        double L[5], R[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            L[n] = Lg[n];
            R[n] = Rg[n];
        }
        rot_g2l(L, H_U, d);
        rot_g2l(R, H_U, d);
        HllAux x;
        if (FLUX_ == 1) {
            hlle_solve(L, R, P.gas.gamma, P.gas.gamma, F, &x);
#pragma unroll
            for (int t = 0; t < NT; ++t) Ft[t] = hlle_tracer_flux(x, L, R, aL[t], aR[t]);
        } else {
            hllc_solve(L, R, P.gas.gamma, P.gas.gamma, F, &x);
#pragma unroll
            for (int t = 0; t < NT; ++t) Ft[t] = hllc_tracer_flux(x, L, R, aL[t], aR[t]);
        }
        rot_l2g(F, 1, d);
    }
};

template <int RECON_> struct FieldPhys {
    static constexpr int NC = 10;
    static constexpr int NQ = 10;
    static constexpr int NF = 8;  // F[mu] = F[ep] = 0 (MFP_field_RH.cpp:33)
    static constexpr int RECON = RECON_;
    static constexpr bool IS_HYDRO = false;
    typedef FieldParams Params;

    static CMFP_DEV bool cons2prim(const Params&, const double* U, double* Q) {
#pragma unroll
        for (int n = 0; n < 10; ++n) Q[n] = U[n];
        return true;
    }
    static CMFP_DEV void prim2cons(const Params&, const double* Q, double* U) {
#pragma unroll
        for (int n = 0; n < 10; ++n) U[n] = Q[n];
    }
    static CMFP_DEV void speed_from_prim(const Params& P, const double*, double* s) {
        s[0] = s[1] = s[2] = P.fastest;  // MFP_field.H:76-79
    }
    static CMFP_DEV void riemann(const Params& P, int d, const double* Lg, const double* Rg, double* F) {
        double L[10], R[10];
#pragma unroll
        for (int n = 0; n < 10; ++n) {
            L[n] = Lg[n];
            R[n] = Rg[n];
        }
        // vector_idx = {Bx, Dx}  (MFP_field.cpp:33-34)
        rot_g2l(L, F_BX, d);
        rot_g2l(L, F_DX, d);
        rot_g2l(R, F_BX, d);
        rot_g2l(R, F_DX, d);
        field_rh_solve(P.c0, P.ch, L, R, F);
        rot_l2g(F, F_BX, d);
        rot_l2g(F, F_DX, d);
    }
};

// ------------------------------------------------ cell reconstruction + trace
// Face values of one cell along one direction from its stencil of primitives.
// st[s][n]: stencil point s (0..SL-1), primitive n.
//
// Hydro: HydroState::calc_reconstruction MFP_hydro.cpp:780-1038 — slopes of
// {rho,u,v,w,eps = p/(gamma-1)}, positivity corrections (:914-964), faces
// (:966-1002).  `st` must hold eps (not p) in slot H_P.
template <int RECON>
CMFP_DEV void hydro_cell_faces(const double gamma, const double (*st)[5], double* lo, double* hi) {
    constexpr int SL = ReconTraits<RECON>::SL;
    constexpr int off = SL / 2;
    double Q[5], sl[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        double s1[SL];
#pragma unroll
        for (int s = 0; s < SL; ++s) s1[s] = st[s][n];
        sl[n] = recon_slope<RECON>(s1);
        Q[n] = s1[off];
    }
    double rho = Q[H_RHO], phi_rho = sl[H_RHO];
    double u = Q[H_U], phi_u = sl[H_U];
    double v = Q[H_V], phi_v = sl[H_V];
    double w = Q[H_W], phi_w = sl[H_W];
    double eps = Q[H_P], phi_eps = sl[H_P];

    if (fabs(phi_rho) > 2 * rho) { phi_rho = 2 * sign_ab(phi_rho, 0.0) * rho; }  // :934 (quirk Q1)
    double rho_lo = rho - 0.5 * phi_rho;
    double rho_hi = rho + 0.5 * phi_rho;
    double abs_phi = phi_u * phi_u + phi_v * phi_v + phi_w * phi_w;
    double eps_face = eps - 0.5 * fabs(phi_eps);
    if (eps_face <= 0.0) {
        phi_u = 0.0;
        phi_v = 0.0;
        phi_w = 0.0;
        phi_eps = 0.0;
    } else {
#ifdef CMFP_FASTMATH
        // the test of :948 multiplied through by rho_lo rho_hi (>= 0 after the clamp above): the quotient is only
        // formed where the rescaling applies
        if ((0.125 * abs_phi) * (rho_lo * rho_hi) > rho * eps_face) {
            double coeff_eps = f_div(rho, rho_lo * rho_hi) * eps_face;
#else
        double coeff_eps = f_div(rho, rho_lo * rho_hi) * eps_face;
        if ((0.125 * abs_phi) > coeff_eps) {
#endif
            double phi_scale = f_sqrt(abs_phi);
            coeff_eps = f_sqrt(8 * coeff_eps);
            phi_u = f_div(phi_u, phi_scale) * coeff_eps;
            phi_v = f_div(phi_v, phi_scale) * coeff_eps;
            phi_w = f_div(phi_w, phi_scale) * coeff_eps;
        }
        abs_phi = phi_u * phi_u + phi_v * phi_v + phi_w * phi_w;
        eps -= f_div(rho_lo * rho_hi, rho) * 0.125 * abs_phi;
    }
    lo[H_RHO] = rho_lo;
    hi[H_RHO] = rho_hi;
    lo[H_U] = u - 0.5 * f_div(rho_hi, rho) * phi_u;
    hi[H_U] = u + 0.5 * f_div(rho_lo, rho) * phi_u;
    lo[H_V] = v - 0.5 * f_div(rho_hi, rho) * phi_v;
    hi[H_V] = v + 0.5 * f_div(rho_lo, rho) * phi_v;
    lo[H_W] = w - 0.5 * f_div(rho_hi, rho) * phi_w;
    hi[H_W] = w + 0.5 * f_div(rho_lo, rho) * phi_w;
    lo[H_P] = (eps - 0.5 * phi_eps) * (gamma - 1.0);
    hi[H_P] = (eps + 0.5 * phi_eps) * (gamma - 1.0);
}

// EulerianState::calc_time_averaged_faces MFP_eulerian.cpp:609-676 for one
// component: in-place half-step trace with the single speed c.
CMFP_DEV void trace_faces(double c, double dt_2dx, double centre, double& lo, double& hi) {
    const double ft = 4.0 / 3.0;
    double lo_face = lo, hi_face = hi;
    double a6 = 6.0 * centre - 3 * (lo_face + hi_face);
    lo += c * dt_2dx * (hi_face - lo_face + (1 - c * ft * dt_2dx) * a6);
    hi -= c * dt_2dx * (hi_face - lo_face - (1 - c * ft * dt_2dx) * a6);
}

// physics adaptors used by the fused kernels (mfp_fused.hip, mfp_gather.hip); the staged
// kernels use the same device functions through HydroPhys / FieldPhys
template <int RECON_, int FLUX_> struct HydroF : HydroPhys<RECON_, FLUX_> {
    typedef HydroParams Params;
    static CMFP_DEV double gamma_of(const Params& P) { return P.gas.gamma; }
    // prim -> cons, += k * dF, cons -> prim  (MFP_eulerian.cpp:1132-1176)
    static CMFP_DEV bool correct(const Params& P, double* Q, double k, const double* dF) {
        double U[5];
        hydro_prim2cons(P.gas, Q, U);
#pragma unroll
        for (int n = 0; n < 5; ++n) U[n] += k * dF[n];
        return hydro_cons2prim(P.gas, U, Q);
    }
    // with NT passive scalars A = alpha of the face state: tracer = alpha rho, += k dT, alpha = tracer / rho' with the
    // corrected density (k_correct in mfp_kernels.hip; MFP_thermally_perfect_gas.cpp:212-214,243-245)
    template <int NT>
    static CMFP_DEV bool correct_tracers(const Params& P, double* Q, double* A, double k, const double* dF, const double* dT) {
        double U[5], T[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            T[t] = A[t] * Q[H_RHO];
            T[t] += k * dT[t];
        }
        hydro_prim2cons(P.gas, Q, U);
#pragma unroll
        for (int n = 0; n < 5; ++n) U[n] += k * dF[n];
        const bool ok = hydro_cons2prim(P.gas, U, Q);
        const double rinv = f_rcp(U[0]);
#pragma unroll
        for (int t = 0; t < NT; ++t) A[t] = T[t] * rinv;
        return ok;
    }
};
template <int RECON_> struct FieldF : FieldPhys<RECON_> {
    typedef FieldParams Params;
    static CMFP_DEV double gamma_of(const Params&) { return 0.0; }
    static CMFP_DEV bool correct(const Params&, double* Q, double k, const double* dF) {
#pragma unroll
        for (int n = 0; n < 8; ++n) Q[n] += k * dF[n];
        return true;
    }
};

}  // namespace CMFP_NS
