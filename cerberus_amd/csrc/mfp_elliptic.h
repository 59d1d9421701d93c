// mfp_elliptic.h — interface of the divergence projection (mfp_elliptic.hip) towards the context.
#pragma once
#include <hip/hip_runtime.h>

#include "mfp_types.h"

namespace cmfp_ell {
struct Cplx {
    double re, im;
};
// S_cc = -cd_scale * (sum_s rho_s q_s / m_s) / ep on the valid cells
struct ChargeArgs {
    int ns;
    const double* rho[8];  // the species' conserved data (Density first; a mixture's tracers at comps 5 ..)
    double r[8];
    cmfp::SpeciesMix mix[8];  // nt != 0: charge / mass of a cell from its mass fractions (MFP_hydro.cpp:1720-1721)
    long sc;                  // component stride of the species' arrays
    const double* ep;
    double cd_scale;
};
// what the context keeps for the projection: node array, cell source, twiddle tables per dimension
struct Work {
    Cplx* nodes = nullptr;
    double* S = nullptr;
    Cplx* tw[3] = {nullptr, nullptr, nullptr};
};
const char* check(const cmfp::Geo& g, const int periodic[3], const int split[3]);  // nullptr: applicable
hipError_t setup(const cmfp::Geo& g, Work& w);
void release(Work& w);
void project(const cmfp::Geo& g, Work& w, double* V, double* phi_cc, const ChargeArgs* charge, hipStream_t st);
}  // namespace cmfp_ell
