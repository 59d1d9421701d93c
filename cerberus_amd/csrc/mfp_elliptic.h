// mfp_elliptic.h — interface of the divergence projection (mfp_elliptic.hip) towards the context.
#pragma once
#include <hip/hip_runtime.h>

#include "mfp_types.h"

namespace cmfp_ell {
struct Cplx {
    double re, im;
};
// LinOpBCType of a domain face as Elliptic's set_field_bcs picks it (MFP_elliptic.cpp:83-122)
enum FaceBC { PERIODIC = 0, NEUMANN = 1, DIRICHLET = 2 };
// The node grid of one solve.  A direction with physical faces has n + 1 nodes; the solve runs on its periodic
// extension of P points (mirror image about a Neumann face, negative mirror image about a Dirichlet face: 2 n points
// for two faces of one kind, 4 n for one of each), on which the node Laplacian is diagonal in Fourier space.
struct Grid {
    int dim;
    int n[3];       // cells
    int P[3];       // points of the periodic extension (n in a periodic direction); 1 in unused directions
    int bc[3][2];   // FaceBC
    int pow2[3];    // P is a power of two: radix-2 FFT, else the direct O(P^2) transform
    double dx[3];
    __host__ __device__ long total() const { return (long)P[0] * P[1] * P[2]; }
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
// what the context keeps for the projection: node array, cell source, twiddle tables per extension length
struct Work {
    Cplx* nodes = nullptr;
    long nodes_cap = 0;
    double* S = nullptr;
    Cplx* tw[3] = {nullptr, nullptr, nullptr};
    int tw_len[3] = {0, 0, 0};
};
// nullptr: applicable; else the reason.  phys_bc[d][side]: the physical boundary type of the vector's first
// component on that face (cmfp_phys_bc of include/cerberus_mfp.h), ignored in periodic directions.
const char* make_grid(const cmfp::Geo& g, const int periodic[3], const int split[3], const int phys_bc[3][2], Grid& out);
hipError_t setup(const Grid& E, const cmfp::Geo& g, Work& w);
void release(Work& w);
// V must have one ghost layer filled (FillPatch) where the domain has physical faces
void project(const cmfp::Geo& g, const Grid& E, Work& w, double* V, double* phi_cc, const ChargeArgs* charge, hipStream_t st);
}  // namespace cmfp_ell
