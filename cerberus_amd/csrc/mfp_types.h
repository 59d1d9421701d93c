// mfp_types.h — POD types shared by the context (host) and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>

namespace cmfp {

struct GasP {
    double mass, charge, gamma;
};
struct HydroParams {
    GasP gas;
    double shock_threshold;
    int nt;  // passive scalars (n_tracers): comps 5 .. 5 + nt - 1 of the conserved / primitive / flux arrays
    // true mixture (cmfp_state_desc::mixture): bit 0 = gamma differs between the species, bit 1 = mass differs.
    // Per species cp = gamma / (mass (gamma - 1)), cv = 1 / (mass (gamma - 1)) and mass; the stage-by-stage kernels
    // then carry gamma as primitive 5 + nt (the reference's Gamma entry of the face states).
    int mix;
    double sp_cp[3], sp_cv[3], sp_mass[3];
};
struct FieldParams {
    double c0;       // MFP::lightspeed
    double ch;       // div_speed = div_transport * lightspeed  (MFP_field.cpp:384)
    double fastest;  // max(c0, ch)                             (MFP_field.cpp:386)
};

// Geometry of one state's ghosted SoA arrays on this block: local cell
// (i,j,k), 0 <= i < n[0], ghosts at negative / >= n indices; i fastest,
// component slowest (the FArrayBox layout).
struct Geo {
    int dim;
    int n[3];
    int ng[3];   // ghost width (0 in unused dims)
    int ext[3];  // n + 2 ng
    long sj, sk, sc;
    double dx[3];
    // device-resident time step (cmfp_advance_steps): when set, a kernel's `dt` argument is a factor
    // (1 for the flux update, 1/2 for a source half step) of *dtp, read on the device, so that a
    // sequence of steps can be enqueued without the host knowing dt
    const double* dtp;
    // 1: every launch takes the kernel instance that carries the domain-face rules, also on a block that touches no
    // physical face.  Set in FAST arithmetic on a domain with a non-periodic direction: the compiler contracts the
    // instances with and without the rules differently, and the ranks of a decomposed run must round alike.
    int bc_instance;
    __host__ __device__ long at(int i, int j, int k) const {
        return (long)(i + ng[0]) + (long)(j + ng[1]) * sj + (long)(k + ng[2]) * sk;
    }
    __host__ __device__ long cells() const { return sc; }
};

struct IBox {
    int lo[3], hi[3];
    __host__ __device__ long npts() const {
        return (long)(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
    }
};

// HydroViscous (hydro/viscous/MFP_hydro_viscous.cpp): kind = cmfp_viscosity, non-dimensional coefficients
struct ViscP {
    int kind;
    double mu0, T0, S, n, Pr;
};

struct StatePhys {
    int kind;   // cmfp_state_kind
    int recon;  // cmfp_recon
    int flux;   // cmfp_flux
    int nc;     // conserved comps (5 / 10)
    int nq;     // primitive comps carried (5 / 10)
    int ng;     // Reconstruction::num_grow of this state (<= the context's common ghost width)
    HydroParams hp;
    ViscP visc;
    FieldParams fp;
    int bc[10][3][2];
    int vacuum;  // field: mu == 1 and ep == 1 on every cell of the block (established on the device; they never change)
};

constexpr int CMFP_RED_BLOCKS = 2048;  // block partials per reduction word
// The 256 atomic slots of several reduction words handled by one launch (k_reduce_many): fold the slots into
// *out (mode 0) and leave them at the identity of the fold (0 for a maximum of positive doubles, all-ones —
// a NaN that fmin ignores — for a minimum) for the next step's atomics; mode 1 only resets.
struct RedJobs {
    static constexpr int MAX = 12;
    int n;
    unsigned long long* slots[MAX];
    unsigned long long* out[MAX];
    int is_max[MAX];
    int mode[MAX];
};

struct Ptr3 {
    double* p[3];
};

// species list of a plasma5 / Lorentz action
// CollisionsRambo (actions/reactions/MFP_collisions_rambo.cpp): per-species constants, the species
// being the hydro states of the action in list order
struct CollArgs {
    int on;
    double m[4], q[4], gamma[4], ez[4];  // gas mass, charge, gamma; the state's effective_zero
    double ccs[4][4];                     // cross sections / x_ref^2
    double c1, c2, c3;                    // MFP_collisions_rambo.cpp:60-62
    int* err;                             // device flag: primitive values outside of physical bounds
};

// collisions_rambo over states with sub-components (tracers): per species of the action the sub-component that takes
// part (cmfp_action_desc::species_component), its mass and charge, and the state's gas (for cons2prim of a mixture)
struct CollMix {
    int comp[4];
    double m[4], q[4];
    HydroParams hp[4];
};

// a species that is a mixture (cmfp_state_desc::mixture): mass and charge of a cell follow its mass fractions
// (HydroGas::get_mass_from_cons / get_charge_from_cons, MFP_hydro_gas.cpp:72-96,123-150).  nt = 0: the constants.
struct SpeciesMix {
    int nt;     // tracers read (comps 5 .. 5 + nt - 1 of the species' conserved data)
    int flags;  // bit 0: mass differs between the species, bit 1: charge differs
    double m[3], q[3];
};

struct SourceArgs {
    int n_species;
    double* sp_U[4];   // species conserved data read
    double* sp_dU[4];  // species dU accumulated (staged) or output (fused)
    double r[4];       // charge / mass  (q / m, MFP_plasma5.cpp:202)
    double* f_U;
    double* f_dU;
    long sp_sc[4];  // component strides
    long f_sc;
    double Larmor, Debye, lightspeed;
    double c_h, div_damp;
    int lorentz_only;
    int implicit;
    // fused CFL reduction (k_source_rk2<..., REDUCE>)
    double gamma[4];                  // species gamma
    int limits_dt;                    // plasma5 explicit: plasma / cyclotron dt limit (MFP_plasma5.cpp:461)
    double* part_speed[4];            // block partials (RED_BLOCKS doubles each)
    double* part_dt;
    unsigned long long* out_speed[4];  // final words
    unsigned long long* out_dt;
    CollArgs coll;  // a collisions_rambo action over the same species, evaluated with this source (fused step)
    SpeciesMix mix[4];  // stage-by-stage kernels only (a state with tracers never runs fused)
};

// Strang source stage C evaluated by the last CTU kernel of the step (the field's) on the cells it updates
struct SourceEpi {
    int ns;             // species (1 or 2)
    SourceArgs a;       // sp_U: the species' new-time data, already updated by their CTU kernels; f_U unused
    double kf[3];       // SourceCoef of the half step (host-side dt; recomputed with a device-resident dt)
    unsigned long long* red_speed[4];  // 256 slots each: max a + |u_d| of what was written (bits of a positive double)
    unsigned long long* red_dt;        // 256 slots: min plasma / cyclotron dt (all-ones = empty)
};

// Strang source stage A evaluated by the 2-D marching CTU kernels on the cells they load (mfp_source.h):
// explicit plasma5 / Lorentz over two species, FAST arithmetic, block without physical faces
struct SourcePro {
    int on;
    const double* sp[2];  // the species' arrays the CTU kernels read (values before stage A)
    const double* fld;    // the field's
    double r[2];          // charge / mass of the species
    double Larmor, Debye, lightspeed, c_h, div_damp;
    int lorentz_only;
    // Physical faces of the block.  A ghost cell there holds +-1 times the values of one valid cell (its mirror
    // image: reflect_even / reflect_odd; or the first valid cell: foextrap) for every component of every state, so
    // stage A of the ghost cell is the same signs applied to stage A of those values: the kernels undo the signs,
    // apply stage A and set the signs again (exact).  odd_*: bit n set = component n is reflect_odd on that face.
    int touch[2][2];          // [d][side]
    unsigned odd_h[2][2];     // hydro components (both species have the same rules)
    unsigned odd_f[2][2];     // field components
};

// Dirichlet data of the physical faces of a 2-D hydro state as the marching kernel reads it
// (cmfp_state_set_face_values; null: none): cell[d][side][n] over the ghost slab of that face (full ghosted extent
// of the other direction, i fastest), face[d][side][n] over the face plane (ghosted extent of the other direction)
struct MarchTab {
    const double* cell[2][2][5];
    const double* face[2][2][5];
};

// ghost fill rule per component for one (dim, side)
enum FillRule { FILL_SKIP = 0, FILL_WRAP = 1, FILL_FOEXTRAP = 2, FILL_EVEN = 3, FILL_ODD = 4 };
struct FillRules {
    int rule[10];
};

// One launch of the ghost fill of dimension d for several states and both sides (k_fill_multi)
struct FillJob {
    double* A;
    int nc, ngs;
    int on[2];          // side switched on
    FillRules rules[2];
};
struct FillJobs {
    int n;
    FillJob j[4];
};

// What MFP::estTimeStep combines (MFP_timestep.cpp:4-21), for the device-side step control
struct DtPlan {
    int n;
    int kind[24];    // 0: hydro state (word = max speed), 1: constant (field state), 2: action (min(cst, word))
    int word[24];
    double cst[24];
    int dim;
    double dx[3];
    double cfl;
    double force_dt;  // > 0: MFP::estTimeStep returns it before anything else (MFP_timestep.cpp:8)
};

// Host-side launchers of one floating-point mode (namespace strict / fast).
struct Launchers {
    // staged kernels: one per FAB-level function of the reference
    void (*prim)(const Geo&, const StatePhys&, const double* U, double* prim, int* err, hipStream_t);
    void (*recon)(const Geo&, const StatePhys&, const double* prim, Ptr3 rlo, Ptr3 rhi, int do_ctu, double dt,
                  hipStream_t);
    void (*face_bc)(const Geo&, const StatePhys&, int d, int side, double* rlo, double* rhi, hipStream_t);
    void (*fluxes)(const Geo&, const StatePhys&, int d, const double* rlo, const double* rhi, double* flux,
                   int grown, hipStream_t);
    void (*correct)(const Geo&, const StatePhys&, int d1, double* rlo, double* rhi, Ptr3 flux, double dt, int* err,
                    hipStream_t);
    void (*divergence)(const Geo&, int nc, Ptr3 flux, double* dU, double dt, hipStream_t);
    // valid-cell elementwise ops
    void (*copy)(const Geo&, int nc, double* dst, const double* src, int with_ghosts, hipStream_t);
    void (*zero)(const Geo&, int nc, double* dst, hipStream_t);
    void (*add)(const Geo&, int nc, double* dst, const double* src, hipStream_t);
    void (*lincomb)(const Geo&, int nc, double* dst, double a, const double* x, double b, const double* y,
                    hipStream_t);
    // sources (staged: accumulate into dU)
    void (*source)(const Geo&, const SourceArgs&, double dt, hipStream_t);
    // dt limits: result combined into *out (bits of a positive double) with atomics
    void (*max_speed)(const Geo&, const StatePhys&, const double* U, double* part, unsigned long long* out, hipStream_t);
    void (*plasma_dt)(const Geo&, const SourceArgs&, double* part, unsigned long long* out, hipStream_t);
    // HydroState::calc_viscous_fluxes for direction d: flux -= viscous stresses (cell data from U)
    void (*viscous)(const Geo&, const StatePhys&, int d, const double* U, double* flux, hipStream_t);
    // HydroViscous::get_min_dt cell loop: max over valid cells of calc_stability_condition
    void (*visc_stability)(const Geo&, const StatePhys&, const double* U, double* part, unsigned long long* out, hipStream_t);
    // dst[cell + shift] = table[cell] over `region`; table laid out over the box tbox, i fastest
    void (*table_store)(const Geo&, const IBox& region, const IBox& tbox, const double* table, double* dst, long shift,
                        hipStream_t);
    // ghosts
    void (*fill_dim)(const Geo&, int nc, int ngs, double* A, int d, int side, FillRules, hipStream_t);
    void (*pack)(const Geo&, int nc, int ngs, double* A, int d, int side, double* buf, int unpack, hipStream_t);
    // fill_dim of up to four states and both sides in one launch (the sides and states are independent)
    void (*fill_multi)(const Geo&, int d, const FillJobs&, hipStream_t);
    // fused paths (return 0 when the option set is not covered)
    int (*ctu_fused)(const Geo&, const StatePhys&, const double* Uin, double* Uout, int do_ctu, double dt,
                     const int touches[3][2], int* err, unsigned long long* red, int part, hipStream_t);
    int (*source_fused)(const Geo&, const SourceArgs&, double dt, int reduce, hipStream_t);
    // Collisions::calc_time_derivative on sp_U -> sp_dU of the first n_species entries (a.coll)
    void (*collisions)(const Geo&, const SourceArgs&, double dt, hipStream_t);
    // tile-free fused CTU update for any dimensionality (mfp_gather.hip); F1 / F2: face-flux scratch, 8 planes per direction
    int (*ctu_gather)(const Geo&, const StatePhys&, const double* Uin, double* Uout, const Ptr3& F1, const Ptr3& F2,
                      int do_ctu, double dt, const int touches[3][2], int* err, unsigned long long* red, hipStream_t);
    // two-pass z-marching update of a 3-D field state (mfp_field3d.hip); F1: sweep-1 flux scratch
    int (*ctu_field3d)(const Geo&, const StatePhys&, const double* Uin, double* Uout, const Ptr3& F1, double* DU, int do_ctu, double dt,
                       const int touches[3][2], const SourceEpi* epi, hipStream_t);
    // the tiled 2-D field update with the explicit Strang source stage C as its epilogue (mfp_fused.hip)
    int (*ctu_field_src)(const Geo&, const StatePhys&, const double* Uin, double* Uout, int do_ctu, double dt,
                         const int touches[3][2], const SourceEpi& epi, int part, hipStream_t);
    // the cells the interior launch (part 1) of ctu_fused / ctu_field_src updates (flags: bit 8 as in their `part`)
    int (*ctu_interior)(const Geo&, const StatePhys&, int do_ctu, int flags, IBox* box);
    // two 2-D hydro states of the same physics in one wave-marching launch (mfp_march.hip); 0: not covered, nothing launched
    // pro (may be null): source stage A as the kernel's prologue; Uin[0] == nullptr probes (with pro: whether the prologue is covered)
    int (*ctu_march2)(const Geo&, const StatePhys* const sp[2], const double* const Uin[2], double* const Uout[2], int do_ctu, double dt,
                      const int touches[3][2], int* err, unsigned long long* const red[2], int part, const SourcePro* pro, hipStream_t);
    // 2-D vacuum field update by the wave-marching kernel (mfp_march_field.hip), epi may be null; 0: not covered
    int (*ctu_march_field)(const Geo&, const StatePhys&, const double* Uin, double* Uout, int do_ctu, double dt,
                           const int touches[3][2], const SourceEpi* epi, int part, const SourcePro* pro, hipStream_t);
    // *flag (zero before the launch) becomes 1 if a valid cell of the field array has mu != 1 or ep != 1
    void (*field_vacuum_check)(const Geo&, const double* U, int* flag, hipStream_t);
    // the three factors of MFP_plasma5.cpp:123-129 as the source kernels compute them on the host
    void (*source_coef)(const SourceArgs&, double dt, double out[3]);
    // source_fused on a sub-box of the valid cells, without reductions
    int (*source_fused_box)(const Geo&, const SourceArgs&, double dt, const IBox& box, hipStream_t);
    // batched leaves
    int (*leaf_reconstruct)(int recon, long n, const double* st, double* slope, double* lo, double* hi, hipStream_t);
    int (*leaf_riemann)(int flux, long n, const double* L, const double* R, double* F, double c0, double ch,
                        double thr, hipStream_t);
    int (*leaf_cons2prim)(long n, double mass, double gamma, const double* U, double* Q, double* sp, hipStream_t);
    int (*leaf_prim2cons)(long n, double mass, double gamma, const double* Q, double* U, hipStream_t);
    void (*reduce_final)(const double* part, int n, int is_max, unsigned long long* out, hipStream_t);
    void (*reduce_many)(const RedJobs&, hipStream_t);
    // *out (all-ones before the launch) becomes the smallest (k * n1 + j) * n0 + i of the valid cells of a hydro state
    // whose conserved values do not give valid primitives (HydroGas::prim_valid MFP_hydro_gas.cpp:152-164)
    void (*first_invalid)(const Geo&, const StatePhys&, const double* U, unsigned long long* out, hipStream_t);
    // MFP::estTimeStep + computeNewDt on the device: log[idx] = dt[0]; dt[0] = min(cfl * est, 1.1 * dt[0])
    void (*new_dt)(const DtPlan&, const unsigned long long* red, double* dt, double* log, int idx, hipStream_t);
    // ctu_fused for a 2-D hydro state with Dirichlet face data, on the marching kernel (mfp_march.hip); 0: not covered
    int (*ctu_march_tab)(const Geo&, const StatePhys&, const double* Uin, double* Uout, int do_ctu, double dt, const int touches[3][2],
                         int* err, unsigned long long* red, int part, const MarchTab& tab, hipStream_t);
    // reduce_many (at most RedJobs::MAX jobs, one workgroup) followed by new_dt, in one launch
    void (*fold_new_dt)(const RedJobs&, const DtPlan&, const unsigned long long* red, double* dt, double* log, int idx, hipStream_t);
    // `collisions` where a state has sub-components (tracers / a mixture): the shares of every sub-component accounted for
    void (*collisions_mix)(const Geo&, const SourceArgs&, const CollMix&, double dt, hipStream_t);
};

}  // namespace cmfp

namespace strict {
const cmfp::Launchers* launchers();
}
namespace fast {
const cmfp::Launchers* launchers();
}
