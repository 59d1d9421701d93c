/*
 * cerberus_mfp.h — C ABI of the MI355X-native per-patch hyperbolic update
 * (five-moment multi-fluid + Maxwell) that replaces the hot path of
 * PlasmaSimUQ/cerberus (MFP).
 *
 * The reference has no FFI: the path sits behind its C++ State/Action plugin
 * API.  Every entry point below names the reference interface it replaces
 * (file:line relative to the reference tree).  All functions return 0 on
 * success and a negative cmfp_status otherwise; cmfp_last_error() gives the
 * text.  Nothing C++ and no torch type crosses this boundary: plain pointers,
 * sizes and POD descriptors only.  State arrays stay resident in HBM between
 * calls; one host thread per context.
 */
#ifndef CERBERUS_MFP_H
#define CERBERUS_MFP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CMFP_MAX_DIM 3
#define CMFP_MAX_COMP 10
#define CMFP_MAX_ACTION_STATES 8
#define CMFP_MAX_TRACERS 2

typedef enum {
    CMFP_OK = 0,
    CMFP_ERR_INVALID = -1,  /* bad argument / unsupported option           */
    CMFP_ERR_HIP = -2,      /* a HIP runtime call failed                    */
    CMFP_ERR_PHYSICS = -3,  /* rho, p or T <= 0 on device: the reference's  */
                            /* "Primitive values outside of physical        */
                            /* bounds!!" abort (MFP_hydro_gas.cpp:152-164)  */
    CMFP_ERR_NODEVICE = -4  /* no HIP device / extension not usable         */
} cmfp_status;

/* State kinds: tags "hydro" (hydro/MFP_hydro.cpp:51) and "field"
 * (field/MFP_field.cpp:42). */
typedef enum { CMFP_STATE_HYDRO = 0, CMFP_STATE_FIELD = 1 } cmfp_state_kind;

/* Reconstruction tags (MFP_reconstruction.cpp:51,73,115,155,208,239,275). */
typedef enum {
    CMFP_RECON_CONSTANT = 0,
    CMFP_RECON_MINMOD = 1,
    CMFP_RECON_VANLEER = 2,
    CMFP_RECON_MC = 3,
    CMFP_RECON_CENTRE = 4,
    CMFP_RECON_O6 = 5,
    CMFP_RECON_WENO = 6
} cmfp_recon;

/* Riemann solver tags: hydro "HLLC" (MFP_hllc.cpp:6), "HLLE"
 * (MFP_hydro_hlle.cpp:8), "HLLE/HLLC" (MFP_hybrid_hll.cpp:9); field
 * "RankineHugoniot" (MFP_field_RH.cpp:11). */
typedef enum {
    CMFP_FLUX_HLLC = 0,
    CMFP_FLUX_HLLE = 1,
    CMFP_FLUX_HYBRID_HLL = 2,
    CMFP_FLUX_FIELD_RH = 10
} cmfp_flux;

/* Mathematical ghost-cell rule per component and domain face; the numeric
 * values are amrex::BCType's (listed in modules/boundary/MFP_fillbc.cpp:100-109).
 * The map physical-BC-name -> these codes (modules/boundary/MFP_bc.cpp:11-31,
 * 99-109) is host-side logic (cerberus_amd/bc.py, cerberus_amd/host). */
typedef enum {
    CMFP_BC_BOGUS = -666,
    CMFP_BC_REFLECT_ODD = -1,
    CMFP_BC_INT_DIR = 0,
    CMFP_BC_REFLECT_EVEN = 1,
    CMFP_BC_FOEXTRAP = 2,
    CMFP_BC_EXT_DIR = 3,
    CMFP_BC_HOEXTRAP = 4
} cmfp_bc;

/* Action tags "CTU" (MFP_CTU_hydro.cpp:10), "plasma5" (MFP_plasma5.cpp:9),
 * "Lorentz" (MFP_lorentz.cpp), "collisions_rambo"
 * (actions/reactions/MFP_collisions_rambo.cpp:7). */
typedef enum {
    CMFP_ACTION_CTU = 0,
    CMFP_ACTION_PLASMA5 = 1,
    CMFP_ACTION_LORENTZ = 2,
    CMFP_ACTION_COLLISIONS_RAMBO = 3
} cmfp_action_kind;

typedef enum { CMFP_SOLVER_EXPLICIT = 0, CMFP_SOLVER_IMPLICIT = 1 } cmfp_source_solver;

/* viscosity.type: "Sutherland", "PowerLaw", "UserDefined" (MFP_hydro_viscous.cpp:117,183,243) */
typedef enum { CMFP_VISC_NONE = 0, CMFP_VISC_SUTHERLAND = 1, CMFP_VISC_POWERLAW = 2, CMFP_VISC_USER = 3 } cmfp_viscosity;

/* time_integration_scheme (MFP_config.cpp:71-87). */
typedef enum { CMFP_TI_EULER = 0, CMFP_TI_RK2 = 1, CMFP_TI_STRANG = 2 } cmfp_time_integrator;

/* Kernel floating-point contraction mode.  STRICT: no FMA contraction, the
 * expression order of the reference (bit-comparable with a g++ build of the
 * same expressions).  FAST: the same kernels compiled with FMA contraction
 * (documented difference, must stay within 1e-12 relative of STRICT). */
typedef enum { CMFP_FP_STRICT = 0, CMFP_FP_FAST = 1 } cmfp_fp_mode;

/* Uniform single-level geometry (AMReX inputs geometry.*, amr.n_cell) plus the
 * block of it this context owns (one block per GPU). */
typedef struct {
    int dim;                        /* AMREX_SPACEDIM: 1, 2 or 3                */
    int n_cell[CMFP_MAX_DIM];       /* global cells (amr.n_cell)                */
    double prob_lo[CMFP_MAX_DIM];   /* geometry.prob_lo                         */
    double dx[CMFP_MAX_DIM];        /* cell size                                */
    int is_periodic[CMFP_MAX_DIM];  /* geometry.is_periodic                     */
    int block_lo[CMFP_MAX_DIM];     /* first global cell index of this block    */
    int block_n[CMFP_MAX_DIM];      /* cells of this block                      */
} cmfp_geometry;

/* Global scalars: MFP::lightspeed, Larmor, Debye (MFP_config.cpp:374-457),
 * cfl (MFP_config.cpp:188), the integrator (MFP_config.cpp:71-87). */
typedef struct {
    double lightspeed;
    double Larmor;
    double Debye;
    double cfl;
    int time_integrator; /* cmfp_time_integrator */
    int fp_mode;         /* cmfp_fp_mode          */
    double n0;           /* MFP::n0 = n_ref x_ref^3 (MFP_config.cpp:390-392); collisions only */
    double force_dt;     /* Lua 'force_dt' (default.lua, MFP_config.cpp): > 0 makes MFP::estTimeStep return it  */
                         /* before anything else (MFP_timestep.cpp:8); <= 0: the CFL estimate                   */
} cmfp_params;

/* One Eulerian state, in the reference's sorted-name order
 * (MFP_config.cpp:102-139).  hydro: states.<name>.{gas,reconstruction,flux,
 * shock_detector} (MFP_hydro.cpp:296-442); field: {reconstruction,flux,
 * div_transport,div_damping} (MFP_field.cpp:273-403). */
typedef struct {
    int kind;           /* cmfp_state_kind */
    int reconstruction; /* cmfp_recon      */
    int flux;           /* cmfp_flux       */
    double mass;        /* gas.mass   (single species)                         */
    double charge;      /* gas.charge                                          */
    double gamma;       /* gas.gamma                                           */
    double shock_threshold; /* shock_detector.threshold (<=0: none)            */
    double div_transport;   /* field: c_h / c                                  */
    double div_damping;     /* field                                           */
    double effective_zero;  /* states.<name>.effective_zero (MFP_eulerian.cpp:62), default 1e-14 */
    /* hydro: states.<name>.viscosity (hydro/viscous/MFP_hydro_viscous.cpp), coefficients already
     * non-dimensionalised as the reference's constructors do (:136-139, :198-200):
     * mu0 / (rho_ref x_ref u_ref), T0 / T_ref, S / T_ref.  2-D and 1-D blocks only. */
    int viscosity;          /* cmfp_viscosity */
    double visc_mu0, visc_T0, visc_S, visc_n, visc_Pr;
    /* hydro: mass fractions `alpha` of a gas of n_tracers + 1 species (value.alpha = {...}; passive scalars where
     * the species have equal mass and gamma, see `mixture` below;
     * HydroState::n_tracers MFP_hydro.cpp:296-323): conserved comps 5 .. 5 + n_tracers - 1 hold
     * rho * alpha (MFP_thermally_perfect_gas.cpp:212-214,243-245).  0 <= n_tracers <= 2. */
    int n_tracers;
    /* BCRec per conserved component: bc[comp][dim][0=lo,1=hi] (cmfp_bc).
     * hydro uses comps 0..4 (+ n_tracers), field 0..9. */
    int bc[CMFP_MAX_COMP][CMFP_MAX_DIM][2];
    /* hydro: a gas of n_tracers + 1 species of different mass / gamma (ThermallyPerfectGas::get_gamma_from_prim /
     * _from_cons, get_cp_*, MFP_thermally_perfect_gas.cpp:51-177; HydroGas::get_mass_from_*, MFP_hydro_gas.cpp:50-96):
     * mixture != 0 reads species_mass / species_gamma [0 .. n_tracers] (mass, gamma above = species 0) and gamma,
     * mass of a cell follow its mass fractions alpha_i = tracer_i / rho, the last species taking 1 - sum(alpha).
     * The plasma5 / Lorentz sources and the plasma dt limit read mass and charge of a cell the same way
     * (get_mass_from_cons / get_charge_from_cons, MFP_plasma5.cpp:199-202,364-365,544-545; species_charge).
     * Built on the stage-by-stage kernels (CTU / RK2 / euler updates, viscous fluxes, Dirichlet faces, sources,
     * dt limits, collisions_rambo) and in the charge density of cmfp_elliptic_project. */
    int mixture;
    double species_mass[CMFP_MAX_TRACERS + 1];
    double species_gamma[CMFP_MAX_TRACERS + 1];
    double species_charge[CMFP_MAX_TRACERS + 1];
} cmfp_state_desc;

/* One action, in sorted-name order (MFP_config.cpp:261-291). */
typedef struct {
    int kind;             /* cmfp_action_kind                                   */
    int corner_transport; /* CTU: 'corner_transport' (MFP_CTU_hydro.cpp:23)      */
    int solver;           /* plasma5: cmfp_source_solver (MFP_plasma5.cpp:22-31) */
    int n_states;
    /* CTU: the states it transports.  plasma5/Lorentz: species in the order
     * of the Lua 'states' list, then the field last.  collisions_rambo: the
     * hydro states in list order (one species of each, see species_component). */
    int states[CMFP_MAX_ACTION_STATES];
    /* collisions_rambo: hard-sphere cross sections between species a and b of the
     * list, already divided by x_ref^2 and symmetric (MFP_collisions_rambo.cpp:39-55);
     * only pairs with a neutral partner use them. */
    double cross_sections[CMFP_MAX_ACTION_STATES][CMFP_MAX_ACTION_STATES];
    /* collisions_rambo: the sub-component of state n that takes part (the entry of the Lua 'species' list found in
     * gas.names, Collisions::SpeciesInfo::alpha_idx MFP_collisions.cpp:43-57); 0 .. n_tracers of that state, 0 for
     * a single-species gas.  One species per state, in list order: the reference reads the species' data with the
     * species index (:181-190), so any other arrangement is refused by cmfp_create. */
    int species_component[CMFP_MAX_ACTION_STATES];
} cmfp_action_desc;

typedef struct cmfp_context cmfp_context;

/* Library / device probe.  cmfp_version: build string.  cmfp_device_count:
 * number of HIP devices visible (0 => every compute entry point fails with
 * CMFP_ERR_NODEVICE; there is no CPU fallback). */
const char* cmfp_version(void);
int cmfp_device_count(void);

/* Builds the per-GPU solver: replaces MFP::variableSetUp + read_config
 * (MFP_setup.cpp:7, MFP_config.cpp:40-356) for the objects on the path.
 * `device` is the HIP device ordinal, `stream` a hipStream_t (NULL = default
 * stream) on which every kernel of this context is launched. */
int cmfp_create(const cmfp_geometry* geom, const cmfp_params* params, int n_states,
                const cmfp_state_desc* states, int n_actions, const cmfp_action_desc* actions,
                int device, void* stream, cmfp_context** out);
int cmfp_destroy(cmfp_context* ctx);
const char* cmfp_last_error(const cmfp_context* ctx);
/* The first offending cell of the last CMFP_ERR_PHYSICS ("error flag + first offending
 * cell"; the reference aborts inside HydroGas::prim_valid, MFP_hydro_gas.cpp:152-164, without a location): index of
 * the hydro state and block-local (i, j, k) of the first valid cell, in state then memory order, whose cell-centred
 * values are invalid when the failure is detected.  CMFP_ERR_INVALID (state = -1) if the last error was another one,
 * or if only a face state failed (after reconstruction / transverse correction), which leaves no invalid cell. */
int cmfp_last_error_cell(const cmfp_context* ctx, int* state, int* cell /* [3] */);

/* Number of conserved comps / ghost width (Reconstruction::num_grow,
 * MFP_reconstruction.cpp:60,82,124,164,217,248,284) of a state. */
int cmfp_state_ncomp(const cmfp_context* ctx, int state);
int cmfp_state_ngrow(const cmfp_context* ctx, int state);

/* State data host <-> HBM: replaces EulerianState::init_data's store into the
 * new-time MultiFab (MFP_eulerian.cpp:80-139) and a plotfile read-back.  Host
 * layout: valid cells of this block only, [comp][k][j][i], i fastest. */
int cmfp_state_upload(cmfp_context* ctx, int state, const double* host);
int cmfp_state_download(cmfp_context* ctx, int state, double* host);
/* Zero-copy view of the HBM-resident new-time data: base pointer of the
 * ghosted SoA array, ghost width and padded extents (i fastest, comp slowest:
 * the FArrayBox layout). */
int cmfp_state_device_view(cmfp_context* ctx, int state, double** dptr, int* ngrow,
                           long long extent[CMFP_MAX_DIM], int* ncomp);

/* Dirichlet data of a physical domain face: the per-face values / functions of (x, y, z, t) the
 * reference reads from states.<name>.bc.<dim>.<side>.<primitive> (BoundaryState::set,
 * modules/boundary/MFP_fillbc.cpp:18-23; HydroState MFP_hydro.cpp:355-358, FieldState
 * MFP_field.cpp:350), tabulated by the caller.  A face with data is the reference's
 * where_is_inflow = ext_dir (:72-87): EulerianState::update_boundary_cells overwrites the primitive
 * in the ghost cells beyond the face (MFP_eulerian.cpp:435-511) and face_bc sets the outside face
 * state before each Riemann sweep (:838-877).  `prim` indexes the carried primitives: hydro
 * {rho, x_vel, y_vel, z_vel, p, alpha_i}, field = conserved comps.  cell_values: the ghost layers
 * beyond the face, i fastest, over the box that has the context's ghost width (cmfp_state_device_view)
 * along `dim` and the block's ghosted extent in the other dimensions; face_values: one plane of the
 * same transverse extent, at the face centres.  Host pointers; NULL, NULL removes the entry.
 * Functions of the local state (the reference passes the interior primitives to the Lua function)
 * are outside this interface.  States with Dirichlet data run on the stage-by-stage kernels; where
 * the domain is split between ranks every rank registers every entry (a table is applied only on a
 * physical face of the rank's own block), so that all ranks take the same kernel path. */
int cmfp_state_set_face_values(cmfp_context* ctx, int state, int dim, int side, int prim,
                               const double* cell_values, const double* face_values);

/* 'dynamic' components of a field state (states.<name>.dynamic = {'x_B', ...}; FieldState::set_udf
 * MFP_field.cpp:175-221 -> EulerianState::dynamic_functions): inside calc_primitives the listed components take the
 * value of their initialisation function at (x, y, z, t) — in the primitives the fluxes are built from and, through
 * prim2cons, in the conserved copy the actions of the step work on (UpdateData::U; MFP_eulerian.cpp:403-423), not in
 * the state's own data.  `values`: the function tabulated by the caller over the block grown by the context's ghost
 * width (cmfp_state_device_view), i fastest, host pointer, at the time the integrator evaluates its fluxes at
 * (strang: time + dt, MFP_advance.cpp:208-217; euler / RK2: time, :72-104); NULL removes the entry.  The reference
 * reads 'dynamic' for field states only: hydro states are refused.  A state with an entry runs on the
 * stage-by-stage kernels, step by step (cmfp_advance_steps returns 1). */
int cmfp_state_set_dynamic(cmfp_context* ctx, int state, int comp, const double* values);

/* ---- Action plugin API, one call per virtual of class Action
 * (actions/MFP_action.H:60-90).  They act on the context's UpdateData
 * {U, dU, status} per state (actions/MFP_action.H:16-29), owned by the
 * library for the duration of one cmfp_advance-equivalent sequence started
 * with cmfp_update_begin. */
int cmfp_update_begin(cmfp_context* ctx); /* Vector<UpdateData> RK_step(...) (MFP_advance.cpp:62,141) */
/* Action::get_data (MFP_action.cpp:15-102; HydroCTU::get_data
 * MFP_CTU_hydro.cpp:43-52; Plasma5::get_data MFP_plasma5.cpp:60-71). */
int cmfp_action_get_data(cmfp_context* ctx, int action, double time);
/* Action::calc_time_derivative (Plasma5: MFP_plasma5.cpp:73-455, Lorentz:
 * MFP_lorentz.cpp:60-198). */
int cmfp_action_calc_time_derivative(cmfp_context* ctx, int action, double time, double dt);
/* HydroCTU::calc_spatial_derivative (MFP_CTU_hydro.cpp:54-532). */
int cmfp_action_calc_spatial_derivative(cmfp_context* ctx, int action, double time, double dt,
                                        double flux_register_scale);
/* Action::get_allowed_time_step (Plasma5: MFP_plasma5.cpp:457-567). */
int cmfp_action_get_allowed_time_step(cmfp_context* ctx, int action, double* dt);
/* EulerianState::get_allowed_time_step (MFP_eulerian.cpp:255-311),
 * FieldState::get_allowed_time_step (MFP_field.cpp:57-67). */
int cmfp_state_get_allowed_time_step(cmfp_context* ctx, int state, double* dt);
/* Read back dU / U of one state's UpdateData (valid cells, host layout as
 * cmfp_state_upload) — test and debugging aid. */
int cmfp_update_download(cmfp_context* ctx, int state, int which /*0=U,1=dU*/, double* host);

/* ---- Level API.
 * cmfp_est_time_step: MFP::estTimeStep (MFP_timestep.cpp:4-21) for this
 * block, without the cross-rank ReduceRealMin (the caller does it: RCCL min).
 * cmfp_compute_new_dt: the growth cap of MFP::computeNewDt
 * (MFP_timestep.cpp:86-93) applied to an already-reduced estimate.
 * cmfp_advance: MFP::advance (MFP_advance.cpp:6-56) incl. advance_RK /
 * advance_strang (MFP_advance.cpp:58-289). */
int cmfp_est_time_step(cmfp_context* ctx, double* dt);
double cmfp_compute_new_dt(double est_dt, double old_dt);
int cmfp_advance(cmfp_context* ctx, double time, double dt);
/* nsteps iterations of the loop around it — Amr::coarseTimeStep -> MFP::advance, then
 * MFP::computeNewDt = min(cfl * estTimeStep, 1.1 * dt) (main.cpp:154-162, MFP_timestep.cpp:4-21,61-93;
 * no stop_time clip) — with dt resident on the device: every kernel reads it there and a one-thread
 * kernel applies estTimeStep / computeNewDt after each step, so nsteps steps are enqueued without a
 * host round trip (per-step launch + read-back latency is what bounds small grids).  dt0: the first
 * step's dt; dts[0..nsteps): the dt each step used; dt_next: the one computeNewDt leaves for the
 * next step.  Same arithmetic as cmfp_est_time_step / cmfp_compute_new_dt: STRICT results are
 * bit-identical to the per-step calls.  Returns 1 without advancing where it does not apply (stage-by-stage
 * path, viscous states, a block split between ranks: the cross-rank dt reduction is the caller's).  If the
 * batch stops early with an error, dts[0..k) holds the dt of the k steps that were completed, dts[k..nsteps)
 * is 0 and dt_next the dt the next step would use, so the caller still knows the time the states reached.
 * params.force_dt > 0 (MFP_timestep.cpp:8) keeps every dt at that value. */
int cmfp_advance_steps(cmfp_context* ctx, double time, double dt0, int nsteps, double* dts, double* dt_next);
/* cmfp_advance_steps for callers that cannot allocate device memory (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same
 * arguments, results and return codes as cmfp_advance_steps, but the device-resident time step, the dt
 * log of the batch and nothing else live in a file-scope `static __device__` array of the library
 * (one copy per GPU, CMFP_STATIC_MAX_STEPS + 1 doubles, held until the process ends) instead of a buffer
 * the call allocates, and the call makes no hipMalloc / hipFree / stream or event creation of any kind:
 * nsteps > CMFP_STATIC_MAX_STEPS is refused with CMFP_ERR_INVALID, and so is a context whose fused step
 * would still have to allocate its ping-pong / flux scratch (cmfp_create reserves it for the default
 * kernel selection; after cmfp_set_fused(2 or 3) or on a block split between ranks run one cmfp_advance
 * first).  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU (two
 * contexts on different streams) race; calls in sequence are independent (the call rewrites every word
 * it reads). */
#define CMFP_STATIC_MAX_STEPS 4096
int cmfp_advance_steps_static(cmfp_context* ctx, double time, double dt0, int nsteps, double* dts, double* dt_next);
/* Elliptic::apply_change with projection = 1 (actions/MFP_elliptic.cpp:70-81): project_divergence_field
 * (:667-782) = two calls of solve_divergence (:784-964) on the new-time data of `field_state` - the B field
 * with psi, then the D field with phi and the cell-centred source -Larmor / (Debye^2 c) * rho_c / ep, rho_c
 * the charge density of the n_species hydro states listed (the species of the LAST plasma5 action, in sorted-name
 * order, associated with the field: the reference zeroes its charge density inside the loop over the associated
 * actions, :722-766, so earlier plasma5 actions on the same field do not contribute).  The host side calls it
 * after cmfp_advance for every `elliptic` action, where the reference's MFP::advance loops apply_change
 * (MFP_advance.cpp:49).  The reference solves with AMReX's nodal projection (MLNodeLaplacian + MLMG, third party,
 * absent from its tree); here the same nodal operators are solved exactly, by FFT on the periodic extension of
 * the node grid (csrc/mfp_elliptic.hip).
 * Domain faces: bc_B / bc_D = the physical boundary type (cmfp_phys_bc) of the field's fill_B_bc / fill_D_bc entry
 * on face [2 * dim + side] - what set_field_bcs (:83-122) reads from phys_fill_bc[Bx] / [Dx]: outflow and
 * symmetry give a Neumann face, inflow a homogeneous Dirichlet face; NULL = 'outflow' everywhere, the default of
 * those entries (MFP_field.cpp:323-325); entries of periodic directions are ignored.  'asymmetry' is refused (the
 * reference maps it to LinOpBCType::reflect_odd, which AMReX's nodal solver does not define).
 * Limits: 2-D or 3-D (1-D: returns CMFP_OK without doing anything, as the reference's body is compiled out, :641);
 * block = domain (single rank); at most 8192 points per transformed line = cells of a periodic direction, twice
 * the cells between two faces of one kind, four times between a Neumann and a Dirichlet face; lines whose length
 * is not a power of two take a direct O(P^2) transform.  Otherwise CMFP_ERR_INVALID with the reason.
 * Parity unpinned (no AMReX here, no fixture in the reference). */
typedef enum {
    CMFP_PHYSBC_INTERIOR = 0,
    CMFP_PHYSBC_INFLOW = 1,
    CMFP_PHYSBC_OUTFLOW = 2,
    CMFP_PHYSBC_SYMMETRY = 3,
    CMFP_PHYSBC_ASYMMETRY = 4   /* FieldState::bc_names, MFP_field.cpp:36-40 */
} cmfp_phys_bc;
int cmfp_elliptic_project_bc(cmfp_context* ctx, int field_state, int n_species, const int* species_states,
                             const int* bc_B /* [6] or NULL */, const int* bc_D /* [6] or NULL */);
/* = cmfp_elliptic_project_bc(ctx, field_state, n_species, species_states, NULL, NULL) */
int cmfp_elliptic_project(cmfp_context* ctx, int field_state, int n_species, const int* species_states);
/* Uses hand-fused kernels inside cmfp_advance where available (default 1: the
 * LDS-tiled kernel for 2-D blocks, the tile-free gather kernels for 1-D and 3-D);
 * 2 forces the gather kernels in any dimensionality; 0 forces the stage-by-stage
 * kernels that mirror the reference's FAB-level functions one to one.  Results
 * are identical. */
int cmfp_set_fused(cmfp_context* ctx, int enable);

/* Path the last cmfp_advance took: 0 stage-by-stage kernels, 1 tiled fused
 * kernels, 2 gather kernels (for at least one state). */
int cmfp_last_advance_fused(const cmfp_context* ctx);

/* Where the Strang source half steps (MFP_advance.cpp:146-197, 235-281) of the last fused
 * cmfp_advance ran: bit 0 set = stage A inside the CTU kernels (applied to every cell they load:
 * FAST arithmetic, 2-D, explicit plasma5 / Lorentz over two species, block without physical faces;
 * CMFP_SRC_PROLOGUE=0 in the environment keeps it a launch of its own), bit 1 set = stage C inside
 * the field's CTU kernel (CMFP_SRC_EPILOGUE=0 likewise).  0: launches of their own.  Diagnostic, like
 * cmfp_last_advance_fused; the reference has one code path (pointwise loops in
 * Plasma5::calc_time_derivative, MFP_plasma5.cpp:73-244). */
int cmfp_last_source_fusion(const cmfp_context* ctx);

/* Directions (bit d) of the last fused cmfp_advance whose ghost cells were not filled because the
 * block spans a periodic direction alone and every CTU state ran on a 2-D marching kernel, which
 * wraps its loads itself (the periodic part of FillPatch, MFP_action.cpp:59, folded into the
 * kernels; CMFP_MARCH_WRAP=0 in the environment keeps the fill launches).  The ghost cells of
 * those directions then hold stale values; cmfp_fill_patch fills them on request.  Diagnostic. */
int cmfp_last_wrapped_dims(const cmfp_context* ctx);

/* Per-kernel device timing: HIP events recorded on the context's stream around
 * every launch of the fused kernels while enabled (the reference's nearest
 * analogue is its per-box wall-time cost, MFP_CTU_hydro.cpp:143,529-530).
 * cmfp_profile_read synchronises, returns launches and summed milliseconds of
 * one slot since the last read, and resets the slot. */
typedef enum {
    CMFP_PROF_CTU_HYDRO = 0,
    CMFP_PROF_CTU_FIELD = 1,
    CMFP_PROF_SOURCE = 2,
    CMFP_PROF_FILL = 3,
    CMFP_PROF_NSLOTS = 4
} cmfp_prof_slot;
int cmfp_profile_enable(cmfp_context* ctx, int on);
int cmfp_profile_read(cmfp_context* ctx, int slot, long long* launches, double* total_ms);

/* Measurement aid (no counterpart in the reference): the HBM bandwidth a device-to-device copy attains on
 * `device` - 16 bytes per lane and access, read + write bytes counted, the best of three grid sizes, `reps`
 * passes over two private buffers of `nbytes` each - in GB/s.  bench.py reports it beside the 8 TB/s spec. */
int cmfp_hbm_copy_probe(int device, void* stream, long long nbytes, int reps, double* gbs);

/* ---- Ghost cells: the FillPatch of Action::get_data (MFP_action.cpp:59) ->
 * FillBC::operator() -> fab_filcc (modules/boundary/MFP_fillbc.cpp:147-172).
 * Dimension d of the new-time data of `state` is filled over the full extent
 * of the lower dimensions' ghosts.  Where this block spans the whole domain in
 * d the library fills it itself (periodic wrap or the bc[] rule).  Where the
 * domain is split in d between ranks, the library calls the registered
 * exchange callback, which must move the packed faces between neighbours
 * (cmfp_halo_pack / cmfp_halo_unpack give and take contiguous device
 * buffers); physical faces on the outer ranks are still filled by the
 * library. */
typedef int (*cmfp_exchange_fn)(void* user, cmfp_context* ctx, int state, int dim);
int cmfp_set_exchange_callback(cmfp_context* ctx, cmfp_exchange_fn fn, void* user);
/* Batched exchanges (default off): inside cmfp_advance the fused Strang step fills the ghost cells
 * of all states together and, with this set, calls the callback once per split dimension with
 * state = CMFP_ALL_STATES; the callback then packs every state of the context into one message
 * per neighbour (one exchange per dimension and step instead of one per state: the reference's
 * FillPatch is per state, MFP_action.cpp:59; the data moved is the same). */
#define CMFP_ALL_STATES (-1)
int cmfp_set_exchange_batched(cmfp_context* ctx, int on);
/* Halo/compute overlap inside cmfp_advance (fused 2-D Strang step).  1 (default): where the block is
 * split between ranks, the flux update of the interior tiles runs on a second, library-owned stream
 * while the ghost layers are packed, exchanged (callback) and unpacked on the context's stream; the
 * frame of boundary tiles follows the fill.  Where the flux launches are not split (3-D, or no interior
 * tiles) the first source half step is: the cells within ng of a face first, the interior on the second
 * stream while the ghost layers travel.  0: one launch per state after the fill.  2: split the
 * launches even on an unsplit block (testing).  Results are identical in all modes.  The exchange
 * callback must only enqueue work on, or synchronise with, the context's stream. */
int cmfp_set_overlap(cmfp_context* ctx, int mode);
int cmfp_fill_patch(cmfp_context* ctx, int state);
/* Cross-rank time-step minimum for cmfp_advance_steps: the ParallelDescriptor::ReduceRealMin of
 * MFP::estTimeStep (MFP_timestep.cpp:18), without the host.  After each step's device-side
 * computeNewDt the library calls fn with the device address of the double holding this rank's next
 * dt; fn must replace it, in stream order on the context's stream, by the minimum over all ranks
 * (e.g. an RCCL all-reduce(MIN) of one double).  min_r min(cfl*est_r, 1.1*dt) = min(cfl*min_r est_r,
 * 1.1*dt): the reference's value.  With a callback registered cmfp_advance_steps also runs on blocks
 * that are split between ranks.  cmfp_copy_async moves a few bytes device-to-device on the context's
 * stream (for bindings whose collectives need their own buffer). */
typedef int (*cmfp_dt_reduce_fn)(void* user, cmfp_context* ctx, double* device_dt);
int cmfp_set_dt_reduce_callback(cmfp_context* ctx, cmfp_dt_reduce_fn fn, void* user);
int cmfp_copy_async(cmfp_context* ctx, void* dst, const void* src, long long nbytes);
/* Number of doubles in one packed face of `state` normal to `dim`. */
long long cmfp_halo_count(const cmfp_context* ctx, int state, int dim);
/* side 0: pack the ng valid layers next to the low face (to send to the low
 * neighbour); side 1: next to the high face.  Unpack side 0 writes the low
 * ghost layers, side 1 the high ghost layers.  dbuf is device memory. */
int cmfp_halo_pack(cmfp_context* ctx, int state, int dim, int side, double* dbuf);
int cmfp_halo_unpack(cmfp_context* ctx, int state, int dim, int side, const double* dbuf);

/* ---- Device-side transport between ranks: the FillPatch exchange (MFP_action.cpp:59) and the ReduceRealMin of
 * MFP::estTimeStep (MFP_timestep.cpp:18) without the host on the per-step path.  Every rank owns an arena of
 * device memory (receive slots for each inter-rank face, progress words, dt slots) that the other ranks map -
 * across processes with HIP IPC (one process per GPU over xGMI, or several processes sharing one GPU), inside one
 * process with the plain device pointer - and write into directly: pack kernel into the neighbour's slot, a
 * system-scope release of its progress word, a bounded device-side wait on the receiver, unpack; the dt minimum
 * likewise (every rank writes its word into every other rank's slot).  With the transport enabled the exchange
 * callback and the dt-reduce callback are no longer called and cmfp_advance_steps runs on split blocks.
 *   1. cmfp_peer_init on every rank: allocates and zeroes the arena, returns its IPC handle
 *      (CMFP_IPC_HANDLE_BYTES bytes = hipIpcMemHandle_t) and, for same-process use, its device address;
 *   2. the caller all-gathers the handles (MPI_Allgather / torch.distributed: set-up only);
 *   3. cmfp_peer_open for EVERY other rank (halos go to neighbours, the dt word to all), cmfp_peer_set_neighbour
 *      for every inter-rank face (periodic wrap included; -1 on physical faces);
 *   4. a barrier of the caller (every rank has zeroed and mapped), then cmfp_peer_enable(ctx, 1).
 * All ranks must make the same sequence of calls afterwards (they do: the step is SPMD).  A wait that sees no
 * partner for CMFP_PEER_TIMEOUT seconds (environment, default 20) gives up: the kernels still run to completion and
 * the next synchronising call returns CMFP_ERR_HIP.  At most 64 ranks. */
#define CMFP_IPC_HANDLE_BYTES 64
int cmfp_peer_init(cmfp_context* ctx, int rank, int world, void* ipc_handle_out /* [CMFP_IPC_HANDLE_BYTES] or NULL */,
                   void** arena_out /* or NULL */);
int cmfp_peer_open(cmfp_context* ctx, int peer_rank, const void* ipc_handle, void* same_process_arena);
int cmfp_peer_set_neighbour(cmfp_context* ctx, int dim, int side, int rank);
int cmfp_peer_enable(cmfp_context* ctx, int on);
int cmfp_peer_enabled(const cmfp_context* ctx);

/* ---- Leaf numerics on device, batched over n independent inputs (SoA:
 * component c of item i at [c*n + i]); device pointers.  They run the very
 * device functions the update kernels call, so the leaves can be pinned
 * against known-answer vectors directly.
 * Reconstruction::get_slope / get_face_values (MFP_reconstruction.H:17-18):
 * stencil has stencil_length(recon) comps. */
int cmfp_leaf_reconstruct(int recon, int fp_mode, long long n, const double* stencil,
                          double* slope, double* lo, double* hi, void* stream);
/* RiemannSolver::solve (MFP_eulerian_riemann.H:23): hydro L,R have 8 comps
 * {rho,u,v,w,p,T,gamma,cp}, F 5; field L,R,F have 10 (c0 = lightspeed,
 * ch = div_speed). */
int cmfp_leaf_riemann(int flux, int fp_mode, long long n, const double* L, const double* R,
                      double* F, double c0, double ch, double shock_threshold, void* stream);
/* HydroGas::cons2prim / prim2cons / get_speed_from_cons
 * (MFP_thermally_perfect_gas.cpp:180-246,294-317): U 5 comps, Q 8, speeds 3. */
int cmfp_leaf_cons2prim(int fp_mode, long long n, double mass, double gamma, const double* U,
                        double* Q, double* speeds, void* stream);
int cmfp_leaf_prim2cons(int fp_mode, long long n, double mass, double gamma, const double* Q,
                        double* U, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CERBERUS_MFP_H */
