// mfp_march.hip — wave-marching fused corner-transport-upwind update of a hydro state (2-D).
//
// Same work as k_ctu2d of mfp_fused.hip — HydroCTU::calc_spatial_derivative
// (MFP_CTU_hydro.cpp:142-531) + `new += dU` (MFP_advance.cpp:220-230) in one launch — organised so
// that nothing is computed twice: one wavefront owns a strip of 64 columns and marches along y, one
// row per iteration, every lane carrying its column's pipeline in registers:
//
//   iteration r:  U(r+1) -> primitives                         (prefetched one row ahead)
//                 reconstruction + half-step trace of row r    (x-stencil: the neighbouring lanes,
//                                                               y-stencil: rows r-1, r, r+1 in registers)
//                 Riemann sweep 1: F_x(r), F_y(r)
//                 transverse correction: y-face states of row r with F_x(r),
//                                        x-face states of row r-1 with F_y(r-1) - F_y(r)
//                 domain-face BCs, Riemann sweep 2: F'_y(r), F'_x(r-1)
//                 divergence of row r-1, Uout = Uin + dU, optional CFL reduction
//
// x-neighbours talk through DPP wave shifts (register to register); there is no LDS and no barrier.
// A strip updates 60 of its 64 columns (two columns of stencil + transverse ring on each side), a
// segment of S rows costs S + 2 iterations: 92 % of the lanes' work lands in updated cells against
// 77 % of the 16 x 16 tile kernel (which also converts the 18 x 18 halo to primitives per tile).
// Arithmetic per value is the same sequence of device functions as the staged and tiled kernels:
// STRICT results are identical to them bit for bit.
#include <cstring>

#include "mfp_march.h"
#include "mfp_source.h"

namespace CMFP_NS {
static constexpr int MW = 60;  // columns a strip updates

// The states one launch updates: one, or two species of the same physics (reconstruction, Riemann solver,
// boundary rules) whose waves share the launch — on blocks of <= 1024^2 cells one state's waves leave
// most SIMDs idle or alone, and two launches in a row cost twice the march of one.
template <class Phys>
struct MarchStates {
    int n;
    typename Phys::Params P[2];
    const double* Uin[2];
    double* Uout[2];
    unsigned long long* red[2];
};

// PRO: Strang source stage A (explicit plasma5 / Lorentz over the launch's two species) applied to every row as
// it is loaded (SourcePro, mfp_source.h): the wave also reads the other species' momentum and D, B of the field
// (14 values per cell instead of 5) and the stage-A values a row's update starts from are rebuilt from its
// primitives instead of being read again (FAST; STRICT carries the conserved stage-A values of two rows instead, which
// keeps the update bit-identical to the separate launch).
// TAB: Dirichlet data on physical faces (MarchTab): the ghost cells beyond such a face get the tabulated primitives
// in place of their own (EulerianState::update_boundary_cells MFP_eulerian.cpp:435-511), the outside face states the
// values at the face centres, once before sweep 1 — kept, like the reference keeps them in its face arrays — and
// again after the fill rules before sweep 2 (face_bc :838-877).  Single-state launches of blocks with physical faces.
// NT: passive scalars (HydroState::n_tracers, species of equal mass / charge / gamma): the mass fractions alpha_t =
// tracer_t / rho ride through the same pipeline — limited slopes and half-step trace like every primitive
// (MFP_hydro.cpp:982-995), the tracer fluxes of HLLC / HLLE from the side and wave speeds of the hydro solve
// (MFP_hllc.cpp:100-183, MFP_hydro_hlle.cpp:75-118), transverse corrections as tracer = alpha rho, += k dF,
// alpha = tracer / rho' (prim2cons / cons2prim, MFP_thermally_perfect_gas.cpp:212-214,243-245).  These instances
// hold about 60 more values per lane and run at one wave per SIMD.  The source prologue leaves them alone.
template <class Phys, bool HAS_BC, bool REDUCE, bool PRO, bool TAB = false, int NT = 0>
__global__ __launch_bounds__(64, (NT || (PRO && Phys::FLUX == 2)) ? 1 : 2) void k_ctu_march(Geo g, MarchStates<Phys> S, double dt, MarchBC bc, int* err, MarchSel sel,
                                                               SourcePro pro, MarchTab tab) {
    static_assert(!TAB || (HAS_BC && !PRO), "face tables: blocks with physical faces, no source prologue");
    static_assert(NT == 0 || !TAB, "passive scalars: no face tables");
    constexpr int NTA = NT ? NT : 1;  // array lengths (no zero-length arrays)
    constexpr int RECON = Phys::RECON;
    static_assert(ReconTraits<RECON>::SL == 3 && Phys::IS_HYDRO, "three-point hydro reconstructions");
    dt = launch_dt(g, dt);  // device-resident time step (cmfp_advance_steps)
    const int lane = threadIdx.x;
    // XCD-aware order (see mfp_fused.hip): neighbouring strips, which share stencil columns, run on one XCD
    const int total = sel.nwaves * S.n;
    const int per = (total + 7) >> 3;
    int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (w >= total) return;
    const int which = w >= sel.nwaves ? 1 : 0;  // workgroup-uniform: the state this wave updates
    w -= which * sel.nwaves;
    const typename Phys::Params P = S.P[which];
    const double* __restrict__ Uin = S.Uin[which];
    double* __restrict__ Uout = S.Uout[which];
    unsigned long long* red = S.red[which];
    MarchRect R = sel.r[0];
#pragma unroll
    for (int t = 1; t < 4; ++t)
        if (t < sel.nrect && w >= sel.r[t].first) R = sel.r[t];
    w -= R.first;
    const int segi = w / R.ns, strip = w - segi * R.ns;
    const int n0 = g.n[0], n1 = g.n[1];
    const int i = R.ilo + strip * MW - 2 + lane;  // this lane's column (>= -2)
    int ic = i < R.ihi + 2 ? i : R.ihi + 2;  // lanes past the rectangle's stencil reach re-read its last column
    if (sel.wrap & 1) ic = wrap_index(ic, n0);  // periodic in x: the ghost columns are the valid columns of the far side
    const int ja = R.jlo + segi * sel.seg;
    const int jb = (ja + sel.seg - 1 < R.jhi) ? ja + sel.seg - 1 : R.jhi;
    const bool inner = lane >= 2 && lane <= 61;
    const bool upd = inner && i <= R.ihi;
    const double gamma = Phys::gamma_of(P), gm1 = gamma - 1.0;
    const double kx = uni(0.5 * dt / g.dx[0]), ky = uni(0.5 * dt / g.dx[1]);  // wave-uniform: scalar registers
    const double dtdx = uni(dt / g.dx[0]), dtdy = uni(dt / g.dx[1]);
    const long col = g.at(ic, 0, 0);
    bool bad = false;
    constexpr int NL = PRO ? 14 : 5;  // values of a cell in flight
    const double* __restrict__ Uoth = PRO ? S.Uin[1 - which] : nullptr;
    const double* __restrict__ Fin = PRO ? pro.fld : nullptr;
    const ProArgs pa = pro_args(pro, which);
    const double hdt = 0.5 * dt;
    const SourceCoef kf = source_coef(pa, hdt);

    auto load_row = [&](int row, double* U) {
        if (sel.wrap & 2) row = wrap_index(row, n1);
        const long c = col + (long)row * g.sj;
#pragma unroll
        for (int n = 0; n < 5; ++n) U[n] = Uin[c + n * g.sc];
    };
    auto load_row_tr = [&](int row, double* T) {
        if constexpr (NT > 0) {
            if (sel.wrap & 2) row = wrap_index(row, n1);
            const long c = col + (long)row * g.sj;
#pragma unroll
            for (int t = 0; t < NT; ++t) T[t] = Uin[c + (5 + t) * g.sc];
        }
    };
    // alpha = tracer * rhoinv (MFP_thermally_perfect_gas.cpp:212-214)
    auto to_alpha = [&](const double* U, const double* T, double* A) {
        if constexpr (NT > 0) {
            const double rhoinv = f_rcp(U[0]);
#pragma unroll
            for (int t = 0; t < NT; ++t) A[t] = T[t] * rhoinv;
        }
    };
    // what stage A needs beside the species' own values: the other species' momentum, D and B
    auto load_row_pro = [&](int row, double* U) {
        if constexpr (PRO) {
            if (sel.wrap & 2) row = wrap_index(row, n1);
            const long c = col + (long)row * g.sj;
#pragma unroll
            for (int n = 0; n < 3; ++n) U[5 + n] = Uoth[c + (1 + n) * g.sc];
#pragma unroll
            for (int n = 0; n < 6; ++n) U[8 + n] = Fin[c + n * g.sc];
        }
    };
    // [stage A,] conserved -> {rho, u, v, w, eps = p / (gamma - 1)} (MFP_hydro.cpp:847-862)
    auto to_prim = [&](double* U, double* Q, int row) {
        if constexpr (PRO) {
            ProMask m{0u, 0u};
            bool any = false;
            if constexpr (HAS_BC) {  // ghost cells of a physical face: stage A of the valid cell they mirror, signs set again
                m = pro_mask(pro, i, row, n0, n1);
                any = __builtin_amdgcn_ballot_w64((m.h | m.f) != 0u) != 0;
                if (any) {
#pragma unroll
                    for (int n = 0; n < 5; ++n) U[n] = pro_flip(U[n], m.h, n);
#pragma unroll
                    for (int n = 0; n < 3; ++n) U[5 + n] = pro_flip(U[5 + n], m.h, 1 + n);
#pragma unroll
                    for (int n = 0; n < 6; ++n) U[8 + n] = pro_flip(U[8 + n], m.f, n);
                }
            }
            source_prologue_species(pa, kf, hdt, U, U + 5, U + 8);
            if constexpr (HAS_BC) {
                if (any) {
#pragma unroll
                    for (int n = 0; n < 5; ++n) U[n] = pro_flip(U[n], m.h, n);
                }
            }
        }
        if (!Phys::cons2prim(P, U, Q)) bad = true;
        if constexpr (TAB) {  // ghost cells of faces with Dirichlet data, in the order of State::get_bc_limits
            const int gi = i + g.ng[0], gj = row + g.ng[1];
            if (i < n0 + g.ng[0]) {
                if (bc.touch[0][0] && i < 0) {
#pragma unroll
                    for (int n = 0; n < 5; ++n)
                        if (tab.cell[0][0][n]) Q[n] = tab.cell[0][0][n][gi + (long)gj * g.ng[0]];
                }
                if (bc.touch[0][1] && i >= n0) {
#pragma unroll
                    for (int n = 0; n < 5; ++n)
                        if (tab.cell[0][1][n]) Q[n] = tab.cell[0][1][n][(i - n0) + (long)gj * g.ng[0]];
                }
                if (bc.touch[1][0] && row < 0) {
#pragma unroll
                    for (int n = 0; n < 5; ++n)
                        if (tab.cell[1][0][n]) Q[n] = tab.cell[1][0][n][gi + (long)gj * g.ext[0]];
                }
                if (bc.touch[1][1] && row >= n1) {
#pragma unroll
                    for (int n = 0; n < 5; ++n)
                        if (tab.cell[1][1][n]) Q[n] = tab.cell[1][1][n][gi + (long)(row - n1) * g.ext[0]];
                }
            }
        }
        Q[H_P] = f_div(Q[H_P], gm1);
    };

    double Qa[5], Qb[5], Un[NL];
    double Aa[NTA], Ab[NTA], Tn[NTA];  // passive scalars: alpha of rows r - 1 and r, conserved tracers of row r + 1
    double Sa[5], Sb[5];               // STRICT with the prologue: conserved stage-A values of rows r - 1 and r
    {
        double U[NL], T[NTA];
        load_row(ja - 2, U);
        load_row_pro(ja - 2, U);
        load_row_tr(ja - 2, T);
        to_alpha(U, T, Aa);
        to_prim(U, Qa, ja - 2);
#pragma unroll
        for (int n = 0; n < 5; ++n) Sa[n] = U[n];
        load_row(ja - 1, U);
        load_row_pro(ja - 1, U);
        load_row_tr(ja - 1, T);
        to_alpha(U, T, Ab);
        to_prim(U, Qb, ja - 1);
#pragma unroll
        for (int n = 0; n < 5; ++n) Sb[n] = U[n];
        load_row(ja, Un);
        load_row_pro(ja, Un);
        load_row_tr(ja, Tn);
    }
    // carried from the previous iteration (row r - 1)
    double xlo_p[5], xhi_p[5];  // traced x-face states, not yet corrected
    double yhiU_p[5];           // traced high y-face state (sweep 1)
    double yhiC_p[5];           // the same, transversally corrected (sweep 2)
    double fy_p[5], fy2_p[5];   // F_y and F'_y on the low y-face
#pragma unroll
    for (int n = 0; n < 5; ++n) xlo_p[n] = xhi_p[n] = yhiU_p[n] = yhiC_p[n] = fy_p[n] = fy2_p[n] = 0.0;
    // the same for the mass fractions and the tracer fluxes
    double axlo_p[NTA], axhi_p[NTA], ayhiU_p[NTA], ayhiC_p[NTA], tfy_p[NTA], tfy2_p[NTA];
#pragma unroll
    for (int t = 0; t < NTA; ++t) axlo_p[t] = axhi_p[t] = ayhiU_p[t] = ayhiC_p[t] = tfy_p[t] = tfy2_p[t] = 0.0;
    double red_speed = 2.2250738585072014e-308;  // std::numeric_limits<Real>::min(), MFP_eulerian.cpp:265

    // One row of the march.  The loop below runs it twice per trip: what a row hands to the next one then
    // alternates between two sets of registers instead of being copied (a value of row r is still in use
    // while row r + 1 computes its own); a trip past the last row (odd row count) computes and drops a row.
    auto march_row = [&](const int r) __attribute__((always_inline)) {
        const bool live = r <= jb + 1;
        double Qc[5], Ac[NTA];
        to_alpha(Un, Tn, Ac);
        to_prim(Un, Qc, r + 1 < jb + 2 ? r + 1 : jb + 2);
#ifndef CMFP_FASTMATH
        double Sc[5];  // (PRO) row r + 1 after stage A, conserved
        if constexpr (PRO) {
#pragma unroll
            for (int n = 0; n < 5; ++n) Sc[n] = Un[n];
        }
#endif
        const int rn = r + 2 < jb + 2 ? r + 2 : jb + 2;  // rows past the segment's stencil reach are not touched
        load_row(rn, Un);
        load_row_tr(rn, Tn);
        double Uold[5];  // what the update of row r - 1 starts from (used once r - 1 >= ja)
        if constexpr (PRO) {
#ifdef CMFP_FASTMATH
            // the stage-A values of row r - 1 from its primitives (eps = E - ke); contractions written out so that
            // both copies of this row pipeline round alike
#pragma clang fp contract(off)
            const double q2 = fma(Qa[H_W], Qa[H_W], fma(Qa[H_V], Qa[H_V], Qa[H_U] * Qa[H_U]));
            Uold[0] = Qa[H_RHO], Uold[1] = Qa[H_U] * Qa[H_RHO], Uold[2] = Qa[H_V] * Qa[H_RHO], Uold[3] = Qa[H_W] * Qa[H_RHO];
            Uold[4] = fma(0.5 * Qa[H_RHO], q2, Qa[H_P]);
#else
            // STRICT: the conserved stage-A values themselves, carried from the iteration that converted the row
#pragma unroll
            for (int n = 0; n < 5; ++n) Uold[n] = Sa[n];
#endif
        } else {
            const long c = col + (long)(r - 1) * g.sj;
#pragma unroll
            for (int n = 0; n < 5; ++n) Uold[n] = Uin[c + n * g.sc];
        }
        double Told[NTA];
        load_row_tr(r - 1, Told);

        // ---- reconstruction + half-step trace of row r (HydroState::calc_reconstruction,
        //      calc_time_averaged_faces)
        double xlo[5], xhi[5], ylo[5], yhi[5];
        double axlo[NTA], axhi[NTA], aylo[NTA], ayhi[NTA];
        {
            double centre[5], speed[3];
#pragma unroll
            for (int n = 0; n < 5; ++n) centre[n] = Qb[n];
            centre[H_P] = centre[H_P] * gm1;  // restored pressure, MFP_hydro.cpp:1025-1035
            Phys::speed_from_prim(P, centre, speed);
            double st[3][5];
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                st[0][n] = lane_left(Qb[n]);
                st[1][n] = Qb[n];
                st[2][n] = lane_right(Qb[n]);
            }
            hydro_cell_faces<RECON>(gamma, st, xlo, xhi);
#pragma unroll
            for (int n = 0; n < 5; ++n) trace_faces(speed[0], kx, centre[n], xlo[n], xhi[n]);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                st[0][n] = Qa[n];
                st[2][n] = Qc[n];
            }
            hydro_cell_faces<RECON>(gamma, st, ylo, yhi);
#pragma unroll
            for (int n = 0; n < 5; ++n) trace_faces(speed[1], ky, centre[n], ylo[n], yhi[n]);
            if constexpr (NT > 0) {  // alpha -+ slope / 2, traced like every primitive (MFP_hydro.cpp:982-995)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    double sa[3] = {lane_left(Ab[t]), Ab[t], lane_right(Ab[t])};
                    double phi = recon_slope<RECON>(sa);
                    axlo[t] = Ab[t] - 0.5 * phi;
                    axhi[t] = Ab[t] + 0.5 * phi;
                    trace_faces(speed[0], kx, Ab[t], axlo[t], axhi[t]);
                    sa[0] = Aa[t], sa[2] = Ac[t];
                    phi = recon_slope<RECON>(sa);
                    aylo[t] = Ab[t] - 0.5 * phi;
                    ayhi[t] = Ab[t] + 0.5 * phi;
                    trace_faces(speed[1], ky, Ab[t], aylo[t], ayhi[t]);
                }
            }
        }

        if constexpr (TAB) {  // face_bc, do_all = false: the outside states of faces with Dirichlet data (valid face cells)
            if (r >= 0 && r < n1) {
                if (bc.touch[0][0] && i == -1) {  // cell -1's high face = L of face 0
#pragma unroll
                    for (int n = 0; n < 5; ++n)
                        if (tab.face[0][0][n]) xhi[n] = tab.face[0][0][n][r + g.ng[1]];
                }
                if (bc.touch[0][1] && i == n0) {  // cell n0's low face = R of face n0
#pragma unroll
                    for (int n = 0; n < 5; ++n)
                        if (tab.face[0][1][n]) xlo[n] = tab.face[0][1][n][r + g.ng[1]];
                }
            }
            if (i >= 0 && i < n0) {
                if (bc.touch[1][0] && r == -1) {
#pragma unroll
                    for (int n = 0; n < 5; ++n)
                        if (tab.face[1][0][n]) yhi[n] = tab.face[1][0][n][i + g.ng[0]];
                }
                if (bc.touch[1][1] && r == n1) {
#pragma unroll
                    for (int n = 0; n < 5; ++n)
                        if (tab.face[1][1][n]) ylo[n] = tab.face[1][1][n][i + g.ng[0]];
                }
            }
        }
        // ---- Riemann sweep 1 on the low faces of (i, r)
        double fx[5], fy[5] = {0, 0, 0, 0, 0};
        double tfx[NTA], tfy[NTA];
#pragma unroll
        for (int t = 0; t < NTA; ++t) tfx[t] = tfy[t] = 0.0;
        {
            double L[5];
#pragma unroll
            for (int n = 0; n < 5; ++n) L[n] = lane_left(xhi[n]);
            if constexpr (NT > 0) {
                double aL[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) aL[t] = lane_left(axhi[t]);
                Phys::template riemann_tracers<NT>(P, 0, L, xlo, aL, axlo, fx, tfx);
                if (r >= ja) Phys::template riemann_tracers<NT>(P, 1, yhiU_p, ylo, ayhiU_p, aylo, fy, tfy);
            } else {
                Phys::riemann(P, 0, L, xlo, fx);
                if (r >= ja) Phys::riemann(P, 1, yhiU_p, ylo, fy);
            }
        }
        double yhiU[5], ayhiU[NTA];
#pragma unroll
        for (int n = 0; n < 5; ++n) yhiU[n] = yhi[n];
#pragma unroll
        for (int t = 0; t < NTA; ++t) ayhiU[t] = NT ? ayhi[t] : 0.0;

        // ---- transverse corrections (EulerianState::correct_face_prim MFP_eulerian.cpp:1047-1188,
        //      per cell: each face state is corrected with the transverse fluxes of its own cell)
        {
            double dF[5];
#pragma unroll
            for (int n = 0; n < 5; ++n) dF[n] = fx[n] - lane_right(fx[n]);
            const bool col_in = inner && i >= 0 && i < n0;
            bool ok_hi, ok_lo;
            if constexpr (NT > 0) {
                double dT[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) dT[t] = tfx[t] - lane_right(tfx[t]);
                ok_hi = Phys::template correct_tracers<NT>(P, yhi, ayhi, kx, dF, dT);
                ok_lo = Phys::template correct_tracers<NT>(P, ylo, aylo, kx, dF, dT);
            } else {
                ok_hi = Phys::correct(P, yhi, kx, dF);
                ok_lo = Phys::correct(P, ylo, kx, dF);
            }
            if (live && col_in && ((!ok_hi && r < n1) || (!ok_lo && r >= 0 && r <= n1))) bad = true;
        }
        if (r > ja) {  // x-face states of row r - 1 use F_y(r - 1) - F_y(r)
            double dF[5];
#pragma unroll
            for (int n = 0; n < 5; ++n) dF[n] = fy_p[n] - fy[n];
            bool ok_hi, ok_lo;
            if constexpr (NT > 0) {
                double dT[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) dT[t] = tfy_p[t] - tfy[t];
                ok_hi = Phys::template correct_tracers<NT>(P, xhi_p, axhi_p, ky, dF, dT);
                ok_lo = Phys::template correct_tracers<NT>(P, xlo_p, axlo_p, ky, dF, dT);
            } else {
                ok_hi = Phys::correct(P, xhi_p, ky, dF);
                ok_lo = Phys::correct(P, xlo_p, ky, dF);
            }
            const bool lane_ok = lane >= 1 && lane <= 62;
            if (live && lane_ok && ((!ok_hi && i >= -1 && i < n0) || (!ok_lo && i >= 0 && i <= n0))) bad = true;
        }

        load_row_pro(rn, Un);  // requested here: half an iteration ahead is enough, and fewer values are live than at its start
        // ---- domain-face BCs (update_face_prim, do_all) + Riemann sweep 2
        double fy2[5] = {0, 0, 0, 0, 0};
        double tfy2[NTA];
#pragma unroll
        for (int t = 0; t < NTA; ++t) tfy2[t] = 0.0;
        if (r >= ja) {
            double L[5], Rr[5], aL[NTA], aR[NTA];
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                L[n] = yhiC_p[n];
                Rr[n] = ylo[n];
            }
#pragma unroll
            for (int t = 0; t < NTA; ++t) {
                aL[t] = ayhiC_p[t];
                aR[t] = NT ? aylo[t] : 0.0;
            }
            if (HAS_BC && bc.touch[1][0] && r == 0) {
#pragma unroll
                for (int n = 0; n < 5; ++n)
                    if (bc.rule[1][0][n]) L[n] = bc.rule[1][0][n] == FILL_ODD ? -Rr[n] : Rr[n];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (bc.rule[1][0][5 + t]) aL[t] = bc.rule[1][0][5 + t] == FILL_ODD ? -aR[t] : aR[t];
            }
            if (HAS_BC && bc.touch[1][1] && r == n1) {
#pragma unroll
                for (int n = 0; n < 5; ++n)
                    if (bc.rule[1][1][n]) Rr[n] = bc.rule[1][1][n] == FILL_ODD ? -L[n] : L[n];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (bc.rule[1][1][5 + t]) aR[t] = bc.rule[1][1][5 + t] == FILL_ODD ? -aL[t] : aL[t];
            }
            if constexpr (TAB) {
                if (i >= 0 && i < n0) {
                    if (bc.touch[1][0] && r == 0) {
#pragma unroll
                        for (int n = 0; n < 5; ++n)
                            if (tab.face[1][0][n]) L[n] = tab.face[1][0][n][i + g.ng[0]];
                    }
                    if (bc.touch[1][1] && r == n1) {
#pragma unroll
                        for (int n = 0; n < 5; ++n)
                            if (tab.face[1][1][n]) Rr[n] = tab.face[1][1][n][i + g.ng[0]];
                    }
                }
            }
            if constexpr (NT > 0)
                Phys::template riemann_tracers<NT>(P, 1, L, Rr, aL, aR, fy2, tfy2);
            else
                Phys::riemann(P, 1, L, Rr, fy2);
        }
        if (r > ja) {
            double L[5], Rr[5], fx2[5], aL[NTA], aR[NTA], tfx2[NTA];
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                L[n] = lane_left(xhi_p[n]);
                Rr[n] = xlo_p[n];
            }
#pragma unroll
            for (int t = 0; t < NTA; ++t) {
                aL[t] = NT ? lane_left(axhi_p[t]) : 0.0;
                aR[t] = axlo_p[t];
                tfx2[t] = 0.0;
            }
            if (HAS_BC && bc.touch[0][0] && i == 0) {
#pragma unroll
                for (int n = 0; n < 5; ++n)
                    if (bc.rule[0][0][n]) L[n] = bc.rule[0][0][n] == FILL_ODD ? -Rr[n] : Rr[n];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (bc.rule[0][0][5 + t]) aL[t] = bc.rule[0][0][5 + t] == FILL_ODD ? -aR[t] : aR[t];
            }
            if (HAS_BC && bc.touch[0][1] && i == n0) {
#pragma unroll
                for (int n = 0; n < 5; ++n)
                    if (bc.rule[0][1][n]) Rr[n] = bc.rule[0][1][n] == FILL_ODD ? -L[n] : L[n];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    if (bc.rule[0][1][5 + t]) aR[t] = bc.rule[0][1][5 + t] == FILL_ODD ? -aL[t] : aL[t];
            }
            if constexpr (TAB) {
                if (r - 1 >= 0 && r - 1 < n1) {
                    if (bc.touch[0][0] && i == 0) {
#pragma unroll
                        for (int n = 0; n < 5; ++n)
                            if (tab.face[0][0][n]) L[n] = tab.face[0][0][n][r - 1 + g.ng[1]];
                    }
                    if (bc.touch[0][1] && i == n0) {
#pragma unroll
                        for (int n = 0; n < 5; ++n)
                            if (tab.face[0][1][n]) Rr[n] = tab.face[0][1][n][r - 1 + g.ng[1]];
                    }
                }
            }
            if constexpr (NT > 0)
                Phys::template riemann_tracers<NT>(P, 0, L, Rr, aL, aR, fx2, tfx2);
            else
                Phys::riemann(P, 0, L, Rr, fx2);
            // ---- divergence of row r - 1 (calc_divergence MFP_eulerian.cpp:1614-1662), new += dU
            double Unew[5], Tnew[NTA];
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const double fxh = lane_right(fx2[n]);
                double du = 0.0;
                du += dtdx * (fx2[n] - fxh);
                du += dtdy * (fy2_p[n] - fy2[n]);
                Unew[n] = Uold[n] + du;
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const double fxh = lane_right(tfx2[t]);
                double du = 0.0;
                du += dtdx * (tfx2[t] - fxh);
                du += dtdy * (tfy2_p[t] - tfy2[t]);
                Tnew[t] = Told[t] + du;
            }
            if (upd && live) {
                const long c = col + (long)(r - 1) * g.sj;
#pragma unroll
                for (int n = 0; n < 5; ++n) Uout[c + n * g.sc] = Unew[n];
#pragma unroll
                for (int t = 0; t < NT; ++t) Uout[c + (5 + t) * g.sc] = Tnew[t];
                if constexpr (REDUCE) {  // EulerianState::get_allowed_time_step MFP_eulerian.cpp:290-296
                    double s[3];
                    hydro_speed_from_cons(P.gas, Unew, s);
                    red_speed = std_max(std_max(red_speed, s[0]), s[1]);
                }
            }
        }
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            Qa[n] = Qb[n];
            Qb[n] = Qc[n];
            xlo_p[n] = xlo[n];
            xhi_p[n] = xhi[n];
            yhiU_p[n] = yhiU[n];
            yhiC_p[n] = yhi[n];
            fy_p[n] = fy[n];
            fy2_p[n] = fy2[n];
        }
#ifndef CMFP_FASTMATH
        if constexpr (PRO) {
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                Sa[n] = Sb[n];
                Sb[n] = Sc[n];
            }
        }
#endif
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            Aa[t] = Ab[t];
            Ab[t] = Ac[t];
            axlo_p[t] = axlo[t];
            axhi_p[t] = axhi[t];
            ayhiU_p[t] = ayhiU[t];
            ayhiC_p[t] = ayhi[t];
            tfy_p[t] = tfy[t];
            tfy2_p[t] = tfy2[t];
        }
    };
    for (int r = ja - 1; r <= jb + 1; r += 2) {
        march_row(r);
        march_row(r + 1);
    }
    if (bad) *err = 1;
    if constexpr (REDUCE) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) red_speed = fmax(red_speed, __shfl_xor(red_speed, o, 64));
        if (lane == 0) atomicMax(&red[blockIdx.x & 255], (unsigned long long)__double_as_longlong(red_speed));
    }
}

template <class Phys>
static int launch_march(const Geo& g, int n, const StatePhys* const sp[2], const double* const Uin[2], double* const Uout[2],
                        double dt, const int touches[3][2], int* err, unsigned long long* const red[2], int part, const SourcePro* pro,
                        hipStream_t st, const MarchTab* tabs = nullptr) {
    if (g.ng[0] < 2 || g.ng[1] < 2) return 0;
    const bool has_int = g.n[0] >= 5 && g.n[1] >= 5;  // cells whose stencil footprint (+-2) stays inside valid cells
    MarchSel sel;
    if (Uin[0] && !march_select(g, part & 0xff, 2, MW, n, sel)) return 1;
    sel.wrap = part_wrap(part);
    part &= 0xff;
    const bool has_bc = touches[0][0] || touches[0][1] || touches[1][0] || touches[1][1] || g.bc_instance;
    if (pro) {  // the prologue: both species in the launch, stage C follows (no reduction here)
        if (n != 2 || red[0]) return 0;
    }
    if (tabs && (n != 1 || !has_bc || pro)) return 0;  // Dirichlet face data: one state, a block with physical faces, no prologue
    // passive scalars (species of equal mass / gamma): their own kernel instances; true mixtures (per-cell gamma), the
    // hybrid flux (no tracer flux in the reference, Q5), the prologue and face tables stay outside
    const int nt = sp[0]->hp.nt;
    if (nt && (sp[0]->hp.mix || Phys::FLUX == 2 || tabs || nt > 2)) return 0;
    if (n == 2 && sp[1]->hp.nt != nt) return 0;
    if (!Uin[0]) return part == 1 ? has_int : 1;
    const MarchBC bc = make_march_bc(*sp[0], touches);
    MarchStates<Phys> S;
    std::memset(&S, 0, sizeof(S));
    S.n = n;
    for (int k = 0; k < n; ++k) {
        S.P[k] = sp[k]->hp;
        S.Uin[k] = Uin[k];
        S.Uout[k] = Uout[k];
        S.red[k] = red[k];
    }
    const dim3 grid(((sel.nwaves * n + 7) / 8) * 8), block(64);
    SourcePro pr;
    std::memset(&pr, 0, sizeof(pr));
    MarchTab tb;
    std::memset(&tb, 0, sizeof(tb));
    if (tabs) {
        tb = *tabs;
        if (red[0])
            hipLaunchKernelGGL((k_ctu_march<Phys, true, true, false, true>), grid, block, 0, st, g, S, dt, bc, err, sel, pr, tb);
        else
            hipLaunchKernelGGL((k_ctu_march<Phys, true, false, false, true>), grid, block, 0, st, g, S, dt, bc, err, sel, pr, tb);
        return 1;
    }
    if constexpr (Phys::FLUX != 2) {
        if (nt && pro) {  // the source terms do not touch the mass fractions: the prologue instance with the tracer lanes
            pr = *pro;
            if (nt == 1)
                hipLaunchKernelGGL((k_ctu_march<Phys, true, false, true, false, 1>), grid, block, 0, st, g, S, dt, bc, err, sel, pr, tb);
            else
                hipLaunchKernelGGL((k_ctu_march<Phys, true, false, true, false, 2>), grid, block, 0, st, g, S, dt, bc, err, sel, pr, tb);
            return 1;
        }
        if (nt) {  // (always the instance with the face rules: half as many kernels to build)
#define CMFP_LAUNCH_T(RED, NTV) \
    hipLaunchKernelGGL((k_ctu_march<Phys, true, RED, false, false, NTV>), grid, block, 0, st, g, S, dt, bc, err, sel, pr, tb)
            if (red[0]) {
                if (nt == 1) CMFP_LAUNCH_T(true, 1); else CMFP_LAUNCH_T(true, 2);
            } else {
                if (nt == 1) CMFP_LAUNCH_T(false, 1); else CMFP_LAUNCH_T(false, 2);
            }
#undef CMFP_LAUNCH_T
            return 1;
        }
    }
#define CMFP_LAUNCH(BC, RED) hipLaunchKernelGGL((k_ctu_march<Phys, BC, RED, false>), grid, block, 0, st, g, S, dt, bc, err, sel, pr, tb)
    if (pro) {
        pr = *pro;
        if (has_bc)
            hipLaunchKernelGGL((k_ctu_march<Phys, true, false, true>), grid, block, 0, st, g, S, dt, bc, err, sel, pr, tb);
        else
            hipLaunchKernelGGL((k_ctu_march<Phys, false, false, true>), grid, block, 0, st, g, S, dt, bc, err, sel, pr, tb);
        return 1;
    }
    if (red[0]) {
        if (has_bc) CMFP_LAUNCH(true, true); else CMFP_LAUNCH(false, true);
    } else {
        if (has_bc) CMFP_LAUNCH(true, false); else CMFP_LAUNCH(false, false);
    }
#undef CMFP_LAUNCH
    return 1;
}

// Same contract as ctu_fused (mfp_fused.hip): part 0 all cells, 1 the interior, 2 the frame;
// Uin == nullptr probes coverage (part 0) / whether an interior exists (part 1).  Returns 0 where the
// option set is outside this kernel (the caller then uses the tiled kernel).
int ctu_march(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, int do_ctu, double dt,
              const int touches[3][2], int* err, unsigned long long* red, int part, hipStream_t st) {
    if (g.dim != 2 || sp.kind != 0 || !do_ctu) return 0;
    const StatePhys* const sps[2] = {&sp, nullptr};
    const double* const in[2] = {Uin, nullptr};
    double* const out[2] = {Uout, nullptr};
    unsigned long long* const reds[2] = {red, nullptr};
#define HY(R, F) \
    if (sp.recon == R && sp.flux == F) return launch_march<HydroF<R, F>>(g, 1, sps, in, out, dt, touches, err, reds, part, nullptr, st);
    HY(1, 0) HY(1, 1) HY(1, 2) HY(2, 0) HY(2, 1) HY(2, 2) HY(3, 0) HY(3, 1) HY(4, 0) HY(4, 1)
#undef HY
    return 0;
}

// ctu_march for a state with Dirichlet face data (MarchTab)
int ctu_march_tab(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, int do_ctu, double dt, const int touches[3][2],
                  int* err, unsigned long long* red, int part, const MarchTab& tab, hipStream_t st) {
    if (g.dim != 2 || sp.kind != 0 || !do_ctu || sp.hp.nt) return 0;
    if (!(touches[0][0] || touches[0][1] || touches[1][0] || touches[1][1])) return 0;
    const StatePhys* const sps[2] = {&sp, nullptr};
    const double* const in[2] = {Uin, nullptr};
    double* const out[2] = {Uout, nullptr};
    unsigned long long* const reds[2] = {red, nullptr};
#define HY(R, F) \
    if (sp.recon == R && sp.flux == F) return launch_march<HydroF<R, F>>(g, 1, sps, in, out, dt, touches, err, reds, part, nullptr, st, &tab);
    HY(1, 0) HY(1, 1) HY(2, 0) HY(2, 1) HY(3, 0) HY(3, 1) HY(4, 0) HY(4, 1)
#undef HY
    return 0;
}

// the cells ctu_march's interior launch (part 1) updates; false: ctu_march does not cover the state or has no interior
bool ctu_march_interior(const Geo& g, const StatePhys& sp, int do_ctu, IBox& box) {
    const int none[3][2] = {{0, 0}, {0, 0}, {0, 0}};
    if (!ctu_march(g, sp, nullptr, nullptr, do_ctu, 0.0, none, nullptr, nullptr, 0, nullptr)) return false;
    return march_interior(g, 2, box);
}

// Two hydro states in one launch (see MarchStates).  Returns 0 — nothing launched — unless both are covered
// by ctu_march with the same kernel instance and the same domain-face rules; Uin[0] == nullptr probes.
int ctu_march3d2(const Geo& g, const StatePhys* const sp[2], const double* const Uin[2], double* const Uout[2], int do_ctu, double dt,
                 const int touches[3][2], int* err, unsigned long long* const red[2], hipStream_t st);
int ctu_march2(const Geo& g, const StatePhys* const sp[2], const double* const Uin[2], double* const Uout[2], int do_ctu, double dt,
               const int touches[3][2], int* err, unsigned long long* const red[2], int part, const SourcePro* pro, hipStream_t st) {
    if (g.dim == 3) return part == 0 && !pro ? ctu_march3d2(g, sp, Uin, Uout, do_ctu, dt, touches, err, red, st) : 0;
    const StatePhys &a = *sp[0], &b = *sp[1];
    if (g.dim != 2 || a.kind != 0 || b.kind != 0 || !do_ctu) return 0;
    if (a.recon != b.recon || a.flux != b.flux || a.nc != b.nc || (red[0] == nullptr) != (red[1] == nullptr)) return 0;
    if (std::memcmp(a.bc, b.bc, sizeof(a.bc)) != 0) return 0;
    static const bool off = std::getenv("CMFP_MARCH_PAIR") && std::atoi(std::getenv("CMFP_MARCH_PAIR")) == 0;
    if (off) return 0;
    // (the hybrid HLLE/HLLC kernel with the prologue does not fit two waves per SIMD: that instance runs at one)
    static const bool hyb_pro = !(std::getenv("CMFP_HYBRID_PROLOGUE") && std::atoi(std::getenv("CMFP_HYBRID_PROLOGUE")) == 0);
    if (pro && a.flux == 2 && !hyb_pro) return 0;
#define HY(R, F) \
    if (a.recon == R && a.flux == F) return launch_march<HydroF<R, F>>(g, 2, sp, Uin, Uout, dt, touches, err, red, part, pro, st);
    HY(1, 0) HY(1, 1) HY(1, 2) HY(2, 0) HY(2, 1) HY(2, 2) HY(3, 0) HY(3, 1) HY(4, 0) HY(4, 1)
#undef HY
    return 0;
}

}  // namespace CMFP_NS
