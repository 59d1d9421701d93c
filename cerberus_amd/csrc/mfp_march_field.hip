// mfp_march_field.hip — wave-marching fused corner-transport-upwind update of the Maxwell state (2-D, O6,
// vacuum materials), optionally with the Strang source stage C as its epilogue.
//
// Same work as k_ctu2d<FieldF<O6>, ..., SRC> of mfp_fused.hip — HydroCTU::calc_spatial_derivative
// (MFP_CTU_hydro.cpp:142-531) + `new += dU` (MFP_advance.cpp:220-230) for the field, then stage C
// (MFP_advance.cpp:235-281) on the cells just updated — organised like the hydro marching kernel
// (mfp_march.hip): one wavefront owns a strip of 64 columns and marches along y, one row per iteration, so that
// nothing is reconstructed or solved twice and no thread idles on a tile ring:
//
//   iteration r:  row r + 3 (prefetched) enters the wave's window of six rows in LDS
//                 O6 interface values of row r:  x: I(i + 1/2) from the lanes i - 2 .. i + 3 of the window row,
//                                                y: I(r + 1/2) from the window rows r - 2 .. r + 3
//                 (the unlimited sixth-order interface value is single valued: the high face of one cell and
//                  the low face of the next are the same sum term by term, MFP_reconstruction.cpp:261-268,
//                  so each is computed once; bit-identical to two evaluations)
//                 half-step trace of row r, Riemann sweep 1: F_x(r), F_y(r)
//                 transverse corrections: y-faces of row r with F_x(r), x-faces of row r - 1 with F_y(r - 1) - F_y(r)
//                 domain-face BCs, Riemann sweep 2: F'_y(r), F'_x(r - 1)
//                 divergence of row r - 1, Uout = Uin + dU, [source stage C of the species at that cell]
//
// The window (6 rows x 8 components x 64 lanes = 24 KB per wave) is lane-private except for the x stencil, which
// reads the neighbouring lanes' entries of the current row; face states, fluxes and the carried rows stay in
// registers; x-neighbour face states and fluxes travel by DPP wave shifts.  No barrier between waves (one wave per
// workgroup).  A strip updates 56 of its 64 columns (87.5 % against 76.6 % useful threads of the 14 x 14 tiles).
//
// Applies where the material coefficients are mu = ep = 1 on the whole block (checked once on the device: they
// never change, F = 0 and no source touches them): their face states are then the same two constants everywhere
// — in STRICT the values the reference's reconstruction and trace give for a flat stencil (1 + 2^-52 before the
// trace for O6), in FAST exactly 1 (the vacuum shortcut of the Riemann solver) — so mu and ep are neither read nor
// rewritten: 8 components of traffic instead of 10.  Same device functions and operation order per value as the
// staged and tiled kernels: STRICT results are bit-identical to them and to the oracle.
#include <cstring>

#include "mfp_march.h"
#include "mfp_source.h"

namespace CMFP_NS {

static constexpr int FLD_MW = 56;   // columns a strip updates: lanes 4 .. 59
static constexpr int FLD_NR = 6;    // window rows
static constexpr int FLD_PAD = 8;   // doubles in front of and behind the window (edge lanes' x stencil reads)
#ifndef CMFP_FLD_DEPTH
#define CMFP_FLD_DEPTH 2
#endif
static constexpr int FLD_DEPTH = CMFP_FLD_DEPTH;  // iterations a row (and the epilogue's species) is requested ahead of its use
#ifndef CMFP_FLD_SPDEPTH
#define CMFP_FLD_SPDEPTH CMFP_FLD_DEPTH
#endif
static constexpr int SP_DEPTH = CMFP_FLD_SPDEPTH;  // the same for the epilogue's species alone (they are used at the end of an iteration)

// the O6 interface value between s3 and s4: recon_faces<5>'s `hi` of the cell holding s3 (= `lo` of the next)
CMFP_DEV double o6_interface(double s1, double s2, double s3, double s4, double s5, double s6) {
    return (37.0 / 60.0) * (s3 + s4) - (2.0 / 15.0) * (s2 + s5) + (1.0 / 60.0) * (s1 + s6);
}

// PRO: Strang source stage A applied to every row as it enters the window (SourcePro, mfp_source.h): the wave
// also reads {rho, mx, my, mz} of both species at the rows it loads (16 values per cell instead of 8).
template <bool HAS_BC, int SRC, bool PRO>
__global__ __launch_bounds__(64, 1) void k_fld_march(Geo g, FieldParams P, const double* __restrict__ Uin, double* __restrict__ Uout,
                                                      double dt, MarchBC bc, MarchSel sel, typename EpiArg<SRC>::type epi, SourcePro pro) {
    typedef FieldF<5> Phys;
    __shared__ double ring_[FLD_NR * 8 * 64 + 2 * FLD_PAD];
    double* const ring = ring_ + FLD_PAD;
    dt = launch_dt(g, dt);
    const int lane = threadIdx.x;
    const int per = (sel.nwaves + 7) >> 3;  // XCD-aware order: neighbouring strips on one XCD
    const int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    double src_speed[SRC > 0 ? SRC : 1];
    SrcDt src_dt;
#pragma unroll
    for (int s = 0; s < (SRC > 0 ? SRC : 1); ++s) src_speed[s] = DBL_MIN;
    if (w < sel.nwaves) {
        const MarchRect R = sel.r[0];
        const int segi = w / R.ns, strip = w - segi * R.ns;
        const int n0 = g.n[0], n1 = g.n[1];
        const int i = R.ilo + strip * FLD_MW - 4 + lane;  // this lane's column (>= -4)
        int ic = i < R.ihi + 4 ? i : R.ihi + 4;            // lanes past the stencil reach re-read the last column
        if (sel.wrap & 1) ic = wrap_index(ic, n0);         // periodic in x: the ghost columns are the far side's valid columns
        const int ja = R.jlo + segi * sel.seg;
        const int jb = (ja + sel.seg - 1 < R.jhi) ? ja + sel.seg - 1 : R.jhi;
        const bool upd = lane >= 4 && lane <= 59 && i <= R.ihi;
        const double speed = P.fastest;
        const double hx = uni(0.5 * dt / g.dx[0]), hy = uni(0.5 * dt / g.dx[1]);
        const double dtdx = uni(dt / g.dx[0]), dtdy = uni(dt / g.dx[1]);
        const long col = g.at(ic, 0, 0);
        const int row0 = ja - 4;  // the row in window slot 0

        // face states of mu, ep (components 8, 9) in x and y: constants of the launch
        double mlo[2], mhi[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#ifdef CMFP_FASTMATH
            mlo[d] = mhi[d] = 1.0;
#else
            const double flat[7] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
            recon_faces<5>(flat, mlo[d], mhi[d]);
            trace_faces(speed, d == 0 ? hx : hy, 1.0, mlo[d], mhi[d]);
#endif
        }

        constexpr int NL = PRO ? 16 : 8;  // values of a cell in flight
        const double* __restrict__ Sa = PRO ? pro.sp[0] : nullptr;
        const double* __restrict__ Sb = PRO ? pro.sp[1] : nullptr;
        const ProArgs pa = pro_args(pro, 0);
        const double hdt = 0.5 * dt;
        const SourceCoef kf = source_coef(pa, hdt);
        auto load_row = [&](int row, double* U) {
            if (sel.wrap & 2) row = wrap_index(row, n1);
            const long c = col + (long)row * g.sj;
#pragma unroll
            for (int n = 0; n < 8; ++n) U[n] = Uin[c + n * g.sc];
            if constexpr (PRO) {
#pragma unroll
                for (int n = 0; n < 4; ++n) U[8 + n] = Sa[c + n * g.sc];
#pragma unroll
                for (int n = 0; n < 4; ++n) U[12 + n] = Sb[c + n * g.sc];
            }
        };
        // a loaded row as the window holds it: after source stage A where that is this kernel's prologue
        auto enter = [&](double* U, int row) {
            if constexpr (PRO) {
                ProMask m{0u, 0u};
                bool any = false;
                if constexpr (HAS_BC) {  // ghost cells of a physical face: stage A of the valid cell they mirror, signs set again
                    m = pro_mask(pro, i, row, n0, n1);
                    any = __builtin_amdgcn_ballot_w64((m.h | m.f) != 0u) != 0;
                    if (any) {
#pragma unroll
                        for (int n = 0; n < 8; ++n) U[n] = pro_flip(U[n], m.f, n);
#pragma unroll
                        for (int n = 0; n < 4; ++n) U[8 + n] = pro_flip(U[8 + n], m.h, n), U[12 + n] = pro_flip(U[12 + n], m.h, n);
                    }
                }
                source_prologue_field(pa, kf, hdt, U, U + 8, U + 12);
                if constexpr (HAS_BC) {
                    if (any) {
#pragma unroll
                        for (int n = 0; n < 8; ++n) U[n] = pro_flip(U[n], m.f, n);
                    }
                }
            }
        };
        auto slot_of = [&](int row) { return (row - row0) % FLD_NR; };
        auto at = [&](int slot, int n, int l) -> double& { return ring[(slot * 8 + n) * 64 + l]; };

        // prologue: rows ja - 4 .. ja + 1 into the window; I_y(ja - 3/2) from them; rows ja + 2, ja + 3 in flight
        // (two rows ahead: at one or two waves per SIMD nothing else hides the latency of the loads)
        double Un[NL], Un2[NL];
        for (int s = 0; s < FLD_NR; ++s) {
            load_row(row0 + s, Un);
            enter(Un, row0 + s);
#pragma unroll
            for (int n = 0; n < 8; ++n) at(s, n, lane) = Un[n];
        }
        load_row(ja + 2, Un);
        if constexpr (FLD_DEPTH == 2) load_row(ja + 3 < jb + 4 ? ja + 3 : jb + 4, Un2);
        double Iy_p[8];
#pragma unroll
        for (int n = 0; n < 8; ++n)
            Iy_p[n] = o6_interface(at(0, n, lane), at(1, n, lane), at(2, n, lane), at(3, n, lane), at(4, n, lane), at(5, n, lane));

        // carried from the previous iteration (row r - 1)
        double xlo_p[8], xhi_p[8];  // traced x-face states, not yet corrected
        double yhiU_p[8];           // traced high y-face state (sweep 1)
        double yhiC_p[8];           // the same, transversally corrected (sweep 2)
        double fy_p[8], fy2_p[8];   // F_y and F'_y on the low y-face
#pragma unroll
        for (int n = 0; n < 8; ++n) xlo_p[n] = xhi_p[n] = yhiU_p[n] = yhiC_p[n] = fy_p[n] = fy2_p[n] = 0.0;
        // source epilogue: the species' values of the row that is updated two iterations later, requested now
        double sp_cur[SRC > 0 ? SRC : 1][5], sp_nxt[SRC > 0 ? SRC : 1][5], sp_nx2[SRC > 0 ? SRC : 1][5];
#pragma unroll
        for (int s = 0; s < (SRC > 0 ? SRC : 1); ++s)
#pragma unroll
            for (int n = 0; n < 5; ++n) sp_cur[s][n] = sp_nxt[s][n] = sp_nx2[s][n] = 1.0;

        // One row of the march.  The loop below runs it twice per trip: what a row hands to the next one then
        // alternates between two sets of registers instead of being copied; a trip past the last row (odd row
        // count) computes and drops a row.
        auto march_row = [&](const int r) __attribute__((always_inline)) {
            const bool live = r <= jb + 1;
            // ---- row r + 3 enters the window (it replaces row r - 3); row r + 5 is requested
            __syncthreads();  // every lane has finished reading the slot that is overwritten
            {
                const int sn = slot_of(r + 3);
                enter(Un, r + 3 < jb + 4 ? r + 3 : jb + 4);
#pragma unroll
                for (int n = 0; n < 8; ++n) at(sn, n, lane) = Un[n];
                if constexpr (FLD_DEPTH == 2) {
#pragma unroll
                    for (int n = 0; n < NL; ++n) Un[n] = Un2[n];
                    const int rn = r + 5 < jb + 4 ? r + 5 : jb + 4;  // rows past the segment's stencil reach are not touched
                    load_row(rn, Un2);
                } else {
                    const int rn = r + 4 < jb + 4 ? r + 4 : jb + 4;
                    load_row(rn, Un);
                }
                if constexpr (SRC > 0) {
                    if constexpr (SP_DEPTH == 2) {
                        if (upd && r + 1 >= ja && r + 1 <= jb) source_epilogue_load<SRC>(g, epi, col + (long)(r + 1) * g.sj, sp_nx2);
                    } else {
                        if (upd && r >= ja && r <= jb) source_epilogue_load<SRC>(g, epi, col + (long)r * g.sj, sp_nxt);
                    }
                }
            }
            __syncthreads();
            // The phases below are ordered so that what is carried from row r - 1 is consumed (and its registers
            // reused) as early as possible.  One wave per SIMD (launch bounds): the iteration's live state plus the
            // rows in flight need about 400 registers; at two waves per SIMD the kernel spilled to scratch and ran
            // 1.25 ms instead of 0.76 ms (4096^2, without the epilogue).
            const int s0 = slot_of(r - 2), s1 = slot_of(r - 1), s2 = slot_of(r), s3 = slot_of(r + 1), s4 = slot_of(r + 2),
                      s5 = slot_of(r + 3);
            // ---- A: y interface values, face states and half-step trace of row r
            double ylo[10], yhi[10];
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const double centre = at(s2, n, lane);
                const double Iy = o6_interface(at(s0, n, lane), at(s1, n, lane), centre, at(s3, n, lane), at(s4, n, lane), at(s5, n, lane));
                ylo[n] = Iy_p[n];
                yhi[n] = Iy;
                Iy_p[n] = Iy;
                trace_faces(speed, hy, centre, ylo[n], yhi[n]);
            }
            ylo[8] = ylo[9] = mlo[1];
            yhi[8] = yhi[9] = mhi[1];
            // ---- B: Riemann sweep 1 on the low y-face of (i, r); C: with it, the x-face states of row r - 1 get their
            //      transverse correction F_y(r - 1) - F_y(r) (EulerianState::correct_face_prim MFP_eulerian.cpp:1047-1188,
            //      per cell), then domain-face BCs (update_face_prim, do_all) and Riemann sweep 2 in x for row r - 1
            // (No test of r against ja here or below: the first two rows of a segment run the same arithmetic on
            //  what the carried registers hold — zeros, then row ja - 1 — which is finite for this linear solver;
            //  nothing of it reaches a stored value, stores start at r = ja + 1.)
            double fx2[10];
            {
                double fy[10];
                {
                    double L[10];
#pragma unroll
                    for (int n = 0; n < 8; ++n) L[n] = yhiU_p[n];
                    L[8] = L[9] = mhi[1];
                    Phys::riemann(P, 1, L, ylo, fy);
                }
#pragma unroll
                for (int n = 0; n < 8; ++n) yhiU_p[n] = yhi[n];
                {
                    double dF[8];
#pragma unroll
                    for (int n = 0; n < 8; ++n) dF[n] = fy_p[n] - fy[n];
                    Phys::correct(P, xhi_p, hy, dF);
                    Phys::correct(P, xlo_p, hy, dF);
                    double L[10], Rr[10];
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        L[n] = lane_left(xhi_p[n]);
                        Rr[n] = xlo_p[n];
                    }
                    L[8] = L[9] = mhi[0];
                    Rr[8] = Rr[9] = mlo[0];
                    if (HAS_BC && bc.touch[0][0] && i == 0) {
#pragma unroll
                        for (int n = 0; n < 10; ++n)
                            if (bc.rule[0][0][n]) L[n] = bc.rule[0][0][n] == FILL_ODD ? -Rr[n] : Rr[n];
                    }
                    if (HAS_BC && bc.touch[0][1] && i == n0) {
#pragma unroll
                        for (int n = 0; n < 10; ++n)
                            if (bc.rule[0][1][n]) Rr[n] = bc.rule[0][1][n] == FILL_ODD ? -L[n] : L[n];
                    }
                    Phys::riemann(P, 0, L, Rr, fx2);
                }
#pragma unroll
                for (int n = 0; n < 8; ++n) fy_p[n] = fy[n];
            }
            // ---- D: x interface values, face states, trace and Riemann sweep 1 of row r; the y-face states of row r
            //      get their transverse correction F_x(i) - F_x(i + 1)
            {
                double xlo[10], xhi[10];
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const double centre = at(s2, n, lane);
                    const double Ix = o6_interface(at(s2, n, lane - 2), at(s2, n, lane - 1), centre, at(s2, n, lane + 1),
                                                   at(s2, n, lane + 2), at(s2, n, lane + 3));
                    xlo[n] = lane_left(Ix);
                    xhi[n] = Ix;
                    trace_faces(speed, hx, centre, xlo[n], xhi[n]);
                }
                xlo[8] = xlo[9] = mlo[0];
                xhi[8] = xhi[9] = mhi[0];
                double fx[10], L[10];
#pragma unroll
                for (int n = 0; n < 8; ++n) L[n] = lane_left(xhi[n]);
                L[8] = L[9] = mhi[0];
                Phys::riemann(P, 0, L, xlo, fx);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    xlo_p[n] = xlo[n];
                    xhi_p[n] = xhi[n];
                }
                double dF[8];
#pragma unroll
                for (int n = 0; n < 8; ++n) dF[n] = fx[n] - lane_right(fx[n]);
                Phys::correct(P, yhi, hx, dF);
                Phys::correct(P, ylo, hx, dF);
            }
            // ---- E: domain-face BCs + Riemann sweep 2 on the low y-face of (i, r)
            double fy2[10];
            {
                double L[10], Rr[10];
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    L[n] = yhiC_p[n];
                    Rr[n] = ylo[n];
                }
                L[8] = L[9] = mhi[1];
                Rr[8] = Rr[9] = mlo[1];
                if (HAS_BC && bc.touch[1][0] && r == 0) {
#pragma unroll
                    for (int n = 0; n < 10; ++n)
                        if (bc.rule[1][0][n]) L[n] = bc.rule[1][0][n] == FILL_ODD ? -Rr[n] : Rr[n];
                }
                if (HAS_BC && bc.touch[1][1] && r == n1) {
#pragma unroll
                    for (int n = 0; n < 10; ++n)
                        if (bc.rule[1][1][n]) Rr[n] = bc.rule[1][1][n] == FILL_ODD ? -L[n] : L[n];
                }
                Phys::riemann(P, 1, L, Rr, fy2);
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) yhiC_p[n] = yhi[n];
            // ---- F: divergence of row r - 1 (calc_divergence MFP_eulerian.cpp:1614-1662), new += dU, [source stage C]
            {
                double Unew[10];
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const double fxh = lane_right(fx2[n]);
                    double du = 0.0;
                    du += dtdx * (fx2[n] - fxh);
                    du += dtdy * (fy2_p[n] - fy2[n]);
                    Unew[n] = at(s1, n, lane) + du;
                }
                Unew[8] = Unew[9] = 1.0;  // mu, ep: F = 0 (MFP_field_RH.cpp:33), vacuum block
                if (upd && live && r > ja) {
                    const long c = col + (long)(r - 1) * g.sj;
                    if constexpr (SRC > 0) source_epilogue_cell_pre<SRC>(g, epi, dt, c, Unew, sp_cur, src_speed, src_dt);
#pragma unroll
                    for (int n = 0; n < 8; ++n) Uout[c + n * g.sc] = Unew[n];
                }
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) fy2_p[n] = fy2[n];
            if constexpr (SRC > 0) {
#pragma unroll
                for (int s = 0; s < SRC; ++s)
#pragma unroll
                    for (int n = 0; n < 5; ++n) {
                        sp_cur[s][n] = sp_nxt[s][n];
                        if constexpr (SP_DEPTH == 2) sp_nxt[s][n] = sp_nx2[s][n];
                    }
            }
        };
        for (int r = ja - 1; r <= jb + 1; r += 2) {
            march_row(r);
            march_row(r + 1);
        }
    }
    if constexpr (SRC > 0) {
        __syncthreads();  // the window is free
        source_epilogue_reduce<SRC>(epi, src_speed, src_dt, ring, lane, 1, (int)(blockIdx.x & 255));
    }
}

// ---- The same update by a pair of waves per strip (round 3).  In two dimensions the vacuum Maxwell system with its
// cleaning potentials splits into two sets of four components that never meet in a flux: TE = {Dx, Dy, Bz, phi} and
// TM = {Dz, Bx, By, psi} (x faces pair (Dy, Bz), (Dz, By), (Dx, phi), (Bx, psi); y faces (Dz, Bx), (Dx, Bz), (Dy, phi),
// (By, psi): every pair lies inside one set).  k_fld_march carries all eight through one wave: 400 registers, one
// wave per SIMD, and a wave alone cannot cover its own instruction latencies (VALU active 58 % of its cycles).  Here a
// workgroup of two waves owns the strip: wave G runs interfaces, traces, Riemann sweeps, corrections and the
// divergence for its four components (half the registers: two waves per SIMD), and what couples the sets — source
// stage A on the rows that enter the window, stage C on the rows that leave — alternates between the two waves row
// by row through LDS, so no arithmetic is done twice.  Per value the same device functions in the same order as
// k_fld_march (the other set's slots of a Riemann solve hold zeros, and the compiler drops what only they feed).
// Result (4096^2 with prologue and epilogue, A/B on one box): 235 - 256 VGPRs, 28.8 KB of LDS per pair, bit-identical
// results, and 1.60 ms against 1.43 ms for k_fld_march — the launch moves 6.2 GB at 4.3 TB/s, 83 % of what a copy
// kernel reaches on this device, so it is the memory system and not the lone wave's latencies that bounds it, and the
// pair adds two barriers per row to that.  Kept selectable (CMFP_FIELD_PAIR=1) and tested; not the default.
CMFP_DEV constexpr int fgc(int G, int q) {  // component q of set G
    return G == 0 ? (q == 0 ? F_DX : q == 1 ? F_DY : q == 2 ? F_BZ : F_PHI) : (q == 0 ? F_DZ : q == 1 ? F_BX : q == 2 ? F_BY : F_PSI);
}
CMFP_DEV constexpr int fgrp(int n) { return (n == F_DX || n == F_DY || n == F_BZ || n == F_PHI) ? 0 : 1; }
CMFP_DEV constexpr int fidx(int n) {
    return n == F_DX ? 0 : n == F_DY ? 1 : n == F_BZ ? 2 : n == F_PHI ? 3 : n == F_DZ ? 0 : n == F_BX ? 1 : n == F_BY ? 2 : 3;
}
template <int G> struct FldSet { static constexpr int value = G; };

template <bool HAS_BC, int SRC, bool PRO>
__global__ __launch_bounds__(128, 2) void k_fld_pair(Geo g, FieldParams P, const double* __restrict__ Uin, double* __restrict__ Uout,
                                                      double dt, MarchBC bc, MarchSel sel, typename EpiArg<SRC>::type epi, SourcePro pro) {
    typedef FieldF<5> Phys;
    constexpr int RG = FLD_NR * 4 * 64;  // one set's window
    __shared__ double ring_[2 * RG + 2 * FLD_PAD + 8 * 64];
    double* const ring = ring_ + FLD_PAD;
    double* const xchg = ring + 2 * RG + FLD_PAD;  // the updated row, all eight components, for stage C
    dt = launch_dt(g, dt);
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int per = (sel.nwaves + 7) >> 3;  // XCD-aware order: neighbouring strips on one XCD
    const int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    double src_speed[SRC > 0 ? SRC : 1];
    SrcDt src_dt;
#pragma unroll
    for (int s = 0; s < (SRC > 0 ? SRC : 1); ++s) src_speed[s] = DBL_MIN;
    if (w < sel.nwaves) {
        const MarchRect R = sel.r[0];
        const int segi = w / R.ns, strip = w - segi * R.ns;
        const int n0 = g.n[0], n1 = g.n[1];
        const int i = R.ilo + strip * FLD_MW - 4 + lane;
        int ic = i < R.ihi + 4 ? i : R.ihi + 4;
        if (sel.wrap & 1) ic = wrap_index(ic, n0);
        const int ja = R.jlo + segi * sel.seg;
        const int jb = (ja + sel.seg - 1 < R.jhi) ? ja + sel.seg - 1 : R.jhi;
        const bool upd = lane >= 4 && lane <= 59 && i <= R.ihi;
        const double speed = P.fastest;
        const double hx = uni(0.5 * dt / g.dx[0]), hy = uni(0.5 * dt / g.dx[1]);
        const double dtdx = uni(dt / g.dx[0]), dtdy = uni(dt / g.dx[1]);
        const long col = g.at(ic, 0, 0);
        const int row0 = ja - 4;
        double mlo[2], mhi[2];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#ifdef CMFP_FASTMATH
            mlo[d] = mhi[d] = 1.0;
#else
            const double flat[7] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
            recon_faces<5>(flat, mlo[d], mhi[d]);
            trace_faces(speed, d == 0 ? hx : hy, 1.0, mlo[d], mhi[d]);
#endif
        }
        constexpr int NL = PRO ? 16 : 8;
        const double* __restrict__ Sa = PRO ? pro.sp[0] : nullptr;
        const double* __restrict__ Sb = PRO ? pro.sp[1] : nullptr;
        const ProArgs pa = pro_args(pro, 0);
        const double hdt = 0.5 * dt;
        const SourceCoef kf = source_coef(pa, hdt);
        auto load_row = [&](int row, double* U) {
            if (sel.wrap & 2) row = wrap_index(row, n1);
            const long c = col + (long)row * g.sj;
#pragma unroll
            for (int n = 0; n < 8; ++n) U[n] = Uin[c + n * g.sc];
            if constexpr (PRO) {
#pragma unroll
                for (int n = 0; n < 4; ++n) U[8 + n] = Sa[c + n * g.sc];
#pragma unroll
                for (int n = 0; n < 4; ++n) U[12 + n] = Sb[c + n * g.sc];
            }
        };
        auto enter = [&](double* U, int row) {
            if constexpr (PRO) {
                ProMask m{0u, 0u};
                bool any = false;
                if constexpr (HAS_BC) {
                    m = pro_mask(pro, i, row, n0, n1);
                    any = __builtin_amdgcn_ballot_w64((m.h | m.f) != 0u) != 0;
                    if (any) {
#pragma unroll
                        for (int n = 0; n < 8; ++n) U[n] = pro_flip(U[n], m.f, n);
#pragma unroll
                        for (int n = 0; n < 4; ++n) U[8 + n] = pro_flip(U[8 + n], m.h, n), U[12 + n] = pro_flip(U[12 + n], m.h, n);
                    }
                }
                source_prologue_field(pa, kf, hdt, U, U + 8, U + 12);
                if constexpr (HAS_BC) {
                    if (any) {
#pragma unroll
                        for (int n = 0; n < 8; ++n) U[n] = pro_flip(U[n], m.f, n);
                    }
                }
            }
        };
        auto slot_of = [&](int row) { return (row - row0) % FLD_NR; };
        // a row into both sets' windows
        auto put_row = [&](int slot, const double* U) {
#pragma unroll
            for (int n = 0; n < 8; ++n) ring[fgrp(n) * RG + (slot * 4 + fidx(n)) * 64 + lane] = U[n];
        };

        auto body = [&](auto GT) {
            constexpr int G = decltype(GT)::value;
            double* const ringG = ring + G * RG;
            auto at = [&](int slot, int q, int l) -> double& { return ringG[(slot * 4 + q) * 64 + l]; };
            auto expand = [&](const double* Q4, double m, double* Q10) {
#pragma unroll
                for (int n = 0; n < 8; ++n) Q10[n] = 0.0;
#pragma unroll
                for (int q = 0; q < 4; ++q) Q10[fgc(G, q)] = Q4[q];
                Q10[8] = Q10[9] = m;
            };
            auto pick = [&](const double* F10, double* F4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) F4[q] = F10[fgc(G, q)];
            };
            // the six rows of the prologue: this wave loads (and takes through stage A) those of its parity
            // (rows alternate between the waves: row `row` is loaded by wave ((row - row0) & 1) ^ 1, so that in every
            //  iteration one wave takes the entering row through stage A while the other finishes stage C of the row
            //  that left)
            double Un[NL];
            for (int s = G ^ 1; s < FLD_NR; s += 2) {
                load_row(row0 + s, Un);
                enter(Un, row0 + s);
                put_row(s, Un);
            }
            load_row(ja + 2 + (G ^ 1) < jb + 4 ? ja + 2 + (G ^ 1) : jb + 4, Un);  // this wave's next row
            __syncthreads();
            double Iy_p[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                Iy_p[q] = o6_interface(at(0, q, lane), at(1, q, lane), at(2, q, lane), at(3, q, lane), at(4, q, lane), at(5, q, lane));
            double xlo_p[4], xhi_p[4], yhiU_p[4], yhiC_p[4], fy_p[4], fy2_p[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xlo_p[q] = xhi_p[q] = yhiU_p[q] = yhiC_p[q] = fy_p[q] = fy2_p[q] = 0.0;
            double sp[SRC > 0 ? SRC : 1][5];
#pragma unroll
            for (int s = 0; s < (SRC > 0 ? SRC : 1); ++s)
#pragma unroll
                for (int n = 0; n < 5; ++n) sp[s][n] = 1.0;

            // stage C of row `row`, whose updated values the previous iteration left in the exchange row
            auto finish_row = [&](int row) {
                if constexpr (SRC > 0) {
                    if (upd && row >= ja && row <= jb) {
                        const long c = col + (long)row * g.sj;
                        double Unew[10];
#pragma unroll
                        for (int n = 0; n < 8; ++n) Unew[n] = xchg[n * 64 + lane];
                        Unew[8] = Unew[9] = 1.0;  // mu, ep: F = 0 (MFP_field_RH.cpp:33), vacuum block
                        source_epilogue_cell_pre<SRC>(g, epi, dt, c, Unew, sp, src_speed, src_dt);
#pragma unroll
                        for (int n = 0; n < 8; ++n) Uout[c + n * g.sc] = Unew[n];
                    }
                }
            };
            auto march_row = [&](const int r) __attribute__((always_inline)) {
                const bool live = r <= jb + 1;
                __syncthreads();  // both waves have finished reading the slot that is overwritten; the exchange row is written
                if ((((r + 3 - row0) & 1) ^ 1) == G) {
                    // row r + 3 enters the windows (it replaces row r - 3), row r + 5 is requested; the species of row
                    // r - 1, whose stage C is this wave's in the next iteration, are requested too
                    enter(Un, r + 3 < jb + 4 ? r + 3 : jb + 4);
                    put_row(slot_of(r + 3), Un);
                    load_row(r + 5 < jb + 4 ? r + 5 : jb + 4, Un);
                    if constexpr (SRC > 0) {
                        if (upd && r - 1 >= ja && r - 1 <= jb) source_epilogue_load<SRC>(g, epi, col + (long)(r - 1) * g.sj, sp);
                    }
                } else {
                    finish_row(r - 2);  // updated by the previous iteration
                }
                __syncthreads();
                const int s0 = slot_of(r - 2), s1 = slot_of(r - 1), s2 = slot_of(r), s3 = slot_of(r + 1), s4 = slot_of(r + 2),
                          s5 = slot_of(r + 3);
                // ---- A: y interface values, face states and half-step trace of row r
                double ylo[4], yhi[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const double centre = at(s2, q, lane);
                    const double Iy = o6_interface(at(s0, q, lane), at(s1, q, lane), centre, at(s3, q, lane), at(s4, q, lane), at(s5, q, lane));
                    ylo[q] = Iy_p[q];
                    yhi[q] = Iy;
                    Iy_p[q] = Iy;
                    trace_faces(speed, hy, centre, ylo[q], yhi[q]);
                }
                // ---- B, C: sweep 1 on the low y-face of (i, r); x-face states of row r - 1 corrected with F_y(r - 1) - F_y(r),
                //      domain-face rules, sweep 2 in x for row r - 1
                double fx2[4];
                {
                    double fy[4];
                    {
                        double L[10], Rr[10], F[10];
                        expand(yhiU_p, mhi[1], L);
                        expand(ylo, mlo[1], Rr);
                        Phys::riemann(P, 1, L, Rr, F);
                        pick(F, fy);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) yhiU_p[q] = yhi[q];
                    {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const double dF = fy_p[q] - fy[q];
                            xhi_p[q] += hy * dF;
                            xlo_p[q] += hy * dF;
                        }
                        double L4[4], L[10], Rr[10], F[10];
#pragma unroll
                        for (int q = 0; q < 4; ++q) L4[q] = lane_left(xhi_p[q]);
                        expand(L4, mhi[0], L);
                        expand(xlo_p, mlo[0], Rr);
                        if (HAS_BC && bc.touch[0][0] && i == 0) {
#pragma unroll
                            for (int n = 0; n < 10; ++n)
                                if (bc.rule[0][0][n]) L[n] = bc.rule[0][0][n] == FILL_ODD ? -Rr[n] : Rr[n];
                        }
                        if (HAS_BC && bc.touch[0][1] && i == n0) {
#pragma unroll
                            for (int n = 0; n < 10; ++n)
                                if (bc.rule[0][1][n]) Rr[n] = bc.rule[0][1][n] == FILL_ODD ? -L[n] : L[n];
                        }
                        Phys::riemann(P, 0, L, Rr, F);
                        pick(F, fx2);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) fy_p[q] = fy[q];
                }
                // ---- D: x interface values, face states, trace and sweep 1 of row r; y-face states corrected with F_x(i) - F_x(i + 1)
                {
                    double xlo[4], xhi[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double centre = at(s2, q, lane);
                        const double Ix = o6_interface(at(s2, q, lane - 2), at(s2, q, lane - 1), centre, at(s2, q, lane + 1), at(s2, q, lane + 2),
                                                       at(s2, q, lane + 3));
                        xlo[q] = lane_left(Ix);
                        xhi[q] = Ix;
                        trace_faces(speed, hx, centre, xlo[q], xhi[q]);
                    }
                    double fx[4];
                    {
                        double L4[4], L[10], Rr[10], F[10];
#pragma unroll
                        for (int q = 0; q < 4; ++q) L4[q] = lane_left(xhi[q]);
                        expand(L4, mhi[0], L);
                        expand(xlo, mlo[0], Rr);
                        Phys::riemann(P, 0, L, Rr, F);
                        pick(F, fx);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        xlo_p[q] = xlo[q];
                        xhi_p[q] = xhi[q];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double dF = fx[q] - lane_right(fx[q]);
                        yhi[q] += hx * dF;
                        ylo[q] += hx * dF;
                    }
                }
                // ---- E: domain-face rules + sweep 2 on the low y-face of (i, r)
                double fy2[4];
                {
                    double L[10], Rr[10], F[10];
                    expand(yhiC_p, mhi[1], L);
                    expand(ylo, mlo[1], Rr);
                    if (HAS_BC && bc.touch[1][0] && r == 0) {
#pragma unroll
                        for (int n = 0; n < 10; ++n)
                            if (bc.rule[1][0][n]) L[n] = bc.rule[1][0][n] == FILL_ODD ? -Rr[n] : Rr[n];
                    }
                    if (HAS_BC && bc.touch[1][1] && r == n1) {
#pragma unroll
                        for (int n = 0; n < 10; ++n)
                            if (bc.rule[1][1][n]) Rr[n] = bc.rule[1][1][n] == FILL_ODD ? -L[n] : L[n];
                    }
                    Phys::riemann(P, 1, L, Rr, F);
                    pick(F, fy2);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) yhiC_p[q] = yhi[q];
                // ---- F: divergence of row r - 1, new += dU, [stage C by the wave of the row's parity]
                {
                    double Un4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const double fxh = lane_right(fx2[q]);
                        double du = 0.0;
                        du += dtdx * (fx2[q] - fxh);
                        du += dtdy * (fy2_p[q] - fy2[q]);
                        Un4[q] = at(s1, q, lane) + du;
                    }
                    const long c = col + (long)(r - 1) * g.sj;
                    if constexpr (SRC > 0) {  // left for the other wave, which finishes the row in the next iteration
#pragma unroll
                        for (int q = 0; q < 4; ++q) xchg[fgc(G, q) * 64 + lane] = Un4[q];
                    } else {
                        if (upd && live && r > ja) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) Uout[c + fgc(G, q) * g.sc] = Un4[q];
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) fy2_p[q] = fy2[q];
            };
            int r = ja - 1;
            for (; r <= jb + 1; r += 2) {
                march_row(r);
                march_row(r + 1);
            }
            if constexpr (SRC > 0) {  // the row the last iteration (r - 1) updated
                __syncthreads();
                if ((((r + 3 - row0) & 1) ^ 1) != G) finish_row(r - 2);
            }
        };
        if (grp == 0)
            body(FldSet<0>{});
        else
            body(FldSet<1>{});
    }
    if constexpr (SRC > 0) {
        __syncthreads();  // the windows are free
        source_epilogue_reduce<SRC>(epi, src_speed, src_dt, ring, tid, 2, (int)(blockIdx.x & 255));
    }
}

// one flag word: 0 if every valid cell has mu == 1 and ep == 1 (set to 1 by any cell that has not); *flag must be
// zero before the launch
__global__ void k_fld_vacuum(Geo g, const double* __restrict__ U, int* flag) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nn = (long)g.n[0] * g.n[1] * g.n[2];
    if (t >= nn) return;
    const int i = (int)(t % g.n[0]), j = (int)((t / g.n[0]) % g.n[1]), k = (int)(t / ((long)g.n[0] * g.n[1]));
    const long c = g.at(i, j, k);
    if (U[c + F_MU * g.sc] != 1.0 || U[c + F_EP * g.sc] != 1.0) *flag = 1;
}
void field_vacuum_check(const Geo& g, const double* U, int* flag, hipStream_t st) {
    const long nn = (long)g.n[0] * g.n[1] * g.n[2];
    hipLaunchKernelGGL(k_fld_vacuum, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, g, U, flag);
}

// wave slots of the device for this kernel: 24 KB of LDS per wave -> six waves per CU
static int fld_wave_slots() {
    static int slots = 0;
    if (!slots) {
        int dev = 0;
        hipDeviceProp_t p;
        slots = 1536;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0)
            slots = p.multiProcessorCount * 6;
    }
    return slots;
}

// The field update of a vacuum block by the marching kernel; epi (may be null): the explicit Strang source
// stage C as its epilogue.  Returns 0 where it does not apply (the caller then uses the tiled kernel).
int ctu_march_field(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, int do_ctu, double dt,
                    const int touches[3][2], const SourceEpi* epi, int part, const SourcePro* pro, hipStream_t st) {
    static const int off = std::getenv("CMFP_FIELD_MARCH") ? std::atoi(std::getenv("CMFP_FIELD_MARCH")) == 0 : 0;
    if (off || g.dim != 2 || sp.kind != 1 || !do_ctu || sp.recon != 5 || sp.flux != 10 || !sp.vacuum || (part & 0xff) != 0) return 0;
    if (g.ng[0] < 4 || g.ng[1] < 4 || g.n[0] < 1 || g.n[1] < 1) return 0;
    if (epi && (epi->a.implicit || epi->a.coll.on || epi->ns < 1 || epi->ns > 2)) return 0;
    const bool has_bc = touches[0][0] || touches[0][1] || touches[1][0] || touches[1][1] || g.bc_instance;
    if (pro) {  // source stage A as the prologue: two species with the epilogue
        if (!epi || epi->ns != 2) return 0;
    }
    if (!Uin) return 1;  // coverage probe
    MarchSel sel;
    sel.nrect = 1;
    MarchRect& R = sel.r[0];
    R.ilo = 0, R.ihi = g.n[0] - 1, R.jlo = 0, R.jhi = g.n[1] - 1;
    R.ns = (g.n[0] + FLD_MW - 1) / FLD_MW;
    R.first = 0;
    // segments: one resident round of waves; a segment of S rows costs S + 2 iterations and a six-row prologue
    static const int seg_env = std::getenv("CMFP_FIELD_SEG") ? std::atoi(std::getenv("CMFP_FIELD_SEG")) : 0;
    int seg = seg_env;
    if (seg <= 0) {
        const int rounds = fld_wave_slots() * 4 / 6 / R.ns;  // one wave per SIMD (the kernel's register budget)
        seg = rounds > 0 ? (g.n[1] + rounds - 1) / rounds : g.n[1];
        // small blocks: short segments that put one wave on every SIMD beat long ones on a few (512^2: 6 rows
        // against 24, launch 0.10 -> see DESIGN.md section 3); the eight rows of start-up only cost where waves queue
        static const int seg_min = std::getenv("CMFP_FIELD_SEG_MIN") ? std::atoi(std::getenv("CMFP_FIELD_SEG_MIN")) : 6;
        if (seg < seg_min) seg = seg_min;
        seg += seg & 1;  // the kernel runs two rows per loop trip over seg + 2 rows
    }
    if (seg > g.n[1]) seg = g.n[1];
    sel.seg = seg;
    sel.wrap = part_wrap(part);
    R.nseg = (g.n[1] + seg - 1) / seg;
    sel.nwaves = R.ns * R.nseg;
    const MarchBC bc = make_march_bc(sp, touches);
    const dim3 grid(((sel.nwaves + 7) / 8) * 8), block(64);
    SourcePro pr;
    std::memset(&pr, 0, sizeof(pr));
    // two waves per strip (k_fld_pair) with CMFP_FIELD_PAIR=1: measured slower than one (see the kernel's comment)
    static const bool pair = std::getenv("CMFP_FIELD_PAIR") && std::atoi(std::getenv("CMFP_FIELD_PAIR")) == 1;
    if (pair) {
        const dim3 block2(128);
#define FLDP(BC, NS, PR, E) hipLaunchKernelGGL((k_fld_pair<BC, NS, PR>), grid, block2, 0, st, g, sp.fp, Uin, Uout, dt, bc, sel, E, pr)
        if (pro) {
            pr = *pro;
            if (has_bc) FLDP(true, 2, true, *epi); else FLDP(false, 2, true, *epi);
        } else if (epi && epi->ns == 2) {
            if (has_bc) FLDP(true, 2, false, *epi); else FLDP(false, 2, false, *epi);
        } else if (epi && epi->ns == 1) {
            if (has_bc) FLDP(true, 1, false, *epi); else FLDP(false, 1, false, *epi);
        } else {
            if (has_bc) FLDP(true, 0, false, NoEpi{}); else FLDP(false, 0, false, NoEpi{});
        }
#undef FLDP
        return 1;
    }
#define FLD(BC, NS, E) hipLaunchKernelGGL((k_fld_march<BC, NS, false>), grid, block, 0, st, g, sp.fp, Uin, Uout, dt, bc, sel, E, pr)
    if (pro) {
        pr = *pro;
        if (has_bc)
            hipLaunchKernelGGL((k_fld_march<true, 2, true>), grid, block, 0, st, g, sp.fp, Uin, Uout, dt, bc, sel, *epi, pr);
        else
            hipLaunchKernelGGL((k_fld_march<false, 2, true>), grid, block, 0, st, g, sp.fp, Uin, Uout, dt, bc, sel, *epi, pr);
        return 1;
    }
    if (epi && epi->ns == 2) {
        if (has_bc) FLD(true, 2, *epi); else FLD(false, 2, *epi);
    } else if (epi && epi->ns == 1) {
        if (has_bc) FLD(true, 1, *epi); else FLD(false, 1, *epi);
    } else {
        if (has_bc) FLD(true, 0, NoEpi{}); else FLD(false, 0, NoEpi{});
    }
#undef FLD
    return 1;
}

}  // namespace CMFP_NS
