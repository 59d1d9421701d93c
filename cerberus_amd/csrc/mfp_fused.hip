// mfp_fused.hip — hand-fused corner-transport-upwind update for gfx950.
//
// One launch does, for one state on one 2-D block, everything
// HydroCTU::calc_spatial_derivative (MFP_CTU_hydro.cpp:142-531) does with eight
// separate full-array passes, plus the `new += dU` of the integrator
// (MFP_advance.cpp:220-230):
//
//   U -> primitives -> slope-limited faces -> half-step trace -> Riemann sweep 1
//     -> transverse correction -> domain-face BCs -> Riemann sweep 2 -> divergence
//
// Each conserved component is read from HBM once (plus the tile halo, served by
// L2) and written once.  A workgroup owns a (BX-2) x (BY-2) tile of cells and
// runs one thread per cell of the tile grown by one (the cells whose face states
// the tile consumes).  Face states stay in registers; only what a neighbouring
// cell needs (the high-side face states and the face fluxes) is exchanged
// through LDS, in five barrier-separated phases.  Arithmetic per value is the
// same sequence of operations as the staged kernels (same device functions), so
// results are identical to them bit for bit.
#include "mfp_device.h"
#include "mfp_source.h"

namespace CMFP_NS {
using namespace cmfp;

#ifdef CMFP_STAMPS
// diagnostic build only: per-phase shader cycles of wave 0 of every workgroup
__device__ unsigned long long g_stamps[16];
#define STAMP(k)                                                          \
    do {                                                                  \
        unsigned long long t_ = __builtin_readcyclecounter();             \
        if (tid == 0) atomicAdd(&g_stamps[k], t_ - t_prev_);              \
        t_prev_ = t_;                                                     \
    } while (0)
#else
#define STAMP(k)
#endif

// value of the lane one to the left / right inside a row of 16 lanes (= the x-neighbour thread for
// BX = 16): DPP row shifts, register to register, no LDS.  Every lane of the wave must execute these.
CMFP_DEV double from_left(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, true);  // row_shr:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
CMFP_DEV double from_right(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x101, 0xf, 0xf, true);  // row_shl:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x101, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

struct FaceBC {
    int touch[2][2];   // [d][side]: physical (non-periodic) domain face on this block
    int rule[2][2][10];  // FILL_SKIP / FILL_EVEN / FILL_ODD per primitive comp
};

// Which tiles a launch covers.  part 0: all gx x gy tiles.  The other two split the tile grid for
// halo/compute overlap: part 1 = the tiles [ix0, ix1] x [iy0, iy1] whose whole stencil footprint lies
// in valid cells (they can run while ghost cells are still in flight), part 2 = the frame around them.
struct TileSel {
    int part, gx, gy, ix0, ix1, iy0, iy1, n;  // n: tiles in this launch
};
__device__ __forceinline__ void tile_of(const TileSel& s, int t, int& tix, int& tiy) {
    if (s.part == 0) {
        tix = t % s.gx;
        tiy = t / s.gx;
    } else if (s.part == 1) {
        const int w = s.ix1 - s.ix0 + 1;
        tix = s.ix0 + t % w;
        tiy = s.iy0 + t / w;
    } else {
        const int rows_b = s.iy0 + (s.gy - 1 - s.iy1);  // full tile rows below and above the interior
        if (t < rows_b * s.gx) {
            const int ry = t / s.gx;
            tix = t - ry * s.gx;
            tiy = ry < s.iy0 ? ry : s.iy1 + 1 + (ry - s.iy0);
        } else {
            const int u = t - rows_b * s.gx;
            const int cols_b = s.ix0 + (s.gx - 1 - s.ix1);
            const int ry = u / cols_b, cx = u - ry * cols_b;
            tiy = s.iy0 + ry;
            tix = cx < s.ix0 ? cx : s.ix1 + 1 + (cx - s.ix0);
        }
    }
}

// SRC > 0 (field states): the explicit Strang source stage C (k_source_rk2's arithmetic) of SRC species is
// applied to the cells this kernel updates, with the CFL reductions of what it writes.
template <class Phys, int BX, int BY, bool HAS_BC, bool REDUCE, int SRC = 0>
__global__ __launch_bounds__(BX* BY, (Phys::IS_HYDRO ? 1024 : 768) / (BX * BY)) void k_ctu2d(Geo g, typename Phys::Params P, const double* __restrict__ Uin,
                                                   double* __restrict__ Uout, int do_ctu, double dt, FaceBC bc, int* err,
                                                   TileSel sel, unsigned long long* red, typename EpiArg<SRC>::type epi) {
    constexpr int NQ = Phys::NQ, NC = Phys::NC, NF = Phys::NF, RECON = Phys::RECON;
    constexpr int SL = ReconTraits<RECON>::SL, H = SL / 2;
    constexpr int TX = BX - 2, TY = BY - 2;
    dt = launch_dt(g, dt);  // device-resident time step (cmfp_advance_steps)
    // wave-uniform factors, held in scalar registers (the field kernel sits at its VGPR budget)
    const double h_dx[2] = {uni(0.5 * dt / g.dx[0]), uni(0.5 * dt / g.dx[1])};
    const double dt_dx[2] = {uni(dt / g.dx[0]), uni(dt / g.dx[1])};
    constexpr int PX = BX + 2 * H, PY = BY + 2 * H;
    constexpr int C = BX * BY;
    static_assert(BX == 16, "the x-neighbour exchange uses DPP shifts inside rows of 16 lanes");
    // primitives go through LDS NH components at a time: all of them for hydro (cons2prim couples
    // them), two halves for the field, whose 22 x 22 x 10 tile alone would cap the CU at 2 workgroups
    constexpr int NH = Phys::IS_HYDRO ? NQ : (NQ + 1) / 2;
    constexpr int NA = (NH * PX * PY > NF * C) ? NH * PX * PY : NF * C;
    __shared__ double bufA[NA];      // phase 0-1: primitives; phase 2-5: y-face fluxes
    __shared__ double bufR[NQ * C];  // high-side y-face states of every cell (x-neighbours talk by DPP)
    double* const Pq = bufA;
    double* const Fb = bufA;

    const int tx = threadIdx.x, ty = threadIdx.y;
    const int tid = ty * BX + tx;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b + 8
    // share one), each with its own L2.  Give every XCD a contiguous band of tile rows so
    // that the halo overlap of neighbouring tiles is served by that XCD's L2 instead of HBM.
    const int per = (sel.n + 7) >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (tile >= sel.n) return;
    int tix, tiy;
    tile_of(sel, tile, tix, tiy);
    const int i0 = tix * TX, j0 = tiy * TY;
    const int i = i0 - 1 + tx, j = j0 - 1 + ty;  // this thread's cell
    const int n0 = g.n[0], n1 = g.n[1];
    const double gm1 = Phys::IS_HYDRO ? Phys::gamma_of(P) - 1.0 : 0.0;
#ifdef CMFP_STAMPS
    unsigned long long t_prev_ = __builtin_readcyclecounter();
#endif

    // ---- phases 0 + 1, NH components at a time:
    //   0: conserved -> primitives for the tile grown by 1 + H, into LDS
    //   1: reconstruction + half-step trace of this cell, both directions
    double lo[2][NQ], hi[2][NQ];
    double speed[3];
    unsigned flat = 0;  // FAST mode, field: (direction, component) pairs whose stencil is flat
    const int pc = (ty + H) * PX + (tx + H);
#pragma unroll
    for (int h0 = 0; h0 < NQ; h0 += NH) {
        for (int t = tid; t < PX * PY; t += C) {
            const int px = t % PX, py = t / PX;
            const int ci = i0 - 1 - H + px, cj = j0 - 1 - H + py;
            if (ci >= -(1 + H) && ci <= n0 + H && cj >= -(1 + H) && cj <= n1 + H) {
                const long c = g.at(ci, cj, 0);
                if constexpr (Phys::IS_HYDRO) {
                    double U[NC], Q[NQ];
#pragma unroll
                    for (int n = 0; n < NC; ++n) U[n] = Uin[c + n * g.sc];
                    if (!Phys::cons2prim(P, U, Q)) *err = 1;
                    Q[H_P] = f_div(Q[H_P], gm1);  // eps = p / (gamma - 1), MFP_hydro.cpp:847-862
#pragma unroll
                    for (int n = 0; n < NQ; ++n) Pq[n * (PX * PY) + t] = Q[n];
                } else {
#pragma unroll
                    for (int n = 0; n < NH; ++n)
                        if (h0 + n < NQ) Pq[n * (PX * PY) + t] = Uin[c + (h0 + n) * g.sc];  // FieldState: prim = cons
                }
            }
        }
        STAMP(0);
        __syncthreads();
        STAMP(1);
        if constexpr (Phys::IS_HYDRO) {
            double centre[NQ];
#pragma unroll
            for (int n = 0; n < NQ; ++n) centre[n] = Pq[n * (PX * PY) + pc];
            centre[H_P] = centre[H_P] * gm1;  // restored pressure, MFP_hydro.cpp:1025-1035
            Phys::speed_from_prim(P, centre, speed);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int sd = d == 0 ? 1 : PX;
                double st[SL][5];
#pragma unroll
                for (int s = 0; s < SL; ++s)
#pragma unroll
                    for (int n = 0; n < 5; ++n) st[s][n] = Pq[n * (PX * PY) + pc + (s - H) * sd];
                hydro_cell_faces<RECON>(Phys::gamma_of(P), st, lo[d], hi[d]);
                if (do_ctu) {
                    const double dt_2dx = h_dx[d];
#pragma unroll
                    for (int n = 0; n < NQ; ++n) trace_faces(speed[d], dt_2dx, centre[n], lo[d][n], hi[d][n]);
                }
            }
        } else {
            Phys::speed_from_prim(P, nullptr, speed);
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int sd = d == 0 ? 1 : PX;
                const double dt_2dx = h_dx[d];
#pragma unroll
                for (int m = 0; m < NH; ++m) {
                    const int n = h0 + m;
                    if (n >= NQ) continue;
                    double st[SL];
#pragma unroll
                    for (int s = 0; s < SL; ++s) st[s] = Pq[m * (PX * PY) + pc + (s - H) * sd];
#ifdef CMFP_FASTMATH
                    // FAST mode: the material coefficients mu, ep never change (F = 0, no source);
                    // where their stencil is flat the face values are the cell value itself (the
                    // weights sum to one; the STRICT evaluation gives 1 + 2^-52 for O6) and the
                    // trace is the identity, so the Riemann solver sees exact vacuum values.
                    if (n >= F_MU) {
                        bool same = true;
#pragma unroll
                        for (int s = 0; s < SL; ++s) same = same && (st[s] == st[H]);
                        if (same) {
                            lo[d][n] = hi[d][n] = st[H];
                            flat |= 1u << (d * 16 + n);
                            continue;
                        }
                    }
#endif
                    recon_faces<RECON>(st, lo[d][n], hi[d][n]);
                    if (do_ctu) trace_faces(speed[d], dt_2dx, st[H], lo[d][n], hi[d][n]);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < NH; ++m)
            if (h0 + m < NQ) bufR[(h0 + m) * C + tid] = hi[1][h0 + m];
        STAMP(2);
        __syncthreads();
        STAMP(3);
    }
    (void)flat;

    // ---- phase 2: Riemann sweep 1 on the low faces of this cell
    double fx[NC] = {0}, fy[NC] = {0};
    {
        double L[NQ];
#pragma unroll
        for (int n = 0; n < NQ; ++n) L[n] = from_left(hi[0][n]);
        if (tx >= 1) Phys::riemann(P, 0, L, lo[0], fx);
        if (ty >= 1) {
#pragma unroll
            for (int n = 0; n < NQ; ++n) L[n] = bufR[n * C + tid - BX];
            Phys::riemann(P, 1, L, lo[1], fy);
        }
    }
    if (do_ctu) {
#pragma unroll
        for (int n = 0; n < NF; ++n) Fb[n * C + tid] = fy[n];
        double fxr[NF];  // x-flux at the high x-face = the right neighbour's low face
#pragma unroll
        for (int n = 0; n < NF; ++n) fxr[n] = from_right(fx[n]);
        STAMP(4);
        __syncthreads();
        STAMP(5);

        // ---- phase 3: transverse correction of this cell's face states
        // (EulerianState::correct_face_prim MFP_eulerian.cpp:1047-1188, written per cell:
        //  each face state is corrected with the transverse fluxes of its own cell)
        const bool row_in = (ty >= 1 && ty <= BY - 2 && j < n1);   // cell row inside the block
        const bool col_in = (tx >= 1 && tx <= BX - 2 && i < n0);
        if (row_in) {
            // x-face states use F_y(i,j) - F_y(i,j+1)
            double dF[NF];
            const double k = h_dx[1];
#pragma unroll
            for (int n = 0; n < NF; ++n) dF[n] = fy[n] - Fb[n * C + tid + BX];
            if (tx <= BX - 2 && i < n0) {
                if (!Phys::correct(P, hi[0], k, dF)) *err = 1;
            }
            if (tx >= 1 && i <= n0) {
                if (!Phys::correct(P, lo[0], k, dF)) *err = 1;
            }
        }
        if (col_in) {
            double dF[NF];
            const double k = h_dx[0];
#pragma unroll
            for (int n = 0; n < NF; ++n) dF[n] = fx[n] - fxr[n];
            if (ty <= BY - 2 && j < n1) {
                if (!Phys::correct(P, hi[1], k, dF)) *err = 1;
            }
            if (ty >= 1 && j <= n1) {
                if (!Phys::correct(P, lo[1], k, dF)) *err = 1;
            }
        }
#pragma unroll
        for (int n = 0; n < NQ; ++n) bufR[n * C + tid] = hi[1][n];
        STAMP(6);
        __syncthreads();
        STAMP(7);

        // ---- phase 4: domain-face BCs (update_face_prim, do_all) + Riemann sweep 2
        {
            double L[NQ], R[NQ];
#pragma unroll
            for (int n = 0; n < NQ; ++n) L[n] = from_left(hi[0][n]);
            if (tx >= 1) {
#pragma unroll
                for (int n = 0; n < NQ; ++n) R[n] = lo[0][n];
                if (HAS_BC && bc.touch[0][0] && i == 0) {
#pragma unroll
                    for (int n = 0; n < NQ; ++n)
                        if (bc.rule[0][0][n]) L[n] = bc.rule[0][0][n] == FILL_ODD ? -R[n] : R[n];
                }
                if (HAS_BC && bc.touch[0][1] && i == n0) {
#pragma unroll
                    for (int n = 0; n < NQ; ++n)
                        if (bc.rule[0][1][n]) R[n] = bc.rule[0][1][n] == FILL_ODD ? -L[n] : L[n];
                }
                Phys::riemann(P, 0, L, R, fx);
            }
            if (ty >= 1) {
#pragma unroll
                for (int n = 0; n < NQ; ++n) {
                    L[n] = bufR[n * C + tid - BX];
                    R[n] = lo[1][n];
                }
                if (HAS_BC && bc.touch[1][0] && j == 0) {
#pragma unroll
                    for (int n = 0; n < NQ; ++n)
                        if (bc.rule[1][0][n]) L[n] = bc.rule[1][0][n] == FILL_ODD ? -R[n] : R[n];
                }
                if (HAS_BC && bc.touch[1][1] && j == n1) {
#pragma unroll
                    for (int n = 0; n < NQ; ++n)
                        if (bc.rule[1][1][n]) R[n] = bc.rule[1][1][n] == FILL_ODD ? -L[n] : L[n];
                }
                Phys::riemann(P, 1, L, R, fy);
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NF; ++n) Fb[n * C + tid] = fy[n];
    double fxh[NF];
#pragma unroll
    for (int n = 0; n < NF; ++n) fxh[n] = from_right(fx[n]);
    STAMP(8);
    __syncthreads();
    STAMP(9);

    double red_speed = 2.2250738585072014e-308;  // std::numeric_limits<Real>::min(), MFP_eulerian.cpp:265
    double src_speed[SRC > 0 ? SRC : 1];
    SrcDt src_dt;
#pragma unroll
    for (int s = 0; s < (SRC > 0 ? SRC : 1); ++s) src_speed[s] = DBL_MIN;
    // ---- phase 5: divergence (calc_divergence MFP_eulerian.cpp:1614-1662) and new += dU
    if (tx >= 1 && tx <= BX - 2 && ty >= 1 && ty <= BY - 2 && i < n0 && j < n1) {
        const long c = g.at(i, j, 0);
        const double dtdx = dt_dx[0], dtdy = dt_dx[1];
        double Unew[NC];
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            double du = 0.0;
            if (n < NF) {
                du += dtdx * (fx[n] - fxh[n]);
                du += dtdy * (fy[n] - Fb[n * C + tid + BX]);
            } else {
                du += dtdx * (0.0 - 0.0);
                du += dtdy * (0.0 - 0.0);
            }
            Unew[n] = Uin[c + n * g.sc] + du;
            if constexpr (SRC == 0) Uout[c + n * g.sc] = Unew[n];
        }
        if constexpr (SRC > 0) {
            // Strang stage C on this cell (MFP_advance.cpp:235-281): the species were updated by their own
            // CTU kernels before this launch; in place on their new-time data
            source_epilogue_cell<SRC>(g, epi, dt, c, Unew, src_speed, src_dt);
#pragma unroll
            for (int n = 0; n < NC; ++n) Uout[c + n * g.sc] = Unew[n];
        }
        if constexpr (Phys::IS_HYDRO && REDUCE) {
            {  // EulerianState::get_allowed_time_step MFP_eulerian.cpp:290-296 on the values just written
                double s[3];
                hydro_speed_from_cons(P.gas, Unew, s);
                red_speed = std_max(std_max(red_speed, s[0]), s[1]);
            }
        }
    }
    if constexpr (Phys::IS_HYDRO && REDUCE) {
        {
            // CFL reduction fused into the last kernel that writes this state: wave max,
            // workgroup max through LDS, then one 64-bit integer atomic per workgroup on one of
            // 256 slots (positive doubles order like their bit patterns)
            double* const shr = bufR;  // free since the barrier that ended phase 4
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) red_speed = fmax(red_speed, __shfl_xor(red_speed, o, 64));
            if ((tid & 63) == 0) shr[tid >> 6] = red_speed;
            __syncthreads();
            if (tid == 0) {
                for (int w2 = 1; w2 < C / 64; ++w2) red_speed = fmax(red_speed, shr[w2]);
                atomicMax(&red[tile & 255], (unsigned long long)__double_as_longlong(red_speed));
            }
        }
    }
    if constexpr (SRC > 0) {
        // the reductions k_source_rk2<REDUCE> leaves (bufR is free since the barrier that ended phase 4)
        source_epilogue_reduce<SRC>(epi, src_speed, src_dt, bufR, tid, C / 64, tile & 255);
    }
    STAMP(10);
}

static FaceBC make_bc(const StatePhys& sp, const int touches[3][2]) {
    FaceBC b;
    for (int d = 0; d < 2; ++d)
        for (int s = 0; s < 2; ++s) {
            b.touch[d][s] = touches[d][s];
            for (int n = 0; n < 10; ++n) {
                int code = (n < sp.nc) ? sp.bc[n][d][s] : -666;  // MFP_eulerian.cpp:796-835
                b.rule[d][s][n] = (code == 2 || code == 4 || code == 1) ? FILL_EVEN : (code == -1 ? FILL_ODD : FILL_SKIP);
            }
        }
    return b;
}

// interior tiles of a launch (part 1): footprint [i0 - 1 - H, i0 + TX + H] inside [0, n - 1]; returns their number
template <int TX, int TY, int H> static int tile_interior(const Geo& g, TileSel& sel) {
    sel.ix0 = sel.iy0 = 1;
    sel.ix1 = g.n[0] - 1 - H - TX >= 0 ? (g.n[0] - 1 - H - TX) / TX : -1;
    sel.iy1 = g.n[1] - 1 - H - TY >= 0 ? (g.n[1] - 1 - H - TY) / TY : -1;
    return (sel.ix1 >= sel.ix0 && sel.iy1 >= sel.iy0) ? (sel.ix1 - sel.ix0 + 1) * (sel.iy1 - sel.iy0 + 1) : 0;
}
// the cells those tiles update
template <int BX, int BY, int H> static bool tile_interior_box(const Geo& g, IBox& box) {
    TileSel sel;
    if (tile_interior<BX - 2, BY - 2, H>(g, sel) <= 0) return false;
    box.lo[0] = sel.ix0 * (BX - 2), box.hi[0] = (sel.ix1 + 1) * (BX - 2) - 1;
    box.lo[1] = sel.iy0 * (BY - 2), box.hi[1] = (sel.iy1 + 1) * (BY - 2) - 1;
    box.lo[2] = box.hi[2] = 0;
    return true;
}

template <class Phys, int BX, int BY>
static int launch2d(const Geo& g, const typename Phys::Params& P, const StatePhys& sp, const double* Uin, double* Uout,
                    int do_ctu, double dt, const int touches[3][2], int* err, unsigned long long* red, int part,
                    hipStream_t st) {
    if (!Uin && part == 0) return 1;  // coverage probe
    constexpr int TX = BX - 2, TY = BY - 2, H = ReconTraits<Phys::RECON>::SL / 2;
    TileSel sel;
    sel.part = part;
    sel.gx = (g.n[0] + TX - 1) / TX;
    sel.gy = (g.n[1] + TY - 1) / TY;
    const int n_int = tile_interior<TX, TY, H>(g, sel);
    if (!Uin) return n_int > 0;  // split probe: is there an interior to overlap with?
    sel.n = part == 0 ? sel.gx * sel.gy : part == 1 ? n_int : sel.gx * sel.gy - n_int;
    if (sel.n <= 0) return 1;
    const bool has_bc = touches[0][0] || touches[0][1] || touches[1][0] || touches[1][1] || g.bc_instance;
    const dim3 grid(((sel.n + 7) / 8) * 8), block(BX, BY);
    const FaceBC bc = make_bc(sp, touches);
#define CMFP_LAUNCH(BC, RED) \
    hipLaunchKernelGGL((k_ctu2d<Phys, BX, BY, BC, RED>), grid, block, 0, st, g, P, Uin, Uout, do_ctu, dt, bc, err, sel, red, NoEpi{})
    if constexpr (Phys::IS_HYDRO) {
        if (red) {
            if (has_bc) CMFP_LAUNCH(true, true); else CMFP_LAUNCH(false, true);
            return 1;
        }
    }
    if (has_bc) CMFP_LAUNCH(true, false); else CMFP_LAUNCH(false, false);
#undef CMFP_LAUNCH
    return 1;
}

// the field update with the source epilogue: same tile selection as launch2d
template <class Phys, int BX, int BY, int NS>
static int launch2d_src(const Geo& g, const typename Phys::Params& P, const StatePhys& sp, const double* Uin, double* Uout,
                        int do_ctu, double dt, const int touches[3][2], const SourceEpi& epi, int part, hipStream_t st) {
    if (!Uin) return 1;  // coverage probe
    constexpr int TX = BX - 2, TY = BY - 2, H = ReconTraits<Phys::RECON>::SL / 2;
    TileSel sel;
    sel.part = part;
    sel.gx = (g.n[0] + TX - 1) / TX;
    sel.gy = (g.n[1] + TY - 1) / TY;
    const int n_int = tile_interior<TX, TY, H>(g, sel);
    sel.n = part == 0 ? sel.gx * sel.gy : part == 1 ? n_int : sel.gx * sel.gy - n_int;
    if (sel.n <= 0) return 1;
    const bool has_bc = touches[0][0] || touches[0][1] || touches[1][0] || touches[1][1] || g.bc_instance;
    const dim3 grid(((sel.n + 7) / 8) * 8), block(BX, BY);
    const FaceBC bc = make_bc(sp, touches);
    if (has_bc)
        hipLaunchKernelGGL((k_ctu2d<Phys, BX, BY, true, false, NS>), grid, block, 0, st, g, P, Uin, Uout, do_ctu, dt, bc, nullptr, sel,
                           nullptr, epi);
    else
        hipLaunchKernelGGL((k_ctu2d<Phys, BX, BY, false, false, NS>), grid, block, 0, st, g, P, Uin, Uout, do_ctu, dt, bc, nullptr, sel,
                           nullptr, epi);
    return 1;
}

#ifndef CMFP_HBX
#define CMFP_HBX 16
#define CMFP_HBY 16
#endif
#ifndef CMFP_FBX
#define CMFP_FBX 16
#define CMFP_FBY 16
#endif

#ifdef CMFP_STAMPS
extern "C" void cmfp_debug_stamps(unsigned long long* out, int reset) {
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[16] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z));
    }
}
#endif

// part: 0 all tiles, 1 interior tiles, 2 the frame of boundary tiles (see TileSel).  Uin == nullptr:
// part 0 probes coverage, part 1 probes whether the block has interior tiles.
// defined in mfp_march.hip
int ctu_march(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, int do_ctu, double dt,
              const int touches[3][2], int* err, unsigned long long* red, int part, hipStream_t st);

// defined in mfp_march_field.hip
int ctu_march_field(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, int do_ctu, double dt,
                    const int touches[3][2], const SourceEpi* epi, int part, const SourcePro* pro, hipStream_t st);

int ctu_march3d(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, int do_ctu, double dt,
                const int touches[3][2], int* err, unsigned long long* red, hipStream_t st);

// part bit 8 set: the LDS-tiled kernel even where the wave-marching kernel (mfp_march.hip) applies
int ctu_fused(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, int do_ctu, double dt,
              const int touches[3][2], int* err, unsigned long long* red, int part, hipStream_t st) {
    const bool tile_only = part & 0x100;
    const int wrap = part & 0x600;  // in-kernel periodic wrap: the marching kernels only
    part &= 0xff;
    if (g.dim == 3)  // z-marching kernel (mfp_march3d.hip): whole block in one launch
        return (tile_only || part != 0 || wrap) ? 0 : ctu_march3d(g, sp, Uin, Uout, do_ctu, dt, touches, err, red, st);
    if (g.dim != 2) return 0;
    if (!tile_only && ctu_march(g, sp, Uin, Uout, do_ctu, dt, touches, err, red, part | wrap, st)) return 1;
    if (!tile_only && ctu_march_field(g, sp, Uin, Uout, do_ctu, dt, touches, nullptr, part | wrap, nullptr, st)) return 1;
    if (wrap) return 0;
    if (sp.kind == 0 && sp.hp.nt) return 0;  // passive scalars: the marching kernel only
#define HY(R, F)                                                                                                    \
    if (sp.recon == R && sp.flux == F)                                                                              \
        return launch2d<HydroF<R, F>, CMFP_HBX, CMFP_HBY>(g, sp.hp, sp, Uin, Uout, do_ctu, dt, touches, err, red, part, st);
#define FI(R)                                                                                                       \
    if (sp.recon == R) return launch2d<FieldF<R>, CMFP_FBX, CMFP_FBY>(g, sp.fp, sp, Uin, Uout, do_ctu, dt, touches, err, red, part, st);
    if (sp.kind == 0) {
        HY(1, 0) HY(1, 1) HY(1, 2) HY(2, 0) HY(2, 1) HY(2, 2) HY(3, 0) HY(3, 1) HY(4, 0) HY(4, 1)
    } else {
        FI(5) FI(1) FI(2) FI(6)
    }
#undef HY
#undef FI
    return 0;
}

// The cells the interior launch (part 1) of ctu_fused / ctu_field_src updates for this state, from the same
// selection code the launches use.  The split launches with the source epilogue rely on: field interior inside
// every species' interior (the context checks it with this).  false: no such launch / no interior.
bool ctu_march_interior(const Geo& g, const StatePhys& sp, int do_ctu, IBox& box);
int ctu_interior(const Geo& g, const StatePhys& sp, int do_ctu, int flags, IBox* box) {
    if (g.dim != 2) return 0;
    if (!(flags & 0x100) && ctu_march_interior(g, sp, do_ctu, *box)) return 1;
    constexpr int HH = ReconTraits<1>::SL / 2;  // the tiled hydro kernels: three-point reconstructions
    if (sp.kind == 0) return tile_interior_box<CMFP_HBX, CMFP_HBY, HH>(g, *box);
    if (sp.recon == 5 || sp.recon == 6) return tile_interior_box<CMFP_FBX, CMFP_FBY, ReconTraits<5>::SL / 2>(g, *box);
    return tile_interior_box<CMFP_FBX, CMFP_FBY, ReconTraits<1>::SL / 2>(g, *box);
}

// The tiled 2-D field update followed, per cell, by the explicit Strang source stage C of epi.ns species
// (plasma5 explicit / Lorentz).  Same part semantics as ctu_fused; 0 where the option set is not covered.
int ctu_field_src(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, int do_ctu, double dt,
                  const int touches[3][2], const SourceEpi& epi, int part, hipStream_t st) {
    if (g.dim != 2 || sp.kind != 1 || epi.a.implicit || epi.a.coll.on) return 0;
    if (!(part & 0x100) && ctu_march_field(g, sp, Uin, Uout, do_ctu, dt, touches, &epi, part, nullptr, st)) return 1;
    if (part & 0x600) return 0;  // in-kernel periodic wrap: the marching kernel only
    part &= 0xff;
#define FS(R, NS) \
    if (sp.recon == R && epi.ns == NS) \
        return launch2d_src<FieldF<R>, CMFP_FBX, CMFP_FBY, NS>(g, sp.fp, sp, Uin, Uout, do_ctu, dt, touches, epi, part, st);
    FS(5, 2) FS(5, 1) FS(1, 2) FS(2, 2) FS(6, 2)
#undef FS
    return 0;
}

}  // namespace CMFP_NS
