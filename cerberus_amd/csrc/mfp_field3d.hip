// mfp_field3d.hip — z-marching corner-transport-upwind update of the field state (3-D), two passes.
//
// HydroCTU::calc_spatial_derivative (MFP_CTU_hydro.cpp:142-531) + `new += dU` (MFP_advance.cpp:220-230)
// for the Maxwell state on a 3-D block.  The one-launch pipeline of mfp_march3d.hip does not fit the
// field: ten components with a 7-point stencil need a 70-value window per column along the marching
// direction and ~90 carried values per column.  What makes the hydro pipeline deep are the transverse
// corrections that wait for F_z of the next plane; with the sweep-1 fluxes in memory nothing waits:
//
//   pass 1  k_f3d_flux1:   reconstruction + half-step trace of every cell, once, direction by direction;
//                          Riemann sweep 1 -> F1[x], F1[y], F1[z] (8 components each) to memory
//   pass 2  k_f3d_update:  the same reconstruction again (linear, cheap), transverse corrections from F1
//                          (EulerianState::correct_face_prim MFP_eulerian.cpp:1047-1188: (d1+1)%3 then
//                          (d1+2)%3), domain-face BCs, Riemann sweep 2, divergence, Uout = Uin + dU
//
// Both are 16 x 16 workgroups that own a 14 x 14 column of cells and march along z: the x and y
// stencils come from an LDS tile of the current plane (grown by the stencil half width), the
// x-neighbour's face state through DPP row shifts, the y-neighbour's through LDS, the z-neighbour's is
// carried by the thread; the z stencil is read from memory (each plane is read as centre once and six
// times as a neighbour: L2 / Infinity Cache hits).  HBM traffic: U + F1 written once and read once
// (624 B per cell against 160 B algorithmic) instead of the gather kernels' two flux sets and, per
// face, two reconstructions whose 8-point stencils are gathered through L1/L2.  Same device functions
// and operation order per value: STRICT results are bit-identical to the oracle and the gather kernels.
#include "mfp_march.h"

namespace CMFP_NS {

CMFP_DEV double f3_row_left(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, false);  // row_shr:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
CMFP_DEV double f3_row_right(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x101, 0xf, 0xf, false);  // row_shl:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x101, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

struct FaceBC3f {
    int touch[3][2];
    int rule[3][2][10];
};
struct SelF3 {
    int gx, gy, seg, n;  // tiles in x and y, planes per z-segment, workgroups
    int lo;              // first cell index of the covered range in every dimension (-1: pass 1, 0: pass 2)
    int np;              // planes covered
};

// reconstruction + half-step trace of one component (EulerianState::calc_reconstruction
// MFP_eulerian.cpp:513-607, calc_time_averaged_faces :609-676); FAST mode: flat mu / ep stencils return
// the cell value (see mfp_fused.hip phase 1)
template <int RECON> CMFP_DEV void f3_faces(const double* st, const int n, const double speed, const double dt_2dx, double& lo, double& hi) {
    constexpr int SL = ReconTraits<RECON>::SL, H = SL / 2;
#ifdef CMFP_FASTMATH
    if (n >= F_MU) {
        bool same = true;
#pragma unroll
        for (int s = 0; s < SL; ++s) same = same && (st[s] == st[H]);
        if (same) {
            lo = hi = st[H];
            return;
        }
    }
#endif
    recon_faces<RECON>(st, lo, hi);
    trace_faces(speed, dt_2dx, st[H], lo, hi);
}

// the LDS tile of plane kp: conserved = primitive values of the thread tile grown by H (clamped to the array)
template <int H, int PX, int NP>
CMFP_DEV void f3_fill(const Geo& g, const double* __restrict__ U, double (*dst)[NP], int ia, int ja, int kp, int tid) {
    const int xlo = -g.ng[0], xhi = g.n[0] + g.ng[0] - 1, ylo = -g.ng[1], yhi = g.n[1] + g.ng[1] - 1;
    for (int t = tid; t < NP; t += 256) {
        const int py = t / PX, px = t - py * PX;
        int ci = ia + px, cj = ja + py;
        ci = ci < xlo ? xlo : (ci > xhi ? xhi : ci);  // cells outside the array feed only values nobody uses
        cj = cj < ylo ? ylo : (cj > yhi ? yhi : cj);
        const long c = g.at(ci, cj, kp);
#pragma unroll
        for (int n = 0; n < 10; ++n) dst[n][t] = U[c + n * g.sc];
    }
}

// one direction of one cell: stencils of all ten components -> traced face states
template <int RECON, int NP>
CMFP_DEV void f3_recon_lds(const double (*Pm)[NP], int pc, int stride, double speed, double h, double* lo, double* hi) {
    constexpr int SL = ReconTraits<RECON>::SL, H = SL / 2;
#pragma unroll
    for (int n = 0; n < 10; ++n) {
        double st[SL];
#pragma unroll
        for (int s = 0; s < SL; ++s) st[s] = Pm[n][pc + (s - H) * stride];
        f3_faces<RECON>(st, n, speed, h, lo[n], hi[n]);
    }
}
template <int RECON, int NP>
CMFP_DEV void f3_recon_z(const Geo& g, const double* __restrict__ U, const double (*Pm)[NP], int pc, long c, int kmin, int kmax,
                         int m, double speed, double h, double* lo, double* hi) {
    constexpr int SL = ReconTraits<RECON>::SL, H = SL / 2;
#pragma unroll
    for (int n = 0; n < 10; ++n) {
        double st[SL];
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            int k = m + s - H;
            k = k < kmin ? kmin : (k > kmax ? kmax : k);  // planes outside the array feed only values nobody uses
            st[s] = (s == H) ? Pm[n][pc] : U[c + (long)(k - m) * g.sk + n * g.sc];
        }
        f3_faces<RECON>(st, n, speed, h, lo[n], hi[n]);
    }
}

// ------------------------------------------------------------------------------------------ pass 1
// F1[d] on cells with index_d in [0, n_d] and the other indices in [-1, n] (the flux boxes grown by one
// transversally, MFP_CTU_hydro.cpp:254-268).  Threads cover the index range [-1, n] plus a ring.
template <int RECON>
__global__ __launch_bounds__(256, 2) void k_f3d_flux1(Geo g, FieldParams P, const double* __restrict__ U, Ptr3 F1, double dt,
                                                       SelF3 sel) {
    constexpr int H = ReconTraits<RECON>::SL / 2, BX = 16, PX = BX + 2 * H, NP = PX * PX, C = 256;
    __shared__ double Pq[10][NP];
    __shared__ double bufR[10][C];
    dt = launch_dt(g, dt);
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * BX + tx;
    const int per = (sel.n + 7) >> 3;
    const int wg = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (wg >= sel.n) return;
    const int tiles = sel.gx * sel.gy;
    const int segi = wg / tiles, tile = wg - segi * tiles;
    const int tiy = tile / sel.gx, tix = tile - tiy * sel.gx;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const int i = sel.lo + tix * 14 - 1 + tx, j = sel.lo + tiy * 14 - 1 + ty;  // this thread's column
    const int ka = sel.lo + segi * sel.seg;
    const int kend = sel.lo + sel.np - 1;
    const int kb = (ka + sel.seg - 1 < kend) ? ka + sel.seg - 1 : kend;
    const int pc = (ty + H) * PX + (tx + H);
    const double speed = P.fastest;
    const double hx = uni(0.5 * dt / g.dx[0]), hy = uni(0.5 * dt / g.dx[1]), hz = uni(0.5 * dt / g.dx[2]);
    const bool inner = tx >= 1 && tx <= 14 && ty >= 1 && ty <= 14;
    const bool i_ok = i >= -1 && i <= n0, j_ok = j >= -1 && j <= n1;
    const int kmin = -g.ng[2], kmax = n2 + g.ng[2] - 1;
    const int ic = i < -g.ng[0] ? -g.ng[0] : (i > n0 + g.ng[0] - 1 ? n0 + g.ng[0] - 1 : i);
    const int jc = j < -g.ng[1] ? -g.ng[1] : (j > n1 + g.ng[1] - 1 ? n1 + g.ng[1] - 1 : j);
    const long col = g.at(ic, jc, 0);

    double zhi_p[10];
#pragma unroll
    for (int n = 0; n < 10; ++n) zhi_p[n] = 0.0;
    // the first plane of a segment that is not the first plane of the range only supplies zhi
    const int m0 = ka > sel.lo ? ka - 1 : ka;
    for (int m = m0; m <= kb; ++m) {
        const bool emit = m >= ka;
        __syncthreads();
        f3_fill<H, PX, NP>(g, U, Pq, i - tx - H, j - ty - H, m, tid);
        __syncthreads();
        const long c = col + (long)m * g.sk;
        double lo[10], hi[10], F[10];
        if (emit) {
            // ---- x
            f3_recon_lds<RECON, NP>(Pq, pc, 1, speed, hx, lo, hi);
            double L[10];
#pragma unroll
            for (int n = 0; n < 10; ++n) L[n] = f3_row_left(hi[n]);
            FieldF<RECON>::riemann(P, 0, L, lo, F);
            if (inner && i >= 0 && i <= n0 && j_ok) {
#pragma unroll
                for (int n = 0; n < 8; ++n) F1.p[0][c + n * g.sc] = F[n];
            }
            // ---- y
            f3_recon_lds<RECON, NP>(Pq, pc, PX, speed, hy, lo, hi);
#pragma unroll
            for (int n = 0; n < 10; ++n) bufR[n][tid] = hi[n];
        }
        __syncthreads();
        if (emit) {
            double L[10];
#pragma unroll
            for (int n = 0; n < 10; ++n) L[n] = bufR[n][ty >= 1 ? tid - BX : tid];
            FieldF<RECON>::riemann(P, 1, L, lo, F);
            if (inner && j >= 0 && j <= n1 && i_ok) {
#pragma unroll
                for (int n = 0; n < 8; ++n) F1.p[1][c + n * g.sc] = F[n];
            }
        }
        // ---- z
        f3_recon_z<RECON, NP>(g, U, Pq, pc, c, kmin, kmax, m, speed, hz, lo, hi);
        if (emit && m >= 0) {
            FieldF<RECON>::riemann(P, 2, zhi_p, lo, F);
            if (inner && i_ok && j_ok) {
#pragma unroll
                for (int n = 0; n < 8; ++n) F1.p[2][c + n * g.sc] = F[n];
            }
        }
#pragma unroll
        for (int n = 0; n < 10; ++n) zhi_p[n] = hi[n];
    }
}

// ------------------------------------------------------------------------------------------ pass 2
template <int RECON, bool HAS_BC>
__global__ __launch_bounds__(256, 2) void k_f3d_update(Geo g, FieldParams P, const double* __restrict__ Uin, double* __restrict__ Uout,
                                                        Ptr3 F1, double dt, FaceBC3f bc, SelF3 sel) {
    constexpr int H = ReconTraits<RECON>::SL / 2, BX = 16, PX = BX + 2 * H, NP = PX * PX, C = 256;
    __shared__ double Pq[10][NP];
    __shared__ double bufR[10][C];
    __shared__ double Fb[8][C];
    dt = launch_dt(g, dt);
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * BX + tx;
    const int per = (sel.n + 7) >> 3;
    const int wg = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (wg >= sel.n) return;
    const int tiles = sel.gx * sel.gy;
    const int segi = wg / tiles, tile = wg - segi * tiles;
    const int tiy = tile / sel.gx, tix = tile - tiy * sel.gx;
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const int i = tix * 14 - 1 + tx, j = tiy * 14 - 1 + ty;  // this thread's column (the tile grown by one)
    const int ka = segi * sel.seg;
    const int kb = (ka + sel.seg - 1 < n2 - 1) ? ka + sel.seg - 1 : n2 - 1;
    const int pc = (ty + H) * PX + (tx + H);
    const double speed = P.fastest;
    const double hx = uni(0.5 * dt / g.dx[0]), hy = uni(0.5 * dt / g.dx[1]), hz = uni(0.5 * dt / g.dx[2]);
    const double dtdx = uni(dt / g.dx[0]), dtdy = uni(dt / g.dx[1]), dtdz = uni(dt / g.dx[2]);
    const bool own = tx >= 1 && tx <= 14 && ty >= 1 && ty <= 14 && i < n0 && j < n1;
    const int kmin = -g.ng[2], kmax = n2 + g.ng[2] - 1;
    const int ic = i > n0 ? n0 : i, jc = j > n1 ? n1 : j;  // i, j >= -1; columns past the ring re-read it
    const long col = g.at(ic, jc, 0);
    const long sx = 1, sy = g.sj, sz = g.sk;

    // carried from plane m - 1: corrected high z-face state, F'_z on its low face, dU after x and y
    double zhiC_p[10], fz2_p[8], du_p[8];
#pragma unroll
    for (int n = 0; n < 10; ++n) zhiC_p[n] = 0.0;
#pragma unroll
    for (int n = 0; n < 8; ++n) fz2_p[n] = du_p[n] = 0.0;

    for (int m = ka - 1; m <= kb + 1; ++m) {
        __syncthreads();
        f3_fill<H, PX, NP>(g, Uin, Pq, i - tx - H, j - ty - H, m, tid);
        __syncthreads();
        const long c = col + (long)m * sz;
        // transverse flux differences of this cell, dF_d = F1[d](c) - F1[d](c + e_d): two of the three are
        // live at a time (z-faces: x, y; x-faces: y, z; y-faces: z, x), the third is (re)loaded
        double dFx[8], dFy[8], dFz[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            dFx[n] = F1.p[0][c + n * g.sc] - F1.p[0][c + sx + n * g.sc];
            dFy[n] = F1.p[1][c + n * g.sc] - F1.p[1][c + sy + n * g.sc];
        }
        const bool full = m >= ka && m <= kb;  // plane whose x and y fluxes are needed (the others only supply z-face states)
        double lo[10], hi[10], F[10];
        // ---- z faces first (they complete plane m - 1, whose carried values then die): dF_x then dF_y
        f3_recon_z<RECON, NP>(g, Uin, Pq, pc, c, kmin, kmax, m, speed, hz, lo, hi);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            hi[n] += hx * dFx[n];
            lo[n] += hx * dFx[n];
            hi[n] += hy * dFy[n];
            lo[n] += hy * dFy[n];
        }
        if (m >= ka) {
            double L[10], R[10], fz2[10];
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                L[n] = zhiC_p[n];
                R[n] = lo[n];
            }
            if (HAS_BC && bc.touch[2][0] && m == 0) {
#pragma unroll
                for (int n = 0; n < 10; ++n)
                    if (bc.rule[2][0][n]) L[n] = bc.rule[2][0][n] == FILL_ODD ? -R[n] : R[n];
            }
            if (HAS_BC && bc.touch[2][1] && m == n2) {
#pragma unroll
                for (int n = 0; n < 10; ++n)
                    if (bc.rule[2][1][n]) R[n] = bc.rule[2][1][n] == FILL_ODD ? -L[n] : L[n];
            }
            FieldF<RECON>::riemann(P, 2, L, R, fz2);
            // ---- plane m - 1 is complete: divergence (calc_divergence MFP_eulerian.cpp:1614-1662), new += dU
            if (m > ka && own) {
                const long cm = c - sz;
#pragma unroll
                for (int n = 0; n < 10; ++n) {
                    double d = 0.0;
                    if (n < 8) {
                        d = du_p[n];  // 0 + dt/dx (F_x - F_x+) + dt/dy (F_y - F_y+), accumulated in that order
                        d += dtdz * (fz2_p[n] - fz2[n]);
                    } else {  // F[mu] = F[ep] = 0 (MFP_field_RH.cpp:33)
                        d += dtdx * (0.0 - 0.0);
                        d += dtdy * (0.0 - 0.0);
                        d += dtdz * (0.0 - 0.0);
                    }
                    Uout[cm + n * g.sc] = Uin[cm + n * g.sc] + d;
                }
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) fz2_p[n] = fz2[n];
        }
#pragma unroll
        for (int n = 0; n < 10; ++n) zhiC_p[n] = hi[n];

        double du[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) du[n] = 0.0;
        if (full) {
#pragma unroll
            for (int n = 0; n < 8; ++n) dFz[n] = F1.p[2][c + n * g.sc] - F1.p[2][c + sz + n * g.sc];
            // ---- x faces: corrected with dF_y then dF_z
            f3_recon_lds<RECON, NP>(Pq, pc, 1, speed, hx, lo, hi);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                hi[n] += hy * dFy[n];
                lo[n] += hy * dFy[n];
                hi[n] += hz * dFz[n];
                lo[n] += hz * dFz[n];
            }
            double L[10], R[10];
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                L[n] = f3_row_left(hi[n]);
                R[n] = lo[n];
            }
            if (HAS_BC && bc.touch[0][0] && i == 0) {
#pragma unroll
                for (int n = 0; n < 10; ++n)
                    if (bc.rule[0][0][n]) L[n] = bc.rule[0][0][n] == FILL_ODD ? -R[n] : R[n];
            }
            if (HAS_BC && bc.touch[0][1] && i == n0) {
#pragma unroll
                for (int n = 0; n < 10; ++n)
                    if (bc.rule[0][1][n]) R[n] = bc.rule[0][1][n] == FILL_ODD ? -L[n] : L[n];
            }
            FieldF<RECON>::riemann(P, 0, L, R, F);
#pragma unroll
            for (int n = 0; n < 8; ++n) du[n] += dtdx * (F[n] - f3_row_right(F[n]));
            // ---- y faces: corrected with dF_z then dF_x (reloaded)
            f3_recon_lds<RECON, NP>(Pq, pc, PX, speed, hy, lo, hi);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const double dfx = F1.p[0][c + n * g.sc] - F1.p[0][c + sx + n * g.sc];
                hi[n] += hz * dFz[n];
                lo[n] += hz * dFz[n];
                hi[n] += hx * dfx;
                lo[n] += hx * dfx;
            }
#pragma unroll
            for (int n = 0; n < 10; ++n) bufR[n][tid] = hi[n];
        }
        __syncthreads();
        if (full) {
            double L[10], R[10];
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                L[n] = bufR[n][ty >= 1 ? tid - BX : tid];
                R[n] = lo[n];
            }
            if (HAS_BC && bc.touch[1][0] && j == 0) {
#pragma unroll
                for (int n = 0; n < 10; ++n)
                    if (bc.rule[1][0][n]) L[n] = bc.rule[1][0][n] == FILL_ODD ? -R[n] : R[n];
            }
            if (HAS_BC && bc.touch[1][1] && j == n1) {
#pragma unroll
                for (int n = 0; n < 10; ++n)
                    if (bc.rule[1][1][n]) R[n] = bc.rule[1][1][n] == FILL_ODD ? -L[n] : L[n];
            }
            FieldF<RECON>::riemann(P, 1, L, R, F);
#pragma unroll
            for (int n = 0; n < 8; ++n) Fb[n][tid] = F[n];
        }
        __syncthreads();
        if (full) {
#pragma unroll
            for (int n = 0; n < 8; ++n) du[n] += dtdy * (F[n] - Fb[n][ty <= 14 ? tid + BX : tid]);
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) du_p[n] = du[n];
    }
}

static SelF3 make_sel(const Geo& g, int lo, int ext) {
    SelF3 s;
    s.lo = lo;
    s.gx = (g.n[0] + ext + 13) / 14;
    s.gy = (g.n[1] + ext + 13) / 14;
    s.np = g.n[2] + ext;
    const int tiles = s.gx * s.gy;
    int nseg = (2 * wave_slots() + tiles - 1) / tiles;  // as in mfp_march3d.hip: eight rounds of resident workgroups
    if (nseg < 1) nseg = 1;
    int seg = (s.np + nseg - 1) / nseg;
    if (seg < 16) seg = 16;
    if (seg > s.np) seg = s.np;
    s.seg = seg;
    s.n = tiles * ((s.np + seg - 1) / seg);
    return s;
}

template <int RECON>
static int launch_f3d(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, const Ptr3& F1, double dt,
                      const int touches[3][2], hipStream_t st) {
    constexpr int H = ReconTraits<RECON>::SL / 2;
    if (g.ng[0] < H + 1 || g.ng[1] < H + 1 || g.ng[2] < H + 1) return 0;
    if (!Uin) return 1;  // coverage probe
    FaceBC3f bc;
    bool has_bc = false;
    for (int d = 0; d < 3; ++d)
        for (int s = 0; s < 2; ++s) {
            bc.touch[d][s] = touches[d][s];
            has_bc = has_bc || touches[d][s];
            for (int n = 0; n < 10; ++n) {
                const int code = sp.bc[n][d][s];  // MFP_eulerian.cpp:796-835
                bc.rule[d][s][n] = (code == 2 || code == 4 || code == 1) ? FILL_EVEN : (code == -1 ? FILL_ODD : FILL_SKIP);
            }
        }
    const dim3 block(16, 16);
    const SelF3 s1 = make_sel(g, -1, 2);  // cells [-1, n] in every dimension
    hipLaunchKernelGGL((k_f3d_flux1<RECON>), dim3(((s1.n + 7) / 8) * 8), block, 0, st, g, sp.fp, Uin, F1, dt, s1);
    const SelF3 s2 = make_sel(g, 0, 0);
    if (has_bc)
        hipLaunchKernelGGL((k_f3d_update<RECON, true>), dim3(((s2.n + 7) / 8) * 8), block, 0, st, g, sp.fp, Uin, Uout, F1, dt, bc, s2);
    else
        hipLaunchKernelGGL((k_f3d_update<RECON, false>), dim3(((s2.n + 7) / 8) * 8), block, 0, st, g, sp.fp, Uin, Uout, F1, dt, bc, s2);
    return 1;
}

// Two-pass z-marching update of a 3-D field state; F1: sweep-1 flux scratch (8 planes per direction).
// Returns 0 where the option set is outside these kernels (the caller then uses the gather kernels).
int ctu_field3d(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, const Ptr3& F1, int do_ctu, double dt,
                const int touches[3][2], hipStream_t st) {
    if (g.dim != 3 || sp.kind != 1 || !do_ctu || sp.flux != 10) return 0;
    if (sp.recon == 5) return launch_f3d<5>(g, sp, Uin, Uout, F1, dt, touches, st);
    if (sp.recon == 2) return launch_f3d<2>(g, sp, Uin, Uout, F1, dt, touches, st);
    return 0;
}

}  // namespace CMFP_NS
