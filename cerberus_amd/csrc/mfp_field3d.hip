// mfp_field3d.hip — corner-transport-upwind update of the field state (3-D) from LDS plane tiles.
//
// HydroCTU::calc_spatial_derivative (MFP_CTU_hydro.cpp:142-531) + `new += dU` (MFP_advance.cpp:220-230)
// for the Maxwell state on a 3-D block.  The one-launch pipeline of mfp_march3d.hip does not fit the
// field (ten components with a 7-point stencil: a 70-value window per column along the marching
// direction), and a stencil gathered from memory along the marching direction misses L2 (measured:
// 14 + 23 GB fetched per step at 256^3 against 1.3 + 4.6 GB needed; the resident workgroups' windows
// are 9 MB per XCD against 4 MB of L2).  With the sweep-1 fluxes in memory nothing has to be carried
// from plane to plane, so every direction can take its stencil from an LDS tile that contains it:
//
//   pass 1  k_f3d_xy<1>:  (x, y) tiles marching along z: reconstruction + half-step trace in x and y,
//                         Riemann sweep 1 -> F1[x], F1[y] (8 components each), stored as the difference
//                         dF_d = F_d(c) - F_d(c + e_d) of each cell: all that the corrections use
//           k_f3d_z<1>:   (x, z) tiles marching along y: the same in z -> F1[z]
//   pass 2  k_f3d_xy<2>:  x and y again (linear, cheap), transverse corrections from F1
//                         (EulerianState::correct_face_prim MFP_eulerian.cpp:1047-1188: (d1+1)%3 then
//                         (d1+2)%3), domain-face BCs, Riemann sweep 2, dU_xy = 0 + dt/dx dF_x + dt/dy dF_y
//           k_f3d_z<2>:   z faces likewise, Uout = Uin + (dU_xy + dt/dz dF_z)   (calc_divergence order,
//                         MFP_eulerian.cpp:1614-1662)
//
// (x, y) workgroups are 16 x 16 threads owning 14 x 14 cells (the x-neighbour's face state comes
// through DPP row shifts, the y-neighbour's through LDS); (x, z) workgroups are 16 cells wide in x
// (no x neighbour is needed) and own 14 cells in z.  HBM traffic: U read four times (with the tile
// halos), F1 and dU_xy written once and read once or twice.  Same device functions and operation order
// per value: STRICT results are bit-identical to the oracle and the gather kernels.
#include "mfp_march.h"
#include "mfp_source.h"

namespace CMFP_NS {

CMFP_DEV double f3_row_left(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, true);  // row_shr:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
CMFP_DEV double f3_row_right(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x101, 0xf, 0xf, true);  // row_shl:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x101, 0xf, 0xf, true);
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

// the (x, z) tile of row j: rows pz = 0 .. BZ + 2H - 1 are the planes kbase - H + pz (clamped), 16 cells in x
template <int H, int NPZ>
CMFP_DEV void f3z_fill(const Geo& g, const double* __restrict__ U, double (*dst)[NPZ], int i0, int j, int k0, int tid) {
    const int xhi = g.n[0] + g.ng[0] - 1, zlo = -g.ng[2], zhi = g.n[2] + g.ng[2] - 1;
    for (int t = tid; t < NPZ; t += 256) {
        const int pz = t >> 4, px = t & 15;
        int ci = i0 + px, ck = k0 + pz;
        ci = ci > xhi ? xhi : ci;  // cells outside the array feed only values nobody uses
        ck = ck < zlo ? zlo : (ck > zhi ? zhi : ck);
        const long c = g.at(ci, j, ck);
#pragma unroll
        for (int n = 0; n < 10; ++n) dst[n][t] = U[c + n * g.sc];
    }
}

CMFP_DEV void f3_apply_bc(const int* rule, bool outside_is_L, double* L, double* R) {
#pragma unroll
    for (int n = 0; n < 10; ++n)
        if (rule[n]) {
            if (outside_is_L)
                L[n] = rule[n] == FILL_ODD ? -R[n] : R[n];
            else
                R[n] = rule[n] == FILL_ODD ? -L[n] : L[n];
        }
}

// -------------------------------------------------------------------------------- x and y directions
// PASS 1: F1[d] on cells with index_d in [0, n_d] and the other indices in [-1, n] (the flux boxes grown
// by one transversally, MFP_CTU_hydro.cpp:254-268); threads cover [-1, n] plus a ring (sel.lo = -1).
// PASS 2: corrected face states, sweep 2, dU_xy on the valid cells (sel.lo = 0).
template <int RECON, int PASS, bool HAS_BC>
__global__ __launch_bounds__(256, 2) void k_f3d_xy(Geo g, FieldParams P, const double* __restrict__ U, Ptr3 F1, double* __restrict__ DU,
                                                    double dt, FaceBC3f bc, SelF3 sel) {
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
    const int n0 = g.n[0], n1 = g.n[1];
    const int i = sel.lo + tix * 14 - 1 + tx, j = sel.lo + tiy * 14 - 1 + ty;  // this thread's column
    const int ka = sel.lo + segi * sel.seg;
    const int kend = sel.lo + sel.np - 1;
    const int kb = (ka + sel.seg - 1 < kend) ? ka + sel.seg - 1 : kend;
    const int pc = (ty + H) * PX + (tx + H);
    const double speed = P.fastest;
    const double hx = uni(0.5 * dt / g.dx[0]), hy = uni(0.5 * dt / g.dx[1]), hz = uni(0.5 * dt / g.dx[2]);
    const double dtdx = uni(dt / g.dx[0]), dtdy = uni(dt / g.dx[1]);
    const bool inner = tx >= 1 && tx <= 14 && ty >= 1 && ty <= 14;
    const bool i_ok = i >= -1 && i <= n0, j_ok = j >= -1 && j <= n1;
    const int ic = i < -g.ng[0] ? -g.ng[0] : (i > n0 + g.ng[0] - 2 ? n0 + g.ng[0] - 2 : i);  // c + sx stays inside
    const int jc = j < -g.ng[1] ? -g.ng[1] : (j > n1 + g.ng[1] - 2 ? n1 + g.ng[1] - 2 : j);
    const long col = g.at(ic, jc, 0);
    const long sz = g.sk;

    for (int m = ka; m <= kb; ++m) {
        const long c = col + (long)m * sz;
        // transverse flux differences of this cell, dF_d = F1[d](c) - F1[d](c + e_d): requested before the
        // tile fill so that all of the iteration's memory latency is paid once
        double dFx[8], dFy[8], dFz[8];
        if (PASS == 2) {
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                dFx[n] = F1.p[0][c + n * g.sc];
                dFy[n] = F1.p[1][c + n * g.sc];
                dFz[n] = F1.p[2][c + n * g.sc];
            }
        }
        __syncthreads();
        f3_fill<H, PX, NP>(g, U, Pq, i - tx - H, j - ty - H, m, tid);
        __syncthreads();
        double lo[10], hi[10], F[10], L[10];
        if (PASS == 1) {
            // ---- x
            f3_recon_lds<RECON, NP>(Pq, pc, 1, speed, hx, lo, hi);
#pragma unroll
            for (int n = 0; n < 10; ++n) L[n] = f3_row_left(hi[n]);
            FieldF<RECON>::riemann(P, 0, L, lo, F);
            {
                double dF[8];  // F_x(i) - F_x(i + 1): what the transverse corrections of this cell use
#pragma unroll
                for (int n = 0; n < 8; ++n) dF[n] = F[n] - f3_row_right(F[n]);
                if (inner && i >= 0 && i < n0 && j_ok) {
#pragma unroll
                    for (int n = 0; n < 8; ++n) F1.p[0][c + n * g.sc] = dF[n];
                }
            }
            // ---- y
            f3_recon_lds<RECON, NP>(Pq, pc, PX, speed, hy, lo, hi);
#pragma unroll
            for (int n = 0; n < 10; ++n) bufR[n][tid] = hi[n];
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 10; ++n) L[n] = bufR[n][ty >= 1 ? tid - BX : tid];
            FieldF<RECON>::riemann(P, 1, L, lo, F);
            __syncthreads();  // every thread has read its neighbour's face state: the buffer takes the fluxes
#pragma unroll
            for (int n = 0; n < 8; ++n) bufR[n][tid] = F[n];
            __syncthreads();
            if (inner && j >= 0 && j < n1 && i_ok) {
#pragma unroll
                for (int n = 0; n < 8; ++n) F1.p[1][c + n * g.sc] = F[n] - bufR[n][tid + BX];
            }
        } else {
            double du[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) du[n] = 0.0;
            // ---- x faces: corrected with dF_y then dF_z
            f3_recon_lds<RECON, NP>(Pq, pc, 1, speed, hx, lo, hi);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                hi[n] += hy * dFy[n];
                lo[n] += hy * dFy[n];
                hi[n] += hz * dFz[n];
                lo[n] += hz * dFz[n];
            }
#pragma unroll
            for (int n = 0; n < 10; ++n) L[n] = f3_row_left(hi[n]);
            if (HAS_BC && bc.touch[0][0] && i == 0) f3_apply_bc(bc.rule[0][0], true, L, lo);
            if (HAS_BC && bc.touch[0][1] && i == n0) f3_apply_bc(bc.rule[0][1], false, L, lo);
            FieldF<RECON>::riemann(P, 0, L, lo, F);
#pragma unroll
            for (int n = 0; n < 8; ++n) du[n] += dtdx * (F[n] - f3_row_right(F[n]));
            // ---- y faces: corrected with dF_z then dF_x
            f3_recon_lds<RECON, NP>(Pq, pc, PX, speed, hy, lo, hi);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                hi[n] += hz * dFz[n];
                lo[n] += hz * dFz[n];
                hi[n] += hx * dFx[n];
                lo[n] += hx * dFx[n];
            }
#pragma unroll
            for (int n = 0; n < 10; ++n) bufR[n][tid] = hi[n];
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 10; ++n) L[n] = bufR[n][ty >= 1 ? tid - BX : tid];
            if (HAS_BC && bc.touch[1][0] && j == 0) f3_apply_bc(bc.rule[1][0], true, L, lo);
            if (HAS_BC && bc.touch[1][1] && j == n1) f3_apply_bc(bc.rule[1][1], false, L, lo);
            FieldF<RECON>::riemann(P, 1, L, lo, F);
            __syncthreads();  // every thread has read its neighbour's face state: the buffer takes the fluxes
#pragma unroll
            for (int n = 0; n < 8; ++n) bufR[n][tid] = F[n];
            __syncthreads();
            if (inner && i < n0 && j < n1) {
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    du[n] += dtdy * (F[n] - bufR[n][tid + BX]);
                    DU[c + n * g.sc] = du[n];
                }
            }
        }
    }
}

// -------------------------------------------------------------------------------------- z direction
// Thread (tx, tz) of tile (tix, tiz): cell i = lo + 16 tix + tx, k = 14 tiz - 1 + tz; the workgroup marches
// along y.  Faces k (between k - 1 and k) are computed by tz >= 1, cells are updated by tz in [1, 14].
// SRC > 0 (pass 2): the explicit Strang source stage C of SRC species on the cells just updated (see mfp_source.h)
template <int RECON, int PASS, bool HAS_BC, int SRC = 0>
__global__ __launch_bounds__(256, SRC > 0 ? 2 : 3) void k_f3d_z(Geo g, FieldParams P, const double* __restrict__ Uin, double* __restrict__ Uout,
                                                   Ptr3 F1, const double* __restrict__ DU, double dt, FaceBC3f bc, SelF3 sel,
                                                   typename EpiArg<SRC>::type epi) {
    constexpr int H = ReconTraits<RECON>::SL / 2, BX = 16, NPZ = (16 + 2 * H) * BX, C = 256;
    __shared__ double Pz[10][NPZ];
    __shared__ double bufR[10][C];
    dt = launch_dt(g, dt);
    const int tx = threadIdx.x, tz = threadIdx.y, tid = tz * BX + tx;
    const int per = (sel.n + 7) >> 3;
    const int wg = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (wg >= sel.n) return;
    const int tiles = sel.gx * sel.gy;  // gy: tiles along z
    const int segi = wg / tiles, tile = wg - segi * tiles;
    const int tiz = tile / sel.gx, tix = tile - tiz * sel.gx;
    const int n0 = g.n[0], n2 = g.n[2];
    const int i = sel.lo + tix * 16 + tx, k = tiz * 14 - 1 + tz;
    const int ja = sel.lo + segi * sel.seg;
    const int jend = sel.lo + sel.np - 1;
    const int jb = (ja + sel.seg - 1 < jend) ? ja + sel.seg - 1 : jend;
    const int pc = (tz + H) * BX + tx;
    const double speed = P.fastest;
    const double hx = uni(0.5 * dt / g.dx[0]), hy = uni(0.5 * dt / g.dx[1]), hz = uni(0.5 * dt / g.dx[2]);
    const double dtdx = uni(dt / g.dx[0]), dtdy = uni(dt / g.dx[1]), dtdz = uni(dt / g.dx[2]);
    const int ic = i > n0 + g.ng[0] - 2 ? n0 + g.ng[0] - 2 : i;  // i >= -1; c + sx stays inside
    const int kc = k > n2 + g.ng[2] - 1 ? n2 + g.ng[2] - 1 : k;  // k >= -1
    const long base = g.at(ic, 0, kc);
    const long sy = g.sj;
    double src_speed[SRC > 0 ? SRC : 1];
    SrcDt src_dt;
#pragma unroll
    for (int s = 0; s < (SRC > 0 ? SRC : 1); ++s) src_speed[s] = DBL_MIN;

    for (int j = ja; j <= jb; ++j) {
        const long c = base + (long)j * sy;
        // requested before the tile fill so that all of the iteration's memory latency is paid once
        double dFx[8], dFy[8], du[8];
        if (PASS == 2) {
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                dFx[n] = F1.p[0][c + n * g.sc];
                dFy[n] = F1.p[1][c + n * g.sc];
                du[n] = DU[c + n * g.sc];
            }
        }
        __syncthreads();
        f3z_fill<H, NPZ>(g, Uin, Pz, i - tx, j, k - tz - H, tid);
        __syncthreads();
        double lo[10], hi[10], F[10], L[10];
        f3_recon_lds<RECON, NPZ>(Pz, pc, BX, speed, hz, lo, hi);
        if (PASS == 2) {
            // z faces: corrected with dF_x then dF_y
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                hi[n] += hx * dFx[n];
                lo[n] += hx * dFx[n];
                hi[n] += hy * dFy[n];
                lo[n] += hy * dFy[n];
            }
        }
#pragma unroll
        for (int n = 0; n < 10; ++n) bufR[n][tid] = hi[n];
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 10; ++n) L[n] = bufR[n][tz >= 1 ? tid - BX : tid];
        if (PASS == 2) {
            if (HAS_BC && bc.touch[2][0] && k == 0) f3_apply_bc(bc.rule[2][0], true, L, lo);
            if (HAS_BC && bc.touch[2][1] && k == n2) f3_apply_bc(bc.rule[2][1], false, L, lo);
        }
        FieldF<RECON>::riemann(P, 2, L, lo, F);
        if (PASS == 1) {
            __syncthreads();  // every thread has read its neighbour's face state: the buffer takes the fluxes
#pragma unroll
            for (int n = 0; n < 8; ++n) bufR[n][tid] = F[n];
            __syncthreads();
            if (tz >= 1 && tz <= 14 && k < n2 && i <= n0) {  // F_z(k) - F_z(k + 1) of cell k (k >= 0 here)
#pragma unroll
                for (int n = 0; n < 8; ++n) F1.p[2][c + n * g.sc] = F[n] - bufR[n][tid + BX];
            }
        } else {
            __syncthreads();  // every thread has read its neighbour's face state: the buffer takes the fluxes
#pragma unroll
            for (int n = 0; n < 8; ++n) bufR[n][tid] = F[n];
            __syncthreads();
            if (tz >= 1 && tz <= 14 && k < n2 && i < n0) {
                double Unew[10];
#pragma unroll
                for (int n = 0; n < 10; ++n) {
                    double d = 0.0;
                    if (n < 8) {
                        d = du[n];  // 0 + dt/dx (F_x - F_x+) + dt/dy (F_y - F_y+), accumulated in that order
                        d += dtdz * (F[n] - bufR[n][tid + BX]);
                    } else {  // F[mu] = F[ep] = 0 (MFP_field_RH.cpp:33)
                        d += dtdx * (0.0 - 0.0);
                        d += dtdy * (0.0 - 0.0);
                        d += dtdz * (0.0 - 0.0);
                    }
                    Unew[n] = Pz[n][pc] + d;  // the tile holds Uin
                }
                if constexpr (SRC > 0) source_epilogue_cell<SRC>(g, epi, dt, c, Unew, src_speed, src_dt);
#pragma unroll
                for (int n = 0; n < 10; ++n) Uout[c + n * g.sc] = Unew[n];
            }
        }
    }
    if constexpr (SRC > 0) {
        __syncthreads();  // the buffer is free
        source_epilogue_reduce<SRC>(epi, src_speed, src_dt, &bufR[0][0], tid, C / 64, wg & 255);
    }
}

// defined in mfp_march_field3d.hip: both passes on the marching kernels (O6, vacuum block); 0 where it does not apply
int ctu_field3d_march(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, const Ptr3& dF, double* DU, double dt,
                      const int touches[3][2], const SourceEpi* epi, hipStream_t st);

// tiles of `tw` cells over `nx` cells in the first tile dimension and 14 over `ny` in the second, marching over `np`
static SelF3 make_sel(int nx, int tw, int ny, int np, int lo) {
    SelF3 s;
    s.lo = lo;
    s.gx = (nx + tw - 1) / tw;
    s.gy = (ny + 13) / 14;
    s.np = np;
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
static int launch_f3d(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, const Ptr3& F1, double* DU, double dt,
                      const int touches[3][2], const SourceEpi* epi, hipStream_t st) {
    constexpr int H = ReconTraits<RECON>::SL / 2;
    if (g.ng[0] < H + 1 || g.ng[1] < H + 1 || g.ng[2] < H + 1) return 0;
    if (!Uin) return 1;  // coverage probe
    FaceBC3f bc;
    bool has_bc = g.bc_instance != 0;
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
    auto grid = [](const SelF3& s) { return dim3(((s.n + 7) / 8) * 8); };
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    // pass 1: cells [-1, n] in every dimension (z faces: 0 .. n2)
    if (RECON == 5 && ctu_field3d_march(g, sp, Uin, Uout, F1, DU, dt, touches, epi, st)) return 1;
    const SelF3 a = make_sel(n0 + 2, 14, n1 + 2, n2 + 2, -1);
    hipLaunchKernelGGL((k_f3d_xy<RECON, 1, false>), grid(a), block, 0, st, g, sp.fp, Uin, F1, DU, dt, bc, a);
    const SelF3 b = make_sel(n0 + 2, 16, n2 + 1, n1 + 2, -1);
    hipLaunchKernelGGL((k_f3d_z<RECON, 1, false>), grid(b), block, 0, st, g, sp.fp, Uin, Uout, F1, DU, dt, bc, b, NoEpi{});
    // pass 2: the valid cells
    const SelF3 c = make_sel(n0, 14, n1, n2, 0);
    const SelF3 d = make_sel(n0, 16, n2, n1, 0);
    if (has_bc)
        hipLaunchKernelGGL((k_f3d_xy<RECON, 2, true>), grid(c), block, 0, st, g, sp.fp, Uin, F1, DU, dt, bc, c);
    else
        hipLaunchKernelGGL((k_f3d_xy<RECON, 2, false>), grid(c), block, 0, st, g, sp.fp, Uin, F1, DU, dt, bc, c);
#define F3Z(BC, NS, E) hipLaunchKernelGGL((k_f3d_z<RECON, 2, BC, NS>), grid(d), block, 0, st, g, sp.fp, Uin, Uout, F1, DU, dt, bc, d, E)
    if (epi && epi->ns == 2) {
        if (has_bc) F3Z(true, 2, *epi); else F3Z(false, 2, *epi);
    } else if (epi && epi->ns == 1) {
        if (has_bc) F3Z(true, 1, *epi); else F3Z(false, 1, *epi);
    } else {
        if (has_bc) F3Z(true, 0, NoEpi{}); else F3Z(false, 0, NoEpi{});
    }
#undef F3Z
    return 1;
}

// Update of a 3-D field state from LDS plane tiles; F1: sweep-1 flux scratch, DU: dU_xy scratch (8 planes
// each); epi (may be null): the explicit Strang source stage C as the epilogue of the last launch.
// Returns 0 where the option set is outside these kernels (the caller then uses the gather kernels).
int ctu_field3d(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, const Ptr3& F1, double* DU, int do_ctu,
                double dt, const int touches[3][2], const SourceEpi* epi, hipStream_t st) {
    if (g.dim != 3 || sp.kind != 1 || !do_ctu || sp.flux != 10) return 0;
    if (epi && (epi->a.implicit || epi->a.coll.on || epi->ns < 1 || epi->ns > 2)) return 0;
    if (sp.recon == 5) return launch_f3d<5>(g, sp, Uin, Uout, F1, DU, dt, touches, epi, st);
    if (sp.recon == 1) return launch_f3d<1>(g, sp, Uin, Uout, F1, DU, dt, touches, epi, st);
    if (sp.recon == 2) return launch_f3d<2>(g, sp, Uin, Uout, F1, DU, dt, touches, epi, st);
    return 0;
}

}  // namespace CMFP_NS
