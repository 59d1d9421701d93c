// mfp_march_field3d.hip — pass 2 of the 3-D field update (mfp_field3d.hip) as two wave-marching kernels
// (O6, vacuum materials), with the Strang source stage C as the epilogue of the second.
//
// Pass 1 (mfp_field3d.hip) leaves dF_d = F1_d(c) - F1_d(c + e_d) of every cell.  Pass 2 of the plane-tile
// kernels re-read U through 22 x 22 / 22 x 16 LDS tiles with 128-byte rows (measured 21 GB per 256^3 step for
// 8 GB needed).  Here one wavefront owns 64 columns along x and marches along the direction of its stencil
// with a six-line window in LDS (as mfp_march_field.hip does in 2-D), so every array is read once in 512-byte
// rows and the O6 interface values are computed once:
//
//   k_f3m_z   march along z at fixed (x strip, j):  z-face states of plane k, corrected with dF_x then dF_y
//             (EulerianState::correct_face_prim MFP_eulerian.cpp:1083-1093: (d1+1)%3 then (d1+2)%3), domain-face
//             BCs, Riemann sweep 2, G_z = dt/dz (F'_z(k) - F'_z(k+1)) -> DU
//   k_f3m_yx  march along y at fixed (x strip, k):  y-face states (dF_z then dF_x) and x-face states (dF_y then
//             dF_z; x stencil from the window row), sweep 2, Uout = Uin + ((0 + dt/dx dF'_x) + dt/dy dF'_y) + G_z
//             (calc_divergence order, MFP_eulerian.cpp:1614-1662), then stage C of the species at that cell
//
// mu, ep are constants of the launch (see mfp_march_field.hip).  Same device functions and operation order per
// value as the plane-tile kernels: STRICT results are bit-identical to them and to the oracle.
#include "mfp_march.h"
#include "mfp_source.h"

namespace CMFP_NS {

struct FaceBC3m {
    int touch[3][2];
    int rule[3][2][10];
};
struct SelF3m {
    int ns;      // strips along x
    int nline;   // lines (j for the z march, k for the y march)
    int nseg;    // segments along the marching direction
    int seg;     // cells per segment
    int nwaves;
};

static constexpr int F3M_NR = 6;
static constexpr int F3M_PAD = 8;

CMFP_DEV double f3m_interface(double s1, double s2, double s3, double s4, double s5, double s6) {
    return (37.0 / 60.0) * (s3 + s4) - (2.0 / 15.0) * (s2 + s5) + (1.0 / 60.0) * (s1 + s6);
}
CMFP_DEV void f3m_bc(const int* rule, bool outside_is_L, double* L, double* R) {
#pragma unroll
    for (int n = 0; n < 10; ++n)
        if (rule[n]) {
            if (outside_is_L)
                L[n] = rule[n] == FILL_ODD ? -R[n] : R[n];
            else
                R[n] = rule[n] == FILL_ODD ? -L[n] : L[n];
        }
}
// face states of mu, ep along a direction with half-step factor h: constants of the launch
CMFP_DEV void f3m_material_faces(double speed, double h, double& mlo, double& mhi) {
#ifdef CMFP_FASTMATH
    mlo = mhi = 1.0;
    (void)speed;
    (void)h;
#else
    const double flat[7] = {1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0};
    recon_faces<5>(flat, mlo, mhi);
    trace_faces(speed, h, 1.0, mlo, mhi);
#endif
}

// ------------------------------------------------------------------------------------------- z direction
template <bool HAS_BC>
__global__ __launch_bounds__(64, 1) void k_f3m_z(Geo g, FieldParams P, const double* __restrict__ U, Ptr3 dF, double* __restrict__ DU,
                                                  double dt, FaceBC3m bc, SelF3m sel) {
    typedef FieldF<5> Phys;
    __shared__ double ring_[F3M_NR * 8 * 64 + 2 * F3M_PAD];
    double* const ring = ring_ + F3M_PAD;
    dt = launch_dt(g, dt);
    const int lane = threadIdx.x;
    const int per = (sel.nwaves + 7) >> 3;
    int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (w >= sel.nwaves) return;
    const int strip = w % sel.ns;
    w /= sel.ns;
    const int j = w % sel.nline - 1, segi = w / sel.nline;  // lines j = -1 .. n1: the ring lines only leave dF_z
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const int i = strip * 56 - 4 + lane;  // dF_x of the cell comes from the x stencil of the window's centre line
    const int ic = i < n0 + 3 ? i : n0 + 3;
    const bool upd = lane >= 4 && lane <= 59 && i < n0 && j >= 0 && j < n1;
    // dF_z (sweep 1, no x stencil) is left for the cells of the block grown by one in x and y: k_f3m_yx corrects the
    // face states of those ring cells too
    const bool zw = (lane >= 4 && lane <= 59 && i <= n0) || (lane == 3 && i == -1) || (lane == 60 && i == n0);
    const int ka = segi * sel.seg;
    const int kb = (ka + sel.seg - 1 < n2 - 1) ? ka + sel.seg - 1 : n2 - 1;
    const double speed = P.fastest;
    const double hx = uni(0.5 * dt / g.dx[0]), hy = uni(0.5 * dt / g.dx[1]), hz = uni(0.5 * dt / g.dx[2]);
    const double dtdz = uni(dt / g.dx[2]);
    double mxlo, mxhi;
    f3m_material_faces(speed, hx, mxlo, mxhi);
    const long col = g.at(ic, j, 0);
    const long sz = g.sk;
    const int pl0 = ka - 4;
    double mlo, mhi;
    f3m_material_faces(speed, hz, mlo, mhi);

    auto load_plane = [&](int k, double* V) {
        const long c = col + (long)k * sz;
#pragma unroll
        for (int n = 0; n < 8; ++n) V[n] = U[c + n * g.sc];
    };
    auto load_dF = [&](int k, double* b) {
        const long c = col + (long)k * sz;
#pragma unroll
        for (int n = 0; n < 8; ++n) b[n] = dF.p[1][c + n * g.sc];
    };
    auto slot_of = [&](int k) { return (k - pl0) % F3M_NR; };
    auto at = [&](int slot, int n) -> double& { return ring[(slot * 8 + n) * 64 + lane]; };
    auto atl = [&](int slot, int n, int l) -> double& { return ring[(slot * 8 + n) * 64 + l]; };

    double Un[8], dyn[8];
    for (int s = 0; s < F3M_NR; ++s) {
        load_plane(pl0 + s, Un);
#pragma unroll
        for (int n = 0; n < 8; ++n) at(s, n) = Un[n];
    }
    load_plane(ka + 2, Un);
    load_dF(ka - 1, dyn);
    double Iz_p[8], zhiC_p[8], fz_p[8], zhiT_p[8], f1z_p[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        Iz_p[n] = f3m_interface(at(0, n), at(1, n), at(2, n), at(3, n), at(4, n), at(5, n));
        zhiC_p[n] = fz_p[n] = zhiT_p[n] = f1z_p[n] = 0.0;
    }

    for (int k = ka - 1; k <= kb + 1; ++k) {
        double dFx[8], dFy[8];
        __syncthreads();  // every lane has finished reading the slot that is overwritten
        {
            const int sn = slot_of(k + 3);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                at(sn, n) = Un[n];
                dFy[n] = dyn[n];
            }
            const int kn = k + 4 < kb + 4 ? k + 4 : kb + 4;
            load_plane(kn, Un);
            const int kd = k + 1 < kb + 1 ? k + 1 : kb + 1;
            load_dF(kd, dyn);
        }
        __syncthreads();
        const int s0 = slot_of(k - 2), s1 = slot_of(k - 1), s2 = slot_of(k), s3 = slot_of(k + 1), s4 = slot_of(k + 2), s5 = slot_of(k + 3);
        {
            // dF_x of this cell: sweep-1 x fluxes of the line (j, k), recomputed from the window's centre line
            // (the same traced face states pass 1 would have used; nothing is stored for x)
            double xlo[10], xhi[10], L[10], fx[10];
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                const double centre = at(s2, n);
                const double Ix = f3m_interface(atl(s2, n, lane - 2), atl(s2, n, lane - 1), centre, atl(s2, n, lane + 1), atl(s2, n, lane + 2),
                                                atl(s2, n, lane + 3));
                xlo[n] = lane_left(Ix);
                xhi[n] = Ix;
                trace_faces(speed, hx, centre, xlo[n], xhi[n]);
            }
            xlo[8] = xlo[9] = mxlo;
            xhi[8] = xhi[9] = mxhi;
#pragma unroll
            for (int n = 0; n < 8; ++n) L[n] = lane_left(xhi[n]);
            L[8] = L[9] = mxhi;
            Phys::riemann(P, 0, L, xlo, fx);
#pragma unroll
            for (int n = 0; n < 8; ++n) dFx[n] = fx[n] - lane_right(fx[n]);
        }
        double zlo[10], zhi[10];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const double centre = at(s2, n);
            const double Iz = f3m_interface(at(s0, n), at(s1, n), centre, at(s3, n), at(s4, n), at(s5, n));
            zlo[n] = Iz_p[n];
            zhi[n] = Iz;
            Iz_p[n] = Iz;
            trace_faces(speed, hz, centre, zlo[n], zhi[n]);
        }
        zlo[8] = zlo[9] = mlo;
        zhi[8] = zhi[9] = mhi;
        {
            // Riemann sweep 1 on the low z-face with the traced, uncorrected states: dF_z of plane k - 1 for k_f3m_yx
            // (the first plane of a segment solves on the zero-initialised carried state; nothing of it is stored)
            double L[10], f1z[10];
#pragma unroll
            for (int n = 0; n < 8; ++n) L[n] = zhiT_p[n];
            L[8] = L[9] = mhi;
            Phys::riemann(P, 2, L, zlo, f1z);
            if (k > ka && zw) {
                const long c = col + (long)(k - 1) * sz;
#pragma unroll
                for (int n = 0; n < 8; ++n) dF.p[2][c + n * g.sc] = f1z_p[n] - f1z[n];
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                zhiT_p[n] = zhi[n];
                f1z_p[n] = f1z[n];
            }
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            // z faces: corrected with dF_x then dF_y
            zhi[n] += hx * dFx[n];
            zlo[n] += hx * dFx[n];
            zhi[n] += hy * dFy[n];
            zlo[n] += hy * dFy[n];
        }
        double fz[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (k >= ka) {
            double L[10];
#pragma unroll
            for (int n = 0; n < 8; ++n) L[n] = zhiC_p[n];
            L[8] = L[9] = mhi;
            if (HAS_BC && bc.touch[2][0] && k == 0) f3m_bc(bc.rule[2][0], true, L, zlo);
            if (HAS_BC && bc.touch[2][1] && k == n2) f3m_bc(bc.rule[2][1], false, L, zlo);
            Phys::riemann(P, 2, L, zlo, fz);
        }
        if (k > ka && upd) {
            const long c = col + (long)(k - 1) * sz;
#pragma unroll
            for (int n = 0; n < 8; ++n) DU[c + n * g.sc] = dtdz * (fz_p[n] - fz[n]);
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            zhiC_p[n] = zhi[n];
            fz_p[n] = fz[n];
        }
    }
}

// ------------------------------------------------------------------------------------ y and x directions
template <bool HAS_BC, int SRC>
__global__ __launch_bounds__(64, 1) void k_f3m_yx(Geo g, FieldParams P, const double* __restrict__ Uin, double* __restrict__ Uout, Ptr3 dF,
                                                   const double* __restrict__ DU, double dt, FaceBC3m bc, SelF3m sel,
                                                   typename EpiArg<SRC>::type epi) {
    typedef FieldF<5> Phys;
    __shared__ double ring_[F3M_NR * 8 * 64 + 2 * F3M_PAD];
    double* const ring = ring_ + F3M_PAD;
    dt = launch_dt(g, dt);
    const int lane = threadIdx.x;
    const int per = (sel.nwaves + 7) >> 3;
    int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    double src_speed[SRC > 0 ? SRC : 1];
    SrcDt src_dt;
#pragma unroll
    for (int s = 0; s < (SRC > 0 ? SRC : 1); ++s) src_speed[s] = DBL_MIN;
    if (w < sel.nwaves) {
        const int strip = w % sel.ns;
        w /= sel.ns;
        const int k = w % sel.nline, segi = w / sel.nline;
        const int n0 = g.n[0], n1 = g.n[1];
        const int i = strip * 56 - 4 + lane;
        const int ic = i < n0 + 3 ? i : n0 + 3;
        const bool upd = lane >= 4 && lane <= 59 && i < n0;
        const int ja = segi * sel.seg;
        const int jb = (ja + sel.seg - 1 < n1 - 1) ? ja + sel.seg - 1 : n1 - 1;
        const double speed = P.fastest;
        const double hx = uni(0.5 * dt / g.dx[0]), hy = uni(0.5 * dt / g.dx[1]), hz = uni(0.5 * dt / g.dx[2]);
        const double dtdx = uni(dt / g.dx[0]), dtdy = uni(dt / g.dx[1]);
        const long col = g.at(ic, 0, k);
        const long sy = g.sj;
        const int row0 = ja - 4;
        double mlo[2], mhi[2];
        f3m_material_faces(speed, hx, mlo[0], mhi[0]);
        f3m_material_faces(speed, hy, mlo[1], mhi[1]);

        auto load_row = [&](int row, double* V) {
            const long c = col + (long)row * sy;
#pragma unroll
            for (int n = 0; n < 8; ++n) V[n] = Uin[c + n * g.sc];
        };
        auto load_dF = [&](int row, double* cz) {
            const long c = col + (long)row * sy;
#pragma unroll
            for (int n = 0; n < 8; ++n) cz[n] = dF.p[2][c + n * g.sc];
        };
        auto slot_of = [&](int row) { return (row - row0) % F3M_NR; };
        auto at = [&](int slot, int n, int l) -> double& { return ring[(slot * 8 + n) * 64 + l]; };

        double Un[8], dzn[8], gzn[8];
        for (int s = 0; s < F3M_NR; ++s) {
            load_row(row0 + s, Un);
#pragma unroll
            for (int n = 0; n < 8; ++n) at(s, n, lane) = Un[n];
        }
        load_row(ja + 2, Un);
        load_dF(ja - 1, dzn);
#pragma unroll
        for (int n = 0; n < 8; ++n) gzn[n] = 0.0;
        // carried from row r - 1: the traced x-face states (corrected one iteration later, when dF_y of their row is
        // known: dF_y(r - 1) = F1_y(r - 1) - F1_y(r) is recomputed here, not read), dF_z of that row, the traced
        // uncorrected high y-face and F1_y on the low y-face (sweep 1), the corrected high y-face and F'_y (sweep 2)
        double Iy_p[8], yhiC_p[8], fy_p[8], gz[8], xlo_p[10], xhi_p[10], dFz_p[8], yhiT_p[8], f1y_p[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            Iy_p[n] = f3m_interface(at(0, n, lane), at(1, n, lane), at(2, n, lane), at(3, n, lane), at(4, n, lane), at(5, n, lane));
            yhiC_p[n] = fy_p[n] = gz[n] = xlo_p[n] = xhi_p[n] = dFz_p[n] = yhiT_p[n] = f1y_p[n] = 0.0;
        }
        xlo_p[8] = xlo_p[9] = mlo[0];
        xhi_p[8] = xhi_p[9] = mhi[0];
        double sp_cur[SRC > 0 ? SRC : 1][5], sp_nxt[SRC > 0 ? SRC : 1][5];
#pragma unroll
        for (int s = 0; s < (SRC > 0 ? SRC : 1); ++s)
#pragma unroll
            for (int n = 0; n < 5; ++n) sp_cur[s][n] = sp_nxt[s][n] = 1.0;

        for (int r = ja - 1; r <= jb + 1; ++r) {
            double dFx[8], dFy_p[8], dFz[8];
            __syncthreads();  // every lane has finished reading the slot that is overwritten
            {
                const int sn = slot_of(r + 3);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    at(sn, n, lane) = Un[n];
                    dFz[n] = dzn[n];
                    gz[n] = gzn[n];  // G_z of row r - 1 (requested in the previous iteration)
                }
                const int rn = r + 4 < jb + 4 ? r + 4 : jb + 4;
                load_row(rn, Un);
                const int rd = r + 1 < jb + 1 ? r + 1 : jb + 1;
                load_dF(rd, dzn);
                if (upd && r >= ja && r <= jb) {
                    const long c = col + (long)r * sy;
#pragma unroll
                    for (int n = 0; n < 8; ++n) gzn[n] = DU[c + n * g.sc];
                    if constexpr (SRC > 0) source_epilogue_load<SRC>(g, epi, c, sp_nxt);
                }
            }
            __syncthreads();
            const int s0 = slot_of(r - 2), s1 = slot_of(r - 1), s2 = slot_of(r), s3 = slot_of(r + 1), s4 = slot_of(r + 2), s5 = slot_of(r + 3);
            // ---- x faces of row r, traced; their sweep-1 fluxes give dF_x of the cell (recomputed: nothing is stored for x)
            double xlo[10], xhi[10];
            {
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const double centre = at(s2, n, lane);
                    const double Ix = f3m_interface(at(s2, n, lane - 2), at(s2, n, lane - 1), centre, at(s2, n, lane + 1), at(s2, n, lane + 2),
                                                    at(s2, n, lane + 3));
                    xlo[n] = lane_left(Ix);
                    xhi[n] = Ix;
                    trace_faces(speed, hx, centre, xlo[n], xhi[n]);
                }
                xlo[8] = xlo[9] = mlo[0];
                xhi[8] = xhi[9] = mhi[0];
                double L[10], fx[10];
#pragma unroll
                for (int n = 0; n < 8; ++n) L[n] = lane_left(xhi[n]);
                L[8] = L[9] = mhi[0];
                Phys::riemann(P, 0, L, xlo, fx);
#pragma unroll
                for (int n = 0; n < 8; ++n) dFx[n] = fx[n] - lane_right(fx[n]);
            }
            // ---- y faces of row r: trace, corrected with dF_z then dF_x; BCs; sweep 2 on the low y-face
            double fy[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            {
                double ylo[10], yhi[10];
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const double centre = at(s2, n, lane);
                    const double Iy = f3m_interface(at(s0, n, lane), at(s1, n, lane), centre, at(s3, n, lane), at(s4, n, lane), at(s5, n, lane));
                    ylo[n] = Iy_p[n];
                    yhi[n] = Iy;
                    Iy_p[n] = Iy;
                    trace_faces(speed, hy, centre, ylo[n], yhi[n]);
                }
                ylo[8] = ylo[9] = mlo[1];
                yhi[8] = yhi[9] = mhi[1];
                {
                    // Riemann sweep 1 on the low y-face with the traced, uncorrected states: dF_y of row r - 1 (the
                    // first rows of a segment solve on the zero-initialised carried state; nothing of it is stored)
                    double L[10], f1y[10];
#pragma unroll
                    for (int n = 0; n < 8; ++n) L[n] = yhiT_p[n];
                    L[8] = L[9] = mhi[1];
                    Phys::riemann(P, 1, L, ylo, f1y);
#pragma unroll
                    for (int n = 0; n < 8; ++n) {
                        dFy_p[n] = f1y_p[n] - f1y[n];
                        yhiT_p[n] = yhi[n];
                        f1y_p[n] = f1y[n];
                    }
                }
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    yhi[n] += hz * dFz[n];
                    ylo[n] += hz * dFz[n];
                    yhi[n] += hx * dFx[n];
                    ylo[n] += hx * dFx[n];
                }
                if (r >= ja) {
                    double L[10];
#pragma unroll
                    for (int n = 0; n < 8; ++n) L[n] = yhiC_p[n];
                    L[8] = L[9] = mhi[1];
                    if (HAS_BC && bc.touch[1][0] && r == 0) f3m_bc(bc.rule[1][0], true, L, ylo);
                    if (HAS_BC && bc.touch[1][1] && r == n1) f3m_bc(bc.rule[1][1], false, L, ylo);
                    Phys::riemann(P, 1, L, ylo, fy);
                }
#pragma unroll
                for (int n = 0; n < 8; ++n) yhiC_p[n] = yhi[n];
            }
            // ---- x faces of row r - 1: corrected with dF_y then dF_z; BCs; sweep 2; x part of the divergence
            double X_p[8];
            {
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    xhi_p[n] += hy * dFy_p[n];
                    xlo_p[n] += hy * dFy_p[n];
                    xhi_p[n] += hz * dFz_p[n];
                    xlo_p[n] += hz * dFz_p[n];
                }
                double L[10], fx[10];
#pragma unroll
                for (int n = 0; n < 8; ++n) L[n] = lane_left(xhi_p[n]);
                L[8] = L[9] = mhi[0];
                if (HAS_BC && bc.touch[0][0] && i == 0) f3m_bc(bc.rule[0][0], true, L, xlo_p);
                if (HAS_BC && bc.touch[0][1] && i == n0) f3m_bc(bc.rule[0][1], false, L, xlo_p);
                Phys::riemann(P, 0, L, xlo_p, fx);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    double du = 0.0;
                    du += dtdx * (fx[n] - lane_right(fx[n]));
                    X_p[n] = du;
                }
            }
            // ---- divergence of row r - 1 and update
            if (r > ja) {
                double Unew[10];
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    double du = X_p[n];                    // 0 + dt/dx (F'_x - F'_x+)
                    du += dtdy * (fy_p[n] - fy[n]);
                    du += gz[n];                           // + dt/dz (F'_z - F'_z+)
                    Unew[n] = at(s1, n, lane) + du;
                }
                Unew[8] = Unew[9] = 1.0;
                if (upd) {
                    const long c = col + (long)(r - 1) * sy;
                    if constexpr (SRC > 0) source_epilogue_cell_pre<SRC>(g, epi, dt, c, Unew, sp_cur, src_speed, src_dt);
#pragma unroll
                    for (int n = 0; n < 8; ++n) Uout[c + n * g.sc] = Unew[n];
                }
            }
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                fy_p[n] = fy[n];
                xlo_p[n] = xlo[n];  // traced, uncorrected: corrected in the next iteration
                xhi_p[n] = xhi[n];
                dFz_p[n] = dFz[n];
            }
            xlo_p[8] = xlo_p[9] = mlo[0];
            if constexpr (SRC > 0) {
#pragma unroll
                for (int s = 0; s < SRC; ++s)
#pragma unroll
                    for (int n = 0; n < 5; ++n) sp_cur[s][n] = sp_nxt[s][n];
            }
        }
    }
    if constexpr (SRC > 0) {
        __syncthreads();
        source_epilogue_reduce<SRC>(epi, src_speed, src_dt, ring, lane, 1, (int)(blockIdx.x & 255));
    }
}

// --------------------------------------------------------------------------------------------- pass 1
// dF_z = F1_z(k) - F1_z(k + 1) for k in [0, n2) on the columns (i, j) in [-1, n0] x [-1, n1] (the flux boxes grown by
// one transversally, MFP_CTU_hydro.cpp:254-268): march along z at fixed (x strip, j); lane l <-> i = -1 + 64 strip + l
__global__ __launch_bounds__(64, 2) void k_f3m_z1(Geo g, FieldParams P, const double* __restrict__ U, double* __restrict__ dFz, double dt,
                                                   SelF3m sel) {
    typedef FieldF<5> Phys;
    __shared__ double ring_[F3M_NR * 8 * 64 + 2 * F3M_PAD];
    double* const ring = ring_ + F3M_PAD;
    dt = launch_dt(g, dt);
    const int lane = threadIdx.x;
    const int per = (sel.nwaves + 7) >> 3;
    int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (w >= sel.nwaves) return;
    const int strip = w % sel.ns;
    w /= sel.ns;
    const int j = w % sel.nline - 1, segi = w / sel.nline;
    const int n0 = g.n[0], n2 = g.n[2];
    const int i = -1 + strip * 64 + lane;
    const int ic = i <= n0 ? i : n0;
    const bool upd = i <= n0;
    const int ka = segi * sel.seg;
    const int kb = (ka + sel.seg - 1 < n2 - 1) ? ka + sel.seg - 1 : n2 - 1;
    const double speed = P.fastest;
    const double hz = uni(0.5 * dt / g.dx[2]);
    const long col = g.at(ic, j, 0);
    const long sz = g.sk;
    const int pl0 = ka - 4;
    double mlo, mhi;
    f3m_material_faces(speed, hz, mlo, mhi);
    auto load_plane = [&](int k, double* V) {
        const long c = col + (long)k * sz;
#pragma unroll
        for (int n = 0; n < 8; ++n) V[n] = U[c + n * g.sc];
    };
    auto slot_of = [&](int k) { return (k - pl0) % F3M_NR; };
    auto at = [&](int slot, int n) -> double& { return ring[(slot * 8 + n) * 64 + lane]; };
    double Un[8], Un2[8];
    for (int s = 0; s < F3M_NR; ++s) {
        load_plane(pl0 + s, Un);
#pragma unroll
        for (int n = 0; n < 8; ++n) at(s, n) = Un[n];
    }
    load_plane(ka + 2, Un);
    load_plane(ka + 3 < kb + 4 ? ka + 3 : kb + 4, Un2);  // two planes in flight: this kernel has little else to hide latency
    double Iz_p[8], zhi_p[8], fz_p[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        Iz_p[n] = f3m_interface(at(0, n), at(1, n), at(2, n), at(3, n), at(4, n), at(5, n));
        zhi_p[n] = fz_p[n] = 0.0;
    }
    for (int k = ka - 1; k <= kb + 1; ++k) {
        {
            const int sn = slot_of(k + 3);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                at(sn, n) = Un[n];
                Un[n] = Un2[n];
            }
            const int kn = k + 5 < kb + 4 ? k + 5 : kb + 4;
            load_plane(kn, Un2);
        }
        const int s0 = slot_of(k - 2), s1 = slot_of(k - 1), s2 = slot_of(k), s3 = slot_of(k + 1), s4 = slot_of(k + 2), s5 = slot_of(k + 3);
        double zlo[10], zhi[10];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const double centre = at(s2, n);
            const double Iz = f3m_interface(at(s0, n), at(s1, n), centre, at(s3, n), at(s4, n), at(s5, n));
            zlo[n] = Iz_p[n];
            zhi[n] = Iz;
            Iz_p[n] = Iz;
            trace_faces(speed, hz, centre, zlo[n], zhi[n]);
        }
        zlo[8] = zlo[9] = mlo;
        zhi[8] = zhi[9] = mhi;
        double fz[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (k >= ka) {
            double L[10];
#pragma unroll
            for (int n = 0; n < 8; ++n) L[n] = zhi_p[n];
            L[8] = L[9] = mhi;
            Phys::riemann(P, 2, L, zlo, fz);
        }
        if (k > ka && upd) {
            const long c = col + (long)(k - 1) * sz;
#pragma unroll
            for (int n = 0; n < 8; ++n) dFz[c + n * g.sc] = fz_p[n] - fz[n];
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            zhi_p[n] = zhi[n];
            fz_p[n] = fz[n];
        }
    }
}

// dF_y = F1_y(j) - F1_y(j + 1) for j in [0, n1) on the columns (i, k) in [-1, n0] x [-1, n2]: march along y at fixed
// (x strip, k); lane l <-> i = -1 + 64 strip + l.  (dF_x is not stored: pass 2 recomputes it from its window.)
__global__ __launch_bounds__(64, 2) void k_f3m_y1(Geo g, FieldParams P, const double* __restrict__ U, double* __restrict__ dFy_out, double dt,
                                                   SelF3m sel) {
    typedef FieldF<5> Phys;
    __shared__ double ring_[F3M_NR * 8 * 64 + 2 * F3M_PAD];
    double* const ring = ring_ + F3M_PAD;
    dt = launch_dt(g, dt);
    const int lane = threadIdx.x;
    const int per = (sel.nwaves + 7) >> 3;
    int w = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (w >= sel.nwaves) return;
    const int strip = w % sel.ns;
    w /= sel.ns;
    const int k = w % sel.nline - 1, segi = w / sel.nline;
    const int n0 = g.n[0], n1 = g.n[1];
    const int i = -1 + strip * 64 + lane;
    const int ic = i <= n0 ? i : n0;
    const bool upd = i <= n0;
    const int ja = segi * sel.seg;
    const int jb = (ja + sel.seg - 1 < n1 - 1) ? ja + sel.seg - 1 : n1 - 1;
    const double speed = P.fastest;
    const double hy = uni(0.5 * dt / g.dx[1]);
    const long col = g.at(ic, 0, k);
    const long sy = g.sj;
    const int row0 = ja - 4;
    double mlo, mhi;
    f3m_material_faces(speed, hy, mlo, mhi);
    auto load_row = [&](int row, double* V) {
        const long c = col + (long)row * sy;
#pragma unroll
        for (int n = 0; n < 8; ++n) V[n] = U[c + n * g.sc];
    };
    auto slot_of = [&](int row) { return (row - row0) % F3M_NR; };
    auto at = [&](int slot, int n) -> double& { return ring[(slot * 8 + n) * 64 + lane]; };
    double Un[8], Un2[8];
    for (int s = 0; s < F3M_NR; ++s) {
        load_row(row0 + s, Un);
#pragma unroll
        for (int n = 0; n < 8; ++n) at(s, n) = Un[n];
    }
    load_row(ja + 2, Un);
    load_row(ja + 3 < jb + 4 ? ja + 3 : jb + 4, Un2);
    double Iy_p[8], yhi_p[8], fy_p[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        Iy_p[n] = f3m_interface(at(0, n), at(1, n), at(2, n), at(3, n), at(4, n), at(5, n));
        yhi_p[n] = fy_p[n] = 0.0;
    }
    for (int r = ja - 1; r <= jb + 1; ++r) {
        {
            const int sn = slot_of(r + 3);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                at(sn, n) = Un[n];  // lane-private window: no synchronisation needed
                Un[n] = Un2[n];
            }
            const int rn = r + 5 < jb + 4 ? r + 5 : jb + 4;
            load_row(rn, Un2);
        }
        const int s0 = slot_of(r - 2), s1 = slot_of(r - 1), s2 = slot_of(r), s3 = slot_of(r + 1), s4 = slot_of(r + 2), s5 = slot_of(r + 3);
        double ylo[10], yhi[10];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const double centre = at(s2, n);
            const double Iy = f3m_interface(at(s0, n), at(s1, n), centre, at(s3, n), at(s4, n), at(s5, n));
            ylo[n] = Iy_p[n];
            yhi[n] = Iy;
            Iy_p[n] = Iy;
            trace_faces(speed, hy, centre, ylo[n], yhi[n]);
        }
        ylo[8] = ylo[9] = mlo;
        yhi[8] = yhi[9] = mhi;
        double fy[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (r >= ja) {
            double L[10];
#pragma unroll
            for (int n = 0; n < 8; ++n) L[n] = yhi_p[n];
            L[8] = L[9] = mhi;
            Phys::riemann(P, 1, L, ylo, fy);
        }
        if (r > ja && upd) {
            const long c = col + (long)(r - 1) * sy;
#pragma unroll
            for (int n = 0; n < 8; ++n) dFy_out[c + n * g.sc] = fy_p[n] - fy[n];
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            yhi_p[n] = yhi[n];
            fy_p[n] = fy[n];
        }
    }
}

static int f3m_slots(int per_cu) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess && p.multiProcessorCount > 0) cus = p.multiProcessorCount;
    }
    return cus * per_cu;
}
// segments along the marching direction: the count that minimises (rounds of resident waves) x (iterations per wave)
static SelF3m f3m_select(int ns, int nline, int nmarch, int slots) {
    SelF3m s;
    s.ns = ns;
    s.nline = nline;
    long best = -1;
    int best_n = 1;
    for (int nseg = 1; nseg <= 16 && nseg <= nmarch; ++nseg) {
        const int seg = (nmarch + nseg - 1) / nseg;
        if (seg < 16 && nseg > 1) break;
        const long waves = (long)ns * nline * ((nmarch + seg - 1) / seg);
        const long cost = ((waves + slots - 1) / slots) * (long)(seg + 8);
        if (best < 0 || cost < best) {
            best = cost;
            best_n = nseg;
        }
    }
    s.seg = (nmarch + best_n - 1) / best_n;
    s.nseg = (nmarch + s.seg - 1) / s.seg;
    s.nwaves = ns * nline * s.nseg;
    return s;
}

// Pass 1 of the 3-D field update on the marching kernels: dF_y and dF_z of every cell (dF_x is recomputed by pass 2)
static void f3m_pass1(const Geo& g, const StatePhys& sp, const double* Uin, const Ptr3& dF, double dt, hipStream_t st) {
    const dim3 block(64);
    // (dF_z: left by k_f3m_z, which marches along z anyway; k_f3m_z1 remains as its stand-alone form)
    const SelF3m b = f3m_select((g.n[0] + 2 + 63) / 64, g.n[2] + 2, g.n[1], f3m_slots(6));
    hipLaunchKernelGGL(k_f3m_y1, dim3(((b.nwaves + 7) / 8) * 8), block, 0, st, g, sp.fp, Uin, dF.p[1], dt, b);
}

// The 3-D field update on the marching kernels, both passes.  dF: scratch for the cell flux differences (y and z;
// 8 planes each), DU: scratch (8 planes) for G_z.  Returns 0 where it does not apply (the caller then runs the
// plane-tile kernels).
int ctu_field3d_march(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, const Ptr3& dF, double* DU, double dt,
                      const int touches[3][2], const SourceEpi* epi, hipStream_t st) {
    static const int off = std::getenv("CMFP_FIELD_MARCH") ? std::atoi(std::getenv("CMFP_FIELD_MARCH")) == 0 : 0;
    if (off || g.dim != 3 || sp.kind != 1 || sp.recon != 5 || sp.flux != 10 || !sp.vacuum) return 0;
    if (g.ng[0] < 4 || g.ng[1] < 4 || g.ng[2] < 4) return 0;
    if (epi && (epi->a.implicit || epi->a.coll.on || epi->ns < 1 || epi->ns > 2)) return 0;
    f3m_pass1(g, sp, Uin, dF, dt, st);
    FaceBC3m bc;
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
    const dim3 block(64);
    const SelF3m a = f3m_select((g.n[0] + 55) / 56, g.n[1] + 2, g.n[2], f3m_slots(4));
    const dim3 ga(((a.nwaves + 7) / 8) * 8);
    if (has_bc)
        hipLaunchKernelGGL((k_f3m_z<true>), ga, block, 0, st, g, sp.fp, Uin, dF, DU, dt, bc, a);
    else
        hipLaunchKernelGGL((k_f3m_z<false>), ga, block, 0, st, g, sp.fp, Uin, dF, DU, dt, bc, a);
    const SelF3m b = f3m_select((g.n[0] + 55) / 56, g.n[2], g.n[1], f3m_slots(4));
    const dim3 gb(((b.nwaves + 7) / 8) * 8);
#define F3M(BC, NS, E) hipLaunchKernelGGL((k_f3m_yx<BC, NS>), gb, block, 0, st, g, sp.fp, Uin, Uout, dF, DU, dt, bc, b, E)
    if (epi && epi->ns == 2) {
        if (has_bc) F3M(true, 2, *epi); else F3M(false, 2, *epi);
    } else if (epi && epi->ns == 1) {
        if (has_bc) F3M(true, 1, *epi); else F3M(false, 1, *epi);
    } else {
        if (has_bc) F3M(true, 0, NoEpi{}); else F3M(false, 0, NoEpi{});
    }
#undef F3M
    return 1;
}

}  // namespace CMFP_NS
