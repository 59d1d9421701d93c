// mfp_march3d.hip — z-marching fused corner-transport-upwind update of a hydro state (3-D).
//
// HydroCTU::calc_spatial_derivative (MFP_CTU_hydro.cpp:142-531) + `new += dU` (MFP_advance.cpp:220-230)
// in one launch for a 3-D block.  A workgroup of 16 x 16 threads owns a 14 x 14 column of cells (one
// thread per cell of the tile grown by one, as in mfp_fused.hip) and marches along z, one plane per
// iteration; what couples neighbouring planes is carried in registers by the thread that owns the
// (i, j) column, what couples neighbouring columns goes through LDS (y) and DPP row shifts (x):
//
//   iteration m:  plane m+1 -> primitives of the tile grown by two, into LDS (double buffered)
//                 reconstruction + half-step trace of plane m in x, y (LDS stencils) and z (registers)
//                 Riemann sweep 1: F_x(m), F_y(m), F_z(m)            [F_z(m) sits between planes m-1, m]
//                 transverse corrections (EulerianState::correct_face_prim MFP_eulerian.cpp:1047-1188: for
//                 face direction d1 the directions (d1+1)%3 then (d1+2)%3, each face state with the sweep-1
//                 fluxes of its own cell): z-faces of plane m with dF_x, dF_y now; x-faces of plane m with
//                 dF_y now and dF_z one iteration later; y-faces with dF_z then dF_x one iteration later
//                 domain-face BCs, Riemann sweep 2: F'_z(m), F'_x(m-1), F'_y(m-1)
//                 divergence of plane m-1, Uout = Uin + dU, optional CFL reduction
//
// Against the gather kernels (mfp_gather.hip), which recompute the reconstruction of both cells of every
// face in both sweeps and move two sets of face fluxes through HBM, every cell is reconstructed once
// and only U is read and written.  Same device functions: STRICT results are bit-identical.
#include <cstring>

#include "mfp_march.h"

namespace CMFP_NS {

// value of the lane one to the left / right inside a row of 16 lanes (the x-neighbour thread): DPP row
// shifts; every lane of the wave must execute these
CMFP_DEV double row_left(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xf, 0xf, true);  // row_shr:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
CMFP_DEV double row_right(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x101, 0xf, 0xf, true);  // row_shl:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x101, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

struct FaceBC3m {
    int touch[3][2];
    int rule[3][2][5];
};
struct Sel3 {
    int gx, gy, nseg, seg, n;  // tiles in x and y, z-segments, planes per segment, workgroups
};

// the states one launch updates: one, or two species of the same physics (mfp_march.hip: MarchStates)
template <class Phys>
struct March3States {
    int n;
    typename Phys::Params P[2];
    const double* Uin[2];
    double* Uout[2];
    unsigned long long* red[2];
};

template <class Phys, bool HAS_BC, bool REDUCE>
__global__ __launch_bounds__(256, 2) void k_ctu3d(Geo g, March3States<Phys> S, double dt, FaceBC3m bc, int* err, Sel3 sel) {
    constexpr int RECON = Phys::RECON;
    static_assert(ReconTraits<RECON>::SL == 3 && Phys::IS_HYDRO, "three-point hydro reconstructions");
    constexpr int BX = 16, BY = 16, TX = BX - 2, TY = BY - 2, PX = BX + 2, PY = BY + 2, C = BX * BY, NP = PX * PY;
    __shared__ double Pq[2][5][NP];  // primitives of two planes (tile grown by two), by plane parity
    __shared__ double bufR[5][C];    // high-side y-face states of every cell
    __shared__ double Fb[5][C];      // y-face fluxes of every cell
    __shared__ double stash[15][C];  // per thread, plane m - 1: traced y-face states (lo, hi) and dF_x, parked here
                                     // between iterations (in registers they push the kernel into scratch)
    dt = launch_dt(g, dt);           // device-resident time step (cmfp_advance_steps)

    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * BX + tx;
    // XCD-aware order (see mfp_fused.hip): a contiguous band of workgroups per XCD
    const int total = sel.n * S.n;
    const int per = (total + 7) >> 3;
    int wg = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (wg >= total) return;
    const int which = wg >= sel.n ? 1 : 0;  // workgroup-uniform: the state this workgroup updates
    wg -= which * sel.n;
    const typename Phys::Params P = S.P[which];
    const double* __restrict__ Uin = S.Uin[which];
    double* __restrict__ Uout = S.Uout[which];
    unsigned long long* red = S.red[which];
    const int tiles = sel.gx * sel.gy;
    const int segi = wg / tiles, tile = wg - segi * tiles;
    const int tiy = tile / sel.gx, tix = tile - tiy * sel.gx;
    const int i0 = tix * TX, j0 = tiy * TY;
    const int i = i0 - 1 + tx, j = j0 - 1 + ty;  // this thread's column
    const int n0 = g.n[0], n1 = g.n[1], n2 = g.n[2];
    const int ka = segi * sel.seg;
    const int kb = (ka + sel.seg - 1 < n2 - 1) ? ka + sel.seg - 1 : n2 - 1;
    const int pc = (ty + 1) * PX + (tx + 1);
    const double gamma = Phys::gamma_of(P), gm1 = gamma - 1.0;
    const double hx = uni(0.5 * dt / g.dx[0]), hy = uni(0.5 * dt / g.dx[1]), hz = uni(0.5 * dt / g.dx[2]);
    const double dtdx = uni(dt / g.dx[0]), dtdy = uni(dt / g.dx[1]), dtdz = uni(dt / g.dx[2]);
    const bool xin = tx >= 1 && tx <= BX - 2, yin = ty >= 1 && ty <= BY - 2;  // the thread's fluxes on both sides exist
    const bool own = xin && yin && i < n0 && j < n1;                          // cells this thread updates
    const bool ci_in = i >= 0 && i < n0, cj_in = j >= 0 && j < n1;
    bool bad = false;

    // conserved -> {rho, u, v, w, eps = p / (gamma - 1)} of plane kp for the tile grown by two
    auto fill_plane = [&](int kp) {
        double(*dst)[NP] = Pq[kp & 1];
        for (int t = tid; t < NP; t += C) {
            const int py = t / PX, px = t - py * PX;
            const int ci = i0 - 2 + px, cj = j0 - 2 + py;
            if (ci <= n0 + 1 && cj <= n1 + 1) {  // ci, cj >= -2 always
                const long c = g.at(ci, cj, kp);
                double U[5], Q[5];
#pragma unroll
                for (int n = 0; n < 5; ++n) U[n] = Uin[c + n * g.sc];
                if (!Phys::cons2prim(P, U, Q)) bad = true;
                Q[H_P] = f_div(Q[H_P], gm1);  // MFP_hydro.cpp:847-862
#pragma unroll
                for (int n = 0; n < 5; ++n) dst[n][t] = Q[n];
            }
        }
    };

    double Qa[5], Qb[5];
    fill_plane(ka - 2);
    fill_plane(ka - 1);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        Qa[n] = Pq[(ka - 2) & 1][n][pc];
        Qb[n] = Pq[(ka - 1) & 1][n][pc];
    }
    // carried from the previous iteration (plane m - 1)
    double zhiU_p[5], zhiC_p[5];  // high z-face state: traced (sweep 1) and corrected (sweep 2)
    double xlo_p[5], xhi_p[5];    // x-face states, corrected with dF_y, waiting for dF_z
    // (stash: y-face states, traced, waiting for dF_z then dF_x; F_x(i) - F_x(i+1) of plane m - 1)
    double fz_p[5], fz2_p[5];     // F_z and F'_z on the low z-face of plane m - 1
#pragma unroll
    for (int n = 0; n < 5; ++n)
        zhiU_p[n] = zhiC_p[n] = xlo_p[n] = xhi_p[n] = fz_p[n] = fz2_p[n] = 0.0;
#pragma unroll
    for (int n = 0; n < 15; ++n) stash[n][tid] = 0.0;
    double red_speed = 2.2250738585072014e-308;  // std::numeric_limits<Real>::min(), MFP_eulerian.cpp:265

    // Plane m + 1 enters during iteration m without a barrier of its own: every thread converts the cell of its own
    // column (kept in registers as Qc: the z stencil) and the first 68 threads one cell each of the tile's outer ring
    // (18 x 18 - 16 x 16); the loads are issued at the top of the iteration and consumed after the in-plane
    // reconstruction, which hides their latency.  The tile of plane m + 1 replaces that of plane m - 1, last read
    // before the first barrier of iteration m - 1, and is first read (as the in-plane stencil) in iteration m + 1,
    // behind the barriers of iteration m.
    int rpx = 0, rpy = 0;  // this thread's ring cell (tid < 68)
    if (tid < PX) rpx = tid, rpy = 0;
    else if (tid < 2 * PX) rpx = tid - PX, rpy = PY - 1;
    else rpx = ((tid - 2 * PX) & 1) ? PX - 1 : 0, rpy = 1 + ((tid - 2 * PX) >> 1);
    const bool ring = tid < NP - C && i0 - 2 + rpx <= n0 + 1 && j0 - 2 + rpy <= n1 + 1;
    const bool own_in = i <= n0 + 1 && j <= n1 + 1;  // i, j >= -1 always
    // byte offsets inside the first (ghost) plane of a component (32 bits: cmfp_create keeps a component array below 4 GiB);
    // the component and plane offsets are wave-uniform and stay in scalar registers
    const unsigned o_own = (unsigned)(g.at(own_in ? i : 0, own_in ? j : 0, -g.ng[2]) * 8);
    const unsigned o_ring = (unsigned)(g.at(ring ? i0 - 2 + rpx : 0, ring ? j0 - 2 + rpy : 0, -g.ng[2]) * 8);
    const int t_ring = rpy * PX + rpx;
    auto ld = [](const double* base, unsigned off) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + off); };
    auto st_ = [](double* base, unsigned off, double v) { *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + off) = v; };

    for (int m = ka - 1; m <= kb + 1; ++m) {
        double Uo[5], Ur[5];
        {
            const double* pl = Uin + (long)(m + 1 + g.ng[2]) * g.sk;
#pragma unroll
            for (int n = 0; n < 5; ++n) Uo[n] = ld(pl + n * g.sc, o_own);
            if (tid < NP - C) {
#pragma unroll
                for (int n = 0; n < 5; ++n) Ur[n] = ld(pl + n * g.sc, o_ring);
            }
        }
        const double(*Pm)[NP] = Pq[m & 1];

        // ---- reconstruction + half-step trace of plane m in x, y, z
        double xlo[5], xhi[5], ylo[5], yhi[5], zlo[5], zhi[5];
        double Qc[5] = {1.0, 0.0, 0.0, 0.0, 1.0};  // (threads outside the loaded tile: a valid state, never used for a stored value)
        {
            double centre[5], speed[3], st[3][5];
#pragma unroll
            for (int n = 0; n < 5; ++n) centre[n] = Qb[n];
            centre[H_P] = centre[H_P] * gm1;  // restored pressure, MFP_hydro.cpp:1025-1035
            Phys::speed_from_prim(P, centre, speed);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                st[0][n] = Pm[n][pc - 1];
                st[1][n] = Qb[n];
                st[2][n] = Pm[n][pc + 1];
            }
            hydro_cell_faces<RECON>(gamma, st, xlo, xhi);
#pragma unroll
            for (int n = 0; n < 5; ++n) trace_faces(speed[0], hx, centre[n], xlo[n], xhi[n]);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                st[0][n] = Pm[n][pc - PX];
                st[2][n] = Pm[n][pc + PX];
            }
            hydro_cell_faces<RECON>(gamma, st, ylo, yhi);
#pragma unroll
            for (int n = 0; n < 5; ++n) trace_faces(speed[1], hy, centre[n], ylo[n], yhi[n]);
            // ---- plane m + 1: conserved -> {rho, u, v, w, eps} of the own column and of one ring cell
            {
                double(*dst)[NP] = Pq[(m + 1) & 1];
                if (own_in) {
                    if (!Phys::cons2prim(P, Uo, Qc)) bad = true;
                    Qc[H_P] = f_div(Qc[H_P], gm1);  // MFP_hydro.cpp:847-862
#pragma unroll
                    for (int n = 0; n < 5; ++n) dst[n][pc] = Qc[n];
                }
                if (ring) {
                    double Q[5];
                    if (!Phys::cons2prim(P, Ur, Q)) bad = true;
                    Q[H_P] = f_div(Q[H_P], gm1);
#pragma unroll
                    for (int n = 0; n < 5; ++n) dst[n][t_ring] = Q[n];
                }
            }
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                st[0][n] = Qa[n];
                st[2][n] = Qc[n];
            }
            hydro_cell_faces<RECON>(gamma, st, zlo, zhi);
#pragma unroll
            for (int n = 0; n < 5; ++n) trace_faces(speed[2], hz, centre[n], zlo[n], zhi[n]);
        }
#pragma unroll
        for (int n = 0; n < 5; ++n) bufR[n][tid] = yhi[n];
        __syncthreads();

        // ---- Riemann sweep 1 on the low faces of (i, j, m)
        double fx[5], fy[5], fz[5] = {0, 0, 0, 0, 0};
        {
            double L[5];
#pragma unroll
            for (int n = 0; n < 5; ++n) L[n] = row_left(xhi[n]);
            Phys::riemann(P, 0, L, xlo, fx);
#pragma unroll
            for (int n = 0; n < 5; ++n) L[n] = bufR[n][ty >= 1 ? tid - BX : tid];
            Phys::riemann(P, 1, L, ylo, fy);
            if (m >= ka) Phys::riemann(P, 2, zhiU_p, zlo, fz);
        }
        double dFx[5], dFy[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            Fb[n][tid] = fy[n];
            dFx[n] = fx[n] - row_right(fx[n]);
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 5; ++n) dFy[n] = fy[n] - Fb[n][ty <= BY - 2 ? tid + BX : tid];

        // ---- transverse corrections of plane m that need in-plane fluxes only
        double zhiU[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) zhiU[n] = zhi[n];
        {
            // z-faces: x then y
            bool ok_hi = Phys::correct(P, zhi, hx, dFx);
            bool ok_lo = Phys::correct(P, zlo, hx, dFx);
            ok_hi = Phys::correct(P, zhi, hy, dFy) && ok_hi;
            ok_lo = Phys::correct(P, zlo, hy, dFy) && ok_lo;
            if (xin && yin && ci_in && cj_in && ((!ok_hi && m < n2) || (!ok_lo && m >= 0 && m <= n2))) bad = true;
            // x-faces: y now, z in the next iteration
            const bool oxh = Phys::correct(P, xhi, hy, dFy);
            const bool oxl = Phys::correct(P, xlo, hy, dFy);
            if (yin && cj_in && m >= 0 && m < n2 && ((!oxh && i >= -1 && i < n0) || (!oxl && i >= 0 && i <= n0))) bad = true;
        }
        // ---- plane m - 1: the corrections that waited for F_z(m)
        double ylo_p[5], yhi_p[5];
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            ylo_p[n] = stash[n][tid];
            yhi_p[n] = stash[5 + n][tid];
        }
        if (m > ka) {
            double dFz[5], dFx_p[5];
#pragma unroll
            for (int n = 0; n < 5; ++n) dFx_p[n] = stash[10 + n][tid];
#pragma unroll
            for (int n = 0; n < 5; ++n) dFz[n] = fz_p[n] - fz[n];
            const bool oxh = Phys::correct(P, xhi_p, hz, dFz);
            const bool oxl = Phys::correct(P, xlo_p, hz, dFz);
            if (yin && cj_in && ((!oxh && i >= -1 && i < n0) || (!oxl && i >= 0 && i <= n0))) bad = true;
            bool oyh = Phys::correct(P, yhi_p, hz, dFz);
            bool oyl = Phys::correct(P, ylo_p, hz, dFz);
            oyh = Phys::correct(P, yhi_p, hx, dFx_p) && oyh;
            oyl = Phys::correct(P, ylo_p, hx, dFx_p) && oyl;
            if (xin && ci_in && ((!oyh && j >= -1 && j < n1) || (!oyl && j >= 0 && j <= n1))) bad = true;
        }
#pragma unroll
        for (int n = 0; n < 5; ++n) bufR[n][tid] = yhi_p[n];  // read in sweep 1 before the last barrier
        __syncthreads();

        // ---- domain-face BCs (update_face_prim, do_all) + Riemann sweep 2
        double fz2[5] = {0, 0, 0, 0, 0};
        if (m >= ka) {
            double L[5], R[5];
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                L[n] = zhiC_p[n];
                R[n] = zlo[n];
            }
            if (HAS_BC && bc.touch[2][0] && m == 0) {
#pragma unroll
                for (int n = 0; n < 5; ++n)
                    if (bc.rule[2][0][n]) L[n] = bc.rule[2][0][n] == FILL_ODD ? -R[n] : R[n];
            }
            if (HAS_BC && bc.touch[2][1] && m == n2) {
#pragma unroll
                for (int n = 0; n < 5; ++n)
                    if (bc.rule[2][1][n]) R[n] = bc.rule[2][1][n] == FILL_ODD ? -L[n] : L[n];
            }
            Phys::riemann(P, 2, L, R, fz2);
        }
        if (m > ka) {
            double L[5], R[5], fx2[5], fy2[5];
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                L[n] = row_left(xhi_p[n]);
                R[n] = xlo_p[n];
            }
            if (HAS_BC && bc.touch[0][0] && i == 0) {
#pragma unroll
                for (int n = 0; n < 5; ++n)
                    if (bc.rule[0][0][n]) L[n] = bc.rule[0][0][n] == FILL_ODD ? -R[n] : R[n];
            }
            if (HAS_BC && bc.touch[0][1] && i == n0) {
#pragma unroll
                for (int n = 0; n < 5; ++n)
                    if (bc.rule[0][1][n]) R[n] = bc.rule[0][1][n] == FILL_ODD ? -L[n] : L[n];
            }
            Phys::riemann(P, 0, L, R, fx2);
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                L[n] = bufR[n][ty >= 1 ? tid - BX : tid];
                R[n] = ylo_p[n];
            }
            if (HAS_BC && bc.touch[1][0] && j == 0) {
#pragma unroll
                for (int n = 0; n < 5; ++n)
                    if (bc.rule[1][0][n]) L[n] = bc.rule[1][0][n] == FILL_ODD ? -R[n] : R[n];
            }
            if (HAS_BC && bc.touch[1][1] && j == n1) {
#pragma unroll
                for (int n = 0; n < 5; ++n)
                    if (bc.rule[1][1][n]) R[n] = bc.rule[1][1][n] == FILL_ODD ? -L[n] : L[n];
            }
            Phys::riemann(P, 1, L, R, fy2);
            // what the update of plane m - 1 starts from: requested before the barrier, which hides most of the latency
            double Uold[5] = {0, 0, 0, 0, 0};
            if (own) {
                const double* pl = Uin + (long)(m - 1 + g.ng[2]) * g.sk;
#pragma unroll
                for (int n = 0; n < 5; ++n) Uold[n] = ld(pl + n * g.sc, o_own);
            }
            double fx2h[5];
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                Fb[n][tid] = fy2[n];
                fx2h[n] = row_right(fx2[n]);
            }
            __syncthreads();
            // ---- divergence of plane m - 1 (calc_divergence MFP_eulerian.cpp:1614-1662), new += dU
            if (own) {
                double* po = Uout + (long)(m - 1 + g.ng[2]) * g.sk;
                double Unew[5];
#pragma unroll
                for (int n = 0; n < 5; ++n) {
                    double du = 0.0;
                    du += dtdx * (fx2[n] - fx2h[n]);
                    du += dtdy * (fy2[n] - Fb[n][tid + BX]);
                    du += dtdz * (fz2_p[n] - fz2[n]);
                    Unew[n] = Uold[n] + du;
                    st_(po + n * g.sc, o_own, Unew[n]);
                }
                if constexpr (REDUCE) {  // EulerianState::get_allowed_time_step MFP_eulerian.cpp:290-296
                    double s[3];
                    hydro_speed_from_cons(P.gas, Unew, s);
                    red_speed = std_max(std_max(std_max(red_speed, s[0]), s[1]), s[2]);
                }
            }
        } else {
            __syncthreads();  // keep the barrier count of the iteration uniform
        }
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            Qa[n] = Qb[n];
            Qb[n] = Qc[n];
            zhiU_p[n] = zhiU[n];
            zhiC_p[n] = zhi[n];
            xlo_p[n] = xlo[n];
            xhi_p[n] = xhi[n];
            stash[n][tid] = ylo[n];
            stash[5 + n][tid] = yhi[n];
            stash[10 + n][tid] = dFx[n];
            fz_p[n] = fz[n];
            fz2_p[n] = fz2[n];
        }
    }
    if (bad) *err = 1;
    if constexpr (REDUCE) {
        __syncthreads();
        double* const shr = &bufR[0][0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) red_speed = fmax(red_speed, __shfl_xor(red_speed, o, 64));
        if ((tid & 63) == 0) shr[tid >> 6] = red_speed;
        __syncthreads();
        if (tid == 0) {
            for (int w2 = 1; w2 < C / 64; ++w2) red_speed = fmax(red_speed, shr[w2]);
            atomicMax(&red[wg & 255], (unsigned long long)__double_as_longlong(red_speed));
        }
    }
}

template <class Phys>
static int launch3d(const Geo& g, int ns, const StatePhys* const sps[2], const double* const Uins[2], double* const Uouts[2], double dt,
                    const int touches[3][2], int* err, unsigned long long* const reds[2], hipStream_t st) {
    if (g.ng[0] < 2 || g.ng[1] < 2 || g.ng[2] < 2) return 0;
    if (!Uins[0]) return 1;  // coverage probe
    const StatePhys& sp = *sps[0];
    March3States<Phys> S;
    std::memset(&S, 0, sizeof(S));
    S.n = ns;
    for (int k = 0; k < ns; ++k) {
        S.P[k] = sps[k]->hp;
        S.Uin[k] = Uins[k];
        S.Uout[k] = Uouts[k];
        S.red[k] = reds[k];
    }
    Sel3 sel;
    sel.gx = (g.n[0] + 13) / 14;
    sel.gy = (g.n[1] + 13) / 14;
    // z-segments: enough workgroups to fill the device several times over (measured at 256^3: 16- to
    // 32-plane segments are best, 4.8 ms for two states; one segment per tile, 361 workgroups: 6.6 ms),
    // at least 12 planes each (96^3: 12-plane segments beat 16 by 8 %; 128^3: no difference)
    const int tiles = sel.gx * sel.gy;
    // wave_slots() / 4 workgroups are resident at 2 waves per SIMD: eight rounds per launch (two states in one
    // launch: half as many, longer segments - 256^3: 43 planes against 22, both species 4.20 -> 4.08 ms)
    int nseg = (2 * wave_slots() / ns + tiles - 1) / tiles;
    if (nseg < 1) nseg = 1;
    int seg = (g.n[2] + nseg - 1) / nseg;
    if (seg < 12) seg = 12;
    static const int seg_env = std::getenv("CMFP_MARCH3D_SEG") ? std::atoi(std::getenv("CMFP_MARCH3D_SEG")) : 0;
    if (seg_env > 0) seg = seg_env;
    if (seg > g.n[2]) seg = g.n[2];
    sel.seg = seg;
    sel.nseg = (g.n[2] + seg - 1) / seg;
    sel.n = tiles * sel.nseg;
    FaceBC3m bc;
    bool has_bc = g.bc_instance != 0;
    for (int d = 0; d < 3; ++d)
        for (int s = 0; s < 2; ++s) {
            bc.touch[d][s] = touches[d][s];
            has_bc = has_bc || touches[d][s];
            for (int n = 0; n < 5; ++n) {
                const int code = sp.bc[n][d][s];  // MFP_eulerian.cpp:796-835
                bc.rule[d][s][n] = (code == 2 || code == 4 || code == 1) ? FILL_EVEN : (code == -1 ? FILL_ODD : FILL_SKIP);
            }
        }
    const dim3 grid(((sel.n * ns + 7) / 8) * 8), block(16, 16);
#define CMFP_LAUNCH(BC, RED) hipLaunchKernelGGL((k_ctu3d<Phys, BC, RED>), grid, block, 0, st, g, S, dt, bc, err, sel)
    if (reds[0]) {
        if (has_bc) CMFP_LAUNCH(true, true); else CMFP_LAUNCH(false, true);
    } else {
        if (has_bc) CMFP_LAUNCH(true, false); else CMFP_LAUNCH(false, false);
    }
#undef CMFP_LAUNCH
    return 1;
}

// Fused update of a 3-D hydro state; returns 0 where the option set is outside this kernel (the caller
// then uses the gather kernels).  Uin == nullptr probes coverage.
int ctu_march3d(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, int do_ctu, double dt,
                const int touches[3][2], int* err, unsigned long long* red, hipStream_t st) {
    if (g.dim != 3 || sp.kind != 0 || !do_ctu || sp.hp.nt) return 0;
    const StatePhys* const sps[2] = {&sp, nullptr};
    const double* const in[2] = {Uin, nullptr};
    double* const out[2] = {Uout, nullptr};
    unsigned long long* const reds[2] = {red, nullptr};
#define HY(R, F) \
    if (sp.recon == R && sp.flux == F) return launch3d<HydroF<R, F>>(g, 1, sps, in, out, dt, touches, err, reds, st);
    HY(1, 0) HY(1, 1) HY(2, 0) HY(2, 1) HY(3, 0) HY(4, 0)
#undef HY
    return 0;
}

// two 3-D hydro states of the same physics in one launch (as ctu_march2 in 2-D); 0: nothing launched
int ctu_march3d2(const Geo& g, const StatePhys* const sp[2], const double* const Uin[2], double* const Uout[2], int do_ctu, double dt,
                 const int touches[3][2], int* err, unsigned long long* const red[2], hipStream_t st) {
    const StatePhys &a = *sp[0], &b = *sp[1];
    if (g.dim != 3 || a.kind != 0 || b.kind != 0 || !do_ctu || a.hp.nt || b.hp.nt) return 0;
    if (a.recon != b.recon || a.flux != b.flux || (red[0] == nullptr) != (red[1] == nullptr)) return 0;
    if (std::memcmp(a.bc, b.bc, sizeof(a.bc)) != 0) return 0;
    static const bool off = std::getenv("CMFP_MARCH_PAIR") && std::atoi(std::getenv("CMFP_MARCH_PAIR")) == 0;
    if (off) return 0;
#define HY(R, F) \
    if (a.recon == R && a.flux == F) return launch3d<HydroF<R, F>>(g, 2, sp, Uin, Uout, dt, touches, err, red, st);
    HY(1, 0) HY(1, 1) HY(2, 0) HY(2, 1) HY(3, 0) HY(4, 0)
#undef HY
    return 0;
}

}  // namespace CMFP_NS
