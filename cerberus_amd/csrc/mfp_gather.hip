// mfp_gather.hip — dimension-generic fused CTU update without tiles (1-D, 3-D; 2-D for tests).
//
// The tiled kernel of mfp_fused.hip keeps a 2-D tile and its ring in LDS.  In 3-D a ring of one
// cell around a tile that fits LDS leaves 42 % useful threads (hydro) and the O6 field tile does
// not fit at all, so this path trades LDS for recomputation: every thread gathers the stencil it
// needs straight from the conserved array (served by L1/L2) and the only intermediates that touch
// HBM are the two sets of face fluxes.  HydroCTU::calc_spatial_derivative
// (MFP_CTU_hydro.cpp:142-531) + `new += dU` become three kernels per state:
//
//   k_g_flux<.., false>  U -> prims -> faces -> trace -> Riemann sweep 1          -> F1[d]
//   k_g_flux<.., true>   the same face states, transverse correction with F1, domain-face BCs,
//                        Riemann sweep 2                                             -> F2[d]
//   k_g_update           divergence of F2 (or F1 without corner transport), Uout = Uin + dU,
//                        optional CFL reduction
//
// against eight full-array passes over prim / r_lo / r_hi / flux / dU arrays in the staged form.
// Per value the arithmetic is the same sequence of device functions as the staged and tiled
// kernels, so STRICT results are identical to them bit for bit.
#include "mfp_device.h"

namespace CMFP_NS {
using namespace cmfp;

struct GBox {
    int lo[3];   // first cell index per dimension
    int ext[3];  // cells per dimension
    unsigned bpp;  // workgroups per k-plane
    unsigned nblk;
};

struct FaceBC3 {
    int touch[3][2];
    int rule[3][2][10];
};

static constexpr int GT = 256;

// XCD-aware workgroup order (see mfp_fused.hip): contiguous bands of rows / planes per XCD
__device__ __forceinline__ bool g_cell(const GBox& b, int& i, int& j, int& k) {
    const unsigned per = (b.nblk + 7u) >> 3;
    const unsigned blk = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
    if (blk >= b.nblk) return false;
    const unsigned plane = blk / b.bpp;
    const unsigned t = (blk - plane * b.bpp) * GT + threadIdx.x;
    if (t >= (unsigned)(b.ext[0] * b.ext[1])) return false;
    const unsigned jj = t / (unsigned)b.ext[0];
    i = b.lo[0] + (int)(t - jj * (unsigned)b.ext[0]);
    j = b.lo[1] + (int)jj;
    k = b.lo[2] + (int)plane;
    return true;
}

// Traced (half-step) face states on the two sides of the low d-face of cell c:
//   L = high-side face state of cell c - e_d,  R = low-side face state of cell c
// (calc_reconstruction + calc_time_averaged_faces, MFP_hydro.cpp:780-1038 / MFP_eulerian.cpp:513-676).
template <class Phys>
CMFP_DEV void face_pair(const Geo& g, const typename Phys::Params& P, const double* __restrict__ U, long c, long sd, int d,
                        int do_ctu, double dt, double* L, double* R, int* err) {
    constexpr int RECON = Phys::RECON, NQ = Phys::NQ;
    constexpr int SL = ReconTraits<RECON>::SL, H = SL / 2;
    const double dt_2dx = 0.5 * dt / g.dx[d];
    if constexpr (Phys::IS_HYDRO) {
        const double gm1 = Phys::gamma_of(P) - 1.0;
        double pr[SL + 1][5];  // cells c - (H+1) .. c + H along d, with eps = p / (gamma - 1) in slot H_P
#pragma unroll
        for (int s = 0; s < SL + 1; ++s) {
            double u[5];
#pragma unroll
            for (int n = 0; n < 5; ++n) u[n] = U[c + (long)(s - H - 1) * sd + n * g.sc];
            if (!Phys::cons2prim(P, u, pr[s])) *err = 1;
            pr[s][H_P] = f_div(pr[s][H_P], gm1);
        }
        double lo[5], hi[5];
        hydro_cell_faces<RECON>(Phys::gamma_of(P), &pr[0], lo, hi);  // cell c - e_d
        if (do_ctu) {
            double centre[5], speed[3];
#pragma unroll
            for (int n = 0; n < 5; ++n) centre[n] = pr[H][n];
            centre[H_P] = centre[H_P] * gm1;
            Phys::speed_from_prim(P, centre, speed);
#pragma unroll
            for (int n = 0; n < 5; ++n) trace_faces(speed[d], dt_2dx, centre[n], lo[n], hi[n]);
        }
#pragma unroll
        for (int n = 0; n < 5; ++n) L[n] = hi[n];
        hydro_cell_faces<RECON>(Phys::gamma_of(P), &pr[1], lo, hi);  // cell c
        if (do_ctu) {
            double centre[5], speed[3];
#pragma unroll
            for (int n = 0; n < 5; ++n) centre[n] = pr[H + 1][n];
            centre[H_P] = centre[H_P] * gm1;
            Phys::speed_from_prim(P, centre, speed);
#pragma unroll
            for (int n = 0; n < 5; ++n) trace_faces(speed[d], dt_2dx, centre[n], lo[n], hi[n]);
        }
#pragma unroll
        for (int n = 0; n < 5; ++n) R[n] = lo[n];
    } else {
        double speed[3];
        Phys::speed_from_prim(P, nullptr, speed);
#pragma unroll
        for (int n = 0; n < NQ; ++n) {
            double v[SL + 1];
#pragma unroll
            for (int s = 0; s < SL + 1; ++s) v[s] = U[c + (long)(s - H - 1) * sd + n * g.sc];
#ifdef CMFP_FASTMATH
            if (n >= F_MU) {  // flat material coefficients: see mfp_fused.hip phase 1
                bool same = true;
#pragma unroll
                for (int s = 0; s < SL + 1; ++s) same = same && (v[s] == v[H]);
                if (same) {
                    L[n] = R[n] = v[H];
                    continue;
                }
            }
#endif
            double lo, hi;
            recon_faces<RECON>(&v[0], lo, hi);
            if (do_ctu) trace_faces(speed[d], dt_2dx, v[H], lo, hi);
            L[n] = hi;
            recon_faces<RECON>(&v[1], lo, hi);
            if (do_ctu) trace_faces(speed[d], dt_2dx, v[H + 1], lo, hi);
            R[n] = lo;
        }
    }
}

// one direction D of k_g_flux for the cell (idx, c)
template <class Phys, int DIM, bool SWEEP2, int D>
CMFP_DEV void g_flux_dir(const Geo& g, const typename Phys::Params& P, const double* __restrict__ U, const Ptr3& F1,
                         const Ptr3& F2, int do_ctu, double dt, const FaceBC3& bc, int* err, const int* idx, long c) {
    constexpr int NQ = Phys::NQ, NC = Phys::NC, NF = Phys::NF;
    const long str[3] = {1, g.sj, g.sk};
    bool on = idx[D] >= 0;
    if (SWEEP2) {
#pragma unroll
        for (int e = 0; e < DIM; ++e)
            if (e != D) on = on && idx[e] < g.n[e];
    }
    if (!on) return;
    double L[NQ], R[NQ], F[NC];
    face_pair<Phys>(g, P, U, c, str[D], D, do_ctu, dt, L, R, err);
    if (SWEEP2) {
        // EulerianState::correct_face_prim MFP_eulerian.cpp:1047-1188: the (DIM-1) transverse
        // corrections one after the other, each face state with the fluxes of its own cell
        const long cm = c - str[D];
#pragma unroll
        for (int dd = 1; dd < DIM; ++dd) {
            const int d2 = (D + dd) % DIM;
            const double* __restrict__ f = F1.p[d2];
            const double kk = 0.5 * dt / g.dx[d2];
            double dF[NF];
#pragma unroll
            for (int n = 0; n < NF; ++n) dF[n] = f[cm + n * g.sc] - f[cm + str[d2] + n * g.sc];
            if (!Phys::correct(P, L, kk, dF)) *err = 1;
#pragma unroll
            for (int n = 0; n < NF; ++n) dF[n] = f[c + n * g.sc] - f[c + str[d2] + n * g.sc];
            if (!Phys::correct(P, R, kk, dF)) *err = 1;
        }
        // update_face_prim, do_all (MFP_eulerian.cpp:882-955, 794-836)
        if (bc.touch[D][0] && idx[D] == 0) {
#pragma unroll
            for (int n = 0; n < NQ; ++n)
                if (bc.rule[D][0][n]) L[n] = bc.rule[D][0][n] == FILL_ODD ? -R[n] : R[n];
        }
        if (bc.touch[D][1] && idx[D] == g.n[D]) {
#pragma unroll
            for (int n = 0; n < NQ; ++n)
                if (bc.rule[D][1][n]) R[n] = bc.rule[D][1][n] == FILL_ODD ? -L[n] : L[n];
        }
    }
    Phys::riemann(P, D, L, R, F);
    double* __restrict__ out = SWEEP2 ? F2.p[D] : F1.p[D];
#pragma unroll
    for (int n = 0; n < NF; ++n) out[c + n * g.sc] = F[n];
}

// SWEEP2 = false: F1[d] on cells with index_d in [0, n_d], the other indices in [-1, n]
// SWEEP2 = true : F2[d] on cells with index_d in [0, n_d], the other indices in [0, n - 1]
template <class Phys, int DIM, bool SWEEP2>
__global__ __launch_bounds__(GT) void k_g_flux(Geo g, GBox box, typename Phys::Params P, const double* __restrict__ U,
                                                Ptr3 F1, Ptr3 F2, int do_ctu, double dt, FaceBC3 bc, int* err) {
    int idx[3];
    if (!g_cell(box, idx[0], idx[1], idx[2])) return;
    dt = launch_dt(g, dt);  // device-resident time step (cmfp_advance_steps)
    const long c = g.at(idx[0], idx[1], idx[2]);
    g_flux_dir<Phys, DIM, SWEEP2, 0>(g, P, U, F1, F2, do_ctu, dt, bc, err, idx, c);
    if constexpr (DIM > 1) g_flux_dir<Phys, DIM, SWEEP2, 1>(g, P, U, F1, F2, do_ctu, dt, bc, err, idx, c);
    if constexpr (DIM > 2) g_flux_dir<Phys, DIM, SWEEP2, 2>(g, P, U, F1, F2, do_ctu, dt, bc, err, idx, c);
}

// calc_divergence MFP_eulerian.cpp:1614-1662 and new = U + dU (MFP_advance.cpp:220-230); with
// REDUCE the CFL speed of the values just written (EulerianState::get_allowed_time_step :255-300)
template <int NC, int NF, int DIM, bool REDUCE>
__global__ __launch_bounds__(GT) void k_g_update(Geo g, GBox box, GasP gas, const double* __restrict__ Uin,
                                                  double* __restrict__ Uout, Ptr3 F, double dt, unsigned long long* red) {
    int i = 0, j = 0, k = 0;
    const bool on = g_cell(box, i, j, k);
    dt = launch_dt(g, dt);  // device-resident time step (cmfp_advance_steps)
    double red_speed = 2.2250738585072014e-308;
    if (on) {
        const long str[3] = {1, g.sj, g.sk};
        const long c = g.at(i, j, k);
        double Unew[NC];
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            double du = 0.0;
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                const double dxinv = dt / g.dx[d];
                if (n < NF)
                    du += dxinv * (F.p[d][c + n * g.sc] - F.p[d][c + str[d] + n * g.sc]);
                else
                    du += dxinv * (0.0 - 0.0);
            }
            Unew[n] = Uin[c + n * g.sc] + du;
            Uout[c + n * g.sc] = Unew[n];
        }
        if constexpr (REDUCE) {
            double s[3];
            hydro_speed_from_cons(gas, Unew, s);
#pragma unroll
            for (int d = 0; d < DIM; ++d) red_speed = std_max(red_speed, s[d]);
        }
    }
    if constexpr (REDUCE) {
        __shared__ double shr[GT / 64];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) red_speed = fmax(red_speed, __shfl_xor(red_speed, o, 64));
        if ((threadIdx.x & 63) == 0) shr[threadIdx.x >> 6] = red_speed;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < GT / 64; ++w) red_speed = fmax(red_speed, shr[w]);
            atomicMax(&red[blockIdx.x & 255], (unsigned long long)__double_as_longlong(red_speed));
        }
    }
}

static GBox make_box(const Geo& g, int dim, const int lo[3], const int hi[3]) {
    GBox b;
    for (int d = 0; d < 3; ++d) {
        b.lo[d] = d < dim ? lo[d] : 0;
        b.ext[d] = d < dim ? hi[d] - lo[d] + 1 : 1;
    }
    b.bpp = (unsigned)(((long)b.ext[0] * b.ext[1] + GT - 1) / GT);
    b.nblk = b.bpp * (unsigned)b.ext[2];
    return b;
}
static dim3 grid_of(const GBox& b) { return dim3(((b.nblk + 7u) / 8u) * 8u); }

static FaceBC3 make_bc3(const StatePhys& sp, const int touches[3][2]) {
    FaceBC3 b;
    for (int d = 0; d < 3; ++d)
        for (int s = 0; s < 2; ++s) {
            b.touch[d][s] = touches[d][s];
            for (int n = 0; n < 10; ++n) {
                int code = (n < sp.nc) ? sp.bc[n][d][s] : -666;  // MFP_eulerian.cpp:796-835
                b.rule[d][s][n] = (code == 2 || code == 4 || code == 1) ? FILL_EVEN : (code == -1 ? FILL_ODD : FILL_SKIP);
            }
        }
    return b;
}

template <class Phys, int DIM>
static int launch_gather(const Geo& g, const typename Phys::Params& P, const StatePhys& sp, const double* Uin, double* Uout,
                         const Ptr3& F1, const Ptr3& F2, int do_ctu, double dt, const int touches[3][2], int* err,
                         unsigned long long* red, hipStream_t st) {
    if (!Uin) return 1;  // coverage probe
    constexpr int NC = Phys::NC, NF = Phys::NF;
    const bool sweep2 = do_ctu && DIM > 1;
    const FaceBC3 bc = make_bc3(sp, touches);
    int lo[3], hi[3];
    // sweep 1: with corner transport the flux box is grown by one transversally (MFP_CTU_hydro.cpp:254-268)
    for (int d = 0; d < 3; ++d) {
        lo[d] = sweep2 ? -1 : 0;
        hi[d] = g.n[d];
    }
    GBox b1 = make_box(g, DIM, lo, hi);
    hipLaunchKernelGGL((k_g_flux<Phys, DIM, false>), grid_of(b1), dim3(GT), 0, st, g, b1, P, Uin, F1, F2, do_ctu, dt, bc, err);
    if (sweep2) {
        for (int d = 0; d < 3; ++d) lo[d] = 0;
        GBox b2 = make_box(g, DIM, lo, hi);
        hipLaunchKernelGGL((k_g_flux<Phys, DIM, true>), grid_of(b2), dim3(GT), 0, st, g, b2, P, Uin, F1, F2, do_ctu, dt, bc, err);
    }
    for (int d = 0; d < 3; ++d) {
        lo[d] = 0;
        hi[d] = g.n[d] - 1;
    }
    GBox b3 = make_box(g, DIM, lo, hi);
    const Ptr3& F = sweep2 ? F2 : F1;
    GasP gas{1.0, 0.0, 1.4};
    if constexpr (Phys::IS_HYDRO) {
        gas = P.gas;
        if (red) {
            hipLaunchKernelGGL((k_g_update<NC, NF, DIM, true>), grid_of(b3), dim3(GT), 0, st, g, b3, gas, Uin, Uout, F, dt, red);
            return 1;
        }
    }
    hipLaunchKernelGGL((k_g_update<NC, NF, DIM, false>), grid_of(b3), dim3(GT), 0, st, g, b3, gas, Uin, Uout, F, dt, red);
    return 1;
}

#ifndef CMFP_GDIM
#error "compile with -DCMFP_GDIM=1|2|3"
#endif
#define CMFP_CAT2(a, b) a##b
#define CMFP_CAT(a, b) CMFP_CAT2(a, b)

// one translation unit per dimensionality (parallel builds): ctu_gather_1 / _2 / _3
int CMFP_CAT(ctu_gather_, CMFP_GDIM)(const Geo& g, const StatePhys& sp, const double* Uin, double* Uout, const Ptr3& F1,
                                     const Ptr3& F2, int do_ctu, double dt, const int touches[3][2], int* err,
                                     unsigned long long* red, hipStream_t st) {
    if (g.dim != CMFP_GDIM) return 0;
#define HY(R, F)                       \
    if (sp.recon == R && sp.flux == F) \
        return launch_gather<HydroF<R, F>, CMFP_GDIM>(g, sp.hp, sp, Uin, Uout, F1, F2, do_ctu, dt, touches, err, red, st);
#define FI(R) \
    if (sp.recon == R) return launch_gather<FieldF<R>, CMFP_GDIM>(g, sp.fp, sp, Uin, Uout, F1, F2, do_ctu, dt, touches, err, red, st);
    if (sp.kind == 0) {
        HY(1, 0) HY(1, 1) HY(1, 2) HY(2, 0) HY(2, 1) HY(2, 2) HY(3, 0) HY(3, 1) HY(4, 0) HY(4, 1)
        HY(0, 0) HY(0, 1) HY(5, 0) HY(5, 1) HY(6, 0) HY(6, 1)
    } else {
        FI(5) FI(1) FI(2) FI(6) FI(3) FI(4) FI(0)
    }
#undef HY
#undef FI
    return 0;
}

}  // namespace CMFP_NS
