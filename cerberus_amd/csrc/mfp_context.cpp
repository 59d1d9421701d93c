// mfp_context.cpp — the C ABI of include/cerberus_mfp.h: per-GPU context,
// HBM-resident state, the Action-API mirror and the level integrator glue
// (MFP::advance / advance_RK / advance_strang / estTimeStep).  Host code only;
// every kernel is reached through the cmfp::Launchers table of the selected
// floating-point mode.  There is no CPU fallback: without a HIP device every
// compute entry point returns CMFP_ERR_NODEVICE.
#include "../../include/cerberus_mfp.h"
#include "mfp_types.h"
#include "mfp_elliptic.h"
#include "mfp_peer.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <string>
#include <vector>

using namespace cmfp;

namespace {

enum Status { Inactive = 0, Local, Filled, Expired, Changed, Zero };  // UpdateData::Status, MFP_action.H:17-24

int recon_num_grow(int r) {
    switch (r) {
    case CMFP_RECON_CONSTANT: return 1;
    case CMFP_RECON_MINMOD:
    case CMFP_RECON_VANLEER:
    case CMFP_RECON_MC:
    case CMFP_RECON_CENTRE: return 2;
    case CMFP_RECON_O6:
    case CMFP_RECON_WENO: return 4;
    }
    return -1;
}

struct StateRT {
    cmfp_state_desc desc;
    StatePhys sp;
    // level data (ghosted, common geometry)
    double* new_data = nullptr;
    double* old_data = nullptr;
    double* alt = nullptr;  // ping-pong partner of new_data for the fused flux kernel
    // UpdateData
    double* U = nullptr;
    double* dU = nullptr;
    Status U_status = Inactive, dU_status = Inactive;
    // staged scratch
    double* prim = nullptr;
    double* rlo[3] = {nullptr, nullptr, nullptr};
    double* rhi[3] = {nullptr, nullptr, nullptr};
    double* flux[3] = {nullptr, nullptr, nullptr};
    // Dirichlet data of the physical faces (cmfp_state_set_face_values): device tables [dim][side][prim]
    double* tab_cell[3][2][CMFP_MAX_COMP] = {};
    double* tab_face[3][2][CMFP_MAX_COMP] = {};
    bool has_tables = false;
    double* dyn[CMFP_MAX_COMP] = {};      // 'dynamic' components (cmfp_state_set_dynamic): values over the ghosted block
    double* dyn_buf[CMFP_MAX_COMP] = {};  // their allocations (kept when an entry is removed, reused when it returns)
    bool has_dynamic = false;
    // field: the materials check (mu == ep == 1 everywhere) has been made for the data now in new_data;
    // the ping-pong partner holds mu, ep too (the vacuum kernel neither reads nor rewrites them)
    bool vac_known = false, alt_mat = false;
    // ghost layers the fills and the halo exchange keep current: the state's own stencil reach (sp.ng), or the
    // field's where the field's CTU kernel applies source stage A to the species it reads (the prologue)
    int fill_ng = 0;
};

}  // namespace

struct cmfp_context {
    cmfp_geometry geom;
    cmfp_params params;
    Geo g;
    int device = 0;
    hipStream_t stream = nullptr;
    const Launchers* L = nullptr;
    std::vector<StateRT> states;
    std::vector<cmfp_action_desc> actions;
    int nsteps = 2;
    int fused = 1;                        // 0 staged, 1 fused (2-D: tiled kernel, else gather kernels), 2 gather kernels
    double* gF1[3] = {nullptr, nullptr, nullptr};  // face-flux scratch of the gather kernels, shared by all states
    double* gF2[3] = {nullptr, nullptr, nullptr};
    int touches[3][2];   // block face is a non-periodic physical boundary
    int split[3];        // domain is split between ranks along d
    cmfp_exchange_fn exchange = nullptr;
    cmfp_dt_reduce_fn dt_reduce = nullptr;
    void* dt_reduce_user = nullptr;
    void* exchange_user = nullptr;
    int exchange_batched = 0;
    int src_epilogue = 1;      // Strang stage C inside the field's CTU kernel where that applies (CMFP_SRC_EPILOGUE=0: never)
    int src_prologue = 1;      // Strang stage A inside the 2-D marching CTU kernels where that applies (CMFP_SRC_PROLOGUE=0: never)
    int march_wrap = 1;        // periodic directions of an unsplit 2-D block wrapped by the marching kernels, no ghost fill (CMFP_MARCH_WRAP=0: never)
    int last_wrap = 0;         // directions the last fused step wrapped in its kernels (bit d)
    // cmfp_advance_steps: the fused step leaves its fold of the reduction words here instead of launching it, and
    // the dt update of the step folds them in the same launch (k_fold_new_dt)
    bool defer_fold = false, fold_pending = false;
    RedJobs pending_fold;
    int last_prologue = 0;     // the last fused step ran stage A as the CTU kernels' prologue
    int last_epilogue = 0;     // ... and stage C as the field kernel's epilogue
    int overlap = 1;           // 1: interior flux tiles overlap the halo exchange where the block is split; 2: always split the launches
    hipStream_t stream2 = nullptr;  // interior tiles run here while ghost cells are in flight on `stream`
    hipEvent_t ev_src = nullptr, ev_int = nullptr, ev_hyd = nullptr;  // one callback per dimension for all states (state = CMFP_ALL_STATES)
    int* d_err = nullptr;
    unsigned long long* d_errcell = nullptr;  // scratch word of the post-mortem scan (first_invalid)
    int err_state = -1, err_cell[3] = {-1, -1, -1};  // first offending cell of the last CMFP_ERR_PHYSICS, if one was found
    int* d_flag = nullptr;      // result word of one-off device checks
    double* d_dtbuf = nullptr;  // cmfp_advance_steps: device-resident dt + dt log
    int d_dtcap = 0;
    cmfp_ell::Work ell;  // divergence projection scratch (cmfp_elliptic_project)
    // device-side halo / dt transport between ranks (cmfp_peer_*, mfp_peer.h)
    struct PeerRT {
        bool enabled = false;
        int rank = -1, world = 0;
        char* arena = nullptr;                         // this rank's arena (fine-grained device memory)
        bool fine = false;
        char* remote[cmfp_peer::MAX_RANKS] = {};       // every rank's arena as this process sees it; remote[rank] = arena
        bool opened[cmfp_peer::MAX_RANKS] = {};        // mapped with hipIpcOpenMemHandle: closed by cmfp_destroy
        int nb[3][2] = {{-1, -1}, {-1, -1}, {-1, -1}}; // neighbour rank of face (d, side), -1: none
        unsigned long long msg[3] = {0, 0, 0};         // halo messages sent so far, per dimension
        unsigned long long dtmsg = 0;
        std::vector<size_t> state_off[3];              // doubles: offset of each state's face inside a slot of dimension d
        cmfp_peer::Layout L;
        double timeout_s = 20.0;                       // bound of every device-side wait (CMFP_PEER_TIMEOUT)
    } peer;
    bool no_alloc = false;  // inside cmfp_advance_steps_static: anything that would allocate fails instead
    unsigned long long* d_red = nullptr;  // one reduction word per state + per action
    double* d_part = nullptr;             // CMFP_RED_BLOCKS block partials per word
    std::vector<char> word_clean;         // per word: its 256 atomic slots hold the identity of its fold (k_reduce_many left them so)
    std::vector<char> red_valid;          // word already holds the reduction of the current new-time data
    std::string err;
    int last_fused = 0;
    int profiling = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof[CMFP_PROF_NSLOTS];
    long prof_uncounted[CMFP_PROF_NSLOTS] = {0, 0, 0, 0};
    std::vector<hipEvent_t> ev_pool;
    std::vector<double*> allocs;
    std::vector<double> staging;
};

namespace {

int fail(cmfp_context* c, int code, const std::string& msg) {
    if (c) {
        c->err = msg;
        // cmfp_last_error_cell describes the *last* error: any failure forgets the cell of an earlier
        // CMFP_ERR_PHYSICS (sync_and_check sets it again after this call where it found one)
        c->err_state = -1;
        c->err_cell[0] = c->err_cell[1] = c->err_cell[2] = -1;
    }
    return code;
}
#define HIPCHK(c, call)                                                                              \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(c, CMFP_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));         \
    } while (0)

// event pair around a kernel launch while profiling is enabled
struct ProfScope {
    cmfp_context* c;
    int slot;
    hipStream_t st;
    bool count;  // false: the time is added to the slot but not counted as a launch (second half of a split launch)
    hipEvent_t a = nullptr, b = nullptr;
    static hipEvent_t get(cmfp_context* c) {
        hipEvent_t e = nullptr;
        if (!c->ev_pool.empty()) {
            e = c->ev_pool.back();
            c->ev_pool.pop_back();
        } else if (c->no_alloc || hipEventCreate(&e) != hipSuccess) {
            e = nullptr;  // no timing for this launch
        }
        return e;
    }
    ProfScope(cmfp_context* ctx, int s) : ProfScope(ctx, s, ctx->stream, true) {}
    ProfScope(cmfp_context* ctx, int s, hipStream_t stream, bool cnt) : c(ctx), slot(s), st(stream), count(cnt) {
        if (!c->profiling) return;
        a = get(c);
        b = get(c);
        if (a) (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (!a || !b) return;
        (void)hipEventRecord(b, st);
        c->prof[slot].push_back({a, b});
        if (!count) c->prof_uncounted[slot]++;
    }
};

int dalloc(cmfp_context* c, double** p, long n) {
    if (*p) return CMFP_OK;
    if (c->no_alloc)
        return fail(c, CMFP_ERR_INVALID, "cmfp_advance_steps_static: the fused step still has scratch to allocate for this kernel "
                                         "selection; run one cmfp_advance first");
    HIPCHK(c, hipMalloc((void**)p, sizeof(double) * (size_t)n));
    c->allocs.push_back(*p);
    return CMFP_OK;
}

int ensure_update(cmfp_context* c, StateRT& s) {
    int r;
    if ((r = dalloc(c, &s.U, s.sp.nc * c->g.sc))) return r;
    if ((r = dalloc(c, &s.dU, s.sp.nc * c->g.sc))) return r;
    if ((r = dalloc(c, &s.old_data, s.sp.nc * c->g.sc))) return r;
    return CMFP_OK;
}
int ensure_staged(cmfp_context* c, StateRT& s) {
    int r;
    if ((r = dalloc(c, &s.prim, s.sp.nq * c->g.sc))) return r;
    for (int d = 0; d < c->g.dim; ++d) {
        if ((r = dalloc(c, &s.rlo[d], s.sp.nq * c->g.sc))) return r;
        if ((r = dalloc(c, &s.rhi[d], s.sp.nq * c->g.sc))) return r;
        if ((r = dalloc(c, &s.flux[d], s.sp.nc * c->g.sc))) return r;
    }
    return CMFP_OK;
}

// ---- device-side transport between ranks (mfp_peer.h) -------------------------------------------------------
long long halo_doubles(const cmfp_context* c, int s, int dim) {
    long long n = (long long)c->states[s].sp.nc * c->states[s].fill_ng;
    for (int t = 0; t < 3; ++t)
        if (t != dim) n *= c->g.ext[t];
    return n;
}
// arena layout: identical on every rank (same block shape, same states, same ghost widths)
void peer_layout(cmfp_context* c) {
    auto& P = c->peer;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        const size_t at = off;
        off += (bytes + 255) & ~(size_t)255;
        return at;
    };
    P.L.ready = take(sizeof(unsigned long long) * 6);
    P.L.dtready = take(sizeof(unsigned long long) * cmfp_peer::MAX_RANKS);
    P.L.dtslot = take(sizeof(double) * 2 * cmfp_peer::MAX_RANKS);
    for (int d = 0; d < 3; ++d) {
        P.state_off[d].assign(c->states.size(), 0);
        size_t n = 0;
        if (d < c->g.dim && c->split[d])
            for (size_t q = 0; q < c->states.size(); ++q) {
                P.state_off[d][q] = n;
                n += (size_t)halo_doubles(c, (int)q, d);
            }
        for (int side = 0; side < 2; ++side)
            for (int par = 0; par < 2; ++par) P.L.recv[d][side][par] = take(sizeof(double) * n);
    }
    P.L.bytes = off;
}
// the ghost layers of `list` across the split dimension d: what the exchange callback does, without the host
int peer_exchange(cmfp_context* c, const int* list, int nlist, int d) {
    auto& P = c->peer;
    const unsigned long long m = ++P.msg[d];
    const int par = (int)(m & 1ull);
    cmfp_peer::Flags2 sig, wt;
    sig.n = wt.n = 0;
    sig.value = wt.value = m;
    for (int side = 0; side < 2; ++side) {
        const int nb = P.nb[d][side];
        if (nb < 0) continue;
        // my face `side` is the neighbour's ghost region on its opposite side
        double* slot = reinterpret_cast<double*>(P.remote[nb] + P.L.recv[d][1 - side][par]);
        for (int k = 0; k < nlist; ++k) {
            const StateRT& st = c->states[list[k]];
            cmfp_peer::pack(c->g, st.sp.nc, st.fill_ng, st.new_data, d, side, slot + P.state_off[d][list[k]], c->stream);
        }
        sig.flag[sig.n++] = reinterpret_cast<unsigned long long*>(P.remote[nb] + P.L.ready) + (2 * d + (1 - side));
        wt.flag[wt.n++] = reinterpret_cast<unsigned long long*>(P.arena + P.L.ready) + (2 * d + side);
    }
    cmfp_peer::signal(sig, c->stream);
    cmfp_peer::wait(wt, c->d_err + 1, P.timeout_s, c->stream);
    for (int side = 0; side < 2; ++side) {
        if (P.nb[d][side] < 0) continue;
        const double* slot = reinterpret_cast<const double*>(P.arena + P.L.recv[d][side][par]);
        for (int k = 0; k < nlist; ++k) {
            StateRT& st = c->states[list[k]];
            cmfp_peer::unpack(c->g, st.sp.nc, st.fill_ng, st.new_data, d, side, slot + P.state_off[d][list[k]], c->stream);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, CMFP_ERR_HIP, std::string("peer exchange launch: ") + hipGetErrorString(e));
    return CMFP_OK;
}
// ParallelDescriptor::ReduceRealMin of the device-resident dt (MFP_timestep.cpp:18)
int peer_dt_min(cmfp_context* c, double* d_dt) {
    auto& P = c->peer;
    cmfp_peer::DtPeers a;
    std::memset(&a, 0, sizeof(a));
    a.rank = P.rank;
    a.world = P.world;
    for (int r = 0; r < P.world; ++r) a.arena[r] = P.remote[r];
    a.dtready = P.L.dtready;
    a.dtslot = P.L.dtslot;
    a.value = ++P.dtmsg;
    cmfp_peer::dt_put(a, d_dt, c->stream);
    cmfp_peer::dt_min(a, d_dt, c->d_err + 1, P.timeout_s, c->stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, CMFP_ERR_HIP, std::string("peer dt minimum launch: ") + hipGetErrorString(e));
    return CMFP_OK;
}

FillRules bc_rules(const StateRT& s, int d, int side) {
    FillRules r;
    for (int n = 0; n < 10; ++n) {
        if (n >= s.sp.nc) {
            r.rule[n] = FILL_SKIP;
            continue;
        }
        switch (s.sp.bc[n][d][side]) {
        case CMFP_BC_REFLECT_EVEN: r.rule[n] = FILL_EVEN; break;
        case CMFP_BC_REFLECT_ODD: r.rule[n] = FILL_ODD; break;
        default: r.rule[n] = FILL_FOEXTRAP; break;  // foextrap; hoextrap/ext_dir/int_dir are not produced by MFP_bc.cpp here
        }
    }
    return r;
}
FillRules wrap_rules(const StateRT& s) {
    FillRules r;
    for (int n = 0; n < 10; ++n) r.rule[n] = n < s.sp.nc ? FILL_WRAP : FILL_SKIP;
    return r;
}

// FillPatch of the new-time data, in place (MFP_action.cpp:59 -> MFP_fillbc.cpp:147-172)
int fill_patch(cmfp_context* c, int si) {
    StateRT& s = c->states[si];
    const Geo& g = c->g;
    ProfScope prof(c, CMFP_PROF_FILL);
    for (int d = 0; d < g.dim; ++d) {
        if (!c->split[d]) {
            for (int side = 0; side < 2; ++side) {
                FillRules r = c->geom.is_periodic[d] ? wrap_rules(s) : bc_rules(s, d, side);
                c->L->fill_dim(g, s.sp.nc, s.fill_ng, s.new_data, d, side, r, c->stream);
            }
        } else {
            for (int side = 0; side < 2; ++side)
                if (c->touches[d][side])
                    c->L->fill_dim(g, s.sp.nc, s.fill_ng, s.new_data, d, side, bc_rules(s, d, side), c->stream);
            if (c->peer.enabled) {
                int r = peer_exchange(c, &si, 1, d);
                if (r) return r;
                continue;
            }
            if (!c->exchange)
                return fail(c, CMFP_ERR_INVALID, "domain is split along a dimension but no exchange callback is set");
            int r = c->exchange(c->exchange_user, c, si, d);
            if (r) return fail(c, CMFP_ERR_INVALID, "exchange callback failed");
        }
    }
    return CMFP_OK;
}

// The same fill for several states at once, dimension by dimension: the states are independent, so
// this equals fill_patch per state; with a batched exchange callback one message per neighbour and
// dimension carries the ghost layers of every state.
int fill_patch_many(cmfp_context* c, const std::vector<int>& list, int skip_dims = 0) {
    const Geo& g = c->g;
    if (skip_dims == (1 << g.dim) - 1) return CMFP_OK;  // nothing to launch
    ProfScope prof(c, CMFP_PROF_FILL);
    for (int d = 0; d < g.dim; ++d) {
        if ((skip_dims >> d) & 1) continue;  // periodic on this block and wrapped by the kernels themselves
        // both sides of up to four states per launch (sides and states are independent of each other)
        FillJobs J;
        J.n = 0;
        auto flush = [&]() {
            if (J.n) c->L->fill_multi(g, d, J, c->stream);
            J.n = 0;
        };
        for (int si : list) {
            StateRT& s = c->states[si];
            FillJob& f = J.j[J.n];
            f.A = s.new_data;
            f.nc = s.sp.nc;
            f.ngs = s.fill_ng;
            for (int side = 0; side < 2; ++side) {
                f.on[side] = !c->split[d] || c->touches[d][side];
                f.rules[side] = (!c->split[d] && c->geom.is_periodic[d]) ? wrap_rules(s) : bc_rules(s, d, side);
            }
            if (f.on[0] || f.on[1]) ++J.n;
            if (J.n == 4) flush();
        }
        flush();
        if (!c->split[d]) continue;
        if (c->peer.enabled) {  // device-side transport: one message per neighbour carries the listed states
            int r = peer_exchange(c, list.data(), (int)list.size(), d);
            if (r) return r;
            continue;
        }
        if (!c->exchange) return fail(c, CMFP_ERR_INVALID, "domain is split along a dimension but no exchange callback is set");
        if (c->exchange_batched) {
            if (c->exchange(c->exchange_user, c, CMFP_ALL_STATES, d)) return fail(c, CMFP_ERR_INVALID, "exchange callback failed");
        } else {
            for (int si : list)
                if (c->exchange(c->exchange_user, c, si, d)) return fail(c, CMFP_ERR_INVALID, "exchange callback failed");
        }
    }
    return CMFP_OK;
}

SourceArgs source_args(cmfp_context* c, const cmfp_action_desc& A, bool on_new) {
    SourceArgs a;
    std::memset(&a, 0, sizeof(a));
    a.n_species = A.n_states - 1;
    for (int n = 0; n < a.n_species; ++n) {
        StateRT& s = c->states[A.states[n]];
        a.sp_U[n] = on_new ? s.new_data : s.U;
        a.sp_dU[n] = s.dU;
        a.r[n] = s.desc.charge / s.desc.mass;  // r = q / m  (MFP_plasma5.cpp:199-202)
        if (s.desc.mixture) {  // get_mass_from_cons / get_charge_from_cons per cell
            SpeciesMix& M = a.mix[n];
            for (int i = 0; i <= s.desc.n_tracers; ++i) {
                M.m[i] = s.desc.species_mass[i];
                M.q[i] = s.desc.species_charge[i];
                if (M.m[i] != M.m[0]) M.flags |= 1;
                if (M.q[i] != M.q[0]) M.flags |= 2;
            }
            M.nt = M.flags ? s.desc.n_tracers : 0;
        }
    }
    StateRT& f = c->states[A.states[A.n_states - 1]];
    a.f_U = on_new ? f.new_data : f.U;
    a.f_dU = f.dU;
    a.Larmor = c->params.Larmor;
    a.Debye = c->params.Debye;
    a.lightspeed = c->params.lightspeed;
    a.c_h = f.sp.fp.ch;
    a.div_damp = f.desc.div_damping;
    a.lorentz_only = A.kind == CMFP_ACTION_LORENTZ;
    a.implicit = (A.kind == CMFP_ACTION_PLASMA5 && A.solver == CMFP_SOLVER_IMPLICIT);
    a.limits_dt = (A.kind == CMFP_ACTION_PLASMA5 && A.solver == CMFP_SOLVER_EXPLICIT);  // MFP_plasma5.cpp:461
    const int ai = (int)(&A - c->actions.data());
    const int ns = (int)c->states.size();
    for (int n = 0; n < a.n_species; ++n) {
        a.gamma[n] = c->states[A.states[n]].desc.gamma;
        a.part_speed[n] = c->d_part + (size_t)A.states[n] * CMFP_RED_BLOCKS;
        a.out_speed[n] = c->d_red + A.states[n];
    }
    a.part_dt = c->d_part + (size_t)(ns + ai) * CMFP_RED_BLOCKS;
    a.out_dt = c->d_red + ns + ai;
    return a;
}

// constants of a collisions_rambo action (MFP_collisions_rambo.cpp:57-64)
CollArgs coll_args(cmfp_context* c, const cmfp_action_desc& A) {
    CollArgs k;
    std::memset(&k, 0, sizeof(k));
    k.on = 1;
    for (int n = 0; n < A.n_states; ++n) {
        const StateRT& s = c->states[A.states[n]];
        k.m[n] = s.desc.mass;
        k.q[n] = s.desc.charge;
        k.gamma[n] = s.desc.gamma;
        k.ez[n] = s.desc.effective_zero;
        for (int b = 0; b < A.n_states; ++b) k.ccs[n][b] = A.cross_sections[n][b];
    }
    const double n0 = c->params.n0, Debye = c->params.Debye;
    k.c1 = 0.02116454531141366 / (n0 * std::pow(Debye, 4));
    k.c2 = 0.4135669939329334;
    k.c3 = 2.127692162140974 * n0;
    k.err = c->d_err;
    return k;
}

// Action::get_data MFP_action.cpp:15-102 for one (state, fill) option
int get_data_one(cmfp_context* c, int si, bool fill) {
    StateRT& s = c->states[si];
    const Geo& g = c->g;
    int r;
    if ((r = ensure_update(c, s))) return r;
    if (fill) {
        if (s.U_status != Filled) {
            if ((r = fill_patch(c, si))) return r;
            c->L->copy(g, s.sp.nc, s.U, s.new_data, 1, c->stream);
            s.U_status = Filled;
        }
    } else {
        if (s.U_status != Local && s.U_status != Filled) {
            c->L->copy(g, s.sp.nc, s.U, s.new_data, 0, c->stream);
            s.U_status = Local;
        }
    }
    if (s.dU_status == Expired || s.dU_status == Inactive) {
        c->L->zero(g, s.sp.nc, s.dU, c->stream);
        s.dU_status = Zero;
    }
    return CMFP_OK;
}

int action_get_data(cmfp_context* c, int ai) {
    const cmfp_action_desc& A = c->actions[ai];
    int r;
    if (A.kind == CMFP_ACTION_CTU) {  // HydroCTU::get_data MFP_CTU_hydro.cpp:43-52
        for (int i = 0; i < A.n_states; ++i)
            if ((r = get_data_one(c, A.states[i], true))) return r;
    } else if (A.kind == CMFP_ACTION_COLLISIONS_RAMBO) {  // Collisions::get_data MFP_collisions.cpp:83-93
        for (int i = 0; i < A.n_states; ++i)
            if ((r = get_data_one(c, A.states[i], false))) return r;
    } else {  // Plasma5::get_data MFP_plasma5.cpp:60-71, Lorentz::get_data MFP_lorentz.cpp:47-58
        if ((r = get_data_one(c, A.states[A.n_states - 1], false))) return r;
        for (int i = 0; i < A.n_states - 1; ++i)
            if ((r = get_data_one(c, A.states[i], false))) return r;
    }
    return CMFP_OK;
}

int action_calc_time_derivative(cmfp_context* c, int ai, double dt) {
    const cmfp_action_desc& A = c->actions[ai];
    if (A.kind == CMFP_ACTION_CTU) return CMFP_OK;  // Action::calc_time_derivative default (MFP_action.H:67-70)
    if (A.kind == CMFP_ACTION_COLLISIONS_RAMBO) {  // Collisions::calc_time_derivative MFP_collisions.cpp:95-267
        SourceArgs a;
        std::memset(&a, 0, sizeof(a));
        a.n_species = A.n_states;
        for (int n = 0; n < A.n_states; ++n) {
            StateRT& s = c->states[A.states[n]];
            a.sp_U[n] = s.U;
            a.sp_dU[n] = s.dU;
            s.dU_status = Changed;
        }
        a.coll = coll_args(c, A);
        bool sub = false;  // a state with sub-components: the reference's accounting over alpha (:218-257)
        for (int n = 0; n < A.n_states; ++n) sub = sub || c->states[A.states[n]].desc.n_tracers > 0;
        if (sub) {
            CollMix cm;
            std::memset(&cm, 0, sizeof(cm));
            for (int n = 0; n < A.n_states; ++n) {
                const StateRT& s = c->states[A.states[n]];
                const int k = A.species_component[n];
                cm.comp[n] = k;
                cm.m[n] = s.desc.mixture ? s.desc.species_mass[k] : s.desc.mass;
                cm.q[n] = s.desc.mixture ? s.desc.species_charge[k] : s.desc.charge;
                cm.hp[n] = s.sp.hp;
            }
            c->L->collisions_mix(c->g, a, cm, dt, c->stream);
        } else {
            c->L->collisions(c->g, a, dt, c->stream);
        }
        return CMFP_OK;
    }
    SourceArgs a = source_args(c, A, false);
    for (int n = 0; n < A.n_states - 1; ++n) c->states[A.states[n]].dU_status = Changed;
    if (A.kind == CMFP_ACTION_PLASMA5) c->states[A.states[A.n_states - 1]].dU_status = Changed;  // MFP_plasma5.cpp:96,255
    c->L->source(c->g, a, dt, c->stream);
    return CMFP_OK;
}

// boxes of the Dirichlet tables of face (d, side) in block-local indices: ghost slab / face plane
IBox table_box(const Geo& g, int d, int side, bool face) {
    IBox b;
    for (int e = 0; e < 3; ++e) {
        b.lo[e] = -g.ng[e];
        b.hi[e] = g.n[e] - 1 + g.ng[e];
    }
    if (face)
        b.lo[d] = b.hi[d] = side == 0 ? 0 : g.n[d];
    else if (side == 0)
        b.hi[d] = -1;
    else
        b.lo[d] = g.n[d];
    return b;
}
// EulerianState::update_boundary_cells (MFP_eulerian.cpp:435-511): ghost cells beyond the faces with
// Dirichlet data, in the order of State::get_bc_limits (MFP_state.cpp:42-112)
void apply_cell_tables(cmfp_context* c, StateRT& s) {
    const Geo& g = c->g;
    for (int d = 0; d < g.dim; ++d)
        for (int side = 0; side < 2; ++side) {
            if (!c->touches[d][side]) continue;
            const IBox tb = table_box(g, d, side, false);
            for (int n = 0; n < s.sp.nq; ++n)
                if (s.tab_cell[d][side][n]) c->L->table_store(g, tb, tb, s.tab_cell[d][side][n], s.prim + n * g.sc, 0, c->stream);
        }
}
// the Dirichlet part of face_bc (MFP_eulerian.cpp:838-877): outside face state = value at the face centre
void apply_face_tables(cmfp_context* c, StateRT& s) {
    const Geo& g = c->g;
    const long str[3] = {1, g.sj, g.sk};
    for (int d = 0; d < g.dim; ++d)
        for (int side = 0; side < 2; ++side) {
            if (!c->touches[d][side]) continue;
            const IBox tb = table_box(g, d, side, true);
            IBox pl;  // the face plane over the valid box (update_face_prim :901-953)
            for (int e = 0; e < 3; ++e) {
                pl.lo[e] = 0;
                pl.hi[e] = g.n[e] - 1;
            }
            pl.lo[d] = pl.hi[d] = tb.lo[d];
            for (int n = 0; n < s.sp.nq; ++n)
                if (s.tab_face[d][side][n]) {
                    if (side == 0)  // dest = L = r_hi of the ghost cell
                        c->L->table_store(g, pl, tb, s.tab_face[d][side][n], s.rhi[d] + n * g.sc, -str[d], c->stream);
                    else            // dest = R = r_lo of the ghost cell
                        c->L->table_store(g, pl, tb, s.tab_face[d][side][n], s.rlo[d] + n * g.sc, 0, c->stream);
                }
        }
}

// HydroCTU::calc_spatial_derivative MFP_CTU_hydro.cpp:54-532, one kernel per stage
int ctu_staged(cmfp_context* c, const cmfp_action_desc& A, double dt) {
    const Geo& g = c->g;
    const int do_ctu = A.corner_transport != 0;
    for (int idx = 0; idx < A.n_states; ++idx) {
        StateRT& s = c->states[A.states[idx]];
        int r;
        if ((r = ensure_staged(c, s))) return r;
        s.dU_status = Changed;  // :88
        Ptr3 rlo{{s.rlo[0], s.rlo[1], s.rlo[2]}}, rhi{{s.rhi[0], s.rhi[1], s.rhi[2]}};
        Ptr3 fl{{s.flux[0], s.flux[1], s.flux[2]}};
        c->L->prim(g, s.sp, s.U, s.prim, c->d_err, c->stream);                       // :182
        if (s.has_dynamic)  // MFP_eulerian.cpp:403-423: Q[n] = f(x, y, z, t), prim2cons back into the action's copy of U
            for (int n = 0; n < s.sp.nc; ++n)
                if (s.dyn[n]) {
                    c->L->copy(g, 1, s.prim + n * g.sc, s.dyn[n], 1, c->stream);
                    c->L->copy(g, 1, s.U + n * g.sc, s.dyn[n], 1, c->stream);
                }
        if (s.has_tables) apply_cell_tables(c, s);                                   // :197
        c->L->recon(g, s.sp, s.prim, rlo, rhi, do_ctu, dt, c->stream);               // :212,230
        if (s.has_tables) apply_face_tables(c, s);                                   // :281 (do_all = false)
        for (int d = 0; d < g.dim; ++d)                                              // :297
            c->L->fluxes(g, s.sp, d, s.rlo[d], s.rhi[d], s.flux[d], do_ctu, c->stream);
        if (g.dim > 1 && do_ctu) {
            for (int d1 = 0; d1 < g.dim; ++d1)                                       // :316
                c->L->correct(g, s.sp, d1, s.rlo[d1], s.rhi[d1], fl, dt, c->d_err, c->stream);
            for (int d = 0; d < g.dim; ++d)                                          // :328
                for (int side = 0; side < 2; ++side)
                    if (c->touches[d][side]) c->L->face_bc(g, s.sp, d, side, s.rlo[d], s.rhi[d], c->stream);
            if (s.has_tables) apply_face_tables(c, s);
            for (int d = 0; d < g.dim; ++d)                                          // :349
                c->L->fluxes(g, s.sp, d, s.rlo[d], s.rhi[d], s.flux[d], do_ctu, c->stream);
        }
        if (s.sp.kind == CMFP_STATE_HYDRO && s.sp.visc.kind)                         // :370-380
            for (int d = 0; d < g.dim; ++d) c->L->viscous(g, s.sp, d, s.U, s.flux[d], c->stream);
        c->L->divergence(g, s.sp.nc, fl, s.dU, dt, c->stream);                       // :504
    }
    return CMFP_OK;
}

int action_calc_spatial_derivative(cmfp_context* c, int ai, double dt) {
    const cmfp_action_desc& A = c->actions[ai];
    if (A.kind != CMFP_ACTION_CTU) return CMFP_OK;
    return ctu_staged(c, A, dt);
}

void swap_levels(cmfp_context* c) {
    // state[i].swapTimeLevels + copy old into new (MFP_advance.cpp:10-25): old := new
    for (auto& s : c->states)
        if (s.old_data) c->L->copy(c->g, s.sp.nc, s.old_data, s.new_data, 0, c->stream);
}

// MFP::advance_RK MFP_advance.cpp:58-135
int advance_RK(cmfp_context* c, double dt) {
    const Geo& g = c->g;
    int r;
    for (auto& s : c->states) s.U_status = s.dU_status = Inactive;
    for (size_t a = 0; a < c->actions.size(); ++a) {
        if ((r = action_get_data(c, (int)a))) return r;
        if ((r = action_calc_time_derivative(c, (int)a, dt))) return r;
        if ((r = action_calc_spatial_derivative(c, (int)a, dt))) return r;
    }
    for (auto& s : c->states) {
        if (s.dU_status != Changed) continue;
        c->L->add(g, s.sp.nc, s.new_data, s.dU, c->stream);
        s.U_status = Expired;
        s.dU_status = Expired;
    }
    if (c->nsteps >= 2) {
        for (size_t a = 0; a < c->actions.size(); ++a) {
            if ((r = action_get_data(c, (int)a))) return r;
            if ((r = action_calc_time_derivative(c, (int)a, dt))) return r;
            if ((r = action_calc_spatial_derivative(c, (int)a, dt))) return r;
        }
        for (auto& s : c->states) {
            if (s.dU_status != Changed) continue;
            c->L->add(g, s.sp.nc, s.U, s.dU, c->stream);
            c->L->lincomb(g, s.sp.nc, s.new_data, 0.5, s.old_data, 0.5, s.U, c->stream);
        }
    }
    return CMFP_OK;
}

// MFP::advance_strang MFP_advance.cpp:137-289
int advance_strang(cmfp_context* c, double dt) {
    const Geo& g = c->g;
    int r;
    for (auto& s : c->states) s.U_status = s.dU_status = Inactive;
    const int na = (int)c->actions.size();

    for (int a = 0; a < na; ++a) {
        if ((r = action_get_data(c, a))) return r;
        if ((r = action_calc_time_derivative(c, a, 0.5 * dt))) return r;
    }
    for (auto& s : c->states) {
        if (s.dU_status != Changed) continue;
        if (c->nsteps >= 2) {
            c->L->add(g, s.sp.nc, s.U, s.dU, c->stream);
            c->L->zero(g, s.sp.nc, s.dU, c->stream);
            s.dU_status = Expired;
        } else {
            c->L->add(g, s.sp.nc, s.new_data, s.dU, c->stream);
        }
    }
    if (c->nsteps >= 2) {
        for (int a = 0; a < na; ++a)
            if ((r = action_calc_time_derivative(c, a, 0.5 * dt))) return r;
        for (auto& s : c->states) {
            if (s.dU_status != Changed) continue;
            c->L->add(g, s.sp.nc, s.U, s.dU, c->stream);
            c->L->zero(g, s.sp.nc, s.dU, c->stream);
            c->L->lincomb(g, s.sp.nc, s.new_data, 0.5, s.old_data, 0.5, s.U, c->stream);
            s.U_status = Expired;
            s.dU_status = Expired;
        }
    }

    for (int a = 0; a < na; ++a) {
        if ((r = action_get_data(c, a))) return r;
        if ((r = action_calc_spatial_derivative(c, a, dt))) return r;
    }
    for (auto& s : c->states) {
        if (s.dU_status != Changed) continue;
        c->L->add(g, s.sp.nc, s.new_data, s.dU, c->stream);
        s.U_status = Expired;
        s.dU_status = Expired;
    }

    for (int a = 0; a < na; ++a) {
        if ((r = action_get_data(c, a))) return r;
        if ((r = action_calc_time_derivative(c, a, 0.5 * dt))) return r;
    }
    for (auto& s : c->states) {
        if (s.dU_status != Changed) continue;
        if (c->nsteps >= 2) {
            c->L->add(g, s.sp.nc, s.U, s.dU, c->stream);
            c->L->zero(g, s.sp.nc, s.dU, c->stream);
            s.dU_status = Inactive;
        } else {
            c->L->add(g, s.sp.nc, s.new_data, s.dU, c->stream);
        }
    }
    if (c->nsteps >= 2) {
        for (int a = 0; a < na; ++a)
            if ((r = action_calc_time_derivative(c, a, 0.5 * dt))) return r;
        for (auto& s : c->states) {
            if (s.dU_status != Changed) continue;
            c->L->add(g, s.sp.nc, s.U, s.dU, c->stream);
            c->L->zero(g, s.sp.nc, s.dU, c->stream);
            c->L->lincomb(g, s.sp.nc, s.new_data, 0.5, s.new_data, 0.5, s.U, c->stream);
        }
    }
    return CMFP_OK;
}

// The same Strang step through the hand-fused kernels: one pointwise RK2
// source kernel per half step (in place on the new-time data), ghost fill in
// place, one fused CTU kernel per state writing new + dU into the ping-pong
// partner.  Returns 1 if it ran, 0 if the configuration is outside what the
// fused kernels cover (caller then takes the staged path), <0 on error.
int advance_strang_fused(cmfp_context* c, double dt, bool reserve_only = false) {
    if (c->params.time_integrator != CMFP_TI_STRANG) return 0;
    for (auto& st : c->states)
        if (st.has_dynamic) return 0;  // 'dynamic' components act on the primitives and on UpdateData::U: stage by stage
    if (!reserve_only)
        for (auto& st : c->states)
            if (st.sp.kind == CMFP_STATE_FIELD && !st.vac_known) {
                // once per upload: are the material coefficients 1 everywhere?  (They never change: F = 0 and no
                // source touches them.)  Then the 2-D O6 update can run on the marching vacuum kernel.
                int h = 1;
                HIPCHK(c, hipMemsetAsync(c->d_flag, 0, sizeof(int), c->stream));
                c->L->field_vacuum_check(c->g, st.new_data, c->d_flag, c->stream);
                HIPCHK(c, hipMemcpyAsync(&h, c->d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                st.sp.vacuum = h == 0;
                st.vac_known = true;
            }
    int n_src = 0, src = -1, coll = -1;
    std::vector<int> seen(c->states.size(), 0);
    for (size_t a = 0; a < c->actions.size(); ++a) {
        const cmfp_action_desc& A = c->actions[a];
        if (A.kind == CMFP_ACTION_CTU) {
            for (int i = 0; i < A.n_states; ++i)
                if (seen[A.states[i]]++) return 0;
        } else if (A.kind == CMFP_ACTION_COLLISIONS_RAMBO) {
            if (coll >= 0) return 0;
            coll = (int)a;
        } else {
            ++n_src;
            src = (int)a;
            if (A.n_states - 1 > 4) return 0;
        }
    }
    if (n_src > 1) return 0;
    if (coll >= 0) {
        // the collisions are evaluated inside the source kernel: same species, same order
        if (src < 0) return 0;
        const cmfp_action_desc &S = c->actions[src], &K = c->actions[coll];
        if (K.n_states != S.n_states - 1) return 0;
        for (int n = 0; n < K.n_states; ++n)
            if (K.states[n] != S.states[n] || c->states[K.states[n]].desc.n_tracers) return 0;  // sub-components: staged kernels
    }
    // probe coverage (null launches): the tiled 2-D kernel where it applies, else the gather kernels
    const int tile_flag = c->fused == 3 ? 0x100 : 0;  // 3: the LDS-tiled kernel also where the wave-marching one applies
    std::vector<char> tiled(c->states.size(), 0), f3d(c->states.size(), 0);
    bool any_gather = false;
    const Ptr3 none{{nullptr, nullptr, nullptr}};
    for (const auto& A : c->actions)
        if (A.kind == CMFP_ACTION_CTU)
            for (int i = 0; i < A.n_states; ++i) {
                const StatePhys& sp = c->states[A.states[i]].sp;
                if (sp.kind == CMFP_STATE_HYDRO && sp.visc.kind) return 0;  // viscous fluxes: stage-by-stage kernels only
                // passive scalars: the 2-D marching kernel (probed below); true mixtures: stage-by-stage kernels only
                // (desc.mixture also where only the charges differ: the sources read a per-cell charge, staged kernels)
                if (sp.kind == CMFP_STATE_HYDRO && sp.hp.nt &&
                    (sp.hp.mix || c->states[A.states[i]].desc.mixture || c->g.dim != 2 || c->fused != 1))
                    return 0;
                if (c->states[A.states[i]].has_tables) {  // Dirichlet face data: the 2-D hydro marching kernel, else stage by stage
                    MarchTab none_tab;
                    std::memset(&none_tab, 0, sizeof(none_tab));
                    if (c->fused != 1 || !c->L->ctu_march_tab(c->g, sp, nullptr, nullptr, A.corner_transport, dt, c->touches, nullptr,
                                                              nullptr, 0, none_tab, c->stream))
                        return 0;
                    tiled[A.states[i]] = 1;
                    continue;
                }
                if (c->fused != 2 && c->L->ctu_fused(c->g, sp, nullptr, nullptr, A.corner_transport, dt, c->touches, nullptr,
                                                     nullptr, 0 | tile_flag, c->stream)) {
                    tiled[A.states[i]] = 1;
                } else if (c->fused == 1 && c->L->ctu_field3d(c->g, sp, nullptr, nullptr, none, nullptr, A.corner_transport, dt, c->touches,
                                                              nullptr, c->stream)) {
                    f3d[A.states[i]] = 1;  // 3-D field: LDS plane-tile kernels (they use the F1 scratch and one F2 array)
                    any_gather = true;
                } else if (c->L->ctu_gather(c->g, sp, nullptr, nullptr, none, none, A.corner_transport, dt, c->touches,
                                            nullptr, nullptr, c->stream)) {
                    any_gather = true;
                } else {
                    return 0;
                }
            }
    if (any_gather)
        for (int d = 0; d < c->g.dim; ++d) {
            int r;
            if ((r = dalloc(c, &c->gF1[d], 8 * c->g.sc))) return r;
            if ((r = dalloc(c, &c->gF2[d], 8 * c->g.sc))) return r;
        }
    std::vector<int> list;  // every state a CTU action updates, with its action's corner_transport flag
    std::vector<int> ctu_flag;
    for (const auto& A : c->actions)
        if (A.kind == CMFP_ACTION_CTU)
            for (int i = 0; i < A.n_states; ++i) {
                list.push_back(A.states[i]);
                ctu_flag.push_back(A.corner_transport);
            }
    // Will the flux launches be split into interior and frame (see below)?  If the block is split between
    // ranks and they will not (3-D: the marching kernels take the whole block), stage A is split instead:
    // the frame of cells within ng of a face first - every ghost fill reads only those - then the interior
    // on a second stream while the ghost layers are packed, exchanged and unpacked.
    // one fused CTU launch of state list[q] (in == nullptr: coverage probe): the marching kernel with Dirichlet
    // face data where the state has tables, else whatever ctu_fused selects
    bool any_tables = false;
    for (int v : list) any_tables = any_tables || c->states[v].has_tables;
    auto ctu_launch = [&](size_t q, const double* in, double* out, unsigned long long* red, int part, hipStream_t st) -> int {
        const StateRT& s = c->states[list[q]];
        if (!s.has_tables) return c->L->ctu_fused(c->g, s.sp, in, out, ctu_flag[q], dt, c->touches, c->d_err, red, part, st);
        MarchTab t;
        std::memset(&t, 0, sizeof(t));
        for (int d = 0; d < 2; ++d)
            for (int side = 0; side < 2; ++side)
                for (int n = 0; n < 5; ++n) {
                    t.cell[d][side][n] = s.tab_cell[d][side][n];
                    t.face[d][side][n] = s.tab_face[d][side][n];
                }
        return c->L->ctu_march_tab(c->g, s.sp, in, out, ctu_flag[q], dt, c->touches, c->d_err, red, part & ~0x100, t, st);
    };
    bool split_any = false;
    for (int d = 0; d < c->g.dim; ++d) split_any = split_any || c->split[d];
    bool overlap = c->overlap == 2 || (c->overlap == 1 && split_any);
    for (size_t q = 0; q < list.size() && overlap; ++q)
        overlap = tiled[list[q]] && ctu_launch(q, nullptr, nullptr, nullptr, 1 | tile_flag, c->stream);
    if ((overlap || (src >= 0 && split_any && c->overlap != 0)) && !c->stream2) {
        if (c->no_alloc)
            return fail(c, CMFP_ERR_INVALID, "cmfp_advance_steps_static: the overlap stream does not exist yet; run one cmfp_advance first");
        HIPCHK(c, hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_src, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_int, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_hyd, hipEventDisableTiming));
    }
    for (int q : list) {  // ping-pong partners of the flux kernels
        int r;
        if ((r = dalloc(c, &c->states[q].alt, c->states[q].sp.nc * c->g.sc))) return r;
        StateRT& sq = c->states[q];
        if (!reserve_only && sq.sp.kind == CMFP_STATE_FIELD && sq.sp.vacuum && !sq.alt_mat) {
            HIPCHK(c, hipMemcpyAsync(sq.alt + 8 * c->g.sc, sq.new_data + 8 * c->g.sc, sizeof(double) * 2 * c->g.sc,
                                     hipMemcpyDeviceToDevice, c->stream));
            sq.alt_mat = true;
        }
    }
    if (reserve_only) return 1;  // cmfp_create: everything this kernel selection allocates now exists
    // Strang stage C as the epilogue of the field's CTU kernel (the last launch of the step): explicit
    // plasma5 / Lorentz over species that are all CTU states, 2-D tiled field kernel
    int epi_q = -1;
    SourceEpi epi;
    std::memset(&epi, 0, sizeof(epi));
    if (src >= 0 && coll < 0 && c->src_epilogue && c->fused == 1) {
        const cmfp_action_desc& SA = c->actions[src];
        const int fs = SA.states[SA.n_states - 1];
        bool ok = SA.n_states - 1 <= 2 && (tiled[fs] || f3d[fs]) && c->states[fs].sp.kind == CMFP_STATE_FIELD &&
                  !(SA.kind == CMFP_ACTION_PLASMA5 && SA.solver == CMFP_SOLVER_IMPLICIT);
        for (size_t q = 0; q < list.size(); ++q)
            if (list[q] == fs) epi_q = (int)q;
        ok = ok && epi_q >= 0;
        for (int n = 0; n < SA.n_states - 1 && ok; ++n) {
            bool in_list = false;
            for (int v : list) in_list = in_list || v == SA.states[n];
            ok = in_list && c->states[SA.states[n]].sp.kind == CMFP_STATE_HYDRO;
            // split launches: the field's interior launch reads the species at its cells on the second stream,
            // so its interior must lie inside every species' interior (launched before it on that stream):
            // true when the species' stencils are no wider than the field's
            if (overlap) {
                ok = ok && c->states[SA.states[n]].sp.ng <= c->states[fs].sp.ng;
                // ... checked on the rectangles the launches themselves select (tile shapes and frame widths are
                // build and kernel options): otherwise stage C stays a launch of its own
                IBox fi, si;
                int sp_flag = 1;
                for (size_t q = 0; q < list.size(); ++q)
                    if (list[q] == SA.states[n]) sp_flag = ctu_flag[q];
                if (ok && c->L->ctu_interior(c->g, c->states[fs].sp, ctu_flag[epi_q >= 0 ? epi_q : 0], tile_flag, &fi) &&
                    c->L->ctu_interior(c->g, c->states[SA.states[n]].sp, sp_flag, tile_flag, &si))
                    for (int d = 0; d < 2; ++d) ok = ok && fi.lo[d] >= si.lo[d] && fi.hi[d] <= si.hi[d];
            }
        }
        if (ok) {
            epi.ns = SA.n_states - 1;
            epi.a = source_args(c, SA, true);
            ok = f3d[fs] ? c->L->ctu_field3d(c->g, c->states[fs].sp, nullptr, nullptr, none, nullptr, ctu_flag[epi_q], dt, c->touches,
                                             &epi, c->stream) != 0
                         : c->L->ctu_field_src(c->g, c->states[fs].sp, nullptr, nullptr, ctu_flag[epi_q], dt, c->touches, epi, 0 | tile_flag,
                                               c->stream) != 0;
        }
        if (!ok) epi_q = -1;
    }
    // Strang stage A as the prologue of the marching kernels (mfp_source.h): 2-D, the two species
    // of an explicit plasma5 / Lorentz action in one paired hydro launch, the field on the marching vacuum kernel
    // with stage C as its epilogue; physical faces where their ghost cells are signed copies of one valid cell.
    // Then no kernel of the step rewrites a state in
    // place: the ghost fills and the exchange move the values of the previous step, the CTU kernels apply stage A
    // to what they load.
    SourcePro pro;
    std::memset(&pro, 0, sizeof(pro));
    if (epi_q >= 0 && c->src_prologue && c->g.dim == 2 && epi.ns == 2 && !tile_flag &&
        list.size() == 3 && !any_tables) {  // (ghost cells with Dirichlet data are not copies of valid cells)
        const cmfp_action_desc& SA = c->actions[src];
        size_t qs[2] = {0, 0};
        int found = 0;
        for (size_t q = 0; q < list.size(); ++q)
            if ((int)q != epi_q) {
                if (found < 2) qs[found] = q;
                ++found;
            }
        const StateRT &sa = c->states[list[qs[0]]], &sb = c->states[list[qs[1]]], &sf = c->states[list[epi_q]];
        bool ok = found == 2 && tiled[list[qs[0]]] && tiled[list[qs[1]]] && ctu_flag[qs[0]] == ctu_flag[qs[1]];
        ok = ok && sa.fill_ng >= sf.sp.ng && sb.fill_ng >= sf.sp.ng;  // the field's kernel reads the species over its own reach
        pro.on = 1;
        pro.fld = sf.new_data;
        pro.Larmor = epi.a.Larmor, pro.Debye = epi.a.Debye, pro.lightspeed = epi.a.lightspeed;
        pro.c_h = epi.a.c_h, pro.div_damp = epi.a.div_damp, pro.lorentz_only = epi.a.lorentz_only;
        for (int k = 0; k < 2; ++k) {  // the field kernel's order: the action's species list
            pro.sp[k] = c->states[SA.states[k]].new_data;
            pro.r[k] = epi.a.r[k];
        }
        // physical faces: every component of the three states must be a signed copy of one and the same valid cell
        // there — all mirror images (reflect_even / reflect_odd) or all the first valid cell (foextrap)
        for (int d = 0; d < 2 && ok; ++d)
            for (int side = 0; side < 2 && ok; ++side) {
                pro.touch[d][side] = c->touches[d][side];
                if (!c->touches[d][side]) continue;
                const FillRules ra = bc_rules(sa, d, side), rb = bc_rules(sb, d, side), rf = bc_rules(sf, d, side);
                int n_mirror = 0, n_copy = 0;
                for (int n = 0; n < 5; ++n) {
                    ok = ok && ra.rule[n] == rb.rule[n];
                    (ra.rule[n] == FILL_FOEXTRAP ? n_copy : n_mirror)++;
                    if (ra.rule[n] == FILL_ODD) pro.odd_h[d][side] |= 1u << n;
                }
                for (int n = 0; n < 8; ++n) {
                    (rf.rule[n] == FILL_FOEXTRAP ? n_copy : n_mirror)++;
                    if (rf.rule[n] == FILL_ODD) pro.odd_f[d][side] |= 1u << n;
                }
                ok = ok && (n_mirror == 0 || n_copy == 0);
            }
        if (ok) {
            const StatePhys* const sp2[2] = {&sa.sp, &sb.sp};
            const double* const in2[2] = {nullptr, nullptr};
            double* const out2[2] = {nullptr, nullptr};
            unsigned long long* const red2[2] = {nullptr, nullptr};
            ok = c->L->ctu_march2(c->g, sp2, in2, out2, ctu_flag[qs[0]], dt, c->touches, nullptr, red2, 0, &pro, c->stream) != 0 &&
                 c->L->ctu_march_field(c->g, sf.sp, nullptr, nullptr, ctu_flag[epi_q], dt, c->touches, &epi, 0, &pro, c->stream) != 0;
        }
        if (!ok) pro.on = 0;
    }
    c->last_prologue = pro.on;
    c->last_epilogue = epi_q >= 0;
    bool src_split = false;
    IBox src_inner;
    if (src >= 0 && !pro.on) {
        SourceArgs a = source_args(c, c->actions[src], true);
        if (coll >= 0) a.coll = coll_args(c, c->actions[coll]);
        IBox inner;
        bool has_inner = !overlap && split_any && c->overlap != 0;
        for (int d = 0; d < 3; ++d) {
            const int w = d < c->g.dim ? c->g.ng[d] : 0;
            inner.lo[d] = w;
            inner.hi[d] = c->g.n[d] - 1 - w;
            has_inner = has_inner && (d >= c->g.dim || inner.hi[d] >= inner.lo[d]);
        }
        if (has_inner) {
            // frame: slabs peeled off dimension by dimension (disjoint), on the main stream
            IBox rest;
            for (int d = 0; d < 3; ++d) {
                rest.lo[d] = 0;
                rest.hi[d] = c->g.n[d] - 1;
            }
            {
                ProfScope prof(c, CMFP_PROF_SOURCE);
                for (int d = c->g.dim - 1; d >= 0; --d) {
                    IBox lo_slab = rest, hi_slab = rest;
                    lo_slab.hi[d] = inner.lo[d] - 1;
                    hi_slab.lo[d] = inner.hi[d] + 1;
                    if (!c->L->source_fused_box(c->g, a, 0.5 * dt, lo_slab, c->stream)) return 0;
                    if (!c->L->source_fused_box(c->g, a, 0.5 * dt, hi_slab, c->stream)) return 0;
                    rest.lo[d] = inner.lo[d];
                    rest.hi[d] = inner.hi[d];
                }
            }
            HIPCHK(c, hipEventRecord(c->ev_src, c->stream));  // everything of the previous step, and the frame
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_src, 0));
            src_split = true;
            src_inner = inner;
        } else {
            ProfScope prof(c, CMFP_PROF_SOURCE);
            if (!c->L->source_fused(c->g, a, 0.5 * dt, 0, c->stream)) return 0;
        }
    }
    // Halo/compute overlap: where the block is split between ranks, the flux update of the tiles
    // whose stencils stay inside valid cells runs on a second stream while the ghost layers are
    // packed, exchanged and unpacked on the main one; the frame of boundary tiles follows the fill.
    // The reduction words this step's kernels write with atomics: one launch at the end of the step folds them
    // and leaves their slots reset for the next step (k_reduce_many); a word that something else has used since
    // is reset by one launch here.
    std::vector<int> fold_word, fold_max, fold_mode, reset_word, reset_max;
    auto use_word = [&](int w, int is_max, int mode) {
        if (!c->word_clean[w]) {
            reset_word.push_back(w);
            reset_max.push_back(is_max);
        }
        c->word_clean[w] = 0;  // until this step's fold has been enqueued
        fold_word.push_back(w);
        fold_max.push_back(is_max);
        fold_mode.push_back(mode);
        return reinterpret_cast<unsigned long long*>(c->d_part + (size_t)w * CMFP_RED_BLOCKS);
    };
    auto launch_jobs = [&](const std::vector<int>& words, const std::vector<int>& is_max, const std::vector<int>* mode) {
        for (size_t b = 0; b < words.size(); b += RedJobs::MAX) {
            RedJobs J;
            std::memset(&J, 0, sizeof(J));
            for (size_t k = b; k < words.size() && k < b + RedJobs::MAX; ++k, ++J.n) {
                J.slots[J.n] = reinterpret_cast<unsigned long long*>(c->d_part + (size_t)words[k] * CMFP_RED_BLOCKS);
                J.out[J.n] = c->d_red + words[k];
                J.is_max[J.n] = is_max[k];
                J.mode[J.n] = mode ? (*mode)[k] : 1;
            }
            c->L->reduce_many(J, c->stream);
        }
    };
    std::vector<unsigned long long*> slots(list.size(), nullptr);
    for (size_t q = 0; q < list.size(); ++q) {
        StateRT& s = c->states[list[q]];
        // a hydro state no source action touches afterwards: the CTU kernel is the last writer of
        // the step and also leaves the CFL reduction of what it wrote
        bool reduce = s.sp.kind == CMFP_STATE_HYDRO;
        if (src >= 0) {
            const cmfp_action_desc& SA = c->actions[src];
            for (int k = 0; k < SA.n_states - 1; ++k)
                if (SA.states[k] == list[q]) reduce = false;
        }
        if (reduce) slots[q] = use_word(list[q], 1, 0);
    }
    std::vector<size_t> order;  // launch order: the field with the source epilogue goes last
    for (size_t q = 0; q < list.size(); ++q)
        if ((int)q != epi_q) order.push_back(q);
    if (epi_q >= 0) {
        order.push_back((size_t)epi_q);
        const cmfp_action_desc& SA = c->actions[src];
        const int ns = (int)c->states.size();
        for (int n = 0; n < epi.ns; ++n) {
            epi.a.sp_U[n] = c->states[SA.states[n]].alt;  // where the species' CTU kernels write
            epi.red_speed[n] = use_word(SA.states[n], 1, 0);
        }
        epi.red_dt = use_word(ns + src, 0, epi.a.limits_dt ? 0 : 1);  // all-ones: empty (a NaN to fmin)
        c->L->source_coef(epi.a, 0.5 * dt, epi.kf);
    }
    launch_jobs(reset_word, reset_max, nullptr);
    // Where the marching vacuum kernel takes the field with the source epilogue it is not split into interior and
    // frame: it runs whole after the ghost fill and after every species launch (the hydro interiors still overlap
    // the exchange, which is what hides the halo traffic of a 2-D block)
    const bool field_whole = overlap && epi_q >= 0 && !f3d[list[epi_q]] &&
                             c->L->ctu_march_field(c->g, c->states[list[epi_q]].sp, nullptr, nullptr, ctu_flag[epi_q], dt, c->touches, &epi,
                                                   0 | tile_flag, nullptr, c->stream) != 0;
    // Periodic directions this block spans alone: where every CTU state runs on a marching kernel, the kernels wrap
    // their loads themselves and no ghost fill is launched for those directions (2 of the 6-7 launches of a step:
    // 8 % of a KH 2048^2 step, 14 % of a two-fluid 512^2 step).  Probed with the launch parts used below.
    int wrap_flag = 0;
    if (c->march_wrap && c->fused == 1 && c->g.dim == 2) {
        for (int d = 0; d < 2; ++d)
            if (!c->split[d] && c->geom.is_periodic[d] && (c->march_wrap == 1 || ((c->march_wrap >> 1) >> d) & 1)) wrap_flag |= 0x200 << d;
        // Measured (tools/ab_wrap.sh): wrapping y is free at every size; wrapping x costs the single-round field
        // kernel 0.17 ms at 4096^2 and 0.03 ms at 3072^2 (nothing at 2048^2 or on the 4096 x 2048 GEM block), more
        // than the fill launch it saves: x is wrapped on blocks of up to 2^23 cells, where launch gaps matter
        if (c->march_wrap == 1 && (long)c->g.n[0] * c->g.n[1] > (1L << 23)) wrap_flag &= ~0x200;
        for (size_t q = 0; q < list.size() && wrap_flag; ++q) {
            const StatePhys& sp = c->states[list[q]].sp;
            bool ok = tiled[list[q]] != 0;
            if ((int)q == epi_q)  // launched whole on the marching field kernel, or not wrapped
                ok = ok && (!overlap || field_whole || pro.on) &&
                     c->L->ctu_march_field(c->g, sp, nullptr, nullptr, ctu_flag[q], dt, c->touches, &epi, 0 | wrap_flag, nullptr, c->stream) != 0;
            else
                ok = ok && ctu_launch(q, nullptr, nullptr, nullptr, (overlap ? 1 : 0) | wrap_flag, c->stream) != 0;
            if (!ok) wrap_flag = 0;
        }
    }
    c->last_wrap = wrap_flag >> 9;
    const Ptr3 F1{{c->gF1[0], c->gF1[1], c->gF1[2]}}, F2{{c->gF2[0], c->gF2[1], c->gF2[2]}};
    auto slot_of = [&](const StateRT& s) { return s.sp.kind == CMFP_STATE_HYDRO ? CMFP_PROF_CTU_HYDRO : CMFP_PROF_CTU_FIELD; };
    // two species of the same physics share one wave-marching launch (MarchStates, mfp_march.hip)
    auto pair_launch = [&](size_t qa, size_t qb, int part, hipStream_t st) -> bool {
        if (tile_flag || (int)qa == epi_q || (int)qb == epi_q || !tiled[list[qa]] || !tiled[list[qb]] || ctu_flag[qa] != ctu_flag[qb])
            return false;
        if (c->states[list[qa]].has_tables || c->states[list[qb]].has_tables) return false;
        StateRT &a = c->states[list[qa]], &b = c->states[list[qb]];
        const StatePhys* const sp[2] = {&a.sp, &b.sp};
        const double* const in[2] = {a.new_data, b.new_data};
        double* const out[2] = {a.alt, b.alt};
        unsigned long long* const red[2] = {slots[qa], slots[qb]};
        SourcePro pr = pro;
        if (pro.on) {  // the launch's order of the species
            const cmfp_action_desc& SA = c->actions[src];
            const int ids[2] = {list[qa], list[qb]};
            for (int k = 0; k < 2; ++k) {
                pr.sp[k] = c->states[ids[k]].new_data;
                pr.r[k] = ids[k] == SA.states[0] ? epi.a.r[0] : epi.a.r[1];
            }
        }
        return c->L->ctu_march2(c->g, sp, in, out, ctu_flag[qa], dt, c->touches, c->d_err, red, part | wrap_flag, pro.on ? &pr : nullptr, st) != 0;
    };
    if (overlap) {
        HIPCHK(c, hipEventRecord(c->ev_src, c->stream));  // source stage A (and the slot resets) done
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_src, 0));
        for (size_t k = 0; k < order.size(); ++k) {
            const size_t q = order[k];
            StateRT& s = c->states[list[q]];
            // the species' interiors are complete here (the epilogue field is last in `order`): the frame launch
            // of that field on the main stream, whose tiles reach into the species' interiors, waits for this
            if ((int)q == epi_q && field_whole) continue;
            if ((int)q == epi_q) HIPCHK(c, hipEventRecord(c->ev_hyd, c->stream2));
            ProfScope prof(c, slot_of(s), c->stream2, true);
            if ((int)q == epi_q)
                c->L->ctu_field_src(c->g, s.sp, s.new_data, s.alt, ctu_flag[q], dt, c->touches, epi, 1 | tile_flag | wrap_flag, c->stream2);
            else if (k + 1 < order.size() && pair_launch(q, order[k + 1], 1, c->stream2))
                ++k;
            else if (pro.on)
                return fail(c, CMFP_ERR_INVALID, "source prologue: the paired hydro launch was refused after its probe succeeded");
            else
                ctu_launch(q, s.new_data, s.alt, slots[q], 1 | tile_flag | wrap_flag, c->stream2);
        }
        HIPCHK(c, hipEventRecord(c->ev_int, c->stream2));
    }
    {
        // ghost cells of every CTU state (stage B FillPatch, MFP_advance.cpp:209); the flux kernels
        // write the ping-pong partners, so all fills can go first
        int r;
        static const bool keep_fill = std::getenv("CMFP_WRAP_KEEPFILL") != nullptr;  // diagnostic
        if ((r = fill_patch_many(c, list, keep_fill ? 0 : wrap_flag >> 9))) return r;
    }
    if (src_split) {
        // the interior of stage A goes to the device after the packs and the exchange have been enqueued:
        // its long-lived workgroups must not keep the transfer kernels waiting for a place on the CUs
        SourceArgs a = source_args(c, c->actions[src], true);
        if (coll >= 0) a.coll = coll_args(c, c->actions[coll]);
        {
            ProfScope prof(c, CMFP_PROF_SOURCE, c->stream2, false);
            c->L->source_fused_box(c->g, a, 0.5 * dt, src_inner, c->stream2);
        }
        HIPCHK(c, hipEventRecord(c->ev_int, c->stream2));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_int, 0));
    }
    for (size_t k = 0; k < order.size(); ++k) {
        const size_t q = order[k];
        StateRT& s = c->states[list[q]];
        // the frame tiles of the epilogue field read and rewrite species cells that the species' interior
        // launches (second stream) update: wait for them
        if (overlap && (int)q == epi_q) HIPCHK(c, hipStreamWaitEvent(c->stream, field_whole ? c->ev_int : c->ev_hyd, 0));
        ProfScope prof(c, slot_of(s), c->stream, !overlap || ((int)q == epi_q && field_whole));
        if ((int)q == epi_q && pro.on) {
            if (!c->L->ctu_march_field(c->g, s.sp, s.new_data, s.alt, ctu_flag[q], dt, c->touches, &epi, 0 | wrap_flag, &pro, c->stream))
                return fail(c, CMFP_ERR_INVALID, "source prologue: the field launch was refused after its probe succeeded");
        } else if ((int)q == epi_q && field_whole)
            c->L->ctu_field_src(c->g, s.sp, s.new_data, s.alt, ctu_flag[q], dt, c->touches, epi, 0 | tile_flag | wrap_flag, c->stream);
        else if ((int)q == epi_q && !f3d[list[q]])
            c->L->ctu_field_src(c->g, s.sp, s.new_data, s.alt, ctu_flag[q], dt, c->touches, epi, (overlap ? 2 : 0) | tile_flag | wrap_flag, c->stream);
        else if (k + 1 < order.size() && pair_launch(q, order[k + 1], overlap ? 2 : 0, c->stream))
            ++k;
        else if (pro.on)
            return fail(c, CMFP_ERR_INVALID, "source prologue: the paired hydro launch was refused after its probe succeeded");
        else if (tiled[list[q]])
            ctu_launch(q, s.new_data, s.alt, slots[q], (overlap ? 2 : 0) | tile_flag | wrap_flag, c->stream);
        else if (f3d[list[q]])
            c->L->ctu_field3d(c->g, s.sp, s.new_data, s.alt, F1, c->gF2[0], ctu_flag[q], dt, c->touches, (int)q == epi_q ? &epi : nullptr,
                              c->stream);
        else
            c->L->ctu_gather(c->g, s.sp, s.new_data, s.alt, F1, F2, ctu_flag[q], dt, c->touches, c->d_err, slots[q], c->stream);
    }
    if (overlap) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_int, 0));
    // the CTU kernels' speed slots and, where stage C ran inside the field kernel, its slots: folded by one launch
    c->fold_pending = false;
    if (c->defer_fold && !fold_word.empty() && fold_word.size() <= (size_t)RedJobs::MAX) {
        RedJobs& J = c->pending_fold;
        std::memset(&J, 0, sizeof(J));
        for (size_t k = 0; k < fold_word.size(); ++k, ++J.n) {
            J.slots[J.n] = reinterpret_cast<unsigned long long*>(c->d_part + (size_t)fold_word[k] * CMFP_RED_BLOCKS);
            J.out[J.n] = c->d_red + fold_word[k];
            J.is_max[J.n] = fold_max[k];
            J.mode[J.n] = fold_mode[k];
        }
        c->fold_pending = true;
    } else {
        launch_jobs(fold_word, fold_max, &fold_mode);
    }
    for (size_t k = 0; k < fold_word.size(); ++k) {
        c->word_clean[fold_word[k]] = 1;
        if (fold_mode[k] == 0) c->red_valid[fold_word[k]] = 1;
    }
    for (size_t q = 0; q < list.size(); ++q) std::swap(c->states[list[q]].new_data, c->states[list[q]].alt);
    if (epi_q < 0 && src >= 0) {
        SourceArgs a = source_args(c, c->actions[src], true);
        if (coll >= 0) a.coll = coll_args(c, c->actions[coll]);
        ProfScope prof(c, CMFP_PROF_SOURCE);
        // last kernel of the step: it also leaves the CFL reductions of what it wrote
        c->L->source_fused(c->g, a, 0.5 * dt, 1, c->stream);
        const cmfp_action_desc& A = c->actions[src];
        for (int n = 0; n < A.n_states - 1; ++n) {
            c->red_valid[A.states[n]] = 1;
            c->word_clean[A.states[n]] = 0;  // the source kernel leaves block partials in its words
        }
        c->word_clean[c->states.size() + src] = 0;
        if (a.limits_dt) c->red_valid[c->states.size() + src] = 1;
    }
    return any_gather ? 2 : 1;
}

// ---- cmfp_advance_steps_static: the device words of cmfp_advance_steps (time step + dt log) in the library
// image instead of an allocation.  One copy of the array per GPU, each at its own address: the name
// `scratch` is a host-side stand-in (never passed to a kernel or a memcpy); the address on a device is
// resolved once with that device current and kept in g_scratch_ptr.
static __device__ __align__(256) double scratch[CMFP_STATIC_MAX_STEPS + 1];
constexpr int kMaxDevices = 64;
double* g_scratch_ptr[kMaxDevices] = {};
std::mutex g_scratch_mutex;

// device address of `scratch` on the context's device (which the caller has made current)
int static_scratch(cmfp_context* c, double** out) {
    if (c->device < 0 || c->device >= kMaxDevices) return fail(c, CMFP_ERR_INVALID, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[c->device]) {
        double* p = nullptr;
        HIPCHK(c, hipGetSymbolAddress((void**)&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(c, CMFP_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address");
        g_scratch_ptr[c->device] = p;
    }
    *out = g_scratch_ptr[c->device];
    return CMFP_OK;
}

int check_ctx(cmfp_context* c) {
    if (!c) return CMFP_ERR_INVALID;
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) return fail(c, CMFP_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return CMFP_OK;
}

int sync_and_check(cmfp_context* c) {
    int h_err2[2] = {0, 0};
    HIPCHK(c, hipMemcpyAsync(h_err2, c->d_err, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_err2[1]) {
        // a bounded wait of the device-side transport expired: a neighbour's message (or dt word) never arrived.
        // Whatever was computed since is meaningless; the words are cleared so that the context can be torn down.
        (void)hipMemsetAsync(c->d_err, 0, 2 * sizeof(int), c->stream);
        return fail(c, CMFP_ERR_HIP, "peer transport: a device-side wait for a neighbour rank timed out (" +
                                         std::to_string(c->peer.timeout_s) + " s); results since the last check are invalid");
    }
    const int h_err = h_err2[0];
    if (!h_err) return CMFP_OK;
    // Post-mortem: the first cell (state order, then memory order) whose cell-centred values are invalid now.  The
    // kernels flag invalid *face* states too (after reconstruction or a transverse correction); those leave no
    // invalid cell behind and are reported without a location.
    std::string where = " (no cell-centred value is invalid: a face state failed after reconstruction or transverse correction)";
    int bad_state = -1, bad_cell[3] = {-1, -1, -1};
    for (size_t s = 0; s < c->states.size() && bad_state < 0; ++s) {
        if (c->states[s].sp.kind != CMFP_STATE_HYDRO) continue;
        unsigned long long h = ~0ull;
        if (hipMemsetAsync(c->d_errcell, 0xff, sizeof(h), c->stream) != hipSuccess) break;
        c->L->first_invalid(c->g, c->states[s].sp, c->states[s].new_data, c->d_errcell, c->stream);
        if (hipMemcpyAsync(&h, c->d_errcell, sizeof(h), hipMemcpyDeviceToHost, c->stream) != hipSuccess) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) break;
        if (h == ~0ull) continue;
        bad_state = (int)s;
        bad_cell[0] = (int)(h % (unsigned long long)c->g.n[0]);
        bad_cell[1] = (int)((h / (unsigned long long)c->g.n[0]) % (unsigned long long)c->g.n[1]);
        bad_cell[2] = (int)(h / ((unsigned long long)c->g.n[0] * (unsigned long long)c->g.n[1]));
        where = " first offending cell: state " + std::to_string(s) + ", cell (" + std::to_string(bad_cell[0]) + ", " +
                std::to_string(bad_cell[1]) + ", " + std::to_string(bad_cell[2]) + ") of this block";
    }
    (void)hipMemsetAsync(c->d_err, 0, sizeof(int), c->stream);  // the flag is per failure, not sticky
    const int rc = fail(c, CMFP_ERR_PHYSICS, "Primitive values outside of physical bounds!!" + where);
    c->err_state = bad_state;
    for (int d = 0; d < 3; ++d) c->err_cell[d] = bad_cell[d];
    return rc;
}

}  // namespace

extern "C" {

const char* cmfp_version(void) { return "cerberus_mfp 0.1 (gfx950, strict+fast)"; }

int cmfp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int cmfp_create(const cmfp_geometry* geom, const cmfp_params* params, int n_states, const cmfp_state_desc* states,
                int n_actions, const cmfp_action_desc* actions, int device, void* stream, cmfp_context** out) {
    if (!geom || !params || !states || !out || n_states < 1) return CMFP_ERR_INVALID;
    if (geom->dim < 1 || geom->dim > 3) return CMFP_ERR_INVALID;
    if (cmfp_device_count() <= device) return CMFP_ERR_NODEVICE;
    cmfp_context* c = new cmfp_context();
    *out = c;  // returned even on failure so that cmfp_last_error can be read; caller destroys it
    c->geom = *geom;
    c->params = *params;
    c->device = device;
    c->stream = (hipStream_t)stream;
    if (const char* e = std::getenv("CMFP_SRC_EPILOGUE")) c->src_epilogue = std::atoi(e) != 0;
    if (const char* e = std::getenv("CMFP_SRC_PROLOGUE")) c->src_prologue = std::atoi(e) != 0;
    if (const char* e = std::getenv("CMFP_MARCH_WRAP")) c->march_wrap = std::atoi(e);  // 0: never; 1: where it applies; 2 / 4: x / y only (diagnostic)
    c->L = params->fp_mode == CMFP_FP_FAST ? fast::launchers() : strict::launchers();
    c->nsteps = params->time_integrator == CMFP_TI_EULER ? 1 : 2;  // MFP_config.cpp:73-81
    HIPCHK(c, hipSetDevice(device));

    int ng = 0;
    for (int s = 0; s < n_states; ++s) {
        int n = recon_num_grow(states[s].reconstruction);
        if (n < 0) return fail(c, CMFP_ERR_INVALID, "unknown reconstruction");
        if (states[s].kind == CMFP_STATE_HYDRO && states[s].viscosity) n = std::max(n, 2);  // MFP_hydro.cpp:108-110
        ng = std::max(ng, n);
    }
    Geo& g = c->g;
    g.dim = geom->dim;
    for (int d = 0; d < 3; ++d) {
        const bool on = d < geom->dim;
        g.n[d] = on ? geom->block_n[d] : 1;
        g.ng[d] = on ? ng : 0;
        g.ext[d] = g.n[d] + 2 * g.ng[d];
        g.dx[d] = on ? geom->dx[d] : 1.0;
        if (on && (geom->block_n[d] < ng || geom->block_lo[d] < 0 || geom->block_lo[d] + geom->block_n[d] > geom->n_cell[d]))
            return fail(c, CMFP_ERR_INVALID, "block does not fit the domain or is thinner than the ghost width");
        c->split[d] = on && geom->block_n[d] != geom->n_cell[d];
        c->touches[d][0] = on && !geom->is_periodic[d] && geom->block_lo[d] == 0;
        c->touches[d][1] = on && !geom->is_periodic[d] && geom->block_lo[d] + geom->block_n[d] == geom->n_cell[d];
    }
    g.bc_instance = 0;
    if (params->fp_mode == CMFP_FP_FAST)
        for (int d = 0; d < geom->dim; ++d)
            if (!geom->is_periodic[d]) g.bc_instance = 1;  // the same kernel instances on every rank of this domain (Geo)
    g.sj = g.ext[0];
    g.sk = (long)g.ext[0] * g.ext[1];
    g.sc = g.sk * g.ext[2];
    // kernels address inside a component plane with 32-bit byte offsets
    if (g.sc * (long)sizeof(double) >= (1L << 32))
        return fail(c, CMFP_ERR_INVALID, "block too large: one ghosted component plane must stay below 4 GiB (2^29 cells)");

    c->states.resize(n_states);
    for (int s = 0; s < n_states; ++s) {
        StateRT& st = c->states[s];
        st.desc = states[s];
        StatePhys& sp = st.sp;
        std::memset(&sp, 0, sizeof(sp));
        sp.kind = states[s].kind;
        sp.recon = states[s].reconstruction;
        sp.flux = states[s].flux;
        sp.ng = recon_num_grow(sp.recon);
        if (sp.kind == CMFP_STATE_HYDRO && states[s].viscosity) {
            if (states[s].viscosity < CMFP_VISC_SUTHERLAND || states[s].viscosity > CMFP_VISC_USER)
                return fail(c, CMFP_ERR_INVALID, "Invalid viscosity option");
            if (geom->dim > 2)
                return fail(c, CMFP_ERR_INVALID, "viscous fluxes: 1-D and 2-D only (the reference's 3-D branch does not compile)");
            sp.ng = std::max(sp.ng, 2);
            sp.visc = ViscP{states[s].viscosity, states[s].visc_mu0, states[s].visc_T0, states[s].visc_S, states[s].visc_n,
                            states[s].visc_Pr};
        }
        if (sp.kind == CMFP_STATE_HYDRO) {
            if (states[s].n_tracers < 0 || states[s].n_tracers > CMFP_MAX_TRACERS)
                return fail(c, CMFP_ERR_INVALID, "n_tracers must be 0..2");
            if (states[s].n_tracers && sp.flux == CMFP_FLUX_HYBRID_HLL)
                return fail(c, CMFP_ERR_INVALID, "HLLE/HLLC with tracers: the reference leaves the blended tracer fluxes undefined (MFP_hybrid_hll.cpp:43-45)");
            sp.nc = sp.nq = 5 + states[s].n_tracers;
            sp.hp.nt = states[s].n_tracers;
            if (sp.flux != CMFP_FLUX_HLLC && sp.flux != CMFP_FLUX_HLLE && sp.flux != CMFP_FLUX_HYBRID_HLL)
                return fail(c, CMFP_ERR_INVALID, "hydro flux must be HLLC, HLLE or HLLE/HLLC");
            if (!(states[s].mass != 0.0) || !(states[s].gamma != 0.0) || states[s].gamma == 1.0)
                return fail(c, CMFP_ERR_INVALID, "mass and gamma cannot be 0");  // MFP_thermally_perfect_gas.cpp:29-30
            sp.hp.gas = GasP{states[s].mass, states[s].charge, states[s].gamma};
            sp.hp.shock_threshold = states[s].shock_threshold;
            sp.hp.mix = 0;
            if (states[s].mixture) {
                // species of different mass / gamma (mass_const / gamma_const false, MFP_thermally_perfect_gas.cpp:32-34)
                const int nsp = states[s].n_tracers + 1;
                for (int i = 0; i < nsp; ++i) {
                    const double m = states[s].species_mass[i], gm = states[s].species_gamma[i];
                    if (!(m != 0.0) || !(gm != 0.0) || gm == 1.0) return fail(c, CMFP_ERR_INVALID, "mass and gamma cannot be 0");
                    sp.hp.sp_mass[i] = m;
                    sp.hp.sp_cp[i] = gm / (m * (gm - 1.0));  // cpi, cvi of get_gamma_from_prim (:69-70)
                    sp.hp.sp_cv[i] = 1.0 / (m * (gm - 1.0));
                    if (gm != states[s].species_gamma[0]) sp.hp.mix |= 1;
                    if (m != states[s].species_mass[0]) sp.hp.mix |= 2;
                }
                if (states[s].species_mass[0] != states[s].mass || states[s].species_gamma[0] != states[s].gamma ||
                    states[s].species_charge[0] != states[s].charge)
                    return fail(c, CMFP_ERR_INVALID, "mixture: mass / charge / gamma must be those of species 0");
                if (sp.hp.mix) sp.nq = sp.nc + 1;  // the face states' Gamma entry travels with the primitives
            }
        } else if (sp.kind == CMFP_STATE_FIELD) {
            sp.nc = sp.nq = 10;
            if (sp.flux != CMFP_FLUX_FIELD_RH) return fail(c, CMFP_ERR_INVALID, "field flux must be RankineHugoniot");
            sp.fp.c0 = params->lightspeed;
            sp.fp.ch = states[s].div_transport * params->lightspeed;  // MFP_field.cpp:382-386
            sp.fp.fastest = std::max(params->lightspeed, sp.fp.ch);
        } else {
            return fail(c, CMFP_ERR_INVALID, "unknown state kind");
        }
        std::memcpy(sp.bc, states[s].bc, sizeof(sp.bc));
        int r;
        if ((r = dalloc(c, &st.new_data, sp.nc * g.sc))) return r;
        HIPCHK(c, hipMemsetAsync(st.new_data, 0, sizeof(double) * sp.nc * g.sc, c->stream));
    }
    c->actions.assign(actions, actions + n_actions);
    for (const auto& A : c->actions) {
        if (A.n_states < 1 || A.n_states > CMFP_MAX_ACTION_STATES) return fail(c, CMFP_ERR_INVALID, "bad action state count");
        for (int i = 0; i < A.n_states; ++i)
            if (A.states[i] < 0 || A.states[i] >= n_states) return fail(c, CMFP_ERR_INVALID, "bad action state index");
        if (A.kind == CMFP_ACTION_CTU) continue;
        if (A.kind == CMFP_ACTION_COLLISIONS_RAMBO) {
            for (int i = 0; i < A.n_states; ++i)  // SpeciesInfo::alpha_idx: a sub-component of the species' own state
                if (A.species_component[i] < 0 || A.species_component[i] > c->states[A.states[i]].desc.n_tracers)
                    return fail(c, CMFP_ERR_INVALID, "collisions_rambo: species_component outside 0 .. n_tracers of its state");
            if (A.n_states > 4) return fail(c, CMFP_ERR_INVALID, "at most 4 species per collisions action");
            for (int i = 0; i < A.n_states; ++i)
                if (c->states[A.states[i]].sp.kind != CMFP_STATE_HYDRO)  // MFP_collisions.cpp:35
                    return fail(c, CMFP_ERR_INVALID, "An invalid state has been defined for the Collisions source");
            if (!(params->n0 > 0.0)) return fail(c, CMFP_ERR_INVALID, "collisions_rambo needs params.n0 > 0");
            continue;
        }
        if (A.kind != CMFP_ACTION_PLASMA5 && A.kind != CMFP_ACTION_LORENTZ) return fail(c, CMFP_ERR_INVALID, "unknown action kind");
        if (A.n_states - 1 > 4) return fail(c, CMFP_ERR_INVALID, "at most 4 species per source action");
        if (c->states[A.states[A.n_states - 1]].sp.kind != CMFP_STATE_FIELD)
            return fail(c, CMFP_ERR_INVALID, "source action: last state must be the field");
        for (int i = 0; i < A.n_states - 1; ++i)
            if (c->states[A.states[i]].sp.kind != CMFP_STATE_HYDRO)
                return fail(c, CMFP_ERR_INVALID, "source action: species must be hydro states");
    }
    for (auto& st : c->states) st.fill_ng = st.sp.ng;
    // Where the source prologue may run (decided per step in advance_strang_fused; the conditions that do not
    // change are checked here) the field's kernel evaluates stage A from the species' values over its own stencil
    // reach: their ghost layers are kept current to the field's width
    if (c->src_prologue && geom->dim == 2 && params->time_integrator == CMFP_TI_STRANG)
        for (const auto& A : c->actions)
            if ((A.kind == CMFP_ACTION_LORENTZ || (A.kind == CMFP_ACTION_PLASMA5 && A.solver != CMFP_SOLVER_IMPLICIT)) && A.n_states == 3)
                for (int i = 0; i < 2; ++i) {
                    StateRT& st = c->states[A.states[i]];
                    st.fill_ng = std::max(st.fill_ng, c->states[A.states[2]].sp.ng);
                }
    HIPCHK(c, hipMalloc((void**)&c->d_err, 2 * sizeof(int)));  // [0]: invalid primitives, [1]: a device-side wait of the peer transport gave up
    HIPCHK(c, hipMalloc((void**)&c->d_errcell, sizeof(unsigned long long)));
    HIPCHK(c, hipMemsetAsync(c->d_err, 0, 2 * sizeof(int), c->stream));
    HIPCHK(c, hipMalloc((void**)&c->d_flag, sizeof(int)));
    HIPCHK(c, hipMalloc((void**)&c->d_red, sizeof(unsigned long long) * (2 * n_states + n_actions + 1)));
    HIPCHK(c, hipMalloc((void**)&c->d_part, sizeof(double) * (size_t)(2 * n_states + n_actions + 1) * CMFP_RED_BLOCKS));
    c->word_clean.assign((size_t)(2 * n_states + n_actions + 1), 0);
    c->red_valid.assign(2 * n_states + n_actions + 1, 0);
    {
        // what the fused Strang step of the default kernel selection allocates (ping-pong partners, flux
        // scratch) is reserved now, so that stepping never allocates: cmfp_advance_steps_static relies on it
        const int r = advance_strang_fused(c, 0.0, true);
        if (r < 0) return r;
        // this device's address of the static scratch array, resolved once per device here (set-up)
        double* p = nullptr;
        int r2;
        if ((r2 = static_scratch(c, &p))) return r2;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return CMFP_OK;
}

int cmfp_destroy(cmfp_context* c) {
    if (!c) return CMFP_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int s = 0; s < CMFP_PROF_NSLOTS; ++s)
        for (auto& pr : c->prof[s]) {
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->stream2) {
        (void)hipStreamSynchronize(c->stream2);
        (void)hipStreamDestroy(c->stream2);
        (void)hipEventDestroy(c->ev_src);
        (void)hipEventDestroy(c->ev_int);
        if (c->ev_hyd) (void)hipEventDestroy(c->ev_hyd);
    }
    for (int r = 0; r < cmfp_peer::MAX_RANKS; ++r)
        if (c->peer.opened[r] && c->peer.remote[r]) (void)hipIpcCloseMemHandle(c->peer.remote[r]);
    if (c->peer.arena) (void)hipFree(c->peer.arena);
    for (double* p : c->allocs) (void)hipFree(p);
    if (c->d_err) (void)hipFree(c->d_err);
    if (c->d_errcell) (void)hipFree(c->d_errcell);
    if (c->d_flag) (void)hipFree(c->d_flag);
    if (c->d_dtbuf) (void)hipFree(c->d_dtbuf);
    cmfp_ell::release(c->ell);
    if (c->d_red) (void)hipFree(c->d_red);
    if (c->d_part) (void)hipFree(c->d_part);
    delete c;
    return CMFP_OK;
}

const char* cmfp_last_error(const cmfp_context* c) { return c ? c->err.c_str() : "null context"; }

int cmfp_last_error_cell(const cmfp_context* c, int* state, int* cell) {
    if (!c || !state || !cell) return CMFP_ERR_INVALID;
    *state = c->err_state;
    for (int d = 0; d < 3; ++d) cell[d] = c->err_cell[d];
    return c->err_state >= 0 ? CMFP_OK : CMFP_ERR_INVALID;
}

int cmfp_state_ncomp(const cmfp_context* c, int s) {
    return (c && s >= 0 && s < (int)c->states.size()) ? c->states[s].sp.nc : CMFP_ERR_INVALID;
}
int cmfp_state_ngrow(const cmfp_context* c, int s) {
    return (c && s >= 0 && s < (int)c->states.size()) ? c->states[s].sp.ng : CMFP_ERR_INVALID;
}

static int copy_state(cmfp_context* c, double* dev, int nc, double* host, bool upload) {
    const Geo& g = c->g;
    c->staging.resize((size_t)nc * g.sc);
    double* stg = c->staging.data();
    const size_t bytes = sizeof(double) * (size_t)nc * g.sc;
    if (!upload) {
        HIPCHK(c, hipMemcpyAsync(stg, dev, bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else {
        std::fill(c->staging.begin(), c->staging.end(), 0.0);
    }
    for (int n = 0; n < nc; ++n)
        for (int k = 0; k < g.n[2]; ++k)
            for (int j = 0; j < g.n[1]; ++j) {
                double* d = stg + g.at(0, j, k) + (size_t)n * g.sc;
                double* h = host + (((size_t)n * g.n[2] + k) * g.n[1] + j) * g.n[0];
                if (upload)
                    std::memcpy(d, h, sizeof(double) * g.n[0]);
                else
                    std::memcpy(h, d, sizeof(double) * g.n[0]);
            }
    if (upload) {
        HIPCHK(c, hipMemcpyAsync(dev, stg, bytes, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    std::vector<double>().swap(c->staging);
    return CMFP_OK;
}

int cmfp_state_upload(cmfp_context* c, int s, const double* host) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (s < 0 || s >= (int)c->states.size() || !host) return fail(c, CMFP_ERR_INVALID, "bad state index / pointer");
    std::fill(c->red_valid.begin(), c->red_valid.end(), 0);
    c->states[s].vac_known = c->states[s].alt_mat = false;
    c->states[s].sp.vacuum = 0;
    return copy_state(c, c->states[s].new_data, c->states[s].sp.nc, const_cast<double*>(host), true);
}
int cmfp_state_download(cmfp_context* c, int s, double* host) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (s < 0 || s >= (int)c->states.size() || !host) return fail(c, CMFP_ERR_INVALID, "bad state index / pointer");
    if ((r = sync_and_check(c))) return r;
    return copy_state(c, c->states[s].new_data, c->states[s].sp.nc, host, false);
}
int cmfp_state_device_view(cmfp_context* c, int s, double** dptr, int* ngrow, long long extent[CMFP_MAX_DIM], int* ncomp) {
    if (!c || s < 0 || s >= (int)c->states.size()) return CMFP_ERR_INVALID;
    if (dptr) *dptr = c->states[s].new_data;
    if (ngrow) *ngrow = c->g.ng[0];
    if (extent)
        for (int d = 0; d < 3; ++d) extent[d] = c->g.ext[d];
    if (ncomp) *ncomp = c->states[s].sp.nc;
    return CMFP_OK;
}

int cmfp_update_begin(cmfp_context* c) {
    int r;
    if ((r = check_ctx(c))) return r;
    for (auto& s : c->states) {
        s.U_status = s.dU_status = Inactive;
        if ((r = ensure_update(c, s))) return r;
    }
    swap_levels(c);
    return CMFP_OK;
}
int cmfp_action_get_data(cmfp_context* c, int a, double) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (a < 0 || a >= (int)c->actions.size()) return fail(c, CMFP_ERR_INVALID, "bad action index");
    return action_get_data(c, a);
}
int cmfp_action_calc_time_derivative(cmfp_context* c, int a, double, double dt) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (a < 0 || a >= (int)c->actions.size()) return fail(c, CMFP_ERR_INVALID, "bad action index");
    return action_calc_time_derivative(c, a, dt);
}
int cmfp_action_calc_spatial_derivative(cmfp_context* c, int a, double, double dt, double) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (a < 0 || a >= (int)c->actions.size()) return fail(c, CMFP_ERR_INVALID, "bad action index");
    return action_calc_spatial_derivative(c, a, dt);
}

int cmfp_update_download(cmfp_context* c, int s, int which, double* host) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (s < 0 || s >= (int)c->states.size() || !host) return fail(c, CMFP_ERR_INVALID, "bad state index / pointer");
    double* p = which == 0 ? c->states[s].U : c->states[s].dU;
    if (!p) return fail(c, CMFP_ERR_INVALID, "UpdateData not defined for this state");
    if ((r = sync_and_check(c))) return r;
    return copy_state(c, p, c->states[s].sp.nc, host, false);
}

static int reduce_words(cmfp_context* c, std::vector<unsigned long long>& h) {
    HIPCHK(c, hipMemcpyAsync(h.data(), c->d_red, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost, c->stream));
    int r;
    if ((r = sync_and_check(c))) return r;
    return CMFP_OK;
}
static double as_double(unsigned long long b) {
    double d;
    std::memcpy(&d, &b, sizeof(d));
    return d;
}

// launches the reduction for state s into word w; hydro only
static int launch_state_dt(cmfp_context* c, int s, int w) {
    c->word_clean[w] = 0;
    c->L->max_speed(c->g, c->states[s].sp, c->states[s].new_data, c->d_part + (size_t)w * CMFP_RED_BLOCKS, c->d_red + w,
                    c->stream);
    return CMFP_OK;
}
static bool state_viscous(const cmfp_context* c, int s) {
    return c->states[s].sp.kind == CMFP_STATE_HYDRO && c->states[s].sp.visc.kind != 0;
}
// word of the viscous stability reduction of state s
static int visc_word(const cmfp_context* c, int s) { return (int)(c->states.size() + c->actions.size()) + s; }
static int launch_visc_dt(cmfp_context* c, int s) {
    const int w = visc_word(c, s);
    c->word_clean[w] = 0;
    c->L->visc_stability(c->g, c->states[s].sp, c->states[s].new_data, c->d_part + (size_t)w * CMFP_RED_BLOCKS, c->d_red + w,
                         c->stream);
    return CMFP_OK;
}
static double state_dt_from(const cmfp_context* c, int s, double max_speed, double visc_max) {
    const StateRT& st = c->states[s];
    const Geo& g = c->g;
    const double sp = st.sp.kind == CMFP_STATE_HYDRO ? max_speed : st.sp.fp.fastest;
    double dt = g.dx[0] / sp;  // MFP_eulerian.cpp:302-303, MFP_field.cpp:63-64
    for (int i = 1; i < g.dim; ++i) dt = std::min(dt, g.dx[i] / sp);
    if (state_viscous(c, s)) {  // HydroState::get_allowed_time_step MFP_hydro.cpp:726-736, MFP_hydro_viscous.cpp:88-96
        double min_dt = std::numeric_limits<double>::max();
        for (int d = 0; d < g.dim; ++d) min_dt = std::min(min_dt, g.dx[d] * g.dx[d] / visc_max);
        dt = std::min(dt, min_dt);
    }
    return dt;
}
static int launch_action_dt(cmfp_context* c, int a, int w) {
    SourceArgs sa = source_args(c, c->actions[a], true);
    c->word_clean[w] = 0;
    c->L->plasma_dt(c->g, sa, c->d_part + (size_t)w * CMFP_RED_BLOCKS, c->d_red + w, c->stream);
    return CMFP_OK;
}
static bool action_limits_dt(const cmfp_action_desc& A) {
    return A.kind == CMFP_ACTION_PLASMA5 && A.solver == CMFP_SOLVER_EXPLICIT;  // MFP_plasma5.cpp:461
}
static double action_dt_from(const cmfp_context* c, int a, double cell_min) {
    const cmfp_action_desc& A = c->actions[a];
    const Geo& g = c->g;
    const StateRT& f = c->states[A.states[A.n_states - 1]];
    double max_speed = c->params.lightspeed;  // MFP_plasma5.cpp:494-498
    max_speed = std::max(max_speed, f.sp.fp.ch);
    double dt = g.dx[0] / max_speed;
    for (int d = 1; d < g.dim; ++d) dt = std::min(dt, g.dx[d] / max_speed);
    return std::min(dt, cell_min);
}

int cmfp_state_get_allowed_time_step(cmfp_context* c, int s, double* dt) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (s < 0 || s >= (int)c->states.size() || !dt) return fail(c, CMFP_ERR_INVALID, "bad state index / pointer");
    const int nw = (int)(2 * c->states.size() + c->actions.size());
    std::vector<unsigned long long> h(nw, 0);
    if (c->states[s].sp.kind == CMFP_STATE_HYDRO) {
        if (!c->red_valid[s] && (r = launch_state_dt(c, s, s))) return r;
        c->red_valid[s] = 1;
        if (state_viscous(c, s) && (r = launch_visc_dt(c, s))) return r;
        if ((r = reduce_words(c, h))) return r;
    }
    *dt = state_dt_from(c, s, as_double(h[s]), as_double(h[visc_word(c, s)]));
    return CMFP_OK;
}
int cmfp_action_get_allowed_time_step(cmfp_context* c, int a, double* dt) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (a < 0 || a >= (int)c->actions.size() || !dt) return fail(c, CMFP_ERR_INVALID, "bad action index / pointer");
    if (!action_limits_dt(c->actions[a])) {
        *dt = std::numeric_limits<double>::max();  // MFP_action.H:87, MFP_plasma5.cpp:461
        return CMFP_OK;
    }
    const int ns0 = (int)c->states.size();
    std::vector<unsigned long long> h(ns0 + c->actions.size(), 0);
    if (!c->red_valid[ns0 + a] && (r = launch_action_dt(c, a, ns0 + a))) return r;
    c->red_valid[ns0 + a] = 1;
    if ((r = reduce_words(c, h))) return r;
    *dt = action_dt_from(c, a, as_double(h[ns0 + a]));
    return CMFP_OK;
}

// MFP::estTimeStep MFP_timestep.cpp:4-21 without the cross-rank ReduceRealMin
int cmfp_est_time_step(cmfp_context* c, double* out) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (!out) return fail(c, CMFP_ERR_INVALID, "null pointer");
    if (c->params.force_dt > 0.0) {  // MFP_timestep.cpp:8
        *out = c->params.force_dt;
        return CMFP_OK;
    }
    const int ns = (int)c->states.size(), na = (int)c->actions.size();
    std::vector<unsigned long long> h(2 * ns + na, 0);
    for (int s = 0; s < ns; ++s) {
        if (c->states[s].sp.kind == CMFP_STATE_HYDRO && !c->red_valid[s]) {
            if ((r = launch_state_dt(c, s, s))) return r;
            c->red_valid[s] = 1;
        }
        if (state_viscous(c, s) && (r = launch_visc_dt(c, s))) return r;
    }
    for (int a = 0; a < na; ++a)
        if (action_limits_dt(c->actions[a]) && !c->red_valid[ns + a]) {
            if ((r = launch_action_dt(c, a, ns + a))) return r;
            c->red_valid[ns + a] = 1;
        }
    if ((r = reduce_words(c, h))) return r;
    double estdt = std::numeric_limits<double>::max();
    for (int s = 0; s < ns; ++s) estdt = std::min(estdt, state_dt_from(c, s, as_double(h[s]), as_double(h[visc_word(c, s)])));
    for (int a = 0; a < na; ++a)
        if (action_limits_dt(c->actions[a])) estdt = std::min(estdt, action_dt_from(c, a, as_double(h[ns + a])));
    estdt *= c->params.cfl;
    *out = estdt;
    return CMFP_OK;
}

double cmfp_compute_new_dt(double est_dt, double old_dt) {
    const double change_max = 1.1;  // MFP_timestep.cpp:89-92
    return std::min(est_dt, change_max * old_dt);
}

// nsteps x {MFP::advance; MFP::computeNewDt} with dt kept on the device (no host round trip per step).
// d_dt: 1 + nsteps doubles of device memory (the time step, then the log of the dt every step used).
static int advance_steps_on(cmfp_context* c, double dt0, int nsteps, double* dts, double* dt_next, double* d_dt) {
    int r;
    bool split_any = false;
    for (int d = 0; d < c->g.dim; ++d) split_any = split_any || c->split[d];
    const int ns = (int)c->states.size(), na = (int)c->actions.size();
    DtPlan plan;
    std::memset(&plan, 0, sizeof(plan));
    plan.dim = c->g.dim;
    for (int d = 0; d < 3; ++d) plan.dx[d] = c->g.dx[d];
    plan.cfl = c->params.cfl;
    plan.force_dt = c->params.force_dt;
    if (ns + na > 24) return 1;
    for (int s = 0; s < ns; ++s) {
        if (state_viscous(c, s)) return 1;
        const int n = plan.n++;
        if (c->states[s].sp.kind == CMFP_STATE_HYDRO) {
            plan.kind[n] = 0;
            plan.word[n] = s;
        } else {
            plan.kind[n] = 1;
            plan.cst[n] = state_dt_from(c, s, 0.0, 0.0);
        }
    }
    for (int a = 0; a < na; ++a)
        if (action_limits_dt(c->actions[a])) {
            const int n = plan.n++;
            plan.kind[n] = 2;
            plan.word[n] = ns + a;
            plan.cst[n] = action_dt_from(c, a, std::numeric_limits<double>::max());
        }
    double* d_log = d_dt + 1;
    // whatever happens below, no later call may find the kernels' dt still redirected to d_dt
    struct DtpGuard {
        cmfp_context* c;
        ~DtpGuard() { c->g.dtp = nullptr; }
    } guard{c};
    HIPCHK(c, hipMemcpyAsync(d_dt, &dt0, sizeof(double), hipMemcpyHostToDevice, c->stream));
    c->g.dtp = d_dt;
    int done = 0, rc = CMFP_OK;
    for (int i = 0; i < nsteps; ++i) {
        std::fill(c->red_valid.begin(), c->red_valid.end(), 0);
        c->defer_fold = true;
        r = advance_strang_fused(c, 1.0);  // dt arguments become factors of *dtp
        c->defer_fold = false;
        if (r <= 0) {
            if (r == 0 && i == 0) rc = 1;  // configuration outside the fused kernels: nothing was advanced
            else if (r == 0) rc = fail(c, CMFP_ERR_INVALID, "fused path became unavailable between steps");
            else rc = r;
            break;
        }
        c->last_fused = r;
        for (int s = 0; s < ns; ++s)
            if (c->states[s].sp.kind == CMFP_STATE_HYDRO && !c->red_valid[s]) {
                launch_state_dt(c, s, s);
                c->red_valid[s] = 1;
            }
        for (int a = 0; a < na; ++a)
            if (action_limits_dt(c->actions[a]) && !c->red_valid[ns + a]) {
                launch_action_dt(c, a, ns + a);
                c->red_valid[ns + a] = 1;
            }
        if (c->fold_pending)
            c->L->fold_new_dt(c->pending_fold, plan, c->d_red, d_dt, d_log, i, c->stream);
        else
            c->L->new_dt(plan, c->d_red, d_dt, d_log, i, c->stream);
        c->fold_pending = false;
        done = i + 1;
        if (split_any && c->peer.enabled) {
            if ((rc = peer_dt_min(c, d_dt))) break;
        } else if (split_any && c->dt_reduce(c->dt_reduce_user, c, d_dt)) {
            rc = fail(c, CMFP_ERR_INVALID, "dt reduce callback failed");
            break;
        }
    }
    c->g.dtp = nullptr;
    // The dt of every completed step and the next one are handed back also when the batch stopped early
    // (dts[done..nsteps) = 0), so that the caller still knows the time the states have reached.
    for (int i = done; i < nsteps; ++i) dts[i] = 0.0;
    *dt_next = dt0;
    hipError_t e = hipSuccess;
    if (done) e = hipMemcpyAsync(dts, d_log, sizeof(double) * (size_t)done, hipMemcpyDeviceToHost, c->stream);
    if (done && e == hipSuccess) e = hipMemcpyAsync(dt_next, d_dt, sizeof(double), hipMemcpyDeviceToHost, c->stream);
    r = sync_and_check(c);
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, CMFP_ERR_HIP, hipGetErrorString(e));
    return r;
}

static int advance_steps_applicable(cmfp_context* c, int nsteps, double* dts, double* dt_next) {
    if (nsteps < 1 || !dts || !dt_next) return fail(c, CMFP_ERR_INVALID, "bad arguments");
    if (!c->fused) return 1;
    bool split_any = false;
    for (int d = 0; d < c->g.dim; ++d) split_any = split_any || c->split[d];
    if (split_any && !c->dt_reduce && !c->peer.enabled) return 1;  // the cross-rank dt minimum needs the peer transport or a callback, else the per-step API
    return CMFP_OK;
}

int cmfp_advance_steps(cmfp_context* c, double, double dt0, int nsteps, double* dts, double* dt_next) {
    int r;
    if ((r = check_ctx(c))) return r;
    if ((r = advance_steps_applicable(c, nsteps, dts, dt_next))) return r;
    if (c->d_dtcap < nsteps + 1) {  // dt word + log, kept with the context and grown on demand
        if (c->d_dtbuf) (void)hipFree(c->d_dtbuf);
        c->d_dtbuf = nullptr;
        c->d_dtcap = 0;
        const int cap = std::max(nsteps + 1, 257);
        HIPCHK(c, hipMalloc((void**)&c->d_dtbuf, sizeof(double) * (size_t)cap));
        c->d_dtcap = cap;
    }
    return advance_steps_on(c, dt0, nsteps, dts, dt_next, c->d_dtbuf);
}

// The same with the dt word and the dt log in the library's file-scope device array: no hipMalloc / hipFree,
// no stream or event creation (include/cerberus_mfp.h).
int cmfp_advance_steps_static(cmfp_context* c, double, double dt0, int nsteps, double* dts, double* dt_next) {
    int r;
    if ((r = check_ctx(c))) return r;
    if ((r = advance_steps_applicable(c, nsteps, dts, dt_next))) return r;
    if (nsteps > CMFP_STATIC_MAX_STEPS)
        return fail(c, CMFP_ERR_INVALID, "cmfp_advance_steps_static: nsteps exceeds CMFP_STATIC_MAX_STEPS (" +
                                             std::to_string(CMFP_STATIC_MAX_STEPS) + "); split the batch");
    double* d_dt = nullptr;
    if ((r = static_scratch(c, &d_dt))) return r;  // resolved by cmfp_create on this device: a table lookup here
    struct Guard {
        cmfp_context* c;
        ~Guard() { c->no_alloc = false; }
    } guard{c};
    c->no_alloc = true;
    return advance_steps_on(c, dt0, nsteps, dts, dt_next, d_dt);
}

int cmfp_state_set_face_values(cmfp_context* c, int s, int dim, int side, int prim, const double* cell_values,
                               const double* face_values) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (s < 0 || s >= (int)c->states.size() || dim < 0 || dim >= c->g.dim || side < 0 || side > 1)
        return fail(c, CMFP_ERR_INVALID, "bad state / face");
    StateRT& st = c->states[s];
    if (prim < 0 || prim >= st.sp.nq) return fail(c, CMFP_ERR_INVALID, "bad primitive index");
    // (a mixture's carried Gamma entry, slot 5 + nt, takes no table: the reference's face data name rho, velocities,
    // p and alpha, and gamma follows alpha)
    if (st.sp.kind == CMFP_STATE_HYDRO && st.sp.hp.mix && prim >= st.sp.nc)
        return fail(c, CMFP_ERR_INVALID, "bad primitive index");
    if (c->geom.is_periodic[dim]) return fail(c, CMFP_ERR_INVALID, "Dirichlet data on a periodic face");
    if ((cell_values == nullptr) != (face_values == nullptr)) return fail(c, CMFP_ERR_INVALID, "cell and face values go together");
    if (!cell_values) {
        // freed with the context (allocs); a removed entry is simply no longer applied
        st.tab_cell[dim][side][prim] = st.tab_face[dim][side][prim] = nullptr;
    } else {
        const long nc = table_box(c->g, dim, side, false).npts(), nf = table_box(c->g, dim, side, true).npts();
        double*& dc = st.tab_cell[dim][side][prim];
        double*& df = st.tab_face[dim][side][prim];
        if (!dc) {
            if ((r = dalloc(c, &dc, nc))) return r;
            if ((r = dalloc(c, &df, nf))) return r;
        }
        HIPCHK(c, hipMemcpyAsync(dc, cell_values, sizeof(double) * nc, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(df, face_values, sizeof(double) * nf, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // the host buffers may be reused on return
    }
    st.has_tables = false;
    for (int d = 0; d < 3; ++d)
        for (int sd = 0; sd < 2; ++sd)
            for (int n = 0; n < CMFP_MAX_COMP; ++n) st.has_tables = st.has_tables || st.tab_cell[d][sd][n];
    return CMFP_OK;
}

int cmfp_state_set_dynamic(cmfp_context* c, int s, int comp, const double* values) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (s < 0 || s >= (int)c->states.size()) return fail(c, CMFP_ERR_INVALID, "bad state");
    StateRT& st = c->states[s];
    if (st.sp.kind != CMFP_STATE_FIELD)  // FieldState::set_udf is the only reader of 'dynamic' (MFP_field.cpp:175-221)
        return fail(c, CMFP_ERR_INVALID, "'dynamic' components: field states only");
    if (comp < 0 || comp >= st.sp.nc) return fail(c, CMFP_ERR_INVALID, "bad component index");
    if (!values) {
        st.dyn[comp] = nullptr;  // (freed with the context)
    } else {
        if ((r = dalloc(c, &st.dyn_buf[comp], c->g.sc))) return r;
        double* d = st.dyn[comp] = st.dyn_buf[comp];
        HIPCHK(c, hipMemcpyAsync(d, values, sizeof(double) * c->g.sc, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // the host buffer may be reused on return
    }
    st.has_dynamic = false;
    for (int n = 0; n < CMFP_MAX_COMP; ++n) st.has_dynamic = st.has_dynamic || st.dyn[n];
    return CMFP_OK;
}

// Elliptic::project_divergence_field MFP_elliptic.cpp:667-782
int cmfp_elliptic_project_bc(cmfp_context* c, int fs, int n_species, const int* species, const int* bc_B, const int* bc_D) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (fs < 0 || fs >= (int)c->states.size() || c->states[fs].sp.kind != CMFP_STATE_FIELD)
        return fail(c, CMFP_ERR_INVALID, "An invalid state has been defined for the 'elliptic' action");  // :66
    if (n_species < 0 || n_species > 8 || (n_species && !species)) return fail(c, CMFP_ERR_INVALID, "bad species list");
    if (c->g.dim < 2) return CMFP_OK;  // "MLNodeLaplacian not implemented for 1D": the reference compiles the body out (:641-650)
    int pb[2][3][2];
    bool any_face = false;
    for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2; ++side) {
            // without a table every face is 'outflow', the default of the field's bc entries (MFP_field.cpp:323-325)
            pb[0][d][side] = bc_B ? bc_B[2 * d + side] : CMFP_PHYSBC_OUTFLOW;
            pb[1][d][side] = bc_D ? bc_D[2 * d + side] : CMFP_PHYSBC_OUTFLOW;
            if (d < c->g.dim && !c->geom.is_periodic[d]) any_face = true;
        }
    cmfp_ell::Grid EB, ED;
    if (const char* why = cmfp_ell::make_grid(c->g, c->geom.is_periodic, c->split, pb[0], EB)) return fail(c, CMFP_ERR_INVALID, why);
    if (const char* why = cmfp_ell::make_grid(c->g, c->geom.is_periodic, c->split, pb[1], ED)) return fail(c, CMFP_ERR_INVALID, why);
    cmfp_ell::ChargeArgs q;
    std::memset(&q, 0, sizeof(q));
    q.ns = n_species;
    for (int n = 0; n < n_species; ++n) {
        if (species[n] < 0 || species[n] >= (int)c->states.size() || c->states[species[n]].sp.kind != CMFP_STATE_HYDRO)
            return fail(c, CMFP_ERR_INVALID, "elliptic: species must be hydro states");
        q.rho[n] = c->states[species[n]].new_data;                                  // Density component
        q.r[n] = c->states[species[n]].desc.charge / c->states[species[n]].desc.mass;
        const cmfp_state_desc& sd = c->states[species[n]].desc;
        if (sd.mixture) {
            SpeciesMix& M = q.mix[n];
            for (int i = 0; i <= sd.n_tracers; ++i) {
                M.m[i] = sd.species_mass[i];
                M.q[i] = sd.species_charge[i];
                if (M.m[i] != M.m[0]) M.flags |= 1;
                if (M.q[i] != M.q[0]) M.flags |= 2;
            }
            M.nt = M.flags ? sd.n_tracers : 0;
        }
    }
    q.sc = c->g.sc;
    double* F = c->states[fs].new_data;
    const long sc = c->g.sc;
    q.ep = F + 9 * sc;
    q.cd_scale = c->params.Larmor / (c->params.Debye * c->params.Debye * c->params.lightspeed);  // :685
    // the cells beyond a physical face enter the nodal divergence: FillPatch of the vector with one ghost layer (:897-904)
    if (any_face && (r = fill_patch(c, fs))) return r;
    HIPCHK(c, cmfp_ell::setup(EB, c->g, c->ell));
    cmfp_ell::project(c->g, EB, c->ell, F + 3 * sc, F + 7 * sc, nullptr, c->stream);  // B with psi   :677-681
    // the second solve's FillPatch sees the projected B only through components it does not read; D is unchanged so far
    HIPCHK(c, cmfp_ell::setup(ED, c->g, c->ell));
    cmfp_ell::project(c->g, ED, c->ell, F + 0 * sc, F + 6 * sc, &q, c->stream);       // D with phi   :775-780
    std::fill(c->red_valid.begin(), c->red_valid.end(), 0);  // B changed: the cyclotron dt limit has to be taken again
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, CMFP_ERR_HIP, std::string("divergence projection launch: ") + hipGetErrorString(e));
    return CMFP_OK;
}
int cmfp_elliptic_project(cmfp_context* c, int fs, int n_species, const int* species) {
    return cmfp_elliptic_project_bc(c, fs, n_species, species, nullptr, nullptr);
}

int cmfp_set_fused(cmfp_context* c, int enable) {
    if (!c) return CMFP_ERR_INVALID;
    c->fused = enable;
    return CMFP_OK;
}

// MFP::advance MFP_advance.cpp:6-56
int cmfp_advance(cmfp_context* c, double, double dt) {
    int r;
    if ((r = check_ctx(c))) return r;
    c->last_fused = 0;
    std::fill(c->red_valid.begin(), c->red_valid.end(), 0);
    if (c->fused) {
        r = advance_strang_fused(c, dt);
        if (r < 0) return r;
        if (r >= 1) {
            c->last_fused = r;  // 1: tiled kernels only, 2: gather kernels used
            return CMFP_OK;
        }
    }
    for (auto& s : c->states)
        if ((r = ensure_update(c, s))) return r;
    swap_levels(c);
    if (c->params.time_integrator == CMFP_TI_STRANG) return advance_strang(c, dt);
    return advance_RK(c, dt);
}

int cmfp_last_advance_fused(const cmfp_context* c) { return c ? c->last_fused : 0; }
int cmfp_last_source_fusion(const cmfp_context* c) { return c && c->last_fused ? (c->last_prologue | (c->last_epilogue << 1)) : 0; }
int cmfp_last_wrapped_dims(const cmfp_context* c) { return c && c->last_fused ? c->last_wrap : 0; }

int cmfp_profile_enable(cmfp_context* c, int on) {
    if (!c) return CMFP_ERR_INVALID;
    c->profiling = on;
    return CMFP_OK;
}
int cmfp_profile_read(cmfp_context* c, int slot, long long* launches, double* total_ms) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (slot < 0 || slot >= CMFP_PROF_NSLOTS || !launches || !total_ms) return fail(c, CMFP_ERR_INVALID, "bad profile slot");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->stream2) HIPCHK(c, hipStreamSynchronize(c->stream2));
    double ms = 0;
    for (auto& pr : c->prof[slot]) {
        float t = 0;
        HIPCHK(c, hipEventElapsedTime(&t, pr.first, pr.second));
        ms += t;
        c->ev_pool.push_back(pr.first);
        c->ev_pool.push_back(pr.second);
    }
    *launches = (long long)c->prof[slot].size() - c->prof_uncounted[slot];
    *total_ms = ms;
    c->prof[slot].clear();
    c->prof_uncounted[slot] = 0;
    return CMFP_OK;
}

int cmfp_set_exchange_callback(cmfp_context* c, cmfp_exchange_fn fn, void* user) {
    if (!c) return CMFP_ERR_INVALID;
    c->exchange = fn;
    c->exchange_user = user;
    return CMFP_OK;
}
int cmfp_set_overlap(cmfp_context* c, int mode) {
    if (!c || mode < 0 || mode > 2) return CMFP_ERR_INVALID;
    c->overlap = mode;
    return CMFP_OK;
}

int cmfp_set_dt_reduce_callback(cmfp_context* c, cmfp_dt_reduce_fn fn, void* user) {
    if (!c) return CMFP_ERR_INVALID;
    c->dt_reduce = fn;
    c->dt_reduce_user = user;
    return CMFP_OK;
}

int cmfp_copy_async(cmfp_context* c, void* dst, const void* src, long long nbytes) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (!dst || !src || nbytes < 0) return fail(c, CMFP_ERR_INVALID, "bad arguments");
    HIPCHK(c, hipMemcpyAsync(dst, src, (size_t)nbytes, hipMemcpyDeviceToDevice, c->stream));
    return CMFP_OK;
}

int cmfp_set_exchange_batched(cmfp_context* c, int on) {
    if (!c) return CMFP_ERR_INVALID;
    c->exchange_batched = on != 0;
    return CMFP_OK;
}

// ---- device-side transport between ranks (include/cerberus_mfp.h, mfp_peer.h)
int cmfp_peer_init(cmfp_context* c, int rank, int world, void* ipc_handle_out, void** arena_out) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (world < 1 || world > cmfp_peer::MAX_RANKS || rank < 0 || rank >= world)
        return fail(c, CMFP_ERR_INVALID, "cmfp_peer_init: rank / world out of range (at most " + std::to_string(cmfp_peer::MAX_RANKS) + " ranks)");
    if (c->peer.arena) return fail(c, CMFP_ERR_INVALID, "cmfp_peer_init: already initialised");
    static_assert(sizeof(hipIpcMemHandle_t) == CMFP_IPC_HANDLE_BYTES, "CMFP_IPC_HANDLE_BYTES");
    auto& P = c->peer;
    P.rank = rank;
    P.world = world;
    if (const char* e = std::getenv("CMFP_PEER_TIMEOUT")) P.timeout_s = std::max(0.05, std::atof(e));
    peer_layout(c);
    // Fine-grained device memory: the progress words and slots are written by kernels of other processes / devices
    // while this device's kernels poll them, which ordinary (coarse-grained) allocations only promise to make
    // visible at kernel boundaries.  CMFP_PEER_COARSE=1 (or a runtime without the flag): plain hipMalloc.
    static const bool coarse = std::getenv("CMFP_PEER_COARSE") && std::atoi(std::getenv("CMFP_PEER_COARSE")) != 0;
    if (coarse || hipExtMallocWithFlags((void**)&P.arena, P.L.bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        P.arena = nullptr;
        HIPCHK(c, hipMalloc((void**)&P.arena, P.L.bytes));
        P.fine = false;
    } else {
        P.fine = true;
    }
    if (std::getenv("CMFP_PEER_DEBUG"))
        std::fprintf(stderr, "cmfp_peer_init: rank %d of %d, arena %zu bytes, %s memory\n", rank, world, (size_t)P.L.bytes,
                     P.fine ? "fine-grained" : "coarse-grained");
    HIPCHK(c, hipMemsetAsync(P.arena, 0, P.L.bytes, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the words are zero before any neighbour can learn the address
    P.remote[rank] = P.arena;
    if (ipc_handle_out) {
        hipIpcMemHandle_t h;
        HIPCHK(c, hipIpcGetMemHandle(&h, P.arena));
        std::memcpy(ipc_handle_out, &h, sizeof(h));
    }
    if (arena_out) *arena_out = P.arena;
    return CMFP_OK;
}
int cmfp_peer_open(cmfp_context* c, int peer_rank, const void* ipc_handle, void* same_process_arena) {
    int r;
    if ((r = check_ctx(c))) return r;
    auto& P = c->peer;
    if (!P.arena) return fail(c, CMFP_ERR_INVALID, "cmfp_peer_open: call cmfp_peer_init first");
    if (peer_rank < 0 || peer_rank >= P.world || peer_rank == P.rank) return fail(c, CMFP_ERR_INVALID, "cmfp_peer_open: bad peer rank");
    if (P.remote[peer_rank]) return fail(c, CMFP_ERR_INVALID, "cmfp_peer_open: peer already attached");
    if (same_process_arena) {
        // a context of this process: its arena pointer is valid here (the caller has enabled peer access where
        // the two contexts sit on different devices)
        P.remote[peer_rank] = static_cast<char*>(same_process_arena);
        return CMFP_OK;
    }
    if (!ipc_handle) return fail(c, CMFP_ERR_INVALID, "cmfp_peer_open: no handle");
    hipIpcMemHandle_t h;
    std::memcpy(&h, ipc_handle, sizeof(h));
    void* p = nullptr;
    HIPCHK(c, hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    P.remote[peer_rank] = static_cast<char*>(p);
    P.opened[peer_rank] = true;
    return CMFP_OK;
}
int cmfp_peer_set_neighbour(cmfp_context* c, int dim, int side, int rank) {
    if (!c) return CMFP_ERR_INVALID;
    if (dim < 0 || dim >= c->g.dim || side < 0 || side > 1 || rank < -1 || rank >= c->peer.world)
        return fail(c, CMFP_ERR_INVALID, "cmfp_peer_set_neighbour: bad face or rank");
    c->peer.nb[dim][side] = rank;
    return CMFP_OK;
}
int cmfp_peer_enable(cmfp_context* c, int on) {
    if (!c) return CMFP_ERR_INVALID;
    auto& P = c->peer;
    if (!on) {
        P.enabled = false;
        return CMFP_OK;
    }
    if (!P.arena) return fail(c, CMFP_ERR_INVALID, "cmfp_peer_enable: call cmfp_peer_init first");
    for (int r = 0; r < P.world; ++r)
        if (!P.remote[r]) return fail(c, CMFP_ERR_INVALID, "cmfp_peer_enable: rank " + std::to_string(r) + " is not attached (the dt minimum writes to every rank)");
    for (int d = 0; d < c->g.dim; ++d)
        for (int side = 0; side < 2; ++side) {
            const bool needs = c->split[d] && !c->touches[d][side];
            if (needs && P.nb[d][side] < 0)
                return fail(c, CMFP_ERR_INVALID, "cmfp_peer_enable: face (" + std::to_string(d) + ", " + std::to_string(side) + ") has no neighbour rank");
            if (!needs && P.nb[d][side] >= 0)
                return fail(c, CMFP_ERR_INVALID, "cmfp_peer_enable: a neighbour is set on a face that is not an inter-rank face");
        }
    P.enabled = true;
    return CMFP_OK;
}
int cmfp_peer_enabled(const cmfp_context* c) { return c && c->peer.enabled ? 1 : 0; }

int cmfp_fill_patch(cmfp_context* c, int s) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (s < 0 || s >= (int)c->states.size()) return fail(c, CMFP_ERR_INVALID, "bad state index");
    return fill_patch(c, s);
}
long long cmfp_halo_count(const cmfp_context* c, int s, int dim) {
    if (!c || s < 0 || s >= (int)c->states.size() || dim < 0 || dim >= c->g.dim) return CMFP_ERR_INVALID;
    long long n = (long long)c->states[s].sp.nc * c->states[s].fill_ng;
    for (int t = 0; t < 3; ++t)
        if (t != dim) n *= c->g.ext[t];
    return n;
}
int cmfp_halo_pack(cmfp_context* c, int s, int dim, int side, double* dbuf) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (s < 0 || s >= (int)c->states.size() || dim < 0 || dim >= c->g.dim || !dbuf)
        return fail(c, CMFP_ERR_INVALID, "bad halo arguments");
    StateRT& st = c->states[s];
    c->L->pack(c->g, st.sp.nc, st.fill_ng, st.new_data, dim, side, dbuf, 0, c->stream);
    return CMFP_OK;
}
int cmfp_halo_unpack(cmfp_context* c, int s, int dim, int side, const double* dbuf) {
    int r;
    if ((r = check_ctx(c))) return r;
    if (s < 0 || s >= (int)c->states.size() || dim < 0 || dim >= c->g.dim || !dbuf)
        return fail(c, CMFP_ERR_INVALID, "bad halo arguments");
    StateRT& st = c->states[s];
    c->L->pack(c->g, st.sp.nc, st.fill_ng, st.new_data, dim, side, const_cast<double*>(dbuf), 1, c->stream);
    return CMFP_OK;
}

static const Launchers* mode(int fp_mode) { return fp_mode == CMFP_FP_FAST ? fast::launchers() : strict::launchers(); }

int cmfp_leaf_reconstruct(int recon, int fp_mode, long long n, const double* stencil, double* slope, double* lo,
                          double* hi, void* stream) {
    if (cmfp_device_count() < 1) return CMFP_ERR_NODEVICE;
    if (n <= 0 || !stencil || !slope || !lo || !hi) return CMFP_ERR_INVALID;
    return mode(fp_mode)->leaf_reconstruct(recon, (long)n, stencil, slope, lo, hi, (hipStream_t)stream) ? CMFP_OK
                                                                                                        : CMFP_ERR_INVALID;
}
int cmfp_leaf_riemann(int flux, int fp_mode, long long n, const double* L, const double* R, double* F, double c0,
                      double ch, double shock_threshold, void* stream) {
    if (cmfp_device_count() < 1) return CMFP_ERR_NODEVICE;
    if (n <= 0 || !L || !R || !F) return CMFP_ERR_INVALID;
    return mode(fp_mode)->leaf_riemann(flux, (long)n, L, R, F, c0, ch, shock_threshold, (hipStream_t)stream) ? CMFP_OK
                                                                                                              : CMFP_ERR_INVALID;
}
int cmfp_leaf_cons2prim(int fp_mode, long long n, double mass, double gamma, const double* U, double* Q, double* speeds,
                        void* stream) {
    if (cmfp_device_count() < 1) return CMFP_ERR_NODEVICE;
    if (n <= 0 || !U || !Q || !speeds) return CMFP_ERR_INVALID;
    return mode(fp_mode)->leaf_cons2prim((long)n, mass, gamma, U, Q, speeds, (hipStream_t)stream) ? CMFP_OK : CMFP_ERR_INVALID;
}
int cmfp_leaf_prim2cons(int fp_mode, long long n, double mass, double gamma, const double* Q, double* U, void* stream) {
    if (cmfp_device_count() < 1) return CMFP_ERR_NODEVICE;
    if (n <= 0 || !Q || !U) return CMFP_ERR_INVALID;
    return mode(fp_mode)->leaf_prim2cons((long)n, mass, gamma, Q, U, (hipStream_t)stream) ? CMFP_OK : CMFP_ERR_INVALID;
}

}  // extern "C"
