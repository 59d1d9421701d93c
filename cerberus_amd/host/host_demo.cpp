// host_demo.cpp — the two-fluid Orszag-Tang problem driven through the C++
// State/Action mirror (MFP_host.H) on one GPU; prints the dt sequence and a
// checksum per state so that tests can compare it with the Python driver.
// `host_demo N STEPS ranks`: the same level cut into two blocks (rows), one context and one host thread per
// block on the same GPU, ghost layers and the dt minimum moved by the library's device-side transport
// (cmfp_peer_*, arenas passed by pointer: the one-process form of INTEGRATION.md 3a).
#include "MFP_host.H"

#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>

using namespace mfp;

// the two HIP runtime calls the ranks mode needs (a stream per context: two contexts that wait for each other's
// messages must not share one queue); declared here so that the demo builds with a plain C++ compiler
extern "C" int hipStreamCreateWithFlags(void** stream, unsigned flags);
extern "C" int hipStreamDestroy(void* stream);

namespace {
struct Barrier {  // ParallelDescriptor::Barrier of the two host threads
    std::mutex m;
    std::condition_variable cv;
    int n, waiting = 0, phase = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(m);
        const int ph = phase;
        if (++waiting == n) {
            waiting = 0;
            ++phase;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return phase != ph; });
        }
    }
};
}  // namespace

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 32;
    const int steps = argc > 2 ? std::atoi(argv[2]) : 5;
    const bool ranks = argc > 3 && std::string(argv[3]) == "ranks";
    const bool batch = (argc > 3 && std::string(argv[3]) == "batch") || ranks;
    const bool project = argc > 3 && std::string(argv[3]) == "elliptic";  // + the divergence projection after every step
    // the ions as a mixture of two kinds (mass = {1, 3}, charge = {1, 2}, gamma = {5/3, 1.4}) with a varying mass fraction
    const bool mixture = argc > 3 && std::string(argv[3]) == "mixture";
    const double mix_m[2] = {1.0, 3.0}, mix_g[2] = {5.0 / 3.0, 1.4};
    const double gam = 5.0 / 3.0, c = 20.0, PI = std::acos(-1.0), L = 2 * PI;
    // configuration + initial data of one block: rows [j0, j0 + nj) of the level
    auto set_up = [&](MFP& level, int j0, int nj, void* stream) {
        Globals& G = level.G;
        G.dim = 2;
        G.n_cell[0] = G.n_cell[1] = n;
        G.prob_hi[0] = G.prob_hi[1] = L;
        G.is_periodic[0] = G.is_periodic[1] = 1;
        G.cfl = 0.3;
        G.lightspeed = c;
        G.skin_depth = 0.2;
        G.beta = 1.0;
        std::map<std::string, StateDef> S;
        for (const char* nm : {"ion", "electron"}) {
            StateDef d;
            d.type = "hydro";
            d.reconstruction = "minmod";
            d.flux = "HLLC";
            d.gamma = gam;
            d.mass = std::string(nm) == "ion" ? 1.0 : 1.0 / 25.0;
            d.charge = std::string(nm) == "ion" ? 1.0 : -1.0;
            if (mixture && std::string(nm) == "ion") {
                d.n_tracers = 1;
                d.species_mass = {mix_m[0], mix_m[1]};
                d.species_charge = {1.0, 2.0};
                d.species_gamma = {mix_g[0], mix_g[1]};
            }
            S[nm] = d;
        }
        StateDef f;
        f.type = "field";
        f.reconstruction = "O6";
        f.flux = "RankineHugoniot";
        f.div_transport = 1.0;
        f.div_damping = 1.0;
        S["field"] = f;
        std::map<std::string, ActionDef> A;
        A["em_fluxes"].type = "CTU";
        A["em_fluxes"].states = {"field"};
        A["hydro_fluxes"].type = "CTU";
        A["hydro_fluxes"].states = {"ion", "electron"};
        A["plasma"].type = "plasma5";
        A["plasma"].solver = "explicit";
        A["plasma"].states = {"ion", "electron", "field"};
        if (project) {
            A["divergence_cleaning"].type = "elliptic";
            A["divergence_cleaning"].projection = 1;
            A["divergence_cleaning"].state = "field";
        }
        if (nj != n) {
            G.block_lo[0] = 0;
            G.block_lo[1] = j0;
            G.block_n[0] = n;
            G.block_n[1] = nj;
        }
        level.read_config(S, A, 0, stream);

        // initial data (HydroState::init_data MFP_hydro.cpp:504-567): cell centres, prim2cons
        const double dx = L / n, nd = gam * gam;
        for (size_t s = 0; s < level.states.size(); ++s) {
            EulerianState& st = level.eulerian(s);
            const int nc = (int)st.n_cons();
            std::vector<double> U((size_t)nc * nj * n, 0.0);
            for (int j = 0; j < nj; ++j)
                for (int i = 0; i < n; ++i) {
                    const double x = 0.0 + (i + 0.5) * dx, y = 0.0 + (j0 + j + 0.5) * dx;
                    const double ux = -std::sin(y), uy = std::sin(x), Bx = -std::sin(y), By = std::sin(2 * x);
                    auto at = [&](int comp) -> double& { return U[((size_t)comp * nj + j) * n + i]; };
                    if (st.get_type() == State::StateType::Hydro) {
                        const double rho = nd * st.def.mass, p = gam / 2;
                        at(0) = rho;
                        at(1) = ux * rho;
                        at(2) = uy * rho;
                        at(3) = 0.0 * rho;
                        double g = gam;
                        if (st.def.n_tracers) {  // prim2cons with get_gamma_from_prim (MFP_thermally_perfect_gas.cpp:51-83, 234)
                            const double alpha = 0.5 + 0.3 * std::sin(x) * std::cos(y);
                            const double a = alpha < 0.0 ? 0.0 : (alpha > 1.0 ? 1.0 : alpha);
                            double s_cp = 0.0, s_cv = 0.0, s_a = 0.0;
                            s_a = s_a + a;
                            s_cp = s_cp + a * (mix_g[0] / (mix_m[0] * (mix_g[0] - 1.0)));
                            s_cv = s_cv + a * (1.0 / (mix_m[0] * (mix_g[0] - 1.0)));
                            s_cp = s_cp + (1.0 - s_a) * (mix_g[1] / (mix_m[1] * (mix_g[1] - 1.0)));
                            s_cv = s_cv + (1.0 - s_a) * (1.0 / (mix_m[1] * (mix_g[1] - 1.0)));
                            g = s_cp / s_cv;
                            at(5) = alpha * rho;
                        }
                        at(4) = p / (g - 1) + 0.5 * rho * (ux * ux + uy * uy + 0.0 * 0.0);
                    } else {
                        at(2) = -(ux * By - uy * Bx) / c;
                        at(3) = Bx;
                        at(4) = By;
                        at(8) = 1.0;
                        at(9) = 1.0;
                    }
                }
            level.check(cmfp_state_upload(level.ctx, (int)s, U.data()));
        }
    };
    auto print_sums = [&](const MFP& level, const std::vector<std::vector<double>>& data) {
        for (size_t s = 0; s < data.size(); ++s) {
            double sum = 0, sq = 0;
            for (double v : data[s]) {
                sum += v;
                sq += v * v;
            }
            std::printf("state %s sum %.17g sumsq %.17g\n", level.state_names[s].c_str(), sum, sq);
        }
    };
    if (ranks) {
        const int W = 2, nj = n / W;
        std::vector<std::unique_ptr<MFP>> lv(W);
        std::vector<void*> arenas(W, nullptr), streams(W, nullptr);
        std::vector<std::vector<double>> whole;      // the level put together again, [state][comp][j][i]
        std::vector<std::vector<Real>> dts(W);
        std::vector<std::string> err(W);
        Barrier bar(W);
        std::mutex mm;
        Real dt_min[2] = {std::numeric_limits<Real>::max(), std::numeric_limits<Real>::max()};
        int dt_turn = 0;
        auto body = [&](int r) {
            bool ok = true;  // a rank that failed still takes every barrier, so that the other one is not left waiting
            auto guard = [&](const std::function<void()>& f) {
                if (!ok) return;
                try {
                    f();
                } catch (const std::exception& e) {
                    err[(size_t)r] = e.what();
                    ok = false;
                }
            };
            guard([&] {
                if (hipStreamCreateWithFlags(&streams[(size_t)r], 1u /* hipStreamNonBlocking */) != 0) Abort("hipStreamCreate");
                lv[(size_t)r].reset(new MFP);
                set_up(*lv[(size_t)r], r * nj, nj, streams[(size_t)r]);
                arenas[(size_t)r] = lv[(size_t)r]->peer_arena(r, W);
            });
            bar.wait();
            guard([&] {
                for (int q = 0; q < W; ++q)
                    if (!arenas[(size_t)q]) Abort("a rank has no arena");
                // rows are cut: both faces of direction 1 lead to the other rank (periodic); direction 0 is whole
                const int nb[3][2] = {{-1, -1}, {1 - r, 1 - r}, {-1, -1}};
                lv[(size_t)r]->attach_local_peers(arenas, nb);
                lv[(size_t)r]->reduce_min = [&](Real v) {  // ReduceRealMin of the two threads
                    int turn;
                    {
                        std::lock_guard<std::mutex> lk(mm);
                        turn = dt_turn;
                        dt_min[turn] = std::min(dt_min[turn], v);
                    }
                    bar.wait();
                    const Real out = dt_min[turn];
                    bar.wait();
                    if (r == 0) {
                        dt_min[turn] = std::numeric_limits<Real>::max();
                        dt_turn = 1 - turn;
                    }
                    bar.wait();
                    return out;
                };
            });
            bar.wait();  // every arena mapped before the first message
            Real dt = 0;
            if (ok) {
                try {
                    dt = lv[(size_t)r]->estTimeStep();
                } catch (const std::exception& e) {  // (inside the reduction's barriers: the other rank would wait)
                    err[(size_t)r] = e.what();
                    std::fprintf(stderr, "Abort: %s\n", e.what());
                    std::exit(2);
                }
            } else {
                std::fprintf(stderr, "Abort: %s\n", err[(size_t)r].c_str());
                std::exit(2);
            }
            guard([&] { lv[(size_t)r]->advance_steps(0.0, dt, steps, dts[(size_t)r]); });
            bar.wait();
            guard([&] {
                MFP& level = *lv[(size_t)r];
                {
                    std::lock_guard<std::mutex> lk(mm);
                    if (whole.empty()) whole.resize(level.states.size());
                }
                for (size_t s = 0; s < level.states.size(); ++s) {
                    const int nc = (int)level.eulerian(s).n_cons();
                    std::vector<double> U((size_t)nc * nj * n);
                    level.check(cmfp_state_download(level.ctx, (int)s, U.data()));
                    std::lock_guard<std::mutex> lk(mm);
                    if (whole[s].empty()) whole[s].assign((size_t)nc * n * n, 0.0);
                    for (int cpt = 0; cpt < nc; ++cpt)
                        for (int j = 0; j < nj; ++j)
                            for (int i = 0; i < n; ++i)
                                whole[s][((size_t)cpt * n + r * nj + j) * n + i] = U[((size_t)cpt * nj + j) * n + i];
                }
            });
        };
        std::thread t1(body, 1);
        body(0);
        t1.join();
        for (int r = 0; r < W; ++r)
            if (!err[(size_t)r].empty()) {
                std::fprintf(stderr, "Abort (rank %d): %s\n", r, err[(size_t)r].c_str());
                return 2;
            }
        if (dts[0] != dts[1]) {
            std::fprintf(stderr, "the ranks disagree on the dt sequence\n");
            return 3;
        }
        for (Real v : dts[0]) std::printf("dt %.17g\n", v);
        print_sums(*lv[0], whole);
        lv.clear();
        for (void* st : streams)
            if (st) hipStreamDestroy(st);
        return 0;
    }
    try {
        MFP level;
        set_up(level, 0, n, nullptr);
        double t = 0, dt = level.estTimeStep();
        if (batch) {  // the whole loop enqueued at once, dt on the device
            std::vector<Real> dts;
            level.advance_steps(t, dt, steps, dts);
            for (Real v : dts) std::printf("dt %.17g\n", v);
        } else {
            for (int k = 0; k < steps; ++k) {
                level.advance(t, dt);
                t += dt;
                std::printf("dt %.17g\n", dt);
                dt = level.computeNewDt(dt);
            }
        }
        std::vector<std::vector<double>> data(level.states.size());
        for (size_t s = 0; s < level.states.size(); ++s) {
            data[s].resize((size_t)level.eulerian(s).n_cons() * n * n);
            level.check(cmfp_state_download(level.ctx, (int)s, data[s].data()));
        }
        print_sums(level, data);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Abort: %s\n", e.what());
        return 2;
    }
    return 0;
}
