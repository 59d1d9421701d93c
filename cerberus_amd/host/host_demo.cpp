// host_demo.cpp — the two-fluid Orszag-Tang problem driven through the C++
// State/Action mirror (MFP_host.H) on one GPU; prints the dt sequence and a
// checksum per state so that tests can compare it with the Python driver.
#include "MFP_host.H"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

using namespace mfp;

int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 32;
    const int steps = argc > 2 ? std::atoi(argv[2]) : 5;
    const bool batch = argc > 3 && std::string(argv[3]) == "batch";
    const bool project = argc > 3 && std::string(argv[3]) == "elliptic";  // + the divergence projection after every step
    // the ions as a mixture of two kinds (mass = {1, 3}, charge = {1, 2}, gamma = {5/3, 1.4}) with a varying mass fraction
    const bool mixture = argc > 3 && std::string(argv[3]) == "mixture";
    const double mix_m[2] = {1.0, 3.0}, mix_g[2] = {5.0 / 3.0, 1.4};
    const double gam = 5.0 / 3.0, c = 20.0, PI = std::acos(-1.0), L = 2 * PI;
    try {
        MFP level;
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
        level.read_config(S, A);

        // initial data (HydroState::init_data MFP_hydro.cpp:504-567): cell centres, prim2cons
        const double dx = L / n, nd = gam * gam;
        for (size_t s = 0; s < level.states.size(); ++s) {
            EulerianState& st = level.eulerian(s);
            const int nc = (int)st.n_cons();
            std::vector<double> U((size_t)nc * n * n, 0.0);
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) {
                    const double x = 0.0 + (i + 0.5) * dx, y = 0.0 + (j + 0.5) * dx;
                    const double ux = -std::sin(y), uy = std::sin(x), Bx = -std::sin(y), By = std::sin(2 * x);
                    auto at = [&](int comp) -> double& { return U[((size_t)comp * n + j) * n + i]; };
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
        for (size_t s = 0; s < level.states.size(); ++s) {
            const int nc = (int)level.eulerian(s).n_cons();
            std::vector<double> U((size_t)nc * n * n);
            level.check(cmfp_state_download(level.ctx, (int)s, U.data()));
            double sum = 0, sq = 0;
            for (double v : U) {
                sum += v;
                sq += v * v;
            }
            std::printf("state %s sum %.17g sumsq %.17g\n", level.state_names[s].c_str(), sum, sq);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Abort: %s\n", e.what());
        return 2;
    }
    return 0;
}
