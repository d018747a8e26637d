// examples/robot of the reference (examples/robot/main.go) in a compiled host language, call for call, on the MI355X engine
// through include/gokalman_amd.hpp (the C++ mirror of gokalman's interfaces over the C ABI): what the reference's own main.go
// reads like once `gokalman.` resolves to this library.  The Go shim (go/gokalman_amd.go) offers the same names to main.go itself;
// there is no Go toolchain in the build image, so this program is the compiled twin that runs in the tests
// (tests/test_examples_gpu.py compares its files with examples/robot.py's, byte for byte, for the same seed and initial state).
//
//   g++ -std=c++17 -O1 -pthread -Iinclude examples/robot.cpp -Lgokalman_amd -lgokalman_amd -Wl,-rpath,$PWD/gokalman_amd -o robot
//   ./robot [outdir [seed [x0_0 x0_1 [sims]]]]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "gokalman_amd.hpp"

using namespace gokalman;

int main(int argc, char **argv) {
    const std::string outdir = argc > 1 ? argv[1] : ".";
    const unsigned long long seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1ull;
    try {
        const double dt = 0.1;                                                    // main.go:16 Δt := 0.1
        const Matrix F(2, 2, {1, dt, 0, 1});                                      // :17
        const Matrix G(2, 1, {0.5 * dt * dt, dt});                                // :18
        const Matrix H(1, 2, {1, 0});                                             // :19
        const Matrix R(1, 1, {0.05});                                             // :20
        const Matrix Q(2, 2, {5e-2, 5e-4, 5e-4, 1e-3});                           // :22 "Q small"
        const Noise noise = NewAWGN(Q, R, seed);                                  // :24 (the seed replaces the reference's wall clock)
        const Vector x0 = NewVector(2, {0, 0});                                   // :25
        const Matrix P0 = ScaledIdentity(2, 2);                                   // :26
        // :27-30 a random initial state ~ N(0, P0): given on the command line so that the run is reproducible
        const Vector mcX0 = NewVector(2, {argc > 4 ? std::atof(argv[3]) : 0.7, argc > 4 ? std::atof(argv[4]) : -0.3});

        auto mcKF = NewPurePredictorVanilla(mcX0, P0, F, G, H, noise).first;      // :31
        auto chiKF = NewVanilla(x0, P0, F, G, H, NewNoiseless(Q, R)).first;       // :32
        const int steps = 120;                                                    // :33
        const int sims = argc > 5 ? std::atoi(argv[5]) : 50;                      // :34
        std::vector<Vector> controls;                                             // :35-38
        for (int k = 0; k < steps; k++) controls.push_back(NewVector(1, {std::cos(0.75 * double(k + 1) * 0.1)}));

        const MonteCarloRuns runs = NewMonteCarloRuns(sims, steps, 1, controls, *mcKF);   // :40
        const std::vector<std::string> headers = {"xi", "xi_dot"};                // :41
        const auto csv = runs.AsCSV(headers);                                     // :42-46
        for (size_t fNo = 0; fNo < csv.size(); fNo++) std::ofstream(outdir + "/montecarlo-" + headers[fNo] + ".csv") << csv[fNo];
        // Run the Chi square tests.                                              // :48-51
        const auto chi = NewChiSquare(*chiKF, runs, controls, true, true);        // (NISmeans, NEESmeans); a failure throws
        // Output the NIS and NEES to a CSV file.                                 // :53-58
        std::FILE *f = std::fopen((outdir + "/chisquare.csv").c_str(), "w");
        if (!f) { std::perror("chisquare.csv"); return 2; }
        std::fputs("NIS,NEES\n", f);
        for (size_t k = 0; k < chi.first.size(); k++) std::fprintf(f, "%f,%f\n", chi.first[k], chi.second[k]);
        std::fclose(f);
    } catch (const Error &e) {
        std::fprintf(stderr, "gokalman error %d: %s\n", e.code, e.what());
        return 3;
    }
    return 0;
}
