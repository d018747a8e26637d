// Replays the reference's examples/jerkcar scenario (main.go:94-161) through the C++ host mirror
// (include/gokalman_amd.hpp) for one filter kind and prints CSVExporter rows (exporter.go:34-45).
// usage: jerkcar_host <vanilla|sqrt|information> <uvec.csv> <yacchist.csv> <yposhist.csv> [channel]
// With "channel" the estimates go through a queue to a consumer thread that writes the rows, as main.go:71-90 does with
// a Go channel and a goroutine per exporter; the consumer is held back until the filter is 500 steps ahead, so every
// row it writes comes from an Estimate the filter has long moved past (the Estimate must own its data).
#include <cmath>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
#include <thread>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "gokalman_amd.hpp"

using namespace gokalman;

static std::vector<double> single_record(const char *path, bool one_per_line) {
    std::ifstream f(path);
    std::vector<double> out;
    std::string line;
    while (std::getline(f, line)) {
        std::stringstream ss(line);
        std::string tok;
        while (std::getline(ss, tok, ',')) {
            double v = std::strtod(tok.c_str(), nullptr);
            if (std::isnan(v)) v = 0.0;  // main.go:58-60
            out.push_back(v);
        }
        if (!one_per_line) break;
    }
    return out;
}

struct EstChan {   // chan gokalman.Estimate
    std::mutex m;
    std::condition_variable cv;
    std::deque<Estimate> q;
    size_t pushed = 0;
    bool closed = false;
    void send(Estimate e) { { std::lock_guard<std::mutex> l(m); q.push_back(std::move(e)); pushed++; } cv.notify_all(); }
    void close() { { std::lock_guard<std::mutex> l(m); closed = true; } cv.notify_all(); }
};

static void write_row(const Estimate &est) {
    const Vector x = est.State();
    const Matrix P = est.Covariance();
    for (int i = 0; i < x.rows; i++) {
        const double c = 2.0 * std::sqrt(P.At(i, i));
        std::printf("%s%.9f,%.9f,%.9f", i ? "," : "", x.At(i, 0), c, -c);
    }
    std::printf("\n");
}

// A fatal signal prints where it hit (glibc backtrace, async-signal-safe enough for a dying test program): round 6 saw ONE SIGSEGV of this program
// in several hundred runs, with core dumps disabled on the GPU boxes.
static void on_fatal(int sig) {
    void *frames[64];
    const int nf = backtrace(frames, 64);
    const char msg[] = "jerkcar_host: fatal signal, backtrace:\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, nf, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

int main(int argc, char **argv) {
    signal(SIGSEGV, on_fatal); signal(SIGBUS, on_fatal); signal(SIGABRT, on_fatal);
    if (argc < 5) return 2;
    try {
        const auto u = single_record(argv[2], true), yacc = single_record(argv[3], false), ypos = single_record(argv[4], false);
        Matrix F(4, 4, {1, 0.01, 0.00005, 0, 0, 1, 0.01, 0, 0, 0, 1, 0, 0, 0, 0, 1.0005125020836});
        Matrix G(4, 1, {0.0, 0.0001, 0.01, 0.0});
        Matrix H1(2, 4, {1, 0, 0, 0, 0, 0, 1, 1}), H2(1, 4, {0, 0, 1, 1});
        Matrix Q(4, 4, {0.0000000000025, 0.000000000625, 0.000000083333333, 0, 0.000000000625, 0.000000166666667, 0.000025, 0,
                        0.000000083333333, 0.000025, 0.005, 0, 0, 0, 0, 0.530265088355421});
        for (auto &v : Q.data) v *= 1e-3;
        const Noise noise1 = NewNoiseless(Q, Matrix(2, 2, {0.5, 0, 0, 0.05})), noise2 = NewNoiseless(Q, Matrix(1, 1, {0.05}));
        Vector x0 = NewVector(4, {0, 0.45, 0, 0.09});
        Matrix P0 = ScaledIdentity(4, 10);
        std::shared_ptr<LDKF> kf;
        const std::string kind = argv[1];
        const bool channel = argc > 5 && std::string(argv[5]) == "channel";
        EstChan chan;
        std::thread consumer;
        if (channel)
            consumer = std::thread([&chan]() {   // processEst (main.go:76-87)
                {   // lag: start draining only when the producer is 500 estimates ahead (or done)
                    std::unique_lock<std::mutex> l(chan.m);
                    chan.cv.wait(l, [&] { return chan.pushed >= 500 || chan.closed; });
                }
                for (;;) {
                    std::unique_lock<std::mutex> l(chan.m);
                    chan.cv.wait(l, [&] { return !chan.q.empty() || chan.closed; });
                    if (chan.q.empty()) break;
                    Estimate e = std::move(chan.q.front());
                    chan.q.pop_front();
                    l.unlock();
                    write_row(e);
                }
            });
        auto emit = [&](const Estimate &e) { if (channel) chan.send(e); else write_row(e); };
        if (kind == "vanilla") { auto pr = NewVanilla(x0, P0, F, G, H2, noise2, 1, 2); kf = pr.first; emit(pr.second); }
        else if (kind == "sqrt") { auto pr = NewSquareRoot(x0, P0, F, G, H2, noise2, 1, 2); kf = pr.first; emit(pr.second); }
        else { auto pr = NewInformation(NewVector(4), Matrix(4, 4), F, G, H2, noise2, 1, 2); kf = pr.first; emit(pr.second); }
        for (size_t k = 0; k < yacc.size(); k++) {
            Vector meas;
            if ((k + 1) % 10 == 0) {
                kf->SetMeasurementMatrix(H1);
                kf->SetNoise(noise1);
                meas = NewVector(2, {ypos[k], yacc[k]});
            } else {
                meas = NewVector(1, {yacc[k]});
            }
            emit(kf->Update(meas, NewVector(1, {u[k]})));
            if ((k + 1) % 10 == 0) {
                kf->SetMeasurementMatrix(H2);
                kf->SetNoise(noise2);
            }
        }
        if (channel) { chan.close(); consumer.join(); }
    } catch (const Error &e) {
        std::fprintf(stderr, "gokalman error %d: %s\n", e.code, e.what());
        return 3;
    }
    return 0;
}
