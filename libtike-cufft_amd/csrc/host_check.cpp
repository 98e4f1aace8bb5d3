// Host-side check of fft_core.hpp: emulates the T cooperating threads of every
// plan sequentially (one "phase" per Stockham step, a barrier between phases)
// and compares with a naive double-precision DFT.  Built by
// tests/test_native_cpu.py::test_fft_core_on_host with clang++ (no GPU involved).
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "fft_core.hpp"

using namespace pty;

template <int N, int DIR>
static double check_plan() {
    using P = Plan<N>;
    using F = Fft<P, DIR>;
    constexpr int T = P::T, E = P::E;
    std::vector<c32> table(N), x(N), bufA(RowLds<N>::FS + 64), bufB(RowLds<N>::FS + 64), y(N);
    for (int k = 0; k < N; ++k) {
        table[k] = c32{(float)std::cos(-2.0 * M_PI * k / N), (float)std::sin(-2.0 * M_PI * k / N)};
        x[k] = c32{(float)std::sin(0.37 * k * k + 0.1), (float)std::cos(1.3 * k) * 0.5f};
    }
    std::vector<F> th(T);
    std::vector<std::vector<c32>> v(T, std::vector<c32>(E));
    for (int j0 = 0; j0 < T; ++j0) th[j0].init(j0, table.data());
    c32* cur = nullptr;
    c32* nxt = bufA.data();
    bool tab_bad = false;
    auto phase = [&](auto stc) {
        constexpr int ST = decltype(stc)::value;
        for (int j0 = 0; j0 < T; ++j0) {
            if (ST == 0)
                th[j0].template load<ST>(v[j0].data(), j0, [&](int i) { return x[i]; });
            else
                th[j0].template load<ST>(v[j0].data(), j0, [&](int i) { return cur[RowLds<N>::at(0, i)]; });
            // compute_tab (twiddles read from the table as they are used, k_tile.hpp) must equal compute bit for bit
            std::vector<c32> vt(v[j0]);
            th[j0].template compute<ST>(v[j0].data());
            th[j0].template compute_tab<ST>(vt.data(), j0, table.data());
            for (int m = 0; m < E; ++m)
                if (vt[m].x != v[j0][m].x || vt[m].y != v[j0][m].y) { std::printf("compute_tab mismatch N=%d step %d\n", N, ST); tab_bad = true; }
        }
        for (int j0 = 0; j0 < T; ++j0) {
            if (ST == P::NSTEP - 1)
                th[j0].template store<ST>(v[j0].data(), j0, [&](int i, c32 val) { y[i] = val; });
            else
                th[j0].template store<ST>(v[j0].data(), j0, [&](int i, c32 val) { nxt[RowLds<N>::at(0, i)] = val; });
        }
        cur = nxt;
        nxt = (nxt == bufA.data()) ? bufB.data() : bufA.data();
    };
    phase(std::integral_constant<int, 0>{});
    if constexpr (P::NSTEP > 1) phase(std::integral_constant<int, 1>{});
    if constexpr (P::NSTEP > 2) phase(std::integral_constant<int, 2>{});
    // natural-order helpers: element m of thread j0 must be index j0 + m*T
    for (int j0 = 0; j0 < T; ++j0) {
        c32 nat[E], back[E];
        F::to_natural(v[j0].data(), nat);
        for (int m = 0; m < E; ++m)
            if (nat[m].x != y[j0 + m * T].x || nat[m].y != y[j0 + m * T].y) { std::printf("to_natural mismatch N=%d\n", N); return 1.0; }
        F::from_natural(nat, back);
        c32 chk[E];
        th[j0].template load<0>(chk, j0, [&](int i) { return y[i]; });
        for (int m = 0; m < E; ++m)
            if (back[m].x != chk[m].x || back[m].y != chk[m].y) { std::printf("from_natural mismatch N=%d\n", N); return 1.0; }
    }
    if (tab_bad) return 1.0;
    double err = 0, nrm = 0;
    for (int k = 0; k < N; ++k) {
        std::complex<double> acc = 0;
        for (int n = 0; n < N; ++n)
            acc += std::complex<double>(x[n].x, x[n].y) * std::polar(1.0, DIR * 2.0 * M_PI * ((long)k * n % N) / N);
        err = std::fmax(err, std::abs(acc - std::complex<double>(y[k].x, y[k].y)));
        nrm = std::fmax(nrm, std::abs(acc));
    }
    std::printf("N=%d dir=%d rel_err=%.3e\n", N, DIR, err / nrm);
    return err / nrm;
}

template <int R, int DIR>
static double check_reg() {
    c32 v[R];
    std::complex<double> x[R];
    for (int i = 0; i < R; ++i) { v[i] = c32{(float)std::sin(i * 1.7 + 0.3), (float)std::cos(i * 0.9)}; x[i] = {v[i].x, v[i].y}; }
    fft_reg<R, DIR>(v);
    double err = 0;
    for (int k = 0; k < R; ++k) {
        std::complex<double> acc = 0;
        for (int n = 0; n < R; ++n) acc += x[n] * std::polar(1.0, DIR * 2.0 * M_PI * k * n / R);
        c32 g = v[oslot(R, k)];
        err = std::fmax(err, std::abs(acc - std::complex<double>(g.x, g.y)));
    }
    std::printf("reg R=%d dir=%d err=%.3e\n", R, DIR, err);
    return err;
}

int main() {
    double worst = 0;
    worst = std::fmax(worst, check_reg<2, -1>());
    worst = std::fmax(worst, check_reg<4, -1>());
    worst = std::fmax(worst, check_reg<8, -1>());
    worst = std::fmax(worst, check_reg<16, -1>());
    worst = std::fmax(worst, check_reg<32, -1>());
    worst = std::fmax(worst, check_reg<16, 1>());
    worst = std::fmax(worst, check_reg<8, 1>());
    worst = std::fmax(worst, check_reg<3, -1>());
    worst = std::fmax(worst, check_reg<5, -1>());
    worst = std::fmax(worst, check_reg<7, -1>());
    worst = std::fmax(worst, check_reg<3, 1>());
    worst = std::fmax(worst, check_reg<5, 1>());
    worst = std::fmax(worst, check_reg<7, 1>());
    worst = std::fmax(worst, check_plan<48, -1>());
    worst = std::fmax(worst, check_plan<80, -1>());
    worst = std::fmax(worst, check_plan<96, -1>());
    worst = std::fmax(worst, check_plan<112, -1>());
    worst = std::fmax(worst, check_plan<192, -1>());
    worst = std::fmax(worst, check_plan<48, 1>());
    worst = std::fmax(worst, check_plan<80, 1>());
    worst = std::fmax(worst, check_plan<96, 1>());
    worst = std::fmax(worst, check_plan<112, 1>());
    worst = std::fmax(worst, check_plan<192, 1>());
    worst = std::fmax(worst, check_plan<16, -1>());
    worst = std::fmax(worst, check_plan<32, -1>());
    worst = std::fmax(worst, check_plan<64, -1>());
    worst = std::fmax(worst, check_plan<128, -1>());
    worst = std::fmax(worst, check_plan<256, -1>());
    worst = std::fmax(worst, check_plan<512, -1>());
    worst = std::fmax(worst, check_plan<1024, -1>());
    worst = std::fmax(worst, check_plan<2048, -1>());
    worst = std::fmax(worst, check_plan<32, 1>());
    worst = std::fmax(worst, check_plan<64, 1>());
    worst = std::fmax(worst, check_plan<128, 1>());
    worst = std::fmax(worst, check_plan<256, 1>());
    worst = std::fmax(worst, check_plan<512, 1>());
    worst = std::fmax(worst, check_plan<1024, 1>());
    worst = std::fmax(worst, check_plan<2048, 1>());
    std::printf("worst=%.3e %s\n", worst, worst < 5e-6 ? "OK" : "FAIL");
    return worst < 5e-6 ? 0 : 1;
}
