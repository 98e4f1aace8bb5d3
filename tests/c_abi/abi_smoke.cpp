// Plain C++ client of the C ABI (include/ptycho_hip.h): no Python, no torch -- only
// hipMalloc'd buffers and the exported functions, the way the reference's C++ class
// ptychofft is driven by its SWIG wrapper (src/cuda/swig/ptychofft.i).  Checks the
// adjoint identities of /root/reference/tests/test_adjoint.py:47-56 on synthetic data and
// the analytic answer for a flat object (fwd = DFT2(pad(prb)) / ndet).
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ptycho_hip.h"

typedef std::complex<float> cf;
#define CK(x) do { int rc_ = (x); if (rc_) { std::printf("FAIL %s -> %d: %s\n", #x, rc_, ptycho_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const size_t ptheta = 1, nz = 160, n = 176, nscan = 49, ndet = 64, nprb = 48;
    ptycho_handle h = nullptr;
    if (ptycho_create(&h, ptheta, 1300, 1300, nscan, 1100, nprb) == 0) { std::printf("FAIL: ndet=1100 accepted\n"); return 1; }   // neither <= 1024 nor a power of two
    CK(ptycho_create(&h, ptheta, nz, n, nscan, ndet, nprb));
    if (ptycho_get(h, 4) != (long long)ndet || ptycho_get(h, 5) != (long long)nprb) { std::printf("FAIL get\n"); return 1; }

    std::vector<cf> psi(nz * n), prb(nprb * nprb), y(nscan * ndet * ndet);
    std::vector<float> scan(nscan * 2);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& v : psi) v = cf(1.0f + 0.2f * rnd(), 0.3f * rnd());
    for (auto& v : prb) v = cf(rnd(), rnd());
    for (auto& v : y) v = cf(rnd(), rnd());
    for (size_t i = 0; i < nscan; ++i) {
        scan[2 * i] = 14.0f * (float)(i / 7) + (rnd() + 0.5f);
        scan[2 * i + 1] = 16.0f * (float)(i % 7) + (rnd() + 0.5f);
    }
    void *d_psi, *d_prb, *d_scan, *d_g, *d_y, *d_f, *d_p;
    HK(hipMalloc(&d_psi, psi.size() * 8)); HK(hipMalloc(&d_prb, prb.size() * 8)); HK(hipMalloc(&d_scan, scan.size() * 4));
    HK(hipMalloc(&d_g, y.size() * 8)); HK(hipMalloc(&d_y, y.size() * 8)); HK(hipMalloc(&d_f, psi.size() * 8)); HK(hipMalloc(&d_p, prb.size() * 8));
    HK(hipMemcpy(d_psi, psi.data(), psi.size() * 8, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_prb, prb.data(), prb.size() * 8, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_scan, scan.data(), scan.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(d_y, y.data(), y.size() * 8, hipMemcpyHostToDevice));
    HK(hipMemset(d_f, 0, psi.size() * 8)); HK(hipMemset(d_p, 0, prb.size() * 8));

    CK(ptycho_fwd(h, d_g, d_psi, d_scan, d_prb, nullptr));          // default stream, like the reference
    CK(ptycho_adj(h, d_f, d_y, d_scan, d_prb, 0, nullptr));
    CK(ptycho_adj(h, d_psi, d_y, d_scan, d_p, 1, nullptr));
    HK(hipDeviceSynchronize());
    std::vector<cf> g(y.size()), aty(psi.size()), bty(prb.size());
    HK(hipMemcpy(g.data(), d_g, g.size() * 8, hipMemcpyDeviceToHost));
    HK(hipMemcpy(aty.data(), d_f, aty.size() * 8, hipMemcpyDeviceToHost));
    HK(hipMemcpy(bty.data(), d_p, bty.size() * 8, hipMemcpyDeviceToHost));
    std::complex<double> lhs = 0, r1 = 0, r2 = 0;
    for (size_t i = 0; i < g.size(); ++i) lhs += std::complex<double>(g[i]) * std::conj(std::complex<double>(y[i]));
    for (size_t i = 0; i < psi.size(); ++i) r1 += std::complex<double>(psi[i]) * std::conj(std::complex<double>(aty[i]));
    for (size_t i = 0; i < prb.size(); ++i) r2 += std::complex<double>(prb[i]) * std::conj(std::complex<double>(bty[i]));
    const double e1 = std::abs(lhs - r1) / std::abs(lhs), e2 = std::abs(lhs - r2) / std::abs(lhs);
    std::printf("<Ax,y>=(%.6e,%.6e)  rel residual object %.2e  probe %.2e\n", lhs.real(), lhs.imag(), e1, e2);
    if (!(e1 < 1e-5 && e2 < 1e-5)) { std::printf("FAIL adjoint\n"); return 1; }

    // flat object: every position gives DFT2(pad(prb)) / ndet, independent of the fractional shift
    for (auto& v : psi) v = cf(1.0f, 0.0f);
    HK(hipMemcpy(d_psi, psi.data(), psi.size() * 8, hipMemcpyHostToDevice));
    CK(ptycho_fwd(h, d_g, d_psi, d_scan, d_prb, nullptr));
    HK(hipMemcpy(g.data(), d_g, g.size() * 8, hipMemcpyDeviceToHost));
    const size_t pad = (ndet - nprb) / 2;
    double worst = 0, scale = 0;
    for (size_t ky = 0; ky < ndet; ky += 7)
        for (size_t kx = 0; kx < ndet; kx += 5) {
            std::complex<double> acc = 0;
            for (size_t iy = 0; iy < nprb; ++iy)
                for (size_t ix = 0; ix < nprb; ++ix)
                    acc += std::complex<double>(prb[iy * nprb + ix]) *
                           std::polar(1.0, -2.0 * M_PI * (double)((ky * (iy + pad) + kx * (ix + pad)) % ndet) / (double)ndet);
            acc /= (double)ndet;
            scale = std::fmax(scale, std::abs(acc));
            for (size_t p : {size_t(0), size_t(17), nscan - 1})
                worst = std::fmax(worst, std::abs(acc - std::complex<double>(g[(p * ndet + ky) * ndet + kx])));
        }
    std::printf("flat-object known answer: max err %.2e of %.2e\n", worst, scale);
    if (!(worst < 2e-5 * scale)) { std::printf("FAIL known answer\n"); return 1; }

    CK(ptycho_free(h));
    CK(ptycho_free(h));                                               // idempotent
    if (ptycho_fwd(h, d_g, d_psi, d_scan, d_prb, nullptr) != PTYCHO_ERR_FREED) { std::printf("FAIL use after free\n"); return 1; }
    CK(ptycho_destroy(h));
    std::printf("C ABI smoke OK\n");
    return 0;
}
