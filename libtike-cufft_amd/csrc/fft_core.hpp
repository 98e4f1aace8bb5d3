// fft_core.hpp -- register/LDS Stockham FFT building blocks for gfx950.
//
// Replaces the closed-source cuFFT calls of the reference
// (/root/reference/src/cuda/ptychofft.cu:14-20,72,85): unnormalised 1-D DFTs of
// length N = 2^a 3^b 5^c 7^d (the powers of two 16 ... 2048 and 48, 80, 96, 112, 192 -- cuFFT takes such sizes natively and the
// reference's own script crops to 112 = 2^4 7, tests/test_fsc.py:115-120), composed into the batched 2-D transform by the
// kernels in ptycho_kernels.hip (columns and rows are separate passes).
//
// One FFT of length N is computed by T = N/E threads (T a power of two), E points per thread (16 for the powers of two)
// held in registers, as 1..3 Stockham steps of radix R_i (prod R_i = N, R_i | E; an odd prime radix comes first).
// Between steps the points are exchanged through LDS.  In step i (Ns = prod of
// earlier radices) butterfly j in [0, N/R) reads x[j + t*N/R], t = 0..R-1,
// multiplies by W_{Ns*R}^{(j mod Ns) t}, does an R-point in-register DFT and
// writes y[(j/Ns)*Ns*R + (j mod Ns) + t'*Ns].
//
// The header is plain C++ (clang vector extensions) so the index arithmetic can
// be exercised on the host: tests/test_native_cpu.py::test_fft_core_on_host builds csrc/host_check.cpp
// with clang++ and compares against numpy.fft.  On the device every loop below is
// fully unrolled and every array lives in VGPRs.
#pragma once

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define PTY_FN __device__ __forceinline__
#elif defined(__HIPCC__)
#define PTY_FN __host__ __device__ __forceinline__
#else
#define PTY_FN inline
#endif

namespace pty {

typedef float c32 __attribute__((ext_vector_type(2)));   // (re, im)

PTY_FN c32 cmul(c32 a, c32 b) { return c32{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
PTY_FN c32 cmulc(c32 a, c32 b) { return c32{a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y}; }  // a*conj(b)
PTY_FN c32 cconj(c32 a) { return c32{a.x, -a.y}; }

// cos/sin(2 pi k / 32)
#define PTY_COS32 { 1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f, 7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f, 0.0f, -1.950903220e-01f, -3.826834324e-01f, -5.555702330e-01f, -7.071067812e-01f, -8.314696123e-01f, -9.238795325e-01f, -9.807852804e-01f, -1.000000000e+00f, -9.807852804e-01f, -9.238795325e-01f, -8.314696123e-01f, -7.071067812e-01f, -5.555702330e-01f, -3.826834324e-01f, -1.950903220e-01f, 0.0f, 1.950903220e-01f, 3.826834324e-01f, 5.555702330e-01f, 7.071067812e-01f, 8.314696123e-01f, 9.238795325e-01f, 9.807852804e-01f }
#define PTY_SIN32 { 0.0f, 1.950903220e-01f, 3.826834324e-01f, 5.555702330e-01f, 7.071067812e-01f, 8.314696123e-01f, 9.238795325e-01f, 9.807852804e-01f, 1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f, 7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f, 0.0f, -1.950903220e-01f, -3.826834324e-01f, -5.555702330e-01f, -7.071067812e-01f, -8.314696123e-01f, -9.238795325e-01f, -9.807852804e-01f, -1.000000000e+00f, -9.807852804e-01f, -9.238795325e-01f, -8.314696123e-01f, -7.071067812e-01f, -5.555702330e-01f, -3.826834324e-01f, -1.950903220e-01f }

// t * exp(-/+ 2 pi i idx / 32); DIR = -1 forward (sign -), +1 inverse.  idx is a
// compile-time constant after unrolling, so the branches fold away.
template <int DIR>
PTY_FN c32 mul_w32(c32 t, int idx) {
    constexpr float C[32] = PTY_COS32;
    constexpr float S[32] = PTY_SIN32;
    constexpr float H = 7.071067812e-01f;
    idx &= 31;
    if (idx == 0) return t;
    if (idx == 16) return -t;
    if (idx == 8) return DIR < 0 ? c32{t.y, -t.x} : c32{-t.y, t.x};
    if (idx == 24) return DIR < 0 ? c32{-t.y, t.x} : c32{t.y, -t.x};
    if (idx == 4) return DIR < 0 ? c32{(t.x + t.y) * H, (t.y - t.x) * H} : c32{(t.x - t.y) * H, (t.y + t.x) * H};
    if (idx == 12) return DIR < 0 ? c32{(t.y - t.x) * H, -(t.x + t.y) * H} : c32{-(t.x + t.y) * H, (t.x - t.y) * H};
    if (idx == 20) return DIR < 0 ? c32{-(t.x + t.y) * H, (t.x - t.y) * H} : c32{(t.y - t.x) * H, -(t.x + t.y) * H};
    if (idx == 28) return DIR < 0 ? c32{(t.x - t.y) * H, (t.x + t.y) * H} : c32{(t.x + t.y) * H, (t.y - t.x) * H};
    const float c = C[idx], s = DIR < 0 ? -S[idx] : S[idx];   // w = c + i s
    return c32{t.x * c - t.y * s, t.x * s + t.y * c};
}

constexpr int ilog2(int x) { return x <= 1 ? 0 : 1 + ilog2(x >> 1); }
constexpr int brev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

// cos / sin (2 pi j / R), j < R, for the odd prime radices
template <int R> struct OddTab;
template <> struct OddTab<3> {
    static constexpr float C[3] = { 1.000000000e+00f, -5.000000000e-01f, -5.000000000e-01f };
    static constexpr float S[3] = { 0.000000000e+00f, 8.660254038e-01f, -8.660254038e-01f };
};
template <> struct OddTab<5> {
    static constexpr float C[5] = { 1.000000000e+00f, 3.090169944e-01f, -8.090169944e-01f, -8.090169944e-01f, 3.090169944e-01f };
    static constexpr float S[5] = { 0.000000000e+00f, 9.510565163e-01f, 5.877852523e-01f, -5.877852523e-01f, -9.510565163e-01f };
};
template <> struct OddTab<7> {
    static constexpr float C[7] = { 1.000000000e+00f, 6.234898019e-01f, -2.225209340e-01f, -9.009688679e-01f, -9.009688679e-01f, -2.225209340e-01f, 6.234898019e-01f };
    static constexpr float S[7] = { 0.000000000e+00f, 7.818314825e-01f, 9.749279122e-01f, 4.338837391e-01f, -4.338837391e-01f, -9.749279122e-01f, -7.818314825e-01f };
};

constexpr bool is_pow2(int x) { return x > 0 && (x & (x - 1)) == 0; }
// slot of output X[k] after fft_reg<R>: bit reversed for the radix-2 networks, natural for the odd primes
constexpr int oslot(int R, int k) { return is_pow2(R) ? brev(k, ilog2(R)) : k; }

// In-register R-point DFT.  R = 2..32 (power of two): decimation in frequency, radix 2; output X[k] is left in
// v[brev(k)].  R = 3, 5, 7: X[k] = x0 + sum_n (x_n + x_{R-n}) cos(2 pi n k / R) -/+ i sum_n (x_n - x_{R-n}) sin(2 pi n k / R),
// n = 1 .. (R-1)/2 (the pair sums / differences are shared by X[k] and X[R-k]); output X[k] in v[k].
template <int R, int DIR>
PTY_FN void fft_reg(c32* v) {
    if constexpr (is_pow2(R)) {
#pragma unroll
        for (int s = R / 2; s >= 1; s >>= 1) {
#pragma unroll
            for (int a = 0; a < R; ++a) {
                if (a & s) continue;
                const int b = a + s;
                const int k = a & (s - 1);
                const c32 t = v[a] - v[b];
                v[a] = v[a] + v[b];
                v[b] = mul_w32<DIR>(t, k * (16 / s));
            }
        }
    } else {
        static_assert(R == 3 || R == 5 || R == 7, "odd radices: 3, 5, 7");
        constexpr int H = (R - 1) / 2;
        c32 sm[H], df[H];
        c32 x0 = v[0], tot = v[0];
#pragma unroll
        for (int n = 1; n <= H; ++n) {
            sm[n - 1] = v[n] + v[R - n];
            df[n - 1] = v[n] - v[R - n];
            tot = tot + sm[n - 1];
        }
        v[0] = tot;
#pragma unroll
        for (int k = 1; k <= H; ++k) {
            c32 a = x0, b = c32{0.0f, 0.0f};
#pragma unroll
            for (int n = 1; n <= H; ++n) {
                a = a + sm[n - 1] * OddTab<R>::C[(n * k) % R];
                b = b + df[n - 1] * OddTab<R>::S[(n * k) % R];
            }
            // forward: X[k] = a - i b, X[R-k] = a + i b; inverse: the other way round.  -i (bx + i by) = by - i bx
            const c32 ib = c32{b.y, -b.x};
            v[k] = DIR < 0 ? a + ib : a - ib;
            v[R - k] = DIR < 0 ? a - ib : a + ib;
        }
    }
}

// ---------------------------------------------------------------------------
// Plans: N = R0*R1*R2, E points per thread, T = N/E threads per transform.
// ---------------------------------------------------------------------------
template <int N_, int R0_, int R1_ = 1, int R2_ = 1, int E_ = 16>
struct PlanT {
    static constexpr int N = N_, E = E_, T = N_ / E_;
    static constexpr int NSTEP = (R1_ == 1) ? 1 : (R2_ == 1 ? 2 : 3);
    static_assert(R0_ * R1_ * R2_ == N_, "radices must multiply to N");
    static_assert(E_ % R0_ == 0 && E_ % R1_ == 0 && E_ % R2_ == 0, "radix must divide E");
    static constexpr int radix(int i) { return i == 0 ? R0_ : (i == 1 ? R1_ : R2_); }
    static constexpr int ns(int i) { return i == 0 ? 1 : (i == 1 ? R0_ : R0_ * R1_); }
};
template <int N> struct Plan;
// sizes with an odd prime factor: T = 4 (8 at 192) threads per transform, the odd radix first (no twiddles before it)
template <> struct Plan<48> : PlanT<48, 3, 4, 4, 12> {};
template <> struct Plan<80> : PlanT<80, 5, 4, 4, 20> {};
template <> struct Plan<96> : PlanT<96, 3, 4, 8, 24> {};
template <> struct Plan<112> : PlanT<112, 7, 4, 4, 28> {};
template <> struct Plan<192> : PlanT<192, 3, 8, 8, 24> {};
template <> struct Plan<16> : PlanT<16, 16> {};
template <> struct Plan<32> : PlanT<32, 4, 8> {};
template <> struct Plan<64> : PlanT<64, 8, 8> {};
template <> struct Plan<128> : PlanT<128, 16, 8> {};
template <> struct Plan<256> : PlanT<256, 16, 16> {};
template <> struct Plan<512> : PlanT<512, 8, 8, 8> {};
template <> struct Plan<1024> : PlanT<1024, 16, 16, 4> {};
template <> struct Plan<2048> : PlanT<2048, 16, 16, 8> {};

// Per-thread FFT state: E points, the inter-step twiddles (loop invariant, so
// they stay in registers across the whole batch loop of a kernel).
template <class P, int DIR>
struct Fft {
    static constexpr int N = P::N, E = P::E, T = P::T;
    c32 tw[(P::NSTEP > 1 ? P::NSTEP - 1 : 1) * E];

    // table[k] = exp(-2 pi i k / N), k in [0, N); j0 in [0, T)
    PTY_FN void init(int j0, const c32* __restrict__ table) {
#pragma unroll
        for (int st = 1; st < P::NSTEP; ++st) {
            const int R = P::radix(st), Ns = P::ns(st);
#pragma unroll
            for (int b = 0; b < E / R; ++b) {
                const int j = j0 + b * T;
#pragma unroll
                for (int t = 0; t < R; ++t) {
                    const int k = ((j % Ns) * t * (N / (Ns * R))) % N;
                    const c32 w = table[k];
                    tw[(st - 1) * E + b * R + t] = DIR < 0 ? w : cconj(w);
                }
            }
        }
    }

    // twiddles of ONE step (kernels that cannot afford all NSTEP - 1 sets in registers at once re-read a set
    // from an LDS copy of the table right before the step that uses it)
    template <int ST>
    PTY_FN void init_step(int j0, const c32* __restrict__ table) {
        static_assert(ST >= 1 && ST < P::NSTEP, "steps 1 .. NSTEP-1 have twiddles");
        constexpr int R = P::radix(ST), Ns = P::ns(ST);
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            const int j = j0 + b * T;
#pragma unroll
            for (int t = 0; t < R; ++t) {
                const int k = ((j % Ns) * t * (N / (Ns * R))) % N;
                const c32 w = table[k];
                tw[(ST - 1) * E + b * R + t] = DIR < 0 ? w : cconj(w);
            }
        }
    }

    // v[b*R + t] = src(j + t*N/R)
    template <int ST, class Src>
    PTY_FN void load(c32* v, int j0, Src src) const {
        constexpr int R = P::radix(ST);
#pragma unroll
        for (int b = 0; b < E / R; ++b)
#pragma unroll
            for (int t = 0; t < R; ++t) v[b * R + t] = src(j0 + b * T + t * (N / R));
    }

    template <int ST>
    PTY_FN void compute(c32* v) const {
        constexpr int R = P::radix(ST);
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            if (ST > 0) {
#pragma unroll
                for (int t = 1; t < R; ++t) v[b * R + t] = cmul(v[b * R + t], tw[(ST > 0 ? ST - 1 : 0) * E + b * R + t]);
            }
            fft_reg<R, DIR>(v + b * R);
        }
    }

    // step ST with its twiddles read from `table` (an LDS copy of the plan's table) as they are used: no twiddle
    // registers at all (the kernels of k_tile.hpp run 1024 threads on 128 registers each)
    template <int ST>
    PTY_FN void compute_tab(c32* v, int j0, const c32* table) const {
        constexpr int R = P::radix(ST), Ns = P::ns(ST);
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            if (ST > 0) {
                const int j = j0 + b * T;
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    const c32 w = table[((j % Ns) * t * (N / (Ns * R))) % N];
                    v[b * R + t] = cmul(v[b * R + t], DIR < 0 ? w : cconj(w));
                }
            }
            fft_reg<R, DIR>(v + b * R);
        }
    }

    // same step in the opposite direction (conjugate twiddles): lets one kernel run a
    // forward and an inverse transform with a single set of twiddle registers
    template <int ST>
    PTY_FN void compute_rev(c32* v) const {
        constexpr int R = P::radix(ST);
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            if (ST > 0) {
#pragma unroll
                for (int t = 1; t < R; ++t) v[b * R + t] = cmulc(v[b * R + t], tw[(ST > 0 ? ST - 1 : 0) * E + b * R + t]);
            }
            fft_reg<R, -DIR>(v + b * R);
        }
    }

    // dst(out_index, value) for every point of this thread
    template <int ST, class Dst>
    PTY_FN void store(const c32* v, int j0, Dst dst) const {
        constexpr int R = P::radix(ST), Ns = P::ns(ST);
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            const int j = j0 + b * T;
            const int base = (j / Ns) * Ns * R + (j % Ns);
#pragma unroll
            for (int t = 0; t < R; ++t) dst(base + t * Ns, v[b * R + oslot(R, t)]);
        }
    }

    // Natural order of a thread's E points: element m is index j0 + m*T, both for the
    // outputs of the last step and for the inputs of step 0 (every N/R is a multiple of T).
    PTY_FN static void to_natural(const c32* v, c32* nat) {   // after compute<LAST>
        constexpr int R = P::radix(P::NSTEP - 1);
#pragma unroll
        for (int b = 0; b < E / R; ++b)
#pragma unroll
            for (int t = 0; t < R; ++t) nat[b + t * (E / R)] = v[b * R + oslot(R, t)];
    }
    PTY_FN static void from_natural(const c32* nat, c32* v) {   // before compute<0>
        constexpr int R = P::radix(0);
#pragma unroll
        for (int b = 0; b < E / R; ++b)
#pragma unroll
            for (int t = 0; t < R; ++t) v[b * R + t] = nat[b + t * (E / R)];
    }

    // index of the point held in slot (b, t) *before* step ST / *after* the last step
    template <int ST> static constexpr int in_index(int j0, int b, int t) {
        return j0 + b * T + t * (N / P::radix(ST));
    }
};

// Row layout in LDS: transform f occupies [f*FS, f*FS + N + N/16); one pad slot
// per 16 points keeps the radix-16 scatter of step 0 conflict free.
template <int N>
struct RowLds {
    static constexpr int FS = N + N / 16 + (((N + N / 16) * 8) % 256 == 0 ? 16 : 0);
    static constexpr int at(int f, int i) { return f * FS + i + (i >> 4); }
};

}  // namespace pty
