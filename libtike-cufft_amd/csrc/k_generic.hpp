// k_generic.hpp -- detector sizes that are not a power of two: Bluestein DFT lines + un-fused probe / object kernels
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
//
// The reference's cuFFT plan takes any ndet (src/cuda/ptychofft.cu:13-20; its own scripts crop 128 -> 112,
// tests/test_fsc.py:115-120).  The register FFT of fft_core.hpp is power-of-two only, so any other length n
// is computed as a chirp-z (Bluestein) transform on the power-of-two plan of length M >= 2 n - 1:
//     X[k] = b[k] sum_j (x[j] b[j]) conj(b)[k - j],   b[m] = exp(-/+ i pi m^2 / n),
// i.e. one forward and one inverse M-point FFT around a pointwise product with H = FFT_M(conj b) / M
// (host table, float64 -> float32).  This is a compatibility path: the transforms are not fused with the
// probe / object product (three plain kernels per operator instead), and it costs about 5x a native plan.
#pragma once

struct LineArgs {
    const c32* src;
    c32* dst;
    const c32* table;   // exp(-2 pi i k / M)
    const c32* chirp;   // b[m] = exp(-i pi m^2 / n), m < n
    const c32* hfilt;   // FFT_M(conj b placed circularly) / M
    long long nlines;
    int n;              // line length
    int ls, es;         // line l starts at (l / n) n^2 + (l % n) ls and steps by es (rows: ls = n, es = 1; columns: ls = 1, es = n)
    const int* tile_index;   // source tile of local tile j (nullptr: j); the destination is always local
};

template <int M, int DIR>
__global__ __launch_bounds__(256) void k_lines_bluestein(const LineArgs a) {
    using P = Plan<M>;
    using F = Fft<P, -1>;
    using L = RowLds<M>;
    constexpr int E = P::E, T = P::T, B = (256 / T) > 0 ? (256 / T) : 1;
    constexpr int NT = B * T;                 // threads that work (M = 2048: T = 128, B = 2)
    constexpr int LAST = P::NSTEP - 1;
    static_assert(NT <= 256, "one workgroup = 256 threads");
    __shared__ c32 lds[P::NSTEP > 1 ? B * L::FS : 1];
    const int tid = threadIdx.x;
    const int f = tid / T, j0 = tid % T;
    F fft;
    fft.init(j0, a.table);
    const c32 zero = c32{0.0f, 0.0f};
    const long long nb = (a.nlines + B - 1) / B;
    for (long long batch = blockIdx.x; batch < nb; batch += gridDim.x) {
        const long long l = batch * B + f;
        const bool ok = l < a.nlines && tid < NT;
        const long long tile = ok ? l / a.n : 0;
        const long long stile = (a.tile_index && ok) ? (long long)a.tile_index[tile] : tile;
        const long long off = (long long)(ok ? l % a.n : 0) * a.ls;
        const c32* sl = a.src + (size_t)stile * a.n * a.n + off;
        c32* dl = a.dst + (size_t)tile * a.n * a.n + off;
        c32 v[E], nat[E];
        fft.template load<0>(v, j0, [&](int i) {
            if (!(ok && i < a.n)) return zero;
            const c32 b = a.chirp[i];
            return DIR < 0 ? cmul(sl[(size_t)i * a.es], b) : cmulc(sl[(size_t)i * a.es], b);
        });
        // forward M-point FFT
        fft.template compute<0>(v);
        if (P::NSTEP > 1) {
            fft.template store<0>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
            __syncthreads();
            fft.template load<1>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            if (P::NSTEP > 2) {
                __syncthreads();
                fft.template compute<1>(v);
                fft.template store<1>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
                __syncthreads();
                fft.template load<2>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            }
            fft.template compute<LAST>(v);
            __syncthreads();
        }
        F::to_natural(v, nat);
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const c32 hh = a.hfilt[j0 + m * T];
            nat[m] = DIR < 0 ? cmul(nat[m], hh) : cmulc(nat[m], hh);
        }
        // inverse M-point FFT with the same twiddle registers (conjugated)
        F::from_natural(nat, v);
        fft.template compute_rev<0>(v);
        if (P::NSTEP > 1) {
            fft.template store<0>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
            __syncthreads();
            fft.template load<1>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            if (P::NSTEP > 2) {
                __syncthreads();
                fft.template compute_rev<1>(v);
                fft.template store<1>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
                __syncthreads();
                fft.template load<2>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            }
            fft.template compute_rev<LAST>(v);
        }
        fft.template store<LAST>(v, j0, [&](int i, c32 val) {
            if (ok && i < a.n) {
                const c32 b = a.chirp[i];
                dl[(size_t)i * a.es] = DIR < 0 ? cmul(val, b) : cmulc(val, b);
            }
        });
        if (P::NSTEP > 1) __syncthreads();
    }
}

// near = c prb . bilerp(psi) in a zero-bordered ndet x ndet frame (kernels.cu:95-107 + the memset of ptychofft.cu:69)
__global__ void k_near_generic(const c32* __restrict__ f, const c32* __restrict__ prb, const float* __restrict__ scan,
                               c32* __restrict__ near, const Geom ge, const long long total_pix) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_pix) return;
    const int N = ge.ndet;
    const int x = (int)(i % N), y = (int)((i / N) % N);
    const int p = (int)(i / ((long long)N * N));
    const int t = p / ge.nscan;
    const int ix = x - ge.pad, iy = y - ge.pad;
    c32 out = c32{0.0f, 0.0f};
    if (ix >= 0 && ix < ge.nprb && iy >= 0 && iy < ge.nprb) {
        const Pos q = decode_pos(scan, p, ge);
        if (q.valid) {
            const c32 patch = bilerp(f + (size_t)t * ge.nz * ge.n, q.sy + iy, q.sx + ix, q, ge);
            out = cmul(prb[((size_t)t * ge.nprb + iy) * ge.nprb + ix] * (1.0f / (float)N), patch);
        }
    }
    near[i] = out;
}

// object adjoint, four bilinear taps per probe pixel (kernels.cu:69-81); near tile of position p at src + (p - p0) n^2
__global__ void k_adj_obj_generic(c32* __restrict__ f, const c32* __restrict__ prb, const float* __restrict__ scan,
                                  const c32* __restrict__ near, const Geom ge, const int p0, const long long npix) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int ix = (int)(i % ge.nprb), iy = (int)((i / ge.nprb) % ge.nprb);
    const int pl = (int)(i / ((long long)ge.nprb * ge.nprb));
    const int p = p0 + pl, t = p / ge.nscan;
    const Pos q = decode_pos(scan, p, ge);
    if (!q.valid) return;
    const int N = ge.ndet;
    const c32 w = prb[((size_t)t * ge.nprb + iy) * ge.nprb + ix] * (1.0f / (float)N);
    const c32 T = cmulc(near[((size_t)pl * N + iy + ge.pad) * N + ix + ge.pad], w);   // conj(c prb) * near
    const float wgt[4] = {(1.0f - q.fx) * (1.0f - q.fy), q.fx * (1.0f - q.fy), (1.0f - q.fx) * q.fy, q.fx * q.fy};
    c32* ft = f + (size_t)t * ge.nz * ge.n;
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
        const int Y = q.sy + iy + (tap >> 1), X = q.sx + ix + (tap & 1);
        if (Y >= 0 && Y < ge.nz && X >= 0 && X < ge.n) {
            float* o = reinterpret_cast<float*>(ft + (size_t)Y * ge.n + X);
            atomicAdd(o, T.x * wgt[tap]);
            atomicAdd(o + 1, T.y * wgt[tap]);
        }
    }
}

// probe adjoint (kernels.cu:82-94): one thread per probe pixel and group of positions, partial sum in registers
__global__ void k_adj_prb_generic(const c32* __restrict__ f, c32* __restrict__ prb, const float* __restrict__ scan,
                                  const c32* __restrict__ near, const Geom ge, const int p0, const int p1, const int pgroup) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= ge.nprb * ge.nprb) return;
    const int ix = pix % ge.nprb, iy = pix / ge.nprb;
    const int N = ge.ndet;
    const int pa = p0 + blockIdx.y * pgroup;
    const int pb = pa + pgroup < p1 ? pa + pgroup : p1;
    c32 acc = c32{0.0f, 0.0f};
    int cur_t = -1;
    for (int p = pa; p < pb; ++p) {
        const int t = p / ge.nscan;
        if (t != cur_t && cur_t >= 0) {
            float* o = reinterpret_cast<float*>(prb + ((size_t)cur_t * ge.nprb + iy) * ge.nprb + ix);
            atomicAdd(o, acc.x * (1.0f / (float)N));
            atomicAdd(o + 1, acc.y * (1.0f / (float)N));
            acc = c32{0.0f, 0.0f};
        }
        cur_t = t;
        const Pos q = decode_pos(scan, p, ge);
        if (!q.valid) continue;
        const c32 patch = bilerp(f + (size_t)t * ge.nz * ge.n, q.sy + iy, q.sx + ix, q, ge);
        acc += cmulc(near[((size_t)(p - p0) * N + iy + ge.pad) * N + ix + ge.pad], patch);
    }
    if (cur_t >= 0) {
        float* o = reinterpret_cast<float*>(prb + ((size_t)cur_t * ge.nprb + iy) * ge.nprb + ix);
        atomicAdd(o, acc.x * (1.0f / (float)N));
        atomicAdd(o + 1, acc.y * (1.0f / (float)N));
    }
}
