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
                                  const c32* __restrict__ near, const Geom ge, const int p0, const int p1, const int pgroup,
                                  long long* __restrict__ det_acc, const DetScale det) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= ge.nprb * ge.nprb) return;
    const float det_sc = det_acc ? det_scale_of(det) : 0.0f;
    auto add = [&](int t, int iy, int ix, c32 v) {   // one group's share of prb[t][iy][ix]
        const size_t e = ((size_t)t * ge.nprb + iy) * ge.nprb + ix;
        if (det_acc) {
            atomicAdd(reinterpret_cast<unsigned long long*>(det_acc + 2 * e), (unsigned long long)__float2ll_rn(v.x * det_sc));
            atomicAdd(reinterpret_cast<unsigned long long*>(det_acc + 2 * e + 1), (unsigned long long)__float2ll_rn(v.y * det_sc));
        } else {
            float* o = reinterpret_cast<float*>(prb + e);
            atomicAdd(o, v.x);
            atomicAdd(o + 1, v.y);
        }
    };
    const int ix = pix % ge.nprb, iy = pix / ge.nprb;
    const int N = ge.ndet;
    const int pa = p0 + blockIdx.y * pgroup;
    const int pb = pa + pgroup < p1 ? pa + pgroup : p1;
    c32 acc = c32{0.0f, 0.0f};
    int cur_t = -1;
    for (int p = pa; p < pb; ++p) {
        const int t = p / ge.nscan;
        if (t != cur_t && cur_t >= 0) {
            add(cur_t, iy, ix, acc * (1.0f / (float)N));
            acc = c32{0.0f, 0.0f};
        }
        cur_t = t;
        const Pos q = decode_pos(scan, p, ge);
        if (!q.valid) continue;
        const c32 patch = bilerp(f + (size_t)t * ge.nz * ge.n, q.sy + iy, q.sx + ix, q, ge);
        acc += cmulc(near[((size_t)(p - p0) * N + iy + ge.pad) * N + ix + ge.pad], patch);
    }
    if (cur_t >= 0) add(cur_t, iy, ix, acc * (1.0f / (float)N));
}


// ---------------------------------------------------------------------------------------------------------------
// Object adjoint for any detector size with the on-chip overlap-add of k_cols_adjwin (k_cols_window.hpp) instead of the
// reference's eight float atomics per probe pixel (kernels.cu:69-81, k_adj_obj_generic above): the near field is already
// fully transformed (Bluestein lines), so a workgroup only forms T = conj(c prb) near for a strip of 16 probe columns,
// combines the four bilinear taps and adds them to an LDS window of the object that follows a run of SORTED positions;
// rows are added to global memory once, when they slide out.  The window (nprb + 8 rows x 20 columns) is dynamic LDS.
// Tile of the k-th sorted position: a.src + (k - a.k_begin) ndet^2.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adjwin_generic(const ColArgs a, const int seglen) {
    extern __shared__ c32 gwin[];
    constexpr int C = 16, WC = C + kBucketPx, NT = 256;
    const Geom ge = a.ge;
    const int N = ge.ndet, H = ge.nprb + 8;
    const int tid = threadIdx.x;
    const int strip = blockIdx.x % a.nstrips, seg = blockIdx.x / a.nstrips;
    const int ix0 = strip * C;                     // first probe column of the strip
    const float cinv = 1.0f / (float)N;
    const c32 zero = c32{0.0f, 0.0f};
    for (int o = tid; o < H * WC; o += NT) gwin[o] = zero;
    const float det_sc = a.det_acc ? det_scale_of(a.det) : 0.0f;   // option deterministic: float -> 64-bit fixed point
    int t_w = -1, X0 = 0, Ybase = 0, Ytop = 0;     // live object rows [Ybase, Ytop), columns [X0, X0 + WC)

    auto flush = [&](int ya, int yb) {             // add rows [ya, yb) to the object and clear them
        if (yb <= ya) return;
        const int cnt = (yb - ya) * WC;
        for (int o = tid; o < cnt; o += NT) {
            const int Y = ya + o / WC, col = o % WC;
            const int slot = (Y % H) * WC + col;
            const c32 v = gwin[slot];
            gwin[slot] = zero;
            const int X = X0 + col;
            if ((v.x != 0.0f || v.y != 0.0f) && Y < ge.nz && X >= 0 && X < ge.n) {
                const size_t e = ((size_t)t_w * ge.nz + Y) * ge.n + X;
                if (a.det_acc) {   // integer atomics: the sum does not depend on the arrival order
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.det_acc + 2 * e), (unsigned long long)__float2ll_rn(v.x * det_sc));
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.det_acc + 2 * e + 1), (unsigned long long)__float2ll_rn(v.y * det_sc));
                } else {
                    float* op = reinterpret_cast<float*>(a.dst + e);
                    atomicAdd(op, v.x);
                    atomicAdd(op + 1, v.y);
                }
            }
        }
    };
    constexpr int NRG = NT / (C + 1), NITEM = (C + 1) * NRG;
    const int rpt = (ge.nprb + 1 + NRG - 1) / NRG;  // output rows per item

    const int kb = a.k_begin + seg * seglen;
    const int ke = kb + seglen < a.k_end ? kb + seglen : a.k_end;
    __shared__ RunMeta rm;
    load_run(rm, a.order, a.scan, kb, ke, tid);
    __syncthreads();
    for (int k = kb; k < ke; ++k) {
        const int p = uni_i(rm.p[k - kb]);
        const int t = p / ge.nscan;
        const Pos q = decode_xy(uni_f(rm.py[k - kb]), uni_f(rm.px[k - kb]), ge);
        if (!q.valid) continue;                     // uniform
        const c32* __restrict__ near = a.src + (size_t)(k - a.k_begin) * N * N;
        const c32* __restrict__ prb = a.aux + (size_t)t * ge.nprb * ge.nprb;
        __syncthreads();                            // the previous position's combine is done
        const int Xa = q.sx + ix0;                  // object column of strip column cc = 0
        const bool fitsw = (t == t_w) && Xa >= X0 && Xa + C < X0 + WC && q.sy >= Ybase;
        if (!fitsw) {
            flush(Ybase, Ytop);
            t_w = t;
            X0 = (q.sx / kBucketPx) * kBucketPx + ix0;
            Ybase = q.sy;
            Ytop = q.sy;
        } else if (q.sy > Ybase) {
            flush(Ybase, q.sy < Ytop ? q.sy : Ytop);
            Ybase = q.sy;
            if (Ytop < Ybase) Ytop = Ybase;
        }
        if (Ytop < q.sy + ge.nprb + 1) Ytop = q.sy + ge.nprb + 1;
        __syncthreads();                            // flushed rows are clean before they re-enter at the top
        // T(y, x) = conj(c prb[y][x]) near[y + pad][x + pad] for the 16 probe columns of THIS strip, zero elsewhere (the
        // neighbouring strips add their own columns' taps to the shared boundary column)
        auto Tat = [&](int y, int x) -> c32 {
            if (y < 0 || y >= ge.nprb || x < ix0 || x >= ix0 + C || x >= ge.nprb) return zero;
            return cmulc(near[(size_t)(y + ge.pad) * N + x + ge.pad], prb[(size_t)y * ge.nprb + x] * cinv);
        };
        const float w00 = (1.0f - q.fx) * (1.0f - q.fy), w01 = q.fx * (1.0f - q.fy);
        const float w10 = (1.0f - q.fx) * q.fy, w11 = q.fx * q.fy;
        for (int item = tid; item < NITEM; item += NT) {
            const int cc = item % (C + 1), rg = item / (C + 1);
            const int ixo = ix0 + cc;               // probe column of tap (., 0)
            if (ixo > ge.nprb) continue;
            const int y0 = rg * rpt;
            int y1 = y0 + rpt;
            if (y1 > ge.nprb + 1) y1 = ge.nprb + 1;
            if (y0 >= y1) continue;
            c32 up0 = Tat(y0 - 1, ixo), up1 = Tat(y0 - 1, ixo - 1);
            int slot = (q.sy + y0) % H;
            const int colw = Xa - X0 + cc;
            for (int y = y0; y < y1; ++y) {
                const c32 t00 = Tat(y, ixo), t01 = Tat(y, ixo - 1);
                gwin[slot * WC + colw] += t00 * w00 + t01 * w01 + up0 * w10 + up1 * w11;   // kernels.cu:73-80
                up0 = t00; up1 = t01;
                slot = slot + 1 == H ? 0 : slot + 1;
            }
        }
    }
    __syncthreads();
    flush(Ybase, Ytop);
}
