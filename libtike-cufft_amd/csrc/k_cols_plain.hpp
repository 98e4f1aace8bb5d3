// k_cols_plain.hpp -- un-windowed column pass (fallback for ndet = 1024, column pass of ptycho_fft2)
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
#pragma once

// ---------------------------------------------------------------------------
// Column pass: DFT over y for a strip of C detector columns, fused with the
// probe / object product.  Thread (c, j0) holds points y = j0 + b*T + t*N/R of
// column x0 + c; LDS image is [y][c] (c fastest), conflict free in every step.
// ---------------------------------------------------------------------------
template <int N, int DIR, int MODE>
__global__ __launch_bounds__(ColCfg<N>::NT) void k_cols(const ColArgs a) {
    using P = Plan<N>;
    using F = Fft<P, DIR>;
    constexpr int E = P::E, T = P::T, C = ColCfg<N>::C, NT = ColCfg<N>::NT;
    constexpr int LAST = P::NSTEP - 1;
    __shared__ c32 lds[N * C];

    const int tid = threadIdx.x;
    const int c = tid % C, j0 = tid / C;
    const int strip = blockIdx.x % a.nstrips, group = blockIdx.x / a.nstrips;
    const int x0 = (a.strip0 + strip) * C;
    const int x = x0 + c;
    const Geom ge = a.ge;
    const int ix = x - ge.pad;
    const bool col_ok = ix >= 0 && ix < ge.nprb;
    const float cinv = 1.0f / (float)N;   // kernels.cu:65

    F fft;
    fft.init(j0, a.table);

    // slot -> y of the points this thread feeds into step 0 / receives from the last step
    // (both are j0 + b*T + t*N/R with the step's own R; they may order slots differently)
    c32 pr[E];   // FWD / ADJ_OBJ: c * probe strip; ADJ_PRB: gradient accumulators
    int cur_t = -1;
    const c32 zero = c32{0.0f, 0.0f};

    auto flush_probe = [&](int t) {
        // ADJ_PRB: add this workgroup's partial sums into prb[t]
        constexpr int R = P::radix(LAST), Ns = P::ns(LAST);
#pragma unroll
        for (int b = 0; b < E / R; ++b) {
            const int j = j0 + b * T;
            const int base = (j / Ns) * Ns * R + (j % Ns);
#pragma unroll
            for (int tt = 0; tt < R; ++tt) {
                const int iy = base + tt * Ns - ge.pad;
                if (col_ok && iy >= 0 && iy < ge.nprb) {
                    float* o = reinterpret_cast<float*>(a.dst + ((size_t)t * ge.nprb + iy) * ge.nprb + ix);
                    const c32 s = pr[b * R + tt] * cinv;
                    atomicAdd(o, s.x);
                    atomicAdd(o + 1, s.y);
                }
            }
        }
    };

    for (int k = a.k_begin + group; k < a.k_end; k += a.ngroups) {
        const int p = a.order ? a.order[k] : k;
        const int t = p / ge.nscan;
        if (MODE == M_FWD || MODE == M_ADJ_OBJ) {
            if (t != cur_t) {
                const c32* prb = a.aux + (size_t)t * ge.nprb * ge.nprb;
                constexpr int R = (MODE == M_FWD) ? P::radix(0) : P::radix(LAST);
                constexpr int Ns = (MODE == M_FWD) ? 1 : P::ns(LAST);
#pragma unroll
                for (int b = 0; b < E / R; ++b) {
                    const int j = j0 + b * T;
#pragma unroll
                    for (int tt = 0; tt < R; ++tt) {
                        const int y = (MODE == M_FWD) ? (j + tt * (N / R)) : ((j / Ns) * Ns * R + (j % Ns) + tt * Ns);
                        const int iy = y - ge.pad;
                        const bool ok = col_ok && iy >= 0 && iy < ge.nprb;
                        pr[b * R + tt] = ok ? prb[(size_t)iy * ge.nprb + ix] * cinv : zero;
                    }
                }
            }
        } else if (MODE == M_ADJ_PRB) {
            if (t != cur_t) {
                if (cur_t >= 0) flush_probe(cur_t);
#pragma unroll
                for (int s = 0; s < E; ++s) pr[s] = zero;
            }
        }
        cur_t = t;

        Pos q;
        if (MODE != M_PLAIN) {
            q = decode_pos(a.scan, p, ge);
            if (MODE != M_FWD && !q.valid) continue;   // uniform across the workgroup
        }
        const c32* ft = nullptr;   // object of this angle
        if (MODE == M_FWD) ft = a.src + (size_t)t * ge.nz * ge.n;
        if (MODE == M_ADJ_PRB) ft = a.aux + (size_t)t * ge.nz * ge.n;
        const c32* tile_in = nullptr;
        if (MODE == M_PLAIN) tile_in = a.src + (size_t)p * N * N;
        if (MODE == M_ADJ_OBJ || MODE == M_ADJ_PRB) tile_in = a.src + (size_t)(a.natural_tiles ? p : (k - a.k_begin)) * N * N;

        c32 v[E];
        // ---- step 0 input ---------------------------------------------------
        {
            constexpr int R = P::radix(0);
#pragma unroll
            for (int b = 0; b < E / R; ++b)
#pragma unroll
                for (int tt = 0; tt < R; ++tt) {
                    const int y = j0 + b * T + tt * (N / R);
                    c32 val;
                    if (MODE == M_FWD) {
                        const int iy = y - ge.pad;
                        const bool ok = q.valid && col_ok && iy >= 0 && iy < ge.nprb;
                        val = ok ? cmul(pr[b * R + tt], bilerp(ft, q.sy + iy, q.sx + ix, q, ge)) : zero;
                    } else {
                        val = tile_in[(size_t)y * N + x];
                    }
                    v[b * R + tt] = val;
                }
        }
        fft.template compute<0>(v);
        if (P::NSTEP > 1) {
            fft.template store<0>(v, j0, [&](int i, c32 val) { lds[i * C + c] = val; });
            __syncthreads();
            fft.template load<1>(v, j0, [&](int i) { return lds[i * C + c]; });
            if (P::NSTEP > 2) {
                __syncthreads();
                fft.template compute<1>(v);
                fft.template store<1>(v, j0, [&](int i, c32 val) { lds[i * C + c] = val; });
                __syncthreads();
                fft.template load<2>(v, j0, [&](int i) { return lds[i * C + c]; });
            }
            fft.template compute<LAST>(v);
        }
        // ---- last step output -----------------------------------------------
        constexpr int RL = P::radix(LAST), NsL = P::ns(LAST);
        if (MODE == M_FWD || MODE == M_PLAIN) {
            c32* tile_out = a.dst + (size_t)p * N * N;
            fft.template store<LAST>(v, j0, [&](int i, c32 val) { tile_out[(size_t)i * N + x] = val; });
            if (P::NSTEP > 1) __syncthreads();   // lds is rewritten by the next position
        } else if (MODE == M_ADJ_PRB) {
#pragma unroll
            for (int b = 0; b < E / RL; ++b) {
                const int j = j0 + b * T;
                const int base = (j / NsL) * NsL * RL + (j % NsL);
#pragma unroll
                for (int tt = 0; tt < RL; ++tt) {
                    const int iy = base + tt * NsL - ge.pad;
                    if (col_ok && iy >= 0 && iy < ge.nprb) {
                        const c32 val = v[b * RL + oslot(RL, tt)];
                        pr[b * RL + tt] += cmulc(val, bilerp(ft, q.sy + iy, q.sx + ix, q, ge));
                    }
                }
            }
            if (P::NSTEP > 1) __syncthreads();
        } else {   // M_ADJ_OBJ: tile T[y][c] = conj(c*prb) * near, then 4-tap combine + atomics
            if (P::NSTEP > 1) __syncthreads();   // everyone finished reading lds
#pragma unroll
            for (int b = 0; b < E / RL; ++b) {
                const int j = j0 + b * T;
                const int base = (j / NsL) * NsL * RL + (j % NsL);
#pragma unroll
                for (int tt = 0; tt < RL; ++tt) {
                    const c32 val = v[b * RL + oslot(RL, tt)];
                    const c32 w = pr[b * RL + tt];
                    lds[(base + tt * NsL) * C + c] = c32{w.x * val.x + w.y * val.y, w.x * val.y - w.y * val.x};
                }
            }
            __syncthreads();
            // output pixel (yy, cc): yy in [0, nprb] (probe rows, +1), cc in [0, C] of this strip
            const float wx0 = 1.0f - q.fx, wy0 = 1.0f - q.fy;
            c32* fo = a.dst + (size_t)t * ge.nz * ge.n;
            const int nout = (ge.nprb + 1) * (C + 1);
            for (int o = tid; o < nout; o += NT) {
                const int yy = o / (C + 1), cc = o % (C + 1);
                const int y = yy + ge.pad;   // nearplane row of tap (0,0)
                // taps: T[y][cc], T[y][cc-1], T[y-1][cc], T[y-1][cc-1]; zero outside the strip / tile.
                // T is zero by construction outside the probe window (pr = 0 there).
                const bool r0 = yy < ge.nprb, r1 = yy >= 1;
                const bool c0 = cc < C, c1 = cc >= 1;
                const c32 t00 = (r0 && c0) ? lds[y * C + cc] : zero;
                const c32 t01 = (r0 && c1) ? lds[y * C + cc - 1] : zero;
                const c32 t10 = (r1 && c0) ? lds[(y - 1) * C + cc] : zero;
                const c32 t11 = (r1 && c1) ? lds[(y - 1) * C + cc - 1] : zero;
                const c32 s = t00 * wx0 * wy0 + t01 * q.fx * wy0 + t10 * wx0 * q.fy + t11 * q.fx * q.fy;
                const int Y = q.sy + yy, X = q.sx + (x0 - ge.pad) + cc;
                if (Y >= 0 && Y < ge.nz && X >= 0 && X < ge.n && (x0 - ge.pad + cc) >= 0 && (x0 - ge.pad + cc) <= ge.nprb) {
                    float* op = reinterpret_cast<float*>(fo + (size_t)Y * ge.n + X);
                    atomicAdd(op, s.x);
                    atomicAdd(op + 1, s.y);
                }
            }
            __syncthreads();
        }
    }
    if (MODE == M_ADJ_PRB && cur_t >= 0) flush_probe(cur_t);
}
