// k_cols_window.hpp -- column pass with the object strip in an LDS window: forward, object adjoint (overlap-add), probe adjoint; sort keys; arg-max pass
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
#pragma once

// ---------------------------------------------------------------------------
// Object adjoint with on-chip overlap-add (replaces the 8 atomics per probe pixel
// of kernels.cu:69-81).  A workgroup owns one strip of C probe columns and a
// contiguous run of positions in SORTED order (same angle, same BX-pixel column
// bucket, ascending row).  Consecutive positions of such a run overlap almost
// completely in the object, so their contributions are summed in an LDS window
// (H rows x WC columns of the object, rows addressed modulo H) and only rows that
// have slid out of the window are added to global memory, once.  Correctness does
// not depend on the order: a position that does not fit the current window
// flushes it and re-anchors.
// ---------------------------------------------------------------------------
constexpr int kBucketPx = 4;   // BX: column bucket of the sort key, and window slack

template <int N>
struct WinCfg {
    static constexpr int C = ColCfg<N>::C;
    static constexpr int WC = C + kBucketPx;   // window columns
    static constexpr int H = N + 8;            // window rows (>= nprb + 1)
    static constexpr bool fits = (size_t)((N + 2) * (C + 2) + H * WC) * sizeof(c32) <= 160 * 1024;
};

// CW: columns per strip (0: ColCfg<N>::C).  Round 4 tried 8-column strips for the CG column stages of small problems (a GPU's
// share of a strongly scaled job): twice the workgroups, half the work per position -- measured SLOWER at 256 / 512 / 1024
// positions (0.79 / 1.37 / 2.61 against 0.77 / 1.23 / 2.32 ms per iteration, profiles/r04/narrow_strips.txt); not instantiated.
template <int N, bool SPLIT = false, int CW = 0>
__global__ __launch_bounds__(Plan<N>::T * (CW ? CW : ColCfg<N>::C)) void k_cols_adjwin(const ColArgs a, const int seglen) {
    // SPLIT (N = 256): the tile already went through the first radix-16 step in k_rows_split;
    // only the twiddled second step runs here, straight from global memory (no exchange).
    using P = Plan<N>;
    using F = Fft<P, +1>;
    constexpr int E = P::E, T = P::T, C = CW ? CW : ColCfg<N>::C, NT = T * C;
    constexpr int LAST = P::NSTEP - 1;
    constexpr int WC = C + kBucketPx, H = WinCfg<N>::H;
    // exchange buffer = T tile, stored with a zero border: element (row i, column c) lives at
    // (i + 1) * CP + (c + 1); the border is written once and never touched again, which makes
    // the four bilinear taps of the combine unconditional loads.
    constexpr int CP = C + 2;
    __shared__ c32 lds[(N + 2) * CP];
    __shared__ c32 win[H * WC];

    const int tid = threadIdx.x;
    const int c = tid % C, j0 = tid / C;
    const int strip = blockIdx.x % a.nstrips, seg = blockIdx.x / a.nstrips;
    const int x0 = (a.strip0 + strip) * C;
    const int x = x0 + c;
    const Geom ge = a.ge;
    const int ix = x - ge.pad;
    const bool col_ok = ix >= 0 && ix < ge.nprb;
    const float cinv = 1.0f / (float)N;
    const c32 zero = c32{0.0f, 0.0f};
    const float det_sc = a.det_acc ? det_scale_of(a.det) : 0.0f;   // deterministic option: float -> fixed point
    auto at = [&](int i) { return (i + 1) * CP + c + 1; };

    F fft;
    fft.init(j0, a.table);
    for (int o = tid; o < H * WC; o += NT) win[o] = zero;
    for (int o = tid; o < (N + 2) * CP; o += NT) lds[o] = zero;

    c32 pr[E];   // c * probe strip, natural order (row j0 + m*T); zero on padding
    int cur_t = -1;
    // window state (uniform across the workgroup)
    int t_w = -1, X0 = 0, Ybase = 0, Ytop = 0;   // live object rows [Ybase, Ytop), columns [X0, X0+WC)

    auto flush = [&](int ya, int yb) {   // add rows [ya, yb) to the object and clear them
        if (yb <= ya) return;
        c32* fo = a.dst + (size_t)t_w * ge.nz * ge.n;
        const int cnt = (yb - ya) * WC;
        for (int o = tid; o < cnt; o += NT) {
            const int Y = ya + o / WC, col = o % WC;
            const int slot = (Y % H) * WC + col;
            const c32 v = win[slot];
            win[slot] = zero;
            const int X = X0 + col;
            if ((PTY_AB & 8) && v.x != 123.456f) continue;   // A/B ablation (wrong results): price of the flush atomics
            if ((v.x != 0.0f || v.y != 0.0f) && Y < ge.nz && X >= 0 && X < ge.n) {
                const size_t e = ((size_t)t_w * ge.nz + Y) * ge.n + X;
                if (a.det_acc) {
                    const float sc = det_sc;
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.det_acc + 2 * e), (unsigned long long)__float2ll_rn(v.x * sc));
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.det_acc + 2 * e + 1), (unsigned long long)__float2ll_rn(v.y * sc));
                } else {
                    float* op = reinterpret_cast<float*>(a.dst + e);
                    atomicAdd(op, v.x);
                    atomicAdd(op + 1, v.y);
                }
            }
        }
    };

    // combine mapping: item -> (output column cc in [0, C], group of consecutive output rows)
    constexpr int NRG = NT / (C + 1) > 0 ? NT / (C + 1) : 1;   // row groups
    constexpr int NITEM = (C + 1) * NRG;
    const int rpt = (ge.nprb + 1 + NRG - 1) / NRG;              // output rows per item

    const int kb = a.k_begin + seg * seglen;
    const int ke = kb + seglen < a.k_end ? kb + seglen : a.k_end;

    __shared__ RunMeta rm;
    load_run(rm, a.order, a.scan, kb, ke, tid);
    struct St { int p, t; Pos q; bool have; };
    auto decode = [&](int k) -> St {
        St st;
        st.have = k < ke;
        st.p = 0; st.t = 0; st.q = Pos{0, 0, 0.f, 0.f, false, false};
        if (!st.have) return st;
        st.p = uni_i(rm.p[k - kb]);
        st.t = st.p / ge.nscan;
        st.q = decode_xy(uni_f(rm.py[k - kb]), uni_f(rm.px[k - kb]), ge);
        return st;
    };
    auto tile_of = [&](const St& st, int k) {
        // A/B ablation bit 16 (wrong results): 128-byte row pieces that straddle two 128-byte lines, as object-space bands
        // (a workgroup owns 16 OBJECT columns: the tile columns it needs move with every position) would read them
        return a.src + (size_t)(a.natural_tiles ? st.p : (k - a.k_begin)) * N * N + (((PTY_AB & 16) && k + 1 < a.k_end && st.p + 1 < ge.ptheta * ge.nscan) ? ((st.p & 7) + 1) : 0);
    };

    __syncthreads();
    STAMP_DECL
    St st = decode(kb);
    c32 v[E];
    if (st.have && st.q.valid) {
        const c32* tile_in = tile_of(st, kb);
        if (a.nt & 8)
            fft.template load<(SPLIT ? 1 : 0)>(v, j0, [&](int i) { return __builtin_nontemporal_load(tile_in + (size_t)i * N + x); });
        else
            fft.template load<(SPLIT ? 1 : 0)>(v, j0, [&](int i) { return tile_in[(size_t)i * N + x]; });
    }
    for (int k = kb; k < ke; ++k) {
        St nx = decode(k + 1);
        if (!st.q.valid) {   // skipped position: nothing to add; fetch the next tile
            if (nx.have && nx.q.valid) {
                const c32* tile_in = tile_of(nx, k + 1);
                fft.template load<(SPLIT ? 1 : 0)>(v, j0, [&](int i) { return tile_in[(size_t)i * N + x]; });
            }
            st = nx;
            continue;
        }
        const Pos q = st.q;
        if (st.t != cur_t || (PTY_AB & 32)) {   // A/B bit 32: reloaded for every position (object-space bands would have to)
            const c32* prb = a.aux + (size_t)st.t * ge.nprb * ge.nprb;
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const int iy = j0 + m * T - ge.pad;
                const bool ok = col_ok && iy >= 0 && iy < ge.nprb;
                const c32 w = prb[ok ? ((size_t)iy * ge.nprb + ix) : 0];
                pr[m] = ok ? w * cinv : zero;
            }
            cur_t = st.t;
        }
        // ---- inverse DFT over y of this strip (tile already in v) ----------------------
        STAMP(0)          // loop head, decode, probe strip
        STAMP_DRAIN();
        STAMP(1)          // wait for the tile loads
        if (SPLIT) {
            fft.template compute<LAST>(v);   // twiddle + radix 16: the step k_rows_split left
        } else {
        fft.template compute<0>(v);
        if (P::NSTEP > 1) {
            fft.template store<0>(v, j0, [&](int i, c32 val) { lds[at(i)] = val; });
            __syncthreads();
            fft.template load<1>(v, j0, [&](int i) { return lds[at(i)]; });
            if (P::NSTEP > 2) {
                __syncthreads();
                fft.template compute<1>(v);
                fft.template store<1>(v, j0, [&](int i, c32 val) { lds[at(i)] = val; });
                __syncthreads();
                fft.template load<2>(v, j0, [&](int i) { return lds[at(i)]; });
            }
            fft.template compute<LAST>(v);
        }
        }
        // ---- T[y][c] = conj(c * prb) * near, written over the slots this thread just read ----
        {
            c32 nat[E];
            F::to_natural(v, nat);
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const c32 w = pr[m];
                nat[m] = c32{w.x * nat[m].x + w.y * nat[m].y, w.x * nat[m].y - w.y * nat[m].x};
            }
            // split kernel: no exchange went through the tile, so the "previous combine is done"
            // barrier sits here, after this position's transform, instead of at the end of the loop
            STAMP(2)      // transform + probe product
            if (SPLIT) __syncthreads();
            STAMP(3)      // barrier: previous combine done
#pragma unroll
            for (int m = 0; m < E; ++m) lds[at(j0 + m * T)] = nat[m];
        }
        STAMP(8)          // T store
        // prefetch the next tile while the combine runs
        if (nx.have && nx.q.valid) {
            const c32* tile_in = tile_of(nx, k + 1);
            if (a.nt & 8)
                fft.template load<(SPLIT ? 1 : 0)>(v, j0, [&](int i) { return __builtin_nontemporal_load(tile_in + (size_t)i * N + x); });
            else
                fft.template load<(SPLIT ? 1 : 0)>(v, j0, [&](int i) { return tile_in[(size_t)i * N + x]; });
        }
        STAMP(9)          // prefetch issue
        // ---- window bookkeeping (all quantities are workgroup-uniform) -------------------
        const int Xa = q.sx + x0 - ge.pad;   // object column of strip column cc = 0
        const bool fitsw = (st.t == t_w) && Xa >= X0 && Xa + C < X0 + WC && q.sy >= Ybase;
        if (!fitsw) {
            __syncthreads();
            flush(Ybase, Ytop);
            t_w = st.t;
            X0 = (q.sx / kBucketPx) * kBucketPx + x0 - ge.pad;
            Ybase = q.sy;
            Ytop = q.sy;
        } else if (q.sy > Ybase) {
            flush(Ybase, q.sy < Ytop ? q.sy : Ytop);   // rows below q.sy: disjoint from this combine
            Ybase = q.sy;
            if (Ytop < Ybase) Ytop = Ybase;
        }
        if (Ytop < q.sy + ge.nprb + 1) Ytop = q.sy + ge.nprb + 1;
        STAMP(4)          // T store, prefetch issue, window bookkeeping, flush
        __syncthreads();   // T tile complete (and, after a re-anchor, the window is clean)
        STAMP(5)          // barrier: T tile complete
        // ---- 4-tap bilinear combine (kernels.cu:73-80) into the window -------------------
        for (int item = tid; item < NITEM; item += NT) {
            const int cc = item % (C + 1), rg = item / (C + 1);
            const int ixo = x0 - ge.pad + cc;                     // probe column of tap (., 0)
            if (ixo < 0 || ixo > ge.nprb) continue;
            const float w00 = (1.0f - q.fx) * (1.0f - q.fy), w01 = q.fx * (1.0f - q.fy);
            const float w10 = (1.0f - q.fx) * q.fy, w11 = q.fx * q.fy;
            const int y0 = rg * rpt;
            int y1 = y0 + rpt;
            if (y1 > ge.nprb + 1) y1 = ge.nprb + 1;
            if (y0 >= y1) continue;
            // padded tile: T[y][cc] at (y + 1) * CP + cc + 1 and T[y][cc - 1] at (y + 1) * CP + cc
            const c32* tp = lds + (y0 + ge.pad) * CP + cc;        // row y - 1 = yy + pad - 1
            c32 up0 = tp[1], up1 = tp[0];                         // T[y-1][cc], T[y-1][cc-1]
            int slot = (q.sy + y0) % H;
            const int colw = Xa - X0 + cc;
            int yy = y0;
            // U rows per trip: the tile taps and the window values of all U rows are requested before the first is used
            // (the one-row loop waits for an LDS round trip per row: the combine is a latency chain, not bandwidth)
            auto trips = [&](auto uc) {
                constexpr int U = decltype(uc)::value;
                for (; yy + U <= y1; yy += U) {
                    c32 t0[U], t1[U], wv[U];
                    c32* wp[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const c32* tq = tp + (u + 1) * CP;
                        int su = slot + u;
                        su = su >= H ? su - H : su;
                        wp[u] = win + su * WC + colw;
                        if (PTY_AB & 64) { t0[u] = c32{w00, w01}; t1[u] = c32{w10, (float)u}; }   // A/B ablation: no tile reads
                        else { t0[u] = tq[1]; t1[u] = tq[0]; }
                    }
                    if (PTY_AB & 128) {   // A/B ablation: no window read-modify-write (one store at the end of the trip keeps the sums alive)
                        c32 accw = zero;
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            accw += t0[u] * w00 + t1[u] * w01 + up0 * w10 + up1 * w11;
                            up0 = t0[u]; up1 = t1[u];
                        }
                        if (accw.x == 123.456f) *wp[0] = accw;
                    } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) wv[u] = *wp[u];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        *wp[u] = wv[u] + (t0[u] * w00 + t1[u] * w01 + up0 * w10 + up1 * w11);
                        up0 = t0[u]; up1 = t1[u];
                    }
                    }
                    tp += U * CP;
                    slot += U;
                    slot = slot >= H ? slot - H : slot;
                }
            };
            trips(std::integral_constant<int, 3>{});
            for (; yy < y1; ++yy) {
                tp += CP;
                const c32 t00 = tp[1], t01 = tp[0];
                win[slot * WC + colw] += t00 * w00 + t01 * w01 + up0 * w10 + up1 * w11;
                up0 = t00; up1 = t01;
                slot = slot + 1 == H ? 0 : slot + 1;
            }
        }
        STAMP(6)          // combine
        if (!SPLIT) __syncthreads();   // combine done: the tile may be overwritten by the next position
        st = nx;
    }
    __syncthreads();
    flush(Ybase, Ytop);
    STAMP(7)
    STAMP_FLUSH(a.stamps ? a.stamps + 12 : a.stamps)
}

// ---------------------------------------------------------------------------
// Forward operator / probe adjoint with the object strip cached in LDS.
// Same run structure as k_cols_adjwin: a workgroup owns C probe columns and a
// contiguous run of SORTED positions; the object rows it needs slide by a few
// pixels from one position to the next, so only the new rows are fetched from
// global memory (the reference re-reads four taps per probe pixel per position,
// kernels.cu:97-104 / :84-91).  Out-of-object taps are stored as zeros, so no
// separate edge path exists.  The hot loop is branch free: padding pixels are
// masked by a zero probe value (window is zero-initialised, so stale rows are finite).
//   M_FWD     : v = (c*prb) * bilerp(window)  -> DFT over y -> strip of g
//   M_ADJ_PRB : IDFT over y of the scratch strip; acc += near * conj(bilerp(window))
// ---------------------------------------------------------------------------
// CW: columns per strip (0: ColCfg<N>::C).  The split forward pass of ndet = 256 runs 32-column strips (512 threads, two
// workgroups = 16 waves per CU, 256-byte row pieces): 0.474 -> 0.455 ms against 16-column strips (round 3, A/B on one box;
// the object adjoint loses with them: one workgroup of eight waves per CU).
template <int N, int MODE, bool SPLIT = false, int NM = 1, int CW = 0>
__global__ __launch_bounds__(Plan<N>::T * (CW ? CW : ColCfg<N>::C)) void k_cols_gatherwin(const ColArgs a, const int seglen) {
    // SPLIT (N = 256): only one radix-16 step of the DFT over y runs here (thread local, no
    // exchange buffer -> 44 KiB of LDS, three workgroups per CU); k_rows_split does the other.
    // NM > 1 (forward only): NM probe modes per launch.  The bilinear patch values of a position are
    // gathered from the window ONCE and multiplied by the NM probe strips (all in VGPRs), giving the
    // strips of NM farplanes dstm[k] (ptycho.py:330-333 calls fwd once per mode and gathers each time).
    static_assert(NM == 1 || (MODE == M_FWD && !SPLIT), "several probe modes: un-split forward pass only");
    if (MODE == M_FWD && a.skip && *a.skip != 0.0) return;   // uniform over the grid
    using P = Plan<N>;
    constexpr int DIR = (MODE == M_FWD) ? -1 : +1;
    using F = Fft<P, DIR>;
    constexpr int E = P::E, T = P::T, C = CW ? CW : ColCfg<N>::C, NT = T * C;
    constexpr int LAST = P::NSTEP - 1;
    constexpr int WC = C + kBucketPx, H = WinCfg<N>::H;
    constexpr int R0 = P::radix(0), RL = P::radix(LAST), NsL = P::ns(LAST);
    __shared__ c32 lds[SPLIT ? 1 : N * C];
    // window rows are addressed modulo H; row H mirrors row 0, so the tap of the row below never wraps and both rows
    // of a bilinear patch are read from one base address with immediate offsets (no address arithmetic for the second)
    __shared__ c32 win[(H + 1) * WC];

    const int tid = threadIdx.x;
    const int c = tid % C, j0 = tid / C;
    const int strip = blockIdx.x % a.nstrips, seg = blockIdx.x / a.nstrips;
    const int x0 = (a.strip0 + strip) * C;
    const int x = x0 + c;
    const Geom ge = a.ge;
    const int ix = x - ge.pad;
    const bool col_ok = ix >= 0 && ix < ge.nprb;
    const float cinv = 1.0f / (float)N;
    const c32 zero = c32{0.0f, 0.0f};
    const float det_sc = (MODE == M_ADJ_PRB && a.det_acc) ? det_scale_of(a.det) : 0.0f;   // deterministic option

    // NM > 1: the inter-step twiddles are re-read from an LDS copy of the table before each step (init_step) instead
    // of living in up to 64 VGPRs next to the shared patch values and the probe strip in flight
    // (Round 3: for ndet <= 256 and two modes the registers could hold both probe strips and the twiddles; hoisting them
    // out of the position loop measured SLOWER -- 1.24 -> 1.49 ms per pass at 4096 x 256^2 -- so every NM > 1 takes this path.)
    __shared__ c32 wtab[NM > 1 ? N : 1];
    F fft;
    if (NM == 1) fft.init(j0, a.table);
    else for (int o = tid; o < N; o += NT) wtab[o] = a.table[o];
    for (int o = tid; o < (H + 1) * WC; o += NT) win[o] = zero;

    // FWD: c * probe strip in step-0 slot order (zero on padding -> masks the gather);
    // ADJ_PRB: gradient accumulators in natural order m (row j0 + m*T)
    // NM > 1: the probe strips are streamed from L2 (next mode's strip in flight during this mode's transform):
    // NM strips do not fit the registers next to three sets of twiddles at ndet = 512
    c32 pr[E];
    c32 prnext[NM > 1 ? E : 1];
    int cur_t = -1;
    auto probe_strip = [&](c32* dst, int km, int t) {
        const c32* prb = (NM == 1 ? a.aux : a.auxm[km]) + (size_t)t * ge.nprb * ge.nprb;
#pragma unroll
        for (int b = 0; b < E / R0; ++b)
#pragma unroll
            for (int tt = 0; tt < R0; ++tt) {
                const int iy = j0 + b * T + tt * (N / R0) - ge.pad;
                const bool ok = col_ok && iy >= 0 && iy < ge.nprb;
                const c32 w = prb[ok ? ((size_t)iy * ge.nprb + ix) : 0];
                dst[b * R0 + tt] = ok ? w * cinv : zero;
            }
    };
    int t_w = -1, X0 = 0, Ylo = 0, Yhi = 0;   // cached object rows [Ylo, Yhi), columns [X0, X0+WC)

    auto flush_probe = [&](int t) {
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int iy = j0 + m * T - ge.pad;
            if (col_ok && iy >= 0 && iy < ge.nprb) {
                const size_t e = ((size_t)t * ge.nprb + iy) * ge.nprb + ix;
                const c32 sacc = pr[m] * cinv;
                if (a.det_acc) {
                    const float sc = det_sc;
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.det_acc + 2 * e), (unsigned long long)__float2ll_rn(sacc.x * sc));
                    atomicAdd(reinterpret_cast<unsigned long long*>(a.det_acc + 2 * e + 1), (unsigned long long)__float2ll_rn(sacc.y * sc));
                } else {
                    float* o = reinterpret_cast<float*>(a.dst + e);
                    atomicAdd(o, sacc.x);
                    atomicAdd(o + 1, sacc.y);
                }
            }
        }
    };

    const int kb = a.k_begin + seg * seglen;
    const int ke = kb + seglen < a.k_end ? kb + seglen : a.k_end;
    __shared__ RunMeta rm;
    load_run(rm, a.order, a.scan, kb, ke, tid);

    struct St { int p, t, Xa; Pos q; bool have; };
    // Window update for position k, split so that the global loads of the rows that slide in
    // overlap the second half of the previous position's transform:
    //   prepare_issue  decodes k (from LDS), slides / re-anchors the window (workgroup-uniform
    //                  bookkeeping) and starts this thread's load of one new element;
    //   prepare_commit stores it to the window.  A re-anchor (more new elements than threads)
    //                  is loaded in place by prepare_issue.
    c32 pre_val = zero;
    int pre_slot = -1;
    bool pre_inb = false;
    auto prepare_issue = [&](int k, int kend) -> St {
        St st;
        st.have = k < kend;
        st.p = 0; st.t = 0; st.Xa = 0; st.q = Pos{0, 0, 0.f, 0.f, false, false};
        pre_slot = -1;
        if (!st.have) return st;
        st.p = uni_i(rm.p[k - kb]);
        st.t = st.p / ge.nscan;
        st.q = decode_xy(uni_f(rm.py[k - kb]), uni_f(rm.px[k - kb]), ge);
        if (!st.q.valid) return st;
        const c32* ft = (MODE == M_FWD ? a.src : a.aux) + (size_t)st.t * ge.nz * ge.n;
        st.Xa = st.q.sx + x0 - ge.pad;
        const int Ra = st.q.sy, Rb = st.q.sy + ge.nprb + 1;
        const bool colfit = (st.t == t_w) && st.Xa >= X0 && st.Xa + C < X0 + WC;
        if (!colfit) {
            t_w = st.t;
            X0 = (st.q.sx / kBucketPx) * kBucketPx + x0 - ge.pad;
            Ylo = Ra; Yhi = Ra;
        } else if (Ra < Ylo || Ra > Yhi) {
            Ylo = Ra; Yhi = Ra;
        } else {
            Ylo = Ra;
        }
        if (Rb > Yhi) {
            const int cnt = (Rb - Yhi) * WC;
            if (cnt <= NT) {
                if (tid < cnt) {
                    const int Y = Yhi + tid / WC, col = tid % WC;
                    const int X = X0 + col;
                    const bool inb = Y < ge.nz && X >= 0 && X < ge.n;
                    pre_val = ft[inb ? ((size_t)Y * ge.n + X) : 0];   // raw: the select waits until prepare_commit, so
                    pre_inb = inb;                                      // nothing here has to wait for the load
                    pre_slot = (Y % H) * WC + col;
                }
            } else {
                for (int o = tid; o < cnt; o += NT) {
                    const int Y = Yhi + o / WC, col = o % WC;
                    const int X = X0 + col;
                    const bool inb = Y < ge.nz && X >= 0 && X < ge.n;
                    const c32 val = ft[inb ? ((size_t)Y * ge.n + X) : 0];
                    const int ws = (Y % H) * WC + col;
                    win[ws] = inb ? val : zero;
                    if (ws < WC) win[H * WC + ws] = inb ? val : zero;   // mirror of row 0
                }
            }
            Yhi = Rb;
        }
        return st;
    };
    auto prepare_commit = [&]() {
        if (pre_slot >= 0) {
            const c32 val = pre_inb ? pre_val : zero;
            win[pre_slot] = val;
            if (pre_slot < WC) win[H * WC + pre_slot] = val;   // mirror of row 0
        }
    };
    auto prepare = [&](int k, int kend) -> St {
        St st = prepare_issue(k, kend);
        prepare_commit();
        return st;
    };

    __syncthreads();
    if constexpr (MODE == M_ADJ_PRB) {
        // Probe adjoint: the window of position k is only needed after its transform, so its
        // update is issued right after the exchange barrier of k, and the tile of k+1 is
        // prefetched before the accumulation of k: two barriers per position.
        auto decode_only = [&](int k) -> St {
            St st;
            st.have = k < ke;
            st.p = 0; st.t = 0; st.Xa = 0; st.q = Pos{0, 0, 0.f, 0.f, false, false};
            if (!st.have) return st;
            st.p = rm.p[k - kb];
            st.t = st.p / ge.nscan;
            st.q = decode_xy(rm.py[k - kb], rm.px[k - kb], ge);
            return st;
        };
        auto load_tile = [&](c32* v, const St& st, int k) {
            const c32* tile_in = a.src + (size_t)(a.natural_tiles ? st.p : (k - a.k_begin)) * N * N;
            fft.template load<(SPLIT ? 1 : 0)>(v, j0, [&](int i) { return tile_in[(size_t)i * N + x]; });
        };
        __syncthreads();   // run metadata visible
        St st = decode_only(kb);
        c32 v[E];
        if (st.have && st.q.valid) load_tile(v, st, kb);
        for (int k = kb; k < ke; ++k) {
            St nx = decode_only(k + 1);
            if (!st.q.valid) {
                if (nx.have && nx.q.valid) load_tile(v, nx, k + 1);
                st = nx;
                continue;
            }
            if (st.t != cur_t) {
                if (cur_t >= 0) flush_probe(cur_t);
#pragma unroll
                for (int m = 0; m < E; ++m) pr[m] = zero;
                cur_t = st.t;
            }
            if (!SPLIT) fft.template compute<0>(v);
            if (!SPLIT && P::NSTEP > 1) {
                fft.template store<0>(v, j0, [&](int i, c32 val) { lds[i * C + c] = val; });
                __syncthreads();   // also: the accumulation of k-1 is over, the window may move
            } else {
                __syncthreads();
            }
            const St cur = prepare_issue(k, ke);   // same decode as st, plus the window update
            if (SPLIT) {
                fft.template compute<LAST>(v);
            } else if (P::NSTEP > 1) {
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
            c32 nat[E];
            F::to_natural(v, nat);
            if (nx.have && nx.q.valid) load_tile(v, nx, k + 1);   // prefetch under the accumulation
            prepare_commit();
            __syncthreads();   // window rows of k in place; exchange buffer free
            const Pos q = cur.q;
            const float w00 = (1.0f - q.fx) * (1.0f - q.fy), w01 = q.fx * (1.0f - q.fy);
            const float w10 = (1.0f - q.fx) * q.fy, w11 = q.fx * q.fy;
            const int colw = cur.Xa - X0 + c;
            int slot = (q.sy + j0 - ge.pad + 2 * H) % H;
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const int iy = j0 + m * T - ge.pad;
                const bool ok = col_ok && iy >= 0 && iy < ge.nprb;
                const c32* r0 = win + slot * WC + colw;
                const c32 patch = r0[0] * w00 + r0[1] * w01 + r0[WC] * w10 + r0[WC + 1] * w11;   // kernels.cu:84-91
                const c32 term = cmulc(nat[m], patch);
                pr[m] += ok ? term : zero;
                slot += T;
                slot = slot >= H ? slot - H : slot;
            }
            st = nx;
        }
        if (cur_t >= 0) flush_probe(cur_t);
        return;
    }
    St st = prepare(kb, ke);
    __syncthreads();
    STAMP_DECL
    for (int k = kb; k < ke; ++k) {
        if (MODE == M_FWD && (NM > 1 || st.t != cur_t)) {
            probe_strip(pr, 0, st.t);
            cur_t = st.t;
        }
        if (MODE == M_ADJ_PRB && st.q.valid && st.t != cur_t) {
            if (cur_t >= 0) flush_probe(cur_t);
#pragma unroll
            for (int m = 0; m < E; ++m) pr[m] = zero;
            cur_t = st.t;
        }
        if (!st.q.valid) {
            if (MODE == M_FWD) {   // skipped position: exact zeros (memset of ptychofft.cu:69)
#pragma unroll
                for (int km = 0; km < NM; ++km) {
                    c32* tile_out = (NM == 1 ? a.dst : a.dstm[km]) + (size_t)st.p * N * N;
#pragma unroll
                    for (int m = 0; m < E; ++m) tile_out[(size_t)(j0 + m * T) * N + x] = zero;
                }
            }
            __syncthreads();
            st = prepare(k + 1, ke);
            __syncthreads();
            continue;
        }
        const Pos q = st.q;
        const float w00 = (1.0f - q.fx) * (1.0f - q.fy), w01 = q.fx * (1.0f - q.fy);
        const float w10 = (1.0f - q.fx) * q.fy, w11 = q.fx * q.fy;
        // bilinear patch value of natural element m (row iy = j0 + m*T - pad); rows advance by T
        // in the window, modulo H, without a division per element.  Padding rows read stale but
        // finite window rows and are masked by the zero probe value / the select below.
        const int colw = st.Xa - X0 + c;
        int slot0 = (q.sy + j0 - ge.pad + 2 * H) % H;
        auto patch_at = [&](int slot) {
            const c32* r0 = win + slot * WC + colw;
            return r0[0] * w00 + r0[1] * w01 + r0[WC] * w10 + r0[WC + 1] * w11;   // kernels.cu:97-104 (row H mirrors row 0)
        };

        c32 v[E];
        c32 vbase[NM > 1 ? E : 1];   // patch values in step-0 slot order, shared by the NM modes
        St nx;
        if (MODE == M_FWD) {
            STAMP(0)      // loop head
            c32 nat[E];
            int slot = slot0;
#pragma unroll
            for (int m = 0; m < E; ++m) {
                nat[m] = patch_at(slot);
                slot += T;
                slot = slot >= H ? slot - H : slot;
            }
            F::from_natural(nat, v);
#ifdef PTY_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int s2 = 0; s2 < E; ++s2) asm volatile("" : "+v"(v[s2]));
#endif
            STAMP(1)      // gather: LDS taps + bilinear weights
            // (Round 3 tried requesting the rows that slide in for position k + 1 HERE, ahead of this position's 16 stores
            // -- vector memory operations complete in issue order, so the load behind the stores waits for them to drain
            // -- : the phase stamps moved as expected, the kernel got 6 % slower (0.465 -> 0.495 ms: 112 instead of 68
            // VGPRs and a worse schedule); profiles/r03/stamps.txt.)
            if (NM > 1) {
#pragma unroll
                for (int s2 = 0; s2 < E; ++s2) vbase[s2] = v[s2];
            }
#pragma unroll
            for (int s2 = 0; s2 < E; ++s2) v[s2] = cmul(pr[s2], v[s2]);
        } else {
            const c32* tile_in = a.src + (size_t)(a.natural_tiles ? st.p : (k - a.k_begin)) * N * N;
            fft.template load<0>(v, j0, [&](int i) { return tile_in[(size_t)i * N + x]; });
        }
        fft.template compute<0>(v);
        if (SPLIT && MODE == M_FWD) {
            // the radix-16 outputs go straight to rows 16 j0 + k1 of the strip (Stockham step 0)
            c32* tile_out = a.dst + (size_t)st.p * N * N;
#ifdef PTY_STAMPS
#pragma unroll
            for (int s2 = 0; s2 < E; ++s2) asm volatile("" : "+v"(v[s2]));
#endif
            STAMP(2)      // probe product + radix-16 step
            fft.template store<0>(v, j0, [&](int i, c32 val) { tile_out[(size_t)i * N + x] = val; });
            STAMP(3)      // store issue
            __syncthreads();   // every bilinear read of position k is done
            STAMP(4)      // barrier
            nx = prepare_issue(k + 1, ke);
            prepare_commit();
            STAMP(5)      // window update (global load of the rows that slide in + LDS store)
            __syncthreads();
            STAMP(6)      // barrier
            st = nx;
            continue;
        }
        int jz = j0;
        if (NM > 1) asm volatile("" : "+v"(jz));   // opaque copy: keeps the twiddle reads inside the position loop
#pragma unroll
        for (int km = 0; km < NM; ++km) {
            if (NM > 1) {
                if (km > 0) {   // next probe mode on the same patch values
#pragma unroll
                    for (int s2 = 0; s2 < E; ++s2) v[s2] = cmul(prnext[s2], vbase[s2]);
                    fft.template compute<0>(v);
                }
                if (km + 1 < NM) probe_strip(prnext, km + 1, st.t);   // in flight during this mode's exchange steps
            }
            if (P::NSTEP > 1) {
                if (MODE == M_FWD) { STAMP(2) }      // un-split forward: probe product + step 0 (+ next probe strip requested)
                fft.template store<0>(v, j0, [&](int i, c32 val) { lds[i * C + c] = val; });
                if (MODE == M_FWD) { STAMP(3) }      // exchange store
                __syncthreads();
                if (MODE == M_FWD) { STAMP(4) }      // barrier A
                if (MODE == M_FWD && km == 0) nx = prepare_issue(k + 1, ke);   // window of k is no longer read
                fft.template load<1>(v, j0, [&](int i) { return lds[i * C + c]; });
                if (P::NSTEP > 2) {
                    __syncthreads();
                    if constexpr (NM > 1 && P::NSTEP > 2) fft.template init_step<1>(jz, wtab);
                    fft.template compute<1>(v);
                    fft.template store<1>(v, j0, [&](int i, c32 val) { lds[i * C + c] = val; });
                    __syncthreads();
                    fft.template load<2>(v, j0, [&](int i) { return lds[i * C + c]; });
                }
                if constexpr (NM > 1 && P::NSTEP > 1) fft.template init_step<LAST>(jz, wtab);
                fft.template compute<LAST>(v);
            } else if (MODE == M_FWD && km == 0) {
                __syncthreads();
                nx = prepare_issue(k + 1, ke);
            }
            if (MODE == M_FWD) {
                STAMP(5)      // window request + exchange load + twiddles + last step
                c32* tile_out = (NM == 1 ? a.dst : a.dstm[km]) + (size_t)st.p * N * N;
                if (a.nt & 4)
                    fft.template store<LAST>(v, j0, [&](int i, c32 val) { __builtin_nontemporal_store(val, tile_out + (size_t)i * N + x); });
                else
                    fft.template store<LAST>(v, j0, [&](int i, c32 val) { tile_out[(size_t)i * N + x] = val; });
                STAMP(6)      // store issue
                if (km == NM - 1) prepare_commit();
                __syncthreads();   // exchange buffer / new window rows visible to everyone
                STAMP(7)      // window commit + barrier B
            }
        }
        if (MODE != M_FWD) {
            c32 nat[E];
            F::to_natural(v, nat);
            int slot = slot0;
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const int iy = j0 + m * T - ge.pad;
                const bool ok = col_ok && iy >= 0 && iy < ge.nprb;
                const c32 term = cmulc(nat[m], patch_at(slot));
                pr[m] += ok ? term : zero;
                slot += T;
                slot = slot >= H ? slot - H : slot;
            }
            __syncthreads();   // everyone is done with the window of k and the exchange buffer
            nx = prepare(k + 1, ke);
            __syncthreads();
        }
        st = nx;
    }
    if (MODE == M_ADJ_PRB && cur_t >= 0) flush_probe(cur_t);
    if (MODE == M_FWD) { STAMP_FLUSH(a.stamps) }
}

// ---------------------------------------------------------------------------
// Processing order of the positions: sorted by (angle, column bucket, row), skipped positions last.
// One launch, no library: every position is RANKED by counting the keys below it (ties broken by
// index, so the ranks are a permutation and the order is deterministic).  Workgroup (iblock, slice)
// counts, for its 256 positions, the smaller keys among the tiles {slice, slice + nslices, ...}; the
// partial counts meet in counts[] (integer atomics) and the last slice of an iblock to arrive writes
// order[rank] = position and clears the counters for the next call.  n^2 / 2 key compares: 1 us of
// work spread over the chip at 4096 positions, ~0.1 ms at the 32768 positions of a configs[3] shard
// (the radix sort it replaces took 30-60 us in several launches at 4096).
// ---------------------------------------------------------------------------
// pc = positions per chunk: the order is chunk-major over equal ranges of positions (one chunk: pc = total), so that
// the sorted ranks [c pc, (c + 1) pc) are exactly the positions of chunk c (chunked multi-mode line search).  Chunk
// and angle are both monotone in p, so (chunk, angle) orders like chunk * ptheta + angle.
__device__ __forceinline__ unsigned long long sort_key(const float* __restrict__ scan, const Geom& ge, const int p, const int pc) {
    const Pos q = decode_pos(scan, p, ge);
    const unsigned long long u = (unsigned long long)(p / pc) * (unsigned long long)ge.ptheta + (unsigned long long)(p / ge.nscan);
    if (!q.valid) return (u << 44) | 0xfffffffffffull;   // skipped positions: last of their (chunk, angle) group
    unsigned long long bx = (unsigned long long)(q.sx / kBucketPx), sy = (unsigned long long)q.sy;
    if (bx > 0x3ffffeull) bx = 0x3ffffeull;
    if (sy > 0x3fffffull) sy = 0x3fffffull;
    return (u << 44) | (bx << 22) | sy;
}

__global__ __launch_bounds__(256) void k_rank_positions(const float* __restrict__ scan, const Geom ge, const int total,
                                                        const int nslices, int* __restrict__ counts,
                                                        int* __restrict__ tickets, int* __restrict__ order, const int pc) {
    __shared__ unsigned long long keys[256];
    __shared__ int last;
    const int tid = threadIdx.x;
    const int iblock = blockIdx.x / nslices, slice = blockIdx.x % nslices;
    const int i = iblock * 256 + tid;
    const unsigned long long ki = i < total ? sort_key(scan, ge, i, pc) : ~0ull;
    const int ntiles = (total + 255) / 256;
    int cnt = 0;
    for (int tile = slice; tile < ntiles; tile += nslices) {
        const int jl = tile * 256 + tid;
        __syncthreads();
        keys[tid] = jl < total ? sort_key(scan, ge, jl, pc) : ~0ull;
        __syncthreads();
        const int j0 = tile * 256;
#pragma unroll 8
        for (int jj = 0; jj < 256; ++jj) {
            const unsigned long long kj = keys[jj];
            cnt += (kj < ki || (kj == ki && j0 + jj < i)) ? 1 : 0;   // j >= total: key ~0 and j > i, never counted
        }
    }
    if (i < total && cnt) atomicAdd(counts + i, cnt);
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(tickets + iblock, 1) == nslices - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (i < total) {
        const int rank = atomicExch(counts + i, 0);   // read the total and clear it for the next call
        order[rank] = i;
    }
    if (tid == 0) tickets[iblock] = 0;
}

// Column pass of the coarse cross-correlation with a fused arg-max (ptycho.py:204-207):
// inverse DFT over y of the slot's tiles, |.|, and per position the first maximum as a packed
// 64-bit key (value bits << 32 | ~flat index) merged with atomicMax.
// N = 256: the step-1 twiddles are re-read per tile from an LDS copy of the table, which brings the kernel under 128
// registers -- four waves per SIMD without spills (forcing four waves with the twiddles in registers spilled 30
// registers and ran 0.49 -> 0.75 ms).
template <int N>
__global__ __launch_bounds__(ColCfg<N>::NT, (N == 256 ? 4 : 1)) void k_cols_argmax(const c32* __restrict__ tiles, const c32* __restrict__ table,
                                                               unsigned long long* __restrict__ best, const int npos,
                                                               const int ngroups) {
    using P = Plan<N>;
    using F = Fft<P, +1>;
    constexpr int E = P::E, T = P::T, C = ColCfg<N>::C, NT = ColCfg<N>::NT;
    constexpr int LAST = P::NSTEP - 1;
    constexpr int NW = (NT + 63) / 64;
    constexpr bool TWLDS = N == 256;
    __shared__ c32 lds[N * C];
    __shared__ unsigned long long red[NW];
    __shared__ c32 wtab[TWLDS ? N : 1];
    const int tid = threadIdx.x;
    const int c = tid % C, j0 = tid / C;
    constexpr int nstrips = N / C;
    const int strip = blockIdx.x % nstrips, group = blockIdx.x / nstrips;
    const int x = strip * C + c;
    F fft;
    int jz = j0;
    if constexpr (TWLDS) {
        for (int o = tid; o < N; o += NT) wtab[o] = table[o];
        __syncthreads();
    } else {
        fft.init(j0, table);
    }
    for (int p = group; p < npos; p += ngroups) {
        if constexpr (TWLDS) asm volatile("" : "+v"(jz));
        const c32* tile = tiles + (size_t)p * N * N;
        c32 v[E];
        fft.template load<0>(v, j0, [&](int i) { return tile[(size_t)i * N + x]; });
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
            if constexpr (TWLDS) fft.template init_step<LAST>(jz, wtab);
            fft.template compute<LAST>(v);
        }
        c32 nat[E];
        F::to_natural(v, nat);
        // key = (bits of |v| << 32) | ~index, maximised: largest |v|, first index among equals -- |v| = sqrtf(|v|^2)
        // correctly rounded as np.abs gives it.  Only elements whose |v|^2 lies within 2^-21 of the thread's largest
        // can reach or tie its square root (distinct squares two ulps apart already round apart), so the 15-instruction
        // sqrtf runs for those only -- one per thread but for near ties.
        float m2[E], m2max = 0.0f;
#pragma unroll
        for (int m = 0; m < E; ++m) {
            m2[m] = nat[m].x * nat[m].x + nat[m].y * nat[m].y;
            m2max = fmaxf(m2max, m2[m]);
        }
        const float near = m2max * (1.0f - 4.76837158e-7f);
        unsigned long long key = 0ull;
#pragma unroll
        for (int m = 0; m < E; ++m) {
            if (m2[m] >= near) {
                const float mag = sqrtf(m2[m]);
                const unsigned idx = (unsigned)((j0 + m * T) * N + x);
                const unsigned long long k2 = ((unsigned long long)__float_as_uint(mag) << 32) | (unsigned long long)(0xffffffffu - idx);
                key = k2 > key ? k2 : key;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long o = __shfl_down(key, off, 64);
            key = o > key ? o : key;
        }
        if ((tid & 63) == 0) red[tid >> 6] = key;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; ++w) key = red[w] > key ? red[w] : key;
            atomicMax(best + p, key);
        }
        __syncthreads();
    }
}
