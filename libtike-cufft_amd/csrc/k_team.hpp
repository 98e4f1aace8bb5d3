// k_team.hpp -- experimental persistent XCD-team forward kernel
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
#pragma once

// ---------------------------------------------------------------------------
// Forward operator as ONE persistent launch of per-XCD teams (experimental,
// option "team").  The two-pass split costs an HBM round trip of the column-pass
// intermediate (2 x 8*ndet^2 B per position).  Here the workgroups that share an
// XCD (identified by HW_REG_XCC_ID, never by an assumed dispatch order) form a team:
// "column workers" (strip, ring slot) write the DFT-over-y strips of Q positions
// into a small ring in global memory, "row workers" read those tiles back with
// L1-bypassing loads and finish the DFT over x into g.  The ring (2 x Q tiles per
// XCD) is rewritten every other round, so it stays resident in that XCD's 4 MiB L2
// and the intermediate never travels to HBM.  Hand-off: producer stores ->
// s_waitcnt vmcnt(0) -> workgroup barrier -> one relaxed agent-scope atomic add on
// a per-round counter; consumer lane 0 polls the counter (bounded), then a
// workgroup barrier, then nontemporal loads.  Producer and consumer share one L2 by
// construction (same XCC id), so no L2 write-back is needed for visibility.
// Every spin is bounded; on timeout an abort word is set and all workers leave.
// ---------------------------------------------------------------------------
struct TeamArgs {
    const c32* f;
    c32* g;
    const c32* prb;
    const float* scan;
    const c32* table;
    Geom ge;
    const int* order;
    int total;
    c32* ring;          // [8 xcc][2][Q] tiles
    unsigned* ctrl;     // [0..7] team size, [8] arrived, [9] abort, [16 + x*2R + r] colDone, [.. + R + r] rowDone
    int R;              // rounds per team (upper bound)
    int Q;              // positions per round
    int strip0, nstrips;
    int xa, xb;
};

constexpr unsigned kSpinLimit = 1u << 22;

__device__ __forceinline__ bool team_wait(unsigned* ctr, unsigned need, unsigned* abort_word) {
    unsigned spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > kSpinLimit || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
            __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

template <int N>
__global__ __launch_bounds__(256) void k_fwd_team(const TeamArgs a) {
    using P = Plan<N>;
    using F = Fft<P, -1>;
    using L = RowLds<N>;
    constexpr int E = P::E, T = P::T, C = ColCfg<N>::C;
    static_assert(ColCfg<N>::NT == 256, "team kernel assumes 256-thread column workgroups");
    constexpr int LAST = P::NSTEP - 1;
    constexpr int WC = WinCfg<N>::WC, H = WinCfg<N>::H;
    constexpr int R0 = P::radix(0);
    constexpr int B = 256 / T;              // rows per row task
    constexpr int NB = N / B;               // row tasks per tile
    constexpr int COL_LDS = N * C + H * WC, ROW_LDS = B * L::FS;
    __shared__ c32 smem[COL_LDS > ROW_LDS ? COL_LDS : ROW_LDS];
    __shared__ int s_info[4];

    const int tid = threadIdx.x;
    const Geom ge = a.ge;
    const c32 zero = c32{0.0f, 0.0f};
    unsigned* abort_word = a.ctrl + 9;
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7;

    // ---- team formation --------------------------------------------------------------
    if (tid == 0) {
        const int m = (int)__hip_atomic_fetch_add(a.ctrl + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.ctrl + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool ok = team_wait(a.ctrl + 8, gridDim.x, abort_word);
        s_info[0] = m;
        s_info[1] = (int)__hip_atomic_load(a.ctrl + xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_info[2] = ok ? 1 : 0;
    }
    __syncthreads();
    const int m = s_info[0], S = s_info[1];
    if (!s_info[2] || S < 2) {
        if (tid == 0) __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const int per = (a.total + 7) / 8;
    const int kb = xcc * per;
    const int ke = kb + per < a.total ? kb + per : a.total;
    const int Q = a.Q;
    const int rounds = kb < ke ? (ke - kb + Q - 1) / Q : 0;
    int Wc = Q * a.nstrips;
    if (Wc > S / 2) Wc = S / 2;
    const int Wr = S - Wc;
    unsigned* colDone = a.ctrl + 16 + (size_t)xcc * 2 * a.R;
    unsigned* rowDone = colDone + a.R;
    c32* ring = a.ring + (size_t)xcc * 2 * Q * N * N;
    auto live = [&](int r) { const int left = ke - (kb + r * Q); return left < Q ? left : Q; };

    F fft;
    if (m < Wc) {
        // =================== column worker ============================================
        const int c = tid % C, j0 = tid / C;
        fft.init(j0, a.table);
        c32* lds = smem;
        c32* win = smem + N * C;
        for (int o = tid; o < H * WC; o += 256) win[o] = zero;
        const float cinv = 1.0f / (float)N;
        c32 pr[E];
        int cur_t = -1, cur_strip = -1;
        int t_w = -1, X0 = 0, Ylo = 0, Yhi = 0;
        __syncthreads();
        for (int r = 0; r < rounds; ++r) {
            if (r >= 2) {   // ring buffer (r & 1) must have been consumed
                if (tid == 0) s_info[3] = team_wait(rowDone + (r - 2), (unsigned)(live(r - 2) * NB), abort_word) ? 1 : 0;
                __syncthreads();
                if (!s_info[3]) return;
            }
            for (int ct = m; ct < Q * a.nstrips; ct += Wc) {
                const int slot = ct / a.nstrips, strip = ct % a.nstrips;
                const int k = kb + r * Q + slot;
                if (k >= ke) continue;
                const int x0 = (a.strip0 + strip) * C;
                const int x = x0 + c;
                const int ix = x - ge.pad;
                const bool col_ok = ix >= 0 && ix < ge.nprb;
                c32* tile = ring + ((size_t)(r & 1) * Q + slot) * N * N;
                const int p = a.order[k];
                const int t = p / ge.nscan;
                const Pos q = decode_pos(a.scan, p, ge);
                if (t != cur_t || strip != cur_strip) {
                    const c32* prb = a.prb + (size_t)t * ge.nprb * ge.nprb;
#pragma unroll
                    for (int b = 0; b < E / R0; ++b)
#pragma unroll
                        for (int tt = 0; tt < R0; ++tt) {
                            const int iy = j0 + b * T + tt * (N / R0) - ge.pad;
                            const bool ok = col_ok && iy >= 0 && iy < ge.nprb;
                            const c32 w = prb[ok ? ((size_t)iy * ge.nprb + ix) : 0];
                            pr[b * R0 + tt] = ok ? w * cinv : zero;
                        }
                    if (strip != cur_strip) t_w = -1;   // window belongs to another strip
                    cur_t = t; cur_strip = strip;
                }
                if (!q.valid) {
#pragma unroll
                    for (int mm = 0; mm < E; ++mm) tile[(size_t)(j0 + mm * T) * N + x] = zero;
                } else {
                    // ---- slide / re-anchor the cached object window -----------------------
                    const c32* ft = a.f + (size_t)t * ge.nz * ge.n;
                    const int Xa = q.sx + x0 - ge.pad;
                    const int Ra = q.sy, Rb = q.sy + ge.nprb + 1;
                    const bool colfit = (t == t_w) && Xa >= X0 && Xa + C < X0 + WC;
                    if (!colfit) {
                        t_w = t;
                        X0 = (q.sx / kBucketPx) * kBucketPx + x0 - ge.pad;
                        Ylo = Ra; Yhi = Ra;
                    } else if (Ra < Ylo || Ra > Yhi) {
                        Ylo = Ra; Yhi = Ra;
                    } else {
                        Ylo = Ra;
                    }
                    if (Rb > Yhi) {
                        const int cnt = (Rb - Yhi) * WC;
                        for (int o = tid; o < cnt; o += 256) {
                            const int Y = Yhi + o / WC, col = o % WC;
                            const int X = X0 + col;
                            const bool inb = Y < ge.nz && X >= 0 && X < ge.n;
                            const c32 val = ft[inb ? ((size_t)Y * ge.n + X) : 0];
                            win[(Y % H) * WC + col] = inb ? val : zero;
                        }
                        Yhi = Rb;
                    }
                    __syncthreads();
                    const float w00 = (1.0f - q.fx) * (1.0f - q.fy), w01 = q.fx * (1.0f - q.fy);
                    const float w10 = (1.0f - q.fx) * q.fy, w11 = q.fx * q.fy;
                    const int colw = Xa - X0 + c;
                    int slotw = (q.sy + j0 - ge.pad + 2 * H) % H;
                    c32 v[E], nat[E];
#pragma unroll
                    for (int mm = 0; mm < E; ++mm) {
                        const int s1 = slotw + 1 == H ? 0 : slotw + 1;
                        const c32* r0 = win + slotw * WC + colw;
                        const c32* r1 = win + s1 * WC + colw;
                        nat[mm] = r0[0] * w00 + r0[1] * w01 + r1[0] * w10 + r1[1] * w11;
                        slotw += T;
                        slotw = slotw >= H ? slotw - H : slotw;
                    }
                    F::from_natural(nat, v);
#pragma unroll
                    for (int s2 = 0; s2 < E; ++s2) v[s2] = cmul(pr[s2], v[s2]);
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
                    fft.template store<LAST>(v, j0, [&](int i, c32 val) { tile[(size_t)i * N + x] = val; });
                }
                // ---- publish: stores complete in L2, then one counter increment ----------
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(colDone + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        // =================== row worker ================================================
        const int wr = m - Wc;
        const int f = tid / T, j0 = tid % T;
        fft.init(j0, a.table);
        c32* lds = smem;
        for (int r = 0; r < rounds; ++r) {
            const int nlive = live(r);
            if (tid == 0) s_info[3] = team_wait(colDone + r, (unsigned)(nlive * a.nstrips), abort_word) ? 1 : 0;
            __syncthreads();
            if (!s_info[3]) return;
            for (int rt = wr; rt < Q * NB; rt += Wr) {
                const int slot = rt / NB, batch = rt % NB;
                const int k = kb + r * Q + slot;
                if (k >= ke) continue;
                const c32* tile = ring + ((size_t)(r & 1) * Q + slot) * N * N;
                const int p = a.order[k];
                const size_t rowoff = (size_t)(batch * B + f) * N;
                const c32* srow = tile + rowoff;
                c32* drow = a.g + (size_t)p * N * N + rowoff;
                c32 v[E];
                fft.template load<0>(v, j0, [&](int i) {
                    const c32 val = __builtin_nontemporal_load(srow + i);   // bypass this CU's L1
                    return (i >= a.xa && i < a.xb) ? val : zero;
                });
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
                }
                fft.template store<LAST>(v, j0, [&](int i, c32 val) { drow[i] = val; });
                __syncthreads();   // every lane has consumed its ring loads (and lds is free)
                if (tid == 0) __hip_atomic_fetch_add(rowDone + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
