// k_rows.hpp -- row pass (DFT over x) and its CG-fused variants, array reductions
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
#pragma once

// ---------------------------------------------------------------------------
// Row pass: DFT over x of contiguous rows, B = 256/T rows per workgroup step.
// ---------------------------------------------------------------------------
// (A full-width variant without the predicates below -- they put every request into a branch of its own -- was measured SLOWER by the
// wall clock: 4096 x 512^2 pair 12.45 -> 12.80 ms, 1024 x 1024^2 33.8 -> 34.15 ms; the kernel is HBM bound and 16 rows per workgroup at
// full occupancy hide the serialisation.  The fused CG stages, which carry more arithmetic per request, did gain: k_rows_fused<..., FW>.)
template <int N, int DIR>
__global__ __launch_bounds__(256) void k_rows(const RowArgs a) {
    using P = Plan<N>;
    using F = Fft<P, DIR>;
    using L = RowLds<N>;
    constexpr int E = P::E, T = P::T, B = 256 / T;
    constexpr int LAST = P::NSTEP - 1;
    __shared__ c32 lds[P::NSTEP > 1 ? B * L::FS : 1];

    const int tid = threadIdx.x;
    const int f = tid / T, j0 = tid % T;
    F fft;
    fft.init(j0, a.table);
    const c32 zero = c32{0.0f, 0.0f};
    const long long nb = (a.nrows + B - 1) / B;
    for (long long batch = blockIdx.x; batch < nb; batch += gridDim.x) {
        const long long r = batch * B + f;
        const bool ok = r < a.nrows;
        const long long tile = r / N;
        const c32* srow = a.src + (size_t)((a.tile_index && ok) ? (long long)a.tile_index[tile] : tile) * N * N + (size_t)(r % N) * N;
        c32* drow = a.dst + ((a.dst_indexed && a.tile_index && ok)
                                 ? (size_t)a.tile_index[tile] * N * N + (size_t)(r % N) * N
                                 : (size_t)r * N);
        c32 v[E];
        if (a.nt & 1)
            fft.template load<0>(v, j0, [&](int i) { return (ok && i >= a.xa && i < a.xb) ? __builtin_nontemporal_load(srow + i) : zero; });
        else
            fft.template load<0>(v, j0, [&](int i) { return (ok && i >= a.xa && i < a.xb) ? srow[i] : zero; });
        fft.template compute<0>(v);
        if (P::NSTEP > 1) {
            fft.template store<0>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
            row_sync<T>();
            fft.template load<1>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            if (P::NSTEP > 2) {
                row_sync<T>();
                fft.template compute<1>(v);
                fft.template store<1>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
                row_sync<T>();
                fft.template load<2>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            }
            fft.template compute<LAST>(v);
        }
        if (a.nt & 2)
            fft.template store<LAST>(v, j0, [&](int i, c32 val) {
                if (ok && i >= a.wa && i < a.wb) __builtin_nontemporal_store(val, drow + i);
            });
        else
            fft.template store<LAST>(v, j0, [&](int i, c32 val) {
                if (ok && i >= a.wa && i < a.wb) drow[i] = val;
            });
        if (P::NSTEP > 1) row_sync<T>();
    }
}

// ---------------------------------------------------------------------------
// Split variant for N = 256 (= 16 x 16): the column kernels do only ONE radix-16 step of
// the DFT over y (thread local, no exchange, no barrier); the other radix-16 step runs
// here, across the 16 rows {jp + 16 t} a workgroup owns, where the kernel is HBM bound
// and has ALU slack.  Stockham algebra: step 0 maps x[j + 16 t] -> y0[16 j + k1]; step 1
// maps y0[j' + 16 t] * W^{j' t} -> X[j' + 16 t'].
//   DIR < 0 (forward): rows j' + 16 t of the intermediate -> twiddle -> radix 16 over t
//                      -> transpose through LDS -> DFT over x -> final rows j' + 16 t'.
//   DIR > 0 (adjoint): rows j' + 16 f of g -> IDFT over x -> transpose -> radix 16 over
//                      the 16 rows (step 0, no twiddle) -> intermediate rows 16 j' + k1.
// ---------------------------------------------------------------------------
template <int N, int DIR>
__global__ __launch_bounds__(256) void k_rows_split(const RowArgs a) {
    static_assert(N == 256, "split row pass is written for the 16 x 16 plan");
    using P = Plan<N>;
    using F = Fft<P, DIR>;
    using L = RowLds<N>;
    constexpr int E = P::E, T = P::T, B = 256 / T;   // 16, 16, 16
    constexpr int LAST = P::NSTEP - 1;
    __shared__ c32 lds[B * L::FS];

    const int tid = threadIdx.x;
    const int f = tid / T, j0 = tid % T;
    F fft;
    fft.init(j0, a.table);
    const c32 zero = c32{0.0f, 0.0f};
    const long long nitems = (a.nrows / N) * 16;
    for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
        const long long tile = item / 16;
        const int jp = (int)(item % 16);
        const long long stile = a.tile_index ? (long long)a.tile_index[tile] : tile;
        const c32* sbase = a.src + (size_t)stile * N * N;
        c32* dbase = a.dst + (size_t)((a.dst_indexed && a.tile_index) ? stile : tile) * N * N;
        c32 v[E];
        if (DIR < 0) {
            // ---- second radix-16 step of the DFT over y, thread = column x ------------------
            const int x = tid;
            const bool colok = x >= a.xa && x < a.xb;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const c32 val = __builtin_nontemporal_load(sbase + (size_t)(jp + 16 * t) * N + x);
                v[t] = colok ? val : zero;
            }
#pragma unroll
            for (int t = 1; t < 16; ++t) v[t] = cmul(v[t], a.table[(jp * t) & (N - 1)]);   // uniform index
            fft_reg<16, -1>(v);
#pragma unroll
            for (int t = 0; t < 16; ++t) lds[L::at(t, x)] = v[brev(t, 4)];
            __syncthreads();
            // ---- DFT over x of row f (final row jp + 16 f) ---------------------------------
            fft.template load<0>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            fft.template compute<0>(v);
            fft.template store<0>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
            row_sync<T>();   // exchange inside row f: the row's 16 threads are lanes of one wave
            fft.template load<1>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            fft.template compute<LAST>(v);
            c32* drow = dbase + (size_t)(jp + 16 * f) * N;
            fft.template store<LAST>(v, j0, [&](int i, c32 val) { __builtin_nontemporal_store(val, drow + i); });
            __syncthreads();
        } else {
            // ---- IDFT over x of row jp + 16 f of g ------------------------------------------
            const c32* srow = sbase + (size_t)(jp + 16 * f) * N;
            fft.template load<0>(v, j0, [&](int i) { return __builtin_nontemporal_load(srow + i); });
            fft.template compute<0>(v);
            fft.template store<0>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
            row_sync<T>();
            fft.template load<1>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            fft.template compute<LAST>(v);
            row_sync<T>();   // every exchange read of row f is done before the row is rewritten
            fft.template store<LAST>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
            __syncthreads();
            // ---- first radix-16 step of the IDFT over y, thread = column x -------------------
            const int x = tid;
#pragma unroll
            for (int t = 0; t < 16; ++t) v[t] = lds[L::at(t, x)];
            fft_reg<16, +1>(v);
            if (x >= a.wa && x < a.wb) {
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1)
                    __builtin_nontemporal_store(v[brev(k1, 4)], dbase + (size_t)(16 * jp + k1) * N + x);
            }
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------
// Row pass fused with the elementwise stages of the CG loop
// (src/libtike/cufft/ptycho.py:325-393 launches each of them as separate CuPy
// kernels over farplane-sized temporaries).  Input rows are column-pass
// intermediates (DFT over y done); the farplane exists only in registers.
//   EP_STATS      I = |g|^2 ; sums += [sum sqrt(I d), sum I]             (ptycho.py:330-343)
//   EP_PROJECT    fpsi = (g s)(1/s'), I' = I s^2,
//                 r = fpsi - sqrt(d) fpsi / (sqrt(I') + 1e-32), cost += (sqrt I' - sqrt d)^2,
//                 out row = IDFT_x(r)                                    (ptycho.py:344-356, 310)
//   EP_LINESEARCH t1 = s g1, t2 = g2: p1,p2,p3 (ptycho.py:383-391) and the cost
//                 sum (sqrt|p1 + y^2 p2 + y p3| - sqrt d)^2 for y = gamma0 * 2^-j, j < ncand,
//                 plus f(p1) -- every trial of line_search_sqr in one pass (ptycho.py:253-281)
// ---------------------------------------------------------------------------
// v_sqrt_f32 / v_rcp_f32 (1 ulp).  The correctly rounded sqrtf()/division expand to ~15
// instructions each; the line search evaluates 17 square roots per farplane element.
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

// An unconditional global load whose value is masked afterwards (m = all ones or zero, made opaque by the caller so that the
// AND does not turn back into a select).  `pred ? load(p) : zero` puts every request into a branch of its own
// (s_and_saveexec / s_cbranch_execz), and the branches pin each request behind the previous one's wait: the fused row
// stages showed 50-100 branches and at most 2-4 requests in flight per thread.
__device__ __forceinline__ c32 load_masked(const c32* __restrict__ p, unsigned m) {
    const c32 val = __builtin_nontemporal_load(p);
    return c32{__uint_as_float(__float_as_uint(val.x) & m), __uint_as_float(__float_as_uint(val.y) & m)};
}
__device__ __forceinline__ float load_masked(const float* __restrict__ p, unsigned m) {
    return __uint_as_float(__float_as_uint(__builtin_nontemporal_load(p)) & m);
}

enum RowEp { EP_STATS = 1, EP_PROJECT = 2, EP_LINESEARCH = 3, EP_CROSS = 6, EP_LINESEARCH_M = 7, EP_STATS_M = 8 };
constexpr int kMaxModes = 8;   // EP_LINESEARCH_M: slot pairs (2k, 2k+1), k < nmodes
constexpr int kMaxCand = 16;
constexpr int kLsGroupsRows = 7;   // = kLsGroupsMax (k_cg_small.hpp): groups of 16 step lengths one pass may price

struct RowFusedArgs {
    const c32* s1;
    const c32* s2;
    c32* out;
    const float* data;
    const c32* table;
    long long nrows;
    double* sums;        // EP_STATS: [2]; EP_PROJECT: [1] cost; EP_LINESEARCH: [ncand + 1]
    const double* ab;    // device scalars a, b of ptycho.py:342-343 (nullptr: scale 1)
    float gamma0;
    int ncand;
    int xa, xb;          // columns outside [xa, xb) of the inputs are zero (never written)
    // multi-mode variants (arrays are float32 [positions][ndet][ndet])
    const float* inten;  // EP_PROJECT: summed intensity of all modes (nullptr: single mode, |g|^2)
    float* acc1;         // EP_STATS_M: summed intensity (may be null)
    int first;           // 1: overwrite the arrays, 0: add to them; EP_PROJECT with inten: 1 = the slot was
                         // made with the probe BEFORE its rescale (fp = (g s)(1/s) as in the single-mode path)
    // EP_LINESEARCH_M: t1_k = s * DFT_x(sm[2k]), t2_k = DFT_x(sm[2k+1]); p1 = inten if given, else sum_k |t1_k|^2
    const c32* sm[2 * kMaxModes];
    int nmodes;
    c32* ip;             // EP_CROSS: image product u1 * conj(u2), [positions][ndet][ndet]
    // device-resident CG state (k_cg_small.hpp; nullptr: the host arguments above are used).  Line search: the
    // pass returns at once if the search is resolved, reads gamma0 / ncand / ngroups from the state and adds the
    // costs of group grp to st[PTYCHO_ST_COSTS + 17 grp ...]; EP_CROSS: gamma = *gamma_dev.
    double* st;
    const double* gamma_dev;
    // EP_PROJECT: if set, max |out| (float bits in the low word, as k_cg_absmax leaves it) -- the deterministic
    // adjoint that follows sizes its fixed point from it and skips its own pass over the slot
    double* maxword;
    // cross-workgroup sums in a fixed order (ptycho_common.hpp); the last workgroup adds the totals to sums
    // (overwrite = 1: stores them -- the native CG stages, which then need no zero fill)
    FoldBuf fold;
    int overwrite;
    // line search on the device-resident state, single GPU: the last workgroup also replays line_search_sqr on the
    // totals (ls_decide_dev), which saves the one-thread decision kernel between two passes.  decide = 0: off
    int decide, decide_which, decide_gamma_word, decide_next;
    // EP_CROSS: if set, best[0 .. nbest) <- 0 for the arg-max column pass that follows (saves its zero fill)
    unsigned long long* best_zero;
    int nbest;
};

// ---- line_search_sqr's accept / shrink loop on the device (ptycho.py:253-281) -------------------------------
// A pass prices ngroups x ncand step lengths gamma0 2^-j in one sweep over the two work buffers; candidate j of group
// grp lands in costs[grp * 17 + j], f(p1) in costs[grp * 17 + ncand].  Accept the first step whose float32 cost is
// not above f(p1), fail below 1e-32.  which: hint slot; gamma_word: where 0.5 * step goes (ptycho.py:393,461);
// next_ngroups: size of the pass that follows if this one did not resolve the search (0: none follows).
__device__ inline void ls_decide_dev(double* __restrict__ st, const int which, const int gamma_word, const int next_ngroups) {
    if (st[PTYCHO_ST_LS_RESOLVED] != 0.0) return;
    const int ngroups = (int)st[PTYCHO_ST_LS_NGROUPS];
    const int ncand = (int)st[PTYCHO_ST_LS_NCAND];
    int tried = (int)st[PTYCHO_ST_LS_TRIED];
    double step = st[PTYCHO_ST_LS_GAMMA0];
    bool done = false;
    for (int grp = 0; grp < ngroups && !done; ++grp) {
        const double* c = st + PTYCHO_ST_COSTS + grp * (kMaxCand + 1);
        const float fp1 = (float)c[ncand];                    // the reference compares float32 costs
        for (int j = 0; j < ncand; ++j) {
            if (!((float)c[j] > fp1)) {
                st[gamma_word] = 0.5 * step;
                st[PTYCHO_ST_HINT + which] = (double)(tried + j);
                done = true;
                break;
            }
            if (step < 1e-32) {                               // "Line search failed for conjugate gradient."
                st[gamma_word] = 0.0;
                st[PTYCHO_ST_HINT + which] = 14.0;
                st[PTYCHO_ST_LS_FAILED] += 1.0;
                done = true;
                break;
            }
            step *= 0.5;
        }
        if (!done) tried += ncand;
    }
    if (!done && next_ngroups == 0) {   // cannot happen with 2..16 + 16 + 32 + 64 (or 2..16 + 112) step lengths (2^-106 < 1e-32); fail safe
        st[gamma_word] = 0.0;
        st[PTYCHO_ST_LS_FAILED] += 1.0;
        done = true;
    }
    st[PTYCHO_ST_LS_RESOLVED] = done ? 1.0 : 0.0;
    st[PTYCHO_ST_LS_TRIED] = (double)tried;
    st[PTYCHO_ST_LS_GAMMA0] = step;
    st[PTYCHO_ST_LS_NCAND] = (double)kMaxCand;
    st[PTYCHO_ST_LS_NGROUPS] = (double)next_ngroups;
}


// Waves per SIMD the register allocation must leave room for.  Left alone the compiler takes 270..310 registers
// (VGPR + AGPR) for the two-input stages, i.e. ONE wave per SIMD, and these kernels are bound by their VALU /
// latency chains: two waves per SIMD (<= 256 registers) took the ndet = 256 line search from 2.0 to 1.3 ms.
// (The plain row pass k_rows<512> is the opposite case: forced from 2 to 3 or 4 waves per SIMD, with the twiddles
// in registers or re-read from LDS, it went 3.24 -> 3.41 / 3.43 ms; it keeps the compiler's 185 registers.)
#ifndef PTY_AB
#define PTY_AB 0
#endif
template <int N, int EP>
constexpr int fused_min_waves() {
    if ((PTY_AB & 1) && N == 512 && EP == EP_LINESEARCH_M) return 1;   // A/B: no spills at one wave per SIMD
    if (!is_pow2(N)) return 1;   // 48 ... 192 with an odd factor: 12-28 points per thread, no register cap (no spills)
    return (N <= 512 && (EP == EP_LINESEARCH || EP == EP_LINESEARCH_M || EP == EP_PROJECT || EP == EP_CROSS)) ? 2
           : (N == 512 && (EP == EP_STATS || EP == EP_STATS_M)) ? 3 : 1;
}
template <>
constexpr int fused_min_waves<256, EP_STATS>() { return 4;   // 0.527 -> 0.505 ms (PROJECT at three waves spills: 1.03 -> 1.31)
}
// FW: the launch covers the full width (xa = 0, xb = N -- the probe fills the detector): no column predicate, loads unconditional
template <int N, int EP, bool FW = false>
__global__ __launch_bounds__(256, (fused_min_waves<N, EP>())) void k_rows_fused(const RowFusedArgs a) {
    using P = Plan<N>;
    using F = Fft<P, -1>;
    using L = RowLds<N>;
    constexpr int E = P::E, T = P::T, B = 256 / T;
    constexpr int LAST = P::NSTEP - 1;
    constexpr bool LS = EP == EP_LINESEARCH || EP == EP_LINESEARCH_M;
    constexpr int NACC = (EP == EP_STATS || EP == EP_STATS_M) ? 2 : (LS ? kMaxCand + 1 : 1);
    __shared__ c32 lds[P::NSTEP > 1 ? B * L::FS : 1];
    __shared__ double red[4 * NACC];
    __shared__ double vals[EP == EP_CROSS ? 1 : kFoldStride];    // this workgroup's sums, in the layout of the output
    __shared__ double fscr[EP == EP_CROSS ? 1 : 256];
    __shared__ c32 stash[(EP == EP_CROSS || EP == EP_LINESEARCH_M) ? E * 256 : 1];
    // single-mode line search with several groups of 16 step lengths in ONE sweep: per batch every wave folds its
    // partial costs of a group into these float64 accumulators (one row per wave) instead of keeping 17 registers per
    // group alive across the batch loop; the inputs are read and transformed once for all groups
    __shared__ double gacc[EP == EP_LINESEARCH ? 4 * kLsGroupsRows * (kMaxCand + 1) : 1];
    // three-step plans at two waves per SIMD: the 2 x 16 inter-step twiddles do not stay in registers across the
    // batch loop (64 VGPRs); each step re-reads its set from an LDS copy of the table (Fft::init_step)
    constexpr bool TWLDS = (P::NSTEP > 2 && fused_min_waves<N, EP>() > 1) || (N <= 256 && P::NSTEP > 1 && (EP == EP_LINESEARCH || EP == EP_LINESEARCH_M));
    __shared__ c32 wtab[TWLDS ? N : 1];

    const int tid = threadIdx.x;
    const int f = tid / T, j0 = tid % T;
    int jz = j0;   // TWLDS: made opaque once per batch, which keeps the twiddle reads inside the batch loop
    F fft;
    if constexpr (TWLDS) {
        for (int o = tid; o < N; o += 256) wtab[o] = a.table[o];
        __syncthreads();
    } else {
        fft.init(j0, a.table);
    }
    const c32 zero = c32{0.0f, 0.0f};
    if (EP != EP_CROSS && tid < kFoldStride) vals[tid] = 0.0;
    float acc[NACC];
    c32 acc2[LS ? NACC : 1];   // line search: even / odd pixel partial sums
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
#pragma unroll
    for (int i = 0; i < (LS ? NACC : 1); ++i) acc2[i] = c32{0.0f, 0.0f};

    // scale factors of ptycho.py:344-351 in float32, as the reference computes them
    float s = 1.0f, sinv = 1.0f;
    if (a.ab) {
        const float af = (float)a.ab[0], bf = (float)a.ab[1];
        s = af / bf;
        sinv = bf / af;
    }

    float vmax2 = 0.0f;   // EP_PROJECT with maxword: running max |out|^2 of this thread
    float gamma0 = a.gamma0;
    int ncand = a.ncand, ngroups = 1;
    double* sums = a.sums;
    if (LS && a.st) {
        if (a.st[PTYCHO_ST_LS_RESOLVED] != 0.0) return;   // uniform over the grid
        gamma0 = (float)a.st[PTYCHO_ST_LS_GAMMA0];
        ncand = (int)a.st[PTYCHO_ST_LS_NCAND];
        ngroups = (int)a.st[PTYCHO_ST_LS_NGROUPS];
        sums = a.st + PTYCHO_ST_COSTS;
    }
    if (EP == EP_CROSS && a.gamma_dev) gamma0 = (float)*a.gamma_dev;
    if (EP == EP_CROSS && a.best_zero) {
        for (int i = blockIdx.x * 256 + tid; i < a.nbest; i += gridDim.x * 256) a.best_zero[i] = 0ull;
    }

    // forward DFT over x of one row held as step-0 inputs in v; result in natural order
    auto fwd_row = [&](c32* v, c32* nat) {
        fft.template compute<0>(v);
        if (P::NSTEP > 1) {
            fft.template store<0>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
            row_sync<T>();
            fft.template load<1>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            if (P::NSTEP > 2) {
                row_sync<T>();
                if constexpr (TWLDS) fft.template init_step<1>(jz, wtab);
                fft.template compute<1>(v);
                fft.template store<1>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
                row_sync<T>();
                fft.template load<2>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
            }
            if constexpr (TWLDS) fft.template init_step<LAST>(jz, wtab);
            fft.template compute<LAST>(v);
            row_sync<T>();   // lds free for the next transform
        }
        F::to_natural(v, nat);
    };

    const long long nb = (a.nrows + B - 1) / B;
    const bool multi = EP == EP_LINESEARCH && ngroups > 1;   // uniform; full groups (ncand = 16) by construction
    if (EP == EP_LINESEARCH && multi) {
        for (int o = tid; o < 4 * kLsGroupsRows * (kMaxCand + 1); o += 256) gacc[o] = 0.0;
        __syncthreads();
    }
    const int nsweeps = multi ? 1 : ngroups;
    for (int grp = 0; grp < nsweeps; ++grp) {   // multi-mode line search: one sweep per group of 16 step lengths (else one pass)
    const float gam_first = gamma0 * exp2f(-16.0f * (float)grp);
    if (grp > 0) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int i = 0; i < (LS ? NACC : 1); ++i) acc2[i] = c32{0.0f, 0.0f};
    }
    for (long long batch = blockIdx.x; batch < nb; batch += gridDim.x) {
        const long long r = batch * B + f;
        const bool ok = r < a.nrows;
        if constexpr (TWLDS) asm volatile("" : "+v"(jz));
        // addresses = wave-uniform batch base (scalar registers) + 32-bit per-thread offset: the
        // loads / stores take the saddr form and no 64-bit address lives in a VGPR
        const size_t boff = (size_t)batch * B * N;
        const unsigned fN = (unsigned)(f * N);
        const unsigned fNs = ok ? fN : 0u;   // FW: a row past the end reads row 0 of its batch (always there) ...
        unsigned rowm = ok ? 0xffffffffu : 0u;   // ... and masks it
        if constexpr (FW) asm volatile("" : "+v"(rowm));
        c32 v[E], g1[E];
        const c32* __restrict__ first_src = (EP == EP_LINESEARCH_M || EP == EP_STATS_M) ? a.sm[0] : a.s1;
        auto request = [&](const c32* __restrict__ src, c32* dstv) {
            fft.template load<0>(dstv, j0, [&](int i) { if constexpr (FW) return load_masked(src + boff + (fNs + (unsigned)i), rowm); else return (ok && i >= a.xa && i < a.xb) ? __builtin_nontemporal_load(src + boff + (fN + (unsigned)i)) : zero; });
        };
        request(first_src, v);
        // (Round 4 tried requesting the 2 nmodes input rows of the multi-mode line search one transform AHEAD, ping-pong between two
        // register sets: 124.8 -> 125.6-126.2 ms per configs[2] iteration -- the kernel is not waiting on its loads (SQ counters: VALU
        // active 47 % of the SIMD cycles, issue stalls 24 %, waits 30 %); a first form that copied the prefetched row and kept the
        // masked loads exploded to 581-1116 spilled registers and 441 ms.  profiles/r04/cfg3_experiments.txt.)
        fwd_row(v, g1);
        float d[E];
        auto load_data = [&]() {
#pragma unroll
            for (int m = 0; m < E; ++m) {
                if constexpr (FW) d[m] = load_masked(a.data + boff + (fNs + (unsigned)(j0 + m * T)), rowm);
                else d[m] = ok ? __builtin_nontemporal_load(a.data + boff + (fN + (unsigned)(j0 + m * T))) : 0.0f;
            }
        };
        if (EP == EP_STATS || EP == EP_PROJECT) load_data();
        if (EP == EP_CROSS) {
            // position correction (ptycho.py:398-403,198-204): u1 = G psi, u2 = G(psi + gamma dpsi)
            // = u1 + gamma G dpsi (ones probe); image product u1 conj(u2) is kept for the zoomed
            // DFT and its inverse row DFT goes back into the slot (column pass + arg-max follow).
            c32 g2[E], rr[E];
            // u1 waits in LDS (private slots, no barrier) while the second row is transformed: the
            // kernel then fits 256 VGPRs and two waves per SIMD instead of one
#pragma unroll
            for (int m = 0; m < E; ++m) stash[m * 256 + tid] = g1[m];
            fft.template load<0>(v, j0, [&](int i) { if constexpr (FW) return load_masked(a.s2 + boff + (fNs + (unsigned)i), rowm); else return (ok && i >= a.xa && i < a.xb) ? __builtin_nontemporal_load(a.s2 + boff + (fN + (unsigned)i)) : zero; });
            fwd_row(v, g2);
#pragma unroll
            for (int m = 0; m < E; ++m) g1[m] = stash[m * 256 + tid];
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const c32 u2 = g1[m] + g2[m] * gamma0;
                rr[m] = cmulc(g1[m], u2);
                if (ok) __builtin_nontemporal_store(rr[m], a.ip + boff + (fN + (unsigned)(j0 + m * T)));
            }
            F::from_natural(rr, v);
            fft.template compute_rev<0>(v);
            if (P::NSTEP > 1) {
                fft.template store<0>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
                row_sync<T>();
                fft.template load<1>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
                if (P::NSTEP > 2) {
                    row_sync<T>();
                    if constexpr (TWLDS) fft.template init_step<1>(jz, wtab);
                    fft.template compute_rev<1>(v);
                    fft.template store<1>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
                    row_sync<T>();
                    fft.template load<2>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
                }
                if constexpr (TWLDS) fft.template init_step<LAST>(jz, wtab);
                fft.template compute_rev<LAST>(v);
            }
            fft.template store<LAST>(v, j0, [&](int i, c32 val) {
                if (ok) __builtin_nontemporal_store(val, a.out + boff + (fN + (unsigned)i));
            });
            if (P::NSTEP > 1) row_sync<T>();
            continue;
        }

        if (EP == EP_STATS) {
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const float I = g1[m].x * g1[m].x + g1[m].y * g1[m].y;
                acc[0] += fsqrt(I * d[m]);
                acc[1] += I;
            }
        } else if (EP == EP_STATS_M) {
            // summed intensity of nmodes slots sm[0..nmodes) (ptycho.py:329-333): written to acc1 if
            // given, and the statistics of :342-343 added to sums if given -- one pass instead of
            // one accumulate pass per mode plus an array reduction
            float I[E];
#pragma unroll
            for (int m = 0; m < E; ++m) I[m] = g1[m].x * g1[m].x + g1[m].y * g1[m].y;
            for (int k = 1; k < a.nmodes; ++k) {
                const c32* __restrict__ src = a.sm[k];
                fft.template load<0>(v, j0, [&](int i) { if constexpr (FW) return load_masked(src + boff + (fNs + (unsigned)i), rowm); else return (ok && i >= a.xa && i < a.xb) ? __builtin_nontemporal_load(src + boff + (fN + (unsigned)i)) : zero; });
                fwd_row(v, g1);
#pragma unroll
                for (int m = 0; m < E; ++m) I[m] += g1[m].x * g1[m].x + g1[m].y * g1[m].y;
            }
            if (a.acc1 && ok) {
#pragma unroll
                for (int m = 0; m < E; ++m) a.acc1[boff + (fN + (unsigned)(j0 + m * T))] = I[m];
            }
            if (a.sums) {
                load_data();
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    acc[0] += fsqrt(I[m] * d[m]);
                    acc[1] += I[m];
                }
            }
        } else if (EP == EP_PROJECT) {
            const float s2 = s * s;
            c32 rr[E];
#pragma unroll
            for (int m = 0; m < E; ++m) {
                // single mode: S comes from the unscaled probe -> I' = |g|^2 s^2, fpsi = (g s)(1/s');
                // multi mode: S comes from the rescaled probe and I is the summed intensity array
                const float I = a.inten ? (ok ? a.inten[boff + (fN + (unsigned)(j0 + m * T))] : 0.0f) * s2
                                        : (g1[m].x * g1[m].x + g1[m].y * g1[m].y) * s2;
                const c32 fp = (a.inten && !a.first) ? g1[m] * sinv : (g1[m] * s) * sinv;
                const float sd = fsqrt(d[m]), sI = fsqrt(I);
                rr[m] = fp - (fp * sd) * frcp(sI + 1e-32f);
                const float df = sI - sd;
                acc[0] += ok ? df * df : 0.0f;
            }
            // inverse DFT over x of the projected row, same twiddle registers (conjugated)
            F::from_natural(rr, v);
            fft.template compute_rev<0>(v);
            if (P::NSTEP > 1) {
                fft.template store<0>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
                row_sync<T>();
                fft.template load<1>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
                if (P::NSTEP > 2) {
                    row_sync<T>();
                    if constexpr (TWLDS) fft.template init_step<1>(jz, wtab);
                    fft.template compute_rev<1>(v);
                    fft.template store<1>(v, j0, [&](int i, c32 val) { lds[L::at(f, i)] = val; });
                    row_sync<T>();
                    fft.template load<2>(v, j0, [&](int i) { return lds[L::at(f, i)]; });
                }
                if constexpr (TWLDS) fft.template init_step<LAST>(jz, wtab);
                fft.template compute_rev<LAST>(v);
            }
            if (a.maxword && ok) {
#pragma unroll
                for (int m = 0; m < E; ++m) vmax2 = fmaxf(vmax2, v[m].x * v[m].x + v[m].y * v[m].y);
            }
            fft.template store<LAST>(v, j0, [&](int i, c32 val) {
                if (ok) __builtin_nontemporal_store(val, a.out + boff + (fN + (unsigned)i));
            });
            if (P::NSTEP > 1) row_sync<T>();
        } else if (EP == EP_LINESEARCH_M) {
            // multi-mode line search (ptycho.py:383-393 with the sums over k of :386-391): the terms
            // p1, p2, p3 are accumulated over the modes in registers, never stored
            float p1[E], p2[E], p3[E];
#pragma unroll
            for (int m = 0; m < E; ++m) { p1[m] = 0.0f; p2[m] = 0.0f; p3[m] = 0.0f; }
            for (int k = 0; k < a.nmodes; ++k) {
                if (k > 0) {
                    request(a.sm[2 * k], v);
                    fwd_row(v, g1);
                }
#pragma unroll
                for (int m = 0; m < E; ++m) stash[m * 256 + tid] = g1[m];   // t1 waits in LDS (private slots)
                c32 g2[E];
                request(a.sm[2 * k + 1], v);
                fwd_row(v, g2);
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    const c32 t1 = stash[m * 256 + tid] * s;
                    p1[m] += t1.x * t1.x + t1.y * t1.y;
                    p2[m] += g2[m].x * g2[m].x + g2[m].y * g2[m].y;
                    p3[m] += 2.0f * (t1.x * g2[m].x + t1.y * g2[m].y);
                }
            }
            load_data();
            if (a.inten) {
#pragma unroll
                for (int m = 0; m < E; ++m) p1[m] = ok ? a.inten[boff + (fN + (unsigned)(j0 + m * T))] : 0.0f;
            }
#pragma unroll
            for (int m = 0; m < E; m += 2) {
                const c32 q1 = c32{p1[m], p1[m + 1]}, q2 = c32{p2[m], p2[m + 1]}, q3 = c32{p3[m], p3[m + 1]};
                const c32 sd = c32{fsqrt(d[m]), fsqrt(d[m + 1])};
                c32 df = c32{fsqrt(fabsf(q1.x)), fsqrt(fabsf(q1.y))} - sd;
                acc2[kMaxCand] += df * df;
                float gam = gam_first;
#pragma unroll
                for (int j0c = 0; j0c < kMaxCand; j0c += 4) {
                    if (j0c < ncand) {
#pragma unroll
                        for (int j = j0c; j < j0c + 4; ++j) {
                            const c32 xx = q1 + q2 * (gam * gam) + q3 * gam;
                            df = c32{fsqrt(fabsf(xx.x)), fsqrt(fabsf(xx.y))} - sd;
                            acc2[j] += df * df;
                            gam *= 0.5f;
                        }
                    }
                }
            }
        } else {   // EP_LINESEARCH
            c32 g2[E];
            fft.template load<0>(v, j0, [&](int i) { if constexpr (FW) return load_masked(a.s2 + boff + (fNs + (unsigned)i), rowm); else return (ok && i >= a.xa && i < a.xb) ? __builtin_nontemporal_load(a.s2 + boff + (fN + (unsigned)i)) : zero; });
            fwd_row(v, g2);
            load_data();   // after the second transform: keeps 16 registers free during it
            // two detector pixels per step in packed float32 (v_pk_fma_f32 / v_pk_add_f32), four
            // step lengths per uniform branch: per trial and pixel 2 FMA for p1 + y^2 p2 + y p3,
            // one v_sqrt_f32, one subtract, one FMA into the cost
            const int ginner = multi ? ngroups : 1;
            for (int gi = 0; gi < ginner; ++gi) {
            const float gfirst = multi ? gamma0 * exp2f(-16.0f * (float)gi) : gam_first;
            if (multi) {
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc2[i] = c32{0.0f, 0.0f};
            }
#pragma unroll
            for (int m = 0; m < E; m += 2) {
                const c32 ta = g1[m] * s, tb = g1[m + 1] * s;
                const c32 p1 = c32{ta.x * ta.x + ta.y * ta.y, tb.x * tb.x + tb.y * tb.y};
                const c32 p2 = c32{g2[m].x * g2[m].x + g2[m].y * g2[m].y, g2[m + 1].x * g2[m + 1].x + g2[m + 1].y * g2[m + 1].y};
                const c32 p3 = c32{2.0f * (ta.x * g2[m].x + ta.y * g2[m].y), 2.0f * (tb.x * g2[m + 1].x + tb.y * g2[m + 1].y)};
                const c32 sd = c32{fsqrt(d[m]), fsqrt(d[m + 1])};
                c32 df = c32{fsqrt(fabsf(p1.x)), fsqrt(fabsf(p1.y))} - sd;
                acc2[kMaxCand] += df * df;
                float gam = gfirst;
#pragma unroll
                for (int j0c = 0; j0c < kMaxCand; j0c += 4) {
                    if (j0c < ncand) {
#pragma unroll
                        for (int j = j0c; j < j0c + 4; ++j) {
                            const c32 xx = p1 + p2 * (gam * gam) + p3 * gam;
                            df = c32{fsqrt(fabsf(xx.x)), fsqrt(fabsf(xx.y))} - sd;
                            acc2[j] += df * df;
                            gam *= 0.5f;
                        }
                    }
                }
            }
            if (multi) {
                // wave reduction of the 16 + 1 partial costs: halve the values per lane at every exchange (15 + 2
                // shuffles for the 16, 6 for f(p1)); lane l < 16 ends up with the wave total of value brev4(l)
                const int lane = tid & 63, wave = tid >> 6;
                float r[kMaxCand];
#pragma unroll
                for (int i = 0; i < kMaxCand; ++i) r[i] = acc2[i].x + acc2[i].y;
                float x16 = acc2[kMaxCand].x + acc2[kMaxCand].y;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const bool up = lane & 1; const float keep = up ? r[i + 8] : r[i], send = up ? r[i] : r[i + 8]; r[i] = keep + __shfl_xor(send, 1, 64); }
#pragma unroll
                for (int i = 0; i < 4; ++i) { const bool up = lane & 2; const float keep = up ? r[i + 4] : r[i], send = up ? r[i] : r[i + 4]; r[i] = keep + __shfl_xor(send, 2, 64); }
#pragma unroll
                for (int i = 0; i < 2; ++i) { const bool up = lane & 4; const float keep = up ? r[i + 2] : r[i], send = up ? r[i] : r[i + 2]; r[i] = keep + __shfl_xor(send, 4, 64); }
                { const bool up = lane & 8; const float keep = up ? r[1] : r[0], send = up ? r[0] : r[1]; r[0] = keep + __shfl_xor(send, 8, 64); }
                float x = r[0];
                x += __shfl_xor(x, 16, 64);
                x += __shfl_xor(x, 32, 64);
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) x16 += __shfl_xor(x16, off, 64);
                double* ga = gacc + (wave * kLsGroupsRows + gi) * (kMaxCand + 1);
                if (lane < 16) ga[((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3)] += (double)x;
                if (lane == 0) ga[kMaxCand] += (double)x16;
            }
            }   // gi
        }
    }
    if (EP == EP_LINESEARCH && multi) {   // fold the four waves' accumulators into the state (f(p1) of a group behind its 16 costs)
        __syncthreads();
        for (int o = tid; o < ngroups * (kMaxCand + 1); o += 256) {
            const int gi = o / (kMaxCand + 1), i = o % (kMaxCand + 1);
            double x = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) x += gacc[(w * kLsGroupsRows + gi) * (kMaxCand + 1) + i];
            vals[o] = x;
        }
    }
    if (EP == EP_CROSS) return;
    if (EP == EP_STATS_M && !a.sums) return;
    if (!(EP == EP_LINESEARCH && multi)) {
    if (LS) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = acc2[i].x + acc2[i].y;
    }
    // ---- block reduction (float partials -> double) in a fixed order: shuffle tree per wave, waves in index order
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
        double x = (double)acc[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0) red[wave * NACC + i] = x;
    }
    __syncthreads();
    if (tid < NACC) {
        const double x = red[tid] + red[NACC + tid] + red[2 * NACC + tid] + red[3 * NACC + tid];
        if (!LS || tid < ncand || tid == kMaxCand)
            vals[(LS ? grp * (kMaxCand + 1) : 0) + (LS && tid == kMaxCand ? ncand : tid)] = x;
    }
    __syncthreads();
    }
    }   // grp
    // ---- totals over the workgroups in a fixed order; the last workgroup to arrive publishes them ----------
    int nv = LS ? (ngroups - 1) * (kMaxCand + 1) + ncand + 1 : NACC, nmax = 0;   // line search: costs[.. ncand] of the last group, no further
    if (EP == EP_PROJECT && a.maxword) {   // max |out| rides along as one more value (folded with max)
        float m = vmax2;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
        if ((tid & 63) == 0) fscr[tid >> 6] = (double)(sqrtf(m) * 1.0000002f);
        __syncthreads();
        if (tid == 0) vals[1] = fmax(fmax(fscr[0], fscr[1]), fmax(fscr[2], fscr[3]));
        nv = 2; nmax = 1;
    }
    if (!fold_across_workgroups(a.fold, vals, nv, nmax, fscr)) return;
    const int nsum = nv - nmax;
    if (tid < nsum) sums[tid] = a.overwrite ? vals[tid] : sums[tid] + vals[tid];
    if (EP == EP_PROJECT && a.maxword && tid == 0)
        *reinterpret_cast<unsigned long long*>(a.maxword) = (unsigned long long)__float_as_uint((float)vals[1]);
    if (LS && a.st && a.decide) {
        __syncthreads();
        if (tid == 0) ls_decide_dev(a.st, a.decide_which, a.decide_gamma_word, a.decide_next);
    }
}
