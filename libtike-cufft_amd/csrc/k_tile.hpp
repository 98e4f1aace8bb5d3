// k_tile.hpp -- one-launch operators for detector sizes whose tile fits on ONE compute unit (ndet = 16 ... 128)
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
#pragma once

// ---------------------------------------------------------------------------
// A 128^2 complex64 tile is 128 KiB: it fits the 160 KiB of LDS of a CU, so both passes of the 2-D DFT run in one
// workgroup and the column <-> row intermediate never leaves the chip (for ndet = 256 it has to).
// HBM traffic of the forward operator = its algorithmic bytes (8 ndet^2 per position written once); the probe adjoint
// reads each tile once.
//
// Thread (c, j0) of a tile, c = t % NL (lanes run along c), j0 = t / NL < T = N / 16, owns the lines l = c + h NL,
// h < CPT (one line per thread: 1024 threads with 128 registers each at ndet = 128), in two roles; the inter-step
// twiddles depend on j0 only and are read from an LDS copy of the table as they are used:
//   column role: column x = l, points y = j0 + b T + t N/R of the DFT over y   (LDS image [y][x]: lanes contiguous)
//   row role   : row   y = l, points x = j0 + b T + t N/R of the DFT over x   (lanes stride LS = N + 1 elements:
//                an odd stride, so the 32 lanes of a half-wave hit 32 different bank pairs)
//   fwd   (ptychofft.cu:60-73, kernels.cu:95-107): exit wave of 16 consecutive rows per thread (tile_exit_block: every
//          object element requested once, L2) x c*probe -> tile -> DFT over y -> tile -> DFT over x -> tile -> 16 bytes
//          per lane, whole rows, nontemporal, to g
//   adj_probe (ptychofft.cu:76-88 flg 1, kernels.cu:82-94): g -> tile (16 bytes per lane) -> IDFT over x -> IDFT
//          over y -> acc += near * conj(bilerp(psi)) in registers over all positions of the workgroup -> one float
//          atomic pair per probe pixel and workgroup at the end
// Tiles of 16^2 and 32^2 are packed 16 / 4 to a workgroup (256 threads).  Workgroups are persistent: position
// p = blockIdx.x * TPW + w, stride gridDim.x * TPW.
// ---------------------------------------------------------------------------
template <int N>
struct TileCfg {
    static constexpr int T = Plan<N>::T;
    static constexpr int CPT = 1;                             // lines per thread (2 at ndet = 128, i.e. 512 threads with 256 registers
                                                              // each: forward 0.341 ms against 0.274 ms with 1024 threads, round 3)
    static constexpr int NL = N / CPT;                        // lanes along a line index
    static constexpr int TT = NL * T;                         // threads per tile
    static constexpr int TPW = TT >= 256 ? 1 : 256 / TT;      // tiles per workgroup
    static constexpr int NT = TT * TPW;
    static constexpr int LS = N + 1;                          // LDS row stride (elements)
    static constexpr size_t lds_bytes = (size_t)TPW * N * LS * sizeof(c32) + N * sizeof(c32);
    // every wave lies inside ONE j0 (its 64 lanes are consecutive columns of the same row block): ndet a multiple of 64.
    // Then the row arithmetic of the gathers is wave-uniform (scalar unit); otherwise (16, 32 and 48, 80, 96, 112) per lane
    static constexpr bool WAVE_J0 = NL % 64 == 0;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TileArgs {
    const c32* obj;      // object f [ptheta][nz][n]
    const c32* prb;      // fwd: probe [ptheta][nprb][nprb]
    c32* g;              // fwd: farplane out; adj_probe: farplane in
    c32* out;            // adj_probe: probe gradient [ptheta][nprb][nprb] (accumulated into)
    const float* scan;
    const c32* table;
    Geom ge;
    int npos;            // ptheta * nscan
};

// One in-LDS transform of the thread's CPT lines: along the columns (role column) or the rows (role row) of the tile.
// FROM_LDS: step-0 inputs are read from the tile (else they are in v already).  Leaves the outputs of the last step in v
// (slot order); the caller stores them after a barrier of its own when the tile is their destination.
template <int N, int DIR, bool ROWROLE, bool FROM_LDS>
__device__ __forceinline__ void tile_dft(c32 (*v)[Plan<N>::E], c32* tile, const c32* wtab, int c, int j0) {
    using P = Plan<N>;
    using F = Fft<P, DIR>;
    constexpr int LS = TileCfg<N>::LS, CPT = TileCfg<N>::CPT, NL = TileCfg<N>::NL, LAST = P::NSTEP - 1;
    const F fft{};
    auto at = [&](int h, int i) { const int l = c + h * NL; return ROWROLE ? (l * LS + i) : (i * LS + l); };
#pragma unroll
    for (int h = 0; h < CPT; ++h) {
        if (FROM_LDS) fft.template load<0>(v[h], j0, [&](int i) { return tile[at(h, i)]; });
        fft.template compute<0>(v[h]);
    }
    if constexpr (P::NSTEP > 1) {
        if (FROM_LDS) __syncthreads();   // every thread of a line has read its step-0 inputs
#pragma unroll
        for (int h = 0; h < CPT; ++h) fft.template store<0>(v[h], j0, [&](int i, c32 val) { tile[at(h, i)] = val; });
        __syncthreads();
#pragma unroll
        for (int h = 0; h < CPT; ++h) fft.template load<1>(v[h], j0, [&](int i) { return tile[at(h, i)]; });
        if constexpr (P::NSTEP > 2) {
#pragma unroll
            for (int h = 0; h < CPT; ++h) fft.template compute_tab<1>(v[h], j0, wtab);
            __syncthreads();
#pragma unroll
            for (int h = 0; h < CPT; ++h) fft.template store<1>(v[h], j0, [&](int i, c32 val) { tile[at(h, i)] = val; });
            __syncthreads();
#pragma unroll
            for (int h = 0; h < CPT; ++h) fft.template load<2>(v[h], j0, [&](int i) { return tile[at(h, i)]; });
        }
#pragma unroll
        for (int h = 0; h < CPT; ++h) fft.template compute_tab<LAST>(v[h], j0, wtab);   // twiddles from the LDS table as they are used
    }
}
// outputs of the last step -> tile (natural order along the line)
template <int N, int DIR, bool ROWROLE>
__device__ __forceinline__ void tile_put(const c32 (*v)[Plan<N>::E], c32* tile, int c, int j0) {
    using P = Plan<N>;
    constexpr int LS = TileCfg<N>::LS, CPT = TileCfg<N>::CPT, NL = TileCfg<N>::NL, LAST = P::NSTEP - 1;
    const Fft<P, DIR> fft{};
#pragma unroll
    for (int h = 0; h < CPT; ++h) {
        const int l = c + h * NL;
        fft.template store<LAST>(v[h], j0, [&](int i, c32 val) { tile[ROWROLE ? (l * LS + i) : (i * LS + l)] = val; });
    }
}

// Bilinear patch values of a thread's E points of column ix (kernels.cu:97-104), natural order m <-> probe row
// j0 + m T - pad.  The patch origin is the same for every lane of a tile (a whole wave for ndet >= 32), so the row
// pointers stay in scalar registers and a lane needs one 32-bit offset; BATCH rows are requested at a time (the scheduling
// barrier keeps the compiler from requesting all 4 E taps at once).  fn(m, ok, value) is called for every m in order.
template <class P, int BATCH, class Fn>
__device__ __forceinline__ void tile_patch(const c32* __restrict__ ft, const Pos& q, const Geom& ge, int j0, int ix, bool col_ok, Fn fn) {
    constexpr int E = P::E, T = P::T;
    // Branch free: every tap is read from a clamped (always valid) element of the object and enters with weight zero
    // where the reference reads nothing (probe padding, kernels.cu:19-20) or this library reads zero (outside the object:
    // bilerp in ptycho_common.hpp).  32-bit element offsets from the object's base (the host checks nz n < 2^28).
    const int X = q.sx + ix;
    const int X0 = X < 0 ? 0 : (X >= ge.n ? ge.n - 1 : X), X1 = X + 1 < 0 ? 0 : (X + 1 >= ge.n ? ge.n - 1 : X + 1);
    const bool x0 = col_ok && X >= 0 && X < ge.n, x1 = col_ok && X + 1 >= 0 && X + 1 < ge.n;
    const float ax = x0 ? 1.0f - q.fx : 0.0f, bx = x1 ? q.fx : 0.0f;   // weights of columns X, X + 1
    const int Ya = q.sy + j0 - ge.pad;
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int iy = j0 + m * T - ge.pad;
        const bool ok = col_ok && iy >= 0 && iy < ge.nprb;
        const int Y = Ya + m * T;
        const int Y0 = Y < 0 ? 0 : (Y >= ge.nz ? ge.nz - 1 : Y), Y1 = Y + 1 < 0 ? 0 : (Y + 1 >= ge.nz ? ge.nz - 1 : Y + 1);
        const bool y0 = iy >= 0 && iy < ge.nprb && Y >= 0 && Y < ge.nz, y1 = iy >= 0 && iy < ge.nprb && Y + 1 >= 0 && Y + 1 < ge.nz;
        const float ay = y0 ? 1.0f - q.fy : 0.0f, by = y1 ? q.fy : 0.0f;
        const c32* r0 = ft + (unsigned)(Y0 * ge.n);
        const c32* r1 = ft + (unsigned)(Y1 * ge.n);
        // kernels.cu:97-104: f00 (1-fx)(1-fy) + f01 fx (1-fy) + f10 (1-fx) fy + f11 fx fy, each weight a product of two factors
        fn(m, ok, r0[X0] * (ax * ay) + r0[X1] * (bx * ay) + r1[X0] * (ax * by) + r1[X1] * (bx * by));
        if (m % BATCH == BATCH - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// The same for tiles whose lanes of a wave share j0 (ndet >= 64: a wave is 64 consecutive columns of one j0; `j0u` comes
// from readfirstlane): the row index, its clamp, the row weights and the two row pointers of every m are wave-uniform
// and live in scalar registers / run on the scalar unit; a lane contributes its two clamped column offsets and column
// weights, computed once.  Per m: four loads with a scalar base, five packed multiply-adds.
template <class P, int BATCH, class Fn>
__device__ __forceinline__ void tile_patch_rows(const c32* __restrict__ ft, const Pos& q, const Geom& ge, int j0u, int ix, bool col_ok, Fn fn) {
    constexpr int E = P::E, T = P::T;
    const int X = q.sx + ix;
    const unsigned X0 = (unsigned)(X < 0 ? 0 : (X >= ge.n ? ge.n - 1 : X)), X1 = (unsigned)(X + 1 < 0 ? 0 : (X + 1 >= ge.n ? ge.n - 1 : X + 1));
    const float ax = (col_ok && X >= 0 && X < ge.n) ? 1.0f - q.fx : 0.0f, bx = (col_ok && X + 1 >= 0 && X + 1 < ge.n) ? q.fx : 0.0f;
    const float wy0 = 1.0f - q.fy, wy1 = q.fy;
#pragma unroll
    for (int m = 0; m < E; ++m) {
        const int iy = j0u + m * T - ge.pad;
        const bool rowok = iy >= 0 && iy < ge.nprb;
        const int Y = q.sy + iy;
        const int Y0 = Y < 0 ? 0 : (Y >= ge.nz ? ge.nz - 1 : Y), Y1 = Y + 1 < 0 ? 0 : (Y + 1 >= ge.nz ? ge.nz - 1 : Y + 1);
        const float ay = (rowok && Y >= 0 && Y < ge.nz) ? wy0 : 0.0f, by = (rowok && Y + 1 >= 0 && Y + 1 < ge.nz) ? wy1 : 0.0f;
        const c32* r0 = ft + (unsigned)(Y0 * ge.n);
        const c32* r1 = ft + (unsigned)(Y1 * ge.n);
        fn(m, rowok && col_ok, (r0[X0] * ax + r0[X1] * bx) * ay + (r1[X0] * ax + r1[X1] * bx) * by);
        if (m % BATCH == BATCH - 1) __builtin_amdgcn_sched_barrier(0);
    }
}
template <int N, int BATCH, class Fn>
__device__ __forceinline__ void tile_exit_taps(const c32* __restrict__ ft, const Pos& q, const Geom& ge, int j0, int ix, bool col_ok, Fn fn) {
    if constexpr (TileCfg<N>::WAVE_J0) tile_patch_rows<Plan<N>, BATCH>(ft, q, ge, uni_i(j0), ix, col_ok, fn);
    else tile_patch<Plan<N>, BATCH>(ft, q, ge, j0, ix, col_ok, fn);
}

// Exit wave of the 16 CONSECUTIVE rows 16 jb ... 16 jb + 15 of detector column c (kernels.cu:95-107): every object element
// is requested ONCE per position -- 17 rows of 16 bytes (elements X, X + 1) per thread; the row-pair sums h = f[X] (1-fx) +
// f[X+1] fx are shared by the two probe rows that tap them -- instead of four 8-byte taps per probe pixel (64 requests per
// thread: an ablation put that gather at half of the forward tile kernel's time, 0.14 of 0.28 ms at 4096 x 128^2).
// Elements outside the object enter as zero, probe padding gives zero (as tile_patch).  ex[k] <-> row 16 jb + k.
typedef float f32x4_a8 __attribute__((ext_vector_type(4), aligned(8)));
template <int N>
__device__ __forceinline__ void tile_exit_block(const c32* __restrict__ ft, const c32* __restrict__ prb, const Pos& q, const Geom& ge,
                                                int jb, int ix, bool col_ok, float cinv, c32* ex) {
    constexpr int E = Plan<N>::E;
    const c32 zero = c32{0.0f, 0.0f};
    const int X = q.sx + ix;
    const int Xb = X < 0 ? 0 : (X > ge.n - 2 ? ge.n - 2 : X);   // the 16 bytes [Xb, Xb + 1] lie inside the row
    const int sel = X - Xb;                                      // 0 except where the patch leaves the object sideways
    const float fx0 = (col_ok && X >= 0 && X < ge.n) ? 1.0f - q.fx : 0.0f;       // weight of element X
    const float fx1 = (col_ok && X + 1 >= 0 && X + 1 < ge.n) ? q.fx : 0.0f;      // weight of element X + 1
    const float wa = sel == 0 ? fx0 : (sel == -1 ? fx1 : 0.0f);   // weight of the first loaded element  (it is X, or X + 1 when X = -1)
    const float wb = sel == 0 ? fx1 : (sel == 1 ? fx0 : 0.0f);    // weight of the second loaded element (it is X + 1, or X when X = n - 1)
    const float wy0 = 1.0f - q.fy, wy1 = q.fy;
    const int r0 = E * jb - ge.pad;   // probe row of k = 0
    c32 hprev = zero;
#pragma unroll
    for (int k = 0; k <= E; ++k) {
        const int Y = q.sy + r0 + k;
        const int Yc = Y < 0 ? 0 : (Y >= ge.nz ? ge.nz - 1 : Y);
        const f32x4_a8 e = *reinterpret_cast<const f32x4_a8*>(ft + (unsigned)(Yc * ge.n + Xb));
        // (no selects on loaded values and no conditional loads: a select turns into a branch around the load, and the
        // branches pinned every request behind the previous one's wait -- 34 L2 round trips per position)
        const bool yin = Y >= 0 && Y < ge.nz;
        const c32 h = c32{e.x, e.y} * (yin ? wa : 0.0f) + c32{e.z, e.w} * (yin ? wb : 0.0f);
        if (k > 0) {
            const int iy = r0 + k - 1;
            const bool ok = col_ok && iy >= 0 && iy < ge.nprb;
            const c32 pv = prb[ok ? iy * ge.nprb + ix : 0];
            ex[k - 1] = cmul(pv * (ok ? cinv : 0.0f), hprev * wy0 + h * wy1);
        }
        hprev = h;
        // rows requested together: two batches at ndet >= 64 (0.233 -> 0.217 ms at 128, 0.052 -> 0.046 at 64), batches of four
        // below (per-lane row arithmetic: larger batches take 200+ registers there)
        constexpr int RB = N >= 64 ? 9 : 4;
        if (k % RB == RB - 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// position of a tile slot; for whole-wave tiles the decoded position is moved to scalar registers
template <int N>
__device__ __forceinline__ Pos tile_pos(const float* __restrict__ scan, int p, const Geom& ge) {
    Pos q = decode_pos(scan, p, ge);
    if (TileCfg<N>::TPW == 1) {   // one tile per workgroup: every lane decodes the same position
        q.sy = uni_i(q.sy); q.sx = uni_i(q.sx);
        q.fy = uni_f(q.fy); q.fx = uni_f(q.fx);
        q.valid = uni_i(q.valid) != 0; q.inside = uni_i(q.inside) != 0;
    }
    return q;
}

template <int N>
__global__ __launch_bounds__(TileCfg<N>::NT) void k_fwd_tile(const TileArgs a) {
    using CF = TileCfg<N>;
    using P = Plan<N>;
    constexpr int E = P::E, TT = CF::TT, TPW = CF::TPW, LS = CF::LS, CPT = CF::CPT, NL = CF::NL;
    __shared__ c32 lds[TPW * N * LS];
    __shared__ c32 wtab[N];
    const int tid = threadIdx.x, w = tid / TT, t = tid % TT;
    const int c = t % NL, j0 = t / NL;
    c32* tile = lds + w * N * LS;
    const Geom ge = a.ge;
    const float cinv = 1.0f / (float)N;   // kernels.cu:65
    const c32 zero = c32{0.0f, 0.0f};
    for (int i = tid; i < N; i += CF::NT) wtab[i] = a.table[i];
    __syncthreads();

    for (int base = blockIdx.x * TPW; base < a.npos; base += gridDim.x * TPW) {
        const bool live = base + w < a.npos;
        const int p = live ? base + w : a.npos - 1;   // an idle tile slot of the last trip repeats a position and stores nothing
        const int th = p / ge.nscan;
        const Pos q = tile_pos<N>(a.scan, p, ge);
        const c32* ft = a.obj + (size_t)th * ge.nz * ge.n;
        c32 v[CPT][E];
        // (plans with more than 16 points per thread -- 48 ... 112 --: an opaque copy of j0 keeps the E probe-row addresses and the
        // 2 x E twiddle addresses of each transform inside the position loop; hoisted out of it they cost k_fwd_tile<112> 115
        // spilled registers)
        int jv = j0;
        if constexpr (E > 16) asm volatile("" : "+v"(jv));
        // ---- exit waves of the thread's columns (kernels.cu:95-107): blocks of 16 consecutive rows -> tile -> DFT over y ----
        // (requesting the NEXT position's rows before this tile streams out and finishing them afterwards was measured
        // slower, twice: 0.322 against 0.217 ms at ndet = 128, 0.070 against 0.046 ms at 64 -- 68 registers of raw rows or 32 of
        // finished values carried across the copy-out cost spills / a wave per SIMD; profiles/r03/knob_sweep.txt)
#pragma unroll
        for (int h = 0; h < CPT; ++h) {
            const int l = c + h * NL, ix = l - ge.pad;
            const bool col_ok = ix >= 0 && ix < ge.nprb;
            c32 ex[E];
            if (q.valid) {
                tile_exit_block<N>(ft, a.prb + (size_t)th * ge.nprb * ge.nprb, q, ge, CF::WAVE_J0 ? uni_i(j0) : jv, ix, col_ok, cinv, ex);
            } else {
#pragma unroll
                for (int k = 0; k < E; ++k) ex[k] = zero;
            }
#pragma unroll
            for (int k = 0; k < E; ++k) tile[(E * jv + k) * LS + l] = ex[k];
        }
        __syncthreads();
        if constexpr (E > 16) asm volatile("" : "+v"(jv));
        tile_dft<N, -1, false, true>(v, tile, wtab, c, jv);
        if (P::NSTEP > 1) __syncthreads();   // the exchange slots have been read: the tile may take the column results
        tile_put<N, -1, false>(v, tile, c, jv);   // [ky][x]
        __syncthreads();
        // ---- DFT over x of the thread's rows ---------------------------------------------------------------
        if constexpr (E > 16) asm volatile("" : "+v"(jv));
        tile_dft<N, -1, true, true>(v, tile, wtab, c, jv);
        if (P::NSTEP > 1) __syncthreads();
        tile_put<N, -1, true>(v, tile, c, jv);    // [ky][kx]
        __syncthreads();
        // ---- whole rows to g, 16 bytes per lane -----------------------------------------------------------
        if (live) {
            f32x4* gt = reinterpret_cast<f32x4*>(a.g + (size_t)p * N * N);
#pragma unroll 4
            for (int u = t; u < N * N / 2; u += TT) {
                const int row = u / (N / 2), cu = (u % (N / 2)) * 2;
                const c32 lo = tile[row * LS + cu], hi = tile[row * LS + cu + 1];
                __builtin_nontemporal_store(f32x4{lo.x, lo.y, hi.x, hi.y}, gt + u);
            }
        }
        __syncthreads();   // the tile is rewritten by the next position
    }
}

template <int N>
__global__ __launch_bounds__(TileCfg<N>::NT) void k_adjprb_tile(const TileArgs a) {
    using CF = TileCfg<N>;
    using P = Plan<N>;
    using F = Fft<P, +1>;
    constexpr int E = P::E, T = P::T, TT = CF::TT, TPW = CF::TPW, LS = CF::LS, CPT = CF::CPT, NL = CF::NL;
    __shared__ c32 lds[TPW * N * LS];
    __shared__ c32 wtab[N];
    const int tid = threadIdx.x, w = tid / TT, t = tid % TT;
    const int c = t % NL, j0 = t / NL;
    c32* tile = lds + w * N * LS;
    const Geom ge = a.ge;
    const float cinv = 1.0f / (float)N;
    const c32 zero = c32{0.0f, 0.0f};
    for (int i = tid; i < N; i += CF::NT) wtab[i] = a.table[i];
    __syncthreads();

    c32 acc[CPT][E];   // probe-gradient sums of pixels (row j0 + m T - pad, column c + h NL - pad), natural order m
#pragma unroll
    for (int h = 0; h < CPT; ++h)
#pragma unroll
        for (int m = 0; m < E; ++m) acc[h][m] = zero;
    int cur_th = -1;
    auto flush = [&](int th) {   // kernels.cu:92-93: one atomic pair per pixel and workgroup
        int jf = j0;
        asm volatile("" : "+v"(jf));   // opaque: keeps the pixel addresses of this rare path out of the loop's live registers
#pragma unroll
        for (int h = 0; h < CPT; ++h) {
            const int ix = c + h * NL - ge.pad;
            const bool col_ok = ix >= 0 && ix < ge.nprb;
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const int iy = jf + m * T - ge.pad;
                if (col_ok && iy >= 0 && iy < ge.nprb && (acc[h][m].x != 0.0f || acc[h][m].y != 0.0f)) {
                    float* o = reinterpret_cast<float*>(a.out + ((size_t)th * ge.nprb + iy) * ge.nprb + ix);
                    atomicAdd(o, acc[h][m].x * cinv);
                    atomicAdd(o + 1, acc[h][m].y * cinv);
                }
                acc[h][m] = zero;
            }
        }
    };

    for (int base = blockIdx.x * TPW; base < a.npos; base += gridDim.x * TPW) {
        const bool live = base + w < a.npos;
        const int p = live ? base + w : a.npos - 1;
        const int th = p / ge.nscan;
        if (th != cur_th) {
            if (cur_th >= 0) flush(cur_th);
            cur_th = th;
        }
        const Pos q = tile_pos<N>(a.scan, p, ge);
        const c32* ft = a.obj + (size_t)th * ge.nz * ge.n;
        // ---- g -> tile, whole rows, 16 bytes per lane ------------------------------------------------------
        {
            const f32x4* gt = reinterpret_cast<const f32x4*>(a.g + (size_t)p * N * N);
#pragma unroll 4
            for (int u = t; u < N * N / 2; u += TT) {
                const int row = u / (N / 2), cu = (u % (N / 2)) * 2;
                const f32x4 val = __builtin_nontemporal_load(gt + u);
                tile[row * LS + cu] = c32{val.x, val.y};
                tile[row * LS + cu + 1] = c32{val.z, val.w};
            }
        }
        __syncthreads();
        c32 v[CPT][E];
        // ---- IDFT over x of the thread's rows, then IDFT over y of its columns (ptychofft.cu:85: unnormalised) ----
        tile_dft<N, +1, true, true>(v, tile, wtab, c, j0);
        if (P::NSTEP > 1) __syncthreads();
        tile_put<N, +1, true>(v, tile, c, j0);   // [ky][x]
        __syncthreads();
        tile_dft<N, +1, false, true>(v, tile, wtab, c, j0);
        // ---- acc += near * conj(bilerp(psi))  (kernels.cu:84-93) -------------------------------------------
        if (live && q.valid) {
#pragma unroll
            for (int h = 0; h < CPT; ++h) {
                const int ix = c + h * NL - ge.pad;
                const bool col_ok = ix >= 0 && ix < ge.nprb;
                c32 nat[E];
                F::to_natural(v[h], nat);
                // (unconditional: the patch value is an exact zero wherever the probe has no pixel -- its weights are zero
                // there --, and an `if (ok)` becomes a branch that pins each row's four requests behind the previous row's wait)
                // (packed tiles, ndet <= 32, keep the branch: without it the per-lane row arithmetic of all 16 rows is hoisted and
                // the kernel takes 255 registers, 0.136 against 0.093 ms at 16384 x 32^2)
                tile_exit_taps<N, 4>(ft, q, ge, j0, ix, col_ok, [&](int m, bool ok, c32 val) {
                    if (CF::WAVE_J0 || ok) acc[h][m] += cmulc(nat[m], val);
                });
            }
        }
        __syncthreads();   // the tile is rewritten by the next position
    }
    if (TPW > 1 && ge.ptheta == 1) {
        // packed tiles (ndet 16 / 32), one angle: the TPW slots of the workgroup add up in LDS first -- every flush lands on
        // the same ndet^2 addresses, and 16 slots x 1024 workgroups of same-address atomics took 1.9 ms at ndet = 16
        static_assert(CPT == 1, "one line per thread");
#pragma unroll
        for (int m = 0; m < E; ++m) lds[(w * E + m) * TT + t] = acc[0][m];
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int m = 0; m < E; ++m) {
                c32 sum = zero;
#pragma unroll 4
                for (int ww = 0; ww < TPW; ++ww) sum += lds[(ww * E + m) * TT + t];
                __builtin_amdgcn_sched_barrier(0);   // (unrolled and hoisted, the 16 x 16 reads took 256 registers)
                acc[0][m] = sum;
            }
            flush(0);
        }
        return;
    }
    if (cur_th >= 0) flush(cur_th);
}

// Row pass of the object adjoint for small tiles (ndet <= 64): dst tile j <- DFT over x of src tile tile_index[j], whole
// rows in and out at 16 bytes per lane through the LDS tile.  (k_rows reads a row as T = ndet / 16 lanes x 16 strided
// points: at ndet = 32 a wave instruction touches 64 separate 8-byte pieces and the pass runs at 0.8 TB/s.)
// COLS: the DFT over y follows in the same launch (ptycho_fft2 of ndet <= 128: the whole 2-D transform of a tile)
template <int N, int DIR, bool COLS = false>
__global__ __launch_bounds__(TileCfg<N>::NT) void k_rows_tile(const c32* src, c32* dst,   // (may alias: ptycho_fft2 in place)
                                                              const int* __restrict__ tile_index, const int ntiles,
                                                              const c32* __restrict__ table) {
    using CF = TileCfg<N>;
    using P = Plan<N>;
    constexpr int E = P::E, TT = CF::TT, TPW = CF::TPW, LS = CF::LS, CPT = CF::CPT, NL = CF::NL;
    __shared__ c32 lds[TPW * N * LS];
    __shared__ c32 wtab[N];
    const int tid = threadIdx.x, w = tid / TT, t = tid % TT;
    const int c = t % NL, j0 = t / NL;
    c32* tile = lds + w * N * LS;
    for (int i = tid; i < N; i += CF::NT) wtab[i] = table[i];
    __syncthreads();
    for (int base = blockIdx.x * TPW; base < ntiles; base += gridDim.x * TPW) {
        const bool live = base + w < ntiles;
        const int j = live ? base + w : ntiles - 1;
        const f32x4* in = reinterpret_cast<const f32x4*>(src + (size_t)(tile_index ? tile_index[j] : j) * N * N);
#pragma unroll 4
        for (int u = t; u < N * N / 2; u += TT) {
            const int row = u / (N / 2), cu = (u % (N / 2)) * 2;
            const f32x4 val = __builtin_nontemporal_load(in + u);
            tile[row * LS + cu] = c32{val.x, val.y};
            tile[row * LS + cu + 1] = c32{val.z, val.w};
        }
        __syncthreads();
        c32 v[CPT][E];
        tile_dft<N, DIR, true, true>(v, tile, wtab, c, j0);
        if (P::NSTEP > 1) __syncthreads();
        tile_put<N, DIR, true>(v, tile, c, j0);
        __syncthreads();
        if constexpr (COLS) {
            tile_dft<N, DIR, false, true>(v, tile, wtab, c, j0);
            if (P::NSTEP > 1) __syncthreads();
            tile_put<N, DIR, false>(v, tile, c, j0);
            __syncthreads();
        }
        if (live) {
            f32x4* out = reinterpret_cast<f32x4*>(dst + (size_t)j * N * N);
#pragma unroll 4
            for (int u = t; u < N * N / 2; u += TT) {
                const int row = u / (N / 2), cu = (u % (N / 2)) * 2;
                const c32 lo = tile[row * LS + cu], hi = tile[row * LS + cu + 1];
                out[u] = f32x4{lo.x, lo.y, hi.x, hi.y};   // read again by the column pass: a plain store
            }
        }
        __syncthreads();
    }
}

// Row pass of the object adjoint for the plans with four threads per row and a tile near the LDS limit (80, 96, 112): NR = 16
// rows per workgroup trip = ONE wave per workgroup (barriers are free), 15 KiB of LDS, eight and more workgroups per CU -- the
// whole-tile k_rows_tile holds 101 KiB at 112, i.e. one workgroup of seven waves per CU whose load / transform / store phases
// do not overlap.  Same 16-byte-per-lane whole-row accesses.
template <int N, int DIR, int NR = 16>
__global__ __launch_bounds__(NR * Plan<N>::T) void k_rows_slab(const c32* __restrict__ src, c32* __restrict__ dst,
                                                              const int* __restrict__ tile_index, const int ntiles,
                                                              const c32* __restrict__ table) {
    using P = Plan<N>;
    constexpr int E = P::E, T = P::T, LS = TileCfg<N>::LS, NTH = NR * T, SPT = N / NR;
    static_assert(N % NR == 0 && TileCfg<N>::CPT == 1, "slabs tile the rows of a tile");
    __shared__ c32 slab[NR * LS];
    __shared__ c32 wtab[N];
    const int tid = threadIdx.x;
    const int c = tid % NR, j0 = tid / NR;
    for (int i = tid; i < N; i += NTH) wtab[i] = table[i];
    __syncthreads();
    const long long nitems = (long long)ntiles * SPT;
    for (long long item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int j = (int)(item / SPT), sb = (int)(item % SPT);
        const f32x4* in = reinterpret_cast<const f32x4*>(src + (size_t)(tile_index ? tile_index[j] : j) * N * N + (size_t)sb * NR * N);
#pragma unroll 4
        for (int u = tid; u < NR * N / 2; u += NTH) {
            const int row = u / (N / 2), cu = (u % (N / 2)) * 2;
            const f32x4 val = __builtin_nontemporal_load(in + u);
            slab[row * LS + cu] = c32{val.x, val.y};
            slab[row * LS + cu + 1] = c32{val.z, val.w};
        }
        __syncthreads();
        c32 v[1][E];
        int jv = j0;
        asm volatile("" : "+v"(jv));   // keeps the twiddle addresses inside the loop (k_fwd_tile)
        tile_dft<N, DIR, true, true>(v, slab, wtab, c, jv);
        if (P::NSTEP > 1) __syncthreads();
        tile_put<N, DIR, true>(v, slab, c, jv);
        __syncthreads();
        f32x4* out = reinterpret_cast<f32x4*>(dst + (size_t)j * N * N + (size_t)sb * NR * N);
#pragma unroll 4
        for (int u = tid; u < NR * N / 2; u += NTH) {
            const int row = u / (N / 2), cu = (u % (N / 2)) * 2;
            const c32 lo = slab[row * LS + cu], hi = slab[row * LS + cu + 1];
            out[u] = f32x4{lo.x, lo.y, hi.x, hi.y};   // read again by the column pass: a plain store
        }
        __syncthreads();
    }
}
