// ptycho_kernels.hip -- gfx950 kernels + C ABI for the ptychography operators.
//
// What it replaces in the reference (paths relative to /root/reference):
//   muloperator flg=2/0/1           src/cuda/kernels.cu:8-108
//   ptychofft ctor/fwd/adj/free     src/cuda/ptychofft.cu:5-88
//   cuFFT batched 2-D C2C           src/cuda/ptychofft.cu:14-20,72,85
//
// Structure: the 2-D DFT is split into a column pass and a row
// pass; the probe/object work is fused into the column pass, which owns a strip
// of C detector columns for a whole group of scan positions and keeps the probe
// strip (or the probe-gradient accumulators) in registers across positions.
//
//   fwd : k_cols<FWD>  gather+bilerp+probe -> DFT over y -> strip of g
//         k_rows       in-place DFT over x on g (zero columns are never read)
//   adj : k_rows       inverse DFT over x, g -> chunk scratch (g untouched)
//         k_cols<ADJ>  inverse DFT over y -> conj(probe) / conj(patch) -> f / prb
//
// Work is issued in chunks of scan positions so that the intermediate of a
// chunk is still resident in the 256 MiB Infinity Cache when the second pass
// reads it.
#include <hip/hip_runtime.h>

#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ptycho_hip.h"
#include "fft_core.hpp"

using namespace pty;

namespace {

struct Geom {
    int ptheta, nz, n, nscan, ndet, nprb, pad;
};

enum Mode { M_PLAIN = 0, M_FWD = 1, M_ADJ_OBJ = 2, M_ADJ_PRB = 3 };

template <int N>
struct ColCfg {
    static constexpr int T = Plan<N>::T;
    static constexpr int C0 = (256 / T) > 16 ? (256 / T) : 16;
    static constexpr int C = C0 > N ? N : C0;   // detector columns per strip
    static constexpr int NT = T * C;            // threads per workgroup
};

struct ColArgs {
    const c32* src;     // FWD: object f; ADJ_*: chunk scratch (tile index k - k_begin); PLAIN: tiles
    c32* dst;           // FWD: farplane g; ADJ_OBJ: object f; ADJ_PRB: probe; PLAIN: tiles
    const c32* aux;     // FWD / ADJ_OBJ: probe; ADJ_PRB: object f
    const float* scan;  // [ptheta][nscan][2]
    const c32* table;   // exp(-2 pi i k / N)
    Geom ge;
    const int* order;   // processing order: position = order[k] (nullptr: identity)
    int natural_tiles;  // ADJ_*: 1 = src tile of position p is tile p (CG work buffers); 0 = tile k - k_begin
    int nt;             // bit 2: nontemporal strip stores (FWD); bit 3: nontemporal tile loads (ADJ)
    int k_begin, k_end; // range of k handled by this launch; ADJ_* read scratch tile k - k_begin
    int ngroups;        // position groups; grid = nstrips * ngroups
    int strip0, nstrips;
};

struct RowArgs {
    const c32* src;
    c32* dst;
    const c32* table;
    long long nrows;
    const int* tile_index;   // source tile of local tile j is tile_index[j] (nullptr: j); dst is always local
    int xa, xb;   // columns outside [xa, xb) are read as zero
    int wa, wb;   // only columns in [wa, wb) are written
    int nt;       // 1: nontemporal loads / stores (streaming data with no reuse)
    int dst_indexed;   // 1: the destination tile is tile_index[j] too (in place on scattered tiles)
};

struct Pos {
    int sy, sx;
    float fy, fx;
    bool valid, inside;
};

// modff split of one scan position -- kernels.cu:27-28,39
__device__ __forceinline__ Pos decode_xy(const float py, const float px, const Geom& ge) {
    Pos q;
    float iy, ix;
    q.fy = modff(py, &iy);
    q.fx = modff(px, &ix);
    // the reference skips sx < 0 || sy < 0; non-finite positions are skipped too
    q.valid = !(ix < 0.0f || iy < 0.0f) && (ix < 1.0e9f) && (iy < 1.0e9f) && (ix == ix) && (iy == iy);
    q.sy = q.valid ? (int)iy : 0;
    q.sx = q.valid ? (int)ix : 0;
    q.inside = q.valid && (q.sy + ge.nprb + 1 <= ge.nz) && (q.sx + ge.nprb + 1 <= ge.n);
    return q;
}
__device__ __forceinline__ Pos decode_pos(const float* __restrict__ scan, int p, const Geom& ge) {
    return decode_xy(scan[2 * (size_t)p], scan[2 * (size_t)p + 1], ge);
}

// Values that are the same in every lane (read from LDS at a uniform index): moving them to
// scalar registers lets the address arithmetic that depends on them run on the scalar unit.
__device__ __forceinline__ int uni_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// positions of one run (<= kRunMax), staged in LDS once per workgroup so that the per-position
// loop has no dependent global loads (order[k] -> scan[p]) on its critical path
constexpr int kRunMax = 128;
struct RunMeta {
    int p[kRunMax];
    float py[kRunMax], px[kRunMax];
};
__device__ __forceinline__ void load_run(RunMeta& rm, const int* __restrict__ order, const float* __restrict__ scan,
                                         int kb, int ke, int tid) {
    const int n = ke - kb;
    for (int i = tid; i < n; i += (int)blockDim.x) {
        const int p = order ? order[kb + i] : kb + i;
        rm.p[i] = p;
        rm.py[i] = scan[2 * (size_t)p];
        rm.px[i] = scan[2 * (size_t)p + 1];
    }
}

// kernels.cu:97-104 -- same taps, same left-to-right weight products
__device__ __forceinline__ c32 bilerp(const c32* __restrict__ ft, int Y, int X, const Pos& q, const Geom& ge) {
    const float wx0 = 1.0f - q.fx, wy0 = 1.0f - q.fy;
    c32 f00, f01, f10, f11;
    if (q.inside) {
        const c32* p = ft + (size_t)Y * ge.n + X;
        f00 = p[0]; f01 = p[1]; f10 = p[ge.n]; f11 = p[ge.n + 1];
    } else {
        const bool y0 = Y >= 0 && Y < ge.nz, y1 = Y + 1 >= 0 && Y + 1 < ge.nz;
        const bool x0 = X >= 0 && X < ge.n, x1 = X + 1 >= 0 && X + 1 < ge.n;
        const c32 z = c32{0.0f, 0.0f};
        f00 = (y0 && x0) ? ft[(size_t)Y * ge.n + X] : z;
        f01 = (y0 && x1) ? ft[(size_t)Y * ge.n + X + 1] : z;
        f10 = (y1 && x0) ? ft[(size_t)(Y + 1) * ge.n + X] : z;
        f11 = (y1 && x1) ? ft[(size_t)(Y + 1) * ge.n + X + 1] : z;
    }
    return f00 * wx0 * wy0 + f01 * q.fx * wy0 + f10 * wx0 * q.fy + f11 * q.fx * q.fy;
}

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
                        const c32 val = v[b * RL + brev(tt, ilog2(RL))];
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
                    const c32 val = v[b * RL + brev(tt, ilog2(RL))];
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

// ---------------------------------------------------------------------------
// Row pass: DFT over x of contiguous rows, B = 256/T rows per workgroup step.
// ---------------------------------------------------------------------------
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
        if (a.nt & 2)
            fft.template store<LAST>(v, j0, [&](int i, c32 val) {
                if (ok && i >= a.wa && i < a.wb) __builtin_nontemporal_store(val, drow + i);
            });
        else
            fft.template store<LAST>(v, j0, [&](int i, c32 val) {
                if (ok && i >= a.wa && i < a.wb) drow[i] = val;
            });
        if (P::NSTEP > 1) __syncthreads();
    }
}

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

template <int N>
__global__ __launch_bounds__(ColCfg<N>::NT) void k_cols_adjwin(const ColArgs a, const int seglen) {
    using P = Plan<N>;
    using F = Fft<P, +1>;
    constexpr int E = P::E, T = P::T, C = ColCfg<N>::C, NT = ColCfg<N>::NT;
    constexpr int LAST = P::NSTEP - 1;
    constexpr int WC = WinCfg<N>::WC, H = WinCfg<N>::H;
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
            if ((v.x != 0.0f || v.y != 0.0f) && Y < ge.nz && X >= 0 && X < ge.n) {
                float* op = reinterpret_cast<float*>(fo + (size_t)Y * ge.n + X);
                atomicAdd(op, v.x);
                atomicAdd(op + 1, v.y);
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
        return a.src + (size_t)(a.natural_tiles ? st.p : (k - a.k_begin)) * N * N;
    };

    __syncthreads();
    St st = decode(kb);
    c32 v[E];
    if (st.have && st.q.valid) {
        const c32* tile_in = tile_of(st, kb);
        if (a.nt & 8)
            fft.template load<0>(v, j0, [&](int i) { return __builtin_nontemporal_load(tile_in + (size_t)i * N + x); });
        else
            fft.template load<0>(v, j0, [&](int i) { return tile_in[(size_t)i * N + x]; });
    }
    for (int k = kb; k < ke; ++k) {
        St nx = decode(k + 1);
        if (!st.q.valid) {   // skipped position: nothing to add; fetch the next tile
            if (nx.have && nx.q.valid) {
                const c32* tile_in = tile_of(nx, k + 1);
                fft.template load<0>(v, j0, [&](int i) { return tile_in[(size_t)i * N + x]; });
            }
            st = nx;
            continue;
        }
        const Pos q = st.q;
        if (st.t != cur_t) {
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
        // ---- T[y][c] = conj(c * prb) * near, written over the slots this thread just read ----
        {
            c32 nat[E];
            F::to_natural(v, nat);
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const c32 w = pr[m];
                lds[at(j0 + m * T)] = c32{w.x * nat[m].x + w.y * nat[m].y, w.x * nat[m].y - w.y * nat[m].x};
            }
        }
        // prefetch the next tile while the combine runs
        if (nx.have && nx.q.valid) {
            const c32* tile_in = tile_of(nx, k + 1);
            if (a.nt & 8)
                fft.template load<0>(v, j0, [&](int i) { return __builtin_nontemporal_load(tile_in + (size_t)i * N + x); });
            else
                fft.template load<0>(v, j0, [&](int i) { return tile_in[(size_t)i * N + x]; });
        }
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
        __syncthreads();   // T tile complete (and, after a re-anchor, the window is clean)
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
            for (int yy = y0; yy < y1; ++yy) {
                tp += CP;
                const c32 t00 = tp[1], t01 = tp[0];
                win[slot * WC + colw] += t00 * w00 + t01 * w01 + up0 * w10 + up1 * w11;
                up0 = t00; up1 = t01;
                slot = slot + 1 == H ? 0 : slot + 1;
            }
        }
        __syncthreads();   // combine done: the tile may be overwritten by the next position
        st = nx;
    }
    __syncthreads();
    flush(Ybase, Ytop);
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
template <int N, int MODE>
__global__ __launch_bounds__(ColCfg<N>::NT) void k_cols_gatherwin(const ColArgs a, const int seglen) {
    using P = Plan<N>;
    constexpr int DIR = (MODE == M_FWD) ? -1 : +1;
    using F = Fft<P, DIR>;
    constexpr int E = P::E, T = P::T, C = ColCfg<N>::C, NT = ColCfg<N>::NT;
    constexpr int LAST = P::NSTEP - 1;
    constexpr int WC = WinCfg<N>::WC, H = WinCfg<N>::H;
    constexpr int R0 = P::radix(0), RL = P::radix(LAST), NsL = P::ns(LAST);
    __shared__ c32 lds[N * C];
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

    F fft;
    fft.init(j0, a.table);
    for (int o = tid; o < H * WC; o += NT) win[o] = zero;

    // FWD: c * probe strip in step-0 slot order (zero on padding -> masks the gather);
    // ADJ_PRB: gradient accumulators in natural order m (row j0 + m*T)
    c32 pr[E];
    int cur_t = -1;
    int t_w = -1, X0 = 0, Ylo = 0, Yhi = 0;   // cached object rows [Ylo, Yhi), columns [X0, X0+WC)

    auto flush_probe = [&](int t) {
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const int iy = j0 + m * T - ge.pad;
            if (col_ok && iy >= 0 && iy < ge.nprb) {
                float* o = reinterpret_cast<float*>(a.dst + ((size_t)t * ge.nprb + iy) * ge.nprb + ix);
                const c32 sacc = pr[m] * cinv;
                atomicAdd(o, sacc.x);
                atomicAdd(o + 1, sacc.y);
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
                    const c32 val = ft[inb ? ((size_t)Y * ge.n + X) : 0];
                    pre_val = inb ? val : zero;
                    pre_slot = (Y % H) * WC + col;
                }
            } else {
                for (int o = tid; o < cnt; o += NT) {
                    const int Y = Yhi + o / WC, col = o % WC;
                    const int X = X0 + col;
                    const bool inb = Y < ge.nz && X >= 0 && X < ge.n;
                    const c32 val = ft[inb ? ((size_t)Y * ge.n + X) : 0];
                    win[(Y % H) * WC + col] = inb ? val : zero;
                }
            }
            Yhi = Rb;
        }
        return st;
    };
    auto prepare_commit = [&]() {
        if (pre_slot >= 0) win[pre_slot] = pre_val;
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
            fft.template load<0>(v, j0, [&](int i) { return tile_in[(size_t)i * N + x]; });
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
            fft.template compute<0>(v);
            if (P::NSTEP > 1) {
                fft.template store<0>(v, j0, [&](int i, c32 val) { lds[i * C + c] = val; });
                __syncthreads();   // also: the accumulation of k-1 is over, the window may move
            } else {
                __syncthreads();
            }
            const St cur = prepare_issue(k, ke);   // same decode as st, plus the window update
            if (P::NSTEP > 1) {
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
                const int s1 = slot + 1 == H ? 0 : slot + 1;
                const c32* r0 = win + slot * WC + colw;
                const c32* r1 = win + s1 * WC + colw;
                const c32 patch = r0[0] * w00 + r0[1] * w01 + r1[0] * w10 + r1[1] * w11;   // kernels.cu:84-91
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
    for (int k = kb; k < ke; ++k) {
        if (MODE == M_FWD && st.t != cur_t) {
            const c32* prb = a.aux + (size_t)st.t * ge.nprb * ge.nprb;
#pragma unroll
            for (int b = 0; b < E / R0; ++b)
#pragma unroll
                for (int tt = 0; tt < R0; ++tt) {
                    const int iy = j0 + b * T + tt * (N / R0) - ge.pad;
                    const bool ok = col_ok && iy >= 0 && iy < ge.nprb;
                    const c32 w = prb[ok ? ((size_t)iy * ge.nprb + ix) : 0];
                    pr[b * R0 + tt] = ok ? w * cinv : zero;
                }
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
                c32* tile_out = a.dst + (size_t)st.p * N * N;
#pragma unroll
                for (int m = 0; m < E; ++m) tile_out[(size_t)(j0 + m * T) * N + x] = zero;
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
            const int s1 = slot + 1 == H ? 0 : slot + 1;
            const c32* r0 = win + slot * WC + colw;
            const c32* r1 = win + s1 * WC + colw;
            return r0[0] * w00 + r0[1] * w01 + r1[0] * w10 + r1[1] * w11;   // kernels.cu:97-104
        };

        c32 v[E];
        if (MODE == M_FWD) {
            c32 nat[E];
            int slot = slot0;
#pragma unroll
            for (int m = 0; m < E; ++m) {
                nat[m] = patch_at(slot);
                slot += T;
                slot = slot >= H ? slot - H : slot;
            }
            F::from_natural(nat, v);
#pragma unroll
            for (int s2 = 0; s2 < E; ++s2) v[s2] = cmul(pr[s2], v[s2]);
        } else {
            const c32* tile_in = a.src + (size_t)(a.natural_tiles ? st.p : (k - a.k_begin)) * N * N;
            fft.template load<0>(v, j0, [&](int i) { return tile_in[(size_t)i * N + x]; });
        }
        fft.template compute<0>(v);
        St nx;
        if (P::NSTEP > 1) {
            fft.template store<0>(v, j0, [&](int i, c32 val) { lds[i * C + c] = val; });
            __syncthreads();
            if (MODE == M_FWD) nx = prepare_issue(k + 1, ke);   // window of k is no longer read
            fft.template load<1>(v, j0, [&](int i) { return lds[i * C + c]; });
            if (P::NSTEP > 2) {
                __syncthreads();
                fft.template compute<1>(v);
                fft.template store<1>(v, j0, [&](int i, c32 val) { lds[i * C + c] = val; });
                __syncthreads();
                fft.template load<2>(v, j0, [&](int i) { return lds[i * C + c]; });
            }
            fft.template compute<LAST>(v);
        } else if (MODE == M_FWD) {
            __syncthreads();
            nx = prepare_issue(k + 1, ke);
        }
        if (MODE == M_FWD) {
            c32* tile_out = a.dst + (size_t)st.p * N * N;
            if (a.nt & 4)
                fft.template store<LAST>(v, j0, [&](int i, c32 val) { __builtin_nontemporal_store(val, tile_out + (size_t)i * N + x); });
            else
                fft.template store<LAST>(v, j0, [&](int i, c32 val) { tile_out[(size_t)i * N + x] = val; });
            prepare_commit();
            __syncthreads();   // exchange buffer / new window rows visible to everyone
        } else {
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
}

// sort key of one position: angle | column bucket | row; skipped positions last
__global__ void k_sort_keys(const float* __restrict__ scan, const Geom ge, const int total,
                            unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const Pos q = decode_pos(scan, p, ge);
    unsigned long long key = ~0ull;
    if (q.valid) {
        const unsigned long long t = (unsigned long long)(p / ge.nscan);
        unsigned long long bx = (unsigned long long)(q.sx / kBucketPx), sy = (unsigned long long)q.sy;
        if (bx > 0x3fffffull) bx = 0x3fffffull;
        if (sy > 0x3fffffull) sy = 0x3fffffull;
        key = (t << 44) | (bx << 22) | sy;
    }
    keys[p] = key;
    vals[p] = p;
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
enum RowEp { EP_STATS = 1, EP_PROJECT = 2, EP_LINESEARCH = 3, EP_ACCUM_I = 4, EP_ACCUM_P = 5, EP_CROSS = 6 };
constexpr int kMaxCand = 16;

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
    float* acc1;         // EP_ACCUM_I: intensity;  EP_ACCUM_P: p1
    float* acc2;         // EP_ACCUM_P: p2
    float* acc3;         // EP_ACCUM_P: p3
    int first;           // 1: overwrite the arrays, 0: add to them
    c32* ip;             // EP_CROSS: image product u1 * conj(u2), [positions][ndet][ndet]
};

template <int N, int EP>
__global__ __launch_bounds__(256) void k_rows_fused(const RowFusedArgs a) {
    using P = Plan<N>;
    using F = Fft<P, -1>;
    using L = RowLds<N>;
    constexpr int E = P::E, T = P::T, B = 256 / T;
    constexpr int LAST = P::NSTEP - 1;
    constexpr int NACC = EP == EP_STATS ? 2 : (EP == EP_LINESEARCH ? kMaxCand + 1 : 1);
    __shared__ c32 lds[P::NSTEP > 1 ? B * L::FS : 1];
    __shared__ double red[4 * NACC];

    const int tid = threadIdx.x;
    const int f = tid / T, j0 = tid % T;
    F fft;
    fft.init(j0, a.table);
    const c32 zero = c32{0.0f, 0.0f};
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;

    // scale factors of ptycho.py:344-351 in float32, as the reference computes them
    float s = 1.0f, sinv = 1.0f;
    if (a.ab) {
        const float af = (float)a.ab[0], bf = (float)a.ab[1];
        s = af / bf;
        sinv = bf / af;
    }

    // forward DFT over x of one row held as step-0 inputs in v; result in natural order
    auto fwd_row = [&](c32* v, c32* nat) {
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
            __syncthreads();   // lds free for the next transform
        }
        F::to_natural(v, nat);
    };

    const long long nb = (a.nrows + B - 1) / B;
    for (long long batch = blockIdx.x; batch < nb; batch += gridDim.x) {
        const long long r = batch * B + f;
        const bool ok = r < a.nrows;
        const size_t rowoff = (size_t)r * N;
        c32 v[E], g1[E];
        fft.template load<0>(v, j0, [&](int i) { return (ok && i >= a.xa && i < a.xb) ? __builtin_nontemporal_load(a.s1 + rowoff + i) : zero; });
        fwd_row(v, g1);
        float d[E];
        auto load_data = [&]() {
#pragma unroll
            for (int m = 0; m < E; ++m) d[m] = ok ? __builtin_nontemporal_load(a.data + rowoff + j0 + m * T) : 0.0f;
        };
        if (EP == EP_STATS || EP == EP_PROJECT) load_data();
        if (EP == EP_CROSS) {
            // position correction (ptycho.py:398-403,198-204): u1 = G psi, u2 = G(psi + gamma dpsi)
            // = u1 + gamma G dpsi (ones probe); image product u1 conj(u2) is kept for the zoomed
            // DFT and its inverse row DFT goes back into the slot (column pass + arg-max follow).
            c32 g2[E], rr[E];
            fft.template load<0>(v, j0, [&](int i) { return (ok && i >= a.xa && i < a.xb) ? __builtin_nontemporal_load(a.s2 + rowoff + i) : zero; });
            fwd_row(v, g2);
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const c32 u2 = g1[m] + g2[m] * a.gamma0;
                rr[m] = cmulc(g1[m], u2);
                if (ok) __builtin_nontemporal_store(rr[m], a.ip + rowoff + j0 + m * T);
            }
            F::from_natural(rr, v);
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
                if (ok) __builtin_nontemporal_store(val, a.out + rowoff + i);
            });
            if (P::NSTEP > 1) __syncthreads();
            continue;
        }

        if (EP == EP_STATS) {
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const float I = g1[m].x * g1[m].x + g1[m].y * g1[m].y;
                acc[0] += sqrtf(I * d[m]);
                acc[1] += I;
            }
        } else if (EP == EP_ACCUM_I) {
            if (ok) {
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    const float I = g1[m].x * g1[m].x + g1[m].y * g1[m].y;
                    float* o = a.acc1 + rowoff + j0 + m * T;
                    *o = a.first ? I : *o + I;
                }
            }
        } else if (EP == EP_ACCUM_P) {
            c32 g2[E];
            fft.template load<0>(v, j0, [&](int i) { return (ok && i >= a.xa && i < a.xb) ? __builtin_nontemporal_load(a.s2 + rowoff + i) : zero; });
            fwd_row(v, g2);
            if (ok) {
#pragma unroll
                for (int m = 0; m < E; ++m) {
                    const size_t o = rowoff + j0 + m * T;
                    const c32 t1 = g1[m] * s;
                    const float p1 = t1.x * t1.x + t1.y * t1.y;
                    const float p2 = g2[m].x * g2[m].x + g2[m].y * g2[m].y;
                    const float p3 = 2.0f * (t1.x * g2[m].x + t1.y * g2[m].y);
                    a.acc1[o] = a.first ? p1 : a.acc1[o] + p1;
                    a.acc2[o] = a.first ? p2 : a.acc2[o] + p2;
                    a.acc3[o] = a.first ? p3 : a.acc3[o] + p3;
                }
            }
        } else if (EP == EP_PROJECT) {
            const float s2 = s * s;
            c32 rr[E];
#pragma unroll
            for (int m = 0; m < E; ++m) {
                // single mode: S comes from the unscaled probe -> I' = |g|^2 s^2, fpsi = (g s)(1/s');
                // multi mode: S comes from the rescaled probe and I is the summed intensity array
                const float I = a.inten ? (ok ? a.inten[rowoff + j0 + m * T] : 0.0f) * s2
                                        : (g1[m].x * g1[m].x + g1[m].y * g1[m].y) * s2;
                const c32 fp = a.inten ? g1[m] * sinv : (g1[m] * s) * sinv;
                const float sd = sqrtf(d[m]), sI = sqrtf(I);
                rr[m] = fp - (fp * sd) / (sI + 1e-32f);
                const float df = sI - sd;
                acc[0] += ok ? df * df : 0.0f;
            }
            // inverse DFT over x of the projected row, same twiddle registers (conjugated)
            F::from_natural(rr, v);
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
                if (ok) __builtin_nontemporal_store(val, a.out + rowoff + i);
            });
            if (P::NSTEP > 1) __syncthreads();
        } else {   // EP_LINESEARCH
            c32 g2[E];
            fft.template load<0>(v, j0, [&](int i) { return (ok && i >= a.xa && i < a.xb) ? __builtin_nontemporal_load(a.s2 + rowoff + i) : zero; });
            fwd_row(v, g2);
            load_data();   // after the second transform: keeps 16 registers free during it
#pragma unroll
            for (int m = 0; m < E; ++m) {
                const c32 t1 = g1[m] * s;
                const float p1 = t1.x * t1.x + t1.y * t1.y;
                const float p2 = g2[m].x * g2[m].x + g2[m].y * g2[m].y;
                const float p3 = 2.0f * (t1.x * g2[m].x + t1.y * g2[m].y);
                const float sd = sqrtf(d[m]);
                float df = sqrtf(fabsf(p1)) - sd;
                acc[kMaxCand] += df * df;
                float gam = a.gamma0;
#pragma unroll
                for (int j = 0; j < kMaxCand; ++j) {
                    if (j < a.ncand) {
                        const float xx = p1 + (gam * gam) * p2 + gam * p3;
                        df = sqrtf(fabsf(xx)) - sd;
                        acc[j] += df * df;
                    }
                    gam *= 0.5f;
                }
            }
        }
    }
    if (EP == EP_ACCUM_I || EP == EP_ACCUM_P || EP == EP_CROSS) return;
    // ---- block reduction (float partials -> double), one atomic per value per workgroup
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
        if (EP != EP_LINESEARCH || tid < a.ncand || tid == kMaxCand)
            atomicAdd(a.sums + (EP == EP_LINESEARCH && tid == kMaxCand ? a.ncand : tid), x);
    }
}

// Column pass of the coarse cross-correlation with a fused arg-max (ptycho.py:204-207):
// inverse DFT over y of the slot's tiles, |.|, and per position the first maximum as a packed
// 64-bit key (value bits << 32 | ~flat index) merged with atomicMax.
template <int N>
__global__ __launch_bounds__(ColCfg<N>::NT) void k_cols_argmax(const c32* __restrict__ tiles, const c32* __restrict__ table,
                                                               unsigned long long* __restrict__ best, const int npos,
                                                               const int ngroups) {
    using P = Plan<N>;
    using F = Fft<P, +1>;
    constexpr int E = P::E, T = P::T, C = ColCfg<N>::C, NT = ColCfg<N>::NT;
    constexpr int LAST = P::NSTEP - 1;
    constexpr int NW = (NT + 63) / 64;
    __shared__ c32 lds[N * C];
    __shared__ unsigned long long red[NW];
    const int tid = threadIdx.x;
    const int c = tid % C, j0 = tid / C;
    constexpr int nstrips = N / C;
    const int strip = blockIdx.x % nstrips, group = blockIdx.x / nstrips;
    const int x = strip * C + c;
    F fft;
    fft.init(j0, table);
    for (int p = group; p < npos; p += ngroups) {
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
            fft.template compute<LAST>(v);
        }
        c32 nat[E];
        F::to_natural(v, nat);
        unsigned long long key = 0ull;
#pragma unroll
        for (int m = 0; m < E; ++m) {
            const float mag = sqrtf(nat[m].x * nat[m].x + nat[m].y * nat[m].y);
            const unsigned idx = (unsigned)((j0 + m * T) * N + x);
            const unsigned long long k2 = ((unsigned long long)__float_as_uint(mag) << 32) | (unsigned long long)(0xffffffffu - idx);
            key = k2 > key ? k2 : key;
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

// elementwise reductions over stored arrays (multi-mode CG path): no DFT involved
//   MODE 0: sums += { sum sqrt(I d), sum I }                                   (ptycho.py:342-343)
//   MODE 1: costs[j] += sum (sqrt|p1 + y_j^2 p2 + y_j p3| - sqrt d)^2, costs[ncand] += f(p1)
constexpr int kArrCand = 32;   // candidates per pass of the array line search

template <int MODE>
__global__ __launch_bounds__(256) void k_array_reduce(const float* __restrict__ p1, const float* __restrict__ p2,
                                                      const float* __restrict__ p3, const float* __restrict__ d,
                                                      const long long n, const float gamma0, const int ncand,
                                                      double* __restrict__ sums) {
    constexpr int NACC = MODE == 0 ? 2 : kArrCand + 1;
    __shared__ double red[4 * NACC];
    float acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float dd = d[i];
        if (MODE == 0) {
            const float I = p1[i];
            acc[0] += sqrtf(I * dd);
            acc[1] += I;
        } else {
            const float a1 = p1[i], a2 = p2[i], a3 = p3[i];
            const float sd = sqrtf(dd);
            float df = sqrtf(fabsf(a1)) - sd;
            acc[kArrCand] += df * df;
            float gam = gamma0;
#pragma unroll
            for (int j = 0; j < kArrCand; ++j) {
                if (j < ncand) {
                    df = sqrtf(fabsf(a1 + (gam * gam) * a2 + gam * a3)) - sd;
                    acc[j] += df * df;
                }
                gam *= 0.5f;
            }
        }
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
        if (MODE == 0 || tid < ncand || tid == kArrCand) atomicAdd(sums + (MODE == 1 && tid == kArrCand ? ncand : tid), x);
    }
}

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

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
thread_local std::string g_err;

// kernel ids for the in-library profiler (ptycho_profile_read)
enum { K_COLS_FWD = 0, K_ROWS_FWD = 1, K_ROWS_INV = 2, K_COLS_ADJ_OBJ = 3, K_COLS_ADJ_PRB = 4, K_COLS_PLAIN = 5, K_SORT = 6, K_ROWS_STATS = 7, K_ROWS_PROJECT = 8, K_ROWS_LINESEARCH = 9, K_FWD_TEAM = 10, K_ROWS_ACCUM = 11, K_ARRAY_REDUCE = 12, K_ROWS_CROSS = 13, K_COLS_ARGMAX = 14, K_COUNT = 15 };

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(PTYCHO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

}  // namespace

struct ptycho_handle_s {
    Geom ge;
    c32* table = nullptr;     // exp(-2 pi i k / ndet)
    c32* scratch = nullptr;   // chunk * ndet^2 complex64
    long long chunk = 0;      // positions per launch pair
    // position sort (object / probe adjoint)
    unsigned long long* keys_a = nullptr;
    unsigned long long* keys_b = nullptr;
    int* vals_a = nullptr;
    int* order = nullptr;
    void* sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    c32* work[2] = {nullptr, nullptr};   // CG work buffers (column-pass intermediates), all positions
    int use_window = 1;       // 0: direct-atomics object adjoint (k_cols<ADJ_OBJ>)
    int use_team = 0;         // 1: forward operator as one persistent XCD-team launch (experimental)
    int use_pipeline = 0;     // 1: column and row passes of neighbouring chunks overlap on two streams (experimental)
    int profile_serial = 0;   // 1: no pipelining (set while the in-library profiler times kernels one by one)
    hipStream_t aux = nullptr;               // second stream of the pipeline
    std::vector<hipEvent_t> evs;             // reusable events (no timing)
    int trust_order = 0;      // 1: caller vouches that scan is unchanged since the last sort
    const float* order_scan = nullptr;   // scan pointer the current order was computed from
    c32* ring = nullptr;      // team ring: 8 XCD x 2 x Q tiles
    unsigned* ctrl = nullptr; // team control words
    int team_R = 0, team_Q = 0;
    int device = 0;
    int n_cu = 256;
    bool freed = false;
    bool profile = false;
    struct Span { int kid; hipEvent_t a, b; };
    std::vector<Span> spans;
};

namespace {

struct ProfSpan {   // brackets one launch with events when profiling is on
    ptycho_handle h;
    hipStream_t st;
    ptycho_handle_s::Span sp;
    bool on;
    ProfSpan(ptycho_handle h_, int kid, hipStream_t st_) : h(h_), st(st_), sp{kid, nullptr, nullptr}, on(h_->profile) {
        if (on) {
            on = hipEventCreate(&sp.a) == hipSuccess && hipEventCreate(&sp.b) == hipSuccess &&
                 hipEventRecord(sp.a, st) == hipSuccess;
        }
    }
    ~ProfSpan() {
        if (on && hipEventRecord(sp.b, st) == hipSuccess) h->spans.push_back(sp);
    }
};

long long default_chunk(const Geom& ge) {
    const char* env = std::getenv("PTYCHO_HIP_CHUNK");
    if (env && std::atoll(env) > 0) return std::atoll(env);
    // Large chunks stream best (measured: the row pass runs at ~5-6 TB/s for chunks
    // >= 256 MiB; small chunks only add launch gaps).  Cap the scratch at 4 GiB.
    const long long per = (long long)ge.ndet * ge.ndet * 8;
    long long c = (4ll << 30) / per;
    if (c < 16) c = 16;
    return c;
}

int alloc_scratch(ptycho_handle h) {
    if (h->scratch) {
        HIP_TRY(hipFree(h->scratch));
        h->scratch = nullptr;
    }
    const long long total = (long long)h->ge.ptheta * h->ge.nscan;
    long long c = h->chunk < total ? h->chunk : total;
    if (c < 1) c = 1;
    HIP_TRY(hipMalloc((void**)&h->scratch, (size_t)c * h->ge.ndet * h->ge.ndet * sizeof(c32)));
    return PTYCHO_OK;
}

int sort_positions(ptycho_handle h, const float* scan, hipStream_t st);   // ptycho_sort.hip-style helper below

template <int N, int DIR, int MODE>
int launch_cols(ptycho_handle h, ColArgs a, hipStream_t st) {
    using CC = ColCfg<N>;
    const int np = a.k_end - a.k_begin;
    if (np <= 0 || a.nstrips <= 0) return PTYCHO_OK;
    int target = h->n_cu * 8;
    int ng = target / a.nstrips;
    if (ng < 1) ng = 1;
    if (ng > np) ng = np;
    a.ngroups = ng;
    constexpr int kid = MODE == M_FWD ? K_COLS_FWD : MODE == M_ADJ_OBJ ? K_COLS_ADJ_OBJ : MODE == M_ADJ_PRB ? K_COLS_ADJ_PRB : K_COLS_PLAIN;
    {
        ProfSpan ps(h, kid, st);
        hipLaunchKernelGGL((k_cols<N, DIR, MODE>), dim3((unsigned)(a.nstrips * ng)), dim3(CC::NT), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

template <int N>
int launch_adjwin(ptycho_handle h, ColArgs a, hipStream_t st, int wg_target = 0) {
    using CC = ColCfg<N>;
    const int np = a.k_end - a.k_begin;
    if (np <= 0 || a.nstrips <= 0) return PTYCHO_OK;
    // contiguous runs of the sorted order; about 4 workgroups per CU in total
    if (wg_target <= 0) wg_target = h->n_cu * 4;
    int nseg = (wg_target + a.nstrips - 1) / a.nstrips;
    if (nseg < 1) nseg = 1;
    int seglen = (np + nseg - 1) / nseg;
    if (seglen < 8) seglen = 8;
    if (seglen > kRunMax) seglen = kRunMax;
    nseg = (np + seglen - 1) / seglen;
    static const int nt_mode_a = std::getenv("PTYCHO_HIP_NT") ? std::atoi(std::getenv("PTYCHO_HIP_NT")) : 0;
    a.nt = nt_mode_a;
    {
        ProfSpan ps(h, K_COLS_ADJ_OBJ, st);
        hipLaunchKernelGGL((k_cols_adjwin<N>), dim3((unsigned)(a.nstrips * nseg)), dim3(CC::NT), 0, st, a, seglen);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

template <int N, int MODE>
int launch_gatherwin(ptycho_handle h, ColArgs a, hipStream_t st, int wg_target = 0) {
    using CC = ColCfg<N>;
    const int np = a.k_end - a.k_begin;
    if (np <= 0 || a.nstrips <= 0) return PTYCHO_OK;
    if (wg_target <= 0) wg_target = h->n_cu * 4;
    int nseg = (wg_target + a.nstrips - 1) / a.nstrips;
    if (nseg < 1) nseg = 1;
    int seglen = (np + nseg - 1) / nseg;
    if (seglen < 8) seglen = 8;
    if (seglen > kRunMax) seglen = kRunMax;
    nseg = (np + seglen - 1) / seglen;
    if (const char* e = std::getenv("PTYCHO_HIP_COLSEGS")) {   // experiment knob: fewer, longer runs
        const int want = std::atoi(e);
        if (want > 0) {
            seglen = (np + want - 1) / want;
            if (seglen > kRunMax) seglen = kRunMax;
            nseg = (np + seglen - 1) / seglen;
        }
    }
    static const int nt_mode_g = std::getenv("PTYCHO_HIP_NT") ? std::atoi(std::getenv("PTYCHO_HIP_NT")) : 0;
    a.nt = nt_mode_g;
    {
        ProfSpan ps(h, MODE == M_FWD ? K_COLS_FWD : K_COLS_ADJ_PRB, st);
        hipLaunchKernelGGL((k_cols_gatherwin<N, MODE>), dim3((unsigned)(a.nstrips * nseg)), dim3(CC::NT), 0, st, a, seglen);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

template <int N, int DIR>
int launch_rows(ptycho_handle h, RowArgs a, hipStream_t st) {
    constexpr int B = 256 / Plan<N>::T;
    if (a.nrows <= 0) return PTYCHO_OK;
    long long nb = (a.nrows + B - 1) / B;
    long long grid = nb < (long long)h->n_cu * 8 ? nb : (long long)h->n_cu * 8;
    // nontemporal row-pass loads and stores: the rows are streamed once (measured 3-4 % on the pair;
    // nontemporal column-pass accesses made no difference).  PTYCHO_HIP_NT overrides (bit mask).
    static const int nt_mode = std::getenv("PTYCHO_HIP_NT") ? std::atoi(std::getenv("PTYCHO_HIP_NT")) : 3;
    a.nt = nt_mode;
    {
        ProfSpan ps(h, DIR < 0 ? K_ROWS_FWD : K_ROWS_INV, st);
        hipLaunchKernelGGL((k_rows<N, DIR>), dim3((unsigned)grid), dim3(256), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

template <int N>
void strip_range(const Geom& ge, int& strip0, int& nstrips) {
    constexpr int C = ColCfg<N>::C;
    strip0 = ge.pad / C;
    const int last = (ge.pad + ge.nprb - 1) / C;
    nstrips = last - strip0 + 1;
}


// ---- two-stream pipeline -----------------------------------------------------------------
// The column pass is latency/VALU bound and the row pass HBM bound, so running them one after
// the other leaves either the memory system or the ALUs idle.  Work is cut into chunks of the
// sorted order; the column pass of a chunk is launched with about one workgroup per CU, which
// leaves LDS and wave slots for the row pass of the neighbouring chunk on a second stream
// (a single column/row pair overlaps to ~0.87 of its summed time).  The caller's stream is
// joined before returning, so the call is still stream ordered from the outside.
// Measured at 4096 x 256^2: 3.13-3.66 ms per fwd+adj pair for 2-16 chunks and 256/512 column
// workgroups against 2.91 ms un-pipelined -- shorter runs, one workgroup per CU and 16+ extra
// launches cost more than the overlap returns.  Off by default (option "pipeline").
int pipeline_ready(ptycho_handle h, size_t nev) {
    if (!h->aux) HIP_TRY(hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
    while (h->evs.size() < nev) {
        hipEvent_t e;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        h->evs.push_back(e);
    }
    return PTYCHO_OK;
}

constexpr int kPipeChunksMax = 32;
static int pipe_chunks() { static const int v = std::getenv("PTYCHO_HIP_PIPE_CHUNKS") ? std::atoi(std::getenv("PTYCHO_HIP_PIPE_CHUNKS")) : 8; return v < 2 ? 2 : (v > kPipeChunksMax ? kPipeChunksMax : v); }
static int pipe_wgs(int n_cu) { static const int v = std::getenv("PTYCHO_HIP_PIPE_WGS") ? std::atoi(std::getenv("PTYCHO_HIP_PIPE_WGS")) : 0; return v > 0 ? v : n_cu; }

template <int N>
int do_fwd_pipelined(ptycho_handle h, c32* g, const c32* f, const float* scan, const c32* prb, hipStream_t st) {
    constexpr int C = ColCfg<N>::C;
    const Geom& ge = h->ge;
    const int total = ge.ptheta * ge.nscan;
    int strip0, nstrips;
    strip_range<N>(ge, strip0, nstrips);
    int rc = sort_positions(h, scan, st);
    if (rc) return rc;
    const int kPipeChunks = pipe_chunks();
    rc = pipeline_ready(h, kPipeChunks + 1);
    if (rc) return rc;
    const int per = (total + kPipeChunks - 1) / kPipeChunks;
    for (int c = 0; c < kPipeChunks; ++c) {
        const int k0 = c * per, k1 = k0 + per < total ? k0 + per : total;
        if (k0 >= k1) break;
        ColArgs ca{};
        ca.src = f; ca.dst = g; ca.aux = prb; ca.scan = scan; ca.table = h->table; ca.ge = ge;
        ca.order = h->order; ca.k_begin = k0; ca.k_end = k1; ca.strip0 = strip0; ca.nstrips = nstrips;
        if constexpr (WinCfg<N>::fits) rc = launch_gatherwin<N, M_FWD>(h, ca, st, pipe_wgs(h->n_cu));
        if (rc) return rc;
        HIP_TRY(hipEventRecord(h->evs[c], st));
        HIP_TRY(hipStreamWaitEvent(h->aux, h->evs[c], 0));
        RowArgs ra{};
        ra.src = g; ra.dst = g; ra.table = h->table; ra.tile_index = h->order + k0; ra.dst_indexed = 1;
        ra.nrows = (long long)(k1 - k0) * N; ra.xa = strip0 * C; ra.xb = (strip0 + nstrips) * C; ra.wa = 0; ra.wb = N;
        rc = launch_rows<N, -1>(h, ra, h->aux);
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(h->evs[kPipeChunks], h->aux));
    HIP_TRY(hipStreamWaitEvent(st, h->evs[kPipeChunks], 0));
    return PTYCHO_OK;
}

template <int N>
int do_adj_pipelined(ptycho_handle h, c32* f, const c32* g, const float* scan, c32* prb, int flg, hipStream_t st) {
    constexpr int C = ColCfg<N>::C;
    const Geom& ge = h->ge;
    const int total = ge.ptheta * ge.nscan;
    int strip0, nstrips;
    strip_range<N>(ge, strip0, nstrips);
    int rc = sort_positions(h, scan, st);
    if (rc) return rc;
    const int kPipeChunks = pipe_chunks();
    rc = pipeline_ready(h, kPipeChunks + 2);
    if (rc) return rc;
    // the row passes run on the second stream; they must see the caller's inputs and the sort
    HIP_TRY(hipEventRecord(h->evs[kPipeChunks + 1], st));
    HIP_TRY(hipStreamWaitEvent(h->aux, h->evs[kPipeChunks + 1], 0));
    const int per = (total + kPipeChunks - 1) / kPipeChunks;
    for (int c = 0; c < kPipeChunks; ++c) {
        const int k0 = c * per, k1 = k0 + per < total ? k0 + per : total;
        if (k0 >= k1) break;
        RowArgs ra{};
        ra.src = g; ra.dst = h->scratch + (size_t)k0 * N * N; ra.table = h->table; ra.tile_index = h->order + k0;
        ra.nrows = (long long)(k1 - k0) * N; ra.xa = 0; ra.xb = N; ra.wa = strip0 * C; ra.wb = (strip0 + nstrips) * C;
        rc = launch_rows<N, +1>(h, ra, h->aux);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(h->evs[c], h->aux));
        HIP_TRY(hipStreamWaitEvent(st, h->evs[c], 0));
        ColArgs ca{};
        ca.src = h->scratch + (size_t)k0 * N * N; ca.scan = scan; ca.table = h->table; ca.ge = ge;
        ca.order = h->order; ca.k_begin = k0; ca.k_end = k1; ca.strip0 = strip0; ca.nstrips = nstrips;
        if (flg == 0) {
            ca.dst = f; ca.aux = prb;
            if constexpr (WinCfg<N>::fits) rc = launch_adjwin<N>(h, ca, st, pipe_wgs(h->n_cu));
        } else {
            ca.dst = prb; ca.aux = f;
            if constexpr (WinCfg<N>::fits) rc = launch_gatherwin<N, M_ADJ_PRB>(h, ca, st, pipe_wgs(h->n_cu));
        }
        if (rc) return rc;
    }
    return PTYCHO_OK;
}

bool pipeline_applies(ptycho_handle h, long long total) {
    // enough positions for 8 chunks of >= 16 runs each, the whole farplane fits the scratch
    return h->use_pipeline && h->use_window && total >= 2048 && total <= h->chunk && !h->profile_serial;
}

template <int N>
int do_fwd_team(ptycho_handle h, c32* g, const c32* f, const float* scan, const c32* prb, hipStream_t st);

template <int N>
int do_fwd(ptycho_handle h, c32* g, const c32* f, const float* scan, const c32* prb, hipStream_t st) {
    constexpr int C = ColCfg<N>::C;
    const Geom& ge = h->ge;
    const long long total = (long long)ge.ptheta * ge.nscan;
    int strip0, nstrips;
    strip_range<N>(ge, strip0, nstrips);
    const bool window = h->use_window && WinCfg<N>::fits;
    int rc = PTYCHO_OK;
    if constexpr (N == 256) {
        if (h->use_team && window) return do_fwd_team<N>(h, g, f, scan, prb, st);
    }
    if constexpr (WinCfg<N>::fits) {
        if (pipeline_applies(h, total)) return do_fwd_pipelined<N>(h, g, f, scan, prb, st);
    }
    if (window) {
        rc = sort_positions(h, scan, st);
        if (rc) return rc;
    }
    // The column pass writes straight into g and the row pass transforms g in place, so the
    // forward operator needs no scratch and is issued as one launch pair over all positions.
    ColArgs ca{};
    ca.src = f; ca.dst = g; ca.aux = prb; ca.scan = scan; ca.table = h->table; ca.ge = ge;
    ca.k_begin = 0; ca.k_end = (int)total; ca.strip0 = strip0; ca.nstrips = nstrips;
    if (window) {
        ca.order = h->order;
        if constexpr (WinCfg<N>::fits) rc = launch_gatherwin<N, M_FWD>(h, ca, st);
    } else {
        ca.order = nullptr;
        rc = launch_cols<N, -1, M_FWD>(h, ca, st);
    }
    if (rc) return rc;
    RowArgs ra{};
    ra.src = g; ra.dst = g; ra.table = h->table; ra.tile_index = nullptr;
    ra.nrows = total * N; ra.xa = strip0 * C; ra.xb = (strip0 + nstrips) * C; ra.wa = 0; ra.wb = N;
    return launch_rows<N, -1>(h, ra, st);
}


template <int N>
int do_fwd_team(ptycho_handle h, c32* g, const c32* f, const float* scan, const c32* prb, hipStream_t st) {
    constexpr int C = ColCfg<N>::C;
    const Geom& ge = h->ge;
    const int total = ge.ptheta * ge.nscan;
    int strip0, nstrips;
    strip_range<N>(ge, strip0, nstrips);
    const size_t tile = (size_t)N * N * sizeof(c32);
    int Q = (int)((1u << 20) / tile);   // about 1 MiB of tiles per ring buffer
    if (Q < 1) Q = 1;
    const int per = (total + 7) / 8;
    const int R = (per + Q - 1) / Q + 1;
    if (!h->ring || h->team_Q != Q || h->team_R < R) {
        if (h->ring) { HIP_TRY(hipFree(h->ring)); h->ring = nullptr; }
        if (h->ctrl) { HIP_TRY(hipFree(h->ctrl)); h->ctrl = nullptr; }
        HIP_TRY(hipMalloc((void**)&h->ring, (size_t)8 * 2 * Q * tile));
        HIP_TRY(hipMalloc((void**)&h->ctrl, (size_t)(16 + 8 * 2 * R) * sizeof(unsigned)));
        h->team_Q = Q; h->team_R = R;
    }
    int rc = sort_positions(h, scan, st);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(h->ctrl, 0, (size_t)(16 + 8 * 2 * h->team_R) * sizeof(unsigned), st));
    TeamArgs ta{};
    ta.f = f; ta.g = g; ta.prb = prb; ta.scan = scan; ta.table = h->table; ta.ge = ge;
    ta.order = h->order; ta.total = total; ta.ring = h->ring; ta.ctrl = h->ctrl; ta.R = h->team_R; ta.Q = Q;
    ta.strip0 = strip0; ta.nstrips = nstrips; ta.xa = strip0 * C; ta.xb = (strip0 + nstrips) * C;
    {
        ProfSpan ps(h, K_FWD_TEAM, st);
        hipLaunchKernelGGL((k_fwd_team<N>), dim3((unsigned)(h->n_cu * 2)), dim3(256), 0, st, ta);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

template <int N>
int do_adj(ptycho_handle h, c32* f, const c32* g, const float* scan, c32* prb, int flg, hipStream_t st) {
    constexpr int C = ColCfg<N>::C;
    const Geom& ge = h->ge;
    const long long total = (long long)ge.ptheta * ge.nscan;
    int strip0, nstrips;
    strip_range<N>(ge, strip0, nstrips);
    if constexpr (WinCfg<N>::fits) {
        if (pipeline_applies(h, total)) return do_adj_pipelined<N>(h, f, g, scan, prb, flg, st);
    }
    const bool window = flg == 0 && h->use_window && WinCfg<N>::fits;
    // positions are visited in sorted order (angle, column bucket, row): neighbours in the
    // object are neighbours in time, which is what the LDS overlap-add window needs
    int rc = sort_positions(h, scan, st);
    if (rc) return rc;
    for (long long k0 = 0; k0 < total; k0 += h->chunk) {
        const long long k1 = k0 + h->chunk < total ? k0 + h->chunk : total;
        RowArgs ra{};
        ra.src = g; ra.dst = h->scratch; ra.table = h->table; ra.tile_index = h->order + k0;
        ra.nrows = (k1 - k0) * N; ra.xa = 0; ra.xb = N; ra.wa = strip0 * C; ra.wb = (strip0 + nstrips) * C;
        rc = launch_rows<N, +1>(h, ra, st);
        if (rc) return rc;
        ColArgs ca{};
        ca.src = h->scratch; ca.scan = scan; ca.table = h->table; ca.ge = ge;
        ca.order = h->order; ca.k_begin = (int)k0; ca.k_end = (int)k1; ca.strip0 = strip0; ca.nstrips = nstrips;
        if (flg == 0) {
            ca.dst = f; ca.aux = prb;
            if (window) {
                if constexpr (WinCfg<N>::fits) rc = launch_adjwin<N>(h, ca, st);
            } else {
                rc = launch_cols<N, +1, M_ADJ_OBJ>(h, ca, st);
            }
        } else {
            ca.dst = prb; ca.aux = f;
            if (h->use_window && WinCfg<N>::fits) {
                if constexpr (WinCfg<N>::fits) rc = launch_gatherwin<N, M_ADJ_PRB>(h, ca, st);
            } else {
                rc = launch_cols<N, +1, M_ADJ_PRB>(h, ca, st);
            }
        }
        if (rc) return rc;
    }
    return PTYCHO_OK;
}

template <int N>
int do_fft2(ptycho_handle h, c32* dst, const c32* src, long long nbatch, int dir, hipStream_t st) {
    constexpr int C = ColCfg<N>::C;
    RowArgs ra{};
    ra.src = src; ra.dst = dst; ra.table = h->table; ra.nrows = nbatch * N; ra.tile_index = nullptr;
    ra.xa = 0; ra.xb = N; ra.wa = 0; ra.wb = N;
    int rc = dir < 0 ? launch_rows<N, -1>(h, ra, st) : launch_rows<N, +1>(h, ra, st);
    if (rc) return rc;
    // column pass in place, in slices small enough for 32-bit position indices
    const long long slice = 1 << 20;
    for (long long b0 = 0; b0 < nbatch; b0 += slice) {
        const long long b1 = b0 + slice < nbatch ? b0 + slice : nbatch;
        ColArgs ca{};
        ca.src = dst + (size_t)b0 * N * N; ca.dst = dst + (size_t)b0 * N * N; ca.table = h->table; ca.ge = h->ge;
        ca.order = nullptr; ca.k_begin = 0; ca.k_end = (int)(b1 - b0); ca.strip0 = 0; ca.nstrips = N / C;
        rc = dir < 0 ? launch_cols<N, -1, M_PLAIN>(h, ca, st) : launch_cols<N, +1, M_PLAIN>(h, ca, st);
        if (rc) return rc;
    }
    return PTYCHO_OK;
}


// ---- CG-stage helpers ----------------------------------------------------------------
int ensure_work(ptycho_handle h, int slot) {
    if (slot < 0 || slot > 1) return fail(PTYCHO_ERR_ARG, "work slot must be 0 or 1");
    if (!h->work[slot]) {
        const size_t total = (size_t)h->ge.ptheta * h->ge.nscan;
        HIP_TRY(hipMalloc((void**)&h->work[slot], total * h->ge.ndet * h->ge.ndet * sizeof(c32)));
    }
    return PTYCHO_OK;
}

template <int N>
int do_cg_fwd_cols(ptycho_handle h, int slot, const c32* f, const float* scan, const c32* prb, hipStream_t st) {
    const Geom& ge = h->ge;
    const long long total = (long long)ge.ptheta * ge.nscan;
    int strip0, nstrips;
    strip_range<N>(ge, strip0, nstrips);
    const bool window = h->use_window && WinCfg<N>::fits;
    int rc = PTYCHO_OK;
    if (window) {
        rc = sort_positions(h, scan, st);
        if (rc) return rc;
    }
    ColArgs ca{};
    ca.src = f; ca.dst = h->work[slot]; ca.aux = prb; ca.scan = scan; ca.table = h->table; ca.ge = ge;
    ca.k_begin = 0; ca.k_end = (int)total; ca.strip0 = strip0; ca.nstrips = nstrips;
    if (window) {
        ca.order = h->order;
        if constexpr (WinCfg<N>::fits) rc = launch_gatherwin<N, M_FWD>(h, ca, st);
    } else {
        ca.order = nullptr;
        rc = launch_cols<N, -1, M_FWD>(h, ca, st);
    }
    return rc;
}

template <int N>
int do_cg_adj_cols(ptycho_handle h, int slot, c32* f, const float* scan, c32* prb, int flg, hipStream_t st) {
    const Geom& ge = h->ge;
    const long long total = (long long)ge.ptheta * ge.nscan;
    int strip0, nstrips;
    strip_range<N>(ge, strip0, nstrips);
    int rc = sort_positions(h, scan, st);
    if (rc) return rc;
    ColArgs ca{};
    ca.src = h->work[slot]; ca.scan = scan; ca.table = h->table; ca.ge = ge; ca.natural_tiles = 1;
    ca.order = h->order; ca.k_begin = 0; ca.k_end = (int)total; ca.strip0 = strip0; ca.nstrips = nstrips;
    const bool window = h->use_window && WinCfg<N>::fits;
    if (flg == 0) {
        ca.dst = f; ca.aux = prb;
        if (window) {
            if constexpr (WinCfg<N>::fits) rc = launch_adjwin<N>(h, ca, st);
        } else {
            rc = launch_cols<N, +1, M_ADJ_OBJ>(h, ca, st);
        }
    } else {
        ca.dst = prb; ca.aux = f;
        if (window) {
            if constexpr (WinCfg<N>::fits) rc = launch_gatherwin<N, M_ADJ_PRB>(h, ca, st);
        } else {
            rc = launch_cols<N, +1, M_ADJ_PRB>(h, ca, st);
        }
    }
    return rc;
}

template <int N, int EP>
int do_cg_rows(ptycho_handle h, RowFusedArgs a, hipStream_t st) {
    constexpr int C = ColCfg<N>::C;
    constexpr int B = 256 / Plan<N>::T;
    int strip0, nstrips;
    strip_range<N>(h->ge, strip0, nstrips);
    a.table = h->table;
    a.nrows = (long long)h->ge.ptheta * h->ge.nscan * N;
    a.xa = strip0 * C; a.xb = (strip0 + nstrips) * C;
    long long nb = (a.nrows + B - 1) / B;
    long long grid = nb < (long long)h->n_cu * 8 ? nb : (long long)h->n_cu * 8;
    {
        ProfSpan ps(h, EP == EP_STATS ? K_ROWS_STATS : EP == EP_PROJECT ? K_ROWS_PROJECT : EP == EP_LINESEARCH ? K_ROWS_LINESEARCH : EP == EP_CROSS ? K_ROWS_CROSS : K_ROWS_ACCUM, st);
        hipLaunchKernelGGL((k_rows_fused<N, EP>), dim3((unsigned)grid), dim3(256), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

#define PTY_DISPATCH(N_, CALL)                                   \
    switch (N_) {                                                \
        case 16: { constexpr int NN = 16; return CALL; }         \
        case 32: { constexpr int NN = 32; return CALL; }         \
        case 64: { constexpr int NN = 64; return CALL; }         \
        case 128: { constexpr int NN = 128; return CALL; }       \
        case 256: { constexpr int NN = 256; return CALL; }       \
        case 512: { constexpr int NN = 512; return CALL; }       \
        case 1024: { constexpr int NN = 1024; return CALL; }     \
        default: return fail(PTYCHO_ERR_ARG, "ndet must be a power of two in [16, 1024]"); \
    }

int check_handle(ptycho_handle h) {
    if (!h) return fail(PTYCHO_ERR_ARG, "null handle");
    if (h->freed) return fail(PTYCHO_ERR_FREED, "handle used after ptycho_free");
    return PTYCHO_OK;
}

int sort_positions(ptycho_handle h, const float* scan, hipStream_t st) {
    const int total = h->ge.ptheta * h->ge.nscan;
    // The order depends only on the scan positions.  A caller that knows they have not
    // changed since the previous call on this handle (option "trust_order") skips the sort.
    if (h->trust_order && h->order_scan == scan) return PTYCHO_OK;
    h->order_scan = scan;
    ProfSpan ps(h, K_SORT, st);
    hipLaunchKernelGGL(k_sort_keys, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, scan, h->ge, total,
                       h->keys_a, h->vals_a);
    HIP_TRY(hipGetLastError());
    size_t bytes = h->sort_tmp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(h->sort_tmp, bytes, h->keys_a, h->keys_b, h->vals_a, h->order,
                                      (size_t)total, 0u, 64u, st));
    return PTYCHO_OK;
}

int alloc_sort(ptycho_handle h) {
    const size_t total = (size_t)h->ge.ptheta * h->ge.nscan;
    HIP_TRY(hipMalloc((void**)&h->keys_a, total * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void**)&h->keys_b, total * sizeof(unsigned long long)));
    HIP_TRY(hipMalloc((void**)&h->vals_a, total * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&h->order, total * sizeof(int)));
    size_t bytes = 0;
    HIP_TRY(rocprim::radix_sort_pairs(nullptr, bytes, h->keys_a, h->keys_b, h->vals_a, h->order, total, 0u, 64u,
                                      (hipStream_t)0));
    h->sort_tmp_bytes = bytes ? bytes : 16;
    HIP_TRY(hipMalloc(&h->sort_tmp, h->sort_tmp_bytes));
    return PTYCHO_OK;
}

void release(ptycho_handle h) {
    void* ptrs[] = {h->table, h->scratch, h->keys_a, h->keys_b, h->vals_a, h->order, h->sort_tmp, h->work[0], h->work[1], h->ring, h->ctrl};
    h->ring = nullptr; h->ctrl = nullptr;
    h->work[0] = nullptr; h->work[1] = nullptr;
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    h->table = nullptr; h->scratch = nullptr; h->keys_a = nullptr; h->keys_b = nullptr;
    h->vals_a = nullptr; h->order = nullptr; h->sort_tmp = nullptr;
    for (auto& sp : h->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    h->spans.clear();
    for (auto& e : h->evs) (void)hipEventDestroy(e);
    h->evs.clear();
    if (h->aux) { (void)hipStreamDestroy(h->aux); h->aux = nullptr; }
}

}  // namespace

extern "C" {

const char* ptycho_last_error(void) { return g_err.c_str(); }
const char* ptycho_version(void) { return "ptychohip 0.2 (gfx950)"; }

int ptycho_create(ptycho_handle* out, size_t ptheta, size_t nz, size_t n, size_t nscan, size_t ndet, size_t nprb) {
    if (!out) return fail(PTYCHO_ERR_ARG, "out is null");
    *out = nullptr;
    if (ptheta == 0 || nz == 0 || n == 0 || nscan == 0 || ndet == 0 || nprb == 0)
        return fail(PTYCHO_ERR_ARG, "all sizes must be positive");
    if (ndet < 16 || ndet > 1024 || (ndet & (ndet - 1)) != 0)
        return fail(PTYCHO_ERR_ARG, "ndet must be a power of two in [16, 1024]");
    if (nprb > ndet) return fail(PTYCHO_ERR_ARG, "nprb must be <= ndet");
    if (ptheta * nscan > (size_t)0x7fffffff / 2 || ptheta > (1u << 19) || nz > 65536 * 4 || n > 65536 * 4)
        return fail(PTYCHO_ERR_ARG, "problem too large for 32-bit position indices");
    ptycho_handle h = new ptycho_handle_s();
    h->ge = Geom{(int)ptheta, (int)nz, (int)n, (int)nscan, (int)ndet, (int)nprb, (int)((ndet - nprb) / 2)};
    hipError_t e = hipGetDevice(&h->device);
    if (e != hipSuccess) {
        delete h;
        return fail(PTYCHO_ERR_HIP, std::string("hipGetDevice: ") + hipGetErrorString(e));
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount > 0)
        h->n_cu = prop.multiProcessorCount;
    std::vector<c32> tab(ndet);
    for (size_t k = 0; k < ndet; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)ndet;
        tab[k] = c32{(float)std::cos(ang), (float)std::sin(ang)};
    }
    e = hipMalloc((void**)&h->table, ndet * sizeof(c32));
    if (e == hipSuccess) e = hipMemcpy(h->table, tab.data(), ndet * sizeof(c32), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        release(h);
        delete h;
        return fail(PTYCHO_ERR_HIP, std::string("twiddle table: ") + hipGetErrorString(e));
    }
    const char* env = std::getenv("PTYCHO_HIP_WINDOW");
    if (env) h->use_window = std::atoi(env) != 0;
    env = std::getenv("PTYCHO_HIP_PIPELINE");
    if (env) h->use_pipeline = std::atoi(env) != 0;
    env = std::getenv("PTYCHO_HIP_TEAM");
    if (env) h->use_team = std::atoi(env) != 0;
    h->chunk = default_chunk(h->ge);
    int rc = alloc_scratch(h);
    if (!rc) rc = alloc_sort(h);
    if (rc) {
        release(h);
        delete h;
        return rc;
    }
    *out = h;
    return PTYCHO_OK;
}

int ptycho_free(ptycho_handle h) {
    if (!h) return fail(PTYCHO_ERR_ARG, "null handle");
    if (!h->freed) {
        h->freed = true;
        release(h);
    }
    return PTYCHO_OK;
}

int ptycho_destroy(ptycho_handle h) {
    if (!h) return PTYCHO_OK;
    ptycho_free(h);
    delete h;
    return PTYCHO_OK;
}

long long ptycho_get(ptycho_handle h, int which) {
    if (!h) return -1;
    switch (which) {
        case 0: return h->ge.ptheta;
        case 1: return h->ge.nz;
        case 2: return h->ge.n;
        case 3: return h->ge.nscan;
        case 4: return h->ge.ndet;
        case 5: return h->ge.nprb;
        case 100: return h->chunk;
        case 101: return h->use_window;
        case 102: {   // team abort word of the last team launch (synchronises the device)
            if (!h->ctrl) return 0;
            unsigned w = 0;
            if (hipDeviceSynchronize() != hipSuccess) return -2;
            if (hipMemcpy(&w, h->ctrl + 9, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -2;
            return (long long)w;
        }
        default: return -1;
    }
}

int ptycho_set_option(ptycho_handle h, const char* name, long long value) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!name) return fail(PTYCHO_ERR_ARG, "null option name");
    if (std::strcmp(name, "chunk") == 0) {
        h->chunk = value > 0 ? value : default_chunk(h->ge);
        HIP_TRY(hipDeviceSynchronize());
        return alloc_scratch(h);
    }
    if (std::strcmp(name, "window") == 0) {
        h->use_window = value != 0;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "trust_order") == 0) {
        h->trust_order = value != 0;
        if (!h->trust_order) h->order_scan = nullptr;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "pipeline") == 0) {
        h->use_pipeline = value != 0;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "team") == 0) {
        h->use_team = value != 0;
        return PTYCHO_OK;
    }
    return fail(PTYCHO_ERR_ARG, std::string("unknown option ") + name);
}

int ptycho_profile(ptycho_handle h, int enable) {
    int rc = check_handle(h);
    if (rc) return rc;
    h->profile = enable != 0;
    return PTYCHO_OK;
}

int ptycho_profile_read(ptycho_handle h, double* ms, long long* launches, int n) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!ms || !launches || n < K_COUNT) return fail(PTYCHO_ERR_ARG, "need arrays of at least 15 entries");
    for (int i = 0; i < n; ++i) { ms[i] = 0.0; launches[i] = 0; }
    for (auto& sp : h->spans) {
        HIP_TRY(hipEventSynchronize(sp.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, sp.a, sp.b));
        ms[sp.kid] += t;
        launches[sp.kid] += 1;
        (void)hipEventDestroy(sp.a);
        (void)hipEventDestroy(sp.b);
    }
    h->spans.clear();
    return PTYCHO_OK;
}

int ptycho_fwd(ptycho_handle h, void* g, const void* f, const void* scan, const void* prb, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!g || !f || !scan || !prb) return fail(PTYCHO_ERR_ARG, "null operand");
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_fwd<NN>(h, (c32*)g, (const c32*)f, (const float*)scan, (const c32*)prb, st)));
}

int ptycho_adj(ptycho_handle h, void* f, const void* g, const void* scan, void* prb, int flg, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!g || !f || !scan || !prb) return fail(PTYCHO_ERR_ARG, "null operand");
    if (flg != 0 && flg != 1) return fail(PTYCHO_ERR_ARG, "flg must be 0 (object) or 1 (probe)");
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_adj<NN>(h, (c32*)f, (const c32*)g, (const float*)scan, (c32*)prb, flg, st)));
}

int ptycho_cg_fwd_cols(ptycho_handle h, int slot, const void* f, const void* scan, const void* prb, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!f || !scan || !prb) return fail(PTYCHO_ERR_ARG, "null operand");
    rc = ensure_work(h, slot);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_fwd_cols<NN>(h, slot, (const c32*)f, (const float*)scan, (const c32*)prb, st)));
}

int ptycho_cg_adj_cols(ptycho_handle h, int slot, void* f, const void* scan, void* prb, int flg, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!f || !scan || !prb) return fail(PTYCHO_ERR_ARG, "null operand");
    if (flg != 0 && flg != 1) return fail(PTYCHO_ERR_ARG, "flg must be 0 (object) or 1 (probe)");
    if (slot < 0 || slot > 1 || !h->work[slot]) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_adj_cols<NN>(h, slot, (c32*)f, (const float*)scan, (c32*)prb, flg, st)));
}

int ptycho_cg_stats(ptycho_handle h, int slot, const void* data, double* sums, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!data || !sums) return fail(PTYCHO_ERR_ARG, "null operand");
    if (slot < 0 || slot > 1 || !h->work[slot]) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    RowFusedArgs a{};
    a.s1 = h->work[slot]; a.data = (const float*)data; a.sums = sums;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_STATS>(h, a, st)));
}

int ptycho_cg_project(ptycho_handle h, int src_slot, int dst_slot, const void* data, const double* ab,
                      double* cost, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!data || !cost) return fail(PTYCHO_ERR_ARG, "null operand");
    if (src_slot < 0 || src_slot > 1 || !h->work[src_slot]) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    rc = ensure_work(h, dst_slot);
    if (rc) return rc;
    RowFusedArgs a{};
    a.s1 = h->work[src_slot]; a.out = h->work[dst_slot]; a.data = (const float*)data; a.sums = cost; a.ab = ab;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_PROJECT>(h, a, st)));
}

int ptycho_cg_linesearch(ptycho_handle h, int slot1, int slot2, const void* data, const double* ab, double gamma0,
                         int ncand, double* costs, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!data || !costs) return fail(PTYCHO_ERR_ARG, "null operand");
    if (ncand < 1 || ncand > kMaxCand) return fail(PTYCHO_ERR_ARG, "ncand must be in [1, 16]");
    if (slot1 < 0 || slot1 > 1 || slot2 < 0 || slot2 > 1 || !h->work[slot1] || !h->work[slot2])
        return fail(PTYCHO_ERR_ARG, "work slot is empty");
    RowFusedArgs a{};
    a.s1 = h->work[slot1]; a.s2 = h->work[slot2]; a.data = (const float*)data; a.sums = costs; a.ab = ab;
    a.gamma0 = (float)gamma0; a.ncand = ncand;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_LINESEARCH>(h, a, st)));
}

int ptycho_cg_project_multi(ptycho_handle h, int src_slot, int dst_slot, const void* data, const void* inten,
                            const double* ab, double* cost, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!data || !cost || !inten) return fail(PTYCHO_ERR_ARG, "null operand");
    if (src_slot < 0 || src_slot > 1 || !h->work[src_slot]) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    rc = ensure_work(h, dst_slot);
    if (rc) return rc;
    RowFusedArgs a{};
    a.s1 = h->work[src_slot]; a.out = h->work[dst_slot]; a.data = (const float*)data; a.sums = cost; a.ab = ab;
    a.inten = (const float*)inten;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_PROJECT>(h, a, st)));
}

int ptycho_cg_accum_intensity(ptycho_handle h, int slot, void* inten, int first, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!inten) return fail(PTYCHO_ERR_ARG, "null operand");
    if (slot < 0 || slot > 1 || !h->work[slot]) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    RowFusedArgs a{};
    a.s1 = h->work[slot]; a.acc1 = (float*)inten; a.first = first;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_ACCUM_I>(h, a, st)));
}

int ptycho_cg_accum_terms(ptycho_handle h, int slot1, int slot2, void* p1, void* p2, void* p3, int first,
                          const double* ab, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!p1 || !p2 || !p3) return fail(PTYCHO_ERR_ARG, "null operand");
    if (slot1 < 0 || slot1 > 1 || slot2 < 0 || slot2 > 1 || !h->work[slot1] || !h->work[slot2])
        return fail(PTYCHO_ERR_ARG, "work slot is empty");
    RowFusedArgs a{};
    a.s1 = h->work[slot1]; a.s2 = h->work[slot2]; a.acc1 = (float*)p1; a.acc2 = (float*)p2; a.acc3 = (float*)p3;
    a.first = first; a.ab = ab;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_ACCUM_P>(h, a, st)));
}

int ptycho_cg_array_stats(ptycho_handle h, const void* inten, const void* data, double* sums, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!inten || !data || !sums) return fail(PTYCHO_ERR_ARG, "null operand");
    const long long n = (long long)h->ge.ptheta * h->ge.nscan * h->ge.ndet * h->ge.ndet;
    hipStream_t st = (hipStream_t)stream;
    {
        ProfSpan ps(h, K_ARRAY_REDUCE, st);
        hipLaunchKernelGGL((k_array_reduce<0>), dim3((unsigned)(h->n_cu * 8)), dim3(256), 0, st, (const float*)inten,
                           (const float*)nullptr, (const float*)nullptr, (const float*)data, n, 0.0f, 0, sums);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

int ptycho_cg_array_costs(ptycho_handle h, const void* p1, const void* p2, const void* p3, const void* data,
                          double gamma0, int ncand, double* costs, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!p1 || !p2 || !p3 || !data || !costs) return fail(PTYCHO_ERR_ARG, "null operand");
    if (ncand < 1 || ncand > kArrCand) return fail(PTYCHO_ERR_ARG, "ncand must be in [1, 32]");
    const long long n = (long long)h->ge.ptheta * h->ge.nscan * h->ge.ndet * h->ge.ndet;
    hipStream_t st = (hipStream_t)stream;
    {
        ProfSpan ps(h, K_ARRAY_REDUCE, st);
        hipLaunchKernelGGL((k_array_reduce<1>), dim3((unsigned)(h->n_cu * 8)), dim3(256), 0, st, (const float*)p1,
                           (const float*)p2, (const float*)p3, (const float*)data, n, (float)gamma0, ncand, costs);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

int ptycho_fft2(ptycho_handle h, void* dst, const void* src, size_t nbatch, int dir, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!dst || !src) return fail(PTYCHO_ERR_ARG, "null operand");
    if (dir != -1 && dir != 1) return fail(PTYCHO_ERR_ARG, "dir must be -1 (forward) or +1 (inverse)");
    if (nbatch == 0) return PTYCHO_OK;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_fft2<NN>(h, (c32*)dst, (const c32*)src, (long long)nbatch, dir, st)));
}

}  // extern "C"

template <int N>
int do_cg_argmax(ptycho_handle h, int slot, unsigned long long* best, hipStream_t st) {
    using CC = ColCfg<N>;
    const int npos = h->ge.ptheta * h->ge.nscan;
    constexpr int nstrips = N / CC::C;
    int ng = (h->n_cu * 8) / nstrips;
    if (ng < 1) ng = 1;
    if (ng > npos) ng = npos;
    HIP_TRY(hipMemsetAsync(best, 0, (size_t)npos * sizeof(unsigned long long), st));
    {
        ProfSpan ps(h, K_COLS_ARGMAX, st);
        hipLaunchKernelGGL((k_cols_argmax<N>), dim3((unsigned)(nstrips * ng)), dim3(CC::NT), 0, st,
                           (const c32*)h->work[slot], (const c32*)h->table, best, npos, ng);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

extern "C" int ptycho_cg_cross(ptycho_handle h, int slot1, int slot2, double gamma, void* image_product, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!image_product) return fail(PTYCHO_ERR_ARG, "null operand");
    if (h->ge.nprb != h->ge.ndet) {
        // with a padded probe the zero columns of the slots are not materialised; the row pass masks them
    }
    if (slot1 < 0 || slot1 > 1 || slot2 < 0 || slot2 > 1 || !h->work[slot1] || !h->work[slot2])
        return fail(PTYCHO_ERR_ARG, "work slot is empty");
    RowFusedArgs a{};
    a.s1 = h->work[slot1]; a.s2 = h->work[slot2]; a.out = h->work[slot2]; a.ip = (c32*)image_product;
    a.gamma0 = (float)gamma;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_CROSS>(h, a, st)));
}

extern "C" int ptycho_cg_argmax(ptycho_handle h, int slot, void* best, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!best) return fail(PTYCHO_ERR_ARG, "null operand");
    if (slot < 0 || slot > 1 || !h->work[slot]) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_argmax<NN>(h, slot, (unsigned long long*)best, st)));
}
