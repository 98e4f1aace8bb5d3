// k_fwd_fused.hpp -- forward operator for ndet = 256 as ONE launch: the column<->row intermediate never leaves the CU
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
//
// Replaces, for ndet = 256, the pair k_cols_gatherwin<FWD> + k_rows_split of ptychofft::fwd
// (/root/reference/src/cuda/ptychofft.cu:60-73: memset + muloperator flg 2 + cufftExecC2C).
//
// A 256 x 256 complex64 tile is 512 KiB and does not fit one CU (160 KiB LDS, 512 KiB VGPR), which is why
// the two-pass operator sends the intermediate through HBM.  Here the DFT over y is split 4 x 64
// (y = 64 n1 + n2, k = k1 + 4 k2):
//
//   C[k1 + 4 k2, x] = sum_n2 W64^(n2 k2) { W256^(n2 k1) sum_n1 W4^(n1 k1) near[64 n1 + n2, x] }
//
// For ONE class k1 the braces are a 64 x 256 tile (128 KiB): it fits LDS, and its 64-point column DFT
// and 256-point row DFT give the 64 complete rows k1 + 4 k2 of g.  The price is that the exit wave
// near = (c prb) . bilerp(psi) is recomputed per class group (the object and the probe come from L2,
// not from HBM): TILES = 1 makes one class per pass (4 passes per position), TILES = 2 makes classes
// {h, h + 2} per pass from the same products (2 passes per position; the second tile waits in VGPRs).
// HBM traffic = the algorithmic bytes: g is written once, nothing else is written.
//
// Workgroup = 1024 threads, one per CU (LDS 155 KiB), persistent over items (position, class group).
//   phase A  8 chunks of 8 values of n2: the object rows of the chunk (4 bands x 9 rows x 258 columns) are
//            staged in LDS (double buffered, fetched one chunk ahead), thread (x, j0) forms the 4 products
//            of each of its 2 slots from LDS taps and the padded probe, folds them over n1 and twiddles;
//            after 8 chunks it holds the 16 step-0 inputs of the 64-point column DFT of column x.
//   column   Plan<64> = 8 x 8: the 4 threads of a column sit in one wave, so the exchange through LDS
//            needs no workgroup barrier; result rows k2 are written in the row-pass layout.
//   row      Plan<256> = 16 x 16 on 64 rows: the 16 threads of a row sit in one wave as well.
#pragma once

struct FusedArgs {
    const c32* f;       // object [ptheta][nz][n]
    c32* g;             // farplane [ptheta*nscan][256][256]
    const c32* prbp;    // padded probe c * prb, [ptheta][256][256], zero outside the probe
    const float* scan;
    const c32* table;   // exp(-2 pi i k / 256)
    const int* order;   // processing order (nullptr: identity)
    Geom ge;
    int total;          // positions
};

// c * prb in a zero-bordered 256 x 256 frame (kernels.cu:48-65: centred pad, c = 1/ndet)
__global__ void k_pad_probe(const c32* __restrict__ prb, c32* __restrict__ out, const Geom ge) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = ge.ndet;
    if (i >= ge.ptheta * N * N) return;
    const int t = i / (N * N), y = (i / N) % N, x = i % N;
    const int iy = y - ge.pad, ix = x - ge.pad;
    const bool ok = iy >= 0 && iy < ge.nprb && ix >= 0 && ix < ge.nprb;
    const float cinv = 1.0f / (float)N;
    out[i] = ok ? prb[((size_t)t * ge.nprb + iy) * ge.nprb + ix] * cinv : c32{0.0f, 0.0f};
}


__device__ __forceinline__ c32 mul_mi(c32 a) { return c32{a.y, -a.x}; }   // a * (-i)
__device__ __forceinline__ c32 mul_pi(c32 a) { return c32{-a.y, a.x}; }   // a * (+i)

template <int TILES>
__global__ __launch_bounds__(1024) void k_fwd_fused256(const FusedArgs a) {
    static_assert(TILES == 1 || TILES == 2, "one or two class tiles per pass");
    constexpr int N = 256;
    using PC = Plan<64>;
    using PR = Plan<256>;
    using FC = Fft<PC, -1>;
    using FR = Fft<PR, -1>;
    using L = RowLds<N>;
    constexpr int FS = L::FS;                 // 272: tile row stride (complex), 16 mod 32 -> conflict free
    constexpr int SW = 272;                   // staging row stride
    constexpr int SR = 36;                    // staging rows per chunk: 4 bands x (8 + 1)
    constexpr int SQ = 129;                   // 16-byte columns staged per row (258 complex)
    constexpr int SBUF = SR * SW;             // complex per staging buffer
    constexpr int NLD = (SR * SQ + 1023) / 1024;   // staging loads per thread per chunk (5)
    constexpr int NCLS = 4 / TILES;
    static_assert(64 * FS <= 2 * SBUF, "the tile aliases the two staging buffers");
    __shared__ __attribute__((aligned(16))) c32 lds[2 * SBUF];
    __shared__ c32 wtab[N];

    typedef float f4 __attribute__((ext_vector_type(4)));
    const Geom ge = a.ge;
    const c32 zero = c32{0.0f, 0.0f};

    if (threadIdx.x < N) wtab[threadIdx.x] = a.table[threadIdx.x];
    __syncthreads();

    const int nitems = a.total * NCLS;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        // Thread coordinates are re-derived per item from an opaque copy of the thread id: everything that
        // depends only on them (probe offsets of 64 rows, staging rows / columns, LDS offsets, twiddles)
        // would otherwise be hoisted out of the item loop and kept live -- far more than the 128 VGPRs a
        // 1024-thread workgroup has per thread.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wv = tid >> 6;
        const int xl = lane & 15, j0 = lane >> 4;     // column phase: thread (x, j0), 4 threads per column in one wave
        const int x = 16 * wv + xl;
        const int fr = tid >> 4, j0r = tid & 15;      // row phase: thread (row fr, j0r), 16 threads per row in one wave
        const int k = item / NCLS, cls = item % NCLS;
        const int p = a.order ? a.order[k] : k;
        const int t = p / ge.nscan;
        const Pos q = decode_xy(a.scan[2 * (size_t)p], a.scan[2 * (size_t)p + 1], ge);
        c32* gt = a.g + (size_t)p * N * N;
        if (!q.valid) {   // skipped position: exact zeros (memset of ptychofft.cu:69)
#pragma unroll
            for (int tl = 0; tl < TILES; ++tl) {
                c32* row = gt + (size_t)(cls + 2 * tl + 4 * fr) * N;
#pragma unroll
                for (int m = 0; m < 16; ++m) row[j0r + 16 * m] = zero;
            }
            continue;
        }
        const c32* ft = a.f + (size_t)t * ge.nz * ge.n;
        const c32* pp = a.prbp + (size_t)t * N * N;
        const int X0 = q.sx - ge.pad, Y0 = q.sy - ge.pad;
        const int Xa = X0 & ~1, dx = X0 - Xa;
        const float w00 = (1.0f - q.fx) * (1.0f - q.fy), w01 = q.fx * (1.0f - q.fy);
        const float w10 = (1.0f - q.fx) * q.fy, w11 = q.fx * q.fy;

        const int zoff = 0;   // (twiddle loads depend on the opaque thread id already)

        // ---- staging of the object rows of chunk c: rows Y0 + 64 band + 8 c + r, r in [0, 9) ----------
        // in two parts (u < 3, u >= 3) so that only 3 of the 5 16-byte values are in flight at a time
        f4 st[3];
        auto stage_load = [&](int c, int part) {
#pragma unroll
            for (int uu = 0; uu < 3; ++uu) {
                const int u = 3 * part + uu;
                if (u >= NLD) continue;
                const int e = tid + 1024 * u;
                const int row = e / SQ, c4 = e - row * SQ;
                const int band = row / 9, r = row - band * 9;
                const int Y = Y0 + 64 * band + 8 * c + r, X = Xa + 2 * c4;
                f4 val = f4{0.0f, 0.0f, 0.0f, 0.0f};
                if (e < SR * SQ && Y >= 0 && Y < ge.nz) {
                    const c32* src = ft + (size_t)Y * ge.n + X;
                    if (X >= 0 && X + 1 < ge.n) {
                        val = *reinterpret_cast<const f4*>(src);
                    } else {
                        if (X >= 0 && X < ge.n) { const c32 lo = src[0]; val.x = lo.x; val.y = lo.y; }
                        if (X + 1 >= 0 && X + 1 < ge.n) { const c32 hi = src[1]; val.z = hi.x; val.w = hi.y; }
                    }
                }
                st[uu] = val;
            }
        };
        auto stage_store = [&](int buf, int part) {
#pragma unroll
            for (int uu = 0; uu < 3; ++uu) {
                const int u = 3 * part + uu;
                if (u >= NLD) continue;
                const int e = tid + 1024 * u;
                const int row = e / SQ, c4 = e - row * SQ;
                if (e < SR * SQ) *reinterpret_cast<f4*>(&lds[buf * SBUF + row * SW + 2 * c4]) = st[uu];
            }
        };
        // padded probe values of slot (c, b): n2 = 8 c + j0 + 4 b, bands n1 = 0..3
        auto probe_load = [&](c32* pr, int c, int b) {
#pragma unroll
            for (int n1 = 0; n1 < 4; ++n1) pr[n1] = pp[(size_t)(64 * n1 + 8 * c + j0 + 4 * b) * N + x];
        };

        stage_load(0, 0);
        stage_store(0, 0);
        stage_load(0, 1);
        stage_store(0, 1);
        c32 prh[4];
        probe_load(prh, 0, 0);
        __syncthreads();

        c32 va[16], vb[TILES == 2 ? 16 : 1];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const c32* sb = lds + (c & 1) * SBUF + dx + x;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                c32 prn[4] = {zero, zero, zero, zero};
                if (c < 7 || b == 0) probe_load(prn, b == 0 ? c : c + 1, b == 0 ? 1 : 0);
                if (c < 7) stage_load(c + 1, b);
                c32 qv[4];
#pragma unroll
                for (int n1 = 0; n1 < 4; ++n1) {
                    const c32* r0 = sb + (n1 * 9 + j0 + 4 * b) * SW;
                    const c32 patch = r0[0] * w00 + r0[1] * w01 + r0[SW] * w10 + r0[SW + 1] * w11;   // kernels.cu:97-104
                    qv[n1] = cmul(prh[n1], patch);
                }
                const int n2 = 8 * c + j0 + 4 * b;
                if (TILES == 1) {
                    c32 u;
                    if (cls == 0) u = (qv[0] + qv[2]) + (qv[1] + qv[3]);
                    else if (cls == 2) u = cmul((qv[0] + qv[2]) - (qv[1] + qv[3]), wtab[(2 * n2) & (N - 1)]);
                    else if (cls == 1) u = cmul((qv[0] - qv[2]) + mul_mi(qv[1] - qv[3]), wtab[n2]);
                    else u = cmul((qv[0] - qv[2]) + mul_pi(qv[1] - qv[3]), wtab[(3 * n2) & (N - 1)]);
                    va[b * 8 + c] = u;
                } else {
                    if (cls == 0) {   // classes 0 and 2
                        const c32 s = qv[0] + qv[2], d = qv[1] + qv[3];
                        va[b * 8 + c] = s + d;
                        vb[b * 8 + c] = cmul(s - d, wtab[(2 * n2) & (N - 1)]);
                    } else {          // classes 1 and 3
                        const c32 s = qv[0] - qv[2], d = qv[1] - qv[3];
                        va[b * 8 + c] = cmul(s + mul_mi(d), wtab[n2]);
                        vb[b * 8 + c] = cmul(s + mul_pi(d), wtab[(3 * n2) & (N - 1)]);
                    }
                }
                if (c < 7) stage_store((c + 1) & 1, b);
#pragma unroll
                for (int n1 = 0; n1 < 4; ++n1) prh[n1] = prn[n1];
            }
            __syncthreads();
        }

        // ---- 64-point DFT over n2 of column x (wave local) ---------------------------------------------
        auto col_fft = [&](c32* v) {
            FC fc;
            // twiddles of the 64-point plan from the 256-entry table: exp(-2 pi i k / 64) = table[4 k]
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int t = 0; t < 8; ++t) fc.tw[b * 8 + t] = wtab[((4 * ((j0 + 4 * b) * t)) & (N - 1)) + zoff];
            fc.template compute<0>(v);
            fc.template store<0>(v, j0, [&](int i, c32 val) { lds[L::at(i, x)] = val; });
            wave_lds_fence();
            fc.template load<1>(v, j0, [&](int i) { return lds[L::at(i, x)]; });
            fc.template compute<1>(v);
            wave_lds_fence();
        };
        auto tile_store = [&](const c32* v) {   // rows k2 = j0 + 4 b + 8 t' of column x, row-pass layout
            FC fc;
            fc.template store<1>(v, j0, [&](int i, c32 val) { lds[L::at(i, x)] = val; });
        };
        // ---- 256-point DFT over x of tile row fr -> row kcls + 4 fr of g (wave local) --------------------
        auto row_fft = [&](int kcls) {
            FR frw;
            frw.init(j0r, wtab + zoff);
            c32 v[16];
            frw.template load<0>(v, j0r, [&](int i) { return lds[L::at(fr, i)]; });
            frw.template compute<0>(v);
            wave_lds_fence();
            frw.template store<0>(v, j0r, [&](int i, c32 val) { lds[L::at(fr, i)] = val; });
            wave_lds_fence();
            frw.template load<1>(v, j0r, [&](int i) { return lds[L::at(fr, i)]; });
            __syncthreads();   // the tile is dead from here on: the next tile / the next item may overwrite it
            frw.template compute<1>(v);
            c32* drow = gt + (size_t)(kcls + 4 * fr) * N;
            frw.template store<1>(v, j0r, [&](int i, c32 val) { __builtin_nontemporal_store(val, drow + i); });
        };

        col_fft(va);
        if (TILES == 2) col_fft(vb);
        tile_store(va);
        __syncthreads();
        row_fft(cls);
        if (TILES == 2) {
            tile_store(vb);
            __syncthreads();
            row_fft(cls + 2);
        }
    }
}
