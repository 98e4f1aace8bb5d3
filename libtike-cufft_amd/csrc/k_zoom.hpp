// k_zoom.hpp -- zoomed (matrix) DFT around the whole-pixel peak with a fused arg-max
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
#pragma once

// ---------------------------------------------------------------------------
// Sub-pixel stage of the position registration (ptycho.py:163-188, 217-235):
//   cross[i, j2, j1] = sum_p sum_k e^{+i th_p (j2 - offy_i)} e^{+i th_k (j1 - offx_i)} ip[i, p, k]
// on an ups x ups window (ups = 150 for upsample_factor = 100), followed by the arg-max of
// |cross| per pattern.  The reference contracts two [nscan, ups, ndet] complex128 kernels
// with einsum; here
//   * the per-pattern offsets leave the kernel as unit-modulus phases px[i,k], py[i,p];
//   * the remaining window kernel e^{i th_k jc} (jc centred) = cos + i sin is numerically
//     low rank over the reals: cos = Lc Vc, sin = Ls Vs with 8-9 + 7 terms at float64
//     accuracy.  The host passes V = [Vc; Vs] (RK = 16 rows) and L = [Lc, Ls];
//   * stage 1 (one thread per detector row p): t[p, r] = sum_k ip[p,k] px[k] V[r,k]
//     -- 2 float64 FMAs per term instead of the 4 of a complex kernel, 16 terms instead
//     of 150; V and px are wave-uniform and travel through scalar loads;
//   * stage 2: core[r2, r1] = i^{[r1>=nc]+[r2>=nc]} sum_p V[r2,p] py[p] t[p, r1]  (via LDS);
//   * stages 3-4 (one thread per window column j): M[r2] = sum_r1 L[j,r1] core[r2,r1];
//     cross[j2, j] = sum_r2 L[j2,r2] M[r2]; running maximum of |cross|^2.
// One workgroup per pattern; float64 throughout (the inputs are complex64).
// ---------------------------------------------------------------------------
constexpr int kZoomRK = 16;
typedef float zf4 __attribute__((ext_vector_type(4)));

template <int NT, int RKC>
__global__ __launch_bounds__(NT) void k_zoom_argmax(const c32* __restrict__ ip, const double2* __restrict__ px,
                                                    const double2* __restrict__ py, const double* __restrict__ vt,
                                                    const double* __restrict__ lz, const int N, const int nc,
                                                    const int ups, int* __restrict__ out,
                                                    const double* __restrict__ coarse, const double up,
                                                    double* __restrict__ shifts, float* __restrict__ scan_add) {
    constexpr int RK = kZoomRK;
    constexpr int NW = NT / 64;
    extern __shared__ double2 ybuf[];          // [N][RKC]
    __shared__ double2 core[RK * RK];
    __shared__ double redv[NW];
    __shared__ int redi[NW];

    const int tid = threadIdx.x;
    const size_t i = blockIdx.x;
    const c32* __restrict__ tile = ip + i * N * N;
    const double2* __restrict__ pxi = px + i * N;

    // ---- stage 1: t[p, r] for this thread's row p -----------------------------------------
    double ar[RK], ai[RK];
#pragma unroll
    for (int r = 0; r < RK; ++r) { ar[r] = 0.0; ai[r] = 0.0; }
    if (tid < N) {
        const zf4* __restrict__ row = reinterpret_cast<const zf4*>(tile + (size_t)tid * N);
        zf4 buf[8], nxt[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) buf[q] = __builtin_nontemporal_load(row + q);
        for (int k0 = 0; k0 < N; k0 += 16) {
            const bool more = k0 + 16 < N;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (more) nxt[q] = __builtin_nontemporal_load(row + (k0 + 16) / 2 + q);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int k = k0 + kk;
                const double2 ph = pxi[k];                                  // wave-uniform
                const double xr = (double)((kk & 1) ? buf[kk / 2].z : buf[kk / 2].x);
                const double xi = (double)((kk & 1) ? buf[kk / 2].w : buf[kk / 2].y);
                const double zr = xr * ph.x - xi * ph.y, zi = xr * ph.y + xi * ph.x;
                const double* __restrict__ vk = vt + (size_t)k * RK;        // wave-uniform
#pragma unroll
                for (int r = 0; r < RK; ++r) {
                    ar[r] = fma(zr, vk[r], ar[r]);
                    ai[r] = fma(zi, vk[r], ai[r]);
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) buf[q] = nxt[q];
        }
    }
    // ---- stage 2: core = V (py . t), reduced over p through LDS, RKC columns at a time -------
    const double2 pyp = tid < N ? py[i * N + tid] : double2{0.0, 0.0};
#pragma unroll
    for (int c0 = 0; c0 < RK; c0 += RKC) {
        __syncthreads();
        if (tid < N) {
#pragma unroll
            for (int r = 0; r < RKC; ++r)
                ybuf[tid * RKC + r] = double2{pyp.x * ar[c0 + r] - pyp.y * ai[c0 + r], pyp.x * ai[c0 + r] + pyp.y * ar[c0 + r]};
        }
        __syncthreads();
        for (int o = tid; o < RK * RKC; o += NT) {
            const int r2 = o / RKC, r1 = o % RKC;
            double sr = 0.0, si = 0.0;
            for (int p = 0; p < N; ++p) {
                const double w = vt[(size_t)p * RK + r2];
                const double2 y = ybuf[p * RKC + r1];
                sr = fma(w, y.x, sr);
                si = fma(w, y.y, si);
            }
            const int turns = ((c0 + r1) >= nc ? 1 : 0) + (r2 >= nc ? 1 : 0);   // times i^turns
            double2 c = double2{sr, si};
            if (turns == 1) c = double2{-si, sr};
            if (turns == 2) c = double2{-sr, -si};
            core[r2 * RK + c0 + r1] = c;
        }
    }
    __syncthreads();
    // ---- stages 3-4: window column j = tid --------------------------------------------------
    double best = -1.0;
    int besti = 0x7fffffff;
    if (tid < ups) {
        const int j = tid;
        double mr[RK], mi[RK];
        {
            double lj[RK];
#pragma unroll
            for (int r = 0; r < RK; ++r) lj[r] = lz[(size_t)j * RK + r];
#pragma unroll
            for (int r2 = 0; r2 < RK; ++r2) {
                double sr = 0.0, si = 0.0;
#pragma unroll
                for (int r1 = 0; r1 < RK; ++r1) {
                    const double2 c = core[r2 * RK + r1];                    // LDS broadcast
                    sr = fma(lj[r1], c.x, sr);
                    si = fma(lj[r1], c.y, si);
                }
                mr[r2] = sr; mi[r2] = si;
            }
        }
        for (int j2 = 0; j2 < ups; ++j2) {
            const double* __restrict__ l2 = lz + (size_t)j2 * RK;           // wave-uniform
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int r2 = 0; r2 < RK; ++r2) {
                sr = fma(l2[r2], mr[r2], sr);
                si = fma(l2[r2], mi[r2], si);
            }
            const double mag = sr * sr + si * si;
            if (mag > best) { best = mag; besti = j2 * ups + j; }
        }
    }
    // ---- first maximum over the window (ties: lowest flat index, as argmax does) -------------
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(besti, off, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    const int lane = tid & 63, wave = tid >> 6;
    if (lane == 0) { redv[wave] = best; redi[wave] = besti; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < NW; ++w)
            if (redv[w] > best || (redv[w] == best && redi[w] < besti)) { best = redv[w]; besti = redi[w]; }
        if (out) out[i] = besti;
        if (shifts) {   // ptycho.py:233-235: shifts + (argmax - dftshift) / upsample_factor
            const double dftshift = (double)(ups / 2);
            const double sy = coarse[2 * i] + ((double)(besti / ups) - dftshift) / up;
            const double sx = coarse[2 * i + 1] + ((double)(besti % ups) - dftshift) / up;
            shifts[2 * i] = sy;
            shifts[2 * i + 1] = sx;
            if (scan_add) {   // scan[0, :] += shifts (ptycho.py:403), float32 like the reference's in-place add
                scan_add[2 * i] += (float)sy;
                scan_add[2 * i + 1] += (float)sx;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Same computation with stages 1-2 on the float64 matrix cores
// (v_mfma_f64_16x16x4_f64: A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[row = (lane >> 4) + 4 reg][col = lane & 15]).  The scalar-operand version above waits
// on a scalar load for every k; here V is the B operand (one coalesced 512-B load per four
// k, reused by the eight MFMAs of a wave's four row tiles) and the rows stream as the A
// operand.  A wave owns 64 detector rows (four 16-row tiles); lane (li, g) reads the 32
// bytes [16 S + 4 g, +4) of its row per super-step S, so the order of the k summation is
// permuted (k-slot g of sub-step s is k = 16 S + 4 g + s) -- identically for A and B.
// Stage 2 takes the accumulators as they are: register v of tile t holds rows
// p = 64 w + 16 t + g + 4 v in k-slot g and column r1 on the lane, which is a valid B
// operand for core[r2, r1] = sum_p V[r2, p] y[p, r1] with A = V[li][p].
// Needs ndet % 64 == 0.
// ---------------------------------------------------------------------------
typedef double zd4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------
// Whole-pixel peak -> offsets and phases (ptycho.py:209-224) for the zoom kernels.
// best[i] is k_cols_argmax's packed result (low word = 0xffffffff - flat index).  The peak
// row / column is wrapped to [-N/2, N/2) as register_translation does (index > N/2 -> - N),
// coarse[i] = (sy, sx), and the per-pattern phases of the window kernel are
//   p[i, k] = exp(+i th_k (c0 - off)),  th_k = 2 pi k' / (N up),  off = fix(ups/2) - s up,
// c0 = (ups - 1) / 2; with an integer shift s the product k' s is reduced modulo N exactly.
// ---------------------------------------------------------------------------
__global__ void k_zoom_prepare(const unsigned long long* __restrict__ best, const int N, const int ups,
                               const double up, double2* __restrict__ px, double2* __restrict__ py,
                               double* __restrict__ coarse) {
    const size_t i = blockIdx.x;
    const unsigned idx = 0xffffffffu - (unsigned)(best[i] & 0xffffffffull);
    int sy = (int)(idx / (unsigned)N), sx = (int)(idx % (unsigned)N);
    if (sy > N / 2) sy -= N;
    if (sx > N / 2) sx -= N;
    if (threadIdx.x == 0) { coarse[2 * i] = (double)sy; coarse[2 * i + 1] = (double)sx; }
    const double delta = 0.5 * (double)(ups - 1) - (double)(ups / 2);
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
        const int kp = k < (N + 1) / 2 ? k : k - N;                       // fftfreq order
        const double fine = delta * (double)kp / ((double)N * up);
        const long long my = ((long long)kp * sy) % N, mx = ((long long)kp * sx) % N;
        double sn, cs;
        sincos(2.0 * M_PI * ((double)my / (double)N + fine), &sn, &cs);
        py[i * N + k] = double2{cs, sn};
        sincos(2.0 * M_PI * ((double)mx / (double)N + fine), &sn, &cs);
        px[i * N + k] = double2{cs, sn};
    }
}


template <int NT>
__global__ __launch_bounds__(NT, (NT == 256 ? 3 : 1)) void k_zoom_mfma(const c32* __restrict__ ip, const double2* __restrict__ px,
                                                  const double2* __restrict__ py, const double* __restrict__ vt,
                                                  const double* __restrict__ lz, const int N, const int nc,
                                                  const int ups, int* __restrict__ out,
                                                  const double* __restrict__ coarse, const double up,
                                                  double* __restrict__ shifts, float* __restrict__ scan_add) {
    constexpr int RK = kZoomRK;
    constexpr int NW = NT / 64;
    __shared__ double2 cpart[NW * RK * RK];
    __shared__ double2 core[RK * RK];
    __shared__ double redv[NW];
    __shared__ int redi[NW];

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6, li = lane & 15, g = lane >> 4;
    const size_t i = blockIdx.x;
    const c32* __restrict__ tile = ip + i * N * N;
    const double2* __restrict__ pxi = px + i * N;
    const int nwa = N / 64;                     // waves with rows

    if (w < nwa) {
        // ---- stage 1 ------------------------------------------------------------------------
        zd4 accr[4], acci[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { accr[t] = zd4{0.0, 0.0, 0.0, 0.0}; acci[t] = zd4{0.0, 0.0, 0.0, 0.0}; }
        const zf4* __restrict__ rows[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) rows[t] = reinterpret_cast<const zf4*>(tile + (size_t)(w * 64 + 16 * t + li) * N + 4 * g);
        zf4 a[4][2], an[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) { a[t][0] = __builtin_nontemporal_load(rows[t]); a[t][1] = __builtin_nontemporal_load(rows[t] + 1); }
        const int nS = N / 16;
        for (int S = 0; S < nS; ++S) {
            const int kb = 16 * S + 4 * g;
            if (S + 1 < nS) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    an[t][0] = __builtin_nontemporal_load(rows[t] + 8 * (S + 1));
                    an[t][1] = __builtin_nontemporal_load(rows[t] + 8 * (S + 1) + 1);
                }
            }
            double2 ph[4];
            double bv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) { ph[s] = pxi[kb + s]; bv[s] = vt[(size_t)(kb + s) * RK + li]; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double xr = (double)((s & 1) ? a[t][s / 2].z : a[t][s / 2].x);
                    const double xi = (double)((s & 1) ? a[t][s / 2].w : a[t][s / 2].y);
                    const double zr = xr * ph[s].x - xi * ph[s].y, zi = xr * ph[s].y + xi * ph[s].x;
                    accr[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(zr, bv[s], accr[t], 0, 0, 0);
                    acci[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(zi, bv[s], acci[t], 0, 0, 0);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) { a[t][0] = an[t][0]; a[t][1] = an[t][1]; }
        }
        // ---- stage 2: this wave's share of core = V (py . t) -----------------------------------
        zd4 cr = zd4{0.0, 0.0, 0.0, 0.0}, ci = zd4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int p = w * 64 + 16 * t + g + 4 * v;
                const double2 q = py[i * N + p];
                const double yr = q.x * accr[t][v] - q.y * acci[t][v], yi = q.x * acci[t][v] + q.y * accr[t][v];
                const double av = vt[(size_t)p * RK + li];
                cr = __builtin_amdgcn_mfma_f64_16x16x4f64(av, yr, cr, 0, 0, 0);
                ci = __builtin_amdgcn_mfma_f64_16x16x4f64(av, yi, ci, 0, 0, 0);
            }
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) cpart[w * RK * RK + (g + 4 * v) * RK + li] = double2{cr[v], ci[v]};
    }
    __syncthreads();
    for (int o = tid; o < RK * RK; o += NT) {
        double sr = 0.0, si = 0.0;
        for (int ww = 0; ww < nwa; ++ww) { sr += cpart[ww * RK * RK + o].x; si += cpart[ww * RK * RK + o].y; }
        const int r2 = o / RK, r1 = o % RK;
        const int turns = (r1 >= nc ? 1 : 0) + (r2 >= nc ? 1 : 0);   // times i^turns
        double2 c = double2{sr, si};
        if (turns == 1) c = double2{-si, sr};
        if (turns == 2) c = double2{-sr, -si};
        core[o] = c;
    }
    __syncthreads();
    // ---- stages 3-4: window column j = tid --------------------------------------------------
    double best = -1.0;
    int besti = 0x7fffffff;
    if (tid < ups) {
        const int j = tid;
        double mr[RK], mi[RK];
        {
            double lj[RK];
#pragma unroll
            for (int r = 0; r < RK; ++r) lj[r] = lz[(size_t)j * RK + r];
#pragma unroll
            for (int r2 = 0; r2 < RK; ++r2) {
                double sr = 0.0, si = 0.0;
#pragma unroll
                for (int r1 = 0; r1 < RK; ++r1) {
                    const double2 c = core[r2 * RK + r1];                    // LDS broadcast
                    sr = fma(lj[r1], c.x, sr);
                    si = fma(lj[r1], c.y, si);
                }
                mr[r2] = sr; mi[r2] = si;
            }
        }
        for (int j2 = 0; j2 < ups; ++j2) {
            const double* __restrict__ l2 = lz + (size_t)j2 * RK;           // wave-uniform
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int r2 = 0; r2 < RK; ++r2) {
                sr = fma(l2[r2], mr[r2], sr);
                si = fma(l2[r2], mi[r2], si);
            }
            const double mag = sr * sr + si * si;
            if (mag > best) { best = mag; besti = j2 * ups + j; }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double ov = __shfl_down(best, off, 64);
        const int oi = __shfl_down(besti, off, 64);
        if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { redv[w] = best; redi[w] = besti; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int ww = 1; ww < NW; ++ww)
            if (redv[ww] > best || (redv[ww] == best && redi[ww] < besti)) { best = redv[ww]; besti = redi[ww]; }
        if (out) out[i] = besti;
        if (shifts) {   // ptycho.py:233-235: shifts + (argmax - dftshift) / upsample_factor
            const double dftshift = (double)(ups / 2);
            const double sy = coarse[2 * i] + ((double)(besti / ups) - dftshift) / up;
            const double sx = coarse[2 * i + 1] + ((double)(besti % ups) - dftshift) / up;
            shifts[2 * i] = sy;
            shifts[2 * i + 1] = sx;
            if (scan_add) {   // scan[0, :] += shifts (ptycho.py:403), float32 like the reference's in-place add
                scan_add[2 * i] += (float)sy;
                scan_add[2 * i + 1] += (float)sx;
            }
        }
    }
}
