// k_cg_small.hpp -- the object- / probe-sized stages of the CG loop and its line-search decisions, on the device
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
//
// src/libtike/cufft/ptycho.py runs these as dozens of small CuPy kernels per iteration and decides every
// line-search trial on the host (one device synchronisation per trial, :274-276).  Here the scalars live in a
// float64 state vector on the device (layout: PTYCHO_ST_* in include/ptycho_hip.h) and the decisions are
// taken by one-thread kernels, so an iteration is a fixed sequence of launches with no host round trip:
//   k_cg_scale_probe   probe *= a / b                                              (ptycho.py:344)
//   k_cg_absmax        max |x|  (np.max(np.abs(.)))                                (ptycho.py:356,431)
//   k_cg_dy_reduce     grad <- grad / max^2 [/ nscan * nmodes]; ||grad||^2, sum conj(d)(grad - grad0)
//   k_cg_dy_update     Dai-Yuan direction with the reference's complex beta         (ptycho.py:366-372,437-447)
//   k_cg_ls_prepare / k_cg_ls_decide   line_search_sqr's accept / shrink loop       (ptycho.py:253-281)
//   k_cg_axpy          x += gamma d                                                (ptycho.py:405,465)
//   k_cg_add_shifts    scan[0] += shifts                                           (ptycho.py:403)
#pragma once

static_assert(kLsGroupsRows == 7, "k_rows.hpp sizes its line-search accumulators for 7 groups");
constexpr int kLsGroupsMax = 7;   // a line-search pass prices up to 7 groups of 16 step lengths

__global__ void k_cg_scale_probe(c32* __restrict__ prb, const long long n, const double* __restrict__ st) {
    const float s = (float)st[PTYCHO_ST_A] / (float)st[PTYCHO_ST_B];   // float32, as ptycho.py:344 computes it
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) prb[i] = prb[i] * s;
}

// max |x| as float bits in the low word of *word (non-negative floats order like unsigned integers)
__global__ __launch_bounds__(256) void k_cg_absmax(const c32* __restrict__ x, const long long n, double* __restrict__ word) {
    __shared__ float red[4];
    float m = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const c32 v = x[i];
        m = fmaxf(m, hypotf(v.x, v.y));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        atomicMax(reinterpret_cast<unsigned*>(word), __float_as_uint(m));
    }
}
__device__ __forceinline__ float absmax_of(const double* word) { return __uint_as_float(*reinterpret_cast<const unsigned*>(word)); }

// g <- ((g / max^2) / div2) * mul3 (div2, mul3 <= 0: skipped), float32 per component like the reference's
// array expressions; unless first: dy += { ||g||^2, Re, Im of sum conj(d) (g - g0) }
__global__ __launch_bounds__(256) void k_cg_dy_reduce(c32* __restrict__ g, const c32* __restrict__ d, const c32* __restrict__ g0,
                                                      const long long n, const double* __restrict__ maxword, const float div2,
                                                      const float mul3, double* __restrict__ dy, const int first) {
    __shared__ double red[4 * 3];
    const float m = absmax_of(maxword);
    const float m2 = m * m;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        c32 v = g[i];
        v = c32{v.x / m2, v.y / m2};
        if (div2 > 0.0f) v = c32{v.x / div2, v.y / div2};
        if (mul3 > 0.0f) v = v * mul3;
        g[i] = v;
        if (!first) {
            const c32 dd = d[i], e = v - g0[i];
            s0 += (double)v.x * v.x + (double)v.y * v.y;
            s1 += (double)dd.x * e.x + (double)dd.y * e.y;     // conj(d) * e
            s2 += (double)dd.x * e.y - (double)dd.y * e.x;
        }
    }
    if (first) return;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s0 += __shfl_down(s0, off, 64);
        s1 += __shfl_down(s1, off, 64);
        s2 += __shfl_down(s2, off, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[w * 3] = s0; red[w * 3 + 1] = s1; red[w * 3 + 2] = s2; }
    __syncthreads();
    if (threadIdx.x < 3) atomicAdd(dy + threadIdx.x, red[threadIdx.x] + red[3 + threadIdx.x] + red[6 + threadIdx.x] + red[9 + threadIdx.x]);
}

// d <- -g + (||g||^2 / sum conj(d)(g - g0)) d   (complex beta, no real part taken); g0 <- g
__global__ void k_cg_dy_update(c32* __restrict__ d, c32* __restrict__ g0, const c32* __restrict__ g, const long long n,
                               const double* __restrict__ dy, const int first) {
    c32 beta = c32{0.0f, 0.0f};
    if (!first) {
        const float n2 = (float)dy[0], sr = (float)dy[1], si = (float)dy[2];
        const float den = sr * sr + si * si;
        beta = c32{n2 * sr / den, -n2 * si / den};
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const c32 gi = g[i];
        d[i] = first ? -gi : -gi + cmul(beta, d[i]);
        g0[i] = gi;
    }
}

// x <- x + gamma d with gamma = (float)*gamma_word; product and sum rounded separately, like the array expression
__global__ void k_cg_axpy(c32* __restrict__ x, const c32* __restrict__ d, const long long n, const double* __restrict__ gamma_word) {
    const float gm = (float)*gamma_word;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const c32 xi = x[i], di = d[i];
        x[i] = c32{__fadd_rn(xi.x, __fmul_rn(gm, di.x)), __fadd_rn(xi.y, __fmul_rn(gm, di.y))};
    }
}

__global__ void k_cg_add_shifts(float* __restrict__ scan, const double* __restrict__ shifts, const int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) scan[i] += (float)shifts[i];
}

// ---- line search on the device --------------------------------------------------------------------
// A pass prices ngroups x ncand step lengths gamma0 2^-j in one sweep over the two work buffers
// (k_rows_fused<EP_LINESEARCH>); candidate j of group grp lands in costs[grp * 17 + j], f(p1) in
// costs[grp * 17 + ncand].  k_cg_ls_decide replays line_search_sqr over them: accept the first step
// whose float32 cost is not above f(p1), fail below 1e-32.  Passes issued after the search is resolved
// return at once, so the host can enqueue the worst case without reading anything back.
__device__ __forceinline__ int ls_first_ncand(const double hint) {
    int nc = (int)hint + 2;
    nc = nc < 2 ? 2 : nc;
    nc = (nc + 3) & ~3;           // the kernel prices step lengths four at a time: the round-up is free
    return nc > kMaxCand ? kMaxCand : nc;
}

__global__ void k_cg_ls_prepare(double* __restrict__ st, const int which) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    st[PTYCHO_ST_LS_GAMMA0] = 1.0;
    // the last search of this kind ended beyond the first 16 step lengths: the next one almost always does too, so
    // the first pass prices whole groups of 16 up to that index (same groups, same sums as the later passes would form)
    const int want = (int)st[PTYCHO_ST_HINT + which] + 2;
    int ng = want > kMaxCand ? (want + kMaxCand - 1) / kMaxCand : 1;
    ng = ng > 4 ? 4 : ng;
    st[PTYCHO_ST_LS_NCAND] = ng > 1 ? (double)kMaxCand : (double)ls_first_ncand(st[PTYCHO_ST_HINT + which]);
    st[PTYCHO_ST_LS_NGROUPS] = (double)ng;
    st[PTYCHO_ST_LS_TRIED] = 0.0;
    st[PTYCHO_ST_LS_RESOLVED] = 0.0;
    for (int i = 0; i < kLsGroupsMax * (kMaxCand + 1); ++i) st[PTYCHO_ST_COSTS + i] = 0.0;
}

// which: hint slot; gamma_word: where 0.5 * step goes (ptycho.py:393,461); next_ngroups: size of the pass that
// follows if this one did not resolve the search (0: none follows)
__global__ void k_cg_ls_decide(double* __restrict__ st, const int which, const int gamma_word, const int next_ngroups) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
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
    for (int i = 0; i < kLsGroupsMax * (kMaxCand + 1); ++i) st[PTYCHO_ST_COSTS + i] = 0.0;
}

// ---- deterministic adjoints: fixed-point accumulation (ColArgs::det_acc) ----------------------------
// scale = 2^e with  2^e * bound < 2^52,  bound = max|g| * max|prb or psi| * ndet >= any single window sum / probe sum
// contribution's magnitude (|near| <= ndet^2 max|g|, times c = 1/ndet, times |prb|): ten more bits of the 63 are
// headroom for the sum over overlapping positions.  words: float bits of max|g| and max|other| (k_cg_absmax).
__global__ void k_det_scale(const double* __restrict__ word_g, const double* __restrict__ word_o, const int ndet, const long long nadd,
                            float* __restrict__ scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float bound = absmax_of(word_g) * absmax_of(word_o) * (float)ndet;
    int e = 0;
    frexpf(bound > 0.0f ? bound : 1.0f, &e);              // bound < 2^e
    int head = 1;
    while ((1ll << head) < nadd && head < 30) ++head;   // additions per element
    int p = 62 - head - e;
    p = p > 120 ? 120 : (p < -120 ? -120 : p);
    *scale = ldexpf(1.0f, p);
}
__global__ void k_det_finish(c32* __restrict__ dst, long long* __restrict__ acc, const long long n, const float* __restrict__ scale) {
    const double inv = 1.0 / (double)*scale;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double re = (double)acc[2 * i] * inv, im = (double)acc[2 * i + 1] * inv;
        acc[2 * i] = 0;
        acc[2 * i + 1] = 0;
        const c32 v = dst[i];
        dst[i] = c32{v.x + (float)re, v.y + (float)im};
    }
}
