// k_cg_small.hpp -- the object- / probe-sized stages of the CG loop and its line-search decisions, on the device
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
//
// src/libtike/cufft/ptycho.py runs these as dozens of small CuPy kernels per iteration and decides every
// line-search trial on the host (one device synchronisation per trial, :274-276).  Here the scalars live in a
// float64 state vector on the device (layout: PTYCHO_ST_* in include/ptycho_hip.h) and the decisions are
// taken by one-thread kernels, so an iteration is a fixed sequence of launches with no host round trip:
//   k_cg_absmax        [probe *= a / b (ptycho.py:344);] max |x|  (np.max(np.abs(.)), ptycho.py:356,431)
//   k_cg_dy_reduce     grad <- grad / max^2 [/ nscan * nmodes]; ||grad||^2, sum conj(d)(grad - grad0)
//   k_cg_dy_update     Dai-Yuan direction with the reference's complex beta (ptycho.py:366-372,437-447), line-search reset
//   k_cg_ls_decide     line_search_sqr's accept / shrink loop (ptycho.py:253-281) for multi-GPU callers; one GPU
//                      decides inside the line-search pass (k_rows_fused)
// Every sum over workgroups is formed in a fixed order (fold_across_workgroups): same inputs, same bits.
//   k_cg_axpy          x += gamma d                                                (ptycho.py:405,465)
//   (scan[0] += shifts, ptycho.py:403, is done by the zoom kernel that finds the shifts)
#pragma once

static_assert(kLsGroupsRows == 7, "k_rows.hpp sizes its line-search accumulators for 7 groups");
constexpr int kLsGroupsMax = 7;   // a line-search pass prices up to 7 groups of 16 step lengths

// max |x| over a grid-stride share of x, reduced over the workgroup: thread 0 returns the workgroup's maximum
__device__ __forceinline__ float block_absmax(float m, float* red4) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
    if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(red4[0], red4[1]), fmaxf(red4[2], red4[3]));
}
__device__ __forceinline__ void store_absmax(double* word, float m) {   // float bits in the low half of the word
    *reinterpret_cast<unsigned long long*>(word) = (unsigned long long)__float_as_uint(m);
}

// max |x| (np.max(np.abs(.)), ptycho.py:356,431) -> *word (and *word2 if given); scale_ab: x *= a / b first (ptycho.py:344,
// float32 like the reference) -- the probe rescale and the maximum the gradient normalisation needs, in one pass
__global__ __launch_bounds__(256) void k_cg_absmax(c32* __restrict__ x, const long long n, double* __restrict__ word,
                                                   double* __restrict__ word2, const double* __restrict__ scale_ab, const FoldBuf fold) {
    __shared__ float red[4];
    __shared__ double vals[1], fscr[256];
    float m = 0.0f;
    float s = 1.0f;
    if (scale_ab) s = (float)scale_ab[0] / (float)scale_ab[1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        c32 v = x[i];
        if (scale_ab) {
            v = v * s;
            x[i] = v;
        }
        m = fmaxf(m, hypotf(v.x, v.y));
    }
    m = block_absmax(m, red);
    if (threadIdx.x == 0) vals[0] = (double)m;
    if (!fold_across_workgroups(fold, vals, 1, 1, fscr)) return;
    if (threadIdx.x == 0) {
        store_absmax(word, (float)vals[0]);
        if (word2) store_absmax(word2, (float)vals[0]);
    }
}

// out <- [out +] acc / scale; acc <- 0 (add = 0: the output is written, not accumulated -- no zero fill needed)
__global__ void k_det_finish(c32* __restrict__ dst, long long* __restrict__ acc, const long long n, const DetScale ds, const int add) {
    const double inv = 1.0 / (double)det_scale_of(ds);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const double re = (double)acc[2 * i] * inv, im = (double)acc[2 * i + 1] * inv;
        acc[2 * i] = 0;
        acc[2 * i + 1] = 0;
        if (add) {
            const c32 v = dst[i];
            dst[i] = c32{v.x + (float)re, v.y + (float)im};
        } else {
            dst[i] = c32{(float)re, (float)im};
        }
    }
}

// g <- ((g / max^2) / div2) * mul3 (div2, mul3 <= 0: skipped), float32 per component like the reference's
// array expressions; unless first: dy <- { ||g||^2, Re, Im of sum conj(d) (g - g0) } (fixed-order sums).
// acc != nullptr: g is still in the fixed-point image of the deterministic adjoint (g = acc / scale, acc <- 0): the
// fold-in pass of its own (k_det_finish) and the zero fill of g are saved.
__global__ __launch_bounds__(256) void k_cg_dy_reduce(c32* __restrict__ g, const c32* __restrict__ d, const c32* __restrict__ g0,
                                                      const long long n, const double* __restrict__ maxword, const float div2,
                                                      const float mul3, double* __restrict__ dy, const int first,
                                                      long long* __restrict__ acc, const DetScale ds, const FoldBuf fold) {
    __shared__ double red[4 * 3];
    __shared__ double vals[3], fscr[256];
    const float m = absmax_of(maxword);
    const float m2 = m * m;
    const double inv = acc ? 1.0 / (double)det_scale_of(ds) : 0.0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        c32 v;
        if (acc) {
            v = c32{(float)((double)acc[2 * i] * inv), (float)((double)acc[2 * i + 1] * inv)};
            acc[2 * i] = 0;
            acc[2 * i + 1] = 0;
        } else {
            v = g[i];
        }
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
    if (threadIdx.x < 3) vals[threadIdx.x] = red[threadIdx.x] + red[3 + threadIdx.x] + red[6 + threadIdx.x] + red[9 + threadIdx.x];
    if (!fold_across_workgroups(fold, vals, 3, 0, fscr)) return;
    if (threadIdx.x < 3) dy[threadIdx.x] = vals[threadIdx.x];
}

__device__ __forceinline__ int ls_first_ncand(const double hint) {
    int nc = (int)hint + 2;
    nc = nc < 2 ? 2 : nc;
    nc = (nc + 3) & ~3;           // the kernel prices step lengths four at a time: the round-up is free
    return nc > kMaxCand ? kMaxCand : nc;
}
// start of a line search: first pass sized from the accepted index of the last search of this kind
__device__ inline void ls_prepare_dev(double* __restrict__ st, const int which) {
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
}

// d <- -g + (||g||^2 / sum conj(d)(g - g0)) d   (complex beta, no real part taken); g0 <- g; and -- the search that
// follows needs it before its first pass -- the line-search state of kind ls_which (< 0: none) is reset
__global__ void k_cg_dy_update(c32* __restrict__ d, c32* __restrict__ g0, const c32* __restrict__ g, const long long n,
                               const double* __restrict__ dy, const int first, double* __restrict__ st, const int ls_which) {
    if (ls_which >= 0 && blockIdx.x == 0 && threadIdx.x == 0) ls_prepare_dev(st, ls_which);
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

// line_search_sqr on the costs a multi-GPU caller has just all-reduced (ls_decide_dev, k_rows.hpp); a single GPU
// takes the decision inside the line-search pass itself
__global__ void k_cg_ls_decide(double* __restrict__ st, const int which, const int gamma_word, const int next_ngroups) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    ls_decide_dev(st, which, gamma_word, next_ngroups);
}
