// ptycho_common.hpp -- geometry, kernel argument structs, position decoding, run metadata
// Part of libptychohip (see ptycho_kernels.hip); included inside its anonymous namespace.
#pragma once


struct Geom {
    int ptheta, nz, n, nscan, ndet, nprb, pad;
};

enum Mode { M_PLAIN = 0, M_FWD = 1, M_ADJ_OBJ = 2, M_ADJ_PRB = 3 };

template <int N>
struct ColCfg {
    static constexpr int T = Plan<N>::T;
    static constexpr int C0 = (256 / T) > 16 ? (256 / T) : 16;
    static constexpr int C1 = C0 > N ? N : C0;
    static constexpr int C2 = T * C1 > 1024 ? 1024 / T : C1;
    static constexpr int divisor_below(int n, int c) { return n % c == 0 ? c : divisor_below(n, c - 1); }
    // detector columns per strip (<= 1024 threads; strips tile the N columns).  Plans with an odd factor (four threads of 12-28
    // points per column): 80, 96, 112 take 16-column strips, i.e. ONE wave per workgroup and four workgroups per CU -- the widest
    // strips that divide N (56 at 112: 224 threads, 110 KiB of LDS, one workgroup per CU) gave 212 / 272 / 234 CG iterations per
    // second at 112 / 96 / 80 against 278 / 340 / 308 (profiles/r04/mixed_radix.txt); 48 keeps its single 48-column strip (1032 against 465)
    // (one-wave workgroups for the small powers of two as well -- 16 columns at 64, 8 at 128 -- measured no different at 64 and
    // 40 % slower at 128 (64-byte row pieces): profiles/r04/narrow_strips.txt)
    static constexpr int C = is_pow2(N) ? C2 : divisor_below(N, N >= 80 ? 16 : C2);
    static constexpr int NT = T * C;            // threads per workgroup
};

// ---- deterministic adjoints: fixed-point accumulation (ColArgs::det_acc) ----------------------------
// scale = 2^p with  2^p * bound < 2^(62 - head),  bound = max|g| * max|prb or psi| * ndet >= any single window sum / probe
// sum contribution's magnitude (|near| <= ndet^2 max|g|, times c = 1/ndet, times |prb|); head = bits of headroom for the
// additions per element.  words: float bits of max|g| and max|other| (k_cg_absmax).  A pure function of the two words:
// the column kernel that accumulates and the kernel that folds the image into the output both evaluate it.
struct DetScale {
    const double* word_g;
    const double* word_o;
    int ndet;
    int head;
};
__device__ __forceinline__ float absmax_of(const double* word) { return __uint_as_float(*reinterpret_cast<const unsigned*>(word)); }
__device__ __forceinline__ float det_scale_of(const DetScale& d) {
    const float bound = absmax_of(d.word_g) * absmax_of(d.word_o) * (float)d.ndet;
    int e = 0;
    frexpf(bound > 0.0f ? bound : 1.0f, &e);              // bound < 2^e
    int p = 62 - d.head - e;
    p = p > 120 ? 120 : (p < -120 ? -120 : p);
    return ldexpf(1.0f, p);
}

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
    // forward pass of several probe modes per launch (k_cols_gatherwin<..., NM > 1>): probes and farplanes
    const c32* auxm[4];
    c32* dstm[4];
    // deterministic adjoints (option "deterministic"): the per-workgroup sums are added to a 64-bit fixed-point
    // image with INTEGER atomics (associative, so the result does not depend on the arrival order) instead of
    // float atomics on dst; det_scale_of(det) is the power of two that converts a float to that fixed point
    long long* det_acc;
    DetScale det;
    // forward passes of a line search whose decisions live on the device: the launch returns at once if *skip != 0
    // (the search is already resolved; the host enqueues the worst case and never reads the state back)
    const double* skip;
#ifdef PTY_STAMPS
    unsigned long long* stamps;   // diagnostic build only (tools/stamps.py): per-phase cycle totals, 12 words per kernel role
#endif
};

// In-kernel phase stamps (cdna_hip_programming.md section 7): compiled in only with -DPTY_STAMPS, a diagnostic build
// that is never shipped; every wave adds the shader cycles it spent in each phase to ColArgs::stamps.
#ifdef PTY_STAMPS
#define STAMP_DECL unsigned long long st_t_ = __builtin_amdgcn_s_memtime(), st_acc_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); st_acc_[i] += now_ - st_t_; st_t_ = now_; }
#define STAMP_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define STAMP_FLUSH(ptr) if ((ptr) && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 12; ++i_) atomicAdd((ptr) + i_, st_acc_[i_]); }
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_DRAIN()
#define STAMP_FLUSH(ptr)
#endif

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

// LDS accesses of one wave are served in order; this only stops the compiler from moving them
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// Synchronisation between the steps of ONE row transform in the row kernels: the T threads of a row exchange through
// their own LDS row only, and for T <= 64 (N <= 1024) they are lanes of one wave -- no workgroup barrier is needed,
// the waves of a workgroup run their rows independently.
template <int T>
__device__ __forceinline__ void row_sync() {
    if constexpr (T <= 64 && 64 % T == 0) wave_lds_fence();
    else __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// Order-free reductions across workgroups.  The CG scalars (a, b, costs, Dai-Yuan sums: ptycho.py:342-343,274-276,
// 370-371) decide float32 accept / reject comparisons, so their value must not depend on the order in which
// workgroups finish (float64 atomicAdd does).  Every workgroup stores its partial values in its own row of a scratch
// table and takes a ticket; the workgroup that arrives last adds the rows in index order (a fixed tree) and gets the
// totals.  Hand-off = write-through (sc1) 8-byte stores, drained, then one agent-scope ticket add per workgroup; the
// last arriver reads the rows with sc1 loads (MI355X_MICROARCH.md, "Valid forms", row 1 of the hand-off table).
// ---------------------------------------------------------------------------------------------------------------
// One ticket and one table per handle: the kernels that fold must be stream-serialised (include/ptycho_hip.h, "Calls on
// ONE handle"), which every caller in this repository is.  The ticket counts modulo the grid size (atomicInc wraps it to
// zero at gridDim.x - 1), so it needs no reset store that a later launch could miss.
constexpr int kFoldStride = 128;   // values per workgroup row (>= 7 groups x 17 line-search costs)
struct FoldBuf {
    unsigned long long* part;   // [rows][kFoldStride] doubles as bits; rows >= gridDim.x of every kernel that folds
    unsigned* ticket;           // zero between kernels
};

// Called by ALL threads of EVERY workgroup of the grid (256 threads), once per kernel, with the workgroup's nv <=
// kFoldStride partial values in LDS vals[].  Returns true in the last workgroup to arrive, with vals[0 .. nv) replaced
// by the totals (sum; the last nmax values: maximum) and visible to all its threads.  scratch: 256 doubles of LDS.
__device__ __forceinline__ bool fold_across_workgroups(const FoldBuf fb, double* vals, const int nv, const int nmax, double* scratch) {
    __shared__ int fold_last;
    const int tid = threadIdx.x;
    unsigned long long* mine = fb.part + (size_t)blockIdx.x * kFoldStride;
    __syncthreads();
    for (int i = tid; i < nv; i += 256)
        __hip_atomic_store(mine + i, (unsigned long long)__double_as_longlong(vals[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) fold_last = atomicInc(fb.ticket, gridDim.x - 1) == gridDim.x - 1 ? 1 : 0;   // wraps to 0 with the last arrival
    __syncthreads();
    if (!fold_last) return false;
    int nvp = 1;
    while (nvp < nv) nvp <<= 1;
    const int G = 256 / nvp;            // threads per value
    const int v = tid % nvp, g = tid / nvp;
    const bool is_max = v >= nv - nmax;
    double acc = 0.0;
    if (v < nv && g < G) {
        for (unsigned b = (unsigned)g; b < gridDim.x; b += (unsigned)G) {
            const double x = __longlong_as_double((long long)__hip_atomic_load(fb.part + (size_t)b * kFoldStride + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            acc = is_max ? fmax(acc, x) : acc + x;
        }
    }
    scratch[tid] = acc;
    __syncthreads();
    if (tid < nv) {
        double t = scratch[tid];
        for (int gg = 1; gg < G; ++gg) t = is_max ? fmax(t, scratch[gg * nvp + tid]) : t + scratch[gg * nvp + tid];
        vals[tid] = t;
    }
    __syncthreads();
    return true;
}
