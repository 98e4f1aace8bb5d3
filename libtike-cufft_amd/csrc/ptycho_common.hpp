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
    static constexpr int C = T * C1 > 1024 ? 1024 / T : C1;   // detector columns per strip (<= 1024 threads)
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
    // forward pass of several probe modes per launch (k_cols_gatherwin<..., NM > 1>): probes and farplanes
    const c32* auxm[4];
    c32* dstm[4];
    // deterministic adjoints (option "deterministic"): the per-workgroup sums are added to a 64-bit fixed-point
    // image with INTEGER atomics (associative, so the result does not depend on the arrival order) instead of
    // float atomics on dst; *det_scale is the power of two that converts a float to that fixed point
    long long* det_acc;
    const float* det_scale;
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
