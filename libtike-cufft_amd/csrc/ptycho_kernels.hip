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
// The adjoint's intermediate lives in a scratch of at most 4 GiB (one launch pair per chunk of
// positions; at 4096 x 256^2 that is a single pair).  For ndet = 256 one radix-16 step of the DFT over
// y moves into the row pass ("split"); k_fwd_fused.hpp holds the single-launch forward.
#include <hip/hip_runtime.h>

#include <cstring>

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

#include "ptycho_common.hpp"
#include "k_cols_plain.hpp"
#include "k_rows.hpp"
#include "k_cols_window.hpp"
#ifdef PTYCHO_EXPERIMENTS
#include "k_fwd_fused.hpp"   // single-launch forward: measured slower, experiments build only
#endif
#include "k_cg_small.hpp"
#include "k_generic.hpp"
#include "k_zoom.hpp"
#include "k_tile.hpp"

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
thread_local std::string g_err;

// Launch-geometry and code-path knobs read from the environment exist only in the experiments build
// (`make -C libtike-cufft_amd/csrc experiments`, -DPTYCHO_EXPERIMENTS); the shipped library uses the defaults.
#ifdef PTYCHO_EXPERIMENTS
int exp_env(const char* name, int dflt) {
    const char* e = std::getenv(name);
    return e ? std::atoi(e) : dflt;
}
#else
constexpr int exp_env(const char*, int dflt) { return dflt; }
#endif

// kernel ids for the in-library profiler (ptycho_profile_read)
enum { K_COLS_FWD = 0, K_ROWS_FWD = 1, K_ROWS_INV = 2, K_COLS_ADJ_OBJ = 3, K_COLS_ADJ_PRB = 4, K_COLS_PLAIN = 5, K_SORT = 6, K_ROWS_STATS = 7, K_ROWS_PROJECT = 8, K_ROWS_LINESEARCH = 9, K_CG_SCALARS = 10, K_FWD_FUSED = 11, K_CG_UPDATE = 12, K_ROWS_CROSS = 13, K_COLS_ARGMAX = 14, K_ZOOM = 15, K_TILE_FWD = 16, K_TILE_ADJ_PRB = 17, K_COUNT = 18 };

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
    c32* table = nullptr;     // exp(-2 pi i k / ndet)  (ndet not a power of two: k / bs_m)
    int bs_m = 0;             // 0: ndet is a power of two; else length of the Bluestein plan (k_generic.hpp)
    c32* bs_chirp = nullptr;  // exp(-i pi m^2 / ndet), m < ndet
    c32* bs_hfilt = nullptr;  // FFT_M of the circular conj-chirp, divided by M
    c32* scratch = nullptr;   // chunk * ndet^2 complex64
    long long chunk = 0;      // positions per launch pair
    // position sort (object / probe adjoint)
    int* order = nullptr;          // processing order: position order[k] is the k-th in (angle, column bucket, row) order
    int* sort_counts = nullptr;    // k_rank_positions: partial ranks [positions] + tickets [ceil(positions / 256)], self-clearing
    static constexpr int kSlots = 2 * kMaxModes;
    c32* work[kSlots] = {};   // CG work buffers (column-pass intermediates), all positions; 0/1 + per-mode pairs
    double* slot_maxw = nullptr;      // [kSlots] max |slot content| left by the PROJECT stage (deterministic option)
    bool slot_max_ok[kSlots] = {};    // ... and whether that word describes what the slot holds now
    void* zoom_phase = nullptr;           // registration: per-pattern phases + whole-pixel shifts
    c32* reg_ip = nullptr;                // native CG loop: image product of the registration [positions][ndet][ndet]
    unsigned long long* reg_best = nullptr;   // whole-pixel peaks [positions]
    double* reg_shifts = nullptr;         // sub-pixel shifts [positions][2]
    int use_window = 1;       // 0: direct-atomics object adjoint (k_cols<ADJ_OBJ>)
    int use_split = 1;        // ndet = 256: one radix-16 step of the DFT over y runs in the row pass
    int use_tile = 1;         // ndet <= 128: one-launch forward / probe adjoint, the tile stays in LDS (k_tile.hpp)
    int deterministic = 0;    // 1: adjoints accumulate in 64-bit fixed point (integer atomics): bitwise reproducible results
    long long* det_acc = nullptr;   // fixed-point image, 2 words per object (or probe) element, kept zero between calls
    double* det_words = nullptr;    // device: max |g|, max |probe or object| as float bits (k_cg_absmax)
    DetScale last_det{};            // scale of the adjoint whose sums sit in det_acc (k_det_finish / k_cg_dy_reduce fold them in)
    bool det_pending = false;       // native CG stages: the gradient is still in det_acc (option "defer_finish")
    int defer_finish = 0;           // 1: ptycho_cg_obj_grad / prb_grad leave the gradient in det_acc for ptycho_cg_*_dir
    int ls_fused_decide = 0;        // 1: line-search passes decide on their own totals (single GPU: nothing to all-reduce)
    bool max_prb_valid = false, max_psi_valid = false;   // state[MAX_PRB / MAX_PSI] were set by the *_grad stage of this step
    FoldBuf fold{};                 // fixed-order cross-workgroup sums (ptycho_common.hpp): n_cu * 8 rows + ticket
    int fold_rows = 0;
    int compact_modes = 0;    // multi-mode CG: 0 = slot pairs (2k, 2k+1); M = compact layout A(k) = k, one shared B = M
    int sort_chunks = 1;      // position order is chunk-major over this many equal position ranges (chunked line search)
    int use_fused = 0;        // ndet = 256 forward as one launch (k_fwd_fused256): 0 off (default: measured slower), 1 / 2 class tiles per pass
    c32* prbp = nullptr;      // fused forward: c * probe in a zero-bordered ndet x ndet frame, per angle
    int trust_order = 0;      // 1: caller vouches that scan is unchanged since the last sort
    int native_order = 0;     // 1: the native CG stages are running and track scan themselves (ptycho_cg_obj_finish re-sorts
                              // after it moved the positions); cleared by ptycho_fwd / ptycho_adj, whose callers own trust_order
    const float* order_scan = nullptr;   // scan pointer the current order was computed from
#ifdef PTY_STAMPS
    unsigned long long* stamps = nullptr;   // diagnostic build: 24 words (forward column pass, object adjoint column pass)
#endif
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
    if (exp_env("PTYCHO_HIP_CHUNK", 0) > 0) return exp_env("PTYCHO_HIP_CHUNK", 0);
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

// shortest run of sorted positions a windowed column workgroup takes (each run pays one window fill)
// 512 positions x 256^2 CG: 8 -> 1.44, 16 -> 1.36, 24 -> 1.51 ms per iteration.  Tiny problems (fewer than one workgroup per CU at
// runs of 16) take shorter runs, down to 4: a workgroup's positions are processed one after the other (~5 us each)
static int min_seglen(int np = 1 << 30, int nstrips = 1, int n_cu = 256) {
    static const int v = exp_env("PTYCHO_HIP_MINSEG", 16);
    int m = v < 1 ? 1 : v;
    const long long fill = (long long)np * nstrips / (n_cu > 0 ? n_cu : 1);   // run length that gives one workgroup per CU
    if (fill < m) m = fill < 4 ? 4 : (int)fill;
    return m;
}

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

template <int N, bool SPLIT = false, int CW = 0>
int launch_adjwin(ptycho_handle h, ColArgs a, hipStream_t st, int wg_target = 0) {
    constexpr int NTHREADS = Plan<N>::T * (CW ? CW : ColCfg<N>::C);
    const int np = a.k_end - a.k_begin;
    if (np <= 0 || a.nstrips <= 0) return PTYCHO_OK;
    // contiguous runs of the sorted order; about 4 workgroups per CU in total
    // ndet 256: two rounds of resident workgroups (one round of runs of 128: within 1 %, six or eight rounds: +4 %);
    // ndet 128: ONE round (two workgroups per CU, runs of 32 positions): 0.234 -> 0.197 ms at 4096 x 128^2; 64 and 32: no gain / worse
    if (wg_target <= 0) wg_target = h->n_cu * (N == 128 ? 2 : 4);   // (512: runs of 128 instead of 64 positions: 1.728 -> 1.709 ms, profiles/r04/stamps.txt)
    int nseg = (wg_target + a.nstrips - 1) / a.nstrips;
    if (nseg < 1) nseg = 1;
    int seglen = (np + nseg - 1) / nseg;
    if (seglen < min_seglen(np, a.nstrips, h->n_cu)) seglen = min_seglen(np, a.nstrips, h->n_cu);
    if (seglen > kRunMax) seglen = kRunMax;
    nseg = (np + seglen - 1) / seglen;
    static const int nt_mode_a = exp_env("PTYCHO_HIP_NT", 0);
    a.nt = nt_mode_a;
#ifdef PTY_STAMPS
    a.stamps = h->stamps;
#endif
    {
        ProfSpan ps(h, K_COLS_ADJ_OBJ, st);
        hipLaunchKernelGGL((k_cols_adjwin<N, SPLIT, CW>), dim3((unsigned)(a.nstrips * nseg)), dim3(NTHREADS), 0, st, a, seglen);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

template <int N, int MODE, bool SPLIT = false, int CW = 0>
int launch_gatherwin(ptycho_handle h, ColArgs a, hipStream_t st, int wg_target = 0) {
    constexpr int NTHREADS = Plan<N>::T * (CW ? CW : ColCfg<N>::C);
    const int np = a.k_end - a.k_begin;
    if (np <= 0 || a.nstrips <= 0) return PTYCHO_OK;
    // whole rounds of resident workgroups (two per CU un-split, three split): a ragged last round costs 5-20 % (round 3: 1.5
    // rounds of the un-split forward pass made the CG iteration 8.43 -> 8.87 ms; one long round 8.49)
    if (wg_target <= 0) wg_target = h->n_cu * (SPLIT ? 6 : 4);
    int nseg = (wg_target + a.nstrips - 1) / a.nstrips;
    if (nseg < 1) nseg = 1;
    int seglen = (np + nseg - 1) / nseg;
    if (seglen < min_seglen(np, a.nstrips, h->n_cu)) seglen = min_seglen(np, a.nstrips, h->n_cu);
    if (seglen > kRunMax) seglen = kRunMax;
    nseg = (np + seglen - 1) / seglen;
#ifdef PTYCHO_EXPERIMENTS
    {   // experiment knob: fewer, longer runs
        const int want = exp_env("PTYCHO_HIP_COLSEGS", 0);
        if (want > 0) {
            seglen = (np + want - 1) / want;
            if (seglen > kRunMax) seglen = kRunMax;
            nseg = (np + seglen - 1) / seglen;
        }
    }
#endif
    static const int nt_mode_g = exp_env("PTYCHO_HIP_NT", 0);
    a.nt = nt_mode_g;
#ifdef PTY_STAMPS
    a.stamps = h->stamps;
#endif
    {
        ProfSpan ps(h, MODE == M_FWD ? K_COLS_FWD : K_COLS_ADJ_PRB, st);
        hipLaunchKernelGGL((k_cols_gatherwin<N, MODE, SPLIT, 1, CW>), dim3((unsigned)(a.nstrips * nseg)), dim3(NTHREADS), 0, st, a, seglen);
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
    static const int nt_mode = exp_env("PTYCHO_HIP_NT", 3);
    a.nt = nt_mode;
    {
        ProfSpan ps(h, DIR < 0 ? K_ROWS_FWD : K_ROWS_INV, st);
        hipLaunchKernelGGL((k_rows<N, DIR>), dim3((unsigned)grid), dim3(256), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

template <int N, int DIR>
int launch_rows_split(ptycho_handle h, RowArgs a, hipStream_t st) {
    if (a.nrows <= 0) return PTYCHO_OK;
    const long long nitems = (a.nrows / N) * 16;
    // measured at 4096 x 256^2: forward best with ~32 workgroups per CU in the grid (0.73 ms vs 0.75
    // at 8), adjoint best with one item per workgroup (0.70 ms vs 0.78); PTYCHO_HIP_ROWGRID overrides
    static const int env_mult = exp_env("PTYCHO_HIP_ROWGRID", 0);
    const int mult = env_mult > 0 ? env_mult : (DIR < 0 ? 32 : 256);
    long long grid = nitems < (long long)h->n_cu * mult ? nitems : (long long)h->n_cu * mult;
    {
        ProfSpan ps(h, DIR < 0 ? K_ROWS_FWD : K_ROWS_INV, st);
        hipLaunchKernelGGL((k_rows_split<N, DIR>), dim3((unsigned)grid), dim3(256), 0, st, a);
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


#ifdef PTYCHO_EXPERIMENTS
int do_fwd_fused(ptycho_handle h, c32* g, const c32* f, const float* scan, const c32* prb, hipStream_t st) {
    const Geom& ge = h->ge;
    const int total = ge.ptheta * ge.nscan;
    const int N = ge.ndet;
    if (!h->prbp) HIP_TRY(hipMalloc((void**)&h->prbp, (size_t)ge.ptheta * N * N * sizeof(c32)));
    // positions in sorted order: the workgroups in flight then touch neighbouring object rows (L2 hits)
    int rc = sort_positions(h, scan, st);
    if (rc) return rc;
    const int npix = ge.ptheta * N * N;
    hipLaunchKernelGGL(k_pad_probe, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, prb, h->prbp, ge);
    FusedArgs fa{};
    fa.f = f; fa.g = g; fa.prbp = h->prbp; fa.scan = scan; fa.table = h->table; fa.order = h->order; fa.ge = ge; fa.total = total;
    const int tiles = h->use_fused >= 2 ? 2 : 1;
    const int nitems = total * (4 / tiles);
    const int grid = nitems < h->n_cu ? nitems : h->n_cu;
    {
        ProfSpan ps(h, K_FWD_FUSED, st);
        if (tiles == 2) hipLaunchKernelGGL((k_fwd_fused256<2>), dim3((unsigned)grid), dim3(1024), 0, st, fa);
        else hipLaunchKernelGGL((k_fwd_fused256<1>), dim3((unsigned)grid), dim3(1024), 0, st, fa);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}
#endif

// ---- one-launch operators for ndet <= 128 (k_tile.hpp): persistent workgroups, as many as fit a CU's LDS ----
template <int N>
unsigned tile_grid(ptycho_handle h, long long npos, int max_per_cu = 0) {
    using CF = TileCfg<N>;
    int per_cu = (int)((160 * 1024) / CF::lds_bytes);
    if (per_cu * CF::NT > 2048) per_cu = 2048 / CF::NT;
    if (max_per_cu > 0 && per_cu > max_per_cu) per_cu = max_per_cu;
    if (per_cu < 1) per_cu = 1;
    long long wg = (npos + CF::TPW - 1) / CF::TPW;
    const long long cap = (long long)h->n_cu * per_cu;
    return (unsigned)(wg < cap ? (wg < 1 ? 1 : wg) : cap);
}
template <int N>
int launch_fwd_tile(ptycho_handle h, c32* g, const c32* f, const float* scan, const c32* prb, hipStream_t st) {
    const Geom& ge = h->ge;
    TileArgs ta{};
    ta.obj = f; ta.prb = prb; ta.g = g; ta.scan = scan; ta.table = h->table; ta.ge = ge;
    ta.npos = (int)((long long)ge.ptheta * ge.nscan);
    {
        ProfSpan ps(h, K_TILE_FWD, st);
        hipLaunchKernelGGL((k_fwd_tile<N>), dim3(tile_grid<N>(h, ta.npos)), dim3(TileCfg<N>::NT), 0, st, ta);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}
template <int N>
int launch_adjprb_tile(ptycho_handle h, c32* prb_out, const c32* g, const float* scan, const c32* f, hipStream_t st) {
    const Geom& ge = h->ge;
    TileArgs ta{};
    ta.obj = f; ta.g = const_cast<c32*>(g); ta.out = prb_out; ta.scan = scan; ta.table = h->table; ta.ge = ge;
    ta.npos = (int)((long long)ge.ptheta * ge.nscan);
    {
        ProfSpan ps(h, K_TILE_ADJ_PRB, st);
                // every workgroup ends with one atomic pair per probe pixel, all on the same ndet^2 addresses: few, long-running
        // workgroups (four per CU: 1.90 ms at 16384 x 16^2, 0.27 at 32^2, 0.108 at 4096 x 64^2; one / two: 0.040, 0.093, 0.096)
        hipLaunchKernelGGL((k_adjprb_tile<N>), dim3(tile_grid<N>(h, ta.npos, N <= 16 ? 1 : 2)), dim3(TileCfg<N>::NT), 0, st, ta);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

template <int N>
int do_fwd(ptycho_handle h, c32* g, const c32* f, const float* scan, const c32* prb, hipStream_t st) {
    constexpr int C = ColCfg<N>::C;
    const Geom& ge = h->ge;
    const long long total = (long long)ge.ptheta * ge.nscan;
    int strip0, nstrips;
    strip_range<N>(ge, strip0, nstrips);
    const bool window = h->use_window && WinCfg<N>::fits;
    int rc = PTYCHO_OK;
    if constexpr (N <= 128) {
        // the tile fits one CU's LDS: one launch, no intermediate in HBM, no position sort (16-byte rows of g)
        if (h->use_tile && ((size_t)g % 16) == 0 && ge.n >= 2 && (long long)ge.nz * ge.n < (1ll << 28)) return launch_fwd_tile<N>(h, g, f, scan, prb, st);
    }
#ifdef PTYCHO_EXPERIMENTS
    if constexpr (N == 256) {
        // single launch, no intermediate in HBM; needs 16-byte aligned object rows
        if (h->use_fused && ge.n % 2 == 0 && ((size_t)f % 16) == 0) return do_fwd_fused(h, g, f, scan, prb, st);
    }
#endif
    if (window) {
        rc = sort_positions(h, scan, st);
        if (rc) return rc;
    }
    // The column pass writes straight into g and the row pass transforms g in place, so the
    // forward operator needs no scratch and is issued as one launch pair over all positions.
    ColArgs ca{};
    ca.src = f; ca.dst = g; ca.aux = prb; ca.scan = scan; ca.table = h->table; ca.ge = ge;
    ca.k_begin = 0; ca.k_end = (int)total; ca.strip0 = strip0; ca.nstrips = nstrips;
    RowArgs ra{};
    ra.src = g; ra.dst = g; ra.table = h->table; ra.tile_index = nullptr;
    ra.nrows = total * N; ra.xa = strip0 * C; ra.xb = (strip0 + nstrips) * C; ra.wa = 0; ra.wb = N;
    if constexpr (N == 256) {
        if (window && h->use_split) {
            // 32-column strips: the strip range and the row pass's column limits follow the wider strips
            constexpr int CF = 32;
            ca.order = h->order;
            ca.strip0 = ge.pad / CF;
            ca.nstrips = (ge.pad + ge.nprb - 1) / CF - ca.strip0 + 1;
            ra.xa = ca.strip0 * CF; ra.xb = (ca.strip0 + ca.nstrips) * CF;
            // exactly one resident round of workgroups (two per CU), runs of 64 positions at 4096: 0.42 -> 0.395 ms against
            // two rounds of shorter runs; 1.5 or 3 rounds (a ragged tail) cost 10-20 % (profiles/r03/knob_sweep.txt)
            rc = launch_gatherwin<N, M_FWD, true, CF>(h, ca, st, h->n_cu * 2);
            if (rc) return rc;
            return launch_rows_split<N, -1>(h, ra, st);
        }
    }
    if (window) {
        ca.order = h->order;
        if constexpr (WinCfg<N>::fits) rc = launch_gatherwin<N, M_FWD>(h, ca, st);
    } else {
        ca.order = nullptr;
        rc = launch_cols<N, -1, M_FWD>(h, ca, st);
    }
    if (rc) return rc;
    return launch_rows<N, -1>(h, ra, st);
}


// ---- deterministic adjoints (option "deterministic"): set-up before / fold-in after the column pass ----
inline unsigned fold_grid(ptycho_handle h, long long n, int per_cu) {   // grid of a kernel that folds: <= fold_rows workgroups
    long long g = (n + 255) / 256;
    long long cap = (long long)h->n_cu * per_cu;
    if (cap > h->fold_rows) cap = h->fold_rows;
    if (g > cap) g = cap;
    return (unsigned)(g < 1 ? 1 : g);
}
int det_begin(ptycho_handle h, ColArgs& ca, const c32* gsrc, long long gcount, const c32* other, long long ocount, int flg, hipStream_t st,
              const double* known_gmax = nullptr,   // max |gsrc| / max |other| are already on the device (k_cg_absmax format)
              const double* known_omax = nullptr) {
    const Geom& ge = h->ge;
    const size_t nobj = (size_t)ge.ptheta * ge.nz * ge.n, nprb = (size_t)ge.ptheta * ge.nprb * ge.nprb;
    // option "defer_finish": a gradient that ptycho_cg_obj_grad / prb_grad left in the fixed-point image has not been
    // folded in yet (ptycho_cg_obj_dir / prb_dir do that); another deterministic adjoint would add into the same image
    if (h->det_pending)
        return fail(PTYCHO_ERR_ARG, "a deferred gradient is pending in the fixed-point image: call ptycho_cg_obj_dir / ptycho_cg_prb_dir first");
    if (!h->det_acc) {
        const size_t words = 2 * (nobj > nprb ? nobj : nprb);
        HIP_TRY(hipMalloc((void**)&h->det_acc, words * sizeof(long long)));
        HIP_TRY(hipMemset(h->det_acc, 0, words * sizeof(long long)));
        HIP_TRY(hipMalloc((void**)&h->det_words, 2 * sizeof(double)));
        HIP_TRY(hipMemset(h->det_words, 0, 2 * sizeof(double)));
    }
    if (!known_gmax) hipLaunchKernelGGL(k_cg_absmax, dim3(fold_grid(h, gcount, 8)), dim3(256), 0, st, (c32*)gsrc, gcount, h->det_words,
                                        (double*)nullptr, (const double*)nullptr, h->fold);
    if (!known_omax) hipLaunchKernelGGL(k_cg_absmax, dim3(fold_grid(h, ocount, 4)), dim3(256), 0, st, (c32*)other, ocount, h->det_words + 1,
                                        (double*)nullptr, (const double*)nullptr, h->fold);
    HIP_TRY(hipGetLastError());
    // additions per element: every position of an angle may touch it, four bilinear taps (object) / once (probe)
    const long long nadd = flg == 0 ? 4ll * ge.nscan : (long long)ge.nscan;
    int head = 1;
    while ((1ll << head) < nadd && head < 30) ++head;
    h->last_det = DetScale{known_gmax ? known_gmax : (const double*)h->det_words,
                           known_omax ? known_omax : (const double*)(h->det_words + 1), ge.ndet, head};
    ca.det_acc = h->det_acc;
    ca.det = h->last_det;
    return PTYCHO_OK;
}
int det_end(ptycho_handle h, c32* dst, int flg, hipStream_t st, int add = 1) {
    const Geom& ge = h->ge;
    const long long n = flg == 0 ? (long long)ge.ptheta * ge.nz * ge.n : (long long)ge.ptheta * ge.nprb * ge.nprb;
    long long g = (n + 255) / 256;
    if (g > (long long)h->n_cu * 4) g = (long long)h->n_cu * 4;
    hipLaunchKernelGGL(k_det_finish, dim3((unsigned)g), dim3(256), 0, st, dst, h->det_acc, n, h->last_det, add);
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
    const bool window = flg == 0 && h->use_window && WinCfg<N>::fits;
    if constexpr (N <= 128) {
        // probe adjoint with the tile in LDS: one launch, g read once, no scratch, no position sort
        if (flg == 1 && h->use_tile && !h->deterministic && ((size_t)g % 16) == 0 && (long long)ge.nz * ge.n < (1ll << 28))
            return launch_adjprb_tile<N>(h, prb, g, scan, f, st);
    }
    // positions are visited in sorted order (angle, column bucket, row): neighbours in the
    // object are neighbours in time, which is what the LDS overlap-add window needs
    int rc = sort_positions(h, scan, st);
    if (rc) return rc;
    if (!h->scratch) {   // the row pass's output (up to 4 GiB), allocated by the first call that gets here
        rc = alloc_scratch(h);
        if (rc) return rc;
    }
    ColArgs det{};
    if (h->deterministic) {
        if (!(h->use_window && WinCfg<N>::fits)) return fail(PTYCHO_ERR_ARG, "option deterministic needs the windowed adjoint kernels (ndet <= 512)");
        rc = det_begin(h, det, g, total * N * N, flg == 0 ? prb : f, flg == 0 ? (long long)ge.ptheta * ge.nprb * ge.nprb : (long long)ge.ptheta * ge.nz * ge.n, flg, st);
        if (rc) return rc;
    }
    for (long long k0 = 0; k0 < total; k0 += h->chunk) {
        const long long k1 = k0 + h->chunk < total ? k0 + h->chunk : total;
        RowArgs ra{};
        ra.src = g; ra.dst = h->scratch; ra.table = h->table; ra.tile_index = h->order + k0;
        ra.nrows = (k1 - k0) * N; ra.xa = 0; ra.xb = N; ra.wa = strip0 * C; ra.wb = (strip0 + nstrips) * C;
        bool split = false, tiled = false;
        if constexpr (N == 256) split = h->use_split && h->use_window;
        if constexpr (N == 256) {
            if (split) rc = launch_rows_split<N, +1>(h, ra, st);
        }
        if constexpr (N <= 128) {
            // whole tiles through LDS, 16 bytes per lane (k_tile.hpp): 0.34 -> 0.047 ms at 16384 x 32^2
            tiled = h->use_tile && ((size_t)g % 16) == 0;
            if (tiled) {
                ProfSpan ps(h, K_ROWS_INV, st);
                if constexpr (!is_pow2(N) && N >= 80) {   // 16-row slabs, one wave per workgroup (k_rows_slab)
                    const long long items = (k1 - k0) * (N / 16);
                    const long long cap = (long long)h->n_cu * 10;
                    hipLaunchKernelGGL((k_rows_slab<N, +1>), dim3((unsigned)(items < cap ? items : cap)), dim3(16 * Plan<N>::T), 0, st, g, h->scratch,
                                       (const int*)(h->order + k0), (int)(k1 - k0), (const c32*)h->table);
                } else {
                    hipLaunchKernelGGL((k_rows_tile<N, +1>), dim3(tile_grid<N>(h, k1 - k0)), dim3(TileCfg<N>::NT), 0, st, g, h->scratch,
                                       (const int*)(h->order + k0), (int)(k1 - k0), (const c32*)h->table);
                }
                HIP_TRY(hipGetLastError());
            }
        }
        if (!split && !tiled) rc = launch_rows<N, +1>(h, ra, st);
        if (rc) return rc;
        ColArgs ca{};
        ca.src = h->scratch; ca.scan = scan; ca.table = h->table; ca.ge = ge;
        ca.order = h->order; ca.k_begin = (int)k0; ca.k_end = (int)k1; ca.strip0 = strip0; ca.nstrips = nstrips;
        ca.det_acc = det.det_acc; ca.det = det.det;
        if constexpr (N == 256) {
            if (split) {
                if (flg == 0) {
                    ca.dst = f; ca.aux = prb;
                    rc = launch_adjwin<N, true>(h, ca, st);
                } else {
                    ca.dst = prb; ca.aux = f;
                    rc = launch_gatherwin<N, M_ADJ_PRB, true>(h, ca, st);
                }
                if (rc) return rc;
                continue;
            }
        }
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
    if (h->deterministic) return det_end(h, flg == 0 ? f : prb, flg, st);
    return PTYCHO_OK;
}

template <int N>
int do_fft2(ptycho_handle h, c32* dst, const c32* src, long long nbatch, int dir, hipStream_t st) {
    constexpr int C = ColCfg<N>::C;
    if constexpr (N <= 128) {
        // the tile fits one CU's LDS: both passes in one launch (k_tile.hpp)
        if (h->use_tile && ((size_t)src % 16) == 0 && ((size_t)dst % 16) == 0 && nbatch < (1ll << 30)) {
            ProfSpan ps(h, K_COLS_PLAIN, st);
            if (dir < 0)
                hipLaunchKernelGGL((k_rows_tile<N, -1, true>), dim3(tile_grid<N>(h, nbatch)), dim3(TileCfg<N>::NT), 0, st, src, dst,
                                   (const int*)nullptr, (int)nbatch, (const c32*)h->table);
            else
                hipLaunchKernelGGL((k_rows_tile<N, +1, true>), dim3(tile_grid<N>(h, nbatch)), dim3(TileCfg<N>::NT), 0, st, src, dst,
                                   (const int*)nullptr, (int)nbatch, (const c32*)h->table);
            HIP_TRY(hipGetLastError());
            return PTYCHO_OK;
        }
    }
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
inline int slot_a(ptycho_handle h, int k) { return h->compact_modes ? k : 2 * k; }
inline int slot_b(ptycho_handle h, int k) { return h->compact_modes ? h->compact_modes : 2 * k + 1; }
inline bool slot_ready(ptycho_handle h, int slot) { return slot >= 0 && slot < ptycho_handle_s::kSlots && h->work[slot]; }
int ensure_work(ptycho_handle h, int slot) {   // called by every stage that is about to write the slot
    if (slot < 0 || slot >= ptycho_handle_s::kSlots) return fail(PTYCHO_ERR_ARG, "work slot out of range");
    h->slot_max_ok[slot] = false;
    if (!h->work[slot]) {
        // kMaxModes spare tiles: the M chunk parts of the shared slot of the compact layout take M ceil(total / M) tiles
        const size_t total = (size_t)h->ge.ptheta * h->ge.nscan + kMaxModes;
        HIP_TRY(hipMalloc((void**)&h->work[slot], total * h->ge.ndet * h->ge.ndet * sizeof(c32)));
        HIP_TRY(hipMemset(h->work[slot], 0, total * h->ge.ndet * h->ge.ndet * sizeof(c32)));
    }
    return PTYCHO_OK;
}

// deterministic option: the PROJECT stage leaves max |dst slot| on the device for the adjoint column pass that follows
int project_maxword(ptycho_handle h, int dst_slot, RowFusedArgs& a, hipStream_t st) {
    if (!h->deterministic) return PTYCHO_OK;
    // option "defer_finish": the pending gradient's fixed-point scale reads the max word of the slot its adjoint consumed; a
    // projection issued before ptycho_cg_obj_dir / prb_dir folded the gradient in would overwrite that word
    if (h->det_pending)
        return fail(PTYCHO_ERR_ARG, "a deferred gradient is pending in the fixed-point image: call ptycho_cg_obj_dir / ptycho_cg_prb_dir first");
    if (!h->slot_maxw) {
        HIP_TRY(hipMalloc((void**)&h->slot_maxw, ptycho_handle_s::kSlots * sizeof(double)));
        HIP_TRY(hipMemset(h->slot_maxw, 0, ptycho_handle_s::kSlots * sizeof(double)));
    }
    a.maxword = h->slot_maxw + dst_slot;   // stored (not accumulated) by the stage's last workgroup
    return PTYCHO_OK;
}

// npos_limit > 0: only the first npos_limit positions (natural order = sorted order for whole angles: the sort key is
// angle major) -- the position correction needs angle 0 only (ptycho.py:399-403)
template <int N>
int do_cg_fwd_cols(ptycho_handle h, int slot, const c32* f, const float* scan, const c32* prb, hipStream_t st, long long npos_limit = 0) {
    const Geom& ge = h->ge;
    const long long total = (long long)ge.ptheta * ge.nscan;
    int strip0, nstrips;
    strip_range<N>(h->ge, strip0, nstrips);
    const bool window = h->use_window && WinCfg<N>::fits;
    int rc = PTYCHO_OK;
    if (window) {
        rc = sort_positions(h, scan, st);
        if (rc) return rc;
    }
    ColArgs ca{};
    ca.src = f; ca.dst = h->work[slot]; ca.aux = prb; ca.scan = scan; ca.table = h->table; ca.ge = ge;
    ca.k_begin = 0; ca.k_end = (int)((npos_limit > 0 && npos_limit < total) ? npos_limit : total); ca.strip0 = strip0; ca.nstrips = nstrips;
    if (window) {
        ca.order = h->order;
        if constexpr (WinCfg<N>::fits) rc = launch_gatherwin<N, M_FWD>(h, ca, st);
    } else {
        ca.order = nullptr;
        rc = launch_cols<N, -1, M_FWD>(h, ca, st);
    }
    return rc;
}

// finish: 1 = add the result to f / prb (public entry point: the caller zero-filled it); 0 = store it (no zero fill
// needed); -1 = leave it in the fixed-point image for ptycho_cg_*_dir (deterministic option only)
template <int N>
int do_cg_adj_cols(ptycho_handle h, int slot, c32* f, const float* scan, c32* prb, int flg, hipStream_t st,
                   const double* known_omax = nullptr, int finish = 1) {
    const Geom& ge = h->ge;
    const long long total = (long long)ge.ptheta * ge.nscan;
    int strip0, nstrips;
    strip_range<N>(h->ge, strip0, nstrips);
    int rc = sort_positions(h, scan, st);
    if (rc) return rc;
    ColArgs ca{};
    ca.src = h->work[slot]; ca.scan = scan; ca.table = h->table; ca.ge = ge; ca.natural_tiles = 1;
    ca.order = h->order; ca.k_begin = 0; ca.k_end = (int)total; ca.strip0 = strip0; ca.nstrips = nstrips;
    const bool window = h->use_window && WinCfg<N>::fits;
    if (h->deterministic) {
        if (!window) return fail(PTYCHO_ERR_ARG, "option deterministic needs the windowed adjoint kernels (ndet <= 512)");
        rc = det_begin(h, ca, h->work[slot], total * N * N, flg == 0 ? prb : f,
                       flg == 0 ? (long long)ge.ptheta * ge.nprb * ge.nprb : (long long)ge.ptheta * ge.nz * ge.n, flg, st,
                       h->slot_max_ok[slot] ? h->slot_maxw + slot : nullptr, known_omax);
        if (rc) return rc;
    }
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
    if (!rc && h->deterministic) {
        if (finish < 0) h->det_pending = true;
        else rc = det_end(h, flg == 0 ? f : prb, flg, st, finish);
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
    if (a.nrows <= 0) a.nrows = (long long)h->ge.ptheta * h->ge.nscan * N;   // preset: a range of positions (chunked line search)
    a.xa = strip0 * C; a.xb = (strip0 + nstrips) * C;
    long long nb = (a.nrows + B - 1) / B;
    long long grid = nb < (long long)h->n_cu * 8 ? nb : (long long)h->n_cu * 8;
    {   // small problems: at least 16 batches per workgroup while two workgroups per CU remain -- start-up (twiddle table),
        // reduction and fold are per workgroup (512 positions x 256^2: 1.37 -> 1.27 ms per CG iteration)
        long long want = nb / 16;
        if (want < (long long)h->n_cu * 2) want = (long long)h->n_cu * 2;
        if (want < grid) grid = want;
        if (grid > nb) grid = nb;
    }
    if (grid > h->fold_rows) grid = h->fold_rows;
    a.fold = h->fold;
    {
        ProfSpan ps(h, (EP == EP_STATS || EP == EP_STATS_M) ? K_ROWS_STATS : EP == EP_PROJECT ? K_ROWS_PROJECT : (EP == EP_LINESEARCH || EP == EP_LINESEARCH_M) ? K_ROWS_LINESEARCH : K_ROWS_CROSS, st);
        // full-width variant (unconditional masked loads): line search 0.292 -> 0.252 ms per pass, cross 1.65 -> 1.61, statistics 0.540 -> 0.517
        // (rocprofv3, 4096 x 256^2), projection 0.98 -> 1.01 (kept on the predicated variant); 8.39 -> 8.33 ms per CG iteration by the wall clock
        if (a.xa == 0 && a.xb == N && EP != EP_PROJECT) hipLaunchKernelGGL((k_rows_fused<N, EP, true>), dim3((unsigned)grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((k_rows_fused<N, EP, false>), dim3((unsigned)grid), dim3(256), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}


// ---- detector sizes that are not a power of two (k_generic.hpp) ------------------------------------
template <int M, int DIR>
int launch_lines(ptycho_handle h, const c32* src, c32* dst, long long ntiles, bool columns, const int* tile_index, hipStream_t st) {
    const int n = h->ge.ndet;
    LineArgs a{};
    a.src = src; a.dst = dst; a.table = h->table; a.chirp = h->bs_chirp; a.hfilt = h->bs_hfilt;
    a.nlines = ntiles * n; a.n = n; a.ls = columns ? 1 : n; a.es = columns ? n : 1; a.tile_index = tile_index;
    constexpr int T = Plan<M>::T, B = (256 / T) > 0 ? (256 / T) : 1;
    const long long nb = (a.nlines + B - 1) / B;
    const long long grid = nb < (long long)h->n_cu * 8 ? nb : (long long)h->n_cu * 8;
    {
        ProfSpan ps(h, DIR < 0 ? K_ROWS_FWD : K_ROWS_INV, st);
        hipLaunchKernelGGL((k_lines_bluestein<M, DIR>), dim3((unsigned)grid), dim3(256), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

template <int M>
int do_fwd_generic(ptycho_handle h, c32* g, const c32* f, const float* scan, const c32* prb, hipStream_t st) {
    const Geom& ge = h->ge;
    const long long total = (long long)ge.ptheta * ge.nscan;
    const long long npix = total * ge.ndet * ge.ndet;
    {
        ProfSpan ps(h, K_COLS_FWD, st);
        hipLaunchKernelGGL(k_near_generic, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, f, prb, scan, g, ge, npix);
    }
    HIP_TRY(hipGetLastError());
    int rc = launch_lines<M, -1>(h, g, g, total, false, nullptr, st);
    if (rc) return rc;
    return launch_lines<M, -1>(h, g, g, total, true, nullptr, st);
}

template <int M>
int do_adj_generic(ptycho_handle h, c32* f, const c32* g, const float* scan, c32* prb, int flg, hipStream_t st) {
    const Geom& ge = h->ge;
    const long long total = (long long)ge.ptheta * ge.nscan;
    const size_t tile = (size_t)ge.ndet * ge.ndet;
    // object adjoint: LDS overlap-add window over runs of sorted positions (k_adjwin_generic) when the window fits
    const size_t win_bytes = (size_t)(ge.nprb + 8) * (16 + kBucketPx) * sizeof(c32);
    if (!h->scratch) {
        int rc0 = alloc_scratch(h);
        if (rc0) return rc0;
    }
    bool windowed = flg == 0 && h->use_window && win_bytes + sizeof(RunMeta) + 256 <= 160 * 1024;
    if (windowed && win_bytes > 48 * 1024 &&
        hipFuncSetAttribute((const void*)k_adjwin_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)win_bytes) != hipSuccess) {
        (void)hipGetLastError();
        windowed = false;
    }
    if (windowed) {
        int rc = sort_positions(h, scan, st);
        if (rc) return rc;
    }
    ColArgs det{};
    if (h->deterministic) {   // per-workgroup sums into the 64-bit fixed-point image (integer atomics), folded in at the end
        if (flg == 0 && !windowed) return fail(PTYCHO_ERR_ARG, "option deterministic needs the windowed object adjoint (option window, nprb <= ~1000)");
        int rc = det_begin(h, det, g, total * (long long)tile, flg == 0 ? prb : f,
                           flg == 0 ? (long long)ge.ptheta * ge.nprb * ge.nprb : (long long)ge.ptheta * ge.nz * ge.n, flg, st);
        if (rc) return rc;
    }
    for (long long k0 = 0; k0 < total; k0 += h->chunk) {
        const long long k1 = k0 + h->chunk < total ? k0 + h->chunk : total;
        // windowed: the chunk is a range of the SORTED order, its tiles are gathered through order[]
        int rc = launch_lines<M, +1>(h, windowed ? g : g + (size_t)k0 * tile, h->scratch, k1 - k0, false, windowed ? h->order + k0 : nullptr, st);
        if (rc) return rc;
        rc = launch_lines<M, +1>(h, h->scratch, h->scratch, k1 - k0, true, nullptr, st);
        if (rc) return rc;
        if (windowed) {
            ColArgs ca{};
            ca.src = h->scratch; ca.dst = f; ca.aux = prb; ca.scan = scan; ca.ge = ge; ca.order = h->order;
            ca.k_begin = (int)k0; ca.k_end = (int)k1; ca.strip0 = 0; ca.nstrips = (ge.nprb + 15) / 16;
            ca.det_acc = det.det_acc; ca.det = det.det;
            const int np = (int)(k1 - k0);
            int nseg = (h->n_cu * 4 + ca.nstrips - 1) / ca.nstrips;
            if (nseg < 1) nseg = 1;
            int seglen = (np + nseg - 1) / nseg;
            if (seglen < min_seglen(np, ca.nstrips, h->n_cu)) seglen = min_seglen(np, ca.nstrips, h->n_cu);
            if (seglen > kRunMax) seglen = kRunMax;
            nseg = (np + seglen - 1) / seglen;
            ProfSpan ps(h, K_COLS_ADJ_OBJ, st);
            hipLaunchKernelGGL(k_adjwin_generic, dim3((unsigned)(ca.nstrips * nseg)), dim3(256), win_bytes, st, ca, seglen);
        } else if (flg == 0) {
            const long long npix = (k1 - k0) * ge.nprb * ge.nprb;
            ProfSpan ps(h, K_COLS_ADJ_OBJ, st);
            hipLaunchKernelGGL(k_adj_obj_generic, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, f, (const c32*)prb, scan,
                               (const c32*)h->scratch, ge, (int)k0, npix);
        } else {
            const int npp = ge.nprb * ge.nprb;
            int groups = (int)((k1 - k0 + 63) / 64);
            if (groups > 1024) groups = 1024;
            const int pgroup = (int)((k1 - k0 + groups - 1) / groups);
            ProfSpan ps(h, K_COLS_ADJ_PRB, st);
            hipLaunchKernelGGL(k_adj_prb_generic, dim3((unsigned)((npp + 255) / 256), (unsigned)groups), dim3(256), 0, st,
                               (const c32*)f, prb, scan, (const c32*)h->scratch, ge, (int)k0, (int)k1, pgroup, det.det_acc, det.det);
        }
        HIP_TRY(hipGetLastError());
    }
    if (h->deterministic) return det_end(h, flg == 0 ? f : prb, flg, st);
    return PTYCHO_OK;
}

template <int M>
int do_fft2_generic(ptycho_handle h, c32* dst, const c32* src, long long nbatch, int dir, hipStream_t st) {
    int rc = dir < 0 ? launch_lines<M, -1>(h, src, dst, nbatch, false, nullptr, st)
                     : launch_lines<M, +1>(h, src, dst, nbatch, false, nullptr, st);
    if (rc) return rc;
    return dir < 0 ? launch_lines<M, -1>(h, dst, dst, nbatch, true, nullptr, st)
                   : launch_lines<M, +1>(h, dst, dst, nbatch, true, nullptr, st);
}

// detector sizes with their own Stockham plan (fft_core.hpp): the powers of two 16 ... 2048 and five sizes with an odd factor
inline bool native_size(size_t n) {
    return ((n & (n - 1)) == 0 && n >= 16 && n <= 2048) || n == 48 || n == 80 || n == 96 || n == 112 || n == 192;
}
#ifdef PTY_FEW_SIZES   // A/B builds (make ab): the benchmarked sizes only -- a third of the compile time
#define PTY_DISPATCH_POW2_CASES(CALL)                             \
        case 64: { constexpr int NN = 64; return CALL; }         \
        case 128: { constexpr int NN = 128; return CALL; }       \
        case 256: { constexpr int NN = 256; return CALL; }       \
        case 512: { constexpr int NN = 512; return CALL; }
#define PTY_DISPATCH(N_, CALL)                                   \
    switch (N_) {                                                \
        PTY_DISPATCH_POW2_CASES(CALL)                            \
        default: return fail(PTYCHO_ERR_ARG, "A/B build: ndet 64, 128, 256 and 512 only"); \
    }
#else
#define PTY_DISPATCH_POW2_CASES(CALL)                             \
        case 16: { constexpr int NN = 16; return CALL; }         \
        case 32: { constexpr int NN = 32; return CALL; }         \
        case 64: { constexpr int NN = 64; return CALL; }         \
        case 128: { constexpr int NN = 128; return CALL; }       \
        case 256: { constexpr int NN = 256; return CALL; }       \
        case 512: { constexpr int NN = 512; return CALL; }       \
        case 1024: { constexpr int NN = 1024; return CALL; }     \
        case 2048: { constexpr int NN = 2048; return CALL; }
#define PTY_DISPATCH(N_, CALL)                                   \
    switch (N_) {                                                \
        PTY_DISPATCH_POW2_CASES(CALL)                            \
        case 48: { constexpr int NN = 48; return CALL; }         \
        case 80: { constexpr int NN = 80; return CALL; }         \
        case 96: { constexpr int NN = 96; return CALL; }         \
        case 112: { constexpr int NN = 112; return CALL; }       \
        case 192: { constexpr int NN = 192; return CALL; }       \
        default: return fail(PTYCHO_ERR_ARG, "this entry point needs a detector size with a Stockham plan: a power of two in [16, 2048] or 48, 80, 96, 112, 192"); \
    }
#endif
// length of a Bluestein plan: always a power of two
#define PTY_DISPATCH_POW2(N_, CALL)                              \
    switch (N_) {                                                \
        PTY_DISPATCH_POW2_CASES(CALL)                            \
        default: return fail(PTYCHO_ERR_ARG, "internal: Bluestein plan length is not a power of two"); \
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
    if ((h->trust_order || h->native_order) && h->order_scan == scan) return PTYCHO_OK;
    h->order_scan = scan;
    const int iblocks = (total + 255) / 256;
    int nslices = (2 * h->n_cu + iblocks - 1) / iblocks;
    if (nslices > iblocks) nslices = iblocks;   // = number of key tiles
    if (nslices < 1) nslices = 1;
    {
        ProfSpan ps(h, K_SORT, st);
        const int pc = (total + h->sort_chunks - 1) / h->sort_chunks;   // positions per chunk
        hipLaunchKernelGGL(k_rank_positions, dim3((unsigned)(iblocks * nslices)), dim3(256), 0, st, scan, h->ge, total, nslices,
                           h->sort_counts, h->sort_counts + total, h->order, pc);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

int alloc_sort(ptycho_handle h) {
    const size_t total = (size_t)h->ge.ptheta * h->ge.nscan;
    const size_t nwords = total + (total + 255) / 256;
    HIP_TRY(hipMalloc((void**)&h->order, total * sizeof(int)));
    HIP_TRY(hipMalloc((void**)&h->sort_counts, nwords * sizeof(int)));
    HIP_TRY(hipMemset(h->sort_counts, 0, nwords * sizeof(int)));
    return PTYCHO_OK;
}

void release(ptycho_handle h) {
    if (h->slot_maxw) { (void)hipFree(h->slot_maxw); h->slot_maxw = nullptr; }
    void* ptrs[] = {h->det_acc, (void*)h->fold.part, (void*)h->fold.ticket, h->det_words, h->bs_chirp, h->bs_hfilt, h->table, h->scratch, h->order, h->sort_counts, h->zoom_phase, h->prbp, h->reg_ip, h->reg_best, h->reg_shifts};
    h->det_acc = nullptr; h->fold.part = nullptr; h->fold.ticket = nullptr; h->det_words = nullptr; h->zoom_phase = nullptr; h->prbp = nullptr; h->bs_chirp = nullptr; h->bs_hfilt = nullptr; h->reg_ip = nullptr; h->reg_best = nullptr; h->reg_shifts = nullptr;
    for (auto& w : h->work) { if (w) (void)hipFree(w); w = nullptr; }
    for (void* q : ptrs)
        if (q) (void)hipFree(q);
    h->table = nullptr; h->scratch = nullptr; h->order = nullptr; h->sort_counts = nullptr;
    for (auto& sp : h->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    h->spans.clear();

}

}  // namespace

extern "C" {

const char* ptycho_last_error(void) { return g_err.c_str(); }
#ifdef PTY_STAMPS
// diagnostic build only: read and clear the in-kernel phase stamps (tools/stamps.py)
int ptycho_debug_stamps(ptycho_handle h, unsigned long long* out24) {
    if (!h || !out24) return PTYCHO_ERR_ARG;
    if (!h->stamps) {
        if (hipMalloc((void**)&h->stamps, 24 * sizeof(unsigned long long)) != hipSuccess) return PTYCHO_ERR_HIP;
        (void)hipMemset(h->stamps, 0, 24 * sizeof(unsigned long long));
    }
    if (hipDeviceSynchronize() != hipSuccess) return PTYCHO_ERR_HIP;
    (void)hipMemcpy(out24, h->stamps, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipMemset(h->stamps, 0, 24 * sizeof(unsigned long long));
    return PTYCHO_OK;
}
#endif
const char* ptycho_version(void) { return "ptychohip 0.4 (gfx950)"; }

int ptycho_create(ptycho_handle* out, size_t ptheta, size_t nz, size_t n, size_t nscan, size_t ndet, size_t nprb) {
    if (!out) return fail(PTYCHO_ERR_ARG, "out is null");
    *out = nullptr;
    if (ptheta == 0 || nz == 0 || n == 0 || nscan == 0 || ndet == 0 || nprb == 0)
        return fail(PTYCHO_ERR_ARG, "all sizes must be positive");
    const bool pow2 = native_size(ndet);   // (name kept: "has a plan of its own"; every other size runs the Bluestein lines)
    if (ndet < 2 || ndet > 2048 || (!pow2 && ndet > 1024))
        return fail(PTYCHO_ERR_ARG, "ndet must be in [2, 1024], or a power of two up to 2048");
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
    size_t tlen = ndet;
    if (!pow2) {   // Bluestein plan: smallest power of two >= 2 ndet - 1 (and >= 16)
        tlen = 16;
        while (tlen < 2 * ndet - 1) tlen *= 2;
        h->bs_m = (int)tlen;
    }
    std::vector<c32> tab(tlen);
    for (size_t k = 0; k < tlen; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)tlen;
        tab[k] = c32{(float)std::cos(ang), (float)std::sin(ang)};
    }
    e = hipMalloc((void**)&h->table, tlen * sizeof(c32));
    if (e == hipSuccess) e = hipMemcpy(h->table, tab.data(), tlen * sizeof(c32), hipMemcpyHostToDevice);
    if (e == hipSuccess && !pow2) {
        // chirp b[m] = exp(-i pi m^2 / n) (phase reduced modulo 2 n exactly) and H = FFT_M(conj b, circular) / M in float64
        const size_t n = ndet, M = tlen;
        std::vector<double> br(n), bi(n), hr(M, 0.0), hi(M, 0.0), Hr(M), Hi(M);
        std::vector<c32> bf(n), Hf(M);
        for (size_t m = 0; m < n; ++m) {
            const double ang = -M_PI * (double)((m * m) % (2 * n)) / (double)n;
            br[m] = std::cos(ang); bi[m] = std::sin(ang);
            bf[m] = c32{(float)br[m], (float)bi[m]};
            hr[m] = br[m]; hi[m] = -bi[m];
            if (m) { hr[M - m] = br[m]; hi[M - m] = -bi[m]; }
        }
        for (size_t k = 0; k < M; ++k) {   // plain O(M^2) DFT, once per handle (M <= 2048)
            double sr = 0.0, si = 0.0;
            for (size_t m = 0; m < M; ++m) {
                if (hr[m] == 0.0 && hi[m] == 0.0) continue;
                const double ang = -2.0 * M_PI * (double)((k * m) % M) / (double)M;
                const double c = std::cos(ang), s2 = std::sin(ang);
                sr += hr[m] * c - hi[m] * s2;
                si += hr[m] * s2 + hi[m] * c;
            }
            Hf[k] = c32{(float)(sr / (double)M), (float)(si / (double)M)};
        }
        e = hipMalloc((void**)&h->bs_chirp, n * sizeof(c32));
        if (e == hipSuccess) e = hipMemcpy(h->bs_chirp, bf.data(), n * sizeof(c32), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc((void**)&h->bs_hfilt, M * sizeof(c32));
        if (e == hipSuccess) e = hipMemcpy(h->bs_hfilt, Hf.data(), M * sizeof(c32), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        release(h);
        delete h;
        return fail(PTYCHO_ERR_HIP, std::string("twiddle table: ") + hipGetErrorString(e));
    }
    h->use_window = exp_env("PTYCHO_HIP_WINDOW", 1) != 0;
    h->use_split = exp_env("PTYCHO_HIP_SPLIT", 1) != 0;
    h->use_fused = exp_env("PTYCHO_HIP_FUSED", 0);

    h->chunk = default_chunk(h->ge);
    h->fold_rows = h->n_cu * 8 > 2048 ? h->n_cu * 8 : 2048;
    e = hipMalloc((void**)&h->fold.part, (size_t)h->fold_rows * kFoldStride * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMalloc((void**)&h->fold.ticket, 64);
    if (e == hipSuccess) e = hipMemset(h->fold.ticket, 0, 64);
    if (e != hipSuccess) {
        release(h);
        delete h;
        return fail(PTYCHO_ERR_HIP, std::string("fold scratch: ") + hipGetErrorString(e));
    }
    int rc = alloc_sort(h);   // the adjoint's scratch (up to 4 GiB) is allocated by the first ptycho_adj call
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
        default:
            if (which >= 200 && which < 200 + ptycho_handle_s::kSlots) return h->work[which - 200] ? 1 : 0;   // CG work slot allocated?
            return -1;
    }
}

int ptycho_set_option(ptycho_handle h, const char* name, long long value) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!name) return fail(PTYCHO_ERR_ARG, "null option name");
    if (std::strcmp(name, "chunk") == 0) {
        h->chunk = value > 0 ? value : default_chunk(h->ge);
        HIP_TRY(hipDeviceSynchronize());
        if (h->scratch) { HIP_TRY(hipFree(h->scratch)); h->scratch = nullptr; }
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "window") == 0) {
        h->use_window = value != 0;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "trust_order") == 0) {
        h->trust_order = value != 0;
        h->native_order = 0;
        if (!h->trust_order) h->order_scan = nullptr;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "split") == 0) {
        h->use_split = value != 0;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "tile") == 0) {
        h->use_tile = value != 0;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "deterministic") == 0) {
        h->deterministic = value != 0;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "compact_modes") == 0) {   // value = number of probe modes (0: slot pairs); also makes the order chunk-major
        if (value < 0 || value > kMaxModes || (value > 0 && (long long)h->ge.ptheta * value > (1 << 19)))
            return fail(PTYCHO_ERR_ARG, "compact_modes must be in [0, 8]");
        h->compact_modes = (int)value;
        h->sort_chunks = value > 1 ? (int)value : 1;
        h->order_scan = nullptr;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "fused") == 0) {
#ifdef PTYCHO_EXPERIMENTS
        h->use_fused = (int)value;
        return PTYCHO_OK;
#else
        if (value == 0) return PTYCHO_OK;
        return fail(PTYCHO_ERR_ARG, "option fused: the single-launch forward is an experiment (measured slower, DESIGN.md); build with -DPTYCHO_EXPERIMENTS");
#endif
    }
    if (std::strcmp(name, "release_scratch") == 0) {   // give back the adjoint's intermediate (<= 4 GiB; the next ptycho_adj allocates it again)
        HIP_TRY(hipDeviceSynchronize());
        if (h->scratch) { HIP_TRY(hipFree(h->scratch)); h->scratch = nullptr; }
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "release_work") == 0) {   // give back one CG work slot (a farplane); the next stage that writes it allocates it again
        if (value < 0 || value >= ptycho_handle_s::kSlots) return fail(PTYCHO_ERR_ARG, "work slot out of range");
        HIP_TRY(hipDeviceSynchronize());
        if (h->work[value]) { HIP_TRY(hipFree(h->work[value])); h->work[value] = nullptr; }
        h->slot_max_ok[value] = false;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "defer_finish") == 0) {
        h->defer_finish = value != 0;
        h->det_pending = false;
        return PTYCHO_OK;
    }
    if (std::strcmp(name, "ls_fused_decide") == 0) {
        h->ls_fused_decide = value != 0;
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
    if (!ms || !launches || n < 16) return fail(PTYCHO_ERR_ARG, "need arrays of at least 16 entries (18 for every kernel id)");
    for (int i = 0; i < n; ++i) { ms[i] = 0.0; launches[i] = 0; }
    for (auto& sp : h->spans) {
        HIP_TRY(hipEventSynchronize(sp.b));
        float t = 0.f;
        HIP_TRY(hipEventElapsedTime(&t, sp.a, sp.b));
        if (sp.kid < n) {   // callers with the 16-entry arrays of earlier headers do not see ids 16 / 17
            ms[sp.kid] += t;
            launches[sp.kid] += 1;
        }
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
    h->native_order = 0;
    if (h->bs_m) { PTY_DISPATCH_POW2(h->bs_m, (do_fwd_generic<NN>(h, (c32*)g, (const c32*)f, (const float*)scan, (const c32*)prb, st))); }
    PTY_DISPATCH(h->ge.ndet, (do_fwd<NN>(h, (c32*)g, (const c32*)f, (const float*)scan, (const c32*)prb, st)));
}

int ptycho_adj(ptycho_handle h, void* f, const void* g, const void* scan, void* prb, int flg, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!g || !f || !scan || !prb) return fail(PTYCHO_ERR_ARG, "null operand");
    if (flg != 0 && flg != 1) return fail(PTYCHO_ERR_ARG, "flg must be 0 (object) or 1 (probe)");
    h->native_order = 0;
    hipStream_t st = (hipStream_t)stream;
    if (h->bs_m) { PTY_DISPATCH_POW2(h->bs_m, (do_adj_generic<NN>(h, (c32*)f, (const c32*)g, (const float*)scan, (c32*)prb, flg, st))); }
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
    if (!slot_ready(h, slot)) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_adj_cols<NN>(h, slot, (c32*)f, (const float*)scan, (c32*)prb, flg, st)));
}

int ptycho_cg_stats(ptycho_handle h, int slot, const void* data, double* sums, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!data || !sums) return fail(PTYCHO_ERR_ARG, "null operand");
    if (!slot_ready(h, slot)) return fail(PTYCHO_ERR_ARG, "work slot is empty");
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
    if (!slot_ready(h, src_slot)) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    rc = ensure_work(h, dst_slot);
    if (rc) return rc;
    RowFusedArgs a{};
    a.s1 = h->work[src_slot]; a.out = h->work[dst_slot]; a.data = (const float*)data; a.sums = cost; a.ab = ab;
    hipStream_t st = (hipStream_t)stream;
    rc = project_maxword(h, dst_slot, a, st);
    if (rc) return rc;
    h->slot_max_ok[dst_slot] = a.maxword != nullptr;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_PROJECT>(h, a, st)));
}

int ptycho_cg_linesearch(ptycho_handle h, int slot1, int slot2, const void* data, const double* ab, double gamma0,
                         int ncand, double* costs, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!data || !costs) return fail(PTYCHO_ERR_ARG, "null operand");
    if (ncand < 1 || ncand > kMaxCand) return fail(PTYCHO_ERR_ARG, "ncand must be in [1, 16]");
    if (!slot_ready(h, slot1) || !slot_ready(h, slot2))
        return fail(PTYCHO_ERR_ARG, "work slot is empty");
    RowFusedArgs a{};
    a.s1 = h->work[slot1]; a.s2 = h->work[slot2]; a.data = (const float*)data; a.sums = costs; a.ab = ab;
    a.gamma0 = (float)gamma0; a.ncand = ncand;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_LINESEARCH>(h, a, st)));
}

int ptycho_cg_project_multi(ptycho_handle h, int src_slot, int dst_slot, const void* data, const void* inten,
                            const double* ab, int slot_unscaled, double* cost, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!data || !cost || !inten) return fail(PTYCHO_ERR_ARG, "null operand");
    if (!slot_ready(h, src_slot)) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    rc = ensure_work(h, dst_slot);
    if (rc) return rc;
    RowFusedArgs a{};
    a.s1 = h->work[src_slot]; a.out = h->work[dst_slot]; a.data = (const float*)data; a.sums = cost; a.ab = ab;
    a.inten = (const float*)inten;
    a.first = slot_unscaled ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    rc = project_maxword(h, dst_slot, a, st);
    if (rc) return rc;
    h->slot_max_ok[dst_slot] = a.maxword != nullptr;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_PROJECT>(h, a, st)));
}

int ptycho_cg_intensity_modes(ptycho_handle h, int nmodes, void* inten, const void* data, double* sums, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (nmodes < 1 || nmodes > kMaxModes) return fail(PTYCHO_ERR_ARG, "nmodes must be in [1, 8]");
    if (!inten && !sums) return fail(PTYCHO_ERR_ARG, "nothing to compute: inten and sums are both null");
    if (sums && !data) return fail(PTYCHO_ERR_ARG, "null operand");
    RowFusedArgs a{};
    for (int k = 0; k < nmodes; ++k) {
        if (!slot_ready(h, slot_a(h, k))) return fail(PTYCHO_ERR_ARG, "work slot is empty");
        a.sm[k] = h->work[slot_a(h, k)];
    }
    a.nmodes = nmodes;
    a.acc1 = (float*)inten; a.data = (const float*)data; a.sums = sums;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_STATS_M>(h, a, st)));
}

int ptycho_cg_linesearch_modes(ptycho_handle h, int mode0, int nmodes, const void* data, const void* inten,
                               const double* ab, double gamma0, int ncand, double* costs, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!data || !costs) return fail(PTYCHO_ERR_ARG, "null operand");
    if (mode0 < 0 || nmodes < 1 || mode0 + nmodes > kMaxModes) return fail(PTYCHO_ERR_ARG, "modes must lie in [0, 8)");
    if (ncand < 1 || ncand > kMaxCand) return fail(PTYCHO_ERR_ARG, "ncand must be in [1, 16]");
    if (h->compact_modes && nmodes != 1) return fail(PTYCHO_ERR_ARG, "compact slot layout: one mode pair per call (use ptycho_cg_linesearch_chunk)");
    RowFusedArgs a{};
    for (int k = 0; k < nmodes; ++k) {
        if (!slot_ready(h, slot_a(h, mode0 + k)) || !slot_ready(h, slot_b(h, mode0 + k))) return fail(PTYCHO_ERR_ARG, "work slot is empty");
        a.sm[2 * k] = h->work[slot_a(h, mode0 + k)];
        a.sm[2 * k + 1] = h->work[slot_b(h, mode0 + k)];
    }
    a.nmodes = nmodes;
    a.data = (const float*)data; a.inten = (const float*)inten; a.sums = costs; a.ab = ab;
    a.gamma0 = (float)gamma0; a.ncand = ncand;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_LINESEARCH_M>(h, a, st)));
}





int ptycho_fft2(ptycho_handle h, void* dst, const void* src, size_t nbatch, int dir, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!dst || !src) return fail(PTYCHO_ERR_ARG, "null operand");
    if (dir != -1 && dir != 1) return fail(PTYCHO_ERR_ARG, "dir must be -1 (forward) or +1 (inverse)");
    if (nbatch == 0) return PTYCHO_OK;
    hipStream_t st = (hipStream_t)stream;
    if (h->bs_m) { PTY_DISPATCH_POW2(h->bs_m, (do_fft2_generic<NN>(h, (c32*)dst, (const c32*)src, (long long)nbatch, dir, st))); }
    PTY_DISPATCH(h->ge.ndet, (do_fft2<NN>(h, (c32*)dst, (const c32*)src, (long long)nbatch, dir, st)));
}

}  // extern "C"

template <int N>
int do_cg_argmax(ptycho_handle h, int slot, unsigned long long* best, hipStream_t st, bool zeroed = false, int npos_limit = 0) {
    using CC = ColCfg<N>;
    const int npos = npos_limit > 0 ? npos_limit : h->ge.ptheta * h->ge.nscan;
    constexpr int nstrips = N / CC::C;
    int ng = (h->n_cu * 8) / nstrips;
    if (ng < 1) ng = 1;
    if (ng > npos) ng = npos;
    if (!zeroed) HIP_TRY(hipMemsetAsync(best, 0, (size_t)npos * sizeof(unsigned long long), st));
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
    if (!image_product) {   // NULL: the image product lives in work slot 2 (free during the position correction)
        rc = ensure_work(h, 2);
        if (rc) return rc;
        if (slot1 == 2 || slot2 == 2) return fail(PTYCHO_ERR_ARG, "slot 2 is taken by the image product");
        image_product = h->work[2];
    }
    if (!slot_ready(h, slot1) || !slot_ready(h, slot2))
        return fail(PTYCHO_ERR_ARG, "work slot is empty");
    RowFusedArgs a{};
    a.s1 = h->work[slot1]; a.s2 = h->work[slot2]; a.out = h->work[slot2]; a.ip = (c32*)image_product;
    h->slot_max_ok[slot2] = false;
    a.gamma0 = (float)gamma;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_CROSS>(h, a, st)));
}

extern "C" int ptycho_cg_argmax(ptycho_handle h, int slot, void* best, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!best) return fail(PTYCHO_ERR_ARG, "null operand");
    if (!slot_ready(h, slot)) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_argmax<NN>(h, slot, (unsigned long long*)best, st)));
}

namespace {
int zoom_impl(ptycho_handle h, const void* image_product, const void* best, const void* vt,
              const void* lz, int nc, int ups, double upsample_factor, void* shifts, float* scan_add, void* stream, int npos_limit = 0);
}
extern "C" int ptycho_cg_zoom(ptycho_handle h, const void* image_product, const void* best, const void* vt,
                              const void* lz, int nc, int ups, double upsample_factor, void* shifts, void* stream) {
    return zoom_impl(h, image_product, best, vt, lz, nc, ups, upsample_factor, shifts, nullptr, stream);
}
namespace {
// scan_add: scan[0, :] += shifts (ptycho.py:403) by the kernel that finds them (native CG stages)
int zoom_impl(ptycho_handle h, const void* image_product, const void* best, const void* vt,
              const void* lz, int nc, int ups, double upsample_factor, void* shifts, float* scan_add, void* stream, int npos_limit) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!image_product) {   // NULL: work slot 2 (see ptycho_cg_cross)
        if (!slot_ready(h, 2)) return fail(PTYCHO_ERR_ARG, "work slot is empty");
        image_product = h->work[2];
    }
    if (!best || !vt || !lz || !shifts) return fail(PTYCHO_ERR_ARG, "null operand");
    const int N = h->ge.ndet;
    const int nthreads = N > 256 ? N : 256;
    if (N % 16 != 0 || N > 1024) return fail(PTYCHO_ERR_ARG, "zoomed DFT kernel needs ndet %% 16 == 0 and ndet <= 1024");
    if (ups < 1 || ups > nthreads || nc < 0 || nc > kZoomRK || !(upsample_factor >= 1.0))
        return fail(PTYCHO_ERR_ARG, "zoomed DFT window, rank split or upsample factor out of range");
    const int npos_all = h->ge.ptheta * h->ge.nscan;
    const int npos = npos_limit > 0 ? npos_limit : npos_all;
    hipStream_t st = (hipStream_t)stream;
    if (!h->zoom_phase) {   // px, py: complex128 [npos][N] each; coarse shifts: float64 [npos][2]
        HIP_TRY(hipMalloc(&h->zoom_phase, (size_t)npos_all * N * 2 * sizeof(double2) + (size_t)npos_all * 2 * sizeof(double)));
        HIP_TRY(hipMemset(h->zoom_phase, 0, (size_t)npos_all * N * 2 * sizeof(double2) + (size_t)npos_all * 2 * sizeof(double)));
    }
    double2* ppx = (double2*)h->zoom_phase;
    double2* ppy = ppx + (size_t)npos * N;
    double* coarse = (double*)(ppy + (size_t)npos * N);
    {
        ProfSpan ps(h, K_ZOOM, st);
        const c32* ip = (const c32*)image_product;
        const double *pv = (const double*)vt, *pl = (const double*)lz;
        hipLaunchKernelGGL(k_zoom_prepare, dim3((unsigned)npos), dim3(N < 256 ? N : 256), 0, st,
                           (const unsigned long long*)best, N, ups, upsample_factor, ppx, ppy, coarse);
        static const bool no_mfma = exp_env("PTYCHO_HIP_ZOOM_SCALAR", 0) != 0;   // comparison knob
        int* none = nullptr;
        if (N % 64 == 0 && !no_mfma) {
            if (N <= 256)
                hipLaunchKernelGGL((k_zoom_mfma<256>), dim3((unsigned)npos), dim3(256), 0, st, ip, ppx, ppy, pv, pl, N, nc, ups, none, coarse, upsample_factor, (double*)shifts, scan_add);
            else if (N <= 512)
                hipLaunchKernelGGL((k_zoom_mfma<512>), dim3((unsigned)npos), dim3(512), 0, st, ip, ppx, ppy, pv, pl, N, nc, ups, none, coarse, upsample_factor, (double*)shifts, scan_add);
            else
                hipLaunchKernelGGL((k_zoom_mfma<1024>), dim3((unsigned)npos), dim3(1024), 0, st, ip, ppx, ppy, pv, pl, N, nc, ups, none, coarse, upsample_factor, (double*)shifts, scan_add);
        } else if (N <= 256)
            hipLaunchKernelGGL((k_zoom_argmax<256, 8>), dim3((unsigned)npos), dim3(256), (size_t)N * 8 * sizeof(double2), st,
                               ip, ppx, ppy, pv, pl, N, nc, ups, none, coarse, upsample_factor, (double*)shifts, scan_add);
        else if (N <= 512)
            hipLaunchKernelGGL((k_zoom_argmax<512, 4>), dim3((unsigned)npos), dim3(512), (size_t)N * 4 * sizeof(double2), st,
                               ip, ppx, ppy, pv, pl, N, nc, ups, none, coarse, upsample_factor, (double*)shifts, scan_add);
        else
            hipLaunchKernelGGL((k_zoom_argmax<1024, 2>), dim3((unsigned)npos), dim3(1024), (size_t)N * 2 * sizeof(double2), st,
                               ip, ppx, ppy, pv, pl, N, nc, ups, none, coarse, upsample_factor, (double*)shifts, scan_add);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}
}  // namespace


// ---------------------------------------------------------------------------------------------------
// Native CG stages (include/ptycho_hip.h, "device-resident CG iteration")
// ---------------------------------------------------------------------------------------------------
namespace {

inline unsigned small_grid(ptycho_handle h, long long n) {
    long long g = (n + 255) / 256;
    const long long cap = (long long)h->n_cu * 4;
    if (g > cap) g = cap;
    return (unsigned)(g < 1 ? 1 : g);
}

// One line-search pass over slots 0 / 1 on the device-resident state.  decide_next >= 0: the pass's last workgroup also
// replays line_search_sqr on the totals and sizes the pass that follows (single GPU); < 0: the caller all-reduces the
// costs and calls k_cg_ls_decide.
template <int N>
int do_ls_pass(ptycho_handle h, const float* data, const double* ab, double* state, hipStream_t st, int which, int decide_next) {
    RowFusedArgs a{};
    a.s1 = h->work[0]; a.s2 = h->work[1]; a.data = data; a.ab = ab; a.st = state;
    a.sums = state + PTYCHO_ST_COSTS; a.gamma0 = 1.0f; a.ncand = kMaxCand;
    a.overwrite = 1;
    a.decide = decide_next >= 0 ? 1 : 0;
    a.decide_which = which;
    a.decide_gamma_word = which == 0 ? (int)PTYCHO_ST_GAMMA_PSI : (int)PTYCHO_ST_GAMMA_PRB;
    a.decide_next = decide_next;
    return do_cg_rows<N, EP_LINESEARCH>(h, a, st);
}
int ls_pass(ptycho_handle h, const void* data, int use_ab, double* state, hipStream_t st, int which, int decide_next) {
    if (!slot_ready(h, 0) || !slot_ready(h, 1)) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    const double* ab = use_ab ? state + PTYCHO_ST_A : nullptr;
    PTY_DISPATCH(h->ge.ndet, (do_ls_pass<NN>(h, (const float*)data, ab, state, st, which, decide_next)));
}
// sizes (groups of 16 step lengths) of the pass that ptycho_cg_ls_next(pass) issues: <= 16 step lengths first (sized from the
// last accepted index), then 16, 32, 64 more: 2^-106 < 1e-32 is covered.  For callers that pay a collective per pass:
// 6 then 7 (32, then the 80 that are left), or 5 (all 112 at once).
constexpr int kLsNext[8] = {0, 1, 2, 4, 0, kLsGroupsMax, 2, 5};

template <int N>
int do_cross_dev(ptycho_handle h, const double* gamma_dev, hipStream_t st, int s1_slot, int s2_slot) {
    RowFusedArgs a{};
    a.s1 = h->work[s1_slot]; a.s2 = h->work[s2_slot]; a.out = h->work[s2_slot]; a.ip = h->reg_ip; a.gamma_dev = gamma_dev;
    a.best_zero = h->reg_best; a.nbest = h->ge.nscan;   // the arg-max pass that follows finds them cleared
    a.nrows = (long long)h->ge.nscan * N;                // angle 0 only (ptycho.py:399-403: fwd(...)[0], scan[0, :] += shifts)
    h->slot_max_ok[s2_slot] = false;
    return do_cg_rows<N, EP_CROSS>(h, a, st);
}
int cross_dev(ptycho_handle h, const double* gamma_dev, hipStream_t st, int s1_slot = 0, int s2_slot = 1) {
    if (!slot_ready(h, s1_slot) || !slot_ready(h, s2_slot)) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    PTY_DISPATCH(h->ge.ndet, (do_cross_dev<NN>(h, gamma_dev, st, s1_slot, s2_slot)));
}

template <int N>
int do_cg_fwd_cols_modes(ptycho_handle h, int nmodes, c32* const* dst, const c32* f, const float* scan, const c32* const* prbs,
                         int k_begin, int k_end, hipStream_t st, const double* skip);   // defined below

// Column passes of fwd(obj, probe) -> slot_p and fwd(obj, ones) -> slot_o in ONE launch that gathers the object patch
// once per position (k_cols_gatherwin<..., NM = 2>): the position correction's operands (ptycho.py:399-402) ride along
// with the passes the object step makes anyway.  Falls back to two passes where the shared-gather kernel does not apply.
template <int N>
int do_fwd_cols_pair(ptycho_handle h, int slot_p, int slot_o, const c32* f, const float* scan, const c32* prb, const c32* ones, hipStream_t st) {
    c32* dst[2] = {h->work[slot_p], h->work[slot_o]};
    const c32* pr[2] = {prb, ones};
    return do_cg_fwd_cols_modes<N>(h, 2, dst, f, scan, pr, 0, h->ge.ptheta * h->ge.nscan, st, nullptr);
}
int fwd_cols_pair(ptycho_handle h, int slot_p, int slot_o, const void* f, const void* scan, const void* prb, const void* ones, hipStream_t st) {
    int rc = ensure_work(h, slot_p);
    if (!rc) rc = ensure_work(h, slot_o);
    if (rc) return rc;
    PTY_DISPATCH(h->ge.ndet, (do_fwd_cols_pair<NN>(h, slot_p, slot_o, (const c32*)f, (const float*)scan, (const c32*)prb, (const c32*)ones, st)));
}

int argmax_native(ptycho_handle h, int slot, unsigned long long* best, hipStream_t st) {   // best was cleared by the CROSS stage
    PTY_DISPATCH(h->ge.ndet, (do_cg_argmax<NN>(h, slot, best, st, true, h->ge.nscan)));   // angle 0 only
}
int fwd_cols_angle0(ptycho_handle h, int slot, const void* f, const void* scan, const void* prb, hipStream_t st) {
    int rc = ensure_work(h, slot);
    if (rc) return rc;
    PTY_DISPATCH(h->ge.ndet, (do_cg_fwd_cols<NN>(h, slot, (const c32*)f, (const float*)scan, (const c32*)prb, st, h->ge.nscan)));
}

int check_stage(ptycho_handle h, const void* state) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!state) return fail(PTYCHO_ERR_ARG, "null state");
    return PTYCHO_OK;
}

// row stages of the native loop: sums are STORED by the stage's last workgroup (no zero fill of the state)
int stats_native(ptycho_handle h, int slot, const void* data, double* sums, hipStream_t st) {
    RowFusedArgs a{};
    a.s1 = h->work[slot]; a.data = (const float*)data; a.sums = sums; a.overwrite = 1;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_STATS>(h, a, st)));
}
int project_native(ptycho_handle h, int src_slot, int dst_slot, const void* data, const double* ab, double* cost, hipStream_t st) {
    int rc = ensure_work(h, dst_slot);
    if (rc) return rc;
    RowFusedArgs a{};
    a.s1 = h->work[src_slot]; a.out = h->work[dst_slot]; a.data = (const float*)data; a.sums = cost; a.ab = ab; a.overwrite = 1;
    rc = project_maxword(h, dst_slot, a, st);
    if (rc) return rc;
    h->slot_max_ok[dst_slot] = a.maxword != nullptr;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_PROJECT>(h, a, st)));
}
int adj_cols_native(ptycho_handle h, int slot, void* f, const void* scan, void* prb, int flg, const double* known_omax, int finish, hipStream_t st) {
    PTY_DISPATCH(h->ge.ndet, (do_cg_adj_cols<NN>(h, slot, (c32*)f, (const float*)scan, (c32*)prb, flg, st, known_omax, finish)));
}

}  // namespace

extern "C" {

int ptycho_cg_obj_begin2(ptycho_handle h, double* state, const void* psi, const void* scan, const void* prb,
                         const void* ones_prb, const void* data, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (!psi || !scan || !prb || !data) return fail(PTYCHO_ERR_ARG, "null operand");
    hipStream_t st = (hipStream_t)stream;
    h->native_order = 1;   // the native loop keeps track of scan itself (ptycho_cg_obj_finish invalidates the order)
    h->det_pending = false;
    if (ones_prb) rc = fwd_cols_pair(h, 0, 2, psi, scan, prb, ones_prb, st);     // + slot 2 <- column pass of fwd(psi, 1)
    else rc = ptycho_cg_fwd_cols(h, 0, psi, scan, prb, stream);
    if (rc) return rc;
    return stats_native(h, 0, data, state + PTYCHO_ST_A, st);
}
int ptycho_cg_obj_begin(ptycho_handle h, double* state, const void* psi, const void* scan, const void* prb,
                        const void* data, void* stream) {
    return ptycho_cg_obj_begin2(h, state, psi, scan, prb, nullptr, data, stream);
}

int ptycho_cg_obj_grad(ptycho_handle h, double* state, const void* scan, void* prb, const void* data, void* grad,
                       void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (!scan || !prb || !data || !grad) return fail(PTYCHO_ERR_ARG, "null operand");
    hipStream_t st = (hipStream_t)stream;
    const Geom& ge = h->ge;
    const long long np = (long long)ge.ptheta * ge.nprb * ge.nprb, no = (long long)ge.ptheta * ge.nz * ge.n;
    // probe *= a / b (ptycho.py:344) and max |probe| (the gradient normalisation of :356 and the fixed-point scale) in one pass
    hipLaunchKernelGGL(k_cg_absmax, dim3(fold_grid(h, np, 4)), dim3(256), 0, st, (c32*)prb, np, state + PTYCHO_ST_MAX_PRB,
                       (double*)nullptr, (const double*)(state + PTYCHO_ST_A), h->fold);
    h->max_prb_valid = true;
    rc = project_native(h, 0, 1, data, state + PTYCHO_ST_A, state + PTYCHO_ST_COST, st);
    if (rc) return rc;
    const bool window = h->use_window && h->ge.ndet <= 512;
    const bool det = h->deterministic && window;
    if (!det) HIP_TRY(hipMemsetAsync(grad, 0, (size_t)no * sizeof(c32), st));   // float atomics accumulate into grad
    return adj_cols_native(h, 1, grad, scan, prb, 0, state + PTYCHO_ST_MAX_PRB, det ? (h->defer_finish ? -1 : 0) : 1, st);
}

int ptycho_cg_obj_dir(ptycho_handle h, double* state, int first, const void* scan, const void* prb, const void* data,
                      void* grad, void* grad0, void* dpsi, void* stream) {
    return ptycho_cg_obj_dir2(h, state, first, scan, prb, nullptr, data, grad, grad0, dpsi, stream);
}
int ptycho_cg_obj_dir2(ptycho_handle h, double* state, int first, const void* scan, const void* prb, const void* ones_prb,
                       const void* data, void* grad, void* grad0, void* dpsi, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (!scan || !prb || !data || !grad || !grad0 || !dpsi) return fail(PTYCHO_ERR_ARG, "null operand");
    hipStream_t st = (hipStream_t)stream;
    const Geom& ge = h->ge;
    const long long np = (long long)ge.ptheta * ge.nprb * ge.nprb, no = (long long)ge.ptheta * ge.nz * ge.n;
    if (!h->max_prb_valid)
        hipLaunchKernelGGL(k_cg_absmax, dim3(fold_grid(h, np, 4)), dim3(256), 0, st, (c32*)prb, np, state + PTYCHO_ST_MAX_PRB,
                           (double*)nullptr, (const double*)nullptr, h->fold);
    h->max_prb_valid = false;
    hipLaunchKernelGGL(k_cg_dy_reduce, dim3(fold_grid(h, no, 4)), dim3(256), 0, st, (c32*)grad, (const c32*)dpsi, (const c32*)grad0, no,
                       (const double*)(state + PTYCHO_ST_MAX_PRB), 0.0f, 0.0f, state + PTYCHO_ST_DY_OBJ, first,
                       h->det_pending ? h->det_acc : (long long*)nullptr, h->last_det, h->fold);
    h->det_pending = false;
    hipLaunchKernelGGL(k_cg_dy_update, dim3(small_grid(h, no)), dim3(256), 0, st, (c32*)dpsi, (c32*)grad0, (const c32*)grad, no,
                       (const double*)(state + PTYCHO_ST_DY_OBJ), first, state, 0);
    if (ones_prb) rc = fwd_cols_pair(h, 1, 3, dpsi, scan, prb, ones_prb, st);    // + slot 3 <- column pass of fwd(dpsi, 1)
    else rc = ptycho_cg_fwd_cols(h, 1, dpsi, scan, prb, stream);
    if (rc) return rc;
    return ls_pass(h, data, 1, state, st, 0, h->ls_fused_decide ? kLsNext[1] : -1);
}

int ptycho_cg_ls_next(ptycho_handle h, double* state, int which, int pass, const void* data, int use_ab, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (which < 0 || which > 1 || pass < 1 || pass > 7 || !data) return fail(PTYCHO_ERR_ARG, "bad line-search stage");
    hipStream_t st = (hipStream_t)stream;
    if (h->ls_fused_decide) {
        // the pass before this call has decided on its own totals and sized this one: just issue it
        if (pass > 3) return pass == 4 ? (int)PTYCHO_OK : fail(PTYCHO_ERR_ARG, "option ls_fused_decide: passes 1, 2, 3, 4 only");
        return ls_pass(h, data, use_ab, state, st, which, kLsNext[pass + 1]);
    }
    const int next_groups = kLsNext[pass];
    hipLaunchKernelGGL(k_cg_ls_decide, dim3(1), dim3(1), 0, st, state, which,
                       which == 0 ? (int)PTYCHO_ST_GAMMA_PSI : (int)PTYCHO_ST_GAMMA_PRB, next_groups);
    HIP_TRY(hipGetLastError());
    if (pass == 4) return PTYCHO_OK;
    return ls_pass(h, data, use_ab, state, st, which, -1);
}

int ptycho_cg_obj_finish(ptycho_handle h, double* state, int correct_positions, void* psi, const void* dpsi, void* scan,
                         const void* ones_prb, const void* vt, const void* lz, int nc, int ups, double upsample_factor,
                         void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (!psi || !dpsi || !scan) return fail(PTYCHO_ERR_ARG, "null operand");
    hipStream_t st = (hipStream_t)stream;
    const Geom& ge = h->ge;
    const long long no = (long long)ge.ptheta * ge.nz * ge.n;
    if (correct_positions) {   // angle 0 only, as in the reference (ptycho.py:399-403)
        if (!ones_prb || !vt || !lz) return fail(PTYCHO_ERR_ARG, "null operand");
        const size_t npos = (size_t)ge.nscan;
        if (!h->reg_ip) {
            HIP_TRY(hipMalloc((void**)&h->reg_ip, npos * ge.ndet * ge.ndet * sizeof(c32)));
            HIP_TRY(hipMalloc((void**)&h->reg_best, npos * sizeof(unsigned long long)));
            HIP_TRY(hipMalloc((void**)&h->reg_shifts, npos * 2 * sizeof(double)));
        }
        // ptycho.py:399-402: tmp1 = fwd(psi, 1), tmp2 = fwd(psi + gamma dpsi, 1) = tmp1 + gamma fwd(dpsi, 1)
        // 2: ptycho_cg_reg_prepare (or ptycho_cg_obj_begin2) left the column pass of tmp1 in slot 2;
        // 3: and ptycho_cg_obj_dir2 the column pass of fwd(dpsi, 1) in slot 3
        const int s1 = correct_positions >= 2 ? 2 : 0, s2 = correct_positions == 3 ? 3 : 1;
        if (s1 == 0) {
            rc = fwd_cols_angle0(h, 0, psi, scan, ones_prb, st);
            if (rc) return rc;
        }
        if (s2 == 1) {
            rc = fwd_cols_angle0(h, 1, dpsi, scan, ones_prb, st);
            if (rc) return rc;
        }
        rc = cross_dev(h, state + PTYCHO_ST_GAMMA_PSI, st, s1, s2);
        if (rc) return rc;
        rc = argmax_native(h, s2, h->reg_best, st);
        if (rc) return rc;
        // the kernel that finds the shifts also adds them to scan[0, :] (ptycho.py:403)
        rc = zoom_impl(h, h->reg_ip, h->reg_best, vt, lz, nc, ups, upsample_factor, h->reg_shifts, (float*)scan, stream, ge.nscan);
        if (rc) return rc;
        h->order_scan = nullptr;   // the positions moved: the next column pass sorts again
    }
    hipLaunchKernelGGL(k_cg_axpy, dim3(small_grid(h, no)), dim3(256), 0, st, (c32*)psi, (const c32*)dpsi, no,
                       (const double*)(state + PTYCHO_ST_GAMMA_PSI));
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

int ptycho_cg_reg_prepare(ptycho_handle h, double* state, const void* psi, const void* scan, const void* ones_prb, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (!psi || !scan || !ones_prb) return fail(PTYCHO_ERR_ARG, "null operand");
    return fwd_cols_angle0(h, 2, psi, scan, ones_prb, (hipStream_t)stream);
}

int ptycho_cg_prb_grad(ptycho_handle h, double* state, const void* psi, const void* scan, const void* prb,
                       const void* data, void* gprb, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (!psi || !scan || !prb || !data || !gprb) return fail(PTYCHO_ERR_ARG, "null operand");
    hipStream_t st = (hipStream_t)stream;
    const Geom& ge = h->ge;
    const long long np = (long long)ge.ptheta * ge.nprb * ge.nprb, no = (long long)ge.ptheta * ge.nz * ge.n;
    rc = ptycho_cg_fwd_cols(h, 0, psi, scan, prb, stream);
    if (rc) return rc;
    // max |psi|: the gradient normalisation of ptycho.py:431 and the fixed-point scale of the probe adjoint
    hipLaunchKernelGGL(k_cg_absmax, dim3(fold_grid(h, no, 4)), dim3(256), 0, st, (c32*)psi, no, state + PTYCHO_ST_MAX_PSI,
                       (double*)nullptr, (const double*)nullptr, h->fold);
    h->max_psi_valid = true;
    rc = project_native(h, 0, 1, data, nullptr, state + PTYCHO_ST_COST2, st);
    if (rc) return rc;
    const bool window = h->use_window && h->ge.ndet <= 512;
    const bool det = h->deterministic && window;
    if (!det) HIP_TRY(hipMemsetAsync(gprb, 0, (size_t)np * sizeof(c32), st));
    return adj_cols_native(h, 1, (void*)psi, scan, gprb, 1, state + PTYCHO_ST_MAX_PSI, det ? (h->defer_finish ? -1 : 0) : 1, st);
}

int ptycho_cg_prb_dir(ptycho_handle h, double* state, int first, double nscan_total, double nmodes, const void* psi,
                      const void* scan, const void* data, void* gprb, void* gprb0, void* dprb, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (!psi || !scan || !data || !gprb || !gprb0 || !dprb) return fail(PTYCHO_ERR_ARG, "null operand");
    hipStream_t st = (hipStream_t)stream;
    const Geom& ge = h->ge;
    const long long np = (long long)ge.ptheta * ge.nprb * ge.nprb, no = (long long)ge.ptheta * ge.nz * ge.n;
    if (!h->max_psi_valid)
        hipLaunchKernelGGL(k_cg_absmax, dim3(fold_grid(h, no, 4)), dim3(256), 0, st, (c32*)psi, no, state + PTYCHO_ST_MAX_PSI,
                           (double*)nullptr, (const double*)nullptr, h->fold);
    h->max_psi_valid = false;
    hipLaunchKernelGGL(k_cg_dy_reduce, dim3(fold_grid(h, np, 4)), dim3(256), 0, st, (c32*)gprb, (const c32*)dprb, (const c32*)gprb0, np,
                       (const double*)(state + PTYCHO_ST_MAX_PSI), (float)nscan_total, (float)nmodes, state + PTYCHO_ST_DY_PRB, first,
                       h->det_pending ? h->det_acc : (long long*)nullptr, h->last_det, h->fold);
    h->det_pending = false;
    hipLaunchKernelGGL(k_cg_dy_update, dim3(small_grid(h, np)), dim3(256), 0, st, (c32*)dprb, (c32*)gprb0, (const c32*)gprb, np,
                       (const double*)(state + PTYCHO_ST_DY_PRB), first, state, 1);
    rc = ptycho_cg_fwd_cols(h, 1, psi, scan, dprb, stream);
    if (rc) return rc;
    return ls_pass(h, data, 0, state, st, 1, h->ls_fused_decide ? kLsNext[1] : -1);
}

int ptycho_cg_prb_finish(ptycho_handle h, double* state, void* prb, const void* dprb, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (!prb || !dprb) return fail(PTYCHO_ERR_ARG, "null operand");
    hipStream_t st = (hipStream_t)stream;
    const long long np = (long long)h->ge.ptheta * h->ge.nprb * h->ge.nprb;
    hipLaunchKernelGGL(k_cg_axpy, dim3(small_grid(h, np)), dim3(256), 0, st, (c32*)prb, (const c32*)dprb, np,
                       (const double*)(state + PTYCHO_ST_GAMMA_PRB));
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

}  // extern "C"


// ---- several probe modes per column pass; compact slot layout with a chunked line search ----
namespace {

// (CW: round 4 tried the two-probe pass of the CG iteration on 32-column strips in one resident round of workgroups -- 512 threads,
// 140 KiB of LDS, one workgroup per CU --: 8.35 against 8.26 ms per iteration, profiles/r04/cg_experiments.txt; 16 columns stay)
template <int N, int NM, int CW = 0>
int launch_gatherwin_modes(ptycho_handle h, ColArgs a, hipStream_t st) {
    constexpr int NTHREADS = Plan<N>::T * (CW ? CW : ColCfg<N>::C);
    const int np = a.k_end - a.k_begin;
    if (np <= 0 || a.nstrips <= 0) return PTYCHO_OK;
    int nseg = (h->n_cu * 4 + a.nstrips - 1) / a.nstrips;
    if (nseg < 1) nseg = 1;
    int seglen = (np + nseg - 1) / nseg;
    if (seglen < min_seglen(np, a.nstrips, h->n_cu)) seglen = min_seglen(np, a.nstrips, h->n_cu);
    if (seglen > kRunMax) seglen = kRunMax;
    nseg = (np + seglen - 1) / seglen;
    a.nt = 0;
#ifdef PTY_STAMPS
    a.stamps = h->stamps;
#endif
    {
        ProfSpan ps(h, K_COLS_FWD, st);
        hipLaunchKernelGGL((k_cols_gatherwin<N, M_FWD, false, NM, CW>), dim3((unsigned)(a.nstrips * nseg)), dim3(NTHREADS), 0, st, a, seglen);
    }
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

template <int N>
int do_cg_fwd_cols_modes(ptycho_handle h, int nmodes, c32* const* dst, const c32* f, const float* scan, const c32* const* prbs,
                         int k_begin, int k_end, hipStream_t st, const double* skip) {
    const Geom& ge = h->ge;
    int strip0, nstrips;
    strip_range<N>(h->ge, strip0, nstrips);
    int rc = sort_positions(h, scan, st);
    if (rc) return rc;
    ColArgs ca{};
    ca.src = f; ca.scan = scan; ca.table = h->table; ca.ge = ge; ca.order = h->order;
    ca.k_begin = k_begin; ca.k_end = k_end; ca.strip0 = strip0; ca.nstrips = nstrips;
    ca.skip = skip;
    static const int nm_max = exp_env("PTYCHO_HIP_NMMAX", (PTY_AB & 2) ? 2 : ((PTY_AB & 4) ? 1 : 4));   // comparison knob
    int k = 0;
    while (k < nmodes) {
        const int left = nmodes - k;
        if constexpr (WinCfg<N>::fits && N <= 512) {
            if (left >= 4 && nm_max >= 4) {
                for (int j = 0; j < 4; ++j) { ca.auxm[j] = prbs[k + j]; ca.dstm[j] = dst[k + j]; }
                rc = launch_gatherwin_modes<N, 4>(h, ca, st);
                if (rc) return rc;
                k += 4;
                continue;
            }
            if (left >= 2 && nm_max >= 2) {
                for (int j = 0; j < 2; ++j) { ca.auxm[j] = prbs[k + j]; ca.dstm[j] = dst[k + j]; }
                rc = launch_gatherwin_modes<N, 2>(h, ca, st);
                if (rc) return rc;
                k += 2;
                continue;
            }
        }
        ca.aux = prbs[k]; ca.dst = dst[k];
        if constexpr (WinCfg<N>::fits) {
            rc = launch_gatherwin<N, M_FWD>(h, ca, st);
        } else {
            ColArgs cb = ca;
            cb.order = nullptr;
            if (k_begin != 0 || k_end != ge.ptheta * ge.nscan) return fail(PTYCHO_ERR_ARG, "position ranges need the windowed column pass (ndet <= 512)");
            rc = launch_cols<N, -1, M_FWD>(h, cb, st);
        }
        if (rc) return rc;
        ++k;
    }
    return PTYCHO_OK;
}

template <int N>
int do_ls_chunk(ptycho_handle h, RowFusedArgs a, hipStream_t st) { return do_cg_rows<N, EP_LINESEARCH_M>(h, a, st); }

}  // namespace

namespace {
int fwd_cols_modes_impl(ptycho_handle h, int nmodes, int mode0, const void* f, const void* scan,
                        const void* const* prbs, int into_b, int chunk, void* stream, const double* skip);
}
extern "C" int ptycho_cg_fwd_cols_modes(ptycho_handle h, int nmodes, int mode0, const void* f, const void* scan,
                                        const void* const* prbs, int into_b, int chunk, void* stream) {
    return fwd_cols_modes_impl(h, nmodes, mode0, f, scan, prbs, into_b, chunk, stream, nullptr);
}
namespace {
int fwd_cols_modes_impl(ptycho_handle h, int nmodes, int mode0, const void* f, const void* scan,
                        const void* const* prbs, int into_b, int chunk, void* stream, const double* skip) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!f || !scan || !prbs || nmodes < 1 || mode0 < 0 || mode0 + nmodes > kMaxModes) return fail(PTYCHO_ERR_ARG, "bad operand");
    const long long total = (long long)h->ge.ptheta * h->ge.nscan;
    const size_t tile = (size_t)h->ge.ndet * h->ge.ndet;
    c32* dst[kMaxModes];
    const c32* pr[kMaxModes];
    int k_begin = 0, k_end = (int)total;
    if (into_b) {   // B sub-slots: mode k of the positions of this chunk at tile k * pc of the shared slot
        if (!h->compact_modes || nmodes != h->compact_modes || mode0 != 0 || chunk < 0 || chunk >= h->sort_chunks)
            return fail(PTYCHO_ERR_ARG, "chunked column pass needs the compact slot layout and all modes");
        const long long pc = (total + h->sort_chunks - 1) / h->sort_chunks;
        k_begin = (int)(chunk * pc);
        k_end = (int)((chunk + 1) * pc < total ? (chunk + 1) * pc : total);
        rc = ensure_work(h, slot_b(h, 0));
        if (rc) return rc;
        for (int k = 0; k < nmodes; ++k) dst[k] = h->work[slot_b(h, 0)] + (size_t)k * pc * tile - (size_t)k_begin * tile;
    } else {
        for (int k = 0; k < nmodes; ++k) {
            rc = ensure_work(h, slot_a(h, mode0 + k));
            if (rc) return rc;
            dst[k] = h->work[slot_a(h, mode0 + k)];
        }
    }
    for (int k = 0; k < nmodes; ++k) {
        if (!prbs[k]) return fail(PTYCHO_ERR_ARG, "null probe");
        pr[k] = (const c32*)prbs[k];
    }
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_fwd_cols_modes<NN>(h, nmodes, dst, (const c32*)f, (const float*)scan, pr, k_begin, k_end, st, skip)));
}
}  // namespace

extern "C" int ptycho_cg_linesearch_chunk(ptycho_handle h, int chunk, const void* data, const double* ab, double gamma0,
                                          int ncand, double* costs, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!data || !costs) return fail(PTYCHO_ERR_ARG, "null operand");
    if (!h->compact_modes || chunk < 0 || chunk >= h->sort_chunks) return fail(PTYCHO_ERR_ARG, "chunked line search needs the compact slot layout");
    if (ncand < 1 || ncand > kMaxCand) return fail(PTYCHO_ERR_ARG, "ncand must be in [1, 16]");
    const int M = h->compact_modes;
    const long long total = (long long)h->ge.ptheta * h->ge.nscan;
    const long long pc = (total + h->sort_chunks - 1) / h->sort_chunks;
    const long long p0 = chunk * pc, p1 = (chunk + 1) * pc < total ? (chunk + 1) * pc : total;
    if (p1 <= p0) return PTYCHO_OK;
    const size_t tile = (size_t)h->ge.ndet * h->ge.ndet;
    RowFusedArgs a{};
    for (int k = 0; k < M; ++k) {
        if (!slot_ready(h, slot_a(h, k)) || !slot_ready(h, slot_b(h, 0))) return fail(PTYCHO_ERR_ARG, "work slot is empty");
        a.sm[2 * k] = h->work[slot_a(h, k)] + (size_t)p0 * tile;
        a.sm[2 * k + 1] = h->work[slot_b(h, 0)] + (size_t)k * pc * tile;
    }
    a.nmodes = M;
    a.data = (const float*)data + (size_t)p0 * tile; a.sums = costs; a.ab = ab;
    a.gamma0 = (float)gamma0; a.ncand = ncand;
    a.nrows = (p1 - p0) * h->ge.ndet;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_ls_chunk<NN>(h, a, st)));
}


// ---- line searches of the multi-mode loop on the device-resident state (ptycho.py:383-393, 451-461 with nmodes > 1) ----
namespace {
__global__ void k_cg_ls_begin(double* __restrict__ st, const int which) {
    if (threadIdx.x == 0 && blockIdx.x == 0) ls_prepare_dev(st, which);
}
template <int N>
int do_ls_modes(ptycho_handle h, RowFusedArgs a, hipStream_t st) { return do_cg_rows<N, EP_LINESEARCH_M>(h, a, st); }
}  // namespace

extern "C" {

int ptycho_cg_ls_begin(ptycho_handle h, double* state, int which, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (which < 0 || which > 1) return fail(PTYCHO_ERR_ARG, "which must be 0 (object) or 1 (probe)");
    hipLaunchKernelGGL(k_cg_ls_begin, dim3(1), dim3(1), 0, (hipStream_t)stream, state, which);
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

int ptycho_cg_ls_decide(ptycho_handle h, double* state, int which, int next_groups, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (which < 0 || which > 1 || next_groups < 0 || next_groups > kLsGroupsMax) return fail(PTYCHO_ERR_ARG, "bad line-search decision");
    hipLaunchKernelGGL(k_cg_ls_decide, dim3(1), dim3(1), 0, (hipStream_t)stream, state, which,
                       which == 0 ? (int)PTYCHO_ST_GAMMA_PSI : (int)PTYCHO_ST_GAMMA_PRB, next_groups);
    HIP_TRY(hipGetLastError());
    return PTYCHO_OK;
}

int ptycho_cg_ls_obj_chunk(ptycho_handle h, double* state, int chunk, const void* dpsi, const void* scan,
                           const void* const* prbs, const void* data, const double* ab, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (!dpsi || !scan || !prbs || !data) return fail(PTYCHO_ERR_ARG, "null operand");
    if (!h->compact_modes || chunk < 0 || chunk >= h->sort_chunks) return fail(PTYCHO_ERR_ARG, "chunked line search needs the compact slot layout");
    const int M = h->compact_modes;
    // direction column passes of this chunk, all modes side by side in the shared slot (skipped once the search is resolved)
    rc = fwd_cols_modes_impl(h, M, 0, dpsi, scan, prbs, 1, chunk, stream, state + PTYCHO_ST_LS_RESOLVED);
    if (rc) return rc;
    const long long total = (long long)h->ge.ptheta * h->ge.nscan;
    const long long pc = (total + h->sort_chunks - 1) / h->sort_chunks;
    const long long p0 = chunk * pc, p1 = (chunk + 1) * pc < total ? (chunk + 1) * pc : total;
    if (p1 <= p0) return PTYCHO_OK;
    const size_t tile = (size_t)h->ge.ndet * h->ge.ndet;
    RowFusedArgs a{};
    for (int k = 0; k < M; ++k) {
        if (!slot_ready(h, slot_a(h, k)) || !slot_ready(h, slot_b(h, 0))) return fail(PTYCHO_ERR_ARG, "work slot is empty");
        a.sm[2 * k] = h->work[slot_a(h, k)] + (size_t)p0 * tile;
        a.sm[2 * k + 1] = h->work[slot_b(h, 0)] + (size_t)k * pc * tile;
    }
    a.nmodes = M;
    a.data = (const float*)data + (size_t)p0 * tile; a.ab = ab;
    a.st = state; a.sums = state + PTYCHO_ST_COSTS; a.gamma0 = 1.0f; a.ncand = kMaxCand;
    a.overwrite = chunk == 0 ? 1 : 0;      // the chunks of one pass accumulate; the first one stores
    a.nrows = (p1 - p0) * h->ge.ndet;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_ls_modes<NN>(h, a, st)));
}

int ptycho_cg_ls_prb_pass(ptycho_handle h, double* state, int mode, const void* data, const void* inten, void* stream) {
    int rc = check_stage(h, state);
    if (rc) return rc;
    if (!data || !inten) return fail(PTYCHO_ERR_ARG, "null operand");
    if (mode < 0 || mode >= kMaxModes) return fail(PTYCHO_ERR_ARG, "modes must lie in [0, 8)");
    if (!slot_ready(h, slot_a(h, mode)) || !slot_ready(h, slot_b(h, mode))) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    RowFusedArgs a{};
    a.sm[0] = h->work[slot_a(h, mode)];
    a.sm[1] = h->work[slot_b(h, mode)];
    a.nmodes = 1;
    a.data = (const float*)data; a.inten = (const float*)inten;
    a.st = state; a.sums = state + PTYCHO_ST_COSTS; a.gamma0 = 1.0f; a.ncand = kMaxCand;
    a.overwrite = 1;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_ls_modes<NN>(h, a, st)));
}

int ptycho_cg_cross_dev(ptycho_handle h, int slot1, int slot2, const double* gamma_dev, void* image_product, void* stream) {
    int rc = check_handle(h);
    if (rc) return rc;
    if (!gamma_dev) return fail(PTYCHO_ERR_ARG, "null operand");
    if (!image_product) {   // NULL: the image product lives in work slot 2 (free during the position correction)
        rc = ensure_work(h, 2);
        if (rc) return rc;
        if (slot1 == 2 || slot2 == 2) return fail(PTYCHO_ERR_ARG, "slot 2 is taken by the image product");
        image_product = h->work[2];
    }
    if (!slot_ready(h, slot1) || !slot_ready(h, slot2)) return fail(PTYCHO_ERR_ARG, "work slot is empty");
    RowFusedArgs a{};
    a.s1 = h->work[slot1]; a.s2 = h->work[slot2]; a.out = h->work[slot2]; a.ip = (c32*)image_product;
    a.gamma_dev = gamma_dev;
    h->slot_max_ok[slot2] = false;
    hipStream_t st = (hipStream_t)stream;
    PTY_DISPATCH(h->ge.ndet, (do_cg_rows<NN, EP_CROSS>(h, a, st)));
}

}  // extern "C"
