/* ptycho_hip.h -- C ABI of libptychohip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the native class `ptychofft` of nikitinvv/libtike-cufft
 * (reference paths relative to /root/reference):
 *
 *   ptychofft::ptychofft(ptheta,nz,n,nscan,ndet,nprb)  src/include/ptychofft.cuh:35-36, src/cuda/ptychofft.cu:5-41
 *   ptychofft::fwd(g,f,scan,prb)                       src/include/ptychofft.cuh:40,    src/cuda/ptychofft.cu:60-73
 *   ptychofft::adj(f,g,scan,prb,flg)                   src/include/ptychofft.cuh:42,    src/cuda/ptychofft.cu:76-88
 *   ptychofft::free() / ~ptychofft()                   src/include/ptychofft.cuh:38,43, src/cuda/ptychofft.cu:44-57
 *   read-only fields ptheta,nz,n,nscan,ndet,nprb       src/cuda/swig/ptychofft.i:11-16
 *
 * Differences from the reference interface, on purpose:
 *   - plain C functions on an opaque handle instead of a SWIG/pybind11 class;
 *   - every entry point returns an int status (0 = ok) and validates its
 *     arguments; ptycho_last_error() describes the last failure of the calling
 *     thread (the reference returns void and checks nothing);
 *   - the caller's HIP stream is passed explicitly (the reference uses the
 *     legacy default stream); no per-call operand state is kept in the handle.
 *     Calls on ONE handle must be stream-serialised (same stream, or ordered by
 *     events): the handle owns scratch that the kernels of a call share -- the
 *     adjoint's intermediate, the position order, the CG work slots, the
 *     fixed-point image of the deterministic adjoints, and the ticket + partial-sum
 *     table through which every CG reduction adds up its workgroups in a fixed
 *     order.  Two stage calls of one handle running concurrently on different
 *     streams would mix their partial sums; use one handle per stream;
 *   - taps outside the object read as zero / are dropped instead of being
 *     undefined behaviour (the reference only rejects negative positions,
 *     src/cuda/kernels.cu:39).
 *
 * All device pointers are borrowed for the duration of the call.  Layouts
 * (C-contiguous, complex64 = interleaved float re,im):
 *   f    object   complex64 [ptheta][nz][n]
 *   g    farplane complex64 [ptheta][nscan][ndet][ndet]   (DC at [0][0])
 *   prb  probe    complex64 [ptheta][nprb][nprb]
 *   scan          float32   [ptheta][nscan][2]  ([..][0] = row/y, [..][1] = column/x)
 * ndet: 2 .. 1024, or a power of two up to 2048.  Sizes with a Stockham plan of their own run the fused kernels and the
 * CG stages below: the powers of two 16 .. 2048 and 48, 80, 96, 112, 192 (mixed radix 3 / 5 / 7 x 4|8 x 4|8; 112 is the crop of the
 * reference's tests/test_fsc.py:115-120); every other size runs a Bluestein transform on the next power-of-two plan (operators
 * and ptycho_fft2 only).  nprb <= ndet.
 */
#ifndef PTYCHO_HIP_H
#define PTYCHO_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ptycho_handle_s* ptycho_handle;

enum {
    PTYCHO_OK = 0,
    PTYCHO_ERR_ARG = 1,      /* invalid argument / unsupported size */
    PTYCHO_ERR_HIP = 2,      /* a HIP runtime call failed */
    PTYCHO_ERR_FREED = 3     /* handle used after ptycho_free */
};

/* ptychofft ctor: builds the twiddle table and the chunk scratch.  ptheta * nscan < 2^30 positions; the processing order of
 * the windowed kernels is computed by a single-launch ranking kernel with n^2 / 2 key compares (39 us at 4096 positions,
 * ~0.1 ms at the 32768 positions of a configs[3] shard, tens of ms at 3e5): shard larger jobs over handles / GPUs by
 * position, as BASELINE.json configs[3] does, or keep scan unchanged between calls (option "trust_order"). */
int ptycho_create(ptycho_handle* out, size_t ptheta, size_t nz, size_t n,
                  size_t nscan, size_t ndet, size_t nprb);
/* ptychofft::free -- idempotent; the handle stays valid for ptycho_get/destroy. */
int ptycho_free(ptycho_handle h);
/* ~ptychofft: free + release the handle itself. */
int ptycho_destroy(ptycho_handle h);
/* read-only size fields: which = 0 ptheta, 1 nz, 2 n, 3 nscan, 4 ndet, 5 nprb;
 * 100: positions per launch pair of the adjoint (option "chunk"), 101: option "window";
 * 200 + slot (slot < 16): 1 if CG work slot `slot` (one farplane, ptheta * nscan * ndet^2 * 8 bytes) is allocated. */
long long ptycho_get(ptycho_handle h, int which);

/* g = F Q f : probe (x) bilinear patch, 1/ndet, centred zero pad, 2-D DFT. */
int ptycho_fwd(ptycho_handle h, void* g, const void* f, const void* scan,
               const void* prb, void* stream);
/* flg = 0: f   += Q* F* g   (f must be zeroed by the caller, as in the reference)
 * flg = 1: prb += O* F* g   (prb must be zeroed by the caller)
 * Same argument order as the reference; g is never modified. */
int ptycho_adj(ptycho_handle h, void* f, const void* g, const void* scan,
               void* prb, int flg, void* stream);

/* Batched unnormalised 2-D DFT of nbatch ndet x ndet complex64 tiles
 * (dir = -1 forward, +1 inverse; dst may equal src).  This is the cuFFT plan of
 * src/cuda/ptychofft.cu:14-20 exposed for the position registration
 * (cp.fft.ifft2 in src/libtike/cufft/ptycho.py:204). */
int ptycho_fft2(ptycho_handle h, void* dst, const void* src, size_t nbatch,
                int dir, void* stream);

/* ---- fused CG-stage entry points ("plus fused CG-stage entry points") ----
 * The elementwise stages of CGPtychoSolver.run (src/libtike/cufft/ptycho.py:325-393) are
 * fused into the row pass of the DFT so that farplanes are never materialised.  The
 * handle owns work buffers ("slots" 0..15, one farplane each, allocated on first use; the
 * single-mode loop uses 0 and 1, the multi-mode loop one pair (2k, 2k+1) per probe mode)
 * that hold column-pass intermediates between calls.  sums/cost/costs and ab are DEVICE
 * pointers to float64 (ab = {a, b} of ptycho.py:342-343; NULL means scale 1); the
 * kernels ADD into sums/cost/costs, the caller zeroes them.  Every such sum is formed in a fixed order (per-workgroup
 * partials, folded by the last workgroup to finish): the same inputs give the same bits, whatever the scheduling.
 *   ptycho_cg_fwd_cols   slot <- column pass of fwd(f, scan, prb)            (ptycho.py:332)
 *   ptycho_cg_stats      sums += { sum sqrt(|g|^2 d), sum |g|^2 }            (ptycho.py:333,342-343)
 *   ptycho_cg_project    dst  <- IDFT_x( fpsi - sqrt(d) fpsi / (sqrt(I)+1e-32) ), cost += ||sqrt I - sqrt d||^2
 *                                                                             (ptycho.py:344-353,310)
 *   ptycho_cg_adj_cols   column pass of adj from a slot into f (flg 0) / prb (flg 1) (ptycho.py:352,429)
 *   ptycho_cg_linesearch costs[j] += f(p1 + y_j^2 p2 + y_j p3), y_j = gamma0 2^-j, j < ncand <= 16;
 *                        costs[ncand] += f(p1); p1,p2,p3 from slot1 (scaled by a/b) and slot2
 *                                                                             (ptycho.py:383-393,253-281) */
int ptycho_cg_fwd_cols(ptycho_handle h, int slot, const void* f, const void* scan,
                       const void* prb, void* stream);
int ptycho_cg_stats(ptycho_handle h, int slot, const void* data, double* sums, void* stream);
int ptycho_cg_project(ptycho_handle h, int src_slot, int dst_slot, const void* data,
                      const double* ab, double* cost, void* stream);
int ptycho_cg_adj_cols(ptycho_handle h, int slot, void* f, const void* scan, void* prb,
                       int flg, void* stream);
int ptycho_cg_linesearch(ptycho_handle h, int slot1, int slot2, const void* data,
                         const double* ab, double gamma0, int ncand, double* costs,
                         void* stream);

/* Multi-mode variants (ptycho.py:330-333,349-356,386-391,425-434 loop over probe modes):
 * the summed intensity is a float32 array [ptheta][nscan][ndet][ndet] owned by the caller;
 * each mode k has its own pair of work slots (2k, 2k+1).
 *   ptycho_cg_intensity_modes  inten = sum_k |g_k|^2 over the slots 2k, k < nmodes (inten may be NULL), and,
 *                              if sums is given, sums += { sum sqrt(inten d), sum inten }: one pass
 *   ptycho_cg_project_multi    as ptycho_cg_project with I = inten * (a/b)^2; slot_unscaled = 0: the slot was
 *                              made with the rescaled probe, 1: with the probe before its rescale
 *   ptycho_cg_linesearch_modes as ptycho_cg_linesearch with p1,p2,p3 summed in registers over the mode
 *                              pairs (slot 2k, slot 2k+1), mode0 <= k < mode0 + nmodes <= 8 (no arrays); p1 = inten if given
 */
int ptycho_cg_intensity_modes(ptycho_handle h, int nmodes, void* inten, const void* data, double* sums,
                              void* stream);
int ptycho_cg_project_multi(ptycho_handle h, int src_slot, int dst_slot, const void* data,
                            const void* inten, const double* ab, int slot_unscaled, double* cost,
                            void* stream);
int ptycho_cg_linesearch_modes(ptycho_handle h, int mode0, int nmodes, const void* data, const void* inten,
                               const double* ab, double gamma0, int ncand, double* costs,
                               void* stream);

/* Compact multi-mode layout (option "compact_modes" = M): mode k keeps its forward column pass in slot k and ALL
 * modes share one further slot M (projected residual of one mode at a time; direction column passes), M + 1
 * farplanes instead of 2 M.  The object line search (ptycho.py:383-393) needs fwd(dpsi, probe_k) of every mode at
 * once, so it runs over M equal ranges ("chunks") of positions: for chunk c the M direction column passes of
 * those positions go side by side into the shared slot and one line-search pass adds the chunk's costs -- the
 * same work as before in M smaller launches.  The option also makes the position order chunk-major.
 *   ptycho_cg_fwd_cols_modes   column passes of fwd(f, scan, prbs[k]) for nmodes modes (mode0 ...), the object
 *                              patch gathered once per position for up to four modes per launch (ptycho.py:330-333
 *                              gathers per mode); into_b = 0: into the modes' own slots, all positions;
 *                              into_b = 1 (compact layout, all modes): the positions of `chunk` into the shared slot
 *   ptycho_cg_linesearch_chunk costs += line-search costs of chunk `chunk` (pairs: slot k, shared slot part k) */
int ptycho_cg_fwd_cols_modes(ptycho_handle h, int nmodes, int mode0, const void* f, const void* scan,
                             const void* const* prbs, int into_b, int chunk, void* stream);
int ptycho_cg_linesearch_chunk(ptycho_handle h, int chunk, const void* data, const double* ab, double gamma0,
                               int ncand, double* costs, void* stream);

/* Position correction (ptycho.py:398-403 + 198-207), fused: with the column passes of
 * fwd(psi, 1) in slot1 and fwd(dpsi, 1) in slot2,
 *   ptycho_cg_cross   image_product = u1 conj(u2), u2 = u1 + gamma G dpsi (complex64
 *                     [ptheta][nscan][ndet][ndet], kept for the zoomed DFT); slot2 <- IDFT_x(product)
 *   ptycho_cg_argmax  IDFT_y of the slot, |.|, first maximum per position as
 *                     best[p] = (float bits of the value << 32) | (0xffffffff - flat index) */
/* image_product = NULL (here and in ptycho_cg_zoom): the product is kept in work slot 2 instead of a caller buffer */
int ptycho_cg_cross(ptycho_handle h, int slot1, int slot2, double gamma, void* image_product,
                    void* stream);
int ptycho_cg_argmax(ptycho_handle h, int slot, void* best, void* stream);

/* Sub-pixel stage of the registration (ptycho.py:209-235): from the whole-pixel peaks
 * (best = output of ptycho_cg_argmax; only the low word, 0xffffffff - flat index, is read)
 * wrap the peak to [-ndet/2, ndet/2), then evaluate the zoomed matrix DFT of the image
 * product on an ups x ups window around it (ups = ceil(1.5 upsample_factor)),
 *   cross[i,j2,j1] = sum_p sum_k e^{+i th_p (j2 - offy_i)} e^{+i th_k (j1 - offx_i)} ip[i,p,k],
 *   th = 2 pi fftfreq(ndet, upsample_factor),  off = fix(ups/2) - shift upsample_factor,
 * take the first maximum of |cross[i]| and return
 *   shifts[i] = shift + (argmax - fix(ups/2)) / upsample_factor     (float64 [ptheta*nscan][2]).
 * The centred window kernel K[j,k] = exp(+i th_k (j - (ups-1)/2)) is passed as real low-rank
 * factors  K = sum_{r<nc} lz[j,r] vt[k,r] + i sum_{r>=nc} lz[j,r] vt[k,r],
 * vt: float64 [ndet][16], lz: float64 [ups][16] (cos terms first, then sin terms).
 * Needs ndet % 16 == 0, ndet <= 1024, ups <= max(256, ndet). */
int ptycho_cg_zoom(ptycho_handle h, const void* image_product, const void* best, const void* vt,
                   const void* lz, int nc, int ups, double upsample_factor, void* shifts,
                   void* stream);

/* ---- device-resident CG iteration (native sequencing of ptycho.py:325-465) -----------------------
 * The reference decides every line-search trial on the host and runs the object- / probe-sized
 * stages (probe rescale, gradient normalisation, Dai-Yuan direction, updates) as dozens of small
 * array kernels per iteration.  Here they are HIP kernels driven by a float64 STATE vector that
 * lives on the device (owned by the caller, PTYCHO_CG_STATE_WORDS doubles, zero-initialised once),
 * and an iteration is issued as a few stage calls with no device synchronisation.  Between the
 * stages a multi-GPU caller all-reduces (sum) the words / arrays named below; a single-GPU caller
 * just calls them back to back.  One probe mode, gaussian model, ptheta = 1 for the position step.
 *
 *   ptycho_cg_obj_begin   slot 0 <- column pass of fwd(psi, probe); state[A,B] <- statistics (stored, like every sum of
 *                         the native stages: the state needs no zero fill) (ptycho.py:330-343).
 *                                                                    all-reduce: state[PTYCHO_ST_A .. +2)
 *   ptycho_cg_obj_grad    probe *= a/b (:344) and state[MAX_PRB] <- max|probe| in one pass; slot 1 <- projected residual,
 *                         state[COST] <- cost (:347-353); grad <- adj (raw, not yet divided by max|probe|^2; stored, no
 *                         zero fill needed with the deterministic adjoints).   all-reduce: grad
 *                         Option "defer_finish" (single GPU): grad stays in the adjoint's fixed-point image and
 *                         ptycho_cg_obj_dir folds it in -- do not read grad between the two calls; a deterministic
 *                         adjoint (ptycho_adj, ptycho_cg_adj_cols, ptycho_cg_prb_grad) issued in between fails with
 *                         PTYCHO_ERR_ARG instead of adding into the pending image.
 *   ptycho_cg_obj_dir     grad /= max|probe|^2 (:356); Dai-Yuan dpsi, grad0 <- grad (:366-373); slot 1 <- column
 *                         pass of fwd(dpsi, probe); first line-search pass (:383-393).
 *                                                                    all-reduce: state[PTYCHO_ST_COSTS .. +119)
 *                         Option "ls_fused_decide" (single GPU): every pass replays line_search_sqr on its own totals
 *                         in its last workgroup; ptycho_cg_ls_next(pass = 1, 2, 3) then only issues the next pass.
 *   ptycho_cg_ls_next     decide on the pass just reduced (line_search_sqr, :253-281); pass = 1, 2, 3: issue the next
 *                         pass (16, 32, 64 step lengths; each returns at once when the search is already
 *                         resolved) -> all-reduce state[COSTS..] again; pass = 4: decide only.  For a multi-GPU
 *                         caller, who pays a collective per pass: pass = 6 (issue 32) then 7 (issue the 80 that are
 *                         left) then 4 -- three collectives per search --, or pass = 5 (all 112 at once) then 4.  The accepted
 *                         step length times 0.5 lands in state[GAMMA_PSI] (which = 0) / state[GAMMA_PRB] (which = 1).
 *   ptycho_cg_reg_prepare (optional, multi-GPU) slot 2 <- column pass of fwd(psi, ones): the first operand of the position
 *                         correction depends on psi and scan only, so a caller that has to wait for the gradient
 *                         all-reduce anyway issues it in that gap; ptycho_cg_obj_finish then takes correct_positions = 2
 *   ptycho_cg_obj_finish  i > 0: position correction (:398-403; needs the zoom factors of ptycho_cg_zoom), scan[0] += shifts
 *                         (by the kernel that finds them; the next column pass re-sorts the positions);
 *                         psi += gamma dpsi (:405).  correct_positions: 0 off, 1 on, 2 on with slot 2 prepared, 3 on with slots 2 and 3 prepared
 *   ptycho_cg_prb_grad    slot 0 <- column pass of fwd(psi, probe); slot 1 <- projected residual (:421-430);
 *                         gprb <- adj_probe (raw).                   all-reduce: gprb
 *   ptycho_cg_prb_dir     gprb <- gprb / max|psi|^2 / nscan_total * nmodes (:431); Dai-Yuan dprb (:437-448);
 *                         slot 1 <- column pass of fwd(psi, dprb); first line-search pass (:451-461).
 *   ptycho_cg_prb_finish  probe += gamma dprb (:465)
 * state[PTYCHO_ST_LS_FAILED] counts failed line searches ("Line search failed for conjugate gradient."). */
enum {
    PTYCHO_ST_A = 0, PTYCHO_ST_B = 1,           /* sum sqrt(I d), sum I                       */
    PTYCHO_ST_COST = 2, PTYCHO_ST_COST2 = 3,    /* start-of-iteration cost; probe-step scratch */
    PTYCHO_ST_DY_OBJ = 4, PTYCHO_ST_DY_PRB = 7, /* 3 words each: ||g||^2, Re, Im sum conj(d)(g - g0) */
    PTYCHO_ST_MAX_PRB = 10, PTYCHO_ST_MAX_PSI = 11,   /* float bits of max|.| in the low half of the word */
    PTYCHO_ST_ZEROED = 12,                      /* (rounds 1-2: words [0, 12) were cleared by ptycho_cg_obj_begin; now every sum is stored) */
    PTYCHO_ST_GAMMA_PSI = 12, PTYCHO_ST_GAMMA_PRB = 13,
    PTYCHO_ST_LS_GAMMA0 = 14, PTYCHO_ST_LS_NCAND = 15, PTYCHO_ST_LS_NGROUPS = 16, PTYCHO_ST_LS_TRIED = 17,
    PTYCHO_ST_LS_RESOLVED = 18, PTYCHO_ST_LS_FAILED = 19,
    PTYCHO_ST_HINT = 20,                        /* [2] accepted index of the last object / probe search (seed with 14) */
    PTYCHO_ST_COSTS = 24,                       /* 7 groups x (16 step lengths + f(p1)) */
    PTYCHO_CG_STATE_WORDS = 160
};
int ptycho_cg_obj_begin(ptycho_handle h, double* state, const void* psi, const void* scan, const void* prb,
                        const void* data, void* stream);
/* obj_begin2 / obj_dir2: as obj_begin / obj_dir; with ones_prb != NULL the operands of the position correction ride along --
 * slot 2 <- column pass of fwd(psi, 1), slot 3 <- column pass of fwd(dpsi, 1) -- in the same launches, which gather the
 * object patch once for both probes; ptycho_cg_obj_finish then takes correct_positions = 3 (both prepared). */
int ptycho_cg_obj_begin2(ptycho_handle h, double* state, const void* psi, const void* scan, const void* prb,
                         const void* ones_prb, const void* data, void* stream);
int ptycho_cg_obj_dir2(ptycho_handle h, double* state, int first, const void* scan, const void* prb, const void* ones_prb,
                       const void* data, void* grad, void* grad0, void* dpsi, void* stream);
int ptycho_cg_obj_grad(ptycho_handle h, double* state, const void* scan, void* prb, const void* data, void* grad,
                       void* stream);
int ptycho_cg_obj_dir(ptycho_handle h, double* state, int first, const void* scan, const void* prb, const void* data,
                      void* grad, void* grad0, void* dpsi, void* stream);
int ptycho_cg_ls_next(ptycho_handle h, double* state, int which, int pass, const void* data, int use_ab, void* stream);
int ptycho_cg_reg_prepare(ptycho_handle h, double* state, const void* psi, const void* scan, const void* ones_prb, void* stream);
int ptycho_cg_obj_finish(ptycho_handle h, double* state, int correct_positions, void* psi, const void* dpsi, void* scan,
                         const void* ones_prb, const void* vt, const void* lz, int nc, int ups, double upsample_factor,
                         void* stream);
int ptycho_cg_prb_grad(ptycho_handle h, double* state, const void* psi, const void* scan, const void* prb,
                       const void* data, void* gprb, void* stream);
int ptycho_cg_prb_dir(ptycho_handle h, double* state, int first, double nscan_total, double nmodes, const void* psi,
                      const void* scan, const void* data, void* gprb, void* gprb0, void* dprb, void* stream);
int ptycho_cg_prb_finish(ptycho_handle h, double* state, void* prb, const void* dprb, void* stream);

/* Line searches of the multi-mode loop (compact slot layout) on the same device-resident state -- the host enqueues the
 * worst case and never reads a cost back (src/libtike/cufft/ptycho.py:274-276 synchronises once per trial):
 *   ptycho_cg_ls_begin      reset the search state of kind `which` (0 object, 1 probe); first pass sized from the last accepted index
 *   ptycho_cg_ls_obj_chunk  one chunk of an object-search pass: column passes of fwd(dpsi, prbs[k]) for the chunk's positions
 *                           (all modes, shared slot) + the line-search pass over them; chunk 0 stores the costs, the others add
 *                           (:383-393 with the sums over the modes of :386-391).  all-reduce after the last chunk: state[COSTS .. +119)
 *   ptycho_cg_ls_prb_pass   one pass of the probe search of mode `mode` (pair slot A(mode) / shared slot; p1 = inten) (:451-461)
 *   ptycho_cg_ls_decide     line_search_sqr on the costs of the pass (:253-281); next_groups = groups of 16 step lengths the
 *                           next pass prices (0: none follows).  Passes issued after the search is resolved return at once,
 *                           their column passes included.
 *   ptycho_cg_cross_dev     ptycho_cg_cross with gamma read from device memory (the accepted step never visits the host) */
int ptycho_cg_ls_begin(ptycho_handle h, double* state, int which, void* stream);
int ptycho_cg_ls_obj_chunk(ptycho_handle h, double* state, int chunk, const void* dpsi, const void* scan,
                           const void* const* prbs, const void* data, const double* ab, void* stream);
int ptycho_cg_ls_prb_pass(ptycho_handle h, double* state, int mode, const void* data, const void* inten, void* stream);
int ptycho_cg_ls_decide(ptycho_handle h, double* state, int which, int next_groups, void* stream);
int ptycho_cg_cross_dev(ptycho_handle h, int slot1, int slot2, const double* gamma_dev, void* image_product, void* stream);

/* Tuning knobs: "chunk" (positions per launch pair, 0 = default);
 * "window" (1 = LDS overlap-add object adjoint [default], 0 = direct atomics);
 * "trust_order" (1 = the caller vouches that the scan buffer passed to the next calls is the
 * same pointer with unchanged contents as in the previous call, so the position sort is
 * reused; set 0 after modifying scan; default 0);
 * "split" (ndet = 256: 1 = one radix-16 step of the DFT over y runs in the row pass [default]);
 * "tile" (ndet <= 128: 1 = forward operator and probe adjoint as ONE launch each, the tile stays in the CU's LDS and the
 * column<->row intermediate never reaches HBM [default]; 0 = the two-pass kernels of the larger sizes);
 * "deterministic" (1 = the adjoints add their per-workgroup sums into a 64-bit fixed-point image with integer
 * atomics and fold it into the output once: bitwise reproducible for a given chunk / run partition, one extra read of
 * g for the scale (none after ptycho_cg_project, which leaves max |slot| on the device); needs the windowed kernels
 * (powers of two up to 512, or any other size with nprb <= ~1000); default 0: float atomics, as kernels.cu:73-80,92-93);
 * "compact_modes" (M = number of probe modes: compact slot layout + chunk-major position order, see above; 0 = slot pairs);
 * "defer_finish", "ls_fused_decide" (native CG stages on one GPU, see above; default 0);
 * "release_scratch" (any value: frees the adjoint's intermediate, which the fused CG stages never use; ptycho_adj re-allocates it);
 * "release_work" (value = slot: frees that CG work slot; the next stage that writes the slot allocates it again.  The native
 * loop with the position correction's shared gathers holds slots 0-3, without them 0-1 (+2 with a process group));
 * "trust_order" is the CALLER's: the native CG stages track the scan buffer themselves and do not touch it.
 * Experiments build only (make experiments, -DPTYCHO_EXPERIMENTS; the shipped library rejects / ignores them):
 * "fused" (ndet = 256: forward operator as ONE launch that keeps the column<->row intermediate on the CU,
 * k_fwd_fused256: 0 = off, 1 / 2 = one / two class tiles per pass; measured slower) and the
 * environment variables PTYCHO_HIP_CHUNK, PTYCHO_HIP_WINDOW, PTYCHO_HIP_SPLIT, PTYCHO_HIP_FUSED (initial values
 * of the options above), PTYCHO_HIP_NT (nontemporal load/store mask), PTYCHO_HIP_ROWGRID, PTYCHO_HIP_COLSEGS,
 * PTYCHO_HIP_MINSEG (launch geometry), PTYCHO_HIP_NMMAX, PTYCHO_HIP_ZOOM_SCALAR (zoomed DFT without the matrix cores). */
int ptycho_set_option(ptycho_handle h, const char* name, long long value);

/* In-library profiler for bench.py: when enabled, every kernel launch is
 * bracketed by HIP events on the caller's stream.  ptycho_profile_read waits for
 * the recorded launches, returns summed milliseconds and launch counts per kernel
 * (index 0 k_cols<FWD>, 1 k_rows<fwd>, 2 k_rows<inv>, 3 k_cols<ADJ_OBJ>,
 * 4 k_cols<ADJ_PRB>, 5 k_cols<PLAIN>, 6 position sort, 7 / 8 / 9 fused CG row passes (statistics / projection /
 * line search), 10 unused, 11 single-launch forward (experiments build), 12 unused, 13 cross row pass,
 * 14 arg-max column pass, 15 zoomed DFT + arg-max, 16 / 17 one-launch forward / probe adjoint of ndet <= 128; n >= 16:
 * arrays of 16 entries, as earlier versions of this header asked for, simply do not receive ids 16 / 17)
 * and clears the record.
 * No counterpart in the reference (it has no timing code). */
int ptycho_profile(ptycho_handle h, int enable);
int ptycho_profile_read(ptycho_handle h, double* ms, long long* launches, int n);

const char* ptycho_last_error(void);
const char* ptycho_version(void);

#ifdef __cplusplus
}
#endif
#endif
