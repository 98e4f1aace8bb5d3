// slab_probe.hip -- what does the data flow of a SINGLE-LAUNCH 2-D FFT cost on MI355X?
//
// One workgroup owns a tile: it writes the column-pass result to a workgroup-private slab
// (rewritten for every tile, so the live footprint is grid x slab bytes), barriers, reads the
// slab back with a different thread mapping (the transpose) and streams the final tile to g.
// No arithmetic: this measures the memory system only (fabric / Infinity Cache / HBM), i.e. the
// floor of VERDICT r01 item 2's "same-workgroup two-pass" design.
//
//   mode 0  fill:      g tile written only (the algorithmic traffic of the forward operator)
//   mode 1  fwd-like:  slab write -> barrier -> slab read -> g write
//   mode 2  slab only: slab write -> barrier -> slab read
//   mode 3  adj-like:  g read -> slab write -> barrier -> slab read (sink)
//   mode 4  read only: g tile read only (the algorithmic traffic of the adjoint)
// flags: bit0 nontemporal g accesses, bit1 nontemporal slab stores, bit2 nontemporal slab loads, bit3 double-buffered slabs
//
// Build: hipcc --offload-arch=gfx950 -O3 slab_probe.hip -o slab_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

// every copy loop moves U = 8 x 16 B per thread per step, all loads issued before the first store
constexpr int U = 8;
template <int T, bool NTL, bool NTS>
__device__ __forceinline__ void copy_tile(const f4* __restrict__ src, f4* __restrict__ dst, int n_f4, int t0) {
    for (int i = t0; i < n_f4; i += T * U) {
        f4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = NTL ? __builtin_nontemporal_load(src + i + u * T) : src[i + u * T];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NTS) __builtin_nontemporal_store(r[u], dst + i + u * T); else dst[i + u * T] = r[u]; }
    }
}
template <int T, bool NTL>
__device__ __forceinline__ f4 read_tile(const f4* __restrict__ src, int n_f4, int t0) {
    f4 acc = {0, 0, 0, 0};
    for (int i = t0; i < n_f4; i += T * U) {
        f4 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = NTL ? __builtin_nontemporal_load(src + i + u * T) : src[i + u * T];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += r[u];
    }
    return acc;
}
template <int T, bool NTS>
__device__ __forceinline__ void fill_tile(f4* __restrict__ dst, int n_f4, int t0, f4 v) {
    for (int i = t0; i < n_f4; i += T) { if (NTS) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
}

// flags are compile-time here: F bit0 nontemporal g accesses, bit1 nontemporal slab stores, bit2 nontemporal slab loads,
// bit3 two slabs per workgroup (double buffer: ONE barrier per tile, read-back of tile i overlaps the slab write of tile i+1)
template <int T, int MODE, int F>
__global__ __launch_bounds__(T) void k(f4* __restrict__ g, f4* __restrict__ slabs, int tile_f4, int slab_f4, int ntiles, float* out) {
    constexpr bool NTG = F & 1, NTSS = F & 2, NTSL = F & 4, DB = F & 8;
    f4* slab0 = slabs + (size_t)blockIdx.x * slab_f4 * (DB ? 2 : 1);
    const int t = threadIdx.x;
    const int tp = (t * 37 + 11) % T;   // a different thread's lanes: the read-back crosses waves
    f4 acc = {0, 0, 0, 0};
    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        f4* gt = g + (size_t)tile * tile_f4;
        f4* slab = slab0 + (DB ? (size_t)(it & 1) * slab_f4 : 0);
        f4 v = {(float)tile, 1.f, 2.f, (float)t};
        if (MODE == 0) { fill_tile<T, NTG>(gt, tile_f4, t, v); continue; }
        if (MODE == 4) { acc += read_tile<T, NTG>(gt, tile_f4, t); continue; }
        if (MODE == 3) copy_tile<T, NTG, NTSS>(gt, slab, slab_f4, t);
        else fill_tile<T, NTSS>(slab, slab_f4, t, v);
        __syncthreads();
        if (MODE == 1) copy_tile<T, NTSL, NTG>(slab, gt, slab_f4, tp);
        else acc += read_tile<T, NTSL>(slab, slab_f4, tp);
        if (!DB) __syncthreads();
    }
    if (acc.x == 123.456f) out[0] = acc.y;
}

template <int T, int MODE, int F>
float run3(f4* g, f4* slabs, int tile_f4, int slab_f4, int ntiles, float* out, int grid, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<T, MODE, F>), dim3(grid), dim3(T), 0, 0, g, slabs, tile_f4, slab_f4, ntiles, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<T, MODE, F>), dim3(grid), dim3(T), 0, 0, g, slabs, tile_f4, slab_f4, ntiles, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return ms / reps;
}
template <int T, int MODE>
float run2(f4* g, f4* slabs, int tile_f4, int slab_f4, int ntiles, int flags, float* out, int grid, int reps) {
    switch (flags) {
#define C(F) case F: return run3<T, MODE, F>(g, slabs, tile_f4, slab_f4, ntiles, out, grid, reps);
        C(0) C(1) C(3) C(5) C(7) C(9) C(15)
#undef C
    }
    return -1.f;
}
template <int T>
float run(f4* g, f4* slabs, int tile_f4, int slab_f4, int ntiles, int mode, int flags, float* out, int grid, int reps) {
    switch (mode) {
        case 0: return run2<T, 0>(g, slabs, tile_f4, slab_f4, ntiles, flags, out, grid, reps);
        case 1: return run2<T, 1>(g, slabs, tile_f4, slab_f4, ntiles, flags, out, grid, reps);
        case 2: return run2<T, 2>(g, slabs, tile_f4, slab_f4, ntiles, flags, out, grid, reps);
        case 3: return run2<T, 3>(g, slabs, tile_f4, slab_f4, ntiles, flags, out, grid, reps);
        default: return run2<T, 4>(g, slabs, tile_f4, slab_f4, ntiles, flags, out, grid, reps);
    }
}

int main(int argc, char** argv) {
    const int ntiles = argc > 1 ? atoi(argv[1]) : 4096;
    const size_t tile_b = 512 * 1024;
    f4 *g, *slabs; float* out;
    if (hipMalloc(&g, ntiles * tile_b) != hipSuccess) { printf("alloc g failed\n"); return 1; }
    if (hipMalloc(&slabs, (size_t)2048 * tile_b) != hipSuccess) { printf("alloc slabs failed\n"); return 1; }
    (void)hipMalloc(&out, 4);
    (void)hipMemset(g, 0, ntiles * tile_b);
    (void)hipMemset(slabs, 0, (size_t)2048 * tile_b);
    struct Cfg { int T, grid, slab_kib, mode, flags; };
    std::vector<Cfg> cfgs;
    for (int mode : {0, 4}) for (int flags : {0, 1}) cfgs.push_back({1024, 256, 512, mode, flags});
    for (int mode : {0, 4}) for (int flags : {1}) { cfgs.push_back({256, 2048, 512, mode, flags}); cfgs.push_back({512, 512, 512, mode, flags}); }
    for (int mode : {1, 2, 3})
        for (int flags : {0, 1, 3, 5, 7}) {
            if (mode == 2 && (flags & 1)) continue;
            cfgs.push_back({1024, 256, 512, mode, flags});
        }
    // double-buffered slabs (one barrier per tile): 256 MiB live at one workgroup per CU
    for (int mode : {1, 3}) for (int flags : {9, 15}) { cfgs.push_back({1024, 256, 512, mode, flags}); cfgs.push_back({1024, 128, 512, mode, flags}); }
    // two workgroups per CU (256 MiB of slabs) and four (512 MiB): does the Infinity Cache still hold them?
    for (int mode : {1, 2, 3}) { cfgs.push_back({512, 512, 512, mode, 1}); cfgs.push_back({256, 1024, 512, mode, 1}); }
    // fewer workgroups than CUs: 128 / 192 slabs
    for (int mode : {1, 3}) { cfgs.push_back({1024, 128, 512, mode, 1}); cfgs.push_back({1024, 192, 512, mode, 1}); }
    // smaller slabs (what an L2-resident hand-off would see): 64 / 128 KiB, 4 and 2 workgroups per CU
    for (int mode : {1, 2}) { cfgs.push_back({256, 1024, 64, mode, 1}); cfgs.push_back({512, 512, 128, mode, 1}); cfgs.push_back({1024, 256, 128, mode, 1}); }
    printf("# ntiles=%d tile=512KiB; ms per pass over all tiles; TB/s counts g bytes only (algorithmic)\n", ntiles);
    for (auto& c : cfgs) {
        const int slab_f4 = c.slab_kib * 1024 / 16;
        const int tile_f4 = (c.mode == 0 || c.mode == 4) ? (int)(tile_b / 16) : slab_f4;
        // with slabs smaller than a tile, process proportionally more "tiles" so the g bytes stay equal
        const int nt = (int)((size_t)ntiles * (tile_b / 16) / tile_f4);
        float ms = 0;
        if (c.T == 1024) ms = run<1024>(g, slabs, tile_f4, slab_f4, nt, c.mode, c.flags, out, c.grid, 5);
        else if (c.T == 512) ms = run<512>(g, slabs, tile_f4, slab_f4, nt, c.mode, c.flags, out, c.grid, 5);
        else ms = run<256>(g, slabs, tile_f4, slab_f4, nt, c.mode, c.flags, out, c.grid, 5);
        const double gbytes = (double)ntiles * tile_b;
        printf("T=%4d grid=%4d slab=%3dKiB live=%4.0fMiB mode=%d flags=%d : %.3f ms  (%.2f TB/s algorithmic)\n", c.T, c.grid, c.slab_kib,
               c.grid * c.slab_kib / 1024.0 * ((c.flags & 8) ? 2 : 1), c.mode, c.flags, ms, gbytes / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
    return 0;
}
