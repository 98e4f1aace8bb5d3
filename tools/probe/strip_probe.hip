// strip_probe.hip -- does the column kernels' access pattern (128-byte pieces at a 2-KiB stride) cost HBM bandwidth?
// Reads / writes 2 GiB (4096 tiles of 256 x 256 complex64) the way k_cols_* do: a workgroup of 256 threads owns one
// 16-column strip of one tile (256 rows x 128 B, row stride 2 KiB), thread (c = tid % 16, j0 = tid / 16) touches rows
// j0 + 16 t, 8 B each -- against the same bytes laid out strip-major (32 KiB contiguous per workgroup, same thread map).
// Build: hipcc --offload-arch=gfx950 -O3 strip_probe.hip -o strip_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float c32 __attribute__((ext_vector_type(2)));
template <int MODE>   // 0 read strided, 1 read contiguous, 2 write strided, 3 write contiguous
__global__ __launch_bounds__(256) void k(c32* __restrict__ buf, int ntiles, int per_wg, float* out) {
    const int tid = threadIdx.x, c = tid % 16, j0 = tid / 16;
    c32 acc = {0.f, 0.f};
    for (int it = 0; it < per_wg; ++it) {
        const long long item = (long long)blockIdx.x * per_wg + it;   // (tile, strip), a run of consecutive tiles per strip like the sorted runs
        const int strip = (int)(item / ((long long)ntiles)) ;
        const long long tile = item % ntiles;
        c32* base = buf + tile * 65536;
        c32 v[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int row = j0 + 16 * t;
            c32* p = (MODE & 1) ? base + strip * 4096 + row * 16 + c : base + row * 256 + strip * 16 + c;
            if (MODE < 2) v[t] = __builtin_nontemporal_load(p);
            else { v[t] = c32{(float)row, (float)c}; *p = v[t]; }
        }
        if (MODE < 2) {
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += v[t];
        }
    }
    if (acc.x == 123.456f) out[0] = acc.y;
}
template <int MODE>
float run(c32* buf, int ntiles, float* out) {
    const int per_wg = 64;
    const int grid = (int)((long long)ntiles * 16 / per_wg);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, buf, ntiles, per_wg, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, buf, ntiles, per_wg, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main() {
    const int ntiles = 4096;
    c32* buf; float* out;
    (void)hipMalloc(&buf, (size_t)ntiles * 65536 * 8); (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 0, (size_t)ntiles * 65536 * 8);
    const double gb = (double)ntiles * 65536 * 8;
    const char* names[4] = {"read  128 B pieces, 2 KiB stride (column kernels today)", "read  strip-major, 32 KiB contiguous per workgroup",
                            "write 128 B pieces, 2 KiB stride (column kernels today)", "write strip-major, 32 KiB contiguous per workgroup"};
    float ms[4] = {run<0>(buf, ntiles, out), run<1>(buf, ntiles, out), run<2>(buf, ntiles, out), run<3>(buf, ntiles, out)};
    for (int i = 0; i < 4; ++i) printf("%-60s %.3f ms  %.2f TB/s\n", names[i], ms[i], gb / (ms[i] * 1e-3) / 1e12);
    return 0;
}
