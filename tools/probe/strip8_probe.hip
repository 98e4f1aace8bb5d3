// strip8_probe.hip -- 64-byte pieces: strips of 8 complex64 columns of a 512 x 512 tile (row stride 4 KiB), as a column
// kernel with C = 8 would touch them, against 128-byte pieces (C = 16).  1024 tiles = 2 GiB.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float c32 __attribute__((ext_vector_type(2)));
template <int C, int WRITE>
__global__ __launch_bounds__(256) void k(c32* __restrict__ buf, int ntiles, int per_wg, float* out) {
    constexpr int N = 512, RPT = 256 / C;   // thread (c, j0): rows j0 + RPT t
    const int tid = threadIdx.x, c = tid % C, j0 = tid / C;
    c32 acc = {0.f, 0.f};
    for (int it = 0; it < per_wg; ++it) {
        const long long item = (long long)blockIdx.x * per_wg + it;
        const int strip = (int)(item / ntiles);
        const long long tile = item % ntiles;
        c32* base = buf + tile * (long long)N * N + strip * C + c;
        constexpr int NR = N / RPT;
        c32 v[NR];
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            c32* p = base + (long long)(j0 + RPT * t) * N;
            if (WRITE) { v[t] = c32{(float)t, (float)c}; *p = v[t]; } else v[t] = __builtin_nontemporal_load(p);
        }
        if (!WRITE) {
#pragma unroll
            for (int t = 0; t < NR; ++t) acc += v[t];
        }
    }
    if (acc.x == 123.456f) out[0] = acc.y;
}
template <int C, int WRITE>
float run(c32* buf, int ntiles, float* out) {
    const int per_wg = 32;
    const int grid = (int)((long long)ntiles * (512 / C) / per_wg);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<C, WRITE>), dim3(grid), dim3(256), 0, 0, buf, ntiles, per_wg, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<C, WRITE>), dim3(grid), dim3(256), 0, 0, buf, ntiles, per_wg, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}
int main() {
    const int ntiles = 1024;
    c32* buf; float* out;
    (void)hipMalloc(&buf, (size_t)ntiles * 512 * 512 * 8); (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 0, (size_t)ntiles * 512 * 512 * 8);
    const double gb = (double)ntiles * 512 * 512 * 8;
    float ms[4] = {run<16, 0>(buf, ntiles, out), run<8, 0>(buf, ntiles, out), run<16, 1>(buf, ntiles, out), run<8, 1>(buf, ntiles, out)};
    const char* nm[4] = {"read  128 B pieces (C = 16)", "read  64 B pieces (C = 8)", "write 128 B pieces (C = 16)", "write 64 B pieces (C = 8)"};
    for (int i = 0; i < 4; ++i) printf("%-30s %.3f ms  %.2f TB/s\n", nm[i], ms[i], gb / (ms[i] * 1e-3) / 1e12);
    return 0;
}
