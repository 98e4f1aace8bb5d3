// Does a workgroup-private write -> read-back round trip stay on chip (L2 / Infinity Cache)?
// Each workgroup owns a region of S bytes, writes it (coalesced 16-B stores), then reads it back,
// R times.  Effective bandwidth = 2 * G * S * R / time.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(f4* buf, size_t region_f4, int reps, float* out, int nt) {
    f4* r = buf + (size_t)blockIdx.x * region_f4;
    f4 acc = {0, 0, 0, 0};
    for (int it = 0; it < reps; ++it) {
        f4 v = {(float)it, 1.f, 2.f, (float)threadIdx.x};
        for (size_t i = threadIdx.x; i < region_f4; i += 256) { if (nt) __builtin_nontemporal_store(v, r + i); else r[i] = v; }
        __threadfence();
        __syncthreads();
        // read back "transposed": thread t reads a different part than it wrote
        for (size_t i = (threadIdx.x * 37 + 11) % 256; i < region_f4; i += 256) { f4 w = nt ? __builtin_nontemporal_load(r + i) : r[i]; acc += w; }
        __syncthreads();
    }
    if (acc.x == 123.456f) out[0] = acc.y;
}
int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256;
    const size_t S = (argc > 2 ? atoi(argv[2]) : 512) * 1024ull;
    const int reps = argc > 3 ? atoi(argv[3]) : 64;
    const int nt = argc > 4 ? atoi(argv[4]) : 0;
    f4* buf; float* out;
    hipMalloc(&buf, G * S); hipMalloc(&out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, buf, S / 16, 2, out, nt);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(G), dim3(256), 0, 0, buf, S / 16, reps, out, nt);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("G=%d S=%zu KiB footprint=%.0f MiB nt=%d: %.3f ms, %.2f TB/s (write+read)\n", G, S / 1024, G * S / 1048576.0, nt, ms,
           2.0 * G * S * reps / (ms * 1e-3) / 1e12);
    return 0;
}
