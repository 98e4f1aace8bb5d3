// lds_atomic_probe.hip -- cost of LDS read-modify-write forms per wave-instruction on gfx950: no-return
// ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_pk_add_f16 against a plain ds_read_b32 + add + ds_write_b32, for a
// conflict-free pattern (lane l -> word l) and for the adjoint window's pattern (16 columns x 4 rows, pitch 20).
// 1024 workgroups of 256 threads, 3 per CU resident; cycles = time * 2.4 GHz / (instructions per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
enum { F32 = 0, U32 = 1, U64 = 2, PKF16 = 3, RMW = 4 };
template <int KIND, int PAT>
__global__ __launch_bounds__(256) void k(int iters, float* out) {
    __shared__ float w[12288];          // 48 KiB
    const int tid = threadIdx.x;
    for (int i = tid; i < 12288; i += 256) w[i] = 0.0f;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    int idx = PAT == 0 ? lane : ((lane >> 4) * 20 + (lane & 15));
    idx += wave * 2048;
    if (KIND == U64) idx *= 2;
    float* p = w + idx;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            float* q = p + u * (KIND == U64 ? 160 : 80);
            if (KIND == F32) __hip_atomic_fetch_add(q, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (KIND == U32) __hip_atomic_fetch_add((unsigned*)q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (KIND == U64) __hip_atomic_fetch_add((unsigned long long*)q, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (KIND == PKF16) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2*)q, h2{(_Float16)1, (_Float16)1});
            }
            if (KIND == RMW) { *q = *q + 1.0f; }
        }
    }
    __syncthreads();
    if (w[tid] == 123.456f) out[0] = w[tid + 1];
}
template <int KIND, int PAT>
void run(const char* name, float* out) {
    const int iters = 2000, grid = 1024;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<KIND, PAT>), dim3(grid), dim3(256), 0, 0, 10, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<KIND, PAT>), dim3(grid), dim3(256), 0, 0, iters, out);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    // wave-instructions per CU: grid/256 workgroups per CU x 4 waves x iters x 16
    const double per_cu = (double)grid / 256 * 4 * iters * 16;
    printf("%-44s %8.3f ms  %6.1f LDS cycles per wave-instruction (CU-wide)\n", name, ms, ms * 1e-3 * 2.4e9 / per_cu);
}
int main() {
    float* out; (void)hipMalloc(&out, 4);
    run<F32, 0>("ds_add_f32, lane -> word", out);
    run<F32, 1>("ds_add_f32, 4 rows x 16 columns, pitch 20", out);
    run<U32, 0>("ds_add_u32, lane -> word", out);
    run<U32, 1>("ds_add_u32, 4 rows x 16 columns, pitch 20", out);
    run<U64, 0>("ds_add_u64, lane -> 8 bytes", out);
    run<U64, 1>("ds_add_u64, 4 rows x 16 columns, pitch 20", out);
    run<PKF16, 0>("ds_pk_add_f16, lane -> word", out);
    run<RMW, 0>("ds_read_b32 + add + ds_write_b32 (dependent)", out);
    run<RMW, 1>("same, 4 rows x 16 columns, pitch 20", out);
    return 0;
}
