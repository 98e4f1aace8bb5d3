// Probe: which XCD does each workgroup land on, and how many are co-resident per CU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k_probe(int* xcc, int* cu, unsigned long long* t0, int lds_kb) {
    extern __shared__ char smem[];
    if (threadIdx.x == 0) {
        // HW_REG_XCC_ID = 20, field [3:0]; HW_REG_HW_ID = 4 (cu id bits [11:8], se id [15:13]...)
        xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);
        cu[blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
        t0[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        smem[0] = 1;
    }
    // keep the block alive ~50 us so that co-residency shows in the start times
    unsigned long long t = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t < 5000) { __builtin_amdgcn_s_sleep(10); }
}
int main() {
    int ngrid = 512;
    int *xcc, *cu; unsigned long long* t0;
    hipMalloc(&xcc, ngrid * 4); hipMalloc(&cu, ngrid * 4); hipMalloc(&t0, ngrid * 8);
    hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipLaunchKernelGGL(k_probe, dim3(ngrid), dim3(256), 80 * 1024, 0, xcc, cu, t0, 80);
    hipDeviceSynchronize();
    std::vector<int> hx(ngrid), hc(ngrid); std::vector<unsigned long long> ht(ngrid);
    hipMemcpy(hx.data(), xcc, ngrid * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), cu, ngrid * 4, hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), t0, ngrid * 8, hipMemcpyDeviceToHost);
    int cnt[16] = {0};
    unsigned long long tmin = ~0ull, tmax = 0;
    for (int i = 0; i < ngrid; ++i) { cnt[hx[i] & 15]++; if (ht[i] < tmin) tmin = ht[i]; if (ht[i] > tmax) tmax = ht[i]; }
    printf("xcc of blocks 0..23:"); for (int i = 0; i < 24; ++i) printf(" %d", hx[i]); printf("\n");
    printf("blocks per xcc:"); for (int i = 0; i < 8; ++i) printf(" %d", cnt[i]); printf("\n");
    printf("start spread (100MHz ticks): %llu\n", tmax - tmin);
    printf("hw_id of blocks 0..7:"); for (int i = 0; i < 8; ++i) printf(" %08x", hc[i]); printf("\n");
    hipError_t e = hipGetLastError(); printf("err=%s\n", hipGetErrorString(e));
    return 0;
}
