// How many 512-thread workgroups does a CU admit as a function of the dynamic LDS size?  (hipcc --offload-arch=gfx950 -o lds_granule lds_granule.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512, 6) k(float *p) { extern __shared__ float s[]; s[threadIdx.x] = p[threadIdx.x]; __syncthreads(); p[threadIdx.x] = s[511 - threadIdx.x]; }
int main() {
    int last = -1;
    for (int b = 30000; b <= 90000; b += 16) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k, 512, b) != hipSuccess) { printf("query failed at %d\n", b); return 1; }
        if (n != last) { printf("from %6d B of dynamic LDS: %d workgroups per CU\n", b, n); last = n; }
    }
    return 0;
}
