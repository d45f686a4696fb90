// How many single-wave blocks does a CU admit for a given LDS size per block? (allocation granularity of gfx950)
// build: hipcc --offload-arch=gfx950 -O2 scripts/micro/lds_occupancy.hip -o scripts/micro/lds_occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(int* out) {
    extern __shared__ int s[];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (out) out[blockIdx.x * 64 + threadIdx.x] = s[63 - threadIdx.x];
}
int main() {
    for (int bytes = 5120; bytes <= 8192; bytes += 128) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 64, bytes) != hipSuccess) return 1;
        printf("%d B per block: %d blocks per CU\n", bytes, n);
    }
    return 0;
}
