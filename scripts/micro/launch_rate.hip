// Micro-benchmark: how fast can the device start single-wave workgroups?  (is k_step dispatch-bound?)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k_empty(unsigned* out) { if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFFFu) out[0] = 1; }
__global__ __launch_bounds__(64, 8) void k_lds(unsigned* out) {
    __shared__ unsigned buf[1088];  // ~4.3 KB like the step kernel
    buf[threadIdx.x] = blockIdx.x;
    __syncthreads();
    if (buf[(threadIdx.x + 1) & 63] == 0xFFFFFFFFu) out[0] = 1;
}
__global__ __launch_bounds__(256) void k_empty256(unsigned* out) { if (threadIdx.x == 0 && blockIdx.x == 0xFFFFFFFFu) out[0] = 1; }
template <typename F> float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.f / reps;
}
int main() {
    unsigned* d; hipMalloc(&d, 4);
    for (int n : {8192, 65536, 262144}) {
        float e = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(n), dim3(64), 0, 0, d); }, 200);
        float l = timeit([&] { hipLaunchKernelGGL(k_lds, dim3(n), dim3(64), 0, 0, d); }, 200);
        float q = timeit([&] { hipLaunchKernelGGL(k_empty256, dim3(n / 4), dim3(256), 0, 0, d); }, 200);
        printf("%7d single-wave blocks: empty %.1f us, with 4.3 KB LDS %.1f us; %d four-wave blocks: %.1f us\n", n, e, l, n / 4, q);
    }
    return 0;
}
