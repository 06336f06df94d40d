// Launch cost of (nearly) empty kernels by workgroup size and dynamic LDS size, 256 workgroups (lab for attn_oproj_fold_kernel):
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/wg_probe tools/wg_probe.hip && /tmp/wg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int THREADS>
__global__ __launch_bounds__(THREADS, 1) void probe(float* out, int touch) {
    extern __shared__ float sm[];
    if (touch) sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = touch ? sm[1] : 1.f;
}
template <int THREADS>
static void run(int lds, int grid, float* out) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<THREADS>, dim3(grid), dim3(THREADS), lds, 0, out, 1);
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(probe<THREADS>, dim3(grid), dim3(THREADS), lds, 0, out, 1);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("threads %4d  lds %6d  grid %4d: %.2f us per launch\n", THREADS, lds, grid, ms * 1000 / 200);
}
int main() {
    float* out;
    hipMalloc(&out, 4096 * 4);
    for (int grid : {8, 256, 1024}) {
        run<256>(4096, grid, out); run<256>(105472, grid, out); run<512>(4096, grid, out); run<512>(105472, grid, out);
        run<768>(4096, grid, out); run<768>(105472, grid, out); run<1024>(4096, grid, out); run<1024>(105472, grid, out);
    }
    return 0;
}
