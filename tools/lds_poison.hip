// Development aid: fill every CU's LDS with a chosen 64-bit pattern, so that a kernel that depends on LDS contents it
// never wrote gives different results for different patterns (tools/lds_poison_probe.py).
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/lds_poison.hip -o tools/_build/liblds_poison.so
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void poison_kernel(uint64_t pattern, int words) {
    extern __shared__ uint64_t buf[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) buf[i] = pattern;
    __syncthreads();
    if (buf[(threadIdx.x * 7) % words] != pattern) __builtin_trap();
}

extern "C" int lds_poison(uint64_t pattern) {
    const int bytes = 160 * 1024;
    if (hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return 1;
    hipLaunchKernelGGL(poison_kernel, dim3(4096), dim3(256), bytes, 0, pattern, bytes / 8);
    return hipDeviceSynchronize() == hipSuccess ? 0 : 2;
}
