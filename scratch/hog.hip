// scratch: a kernel that does nothing but hold `blocks` CUs busy for `ticks` of the 100 MHz wall clock - stands in for a
// communicator kernel (RCCL runs one 256-thread workgroup per channel) next to the tile kernel, whose workgroups each need
// a whole CU (16 waves x 128 VGPRs, 160 KB LDS).
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(256) hog(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
extern "C" int hog_launch(int blocks, long long ticks, void* stream) {
    hipLaunchKernelGGL(hog, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), ticks);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
