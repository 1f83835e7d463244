// lds_poison.hip - TEST INFRASTRUCTURE (tests/test_gpu_boundary.py): fill every CU's LDS with a NaN bit pattern.
//
// LDS is not cleared between workgroups, so whatever a CU's previous workgroup left there is what the next one finds in
// the cells it does not write.  A kernel that READS a cell it has not written (and, say, multiplies it by a zero weight)
// works until the leftover happens to look like a NaN - a rare, box- and schedule-dependent failure (round 3:
// trace_bwd_item_lean with an empty window).  This kernel makes the leftover a NaN everywhere, on purpose.
// Build: hipcc --offload-arch=gfx950 -shared -fPIC -o tests/bin/liblds_poison.so tests/lds_poison.hip   (__graft_entry__.build)
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(1024) void lds_poison_kernel(unsigned pattern, int cells, unsigned* sink)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < cells; i += blockDim.x) lds[i] = pattern;
    __syncthreads();
    if (threadIdx.x == 0 && lds[(blockIdx.x * 7919) % cells] != pattern) atomicAdd(sink, 1u);   // (keeps the stores alive)
}

// 160 KB of LDS per workgroup = one workgroup per CU; four workgroups per CU's worth of grid so that every CU gets one.
extern "C" int lds_poison(unsigned pattern, void* stream_, unsigned* sink)
{
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int bytes = 160 * 1024 - 64;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess)
        return -1;
    hipLaunchKernelGGL(lds_poison_kernel, dim3((unsigned)(4 * cus)), dim3(1024), bytes, stream, pattern, bytes / 4, sink);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
