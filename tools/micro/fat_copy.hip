// A stand-in for a collective's kernel in tools/reserve_cus_probe.py: a few work-groups that stream memory and - like
// RCCL's kernels, unlike a torch copy - occupy many vector registers per lane (~110) and some LDS, so that they do NOT fit
// into what the persistent tile kernel leaves free on a CU it occupies (its work-group takes 448 of a SIMD's 512
// VGPRs and 134 of 160 KB of LDS).  Whether such a kernel starts beside the tile kernel therefore depends on CUs
// being left free ("reserve_cus").
// build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libfatcopy.so tools/micro/fat_copy.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int NREG = 24;  // double2 values a lane holds at once: ~110 VGPRs, more than the 64 a SIMD running the tile kernel has left

// stamps[4 * b .. 4 * b + 3] (optional): work-group b's start and end on the 100 MHz real-time counter, its XCC_ID and
// its HW_ID (which CU it ran on) - the probe counts how many work-groups started while the tile kernel was running
__global__ void __launch_bounds__(256) fat_copy_kernel(const double2 *__restrict__ src, double2 *__restrict__ dst, int64_t n,
                                                       long long *__restrict__ stamps, int reps)
{
    __shared__ double pad[4096];  // 32 KB: more than a CU running the tile kernel has left
    if (threadIdx.x == 0) pad[blockIdx.x & 4095] = 0.0;
    if (stamps && threadIdx.x == 0) {
        unsigned xcc = 0, hw = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        stamps[4 * blockIdx.x] = (long long)__builtin_amdgcn_s_memrealtime();
        stamps[4 * blockIdx.x + 2] = (long long)xcc;
        stamps[4 * blockIdx.x + 3] = (long long)hw;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int rep = 0; rep < reps; ++rep)  // one launch that stays for as long as a collective's kernel does
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; base < n; base += stride * NREG) {
        double2 r[NREG];
#pragma unroll
        for (int q = 0; q < NREG; ++q) {
            const int64_t i = base + q * stride;
            r[q] = i < n ? src[i] : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < NREG; ++q) asm volatile("" : "+v"(r[q].x), "+v"(r[q].y));  // all of them live at once
#pragma unroll
        for (int q = 0; q < NREG; ++q) {
            const int64_t i = base + q * stride;
            if (i < n) dst[i] = r[q];
        }
    }
    if (threadIdx.x == 0 && pad[blockIdx.x & 4095] != 0.0) dst[0].x = 1.0;
    if (stamps && threadIdx.x == 0) stamps[4 * blockIdx.x + 1] = (long long)__builtin_amdgcn_s_memrealtime();
}

extern "C" int fat_copy(void *dst, const void *src, int64_t n_double2, int blocks, void *stream, void *stamps)
{
    hipLaunchKernelGGL(fat_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double2 *)src, (double2 *)dst,
                       n_double2, (long long *)stamps, 1);
    return (int)hipGetLastError();
}

// the same in ONE launch that copies `reps` times - a collective is one long-lived kernel, not a train of short ones
extern "C" int fat_copy_rep(void *dst, const void *src, int64_t n_double2, int blocks, int reps, void *stream, void *stamps)
{
    hipLaunchKernelGGL(fat_copy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double2 *)src, (double2 *)dst,
                       n_double2, (long long *)stamps, reps);
    return (int)hipGetLastError();
}
