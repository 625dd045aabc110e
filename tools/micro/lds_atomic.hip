// Micro-benchmark: LDS fp64 atomic-add issue cost per 64-lane instruction on gfx950, by active-lane
// pattern and against plain read / write of the same shape.  One work-group per CU, all waves looping.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o tools/micro/lds_atomic tools/micro/lds_atomic.hip
// usage: tools/micro/lds_atomic   (results: profiles/r01_lds_atomic_microbench.txt)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ITERS = 2000;
constexpr int UNROLL = 8;

// MODE 0: ds_add_f64   1: ds_read_b64 (sum)   2: ds_write_b64   3: read+add+write (non-atomic RMW)
// 4: ds_add_f32  5: ds_add_f64 at two planes with one address register (offset immediates)
template <int MODE>
__global__ void __launch_bounds__(1024) k(double *out, long long *cyc, unsigned long long mask, int stride)
{
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += blockDim.x) lds[i] = 0.0;
    __syncthreads();
    const bool on = (mask >> lane) & 1;
    double acc = 0.0;
    const double val = 1.0 + lane;
    // each wave works in its own 8 KB region; lanes contiguous, UNROLL steps `stride` doubles apart
    double *base = lds + wave * 1024 + lane;
    long long t0 = clock64();
    if (on) {
        for (int it = 0; it < ITERS; ++it) {
#pragma unroll
            for (int s = 0; s < UNROLL; ++s) {
                double *p = base + ((s * stride) & 511);
                if (MODE == 0) __hip_atomic_fetch_add(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MODE == 1) acc += *(volatile double *)p;
                if (MODE == 2) *(volatile double *)p = val;
                if (MODE == 3) { double x = *(volatile double *)p; *(volatile double *)p = x + val; }
                if (MODE == 4) __hip_atomic_fetch_add((float *)p, (float)val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
    long long t1 = clock64();
    __syncthreads();
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + tid] = acc + lds[tid];
}

template <int MODE>
void run(const char *name, int waves, unsigned long long mask, int stride)
{
    int ncu = 256;
    double *out; long long *cyc;
    CK(hipMalloc(&out, sizeof(double) * ncu * 1024));
    CK(hipMalloc(&cyc, sizeof(long long) * ncu));
    CK(hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<MODE>, dim3(ncu), dim3(waves * 64), 131072, 0, out, cyc, mask, stride);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<long long> h(ncu);
    CK(hipMemcpy(h.data(), cyc, sizeof(long long) * ncu, hipMemcpyDeviceToHost));
    double avg = 0; for (auto c : h) avg += c; avg /= ncu;
    const double instr = (double)ITERS * UNROLL * waves;  // wave-instructions per CU
    printf("%-28s waves %2d mask %016llx stride %3d : %8.3f ms  clock64 %10.0f  -> %6.2f clk/instr (clock64)  %6.2f ns/instr\n",
           name, waves, mask, stride, ms, avg, avg / instr, ms * 1e6 / instr);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main(int argc, char **argv)
{
    const unsigned long long FULL = ~0ull;
    for (int waves : {4, 8, 16}) {
        run<0>("ds_add_f64 full", waves, FULL, 64);
        run<0>("ds_add_f64 lanes 0-47", waves, (1ull << 48) - 1, 64);
        run<0>("ds_add_f64 lanes 0-32", waves, (1ull << 33) - 1, 64);
        run<0>("ds_add_f64 lanes 0-31", waves, (1ull << 32) - 1, 64);
        run<0>("ds_add_f64 lanes 0-15", waves, (1ull << 16) - 1, 64);
        run<0>("ds_add_f64 even lanes", waves, 0x5555555555555555ull, 64);
        run<0>("ds_add_f64 same addr x8", waves, FULL, 0);
        run<1>("ds_read_b64 full", waves, FULL, 64);
        run<2>("ds_write_b64 full", waves, FULL, 64);
        run<3>("read+write b64 full", waves, FULL, 64);
        run<4>("ds_add_f32 full", waves, FULL, 64);
    }
    return 0;
}
