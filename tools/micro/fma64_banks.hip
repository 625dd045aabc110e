// Micro-benchmark: issue cost of v_fma_f64 on gfx950 by where its three 64-bit VGPR operands sit (register index mod 4),
// one wave per SIMD, 16 independent accumulators (no dependency stalls).  The aw kernel builder's fp64 FMAs take
// their operands from double2 values (LDS b128 reads), which the register allocator aligns to 4 VGPRs: a.x, b.x and
// acc.x then all start at index = 0 (mod 4).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/fma64_banks tools/micro/fma64_banks.hip ; run: tools/micro/fma64_banks
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ITERS = 4096;

// 16 FMAs per trip: acc_k (v[2k : 2k+1], k = 0..15 -> registers 0..31) += A * B with A, B at chosen registers.
// MODE 0: A = v[32:33], B = v[36:37]   (both 0 mod 4; accumulators alternate 0 / 2 mod 4)
// MODE 1: A = v[32:33], B = v[38:39]   (A 0 mod 4, B 2 mod 4)
// MODE 2: A = v[34:35], B = v[38:39]   (both 2 mod 4)
// MODE 3: as 0 but only the accumulators at 0 mod 4 (acc, A, B all 0 mod 4: what double2 .x * .x -> .x does)
// MODE 4: as 1 but accumulators at 0 mod 4 only (A 0, B 2, acc 0: what .x * .y -> ... does)
// (64-bit operands must start at an even register on gfx950: 0 or 2 mod 4 are the only cases)
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, long long *cyc)
{
    long long t0 = 0, t1 = 0;
    asm volatile(
        "v_mov_b32 v32, 0\n v_mov_b32 v33, 0x3ff00000\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0x3ff00000\n"
        "v_mov_b32 v36, 0\n v_mov_b32 v37, 0x3ff00000\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0x3ff00000\n"
        ::: "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39");
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < ITERS; ++it) {
#define F(d, a, b) "v_fma_f64 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1], v[" #d ":" #d "+1]\n"
        if (MODE == 0)
            asm volatile(F(0, 32, 36) F(2, 32, 36) F(4, 32, 36) F(6, 32, 36) F(8, 32, 36) F(10, 32, 36) F(12, 32, 36) F(14, 32, 36)
                         F(16, 32, 36) F(18, 32, 36) F(20, 32, 36) F(22, 32, 36) F(24, 32, 36) F(26, 32, 36) F(28, 32, 36) F(30, 32, 36) ::: "memory");
        if (MODE == 1)
            asm volatile(F(0, 32, 38) F(2, 32, 38) F(4, 32, 38) F(6, 32, 38) F(8, 32, 38) F(10, 32, 38) F(12, 32, 38) F(14, 32, 38)
                         F(16, 32, 38) F(18, 32, 38) F(20, 32, 38) F(22, 32, 38) F(24, 32, 38) F(26, 32, 38) F(28, 32, 38) F(30, 32, 38) ::: "memory");
        if (MODE == 2)
            asm volatile(F(0, 34, 38) F(2, 34, 38) F(4, 34, 38) F(6, 34, 38) F(8, 34, 38) F(10, 34, 38) F(12, 34, 38) F(14, 34, 38)
                         F(16, 34, 38) F(18, 34, 38) F(20, 34, 38) F(22, 34, 38) F(24, 34, 38) F(26, 34, 38) F(28, 34, 38) F(30, 34, 38) ::: "memory");
        if (MODE == 3)
            asm volatile(F(0, 32, 36) F(4, 32, 36) F(8, 32, 36) F(12, 32, 36) F(16, 32, 36) F(20, 32, 36) F(24, 32, 36) F(28, 32, 36)
                         F(0, 32, 36) F(4, 32, 36) F(8, 32, 36) F(12, 32, 36) F(16, 32, 36) F(20, 32, 36) F(24, 32, 36) F(28, 32, 36) ::: "memory");
        if (MODE == 4)
            asm volatile(F(0, 32, 38) F(4, 32, 38) F(8, 32, 38) F(12, 32, 38) F(16, 32, 38) F(20, 32, 38) F(24, 32, 38) F(28, 32, 38)
                         F(0, 32, 38) F(4, 32, 38) F(8, 32, 38) F(12, 32, 38) F(16, 32, 38) F(20, 32, 38) F(24, 32, 38) F(28, 32, 38) ::: "memory");
#define G(d, a, b) "v_fmac_f64 v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1]\n"
#define H(d, a, b, n) "v_fmac_f64_dpp v[" #d ":" #d "+1], v[" #a ":" #a "+1], v[" #b ":" #b "+1] row_newbcast:" #n " row_mask:0xf bank_mask:0xf\n"
        if (MODE == 5)  // the VOP2 form the compiler emits for most of the builder's products
            asm volatile(G(0, 32, 36) G(2, 32, 36) G(4, 32, 36) G(6, 32, 36) G(8, 32, 36) G(10, 32, 36) G(12, 32, 36) G(14, 32, 36)
                         G(16, 32, 36) G(18, 32, 36) G(20, 32, 36) G(22, 32, 36) G(24, 32, 36) G(26, 32, 36) G(28, 32, 36) G(30, 32, 36) ::: "memory");
        if (MODE == 6)  // the same with the first factor taken from lane n of each row of 16 (DP ALU DPP: row_newbcast only)
            asm volatile(H(0, 32, 36, 0) H(2, 32, 36, 1) H(4, 32, 36, 2) H(6, 32, 36, 3) H(8, 32, 36, 4) H(10, 32, 36, 5) H(12, 32, 36, 6) H(14, 32, 36, 7)
                         H(16, 32, 36, 8) H(18, 32, 36, 9) H(20, 32, 36, 10) H(22, 32, 36, 11) H(24, 32, 36, 12) H(26, 32, 36, 13) H(28, 32, 36, 14) H(30, 32, 36, 15) ::: "memory");
    }
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (out) out[0] = 0.0;
}

int main()
{
    long long *cyc;
    CK(hipMalloc(&cyc, 8));
    const char *names[] = {"A 0 mod 4, B 0 mod 4, acc alternating 0 / 2", "A 0, B 2, acc alternating", "A 2, B 2, acc alternating",
                           "A 0, B 0, acc 0 (double2 .x * .x -> .x)", "A 0, B 2, acc 0",
                           "v_fmac_f64 (VOP2), operands as mode 0", "v_fmac_f64_dpp row_newbcast:n, operands as mode 0"};
    for (int m = 0; m < 7; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (m) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, nullptr, cyc); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, nullptr, cyc); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), 0, 0, nullptr, cyc); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(256), dim3(256), 0, 0, nullptr, cyc); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, nullptr, cyc); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(256), dim3(256), 0, 0, nullptr, cyc); break;
                case 6: hipLaunchKernelGGL(k<6>, dim3(256), dim3(256), 0, 0, nullptr, cyc); break;
            }
            CK(hipDeviceSynchronize());
        }
        long long h = 0;
        CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
        printf("mode %d  %-46s  %.3f cycles per instruction (one wave per SIMD, 256 work-groups)\n", m, names[m], (double)h / (ITERS * 16.0));
    }
    return 0;
}
