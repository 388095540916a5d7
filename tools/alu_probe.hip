// alu_probe.hip -- measurement tool: issue rate of float64 add / mul / fma and 32-bit integer add on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/alu_probe.hip -o tools/alu_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_alu(double* out, double seed, int iters) {
    double a[8];
    uint32_t u[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { a[k] = seed + k + threadIdx.x; u[k] = threadIdx.x + k; }
    const double m = 1.0000001, c = 0.5;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(m));
                if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                if (OP == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(7u));
                if (OP == 4) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(c)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(7u)); }
            }
        }
    }
    double s = 0; uint32_t t = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { s += a[k]; t += u[k]; }
    if (s == 1.2345 || t == 0x12345u) out[0] = s;
}

int main() {
    double* out; CK(hipMalloc(&out, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000, blocks = 256 * 8;           // 8 blocks x 4 waves per CU = 8 waves per SIMD
    const char* names[] = {"v_add_f64", "v_mul_f64", "v_fma_f64", "v_add_u32", "v_add_f64+v_add_u32"};
    printf("op,ms,wave_instr_per_ns_chip,cycles_per_wave_instr_per_simd_at_2.4GHz\n");
    for (int op = 0; op < 5; ++op) {
        float ms = 0;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            switch (op) {
                case 0: k_alu<0><<<blocks, 256>>>(out, 1.0, iters); break;
                case 1: k_alu<1><<<blocks, 256>>>(out, 1.0, iters); break;
                case 2: k_alu<2><<<blocks, 256>>>(out, 1.0, iters); break;
                case 3: k_alu<3><<<blocks, 256>>>(out, 1.0, iters); break;
                default: k_alu<4><<<blocks, 256>>>(out, 1.0, iters); break;
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double winstr = (double)blocks * 4 * iters * 64 * (op == 4 ? 2 : 1);     // wave-instructions
        const double per_simd = winstr / 1024.0;
        printf("%s,%.3f,%.2f,%.2f\n", names[op], ms, winstr / (ms * 1e6), ms * 1e-3 * 2.4e9 / per_simd);
    }
    return 0;
}
