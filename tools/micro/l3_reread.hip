// Microbenchmark (dev aid, not part of the library): how fast is the SECOND read of a buffer of S bytes when the first read went just
// before it -- i.e. what the Infinity Cache gives a two-pass algorithm at sub-chunk size S.  Plain and nontemporal loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <bool NT>
__global__ __launch_bounds__(256) void k_read(const uint4* p, size_t n, uint32_t* out) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint4 v;
        if (NT) { v.x = __builtin_nontemporal_load(&p[i].x); v.y = __builtin_nontemporal_load(&p[i].y); v.z = __builtin_nontemporal_load(&p[i].z); v.w = __builtin_nontemporal_load(&p[i].w); }
        else v = p[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t maxb = (size_t)2 << 30;
    uint4* buf; uint32_t* out;
    CK(hipMalloc(&buf, maxb)); CK(hipMalloc(&out, 4)); CK(hipMemset(buf, 1, maxb));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t sizes[] = {32u << 20, 64u << 20, 96u << 20, 128u << 20, 160u << 20, 192u << 20, 256u << 20, 512u << 20, (size_t)2 << 30};
    for (int nt1 = 0; nt1 < 2; ++nt1) for (int nt2 = 0; nt2 < 2; ++nt2) for (size_t S : sizes) {
        const size_t n = S / 16;
        float best1 = 1e9f, best2 = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            // flush: read a different 1 GiB region (the upper half) -- only meaningful for S <= 1 GiB
            if (S <= ((size_t)1 << 30)) { k_read<false><<<4096, 256>>>(buf + (maxb / 32), (maxb / 32), out); }
            CK(hipEventRecord(a));
            if (nt1) k_read<true><<<4096, 256>>>(buf, n, out); else k_read<false><<<4096, 256>>>(buf, n, out);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms1; CK(hipEventElapsedTime(&ms1, a, b));
            CK(hipEventRecord(a));
            if (nt2) k_read<true><<<4096, 256>>>(buf, n, out); else k_read<false><<<4096, 256>>>(buf, n, out);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms2; CK(hipEventElapsedTime(&ms2, a, b));
            if (ms1 < best1) best1 = ms1;
            if (ms2 < best2) best2 = ms2;
        }
        printf("first %s second %s  S=%5zu MiB  first %.1f us %.2f TB/s   second %.1f us %.2f TB/s\n", nt1 ? "nt   " : "plain", nt2 ? "nt   " : "plain", S >> 20, best1 * 1e3, S / best1 * 1e-9, best2 * 1e3, S / best2 * 1e-9);
        fflush(stdout);
    }
    return 0;
}
