// mall_probe.hip -- measurement tool, not part of the library.
// How fast can a streaming read go when the same bytes are read again shortly afterwards (Infinity Cache reuse)?
//   1. one buffer of S bytes read 20 times in a row, S from 32 MiB to 4 GiB;
//   2. a 2.25 GiB buffer cut into groups of S bytes, every group read 3 times back to back (the min/max, histogram and
//      mask passes of the LFA front end) before the next group is touched.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o tools/mall_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const uint4* p, size_t n16, unsigned int* sink) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    unsigned int acc = 0;
    for (; i + 3 * step < n16; i += 4 * step) {
        const uint4 a = p[i], b = p[i + step], c = p[i + 2 * step], d = p[i + 3 * step];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += step) { const uint4 a = p[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the histogram pass's traffic shape: 48 bytes in, 16 bytes out per lane and step
__global__ __launch_bounds__(256) void k_read3_write1(const uint4* p, uint4* q, size_t ngroups) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; g < ngroups; g += step) {
        const uint4 a = p[3 * g], b = p[3 * g + 1], c = p[3 * g + 2];
        uint4 o;
        o.x = a.x ^ b.x ^ c.x; o.y = a.y ^ b.y ^ c.y; o.z = a.z ^ b.z ^ c.z; o.w = a.w ^ b.w ^ c.w;
        q[g] = o;
    }
}

// the Gaussian stage's traffic shape: 1 byte in, 8 bytes out per element (u8 -> f64), and write-only
__global__ __launch_bounds__(256) void k_read1_write8(const uint32_t* p, double* q, size_t n4) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    // lane handles 4 consecutive bytes per step but writes them as 4 row-strided stores so that a wave's store is contiguous
    for (size_t base = (size_t)blockIdx.x * 1024; base < n4 * 4; base += (size_t)gridDim.x * 1024) {
        const uint8_t* pb = (const uint8_t*)p + base;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t e = base + k * 256 + threadIdx.x;
            q[e] = (double)pb[k * 256 + threadIdx.x] * (1.0 / 255.0);
        }
    }
    (void)g; (void)step;
}
__global__ __launch_bounds__(256) void k_write8(double* q, size_t n) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; g < n; g += step) q[g] = (double)g;
}
__global__ __launch_bounds__(256) void k_write16(double2* q, size_t n) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; g < n; g += step) q[g] = make_double2((double)g, 1.0);
}

static int grid_for(size_t n16) {
    size_t blocks = (n16 + 256 * 8 - 1) / (256 * 8);      // about 8 x 16 bytes per lane
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

int main() {
    const size_t total = (size_t)2304 << 20;
    const size_t big = (size_t)4096 << 20;
    uint8_t* buf; unsigned int* sink;
    CK(hipMalloc(&buf, big)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, big));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s; CK(hipStreamCreate(&s));
    printf("test,bytes_MiB,ms,GBps\n");
    const size_t sizes[] = {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 2304, 4096};
    for (size_t si = 0; si < sizeof(sizes) / sizeof(sizes[0]); ++si) {
        const size_t S = sizes[si] << 20, n16 = S / 16;
        const int g = grid_for(n16);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, s, (const uint4*)buf, n16, sink);
        CK(hipEventRecord(e0, s));
        const int reps = 20;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, s, (const uint4*)buf, n16, sink);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("repeat,%zu,%.4f,%.1f\n", sizes[si], ms / reps, (double)S * reps / (ms * 1e6));
    }
    const size_t groups[] = {16, 32, 64, 96, 128, 160, 192, 256, 576, 2304};
    for (size_t gi = 0; gi < sizeof(groups) / sizeof(groups[0]); ++gi) {
        const size_t S = groups[gi] << 20, n16 = S / 16;
        const int g = grid_for(n16);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (size_t off = 0; off + S <= total; off += S)
                for (int pass = 0; pass < 3; ++pass)
                    hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, s, (const uint4*)(buf + off), n16, sink);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const size_t covered = (total / S) * S;
        printf("3pass_groups,%zu,%.4f,%.1f\n", groups[gi], best, (double)covered * 3 / (best * 1e6));
    }
    {
        const size_t in_bytes = (size_t)2304 << 20, ngroups = in_bytes / 48;       // 2.25 GiB in, 0.75 GiB out
        uint4* q = (uint4*)(buf + in_bytes);
        for (int blocks : {1024, 2048, 4096, 8192}) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_read3_write1, dim3(blocks), dim3(256), 0, s, (const uint4*)buf, q, ngroups);
            CK(hipEventRecord(e0, s));
            const int reps = 10;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read3_write1, dim3(blocks), dim3(256), 0, s, (const uint4*)buf, q, ngroups);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("read48_write16_blocks%d,%zu,%.4f,%.1f\n", blocks, (in_bytes + ngroups * 16) >> 20, ms / reps,
                   (double)(in_bytes + ngroups * 16) * reps / (ms * 1e6));
        }
    }
    {
        const size_t n = (size_t)2048 << 20;                         // hipMemcpy device to device, 2 GiB (read + write counted)
        for (int w = 0; w < 2; ++w) CK(hipMemcpyAsync(buf + n, buf, n, hipMemcpyDeviceToDevice, s));
        CK(hipEventRecord(e0, s));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(buf + n, buf, n, hipMemcpyDeviceToDevice, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemcpyD2D_read_plus_write,%zu,%.4f,%.1f\n", (2 * n) >> 20, ms / reps, (double)2 * n * reps / (ms * 1e6));
    }
    {
        const size_t n_in = (size_t)384 << 20;                       // 384 MiB of bytes in, 3 GiB of doubles out
        double* q = (double*)(buf + ((size_t)512 << 20));
        for (int blocks : {2048, 4096, 8192, 16384}) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_read1_write8, dim3(blocks), dim3(256), 0, s, (const uint32_t*)buf, q, n_in / 4);
            CK(hipEventRecord(e0, s));
            const int reps = 10;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read1_write8, dim3(blocks), dim3(256), 0, s, (const uint32_t*)buf, q, n_in / 4);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("read1_write8_blocks%d,%zu,%.4f,%.1f\n", blocks, (n_in * 9) >> 20, ms / reps, (double)n_in * 9 * reps / (ms * 1e6));
        }
        const size_t n = ((size_t)3072 << 20) / 8;
        for (int blocks : {2048, 8192}) {
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_write8, dim3(blocks), dim3(256), 0, s, q, n);
            CK(hipEventRecord(e0, s));
            const int reps = 10;
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_write8, dim3(blocks), dim3(256), 0, s, q, n);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("write8_blocks%d,%zu,%.4f,%.1f\n", blocks, (n * 8) >> 20, ms / reps, (double)n * 8 * reps / (ms * 1e6));
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_write16, dim3(blocks), dim3(256), 0, s, (double2*)q, n / 2);
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_write16, dim3(blocks), dim3(256), 0, s, (double2*)q, n / 2);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("write16_blocks%d,%zu,%.4f,%.1f\n", blocks, (n * 8) >> 20, ms / reps, (double)n * 8 * reps / (ms * 1e6));
        }
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 10; ++r) CK(hipMemsetAsync(q, 0, n * 8, s));
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemset,%zu,%.4f,%.1f\n", (n * 8) >> 20, ms / 10, (double)n * 8 * 10 / (ms * 1e6));
    }
    CK(hipGetLastError());
    return 0;
}
