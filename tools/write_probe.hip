// write_probe.hip -- measurement tool, not part of the library.
// What does a write-dominated stream (the Gaussian stage: 1 byte in, 8 bytes out) reach on this GPU, and which store
// shape gets there?  Variants: store width (8/16 B per lane), plain vs nontemporal, grid-stride vs block-contiguous.
// Build: hipcc --offload-arch=gfx950 -O3 tools/write_probe.hip -o tools/write_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

template <int NT>
__device__ __forceinline__ void st16(d2* p, d2 v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <int NT>
__device__ __forceinline__ void st8(double* p, double v) {
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// fill, 16 B per lane, grid-stride
template <int NT>
__global__ __launch_bounds__(256) void k_fill16(d2* q, size_t n) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    d2 v; v.x = 1.0; v.y = 2.0;
    for (; g + 3 * step < n; g += 4 * step) { st16<NT>(q + g, v); st16<NT>(q + g + step, v); st16<NT>(q + g + 2 * step, v); st16<NT>(q + g + 3 * step, v); }
    for (; g < n; g += step) st16<NT>(q + g, v);
}
// fill, block-contiguous: block b writes [b*per, (b+1)*per)
template <int NT>
__global__ __launch_bounds__(256) void k_fill16_contig(d2* q, size_t per) {
    d2* p = q + (size_t)blockIdx.x * per;
    d2 v; v.x = 1.0; v.y = 2.0;
    for (size_t g = threadIdx.x; g < per; g += 256) st16<NT>(p + g, v);
}

// u8 -> f64, byte loads, 8 B stores (what k_gaussian does today)
template <int NT>
__global__ __launch_bounds__(256) void k_cvt_a(const uint8_t* p, double* q, size_t n) {
    for (size_t base = (size_t)blockIdx.x * 1024; base < n; base += (size_t)gridDim.x * 1024) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t e = base + k * 256 + threadIdx.x;
            st8<NT>(q + e, (double)p[e] * (1.0 / 255.0));
        }
    }
}
// u8 -> f64, one dword load per lane, two 16 B stores per lane (lane stride 32 B)
template <int NT>
__global__ __launch_bounds__(256) void k_cvt_b(const uint32_t* p, d2* q, size_t n4) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; g < n4; g += step) {
        const uint32_t x = p[g];
        d2 a, b;
        a.x = (double)(x & 255) * (1.0 / 255.0); a.y = (double)((x >> 8) & 255) * (1.0 / 255.0);
        b.x = (double)((x >> 16) & 255) * (1.0 / 255.0); b.y = (double)(x >> 24) * (1.0 / 255.0);
        st16<NT>(q + 2 * g, a); st16<NT>(q + 2 * g + 1, b);
    }
}
// u8 -> f64, one ushort load per lane (2 elements), one 16 B store per lane: wave writes 1 KiB contiguous per instruction
template <int NT>
__global__ __launch_bounds__(256) void k_cvt_c(const uint16_t* p, d2* q, size_t n2) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; g + step < n2; g += 2 * step) {
        const uint32_t x = p[g], y = p[g + step];
        d2 a, b;
        a.x = (double)(x & 255) * (1.0 / 255.0); a.y = (double)(x >> 8) * (1.0 / 255.0);
        b.x = (double)(y & 255) * (1.0 / 255.0); b.y = (double)(y >> 8) * (1.0 / 255.0);
        st16<NT>(q + g, a); st16<NT>(q + g + step, b);
    }
    for (; g < n2; g += step) {
        const uint32_t x = p[g];
        d2 a; a.x = (double)(x & 255) * (1.0 / 255.0); a.y = (double)(x >> 8) * (1.0 / 255.0);
        st16<NT>(q + g, a);
    }
}

// like cvt_a, but the input index wraps inside 1 MiB (served from L2): is the limit the HBM read/write mix or the instruction path?
template <int NT>
__global__ __launch_bounds__(256) void k_cvt_a_l2(const uint8_t* p, double* q, size_t n) {
    for (size_t base = (size_t)blockIdx.x * 1024; base < n; base += (size_t)gridDim.x * 1024) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const size_t e = base + k * 256 + threadIdx.x;
            st8<NT>(q + e, (double)p[e & 0xfffff] * (1.0 / 255.0));
        }
    }
}
// like cvt_a with 16 loads requested before the first store
template <int NT>
__global__ __launch_bounds__(256) void k_cvt_a16(const uint8_t* p, double* q, size_t n) {
    for (size_t base = (size_t)blockIdx.x * 4096; base < n; base += (size_t)gridDim.x * 4096) {
        uint8_t v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = p[base + k * 256 + threadIdx.x];
#pragma unroll
        for (int k = 0; k < 16; ++k) st8<NT>(q + base + k * 256 + threadIdx.x, (double)v[k] * (1.0 / 255.0));
    }
}
// software-pipelined: the next group's 8 loads are in flight while this group's 8 stores are issued
template <int NT>
__global__ __launch_bounds__(256) void k_cvt_a_pipe(const uint8_t* p, double* q, size_t n) {
    const size_t stride = (size_t)gridDim.x * 2048;
    size_t base = (size_t)blockIdx.x * 2048;
    uint8_t v[8], nx[8];
    if (base < n) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[base + k * 256 + threadIdx.x];
    }
    for (; base < n; base += stride) {
        const size_t nb = base + stride;
        if (nb < n) {
#pragma unroll
            for (int k = 0; k < 8; ++k) nx[k] = p[nb + k * 256 + threadIdx.x];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) st8<NT>(q + base + k * 256 + threadIdx.x, (double)v[k] * (1.0 / 255.0));
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = nx[k];
    }
}

int main() {
    const size_t out_bytes = (size_t)3072 << 20, in_bytes = out_bytes / 8;
    uint8_t* in; double* out;
    CK(hipMalloc(&in, in_bytes)); CK(hipMalloc(&out, out_bytes));
    CK(hipMemset(in, 7, in_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s; CK(hipStreamCreate(&s));
    printf("test,blocks,ms,GBps\n");
    const int reps = 10;
    float ms;
#define RUN(name, bytes, ...) do { \
        for (int w = 0; w < 2; ++w) { __VA_ARGS__; } \
        CK(hipEventRecord(e0, s)); \
        for (int r = 0; r < reps; ++r) { __VA_ARGS__; } \
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); \
        printf("%s,%d,%.4f,%.1f\n", name, blocks, ms / reps, (double)(bytes) * reps / (ms * 1e6)); } while (0)
    for (int blocks : {2048, 8192, 32768}) {
        const size_t n16 = out_bytes / 16;
        RUN("fill16", out_bytes, k_fill16<0><<<dim3(blocks), dim3(256), 0, s>>>((d2*)out, n16));
        RUN("fill16_nt", out_bytes, k_fill16<1><<<dim3(blocks), dim3(256), 0, s>>>((d2*)out, n16));
        RUN("fill16_contig", out_bytes, k_fill16_contig<0><<<dim3(blocks), dim3(256), 0, s>>>((d2*)out, n16 / blocks));
        RUN("fill16_contig_nt", out_bytes, k_fill16_contig<1><<<dim3(blocks), dim3(256), 0, s>>>((d2*)out, n16 / blocks));
    }
    for (int blocks : {2048, 8192, 32768}) {
        const size_t tot = in_bytes * 9;
        RUN("cvt_a_ld1_st8", tot, k_cvt_a<0><<<dim3(blocks), dim3(256), 0, s>>>(in, out, in_bytes));
        RUN("cvt_a_ld1_st8_nt", tot, k_cvt_a<1><<<dim3(blocks), dim3(256), 0, s>>>(in, out, in_bytes));
        RUN("cvt_a_l2input", tot, k_cvt_a_l2<0><<<dim3(blocks), dim3(256), 0, s>>>(in, out, in_bytes));
        RUN("cvt_a_l2input_nt", tot, k_cvt_a_l2<1><<<dim3(blocks), dim3(256), 0, s>>>(in, out, in_bytes));
        RUN("cvt_a16", tot, k_cvt_a16<0><<<dim3(blocks), dim3(256), 0, s>>>(in, out, in_bytes));
        RUN("cvt_a16_nt", tot, k_cvt_a16<1><<<dim3(blocks), dim3(256), 0, s>>>(in, out, in_bytes));
        RUN("cvt_a_pipe", tot, k_cvt_a_pipe<0><<<dim3(blocks), dim3(256), 0, s>>>(in, out, in_bytes));
        RUN("cvt_a_pipe_nt", tot, k_cvt_a_pipe<1><<<dim3(blocks), dim3(256), 0, s>>>(in, out, in_bytes));
        RUN("cvt_b_ld4_st16x2", tot, k_cvt_b<0><<<dim3(blocks), dim3(256), 0, s>>>((const uint32_t*)in, (d2*)out, in_bytes / 4));
        RUN("cvt_b_ld4_st16x2_nt", tot, k_cvt_b<1><<<dim3(blocks), dim3(256), 0, s>>>((const uint32_t*)in, (d2*)out, in_bytes / 4));
        RUN("cvt_c_ld2_st16", tot, k_cvt_c<0><<<dim3(blocks), dim3(256), 0, s>>>((const uint16_t*)in, (d2*)out, in_bytes / 2));
        RUN("cvt_c_ld2_st16_nt", tot, k_cvt_c<1><<<dim3(blocks), dim3(256), 0, s>>>((const uint16_t*)in, (d2*)out, in_bytes / 2));
    }
    {
        int blocks = 0;
        RUN("hipMemset", out_bytes, CK(hipMemsetAsync(out, 0, out_bytes, s)));
    }
    CK(hipGetLastError());
    return 0;
}
