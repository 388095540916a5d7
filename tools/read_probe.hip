// read_probe.hip -- measurement tool, not part of the library: which shape of a pure streaming read gets closest to the HBM
// read ceiling?  Variants: grid-stride (what k_minmax_rgb does) vs one contiguous chunk per block; 4 / 8 / 16 loads in
// flight per lane; default vs nontemporal loads; 256 vs 1024 threads; grid sizes.
// Build: hipcc --offload-arch=gfx950 -O3 tools/read_probe.hip -o tools/read_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool CHUNK>
__global__ void k_read(const u4* p, size_t n16, unsigned int* sink) {
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    size_t i, step, end;
    if (CHUNK) {                                   // block b reads [b*per, (b+1)*per), its threads interleaved inside
        const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
        i = (size_t)blockIdx.x * per + threadIdx.x; step = blockDim.x; end = min(n16, (size_t)(blockIdx.x + 1) * per);
    } else {
        i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; step = nthreads; end = n16;
    }
    unsigned int acc = 0;
    for (; i + (U - 1) * step < end; i += U * step) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * step) : p[i + u * step];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < end; i += step) { const u4 a = p[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the colour passes' shape: a lane owns 48 contiguous bytes (16 pixels) = three 16-byte loads at a lane stride of 48 bytes;
// G groups in flight per lane; block b owns a contiguous run of groups (CHUNK) or the groups are dealt grid-stride
template <int G, bool NT, bool CHUNK>
__global__ void k_read48(const u4* p, size_t ngroups, unsigned int* sink) {
    size_t g, step, end;
    if (CHUNK) {
        const size_t per = (ngroups + gridDim.x - 1) / gridDim.x;
        g = (size_t)blockIdx.x * per + threadIdx.x; step = blockDim.x; end = min(ngroups, (size_t)(blockIdx.x + 1) * per);
    } else {
        g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; step = (size_t)gridDim.x * blockDim.x; end = ngroups;
    }
    unsigned int acc = 0;
    for (; g + (G - 1) * step < end; g += G * step) {
        u4 v[3 * G];
#pragma unroll
        for (int u = 0; u < G; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) v[3 * u + k] = NT ? __builtin_nontemporal_load(p + 3 * (g + u * step) + k) : p[3 * (g + u * step) + k];
#pragma unroll
        for (int u = 0; u < 3 * G; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; g < end; g += step) for (int k = 0; k < 3; ++k) { const u4 a = p[3 * g + k]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int G, bool NT, bool CHUNK>
static void run48(const uint8_t* buf, size_t bytes, unsigned int* sink, int threads, int blocks, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const size_t ng = bytes / 48;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_read48<G, NT, CHUNK>), dim3(blocks), dim3(threads), 0, s, (const u4*)buf, ng, sink);
    CK(hipEventRecord(e0, s));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_read48<G, NT, CHUNK>), dim3(blocks), dim3(threads), 0, s, (const u4*)buf, ng, sink);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("read48,G%d,%s,%s,threads%d,blocks%d,%.4f,%.1f\n", G, NT ? "nt" : "plain", CHUNK ? "chunk" : "stride", threads, blocks, ms / reps,
           (double)bytes * reps / (ms * 1e6));
}

// the min/max pass's launch shape: grid (blocks per image, images), a block strides over its image's 48-byte groups
__global__ void k_read48_img(const u4* p, size_t groups_per_image, unsigned int* sink) {
    const u4* base = p + (size_t)blockIdx.y * groups_per_image * 3;
    unsigned int acc = 0;
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups_per_image; g += (size_t)gridDim.x * blockDim.x) {
        const u4 a = base[3 * g], b = base[3 * g + 1], c = base[3 * g + 2];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
static void run_img(const uint8_t* buf, size_t image_bytes, int n, unsigned int* sink, int bpi, hipStream_t s, hipEvent_t e0, hipEvent_t e1, const char* tag) {
    const size_t gpi = image_bytes / 48;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_read48_img, dim3(bpi, n), dim3(256), 0, s, (const u4*)buf, gpi, sink);
    CK(hipEventRecord(e0, s));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_read48_img, dim3(bpi, n), dim3(256), 0, s, (const u4*)buf, gpi, sink);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("read48_img%s,G1,plain,per_image,threads256,bpi%d,%.4f,%.1f\n", tag, bpi, ms / reps, (double)image_bytes * n * reps / (ms * 1e6));
}

template <int U, bool NT, bool CHUNK>
static void run(const char* name, const uint8_t* buf, size_t bytes, unsigned int* sink, int threads, int blocks, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    const size_t n16 = bytes / 16;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_read<U, NT, CHUNK>), dim3(blocks), dim3(threads), 0, s, (const u4*)buf, n16, sink);
    CK(hipEventRecord(e0, s));
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_read<U, NT, CHUNK>), dim3(blocks), dim3(threads), 0, s, (const u4*)buf, n16, sink);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s,U%d,%s,%s,threads%d,blocks%d,%.4f,%.1f\n", name, U, NT ? "nt" : "plain", CHUNK ? "chunk" : "stride", threads, blocks, ms / reps,
           (double)bytes * reps / (ms * 1e6));
}

int main() {
    const size_t bytes = (size_t)2304 << 20;
    uint8_t* buf; unsigned int* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4)); CK(hipMemset(buf, 1, bytes));
    if (getenv("PROBE_RANDOM")) {                       // image-like content instead of a constant: bytes from a 64-bit LCG
        const size_t n = 64 << 20;
        uint64_t* h = (uint64_t*)malloc(n);
        uint64_t x = 88172645463325252ull;
        for (size_t i = 0; i < n / 8; ++i) { x = x * 6364136223846793005ull + 1442695040888963407ull; h[i] = x ^ (x >> 29); }
        for (size_t off = 0; off < bytes; off += n) CK(hipMemcpy(buf + off, h, n < bytes - off ? n : bytes - off, hipMemcpyHostToDevice));
        free(h);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t s; CK(hipStreamCreate(&s));
    printf("test,unroll,policy,layout,threads,blocks,ms,GBps\n");
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        run<4, false, false>("read", buf, bytes, sink, 256, blocks, s, e0, e1);
        run<8, false, false>("read", buf, bytes, sink, 256, blocks, s, e0, e1);
        run<16, false, false>("read", buf, bytes, sink, 256, blocks, s, e0, e1);
        run<4, true, false>("read", buf, bytes, sink, 256, blocks, s, e0, e1);
        run<8, true, false>("read", buf, bytes, sink, 256, blocks, s, e0, e1);
        run<4, false, true>("read", buf, bytes, sink, 256, blocks, s, e0, e1);
        run<8, false, true>("read", buf, bytes, sink, 256, blocks, s, e0, e1);
        run<8, true, true>("read", buf, bytes, sink, 256, blocks, s, e0, e1);
    }
    for (int blocks : {1280, 2048, 4096, 8192, 16384}) {
        run48<1, false, false>(buf, bytes, sink, 256, blocks, s, e0, e1);
        run48<2, false, false>(buf, bytes, sink, 256, blocks, s, e0, e1);
        run48<1, true, false>(buf, bytes, sink, 256, blocks, s, e0, e1);
        run48<2, true, false>(buf, bytes, sink, 256, blocks, s, e0, e1);
        run48<1, false, true>(buf, bytes, sink, 256, blocks, s, e0, e1);
        run48<2, false, true>(buf, bytes, sink, 256, blocks, s, e0, e1);
        run48<1, true, true>(buf, bytes, sink, 256, blocks, s, e0, e1);
        run48<2, true, true>(buf, bytes, sink, 256, blocks, s, e0, e1);
        run48<4, true, true>(buf, bytes, sink, 256, blocks, s, e0, e1);
    }
    for (int bpi : {4, 8, 16, 32, 64, 192}) run_img(buf, (size_t)1536 * 2048 * 3, 256, sink, bpi, s, e0, e1, "");
    {   // the same out of a 9.6 GB buffer (the bench's batch), reading its 2nd quarter
        uint8_t* big; CK(hipMalloc(&big, (size_t)1024 * 1536 * 2048 * 3)); CK(hipMemset(big, 1, (size_t)1024 * 1536 * 2048 * 3));
        for (int bpi : {16, 64}) run_img(big + (size_t)256 * 1536 * 2048 * 3, (size_t)1536 * 2048 * 3, 256, sink, bpi, s, e0, e1, "_bigbuf");
        CK(hipFree(big));
    }
    for (int blocks : {512, 1024, 2048}) {
        run<4, false, false>("read", buf, bytes, sink, 1024, blocks, s, e0, e1);
        run<8, false, true>("read", buf, bytes, sink, 1024, blocks, s, e0, e1);
        run<4, false, false>("read", buf, bytes, sink, 512, blocks, s, e0, e1);
    }
    return 0;
}
