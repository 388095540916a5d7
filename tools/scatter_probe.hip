// scatter_probe.hip -- measurement tool, not part of the library: what does rocprofv3's FETCH_SIZE report for SCATTERED small
// reads (the row pieces of k_profile: 16..64 bytes at strides of a row), against a count of the bytes / lines known by
// construction?  MI355X_MICROARCH.md calibrates FETCH_SIZE (x2) for wide coalesced streaming reads only.
// Every kernel reads one piece of PIECE bytes (as 16-byte loads) at byte offset OFF of every STRIDE-byte block of a 2 GiB
// buffer, once.  Pieces P = 2 GiB / STRIDE.  Build: hipcc --offload-arch=gfx950 -O3 tools/scatter_probe.hip -o tools/scatter_probe
// Run:   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o pmc --output-format csv -- tools/scatter_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int STRIDE, int OFF, int PIECE>
__global__ void k_scatter(const uint8_t* p, size_t pieces, unsigned int* sink) {
    unsigned int acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pieces; i += (size_t)gridDim.x * blockDim.x) {
        const u4* q = (const u4*)(p + i * STRIDE + OFF);
#pragma unroll
        for (int k = 0; k < PIECE / 16; ++k) { const u4 v = q[k]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int STRIDE, int OFF, int PIECE>
static void run(const uint8_t* buf, size_t bytes, unsigned int* sink) {
    const size_t pieces = bytes / STRIDE;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k_scatter<STRIDE, OFF, PIECE>), dim3(8192), dim3(256), 0, 0, buf, pieces, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    // lines of 128 bytes / sectors of 64 bytes the piece touches
    const int l128 = (OFF + PIECE - 1) / 128 - OFF / 128 + 1, s64 = (OFF + PIECE - 1) / 64 - OFF / 64 + 1;
    printf("k_scatter<%d,%d,%d> pieces %zu piece_bytes %zu bytes_in_128B_lines %zu bytes_in_64B_sectors %zu ms %.3f\n", STRIDE, OFF, PIECE, pieces,
           pieces * PIECE, pieces * l128 * 128, pieces * s64 * 64, ms);
}

int main() {
    const size_t bytes = (size_t)2 << 30;
    uint8_t* buf; unsigned int* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 1, bytes)); CK(hipDeviceSynchronize());
    run<256, 0, 16>(buf, bytes, sink);        // one 16-byte piece in every other 128-byte line
    run<256, 64, 16>(buf, bytes, sink);       // the same in the upper half of the line
    run<256, 0, 64>(buf, bytes, sink);        // 64 bytes = the lower half of a line
    run<256, 32, 64>(buf, bytes, sink);       // 64 bytes across the two halves of one line
    run<256, 96, 64>(buf, bytes, sink);       // 64 bytes across two lines
    run<512, 0, 16>(buf, bytes, sink);
    run<4096, 0, 16>(buf, bytes, sink);       // the analyze path's shape: a piece per 4 KiB row
    run<4096, 112, 48>(buf, bytes, sink);     // 48 bytes across two lines, one per row
    run<128, 0, 128>(buf, bytes, sink);       // everything: the calibrated streaming case (2 GiB)
    return 0;
}
