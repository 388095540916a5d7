// bitplane_probe.hip -- measurement tool, not part of the library.
// What does a thin write stream (the predicted-threshold bit plane: 128 bytes written per 3072 bytes read) cost a pass that
// otherwise reads at the HBM ceiling, and does the SHAPE of the store matter?  One wave takes rounds of 3 KiB as three coalesced
// nontemporal 16-byte loads per lane (the shape of k_minmax_rgb / k_hist_rgb / k_grey_hist) and writes 16 bits per lane and round:
//   0 no store   1 2-byte store per lane (what the passes do)   2 the same, nontemporal   3 8 lanes store 16 bytes (gathered in LDS)
//   4 32 lanes store 4 bytes (pairs joined by DPP)   5 as 3, nontemporal   6 2-byte stores into 512 bytes rewritten over and over
//   9 no store, but the round goes through the 3 KiB LDS turn of the real passes (coalesced pieces in, 48 contiguous bytes per lane out)
//   7 the block keeps its bits in LDS and writes them at its end, 16 bytes per lane (one burst of <= 24 KiB per block)   8 the same, nontemporal
// Build: hipcc --offload-arch=gfx950 -O3 tools/bitplane_probe.hip -o tools/bitplane_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_pass(const u32x4* img, uint8_t* plane, int rounds, uint32_t* sink) {
    __shared__ uint16_t buf[4][64];
    __shared__ u32x4 keep[(MODE == 7 || MODE == 8) ? 1536 : 1];      // 24 KiB: 192 rounds
    __shared__ u32x4 turn[MODE == 9 ? 4 : 1][MODE == 9 ? 192 : 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int per = (rounds + gridDim.x - 1) / gridDim.x;
    const int r_end = min(rounds, (int)(blockIdx.x + 1) * per);
    uint32_t acc = 0;
    u32x4 n0, n1, n2;
    int r = blockIdx.x * per + wv;
    auto request = [&](int rr) {
        const u32x4* p = img + (size_t)rr * 192 + lane;
        n0 = __builtin_nontemporal_load(p); n1 = __builtin_nontemporal_load(p + 64); n2 = __builtin_nontemporal_load(p + 128);
    };
    if (r < r_end) request(r);
    for (; r < r_end; r += 4) {
        u32x4 a = n0, b = n1, c = n2;
        if (MODE == 9) { turn[wv][lane] = a; turn[wv][64 + lane] = b; turn[wv][128 + lane] = c; }
        if (r + 4 < r_end) request(r + 4);
        if (MODE == 9) { a = turn[wv][3 * lane]; b = turn[wv][3 * lane + 1]; c = turn[wv][3 * lane + 2]; }
        const uint32_t x = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w;
        const uint32_t bits = (x ^ (x >> 16)) & 0xffffu;
        acc += bits;
        uint8_t* out = plane + (size_t)r * 128;
        if (MODE == 1) *(uint16_t*)(out + lane * 2) = (uint16_t)bits;
        if (MODE == 2) __builtin_nontemporal_store((uint16_t)bits, (uint16_t*)(out + lane * 2));
        if (MODE == 3 || MODE == 5) {
            buf[wv][lane] = (uint16_t)bits;
            if (lane < 8) {
                const u32x4 v = ((const u32x4*)buf[wv])[lane];
                if (MODE == 3) *(u32x4*)(out + lane * 16) = v; else __builtin_nontemporal_store(v, (u32x4*)(out + lane * 16));
            }
        }
        if (MODE == 4) {
            const uint32_t other = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bits, 0xb1, 0xf, 0xf, false);   // lane ^ 1
            if (!(lane & 1)) *(uint32_t*)(out + lane * 2) = bits | (other << 16);
        }
        if (MODE == 7 || MODE == 8) ((uint16_t*)keep)[(size_t)(r - blockIdx.x * per) * 64 + lane] = (uint16_t)bits;
        if (MODE == 6) *(uint16_t*)(plane + (size_t)(blockIdx.x & 1023) * 512 + wv * 128 + lane * 2) = (uint16_t)bits;
    }
    if (MODE == 7 || MODE == 8) {
        __syncthreads();
        const int mine = max(0, r_end - (int)blockIdx.x * per);                 // rounds of this block, 8 x 16 bytes each
        u32x4* out = (u32x4*)(plane + (size_t)blockIdx.x * per * 128);
        for (int k = threadIdx.x; k < mine * 8; k += 256) { if (MODE == 7) out[k] = keep[k]; else __builtin_nontemporal_store(keep[k], out + k); }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
float run(const u32x4* img, uint8_t* plane, int rounds, uint32_t* sink, int blocks) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(a));
        k_pass<MODE><<<blocks, 256>>>(img, plane, rounds, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep && ms < best) best = ms;
    }
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return best;
}

int main() {
    const size_t img_bytes = (size_t)256 * 9437184, plane_bytes = img_bytes / 24;
    const int rounds = (int)(img_bytes / 3072);
    u32x4* img; uint8_t* plane; uint32_t* sink;
    CK(hipMalloc(&img, img_bytes)); CK(hipMalloc(&plane, plane_bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(img, 3, img_bytes)); CK(hipMemset(plane, 0, plane_bytes));
    printf("mode,blocks,ms,image_GBps\n");
    for (int blocks : {4096, 6144}) {
        const float t[10] = {run<0>(img, plane, rounds, sink, blocks), run<1>(img, plane, rounds, sink, blocks), run<2>(img, plane, rounds, sink, blocks),
                            run<3>(img, plane, rounds, sink, blocks), run<4>(img, plane, rounds, sink, blocks), run<5>(img, plane, rounds, sink, blocks),
                            run<6>(img, plane, rounds, sink, blocks), run<7>(img, plane, rounds, sink, blocks), run<8>(img, plane, rounds, sink, blocks),
                            run<9>(img, plane, rounds, sink, blocks)};
        for (int m = 0; m < 10; ++m) printf("%d,%d,%.4f,%.0f\n", m, blocks, t[m], img_bytes / (t[m] * 1e-3) / 1e9);
        fflush(stdout);
    }
    return 0;
}
