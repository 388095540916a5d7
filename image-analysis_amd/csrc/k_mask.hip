// k_mask.hip -- stage S4a: threshold compare, written as VERTICAL 64-row bit words  (gfx950)
//
// mask = adjust_gamma(gray, 0.4) > threshold_otsu(...)   (LFA-Image_Analysis/detection.py:39, :47), evaluated as an
// integer compare `pixel > x_thr` (2-D images) or a float64 compare `gray > g_thr` (colour), see k_threshold.hip.
//
// Layout: maskT[image][rb][c] is a uint64 whose bit k is mask[rb*64 + k][c].  Lanes run along c, so the image is read
// in whole row segments (coalesced) and the words are written coalesced; the opening and the run extraction that
// follow work along columns and get 64 rows per load.
#include "lfa_internal.h"
#include "lfa_device.h"

namespace lfa {

typedef uint32_t lfa_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint2 nt_load8(const uint8_t* p) {   // read once, by this kernel only
    const lfa_u32x2 v = __builtin_nontemporal_load((const lfa_u32x2*)p);
    return make_uint2(v.x, v.y);
}


// 8 x 8 bit matrix in two dwords (byte i = row i, bit k = column k; lo = rows 0-3): transposed in place, so that byte k
// holds column k with bit i = row i.  Three delta swaps; the first two stay inside a dword, the last one crosses.
__device__ __forceinline__ void transpose_bits8x8(uint32_t& lo, uint32_t& hi) {
    uint32_t t;
    t = (lo ^ (lo >> 7)) & 0x00aa00aau; lo ^= t ^ (t << 7);
    t = (hi ^ (hi >> 7)) & 0x00aa00aau; hi ^= t ^ (t << 7);
    t = (lo ^ (lo >> 14)) & 0x0000ccccu; lo ^= t ^ (t << 14);
    t = (hi ^ (hi >> 14)) & 0x0000ccccu; hi ^= t ^ (t << 14);
    t = (lo ^ (hi << 4)) & 0xf0f0f0f0u; lo ^= t; hi ^= t >> 4;
}

// Predicted-threshold mode, after k_mask: the pixels of the listed groups that lie inside the window (those the bit plane
// left clear without knowing) get the exact compare against the image's threshold and are ORed into the words k_mask
// stored.  Grid (blocks, n); almost always a handful of entries per image.
__global__ __launch_bounds__(256) void k_mask_fix(Job j) {
    const int img = blockIdx.y;
    const uint32_t* eg = j.edges_i + (int64_t)img * kEdgesI;
    if (eg[kEdgesI_Bins] != 2u) return;
    const uint32_t n = min(eg[kEdgesI_PredCnt], (uint32_t)j.pred_cap);
    const uint32_t t_lo = j.pred_lo, t_hi = j.pred_hi;
    const uint32_t* list = j.pred_list + (int64_t)img * j.pred_stride;
    const uint8_t* base = j.images + (int64_t)img * j.img_bytes;
    uint64_t* out = j.maskT + (int64_t)img * j.HB * j.W;
    // Job::pred_fused: the bits stay where the histogram pass wrote them (k_runs_count turns them itself): pixel p is bit p of the plane
    uint32_t* plane = j.pred_fused ? (uint32_t*)(j.bins + (int64_t)img * (((int64_t)j.H * j.W) >> 3)) : nullptr;
    auto set_pixel = [&](int64_t p, int r, int c) {
        if (plane) atomicOr(&plane[p >> 5], 1u << (p & 31));
        else atomicOr((unsigned long long*)&out[(int64_t)(r >> 6) * j.W + c], 1ull << (r & 63));
    };
    if (j.dtype == LFA_U16) {                                             // uint16 frames: pixel >= T = x_thr + 1
        const uint32_t T = (uint32_t)j.info[img].x_thr + 1u;
        for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
            const int64_t p0 = (int64_t)list[e] * 16;
            const int r = (int)(p0 / j.W), c0 = (int)(p0 % j.W);
            const uint16_t* px = (const uint16_t*)base + p0;
            for (int k = 0; k < 16; ++k) {
                const uint32_t x = px[k];
                if (x >= t_lo && x < t_hi && x >= T) set_pixel(p0 + k, r, c0 + k);
            }
        }
        return;
    }
    const uint32_t G_thr = eg[kEdgesI_MaskThr];
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int64_t p0 = (int64_t)list[e] * 16;                         // W % 16 == 0: a group never crosses a row
        const int r = (int)(p0 / j.W), c0 = (int)(p0 % j.W);
        uint32_t w[12];
        load48(base + p0 * 3, w);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint32_t G = gray_G16(w, k);
            if (G >= t_lo && G < t_hi && G >= G_thr) set_pixel(p0 + k, r, c0 + k);
        }
    }
}

// MODE 0: uint8 2-D, 1: uint16 2-D, 2: uint8 colour.  VEC: 16 pixels per lane per row through 16-byte loads.
// PRED: a call in predicted-threshold mode (bit plane or the image itself; a kernel of its own, so that its registers do
// not weigh on the class-plane path).
template <int MODE, int V, bool VEC, bool PRED = false>
__global__ __launch_bounds__(256) void k_mask(Job j) {
    const int img = blockIdx.y;
    const LfaImageInfo inf = j.info[img];
    const uint8_t* base = j.images + (int64_t)img * j.img_bytes;
    const int H = j.H, W = j.W;
    const int x_thr = inf.x_thr;
    const double g_thr = inf.g_thr;
    // first double above g_thr (for the ambiguity census only)
    const long long gb = __double_as_longlong(g_thr);
    const double g_next = (gb >= 0 && gb < 0x7ff0000000000000LL) ? __longlong_as_double(gb + 1) : g_thr;
    uint64_t* out = j.maskT + (int64_t)img * j.HB * W;
    uint32_t amb = 0;
    // colour: integer form of the compare (k_threshold.hip g_threshold) unless the threshold is critical
    uint32_t G_thr = 0;
    bool integer_path = false;
    if (MODE != 0) {
        const uint32_t* eg = j.edges_i + (int64_t)img * kEdgesI;
        if (MODE == 2) {
            G_thr = eg[kEdgesI_MaskThr];
            integer_path = eg[kEdgesI_MaskCrit] == 0;
        }
        if (PRED && VEC && j.bins && eg[kEdgesI_Bins] == 2u && j.pred_fused) return;      // k_runs_count reads the bit plane itself
        if (PRED && VEC && j.bins && eg[kEdgesI_Bins] == 2u) {
            // Predicted-threshold bit plane (Job::pred_mode): bit = G >= pred_hi, row-major, 16 pixels = one 16-bit word.  The
            // image's threshold lies inside the window, so a set bit is a set pixel and a clear bit a clear pixel except for
            // the listed pixels inside the window, which k_mask_fix ORs in afterwards.  What remains here is turning 64 rows
            // x 16 columns of bits into 16 vertical words: 8 x 8 bit blocks, transposed by three delta swaps each.
            const int ncg = W >> 4;
            const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
            if (item >= (int64_t)j.HB * ncg) return;
            const int rb = (int)(item / ncg), cg = (int)(item % ncg);
            const int r0 = rb * 64;
            const int rows = min(64, H - r0);
            const int64_t rowb = (int64_t)W >> 3;
            const uint8_t* bp = j.bins + (int64_t)img * (((int64_t)H * W) >> 3) + (int64_t)r0 * rowb + (int64_t)cg * 2;
            uint32_t rw[64];
            if (__all(rows == 64)) {                                      // (all but the image's last row block: no lane masks the loads)
#pragma unroll
                for (int i = 0; i < 64; ++i)                              // all 64 rows are requested together
                    rw[i] = (uint32_t)__builtin_nontemporal_load((const uint16_t*)(bp + (int64_t)i * rowb));
            } else {
#pragma unroll
                for (int i = 0; i < 64; ++i)
                    rw[i] = (i < rows) ? (uint32_t)__builtin_nontemporal_load((const uint16_t*)(bp + (int64_t)i * rowb)) : 0u;
            }
            uint32_t Ta[8][2], Tb[8][2];                                  // [row byte jb][low / high dword]: columns 0-7 / 8-15
#pragma unroll
            for (int jb = 0; jb < 8; ++jb) {
                const uint32_t* q = rw + jb * 8;
                // byte i of the 64-bit matrix = bits of row 8jb+i: columns 0-7 (byte 0 of the row word) or 8-15 (byte 1)
                const uint32_t a01 = __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0400u), a23 = __builtin_amdgcn_perm(q[3], q[2], 0x0c0c0400u);
                const uint32_t a45 = __builtin_amdgcn_perm(q[5], q[4], 0x0c0c0400u), a67 = __builtin_amdgcn_perm(q[7], q[6], 0x0c0c0400u);
                const uint32_t b01 = __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0501u), b23 = __builtin_amdgcn_perm(q[3], q[2], 0x0c0c0501u);
                const uint32_t b45 = __builtin_amdgcn_perm(q[5], q[4], 0x0c0c0501u), b67 = __builtin_amdgcn_perm(q[7], q[6], 0x0c0c0501u);
                Ta[jb][0] = __builtin_amdgcn_perm(a23, a01, 0x05040100u); Ta[jb][1] = __builtin_amdgcn_perm(a67, a45, 0x05040100u);
                Tb[jb][0] = __builtin_amdgcn_perm(b23, b01, 0x05040100u); Tb[jb][1] = __builtin_amdgcn_perm(b67, b45, 0x05040100u);
                transpose_bits8x8(Ta[jb][0], Ta[jb][1]);
                transpose_bits8x8(Tb[jb][0], Tb[jb][1]);                  // now byte k = column k (+8), bit i = row 8jb+i
            }
            uint64_t* o = out + (int64_t)rb * W + cg * 16;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                uint32_t lo2[2], hi2[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int kk = k + e, half = (kk & 7) >> 2, m = kk & 3;
                    const uint32_t sel = (uint32_t)(((4 + m) << 8) | m);            // byte0 = second operand's byte m, byte1 = first's
                    uint32_t b8[8];
#pragma unroll
                    for (int jb = 0; jb < 8; ++jb) b8[jb] = (kk < 8) ? Ta[jb][half] : Tb[jb][half];
                    const uint32_t p01 = __builtin_amdgcn_perm(b8[1], b8[0], sel), p23 = __builtin_amdgcn_perm(b8[3], b8[2], sel);
                    const uint32_t p45 = __builtin_amdgcn_perm(b8[5], b8[4], sel), p67 = __builtin_amdgcn_perm(b8[7], b8[6], sel);
                    lo2[e] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                    hi2[e] = __builtin_amdgcn_perm(p67, p45, 0x05040100u);
                }
                uint4 v;
                v.x = lo2[0]; v.y = hi2[0]; v.z = lo2[1]; v.w = hi2[1];
                *(uint4*)(o + k) = v;
            }
            return;
        }
        if (!PRED && VEC && j.bins && eg[kEdgesI_Bins] == 1u) {
            // The class plane k_hist_rgb wrote (bin >> 4 of every pixel, 4 bits each) decides every pixel whose class
            // differs from the class q of the Otsu bin: class > q: set, class < q: clear -- half a byte per pixel
            // instead of three.  A pixel of class q is settled by the exact compare on the image itself.
            // uint16 frames: the same with the sixteen value ranges of k_hist_u16 and the class of x_thr (two bytes saved).
            const int ncg = W >> 4;
            const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
            if (item >= (int64_t)j.HB * ncg) return;
            const int rb = (int)(item / ncg), cg = (int)(item % ncg);
            const int r0 = rb * 64;
            const int rows = min(64, H - r0);
            const uint32_t q = (MODE == 2) ? (uint32_t)inf.otsu_bin >> 4 : (eg[kEdgesI_Class] & 0xffu);
            const bool settled = MODE == 1 && (eg[kEdgesI_Class] & 0x100u);     // class q is clear as a whole: no compare
            const uint32_t q8 = q * 0x11111111u;
            const uint8_t* bp = j.bins + (((int64_t)img * H * W + (int64_t)r0 * W) >> 1) + (int64_t)cg * 8;
            const int64_t rowb = (int64_t)W >> 1;
            // Eight pixels per instruction: nibble x > q  <=>  x + (15 - q) carries out of its nibble, and the carries of
            // all nibbles of a dword come from one add on the low 3 bits of each plus the majority of the three top
            // bits.  The carries (bit 3 of each nibble) are split into even and odd pixels -- each then one bit per
            // byte -- and the bits of eight consecutive rows are shifted into those bytes:
            //   Se[j][d] byte m = pixel 8d+2m, So[j][d] byte m = pixel 8d+2m+1, bit i = row 8j+i
            // which is byte j of that pixel's vertical 64-row word.
            const uint32_t c8 = (15u - q) * 0x11111111u, c8lo = c8 & 0x77777777u;
            uint32_t Se[8][2], So[8][2];
            uint64_t eqrows = 0;                                          // rows holding a pixel of class q
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
            uint2 av[32];
#pragma unroll
            for (int i = 0; i < 32; ++i)                                  // 32 rows are requested together
                av[i] = (jq * 32 + i < rows) ? nt_load8(bp + (int64_t)(jq * 32 + i) * rowb) : make_uint2(0u, 0u);
#pragma unroll
            for (int jh = 0; jh < 4; ++jh) {
                const int jb = jq * 4 + jh;
                uint32_t ae[2] = {0u, 0u}, ao[2] = {0u, 0u};
#pragma unroll
                for (int i = 7; i >= 0; --i) {                            // top row first: it ends up shifted to bit 7
                    const int r = jb * 8 + i;
                    const uint32_t w[2] = {av[jh * 8 + i].x, av[jh * 8 + i].y};
                    uint32_t car[2], anyeq = 0;
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const uint32_t x = w[d];
                        const uint32_t low = (x & 0x77777777u) + c8lo;
                        car[d] = ((x & c8) | ((x | c8) & low)) & 0x88888888u;
                        const uint32_t v = x ^ q8;                       // a zero nibble <=> that pixel is of class q
                        anyeq |= (v - 0x11111111u) & ~v;
                    }
                    const bool valid = r < rows;
                    eqrows |= (uint64_t)((valid && (anyeq & 0x88888888u)) ? 1u : 0u) << r;     // settled after the main loop
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        const uint32_t c = valid ? car[d] : 0u;
                        ae[d] = (ae[d] << 1) | ((c >> 3) & 0x01010101u);
                        ao[d] = (ao[d] << 1) | ((c >> 7) & 0x01010101u);
                    }
                }
#pragma unroll
                for (int d = 0; d < 2; ++d) { Se[jb][d] = ae[d]; So[jb][d] = ao[d]; }
            }
            }
            // column k = 8d + 2m + odd: low word = byte m of S[0..3][d], high word = byte m of S[4..7][d]
            uint32_t lo[16], hi[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int d = k >> 3, m = (k & 7) >> 1, odd = k & 1;
                const uint32_t sel = (uint32_t)(((4 + m) << 8) | m);                // byte0 = second operand's byte m, byte1 = first's
                uint32_t b8[8];
#pragma unroll
                for (int jb = 0; jb < 8; ++jb) b8[jb] = odd ? So[jb][d] : Se[jb][d];
                const uint32_t p01 = __builtin_amdgcn_perm(b8[1], b8[0], sel), p23 = __builtin_amdgcn_perm(b8[3], b8[2], sel);
                const uint32_t p45 = __builtin_amdgcn_perm(b8[5], b8[4], sel), p67 = __builtin_amdgcn_perm(b8[7], b8[6], sel);
                lo[k] = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                hi[k] = __builtin_amdgcn_perm(p67, p45, 0x05040100u);
            }
            uint64_t* o = out + (int64_t)rb * W + cg * 16;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                uint4 v;
                v.x = lo[k]; v.y = hi[k]; v.z = lo[k + 1]; v.w = hi[k + 1];
                *(uint4*)(o + k) = v;
            }
            // Pixels of class q: the exact compare on the image itself.  Rare (the classes around the Otsu bin are the
            // valley of the histogram), so this is a small rolled loop that ORs straight into the words just written
            // -- this lane's own words -- instead of 64 unrolled copies inside the loop above.
            if (settled) eqrows = 0;
            while (eqrows) {
                const int r = __ffsll((unsigned long long)eqrows) - 1;
                eqrows &= eqrows - 1;
                const uint2 a = *(const uint2*)(bp + (int64_t)r * rowb);
                const uint8_t* px = base + ((int64_t)(r0 + r) * W + (int64_t)cg * 16) * (MODE == 2 ? 3 : 2);
                // only the nibbles equal to q are visited: the zero-nibble flags of x ^ q8 (a flag may also come up for a
                // nibble just above a zero one, hence the exact re-check)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const uint32_t x = d ? a.y : a.x;
                    const uint32_t v = x ^ q8;
                    uint32_t z = (v - 0x11111111u) & ~v & 0x88888888u;
                    while (z) {
                        const int b = __ffs((int)z) - 1;                  // bit 3 of the nibble
                        z &= z - 1;
                        if (((x >> (b - 3)) & 15u) != q) continue;
                        const int k = d * 8 + (b >> 2);
                        const bool set = (MODE == 2) ? (2125u * px[3 * k] + 7154u * px[3 * k + 1] + 721u * px[3 * k + 2]) >= G_thr
                                                     : (int)((const uint16_t*)px)[k] > x_thr;
                        if (set) atomicOr((unsigned long long*)&o[k], 1ull << r);
                    }
                }
            }
            return;
        }
    }

    if (VEC) {
        const int ncg = W >> 4;
        const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (item < (int64_t)j.HB * ncg) {
            const int rb = (int)(item / ncg), cg = (int)(item % ncg);
            const int r0 = rb * 64;
            const int rows = min(64, H - r0);
            uint32_t lo[16], hi[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) { lo[k] = 0; hi[k] = 0; }
            // Equal-channel image of a grey-mode call (k_grey_hist left channel 0 of every pixel in a byte plane): G = 10000 * b, so
            // `G >= G_thr` is `b >= ceil(G_thr / 10000)` -- one byte per pixel read instead of three.  Integer form of the threshold only.
            const bool grey_bytes = MODE == 2 && j.grey_plane && integer_path && j.edges_i[(int64_t)img * kEdgesI + kEdgesI_Grey] != 0u;
            const int b_thr = (G_thr > 2550000u) ? 256 : (int)((G_thr + 9999u) / 10000u);
            const int64_t px_bytes = grey_bytes ? 1 : ((MODE == 2) ? 3 : (MODE == 1 ? 2 : 1));
            const uint8_t* src = grey_bytes ? j.grey_plane + (int64_t)img * j.grey_plane_stride : base;
            const uint8_t* p = src + ((int64_t)r0 * W + (int64_t)cg * 16) * px_bytes;
            const int64_t row_bytes = (int64_t)W * px_bytes;
            // one row of 16 pixels: compare and drop the result into bit `rbit` of the 16 column words
            auto row = [&](const uint8_t* q, uint32_t (&acc)[16], uint32_t rbit) {
                if (MODE == 0) {
                    const uint4 a = *(const uint4*)q;
                    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[k] |= ((int)byte_of(w, k) > x_thr) ? rbit : 0u;
                } else if (MODE == 1) {
                    const uint4 a = ((const uint4*)q)[0], b = ((const uint4*)q)[1];
                    const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int x = (int)((w[k >> 1] >> ((k & 1) * 16)) & 0xffffu);
                        acc[k] |= (x > x_thr) ? rbit : 0u;
                    }
                } else if (grey_bytes) {
                    const uint4 a = *(const uint4*)q;
                    const uint32_t w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[k] |= ((int)byte_of(w, k) >= b_thr) ? rbit : 0u;
                } else {
                    uint32_t w[12];
                    load48(q, w);
                    if (integer_path) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) acc[k] |= (gray_G16(w, k) >= G_thr) ? rbit : 0u;
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; ++k) {
                            const double g = gray_u8<V>(byte_of(w, 3 * k), byte_of(w, 3 * k + 1), byte_of(w, 3 * k + 2));
                            acc[k] |= (g > g_thr) ? rbit : 0u;
                            amb += near_rel(g, g_next) ? 1u : 0u;
                        }
                    }
                }
            };
            const int rows_lo = min(rows, 32);
#pragma unroll 2
            for (int r = 0; r < rows_lo; ++r) row(p + (int64_t)r * row_bytes, lo, 1u << r);
#pragma unroll 2
            for (int r = 32; r < rows; ++r) row(p + (int64_t)r * row_bytes, hi, 1u << (r - 32));
            uint64_t* o = out + (int64_t)rb * W + cg * 16;
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
                uint4 v;
                v.x = lo[k]; v.y = hi[k]; v.z = lo[k + 1]; v.w = hi[k + 1];
                *(uint4*)(o + k) = v;
            }
        }
    } else {
        const int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (item < (int64_t)j.HB * W) {
            const int rb = (int)(item / W), c = (int)(item % W);
            const int r0 = rb * 64;
            const int rows = min(64, H - r0);
            uint64_t acc = 0;
            for (int r = 0; r < rows; ++r) {
                const int64_t px = (int64_t)(r0 + r) * W + c;
                bool bit;
                if (MODE == 0) bit = (int)base[px] > x_thr;
                else if (MODE == 1) bit = (int)((const uint16_t*)base)[px] > x_thr;
                else if (integer_path) {
                    bit = (2125u * base[3 * px] + 7154u * base[3 * px + 1] + 721u * base[3 * px + 2]) >= G_thr;
                } else {
                    const double g = gray_at<V>(base, px, j.dtype, j.C);
                    bit = g > g_thr;
                    amb += near_rel(g, g_next) ? 1u : 0u;
                }
                acc |= (uint64_t)(bit ? 1 : 0) << r;
            }
            out[(int64_t)rb * W + c] = acc;
        }
    }
    if (MODE == 2) {
        amb = wave_sum_u32(amb);
        if ((threadIdx.x & 63) == 0 && amb) atomicAdd(&j.info[img].ambiguous, amb);
    }
}

template <int MODE, int V>
static void launch_mode(const Job& j, hipStream_t s) {
    const bool vec = j.aligned16 && (j.W % 16 == 0);
    if (vec) {
        const int64_t items = (int64_t)j.HB * (j.W >> 4);
        if (MODE != 0 && j.pred_mode) {
            LFA_LAUNCH("k_mask", (k_mask<MODE, V, true, true>), dim3((unsigned)((items + 255) / 256), j.n), dim3(256), 0, s, j);
            LFA_LAUNCH("k_mask_fix", k_mask_fix, dim3(4, j.n), dim3(256), 0, s, j);
        } else {
            LFA_LAUNCH("k_mask", (k_mask<MODE, V, true>), dim3((unsigned)((items + 255) / 256), j.n), dim3(256), 0, s, j);
        }
    } else {
        const int64_t items = (int64_t)j.HB * j.W;
        LFA_LAUNCH("k_mask_scalar", (k_mask<MODE, V, false>), dim3((unsigned)((items + 255) / 256), j.n), dim3(256), 0, s, j);
    }
}

hipError_t launch_mask(const Job& j, hipStream_t s) {
    if (j.C == 3 || j.dtype == LFA_F64) {                    // gray is a float64: colour images, float64 images
        switch (j.gray_variant) {
            case 0: launch_mode<2, 0>(j, s); break;
            case 1: launch_mode<2, 1>(j, s); break;
            case 2: launch_mode<2, 2>(j, s); break;
            case 3: launch_mode<2, 3>(j, s); break;
            case 4: launch_mode<2, 4>(j, s); break;
            default: launch_mode<2, 5>(j, s); break;
        }
    } else if (j.dtype == LFA_U8) {
        launch_mode<0, 0>(j, s);
    } else {
        launch_mode<1, 0>(j, s);
    }
    return hipGetLastError();
}

}  // namespace lfa
