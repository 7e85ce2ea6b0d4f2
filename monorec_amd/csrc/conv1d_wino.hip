// 3x1 / 1x3 stride-1 convolutions as 1-D Winograd F(2, 3) on the fp32 matrix cores of gfx950 (MI355X).
//
// layers.ConvReLU2 (reference model/layers.py:289-314) splits every k x k convolution of the DepthModule
// (model/monorec/monorec_model.py:485-513) into a k x 1 and a 1 x k one; sixteen of them per keyframe are 3-tap, stride 1, 'same'
// padded (enc{0..4}.1, dec{1,2}.1, dec4.0: 7.3 of the 27.2 GMAC of the depth net at c2).  F(2, 3) computes 2 outputs along the filter
// axis from 4 inputs with 4 multiplies per (cin, cout) instead of 6:
//     y = A^T [ sum_cin (G g) o (B^T d) ],   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],
//     A^T = [1 1 1 0; 0 1 -1 -1]
// - the 1-D factor of the F(2x2, 3x3) kernel in conv_wino.hip, and the same skeleton as its in-register-transform variant:
// workgroup = 8 waves, 8 x 32 output pixels, 16 * MBW output channels (every wave sweeps ALL of them: MBW = 3 covers the 48-channel
// layers without padding), K in chunks of 8 input channels; the haloed region (10 rows x 40 columns, 16-byte aligned; one halo
// direction is unused) and the chunk's U fragments (G g, formed in double, rounded once; 2 KiB per 16 output channels) arrive by LDS-DMA
// in one of two pipeline buffers (<= 42 KB: three to four workgroups share a CU); ONE barrier per chunk.
//   AXIS 0 (1 x 3, along x): wave = output row, lane & 15 = tile of 2 columns; AXIS 1 (3 x 1, along y): wave = (tile of 2 rows, half of
//   the 32 columns), lane & 15 = column.  Either way a lane reads the 4 inputs of its tile for its channel (lane >> 4 of the quad) from
//   the raw region, transforms them (4 adds) and holds the 4 values as MFMA B operands; it ends up with the 4 positions of its
//   (cout, tile) in registers, so the output transform, bias and LeakyReLU run there as well.
// The sum over cin is the exact-order fp32 FMA chain of every MFMA; the transforms round differently from the direct sum (~1e-6
// relative on O(1) outputs, like the 2-D kernel).
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <atomic>

#include "../../include/monorec_hip.h"
#include "cooktoom_1d.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int WCK = 8;                                   // input channels per chunk (one per wave in the DMA phase)
constexpr int RAW_PITCH = 40, RAW_ROWS = 10, RAW_PLANE = RAW_PITCH * RAW_ROWS;   // rows oy0-1 .. oy0+8, columns ox0-4 .. ox0+35
// 1 x 3 (AXIS 0): tile t reads columns 2 t + 3 .. 2 t + 6 of its row - lane stride 2, so dword reads of the two channels of a 32-lane group always collide
// (an even plane pitch keeps the parity of the bank): read as aligned 8-byte pairs from column 2 t + 2 instead, conflict free over the 64 banks of a
// ds_read_b64 when the plane pitch is 32 mod 64 (tools/lds_banks.py)
constexpr int RAW_PLANE_X = 416;
static_assert(RAW_PLANE_X >= RAW_PLANE && RAW_PLANE_X % 64 == 32 && RAW_PLANE % 32 == 16, "plane pitches");
constexpr int NPOS = 4;

struct W1KArgs {
    const float* src[MR_MAX_SOURCES];
    int src_bytes[MR_MAX_SOURCES];
    int src_c[MR_MAX_SOURCES];
    int src_cpad[MR_MAX_SOURCES];       // padded to a multiple of WCK
    int nsrc;
    int H, W;
    float* dst;
    const float* bias;
    int act;
    float p0;
    int Cout, tiles_x, nchunks;
    const float* w;
    long long wgroup_stride;            // packed floats per cout group
    // conv1d_ct_kernel only (round 5: the stride-2 layers as stride-1 forms over [even | odd] views of their input):
    int pitch, cplane;                  // floats between rows / channel planes of a source as stored (dense: W, H * W)
    int dst_split;                      // axis 1: the destination is (2, batch, Cout, H, W / 2) - even columns, odd columns
    long long dst_half;                 // floats of one parity half of such a destination
};

// LDS-DMA through inline asm (see conv_mfma.hip: the builtins make hipcc drain vmcnt before every sweep)
__device__ __forceinline__ void dma_buffer_x4(unsigned lds_byte_addr, int voff, i32x4 srd, int soff) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte_addr), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_global_x4(unsigned lds_byte_addr, const float* g) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte_addr), "v"(g) : "memory");
}
// One aligned 8-byte LDS read that stays one: left to itself hipcc drops the halves a caller does not use and re-pairs the rest into ds_read2_b32 -
// two dword accesses with the 32-bank rule (volatile keeps the access whole; the explicit LDS address space keeps it a ds_ instruction)
__device__ __forceinline__ f32x2 lds_pair(const float* p) {
    return *(const volatile __attribute__((address_space(3))) f32x2*)(__attribute__((address_space(3))) const float*)p;
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ i32x4 make_srd(const void* base, int bytes) {
    const unsigned long long p = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000;
    return r;
}

// none / ReLU / LeakyReLU as ONE branch-free form, max(x, lo) with lo = x (none), 0 (ReLU: -inf -> 0 and no -0.0, like torch.relu),
// x * p0 (LeakyReLU, 0 <= p0 <= 1 - the host side rejects other slopes); lo's selector is wave-uniform: as a
// switch the compiler emitted scalar branches around every stored element of the epilogue (round 4: 200-450 branches per workgroup)
__device__ __forceinline__ float act1(float v, int act, float p0) {
    const unsigned keep = act == MR_ACT_RELU ? 0u : ~0u;          // (an AND, not a select: a uniform select made hipcc clone the store loops)
    const float lo = __uint_as_float(__float_as_uint(v * (act == MR_ACT_LEAKY_RELU ? p0 : 1.f)) & keep);
    return fmaxf(v, lo);
}

template <int AXIS, int MBW>
__global__ __launch_bounds__(512) void conv1d3_wino_kernel(const W1KArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int U_FLOATS = NPOS * 2 * MBW * 64;            // U fragments of one chunk: [p][c4][cout block][64 lanes]
    constexpr int PLANE = AXIS == 0 ? RAW_PLANE_X : RAW_PLANE;
    constexpr int BUF = WCK * PLANE + U_FLOATS;              // one pipeline buffer: raw input region + U
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    const int grp = blockIdx.y, b = blockIdx.z;
    const int oy0 = ty_wg * 8, ox0 = tx_wg * 32;
    const int H = a.H, W = a.W, HW = H * W;

    int voff4[2];                                             // lane l owns the 16-byte groups r = l, l + 64 (< 100): row r / 10, group r % 10
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = lane + 64 * i;
        const int row = r / 10, g4 = r - row * 10;
        const int gy = oy0 - 1 + row, gx = ox0 - 4 + 4 * g4;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff4[i] = r < RAW_ROWS * 10 ? (inb ? (gy * W + gx) * 4 : -1) : -2;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
    const float* wgrp = a.w + (long long)grp * a.wgroup_stride;

    f32x4 acc[NPOS][MBW];
#pragma unroll
    for (int p = 0; p < NPOS; ++p)
#pragma unroll
        for (int m = 0; m < MBW; ++m) acc[p][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;                                      // chunk cursor: source, first channel
    auto issue = [&](int q, int pb) {
        const unsigned buf_addr = lds_base + pb * BUF * 4;
        const unsigned u_addr = buf_addr + WCK * PLANE * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS;
        if (wave < U_FLOATS / 256) dma_global_x4(u_addr + wave * 1024, wsrc + wave * 256 + lane * 4);     // 2 MBW <= 8 pieces of 1 KiB
        const i32x4 srd = make_srd(a.src[cs], a.src_bytes[cs]);
        const bool cok = cc0 + wave < a.src_c[cs];            // padded channels read as zero
        const int so = ((b * a.src_c[cs] + cc0 + (cok ? wave : 0)) * HW) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (voff4[i] != -2) dma_buffer_x4(buf_addr + wave * (PLANE * 4) + i * 1024, cok ? voff4[i] : -1, srd, so);
        cc0 += WCK;
        if (cc0 >= a.src_cpad[cs]) { cc0 = 0; ++cs; }
    };

    issue(0, 0);
    // the 4 inputs of this lane's tile, channel lane >> 4 of a quad: offsets patch0 + i * pstep in its channel plane
    const int t = lane & 15;
    // AXIS 0: aligned pairs from column 2 t + 2 on (inputs = columns 2 t + 3 .. 2 t + 6); AXIS 1: rows 2 (wave >> 1) .. + 3 at column 4 + 16 (wave & 1) + t
    const int patch0 = (lane >> 4) * PLANE + (AXIS == 0 ? (wave + 1) * RAW_PITCH + 2 * t + 2
                                                        : (2 * (wave >> 1)) * RAW_PITCH + 4 + (wave & 1) * 16 + t);
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF;
        const float* ub = raw + WCK * PLANE + lane;
        dma_wait_all();
        __syncthreads();                                      // raw + U of chunk q visible; everyone is done with the other buffer
        if (q + 1 < a.nchunks) issue(q + 1, pb ^ 1);
        float v[2][NPOS];                                     // B operands of this lane: (B^T d)[p] of channels 4 c4 + (lane >> 4)
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const float* rp = raw + patch0 + c4 * 4 * PLANE;
            float d0, d1, d2, d3;
            if constexpr (AXIS == 0) {
                const f32x2 g0 = lds_pair(rp), g1 = lds_pair(rp + 2), g2 = lds_pair(rp + 4);
                d0 = g0.y; d1 = g1.x; d2 = g1.y; d3 = g2.x;
            } else {
                d0 = rp[0]; d1 = rp[RAW_PITCH]; d2 = rp[2 * RAW_PITCH]; d3 = rp[3 * RAW_PITCH];
            }
            v[c4][0] = d0 - d2;
            v[c4][1] = d1 + d2;
            v[c4][2] = d2 - d1;
            v[c4][3] = d1 - d3;
        }
#pragma unroll
        for (int p = 0; p < NPOS; ++p)
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4)
#pragma unroll
                for (int m = 0; m < MBW; ++m) {
                    const float av = ub[((p * 2 + c4) * MBW + m) * 64];
                    acc[p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[c4][p], acc[p][m], 0, 0, 0);
                }
    }
    // ---- output transform y = A^T m per (cout, tile) in registers, epilogue ------------------------------------------------------
    const int ox = ox0 + (AXIS == 0 ? 2 * t : (wave & 1) * 16 + t);
    const int oy = oy0 + (AXIS == 0 ? wave : 2 * (wave >> 1));
    if (ox >= W || oy >= H) return;
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cout = (grp * MBW + m) * 16 + (lane >> 4) * 4 + r;
            if (cout >= a.Cout) continue;
            const float bs = a.bias ? a.bias[cout] : 0.f;
            const float y0 = act1(((acc[0][m][r] + acc[1][m][r]) + acc[2][m][r]) + bs, a.act, a.p0);
            const float y1 = act1(((acc[1][m][r] - acc[2][m][r]) - acc[3][m][r]) + bs, a.act, a.p0);
            float* o = a.dst + ((long long)(b * a.Cout + cout) * H + oy) * W + ox;
            if (AXIS == 0) {
                *(float2*)o = make_float2(y0, y1);             // W % 4 == 0 and ox even: both columns exist
            } else {
                o[0] = y0;
                if (oy + 1 < H) o[W] = y1;
            }
        }
}

// ---- the larger Cook-Toom forms F(4, 3), F(2, 7), F(4, 7) (round 3, numerics checked first: oracle/numerics_study_winograd.py) -----------------
// Same skeleton as conv1d3_wino_kernel; what changes with (M outputs per tile, R taps), N = M + R - 1 positions:
//   AXIS 0 (1 x R): wave = output row, lane & 15 = tile of M columns -> workgroup = 8 rows x 16 M columns, no vertical halo;
//   AXIS 1 (R x 1): wave = (tile of M rows, half of the 32 columns) -> workgroup = 4 M rows x 32 columns, R - 1 halo rows;
//   4 halo columns either side (16-byte groups; >= (R - 1) / 2), channel-plane pitch by the width of the patch reads (CtGeom::PLANE); the transforms are the generated
//   straight-line chains of cooktoom_1d.h (dyadic coefficients), N accumulator sets per output-channel block.
template <int M, int R> struct CtForm;
template <> struct CtForm<4, 3> {
    static __device__ __forceinline__ void in(const float (&d)[6], float (&v)[6]) { ct_input_4_3(d, v); }
    static __device__ __forceinline__ void out(const float (&mm)[6], float (&y)[4]) { ct_output_4_3(mm, y); }
    static const double* g() { return &CT_G_4_3[0][0]; }
};
template <> struct CtForm<2, 7> {
    static __device__ __forceinline__ void in(const float (&d)[8], float (&v)[8]) { ct_input_2_7(d, v); }
    static __device__ __forceinline__ void out(const float (&mm)[8], float (&y)[2]) { ct_output_2_7(mm, y); }
    static const double* g() { return &CT_G_2_7[0][0]; }
};
template <> struct CtForm<4, 7> {
    static __device__ __forceinline__ void in(const float (&d)[10], float (&v)[10]) { ct_input_4_7(d, v); }
    static __device__ __forceinline__ void out(const float (&mm)[10], float (&y)[4]) { ct_output_4_7(mm, y); }
    static const double* g() { return &CT_G_4_7[0][0]; }
};

template <> struct CtForm<4, 4> {         // the 7-tap stride-2 layers: 4 taps over [even | odd] (cooktoom.stride2_as_stride1)
    static __device__ __forceinline__ void in(const float (&d)[7], float (&v)[7]) { ct_input_4_4(d, v); }
    static __device__ __forceinline__ void out(const float (&mm)[7], float (&y)[4]) { ct_output_4_4(mm, y); }
    static const double* g() { return &CT_G_4_4[0][0]; }
};

// floats of one chunk's U block in the packed stream: whole 1 KiB DMA pieces (an odd number of positions - F(4,4): 7 - is padded with zeros)
constexpr int ct_u_stream(int n, int mbw) { return (n * 2 * mbw * 64 + 255) / 256 * 256; }

template <int AXIS, int M, int R>
struct CtGeom {
    static constexpr int N = M + R - 1;
    static constexpr int PL = (R - 1) / 2;                       // 'same' padding of an odd filter at stride 1: (R - 1) / 2 either side
    static constexpr int RH = AXIS == 0 ? 8 : 4 * M;             // output rows / columns per workgroup
    static constexpr int RW = AXIS == 0 ? 16 * M : 32;
    static constexpr int ROWS = RH + (AXIS == 1 ? R - 1 : 0);    // raw region: rows oy0 - PL (axis 1) or oy0 (axis 0) ..., columns ox0 - 4 ...
    static constexpr int PITCH = RW + 8;
    static constexpr int G4 = PITCH / 4;
    static constexpr int NG = ROWS * G4;                         // 16-byte groups of one channel plane
    static constexpr int NI = (NG + 63) / 64;                    // DMA instructions per plane (1 KiB each)
    static constexpr int PLANE0 = ROWS * PITCH;
    // Plane pitch by what reads the patches (lane = 16 channel + tile; MI355X_MICROARCH.md, LDS; tools/lds_banks.py):
    //   AXIS 1: dword reads at channel * PLANE + tile, two groups of 32 lanes over 32 banks -> 16 mod 32;
    //   AXIS 0, M = 4: 16-byte reads at channel * PLANE + 4 tile, groups of 16 lanes ({0-3, 12-15, 20-27}, ...) over 64 banks: the two channels of a
    //           group fill the 16 slots of a 256-byte row exactly when PLANE = 0 mod 64 (rounds 3-4 used 16 mod 32 here too: two cycles per group);
    //   AXIS 0, M = 2: 8-byte reads at channel * PLANE + 2 tile, two groups of 32 lanes over 64 banks -> 32 mod 64.
    static constexpr int PMOD = AXIS == 1 ? 32 : 64, PREM = AXIS == 1 ? 16 : (M == 4 ? 0 : 32);
    static constexpr int PLANE = PLANE0 + ((PREM - PLANE0 % PMOD) + PMOD) % PMOD;
    static_assert(PL <= 4 && PITCH % 4 == 0 && PLANE % 4 == 0 && PLANE % PMOD == PREM && PLANE >= PLANE0, "raw region layout");
};

template <int AXIS, int MBW, int M, int R>
__global__ __launch_bounds__(512) void conv1d_ct_kernel(const W1KArgs a) {
    using Gm = CtGeom<AXIS, M, R>;
    constexpr int N = Gm::N, PITCH = Gm::PITCH, PLANE = Gm::PLANE, PL = Gm::PL;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int U_FLOATS = ct_u_stream(N, MBW);            // U fragments of one chunk: [p][c4][cout block][64 lanes] (+ zero padding to 1 KiB pieces)
    constexpr int U_PIECES = U_FLOATS / 256;                 // 1 KiB pieces
    constexpr int BUF = WCK * PLANE + U_FLOATS;
    static_assert(U_FLOATS % 256 == 0 && U_FLOATS >= N * 2 * MBW * 64, "U pieces");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    const int grp = blockIdx.y, b = blockIdx.z;
    const int oy0 = ty_wg * Gm::RH, ox0 = tx_wg * Gm::RW;
    const int H = a.H, W = a.W, HW = a.cplane;                // (HW: floats between channel planes as stored - H * W unless the source is a strided view)

    int voff4[Gm::NI];                                        // lane l owns the 16-byte groups r = l + 64 i of a plane: row r / G4, group r % G4
#pragma unroll
    for (int i = 0; i < Gm::NI; ++i) {
        const int r = lane + 64 * i;
        const int row = r / Gm::G4, g4 = r - row * Gm::G4;
        const int gy = oy0 - (AXIS == 1 ? PL : 0) + row, gx = ox0 - 4 + 4 * g4;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff4[i] = r < Gm::NG ? (inb ? (gy * a.pitch + gx) * 4 : -1) : -2;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
    const float* wgrp = a.w + (long long)grp * a.wgroup_stride;

    f32x4 acc[N][MBW];
#pragma unroll
    for (int p = 0; p < N; ++p)
#pragma unroll
        for (int m = 0; m < MBW; ++m) acc[p][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;
    auto issue = [&](int q, int pb) {
        const unsigned buf_addr = lds_base + pb * BUF * 4;
        const unsigned u_addr = buf_addr + WCK * PLANE * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS;
#pragma unroll
        for (int pc = 0; pc < (U_PIECES + 7) / 8; ++pc)
            if (wave + 8 * pc < U_PIECES) dma_global_x4(u_addr + (wave + 8 * pc) * 1024, wsrc + (wave + 8 * pc) * 256 + lane * 4);
        const i32x4 srd = make_srd(a.src[cs], a.src_bytes[cs]);
        const bool cok = cc0 + wave < a.src_c[cs];            // padded channels read as zero
        const int so = ((b * a.src_c[cs] + cc0 + (cok ? wave : 0)) * HW) * 4;
#pragma unroll
        for (int i = 0; i < Gm::NI; ++i)
            if (voff4[i] != -2) dma_buffer_x4(buf_addr + wave * (PLANE * 4) + i * 1024, cok ? voff4[i] : -1, srd, so);
        cc0 += WCK;
        if (cc0 >= a.src_cpad[cs]) { cc0 = 0; ++cs; }
    };

    issue(0, 0);
    const int t = lane & 15;
    // AXIS 0: the N inputs of tile t start at raw column 4 + M t - PL; they are read as aligned vectors from column M t on.
    // AXIS 1: input i of the tile is raw row M (wave >> 1) + i at column 4 + 16 (wave & 1) + t.
    const int patch0 = (lane >> 4) * PLANE + (AXIS == 0 ? wave * PITCH + M * t : (M * (wave >> 1)) * PITCH + 4 + (wave & 1) * 16 + t);
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF;
        const float* ub = raw + WCK * PLANE + lane;
        dma_wait_all();
        __syncthreads();                                      // raw + U of chunk q visible; everyone is done with the other buffer
        if (q + 1 < a.nchunks) issue(q + 1, pb ^ 1);
        float v[2][N];                                        // B operands of this lane: (B^T d)[p] of channels 4 c4 + (lane >> 4)
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const float* rp = raw + patch0 + c4 * 4 * PLANE;
            float d[N];
            if constexpr (AXIS == 1) {
#pragma unroll
                for (int i = 0; i < N; ++i) d[i] = rp[i * PITCH];
            } else if constexpr (M == 4) {                     // columns 4 t .. 4 t + 11 as three 16-byte reads, inputs from 4 - PL on
                float x[12];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const f32x4 g = *(const f32x4*)(rp + 4 * j);
                    x[4 * j] = g.x; x[4 * j + 1] = g.y; x[4 * j + 2] = g.z; x[4 * j + 3] = g.w;
                }
                static_assert(4 - PL + N <= 12, "three groups cover the patch");
#pragma unroll
                for (int i = 0; i < N; ++i) d[i] = x[4 - PL + i];
            } else {                                           // M == 2: columns 2 t .. as 8-byte reads
                constexpr int NV = (4 - PL + N + 1) / 2;
                float x[2 * NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const float2 g = *(const float2*)(rp + 2 * j);
                    x[2 * j] = g.x; x[2 * j + 1] = g.y;
                }
#pragma unroll
                for (int i = 0; i < N; ++i) d[i] = x[4 - PL + i];
            }
            CtForm<M, R>::in(d, v[c4]);
        }
#pragma unroll
        for (int p = 0; p < N; ++p)
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4)
#pragma unroll
                for (int m = 0; m < MBW; ++m) {
                    const float av = ub[((p * 2 + c4) * MBW + m) * 64];
                    acc[p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[c4][p], acc[p][m], 0, 0, 0);
                }
    }
    // ---- output transform y = A^T m per (cout, tile) in registers, epilogue ------------------------------------------------------
    const int ox = ox0 + (AXIS == 0 ? M * t : (wave & 1) * 16 + t);
    const int oy = oy0 + (AXIS == 0 ? wave : M * (wave >> 1));
    if (ox >= W || oy >= H) return;
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cout = (grp * MBW + m) * 16 + (lane >> 4) * 4 + r;
            if (cout >= a.Cout) continue;
            const float bs = a.bias ? a.bias[cout] : 0.f;
            float mm[N], y[M];
#pragma unroll
            for (int p = 0; p < N; ++p) mm[p] = acc[p][m][r];
            CtForm<M, R>::out(mm, y);
#pragma unroll
            for (int k = 0; k < M; ++k) y[k] = act1(y[k] + bs, a.act, a.p0);
            if constexpr (AXIS == 0) {                         // W % 4 == 0 and ox a multiple of M: all M columns exist
                float* o = a.dst + ((long long)(b * a.Cout + cout) * H + oy) * W + ox;
                if constexpr (M == 4) *(f32x4*)o = (f32x4){y[0], y[1], y[2], y[3]};
                else *(float2*)o = make_float2(y[0], y[1]);
            } else {
                // dense (batch, Cout, H, W), or - dst_split - the two column parities as two dense (batch, Cout, H, W / 2) tensors one after the
                // other: what the 1 x k stride-(1,2) half of a ConvReLU2 pair then reads as its [even | odd] sources
                const int wd = a.dst_split ? (W >> 1) : W;
                float* o = a.dst + (a.dst_split ? (long long)(ox & 1) * a.dst_half : 0ll) + ((long long)(b * a.Cout + cout) * H + oy) * wd + (a.dst_split ? (ox >> 1) : ox);
#pragma unroll
                for (int k = 0; k < M; ++k)
                    if (oy + k < H) o[(long long)k * wd] = y[k];
            }
        }
}

// ---- layers.Upconv (nearest x2 -> pad (0,1,0,1) -> conv 2x2; model/layers.py:349-356) with 4 multiplies per 2x2 output block ----------
// The four outputs of the block that input position (y, x) expands to read the 2x2 input patch a = in(y, x), b = in(y, x+1),
// c = in(y+1, x), d = in(y+1, x+1) (zero beyond the bottom / right edge = the pad):
//     o00 = a (w00+w01+w10+w11)      o01 = a (w00+w10) + b (w01+w11)      o10 = a (w00+w01) + c (w10+w11)      o11 = a w00 + b w01 + c w10 + d w11
// - 9 multiplies as the plan's phase decomposition runs them (16 in the reference).  Writing b = a + (b - a) etc. leaves 4:
//     V = [a, b - a; c - a, d - b - c + a],   U = [sum w, w01 + w11; w10 + w11, w11],   M = sum_cin U o V,
//     o00 = M00,  o01 = M00 + M01,  o10 = M00 + M10,  o11 = (M00 + M01) + (M10 + M11)
// (the 2-D tensor product of  out_even = a (g0 + g1), out_odd = out_even + (b - a) g1).  All transform coefficients are +-1.
// Same skeleton as the F(2,3) kernel above: workgroup = 8 waves, 8 x 32 INPUT positions (16 x 64 output pixels), 16 * MBW output
// channels; wave = input row, two blocks of 16 columns each; 4 positions -> acc[2][4][MBW].
template <int MBW>
__global__ __launch_bounds__(512) void upconv2x2_wino_kernel(const W1KArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int U_FLOATS = NPOS * 2 * MBW * 64;
    constexpr int BUF = WCK * RAW_PLANE + U_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    const int grp = blockIdx.y, b = blockIdx.z;
    const int iy0 = ty_wg * 8, ix0 = tx_wg * 32;
    const int H = a.H, W = a.W, HW = H * W;

    int voff4[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = lane + 64 * i;
        const int row = r / 10, g4 = r - row * 10;
        const int gy = iy0 - 1 + row, gx = ix0 - 4 + 4 * g4;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff4[i] = r < RAW_ROWS * 10 ? (inb ? (gy * W + gx) * 4 : -1) : -2;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
    const float* wgrp = a.w + (long long)grp * a.wgroup_stride;

    f32x4 acc[2][NPOS][MBW];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int p = 0; p < NPOS; ++p)
#pragma unroll
            for (int m = 0; m < MBW; ++m) acc[nb][p][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;
    auto issue = [&](int q, int pb) {
        const unsigned buf_addr = lds_base + pb * BUF * 4;
        const unsigned u_addr = buf_addr + WCK * RAW_PLANE * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS;
        if (wave < U_FLOATS / 256) dma_global_x4(u_addr + wave * 1024, wsrc + wave * 256 + lane * 4);
        const i32x4 srd = make_srd(a.src[cs], a.src_bytes[cs]);
        const bool cok = cc0 + wave < a.src_c[cs];
        const int so = ((b * a.src_c[cs] + cc0 + (cok ? wave : 0)) * HW) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (voff4[i] != -2) dma_buffer_x4(buf_addr + wave * (RAW_PLANE * 4) + i * 1024, cok ? voff4[i] : -1, srd, so);
        cc0 += WCK;
        if (cc0 >= a.src_cpad[cs]) { cc0 = 0; ++cs; }
    };

    issue(0, 0);
    const int t = lane & 15;
    const int patch0 = (lane >> 4) * RAW_PLANE + (wave + 1) * RAW_PITCH + 4 + t;      // input (iy0 + wave, ix0 + t) of channel lane >> 4
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF;
        const float* ub = raw + WCK * RAW_PLANE + lane;
        dma_wait_all();
        __syncthreads();
        if (q + 1 < a.nchunks) issue(q + 1, pb ^ 1);
        float v[2][2][NPOS];                                  // [column block][channel quad][position]
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4) {
                const float* rp = raw + patch0 + c4 * 4 * RAW_PLANE + nb * 16;
                const float pa = rp[0], pb_ = rp[1], pc = rp[RAW_PITCH], pd = rp[RAW_PITCH + 1];
                v[nb][c4][0] = pa;
                v[nb][c4][1] = pb_ - pa;
                v[nb][c4][2] = pc - pa;
                v[nb][c4][3] = (pd - pb_) - (pc - pa);
            }
#pragma unroll
        for (int p = 0; p < NPOS; ++p)
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4)
#pragma unroll
                for (int m = 0; m < MBW; ++m) {
                    const float av = ub[((p * 2 + c4) * MBW + m) * 64];
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc[nb][p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[nb][c4][p], acc[nb][p][m], 0, 0, 0);
                }
    }
    const int iy = iy0 + wave;
    if (iy >= H) return;
    const int OW = 2 * W;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int ix = ix0 + nb * 16 + t;
        if (ix >= W) continue;
#pragma unroll
        for (int m = 0; m < MBW; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = (grp * MBW + m) * 16 + (lane >> 4) * 4 + r;
                if (cout >= a.Cout) continue;
                const float bs = a.bias ? a.bias[cout] : 0.f;
                const float m00 = acc[nb][0][m][r], m01 = acc[nb][1][m][r], m10 = acc[nb][2][m][r], m11 = acc[nb][3][m][r];
                const float o01 = m00 + m01;
                float* o = a.dst + ((long long)(b * a.Cout + cout) * (2 * H) + 2 * iy) * OW + 2 * ix;
                *(float2*)o = make_float2(act1(m00 + bs, a.act, a.p0), act1(o01 + bs, a.act, a.p0));
                *(float2*)(o + OW) = make_float2(act1((m00 + m10) + bs, a.act, a.p0), act1((o01 + (m10 + m11)) + bs, a.act, a.p0));
            }
    }
}

bool valid_mbw1(int m) { return m >= 1 && m <= 4; }
int pad8(int c) { return (c + 7) & ~7; }

struct W1Derived {
    W1KArgs k;
    dim3 grid;
    size_t lds_bytes;
    int mbw;
};

int derive1(const mr_wino_desc* d, W1Derived* out, bool views = false) {
    if (!d || d->num_src < 1 || d->num_src > MR_MAX_SOURCES || d->batch < 1 || d->height < 1 || d->width < 4 || !d->dst ||
        !d->packed_weights || d->out_channels < 1)
        return MR_ERR_BAD_ARGUMENT;
    if (d->width % 4) return MR_ERR_UNSUPPORTED;              // 16-byte groups entirely inside or outside the image
    if (!views && (d->src_row_pitch || d->src_plane_floats || d->dst_split_columns)) return MR_ERR_UNSUPPORTED;   // mr_conv1d_cooktoom_f32 only
    if (d->src_row_pitch < 0 || d->src_plane_floats < 0) return MR_ERR_BAD_ARGUMENT;
    const int pitch = d->src_row_pitch ? d->src_row_pitch : d->width;
    const long long cplane = d->src_plane_floats ? d->src_plane_floats : (long long)d->height * d->width;
    if (pitch < d->width || (pitch & 3) || cplane < (long long)(d->height - 1) * pitch + d->width || cplane >= (1ll << 30)) return MR_ERR_BAD_ARGUMENT;
    if (d->dst_split_columns && (d->width & 7)) return MR_ERR_UNSUPPORTED;       // each parity half must keep rows of a multiple of 4 columns
    if (d->residual) return MR_ERR_UNSUPPORTED;
    if (!valid_mbw1(d->cout_blocks_per_wave)) return MR_ERR_BAD_ARGUMENT;
    if (d->activation != MR_ACT_NONE && d->activation != MR_ACT_RELU && d->activation != MR_ACT_LEAKY_RELU) return MR_ERR_UNSUPPORTED;
    if (d->activation == MR_ACT_LEAKY_RELU && !(d->act_p0 >= 0.f && d->act_p0 <= 1.f)) return MR_ERR_UNSUPPORTED;   // the epilogue is max(x, x * slope)
    W1KArgs& k = out->k;
    memset(&k, 0, sizeof(k));
    int nchunks = 0;
    for (int s = 0; s < d->num_src; ++s) {
        if (!d->src[s] || d->src_channels[s] < 1) return MR_ERR_BAD_ARGUMENT;
        // bytes the launch may address from src[s]: the stored tensor, less what a view that starts (pitch - width) floats into it leaves behind
        const long long bytes = ((long long)d->batch * d->src_channels[s] * cplane - (pitch - d->width)) * 4;
        if (bytes >= (1ll << 31) || bytes <= 0) return MR_ERR_UNSUPPORTED;
        k.src[s] = d->src[s];
        k.src_bytes[s] = (int)bytes;
        k.src_c[s] = d->src_channels[s];
        k.src_cpad[s] = pad8(d->src_channels[s]);
        nchunks += k.src_cpad[s] / WCK;
    }
    if ((long long)d->batch * d->out_channels * d->height * d->width * 4 >= (1ll << 33)) return MR_ERR_UNSUPPORTED;
    k.nsrc = d->num_src;
    k.H = d->height; k.W = d->width;
    k.pitch = pitch; k.cplane = (int)cplane;
    k.dst_split = d->dst_split_columns ? 1 : 0;
    k.dst_half = (long long)d->batch * d->out_channels * d->height * (d->width / 2);
    k.dst = d->dst; k.bias = d->bias;
    k.act = d->activation; k.p0 = d->act_p0;
    k.Cout = d->out_channels;
    k.tiles_x = (d->width + 31) / 32;
    k.nchunks = nchunks;
    k.w = d->packed_weights;
    const int mbw = d->cout_blocks_per_wave;
    const int ufl = NPOS * 2 * mbw * 64;
    k.wgroup_stride = (long long)nchunks * ufl;
    const int groups = (d->out_channels + 16 * mbw - 1) / (16 * mbw);
    if (d->batch >= 65536 || groups >= 65536) return MR_ERR_UNSUPPORTED;
    out->grid = dim3((unsigned)(k.tiles_x * ((d->height + 7) / 8)), (unsigned)groups, (unsigned)d->batch);
    out->lds_bytes = (size_t)(2 * (WCK * RAW_PLANE_X + ufl)) * 4;     // (the larger of the two axes' plane pitches; the Upconv kernel and AXIS 1 use 400)
    out->mbw = mbw;
    return 0;
}

template <int AXIS, int MBW>
int launch1(const W1Derived& dv, hipStream_t stream) {
    hipLaunchKernelGGL((conv1d3_wino_kernel<AXIS, MBW>), dv.grid, dim3(512), dv.lds_bytes, stream, dv.k);   // <= 42 KB of LDS: no attribute needed
    return (int)hipGetLastError();
}

template <int AXIS>
int launch1_mbw(const W1Derived& dv, hipStream_t stream) {
    switch (dv.mbw) {
        case 1: return launch1<AXIS, 1>(dv, stream);
        case 2: return launch1<AXIS, 2>(dv, stream);
        case 3: return launch1<AXIS, 3>(dv, stream);
        default: return launch1<AXIS, 4>(dv, stream);
    }
}


// ---- host side of the larger Cook-Toom forms ---------------------------------------------------------------------------------------
// F(2, 7) was never selected by a measured table (F(4, 7) halves the 7-tap layers, the direct kernel beats F(2, 7)): its instantiations
// are compiled into the diagnostic library only (python -m monorec_amd.build --timeline).
#ifdef MR_DIAGNOSTIC_FORMS
bool valid_form(int m, int r) { return (m == 4 && r == 3) || (m == 2 && r == 7) || (m == 4 && r == 7) || (m == 4 && r == 4); }
#else
bool valid_form(int m, int r) { return (m == 4 && r == 3) || (m == 4 && r == 7) || (m == 4 && r == 4); }
#endif
bool valid_ct_mbw(int m, int r, int mbw) { return mbw >= 1 && mbw <= (m + r - 1 >= 10 ? 3 : 4); }      // N x MBW accumulator sets in 256 VGPRs

template <int AXIS, int M, int R>
int derive_ct_form(const mr_wino_desc* d, W1Derived* out) {
    using Gm = CtGeom<AXIS, M, R>;
    const int rc = derive1(d, out, true);                     // argument checks, sources (strided views allowed), chunk count (geometry of F(2,3) overwritten below)
    if (rc != 0) return rc;
    if (d->dst_split_columns && AXIS != 1) return MR_ERR_UNSUPPORTED;
    const int mbw = d->cout_blocks_per_wave;
    const int ufl = ct_u_stream(Gm::N, mbw);
    out->k.tiles_x = (d->width + Gm::RW - 1) / Gm::RW;
    out->k.wgroup_stride = (long long)out->k.nchunks * ufl;
    out->grid = dim3((unsigned)(out->k.tiles_x * ((d->height + Gm::RH - 1) / Gm::RH)), out->grid.y, out->grid.z);
    out->lds_bytes = (size_t)(2 * (WCK * Gm::PLANE + ufl)) * 4;
    if (out->lds_bytes > 160 * 1024) return MR_ERR_LDS_BUDGET;
    return 0;
}

template <int AXIS, int MBW, int M, int R>
int launch_ct(const W1Derived& dv, hipStream_t stream) {
    static std::atomic<unsigned long long> attr_set{0};      // dynamic-LDS ceiling once per instantiation AND device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv1d_ct_kernel<AXIS, MBW, M, R>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv1d_ct_kernel<AXIS, MBW, M, R>), dv.grid, dim3(512), dv.lds_bytes, stream, dv.k);
    return (int)hipGetLastError();
}

template <int AXIS, int M, int R>
int run_ct_form(const mr_wino_desc* d, hipStream_t stream, bool launch, int64_t* lds) {
    if (!d || !valid_ct_mbw(M, R, d->cout_blocks_per_wave)) return MR_ERR_BAD_ARGUMENT;
    W1Derived dv;
    const int rc = derive_ct_form<AXIS, M, R>(d, &dv);
    if (rc != 0) return rc;
    if (lds) *lds = (int64_t)dv.lds_bytes;
    if (!launch) return 0;
    switch (dv.mbw) {
        case 1: return launch_ct<AXIS, 1, M, R>(dv, stream);
        case 2: return launch_ct<AXIS, 2, M, R>(dv, stream);
        case 3: return launch_ct<AXIS, 3, M, R>(dv, stream);
        default:
            if constexpr (M + R - 1 < 10) return launch_ct<AXIS, 4, M, R>(dv, stream);
            else return MR_ERR_BAD_ARGUMENT;
    }
}

int run_ct(const mr_wino_desc* d, int axis, int m, int r, hipStream_t stream, bool launch, int64_t* lds) {
    if (!valid_form(m, r) || (axis != 0 && axis != 1)) return MR_ERR_BAD_ARGUMENT;
    if (m == 4 && r == 3) return axis == 0 ? run_ct_form<0, 4, 3>(d, stream, launch, lds) : run_ct_form<1, 4, 3>(d, stream, launch, lds);
    if (m == 4 && r == 4) return axis == 0 ? run_ct_form<0, 4, 4>(d, stream, launch, lds) : run_ct_form<1, 4, 4>(d, stream, launch, lds);
#ifdef MR_DIAGNOSTIC_FORMS
    if (m == 2 && r == 7) return axis == 0 ? run_ct_form<0, 2, 7>(d, stream, launch, lds) : run_ct_form<1, 2, 7>(d, stream, launch, lds);
#endif
    return axis == 0 ? run_ct_form<0, 4, 7>(d, stream, launch, lds) : run_ct_form<1, 4, 7>(d, stream, launch, lds);
}

const double* form_g(int m, int r) { return m == 4 && r == 3 ? CtForm<4, 3>::g() : m == 4 && r == 4 ? CtForm<4, 4>::g() : m == 2 ? CtForm<2, 7>::g() : CtForm<4, 7>::g(); }

}  // namespace

extern "C" size_t mr_wino1d_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t mbw) {
    if (!src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_mbw1(mbw) || out_channels < 1) return 0;
    int nchunks = 0;
    for (int s = 0; s < num_src; ++s) nchunks += pad8(src_channels[s]) / WCK;
    const int groups = (out_channels + 16 * mbw - 1) / (16 * mbw);
    return (size_t)groups * nchunks * (NPOS * 2 * mbw * 64);
}

// weight: (out_channels, sum(src_channels), 3, 1) or (out_channels, sum(src_channels), 1, 3) fp32, nn.Conv2d layout - three taps per
// (cout, cin) either way.  U = G g in double, rounded once to fp32; stream order
// [cout group of 16 mbw][chunk (source-major, 8 channels)][position][channel quad][cout block of the group][64 lanes],
// lane l = (cout l & 15 of the block, channel l >> 4 of the quad).
extern "C" int mr_wino1d_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                          int32_t mbw, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_mbw1(mbw) || out_channels < 1)
        return MR_ERR_BAD_ARGUMENT;
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    const int groups = (out_channels + 16 * mbw - 1) / (16 * mbw);
    size_t o = 0;
    for (int g = 0; g < groups; ++g) {
        int cin_off = 0;
        for (int s = 0; s < num_src; ++s) {
            const int cpad = pad8(src_channels[s]);
            for (int c0 = 0; c0 < cpad; c0 += WCK)
                for (int p = 0; p < NPOS; ++p)
                    for (int c4 = 0; c4 < 2; ++c4)
                        for (int mb = 0; mb < mbw; ++mb)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int cout = (g * mbw + mb) * 16 + (lane & 15);
                                const int cl = c0 + c4 * 4 + (lane >> 4);
                                double u = 0.0;
                                if (cout < out_channels && cl < src_channels[s]) {
                                    const float* gw = weight + ((size_t)cout * cin_total + (cin_off + cl)) * 3;
                                    for (int i = 0; i < 3; ++i) u += G[p][i] * (double)gw[i];
                                }
                                dst[o++] = (float)u;
                            }
            cin_off += src_channels[s];
        }
    }
    return 0;
}

extern "C" int64_t mr_conv1d3_winograd_lds_bytes(const mr_wino_desc* desc) {
    W1Derived dv;
    const int rc = derive1(desc, &dv);
    return rc != 0 ? rc : (int64_t)dv.lds_bytes;
}

extern "C" int mr_conv1d3_winograd_f32(const mr_wino_desc* desc, int32_t axis, void* stream) {
    W1Derived dv;
    const int rc = derive1(desc, &dv);
    if (rc != 0) return rc;
    if (axis == 0) return launch1_mbw<0>(dv, (hipStream_t)stream);
    if (axis == 1) return launch1_mbw<1>(dv, (hipStream_t)stream);
    return MR_ERR_BAD_ARGUMENT;
}

// ---- F(4, 3), F(2, 7), F(4, 7): size, packer, launch ------------------------------------------------------------------------------------
extern "C" size_t mr_cooktoom1d_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t mbw,
                                                     int32_t m, int32_t r) {
    if (!src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_form(m, r) || !valid_ct_mbw(m, r, mbw) || out_channels < 1) return 0;
    int nchunks = 0;
    for (int s = 0; s < num_src; ++s) nchunks += pad8(src_channels[s]) / WCK;
    const int groups = (out_channels + 16 * mbw - 1) / (16 * mbw);
    return (size_t)groups * nchunks * ct_u_stream(m + r - 1, mbw);
}

// weight: (out_channels, sum(src_channels), r, 1) or (.., 1, r) fp32, nn.Conv2d layout - r taps per (cout, cin) either way.  U = G g in
// double (G: cooktoom_1d.h), rounded once to fp32; stream order as mr_wino1d_pack_weights_f32 with m + r - 1 positions.
extern "C" int mr_cooktoom1d_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                              int32_t mbw, int32_t m, int32_t r, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_form(m, r) || !valid_ct_mbw(m, r, mbw) ||
        out_channels < 1)
        return MR_ERR_BAD_ARGUMENT;
    const double* G = form_g(m, r);
    const int npos = m + r - 1;
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    const int groups = (out_channels + 16 * mbw - 1) / (16 * mbw);
    size_t o = 0;
    for (int g = 0; g < groups; ++g) {
        int cin_off = 0;
        for (int s = 0; s < num_src; ++s) {
            const int cpad = pad8(src_channels[s]);
            for (int c0 = 0; c0 < cpad; c0 += WCK) {
                for (int p = 0; p < npos; ++p)
                    for (int c4 = 0; c4 < 2; ++c4)
                        for (int mb = 0; mb < mbw; ++mb)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int cout = (g * mbw + mb) * 16 + (lane & 15);
                                const int cl = c0 + c4 * 4 + (lane >> 4);
                                double u = 0.0;
                                if (cout < out_channels && cl < src_channels[s]) {
                                    const float* gw = weight + ((size_t)cout * cin_total + (cin_off + cl)) * r;
                                    for (int i = 0; i < r; ++i) u += G[p * r + i] * (double)gw[i];
                                }
                                dst[o++] = (float)u;
                            }
                for (int pad = npos * 2 * mbw * 64; pad < ct_u_stream(npos, mbw); ++pad) dst[o++] = 0.f;      // whole 1 KiB pieces per chunk (odd npos)
            }
            cin_off += src_channels[s];
        }
    }
    return 0;
}

extern "C" int64_t mr_conv1d_cooktoom_lds_bytes(const mr_wino_desc* desc, int32_t axis, int32_t m, int32_t r) {
    int64_t lds = 0;
    const int rc = run_ct(desc, axis, m, r, nullptr, false, &lds);
    return rc != 0 ? rc : lds;
}

extern "C" int mr_conv1d_cooktoom_f32(const mr_wino_desc* desc, int32_t axis, int32_t m, int32_t r, void* stream) {
    return run_ct(desc, axis, m, r, (hipStream_t)stream, true, nullptr);
}

// layers.Upconv: weight (out_channels, sum(src_channels), 2, 2) fp32.  U = [sum w, w01 + w11; w10 + w11, w11] in double, rounded once;
// stream order as mr_wino1d_pack_weights_f32 (position p = 2 i + j).
extern "C" int mr_upconv_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                          int32_t mbw, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_mbw1(mbw) || out_channels < 1)
        return MR_ERR_BAD_ARGUMENT;
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    const int groups = (out_channels + 16 * mbw - 1) / (16 * mbw);
    size_t o = 0;
    for (int g = 0; g < groups; ++g) {
        int cin_off = 0;
        for (int s = 0; s < num_src; ++s) {
            const int cpad = pad8(src_channels[s]);
            for (int c0 = 0; c0 < cpad; c0 += WCK)
                for (int p = 0; p < NPOS; ++p)
                    for (int c4 = 0; c4 < 2; ++c4)
                        for (int mb = 0; mb < mbw; ++mb)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int cout = (g * mbw + mb) * 16 + (lane & 15);
                                const int cl = c0 + c4 * 4 + (lane >> 4);
                                double u = 0.0;
                                if (cout < out_channels && cl < src_channels[s]) {
                                    const float* gw = weight + ((size_t)cout * cin_total + (cin_off + cl)) * 4;     // w00 w01 w10 w11
                                    const double w00 = gw[0], w01 = gw[1], w10 = gw[2], w11 = gw[3];
                                    u = p == 0 ? ((w00 + w01) + (w10 + w11)) : p == 1 ? (w01 + w11) : p == 2 ? (w10 + w11) : w11;
                                }
                                dst[o++] = (float)u;
                            }
            cin_off += src_channels[s];
        }
    }
    return 0;
}

// desc: sources (batch, C_s, height, width) = the LOW-resolution input, dst = (batch, out_channels, 2 height, 2 width); bias and
// activation in the epilogue; packed_weights from mr_upconv_pack_weights_f32 (size: mr_wino1d_packed_weight_floats).
extern "C" int mr_upconv2x2_winograd_f32(const mr_wino_desc* desc, void* stream) {
    W1Derived dv;
    const int rc = derive1(desc, &dv);
    if (rc != 0) return rc;
    if ((long long)desc->batch * desc->out_channels * desc->height * desc->width * 16 >= (1ll << 33)) return MR_ERR_UNSUPPORTED;
    if (dv.mbw > 2) return MR_ERR_BAD_ARGUMENT;              // two column blocks per wave: accumulators of at most 32 channels
    if (dv.mbw == 1) hipLaunchKernelGGL((upconv2x2_wino_kernel<1>), dv.grid, dim3(512), dv.lds_bytes, (hipStream_t)stream, dv.k);
    else hipLaunchKernelGGL((upconv2x2_wino_kernel<2>), dv.grid, dim3(512), dv.lds_bytes, (hipStream_t)stream, dv.k);
    return (int)hipGetLastError();
}
