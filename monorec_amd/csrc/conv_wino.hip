// 3x3 stride-1 convolutions as Winograd F(2x2, 3x3) on the fp32 matrix cores of gfx950 (MI355X).
//
// The MaskModule is 3x3 convolutions throughout (reference model/monorec/monorec_model.py:296-343: ConvReLU = PadSameConv2d(3) +
// Conv2d(3) + LeakyReLU, model/layers.py:317-335) - 29.1 of the 61.07 GMAC of a c2 keyframe, 70.6 of 104 at c3 - and the big ones
// are MFMA-bound on the direct kernel (conv_mfma.hip: 105-120 TFLOP/s of the ~130 the sustained clock allows).  F(2x2, 3x3)
// computes a 2x2 output tile from a 4x4 input patch with 16 multiplies per (cin, cout) instead of 36:
//     Y = A^T [ sum_cin (G g G^T) o (B^T d B) ] A
// The 16 elementwise products are 16 independent GEMMs over cin - MFMA work: D_p[cout][tile] += U_p[cout][cin] * V_p[cin][tile].
//
// Workgroup = 8 waves, 8 x 32 output pixels = 4 x 16 tiles, 32 * MBW output channels, K walked in chunks of 8 input channels:
//   * the haloed input region (10 rows x 40 columns, 16-byte aligned) and the chunk's U fragments (host-packed, contiguous) come in
//     by LDS-DMA into one of two pipeline buffers (buffer_load_dwordx4 ... lds with hardware zero fill = padding, as in conv_mfma.hip);
//   * input transform: thread = (channel = wave, tile = lane): 16 LDS reads, 32 adds, 16 LDS writes into V[p][channel][tile]
//     (channel pitch 80 floats = 16 mod 32 banks: the MFMA B reads are conflict free);
//   * sweep: wave = (tile row tb = wave & 3, cout half = wave >> 2): for the 16 positions p and the 2 channel quads of the chunk one
//     B read + MBW A reads + MBW v_mfma_f32_16x16x4_f32 into acc[p][m] - the lane ends up holding all 16 positions of its (cout, tile),
//     so the output transform A^T M A runs in registers; bias / residual / activation / 8-byte stores follow.
// Two barriers per chunk (raw + U visible and V free; V visible); the DMA of chunk q + 1 is issued behind the transform of chunk q
// and lands during its sweep.  Products differ from the direct convolution by the rounding of the transforms (fp32 adds, weights
// transformed in fp64 and rounded once): ~1e-6 relative, far inside the 1e-4 bar; the summation over cin is an exact-order fmaf chain
// per position like every fp32 MFMA.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>

#include "../../include/monorec_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int WCK = 8;                                   // input channels per chunk (one per wave in the DMA / transform phases)
constexpr int RAW_PITCH = 40, RAW_ROWS = 10;             // floats: rows oy0-1 .. oy0+8, columns ox0-4 .. ox0+35
// Plane pitch 32 mod 64 (400 -> 416): the in-register-transform kernels read the patch of tile t - columns 2 t + 3 .. 2 t + 6, lane = 16 channel + t - as
// aligned 8-byte pairs from column 2 t + 2 on; a ds_read_b64 serves 32 lanes = two channels over 64 banks, conflict free when the second channel
// starts 32 banks on.  (Until round 5 these were dword reads at lane stride 2 with a pitch of 16 mod 32: the two channels of a group always met on
// the same 16 banks - an even pitch keeps the bank parity; tools/lds_banks.py.)  The LDS-transform kernel (channel = wave) does not care.
constexpr int RAW_PLANE = 416;
static_assert(RAW_PLANE >= RAW_PITCH * RAW_ROWS && RAW_PLANE % 64 == 32, "plane pitch");
constexpr int V_PITCH = 80;                              // floats per (position, channel): 64 tiles + 16
constexpr int V_FLOATS = 16 * WCK * V_PITCH;

struct WinoKArgs {
    const float* src[MR_MAX_SOURCES];
    int src_bytes[MR_MAX_SOURCES];
    int src_c[MR_MAX_SOURCES];
    int src_cpad[MR_MAX_SOURCES];       // padded to a multiple of WCK
    int nsrc;
    int H, W;
    float* dst;
    const float* bias;
    const float* res;
    int act;
    float p0;
    int Cout, tiles_x, nchunks;
    const float* w;
    long long wgroup_stride;            // packed floats per cout group
    int tail_grp;                       // variant 2 (in-register transform, 32 couts per workgroup): index of the 16-channel tail group
                                        // (out_channels % 32 in 1..16), handled by 16-row workgroups (wino_rb_tail), or -1
    int tiles_y16;                      // 16-row tile rows of the image (tail workgroups)
};

// ---- LDS-DMA through inline asm (see conv_mfma.hip: the builtins make hipcc drain vmcnt before every sweep) -------------------
// (the scalar operands go through readfirstlane: values derived from the wave index are uniform, but the compiler may hold them in
// VGPRs, which the "s" constraints of the asm do not fix up)
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
__device__ __forceinline__ float wino_activate(float v, int act, float p0) {
    const unsigned keep = act == MR_ACT_RELU ? 0u : ~0u;          // (an AND, not a select: a uniform select made hipcc clone the store loops)
    const float lo = __uint_as_float(__float_as_uint(v * (act == MR_ACT_LEAKY_RELU ? p0 : 1.f)) & keep);
    return fmaxf(v, lo);
}

template <int MBW>
__global__ __launch_bounds__(512) void conv3x3_wino_kernel(const WinoKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int U_FLOATS = 16 * 2 * (2 * MBW) * 64;        // U fragments of one chunk: [p][c4][cout block][64 lanes]
    constexpr int BUF = WCK * RAW_PLANE + U_FLOATS;          // one pipeline buffer: raw input region + U
    float* V = lds + 2 * BUF;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    const int grp = blockIdx.y, b = blockIdx.z;
    const int oy0 = ty_wg * 8, ox0 = tx_wg * 32;
    const int H = a.H, W = a.W, HW = H * W;

    // lane l owns the 16-byte groups r = l, l + 64 (< 100) of every channel plane of the region: row r / 10, group r % 10
    int voff4[2];
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

    f32x4 acc[16][MBW];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int m = 0; m < MBW; ++m) acc[p][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;                                      // chunk cursor: source, first channel
    auto issue = [&](int q, int pb) {
        const unsigned buf_addr = lds_base + pb * BUF * 4;
        const unsigned u_addr = buf_addr + WCK * RAW_PLANE * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS;
        for (int kb = wave; kb < U_FLOATS / 256; kb += 8) dma_global_x4(u_addr + kb * 1024, wsrc + kb * 256 + lane * 4);
        const i32x4 srd = make_srd(a.src[cs], a.src_bytes[cs]);
        const bool cok = cc0 + wave < a.src_c[cs];            // padded channels read as zero
        const int so = ((b * a.src_c[cs] + cc0 + (cok ? wave : 0)) * HW) * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (voff4[i] != -2) dma_buffer_x4(buf_addr + wave * (RAW_PLANE * 4) + i * 1024, cok ? voff4[i] : -1, srd, so);
        cc0 += WCK;
        if (cc0 >= a.src_cpad[cs]) { cc0 = 0; ++cs; }
    };

    issue(0, 0);
    const int tb = wave & 3, chalf = wave >> 2;
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF;
        const float* U = raw + WCK * RAW_PLANE;
        dma_wait_all();
        __syncthreads();                                      // raw + U of chunk q visible; everyone is done with V (sweep q - 1)
        {   // ---- input transform V = B^T d B: channel = wave, tile = lane (row lane >> 4, column lane & 15) -------------------
            const float* rp = raw + wave * RAW_PLANE + (2 * (lane >> 4)) * RAW_PITCH + 2 * (lane & 15) + 3;
            float d[4][4], t[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[r][c] = rp[r * RAW_PITCH + c];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = d[0][c] - d[2][c];
                t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c];
                t[3][c] = d[1][c] - d[3][c];
            }
            float* vp = V + wave * V_PITCH + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vp[(r * 4 + 0) * WCK * V_PITCH] = t[r][0] - t[r][2];
                vp[(r * 4 + 1) * WCK * V_PITCH] = t[r][1] + t[r][2];
                vp[(r * 4 + 2) * WCK * V_PITCH] = t[r][2] - t[r][1];
                vp[(r * 4 + 3) * WCK * V_PITCH] = t[r][1] - t[r][3];
            }
        }
        if (q + 1 < a.nchunks) issue(q + 1, pb ^ 1);          // lands during the sweep; buffer pb ^ 1 was last read before the barrier above
        __syncthreads();                                      // V visible
        {   // ---- sweep: 16 positions x 2 channel quads ------------------------------------------------------------------------
            const float* vb = V + (lane >> 4) * V_PITCH + tb * 16 + (lane & 15);
            const float* ub = U + (chalf * MBW) * 64 + lane;
#pragma unroll
            for (int p = 0; p < 16; ++p)
#pragma unroll
                for (int c4 = 0; c4 < 2; ++c4) {
                    const float bv = vb[(p * WCK + c4 * 4) * V_PITCH];
#pragma unroll
                    for (int m = 0; m < MBW; ++m) {
                        const float av = ub[((p * 2 + c4) * (2 * MBW) + m) * 64];
                        acc[p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[p][m], 0, 0, 0);
                    }
                }
        }
    }
    // ---- output transform Y = A^T M A per (cout, tile) in registers, epilogue ---------------------------------------------------
    const int ox = ox0 + 2 * (lane & 15);
    const int oyb = oy0 + 2 * tb;
    if (ox >= W) return;
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cout = grp * (32 * MBW) + (chalf * MBW + m) * 16 + (lane >> 4) * 4 + r;
            if (cout >= a.Cout) continue;
            float s0[4], s1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s0[c] = (acc[0 + c][m][r] + acc[4 + c][m][r]) + acc[8 + c][m][r];
                s1[c] = (acc[4 + c][m][r] - acc[8 + c][m][r]) - acc[12 + c][m][r];
            }
            float y[2][2];
            y[0][0] = (s0[0] + s0[1]) + s0[2];
            y[0][1] = (s0[1] - s0[2]) - s0[3];
            y[1][0] = (s1[0] + s1[1]) + s1[2];
            y[1][1] = (s1[1] - s1[2]) - s1[3];
            const float bs = a.bias ? a.bias[cout] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = oyb + i;
                if (oy >= H) continue;
                const long long idx = ((long long)(b * a.Cout + cout) * H + oy) * W + ox;
                float2 o;
                o.x = y[i][0] + bs;
                o.y = y[i][1] + bs;
                if (a.res) {
                    const float2 rv = *(const float2*)(a.res + idx);
                    o.x += rv.x;
                    o.y += rv.y;
                }
                o.x = wino_activate(o.x, a.act, a.p0);
                o.y = wino_activate(o.y, a.act, a.p0);
                *(float2*)(a.dst + idx) = o;
            }
        }
}


// ---- 16-channel tail of the in-register-transform kernel ------------------------------------------------------------------------------
// out_channels = 32 a + r with 0 < r <= 16 (the 48-channel layers of the MaskModule: monorec_model.py:300-313) would leave half of
// the last 32-channel workgroup multiplying zero weights.  The tail group is produced by workgroups of another shape instead: 16 x 32
// output pixels (8 tile rows, one per wave) x ONE block of 16 channels - the same 8 (tile row, channel block) units of work per
// workgroup, the same sweep (16 positions x 2 channel quads MFMAs per wave and chunk), one A read per MFMA.  Raw region 18 rows x 40
// columns (plane pitch 736 floats = 32 mod 64 banks, like 416), U fragments of a chunk 8 KiB.
constexpr int RAW_ROWS_T = 18, RAW_PLANE_T = 736;         // 720 -> 32 mod 64, as RAW_PLANE
static_assert(RAW_PLANE_T >= RAW_PITCH * RAW_ROWS_T && RAW_PLANE_T % 64 == 32, "plane pitch");
constexpr int U_FLOATS_T = 16 * 2 * 64;
constexpr int BUF_T = WCK * RAW_PLANE_T + U_FLOATS_T;

__device__ __forceinline__ void wino_rb_tail(const WinoKArgs& a, float* lds) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    if (ty_wg >= a.tiles_y16) return;                         // the grid is sized for the 8-row workgroups of the full groups
    const int b = blockIdx.z;
    const int oy0 = ty_wg * 16, ox0 = tx_wg * 32;
    const int H = a.H, W = a.W, HW = H * W;

    int voff4[3];                                             // lane l owns the 16-byte groups r = l + 64 i (< 180): row r / 10, group r % 10
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = lane + 64 * i;
        const int row = r / 10, g4 = r - row * 10;
        const int gy = oy0 - 1 + row, gx = ox0 - 4 + 4 * g4;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff4[i] = r < RAW_ROWS_T * 10 ? (inb ? (gy * W + gx) * 4 : -1) : -2;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
    const float* wgrp = a.w + (long long)a.tail_grp * a.wgroup_stride;

    f32x4 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;
    auto issue = [&](int q, int pb) {
        const unsigned buf_addr = lds_base + pb * BUF_T * 4;
        const unsigned u_addr = buf_addr + WCK * RAW_PLANE_T * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS_T;
        dma_global_x4(u_addr + wave * 1024, wsrc + wave * 256 + lane * 4);          // 8 pieces of 1 KiB, one per wave
        const i32x4 srd = make_srd(a.src[cs], a.src_bytes[cs]);
        const bool cok = cc0 + wave < a.src_c[cs];
        const int so = ((b * a.src_c[cs] + cc0 + (cok ? wave : 0)) * HW) * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (voff4[i] != -2) dma_buffer_x4(buf_addr + wave * (RAW_PLANE_T * 4) + i * 1024, cok ? voff4[i] : -1, srd, so);
        cc0 += WCK;
        if (cc0 >= a.src_cpad[cs]) { cc0 = 0; ++cs; }
    };

    issue(0, 0);
    const int tb = wave;                                      // tile row 0..7
    const int patch0 = (lane >> 4) * RAW_PLANE_T + (2 * tb) * RAW_PITCH + 2 * (lane & 15) + 2;      // aligned pairs; the patch starts one column on
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF_T;
        const float* ub = raw + WCK * RAW_PLANE_T + lane;
        dma_wait_all();
        __syncthreads();
        if (q + 1 < a.nchunks) issue(q + 1, pb ^ 1);
        float v[2][16];
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const float* rp = raw + patch0 + c4 * 4 * RAW_PLANE_T;
            float d[4][4], t[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 g0 = lds_pair(rp + r * RAW_PITCH), g1 = lds_pair(rp + r * RAW_PITCH + 2), g2 = lds_pair(rp + r * RAW_PITCH + 4);
                d[r][0] = g0.y; d[r][1] = g1.x; d[r][2] = g1.y; d[r][3] = g2.x;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = d[0][c] - d[2][c];
                t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c];
                t[3][c] = d[1][c] - d[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[c4][r * 4 + 0] = t[r][0] - t[r][2];
                v[c4][r * 4 + 1] = t[r][1] + t[r][2];
                v[c4][r * 4 + 2] = t[r][2] - t[r][1];
                v[c4][r * 4 + 3] = t[r][1] - t[r][3];
            }
        }
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4) {
                const float av = ub[(p * 2 + c4) * 64];
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[c4][p], acc[p], 0, 0, 0);
            }
    }
    const int ox = ox0 + 2 * (lane & 15);
    const int oyb = oy0 + 2 * tb;
    if (ox >= W) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cout = a.tail_grp * 32 + (lane >> 4) * 4 + r;
        if (cout >= a.Cout) continue;
        float s0[4], s1[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            s0[c] = (acc[0 + c][r] + acc[4 + c][r]) + acc[8 + c][r];
            s1[c] = (acc[4 + c][r] - acc[8 + c][r]) - acc[12 + c][r];
        }
        float y[2][2];
        y[0][0] = (s0[0] + s0[1]) + s0[2];
        y[0][1] = (s0[1] - s0[2]) - s0[3];
        y[1][0] = (s1[0] + s1[1]) + s1[2];
        y[1][1] = (s1[1] - s1[2]) - s1[3];
        const float bs = a.bias ? a.bias[cout] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int oy = oyb + i;
            if (oy >= H) continue;
            const long long idx = ((long long)(b * a.Cout + cout) * H + oy) * W + ox;
            float2 o;
            o.x = y[i][0] + bs;
            o.y = y[i][1] + bs;
            if (a.res) {
                const float2 rv = *(const float2*)(a.res + idx);
                o.x += rv.x;
                o.y += rv.y;
            }
            o.x = wino_activate(o.x, a.act, a.p0);
            o.y = wino_activate(o.y, a.act, a.p0);
            *(float2*)(a.dst + idx) = o;
        }
    }
}

// ---- variant with the input transform in registers (mr_wino_desc.variant = 1; the plan picks per layer shape by measurement) --------
// The B operand of the MFMA for (position p, channel quad c4) is V[p][4 c4 + (lane >> 4)][tile lane & 15] - so the lane that needs it
// can compute it itself: it reads the 4x4 patches of ITS two channels of the chunk at ITS tile from the raw region (32 LDS reads),
// transforms them (64 adds) and holds the 32 values as MFMA operands.  No V buffer (40 KB of LDS), no V round trip (16 writes + 32
// reads per lane and chunk), ONE barrier per chunk instead of two; the price is that the two waves of a tile row (the cout halves)
// both transform its patches.  Same products in the same order per accumulator as the kernel above: bit-identical outputs.
template <int MBW>
__global__ __launch_bounds__(512, MBW == 1 ? 4 : 2) void conv3x3_wino_rb_kernel(const WinoKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (MBW == 1 && (int)blockIdx.y == a.tail_grp) { wino_rb_tail(a, lds); return; }      // 16-channel tail group: 16-row workgroups
    constexpr int U_FLOATS = 16 * 2 * (2 * MBW) * 64;
    constexpr int BUF = WCK * RAW_PLANE + U_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    const int grp = blockIdx.y, b = blockIdx.z;
    const int oy0 = ty_wg * 8, ox0 = tx_wg * 32;
    const int H = a.H, W = a.W, HW = H * W;

    int voff4[2];
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

    f32x4 acc[16][MBW];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int m = 0; m < MBW; ++m) acc[p][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;
    auto issue = [&](int q, int pb) {
        const unsigned buf_addr = lds_base + pb * BUF * 4;
        const unsigned u_addr = buf_addr + WCK * RAW_PLANE * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS;
        for (int kb = wave; kb < U_FLOATS / 256; kb += 8) dma_global_x4(u_addr + kb * 1024, wsrc + kb * 256 + lane * 4);
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
    const int tb = wave & 3, chalf = wave >> 2;
    const int patch0 = (lane >> 4) * RAW_PLANE + (2 * tb) * RAW_PITCH + 2 * (lane & 15) + 2;   // channel lane >> 4, tile (tb, lane & 15): aligned pairs, the patch starts one column on
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF;
        const float* ub = raw + WCK * RAW_PLANE + (chalf * MBW) * 64 + lane;
        dma_wait_all();
        __syncthreads();                                      // raw + U of chunk q visible; everyone is done with the other buffer
        if (q + 1 < a.nchunks) issue(q + 1, pb ^ 1);
        float v[2][16];                                       // B operands of this lane: V[p] of channels 4 c4 + (lane >> 4)
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const float* rp = raw + patch0 + c4 * 4 * RAW_PLANE;
            float d[4][4], t[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 g0 = lds_pair(rp + r * RAW_PITCH), g1 = lds_pair(rp + r * RAW_PITCH + 2), g2 = lds_pair(rp + r * RAW_PITCH + 4);
                d[r][0] = g0.y; d[r][1] = g1.x; d[r][2] = g1.y; d[r][3] = g2.x;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                t[0][c] = d[0][c] - d[2][c];
                t[1][c] = d[1][c] + d[2][c];
                t[2][c] = d[2][c] - d[1][c];
                t[3][c] = d[1][c] - d[3][c];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[c4][r * 4 + 0] = t[r][0] - t[r][2];
                v[c4][r * 4 + 1] = t[r][1] + t[r][2];
                v[c4][r * 4 + 2] = t[r][2] - t[r][1];
                v[c4][r * 4 + 3] = t[r][1] - t[r][3];
            }
        }
#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4)
#pragma unroll
                for (int m = 0; m < MBW; ++m) {
                    const float av = ub[((p * 2 + c4) * (2 * MBW) + m) * 64];
                    acc[p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[c4][p], acc[p][m], 0, 0, 0);
                }
    }
    // ---- output transform Y = A^T M A per (cout, tile) in registers, epilogue (as above) ------------------------------------------
    const int ox = ox0 + 2 * (lane & 15);
    const int oyb = oy0 + 2 * tb;
    if (ox >= W) return;
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cout = grp * (32 * MBW) + (chalf * MBW + m) * 16 + (lane >> 4) * 4 + r;
            if (cout >= a.Cout) continue;
            float s0[4], s1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                s0[c] = (acc[0 + c][m][r] + acc[4 + c][m][r]) + acc[8 + c][m][r];
                s1[c] = (acc[4 + c][m][r] - acc[8 + c][m][r]) - acc[12 + c][m][r];
            }
            float y[2][2];
            y[0][0] = (s0[0] + s0[1]) + s0[2];
            y[0][1] = (s0[1] - s0[2]) - s0[3];
            y[1][0] = (s1[0] + s1[1]) + s1[2];
            y[1][1] = (s1[1] - s1[2]) - s1[3];
            const float bs = a.bias ? a.bias[cout] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int oy = oyb + i;
                if (oy >= H) continue;
                const long long idx = ((long long)(b * a.Cout + cout) * H + oy) * W + ox;
                float2 o;
                o.x = y[i][0] + bs;
                o.y = y[i][1] + bs;
                if (a.res) {
                    const float2 rv = *(const float2*)(a.res + idx);
                    o.x += rv.x;
                    o.y += rv.y;
                }
                o.x = wino_activate(o.x, a.act, a.p0);
                o.y = wino_activate(o.y, a.act, a.p0);
                *(float2*)(a.dst + idx) = o;
            }
        }
}

bool valid_mbw(int m) { return m == 1 || m == 2; }
int pad8(int c) { return (c + 7) & ~7; }

struct WinoDerived {
    WinoKArgs k;
    dim3 grid;
    size_t lds_bytes;
    int mbw;
    bool regb;
};

int wino_derive(const mr_wino_desc* d, WinoDerived* out) {
    if (!d || d->num_src < 1 || d->num_src > MR_MAX_SOURCES || d->batch < 1 || d->height < 1 || d->width < 4 || !d->dst ||
        !d->packed_weights || d->out_channels < 1)
        return MR_ERR_BAD_ARGUMENT;
    if (d->width % 4) return MR_ERR_UNSUPPORTED;              // 16-byte groups entirely inside or outside the image
    if (!valid_mbw(d->cout_blocks_per_wave)) return MR_ERR_BAD_ARGUMENT;
    if (d->activation != MR_ACT_NONE && d->activation != MR_ACT_RELU && d->activation != MR_ACT_LEAKY_RELU) return MR_ERR_UNSUPPORTED;
    if (d->activation == MR_ACT_LEAKY_RELU && !(d->act_p0 >= 0.f && d->act_p0 <= 1.f)) return MR_ERR_UNSUPPORTED;   // the epilogue is max(x, x * slope)
    if (d->src_row_pitch || d->src_plane_floats || d->dst_split_columns) return MR_ERR_UNSUPPORTED;      // strided views: mr_conv1d_cooktoom_f32 only
    WinoKArgs& k = out->k;
    memset(&k, 0, sizeof(k));
    int nchunks = 0;
    for (int s = 0; s < d->num_src; ++s) {
        if (!d->src[s] || d->src_channels[s] < 1) return MR_ERR_BAD_ARGUMENT;
        const long long bytes = (long long)d->batch * d->src_channels[s] * d->height * d->width * 4;
        if (bytes >= (1ll << 31)) return MR_ERR_UNSUPPORTED;
        k.src[s] = d->src[s];
        k.src_bytes[s] = (int)bytes;
        k.src_c[s] = d->src_channels[s];
        k.src_cpad[s] = pad8(d->src_channels[s]);
        nchunks += k.src_cpad[s] / WCK;
    }
    if ((long long)d->batch * d->out_channels * d->height * d->width * 4 >= (1ll << 33)) return MR_ERR_UNSUPPORTED;
    k.nsrc = d->num_src;
    k.H = d->height; k.W = d->width;
    k.dst = d->dst; k.bias = d->bias; k.res = d->residual;
    k.act = d->activation; k.p0 = d->act_p0;
    k.Cout = d->out_channels;
    k.tiles_x = (d->width + 31) / 32;
    k.nchunks = nchunks;
    k.w = d->packed_weights;
    const int mbw = d->cout_blocks_per_wave;
    const int ufl = 16 * 2 * (2 * mbw) * 64;
    k.wgroup_stride = (long long)nchunks * ufl;
    const int groups = (d->out_channels + 32 * mbw - 1) / (32 * mbw);
    if (d->batch >= 65536 || groups >= 65536) return MR_ERR_UNSUPPORTED;
    out->grid = dim3((unsigned)(k.tiles_x * ((d->height + 7) / 8)), (unsigned)groups, (unsigned)d->batch);
    if (d->variant < 0 || d->variant > 2) return MR_ERR_BAD_ARGUMENT;
    const bool regb = d->variant >= 1;                                          // the in-register-transform variant (see there)
    out->regb = regb;
    out->lds_bytes = (size_t)(2 * (WCK * RAW_PLANE + ufl) + (regb ? 0 : V_FLOATS)) * 4;
    out->mbw = mbw;
    k.tail_grp = -1;
    k.tiles_y16 = (d->height + 15) / 16;
    if (d->variant == 2) {                                                      // 32 a + (1..16) channels: the tail by 16-row workgroups
        const int rem = d->out_channels % 32;
        if (mbw != 1 || rem < 1 || rem > 16) return MR_ERR_BAD_ARGUMENT;
        k.tail_grp = d->out_channels / 32;                                      // = groups - 1
        const size_t tail_lds = (size_t)2 * BUF_T * 4;
        if (tail_lds > out->lds_bytes) out->lds_bytes = tail_lds;
    }
    return 0;
}

template <int MBW, bool REGB>
int wino_launch(const WinoDerived& dv, hipStream_t stream) {
    static std::atomic<unsigned long long> attr_set{0};      // dynamic-LDS ceiling once per instantiation AND device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    const void* fn = REGB ? reinterpret_cast<const void*>(&conv3x3_wino_rb_kernel<MBW>) : reinterpret_cast<const void*>(&conv3x3_wino_kernel<MBW>);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    if (REGB) hipLaunchKernelGGL(conv3x3_wino_rb_kernel<MBW>, dv.grid, dim3(512), dv.lds_bytes, stream, dv.k);
    else hipLaunchKernelGGL(conv3x3_wino_kernel<MBW>, dv.grid, dim3(512), dv.lds_bytes, stream, dv.k);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" size_t mr_wino_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t mbw) {
    if (!src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_mbw(mbw) || out_channels < 1) return 0;
    int nchunks = 0;
    for (int s = 0; s < num_src; ++s) nchunks += pad8(src_channels[s]) / WCK;
    const int groups = (out_channels + 32 * mbw - 1) / (32 * mbw);
    return (size_t)groups * nchunks * (16 * 2 * (2 * mbw) * 64);
}

// weight: (out_channels, sum(src_channels), 3, 3) fp32, nn.Conv2d layout.  U = G g G^T in double, rounded once to fp32; stream order
// [cout group][chunk (source-major, 8 channels)][position p = 4a + b][channel quad][cout block of the group][64 lanes], lane l =
// (cout l & 15 of the block, channel l >> 4 of the quad) - one contiguous block per (group, chunk), read lane-linearly.
extern "C" int mr_wino_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                        int32_t mbw, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_mbw(mbw) || out_channels < 1)
        return MR_ERR_BAD_ARGUMENT;
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    const int groups = (out_channels + 32 * mbw - 1) / (32 * mbw);
    size_t o = 0;
    for (int g = 0; g < groups; ++g) {
        int cin_off = 0;
        for (int s = 0; s < num_src; ++s) {
            const int cpad = pad8(src_channels[s]);
            for (int c0 = 0; c0 < cpad; c0 += WCK)
                for (int p = 0; p < 16; ++p)
                    for (int c4 = 0; c4 < 2; ++c4)
                        for (int mb = 0; mb < 2 * mbw; ++mb)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int cout = g * 32 * mbw + mb * 16 + (lane & 15);
                                const int cl = c0 + c4 * 4 + (lane >> 4);
                                double u = 0.0;
                                if (cout < out_channels && cl < src_channels[s]) {
                                    const float* gw = weight + ((size_t)cout * cin_total + (cin_off + cl)) * 9;
                                    const int pa = p >> 2, pb = p & 3;
                                    for (int i = 0; i < 3; ++i)
                                        for (int j = 0; j < 3; ++j) u += G[pa][i] * (double)gw[i * 3 + j] * G[pb][j];
                                }
                                dst[o++] = (float)u;
                            }
            cin_off += src_channels[s];
        }
    }
    return 0;
}

// Variant 2 (out_channels = 32 a + r, 0 < r <= 16): the a full groups exactly as mr_wino_pack_weights_f32(mbw = 1) packs them, then the
// tail group as [chunk][position][channel quad][64 lanes] (one block of 16 channels), lane l = (cout 32 a + (l & 15), channel l >> 4).
extern "C" size_t mr_wino_packed_weight_floats_tail(int32_t out_channels, const int32_t* src_channels, int32_t num_src) {
    if (!src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || out_channels < 1) return 0;
    const int rem = out_channels % 32;
    if (rem < 1 || rem > 16) return 0;
    int nchunks = 0;
    for (int s = 0; s < num_src; ++s) nchunks += pad8(src_channels[s]) / WCK;
    return (size_t)(out_channels / 32) * nchunks * (16 * 2 * 2 * 64) + (size_t)nchunks * U_FLOATS_T;
}

extern "C" int mr_wino_pack_weights_tail_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src, float* dst) {
    if (!weight || !dst || mr_wino_packed_weight_floats_tail(out_channels, src_channels, num_src) == 0) return MR_ERR_BAD_ARGUMENT;
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int full = out_channels / 32;
    if (full > 0) {                                      // the full groups: the standard stream of their 32 a channels, cin layout unchanged
        // (weights of output channel c start at c * cin_total * 9, so the first 32 a channels are a prefix of the tensor)
        const int rc = mr_wino_pack_weights_f32(weight, full * 32, src_channels, num_src, 1, dst);
        if (rc != 0) return rc;
    }
    int cin_total = 0, nchunks = 0;
    for (int s = 0; s < num_src; ++s) { cin_total += src_channels[s]; nchunks += pad8(src_channels[s]) / WCK; }
    size_t o = (size_t)full * nchunks * (16 * 2 * 2 * 64);
    int cin_off = 0;
    for (int s = 0; s < num_src; ++s) {
        const int cpad = pad8(src_channels[s]);
        for (int c0 = 0; c0 < cpad; c0 += WCK)
            for (int p = 0; p < 16; ++p)
                for (int c4 = 0; c4 < 2; ++c4)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int cout = full * 32 + (lane & 15);
                        const int cl = c0 + c4 * 4 + (lane >> 4);
                        double u = 0.0;
                        if (cout < out_channels && cl < src_channels[s]) {
                            const float* gw = weight + ((size_t)cout * cin_total + (cin_off + cl)) * 9;
                            const int pa = p >> 2, pb = p & 3;
                            for (int i = 0; i < 3; ++i)
                                for (int j = 0; j < 3; ++j) u += G[pa][i] * (double)gw[i * 3 + j] * G[pb][j];
                        }
                        dst[o++] = (float)u;
                    }
        cin_off += src_channels[s];
    }
    return 0;
}

extern "C" int64_t mr_conv3x3_winograd_lds_bytes(const mr_wino_desc* desc) {
    WinoDerived dv;
    const int rc = wino_derive(desc, &dv);
    return rc != 0 ? rc : (int64_t)dv.lds_bytes;
}

extern "C" int mr_conv3x3_winograd_f32(const mr_wino_desc* desc, void* stream) {
    WinoDerived dv;
    const int rc = wino_derive(desc, &dv);
    if (rc != 0) return rc;
    if (dv.regb) return dv.mbw == 2 ? wino_launch<2, true>(dv, (hipStream_t)stream) : wino_launch<1, true>(dv, (hipStream_t)stream);   // (variant 2: mbw 1)
    return dv.mbw == 2 ? wino_launch<2, false>(dv, (hipStream_t)stream) : wino_launch<1, false>(dv, (hipStream_t)stream);
}
