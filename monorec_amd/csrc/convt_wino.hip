// ConvTranspose2d(k = 4, s = 2) + centre crop (layers.Refine, reference model/layers.py:380-400; the four decoder stages of the
// DepthModule, model/monorec/monorec_model.py:503-513) as Winograd F(2x2, 2x2) on the fp32 matrix cores of gfx950 (MI355X).
//
// The transposed convolution is four 2x2 stride-1 convolutions on the LOW-resolution input, one per output parity (py, px):
//     out[2y + py][2x + px] = sum_c sum_{t,u in {0,1}} Wp[py,px][cout][c][t][u] * in[c][y - pt + t][x - pl + u],   pt = 1 - py, pl = 1 - px
// (engine.transposed_phase_weights).  F(2x2, 2x2) computes a 2x2 tile of one parity from a 3x3 input patch with 9 multiplies per
// (cin, cout) instead of 16:  Y = A^T [ sum_cin (G g G^T) o (B^T d B) ] A  with
//     B^T = [1 -1 0; 0 1 0; 0 -1 1],  G = [1 0; 1 1; 0 1],  A^T = [1 1 0; 0 1 1]
// - every coefficient is 0 or +-1, so the transforms add no scaling error at all.  The 9 products are 9 independent GEMMs over cin.
//
// Same skeleton as conv_wino.hip: workgroup = 8 waves, one parity (grid z), 8 x 32 parity outputs (= input positions) = 4 x 16 tiles,
// 32 * MBW output channels (MBW = 1, 2, 4), K in chunks of 8 input channels: raw region (10 rows x 40 columns) and the chunk's U
// fragments by LDS-DMA into one of two pipeline buffers; transform thread = (channel = wave, tile = lane) into V[9][channel][tile];
// sweep wave = (tile row, cout half): 9 positions x 2 channel quads x MBW MFMAs; the lane ends up with the 9 positions of its
// (cout, tile): output transform in registers, bias, LeakyReLU, stores with the parity's stride-2 placement.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <atomic>

#include "../../include/monorec_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WCK = 8;
constexpr int RAW_PITCH = 40, RAW_ROWS = 10, RAW_PLANE = RAW_PITCH * RAW_ROWS;   // rows y0 - pt .. (9 used), columns x0 - 4 .. x0 + 35
// (The in-register-transform kernels below read their patches as dwords at lane stride 2: the two channels of a 32-lane group meet on the same 16 banks
// - tools/lds_banks.py.  The fix that pays in conv_wino.hip - aligned 8-byte reads on a 32-mod-64 pitch - was 2-8 % SLOWER here (tools/sessions/r05_s14.sh:
// two pairs cover 4 columns and the 3 wanted ones sit at a parity-dependent offset, a select per element); not kept.)
constexpr int V_PITCH = 80;                                                      // 64 tiles + 16: channel pitch = 16 mod 32 banks
constexpr int NPOS = 9;
constexpr int V_FLOATS = NPOS * WCK * V_PITCH;

struct WinoTKArgs {
    const float* src[MR_MAX_SOURCES];
    int src_bytes[MR_MAX_SOURCES];
    int src_c[MR_MAX_SOURCES];
    int src_cpad[MR_MAX_SOURCES];
    int nsrc;
    int H, W;                           // input plane; the output plane is 2H x 2W
    float* dst;
    const float* bias;
    int act;
    float p0;
    int Cout, tiles_x, nchunks, ngroups;
    const float* w;                     // [phase][cout group][chunk][9][2][2 MBW][64]
    long long wgroup_stride, wphase_stride;
    int tail_grp;                       // variant 2: index of the 16-channel tail group (out_channels % 32 in 1..16) or -1, see convt_rb_tail
    int tiles_y16;
};

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
__device__ __forceinline__ void dma_global_x1(unsigned lds_byte_addr, const float* g) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte_addr), "v"(g) : "memory");
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
__device__ __forceinline__ float act_t(float v, int act, float p0) {
    const unsigned keep = act == MR_ACT_RELU ? 0u : ~0u;          // (an AND, not a select: a uniform select made hipcc clone the store loops)
    const float lo = __uint_as_float(__float_as_uint(v * (act == MR_ACT_LEAKY_RELU ? p0 : 1.f)) & keep);
    return fmaxf(v, lo);
}

template <int MBW>
__global__ __launch_bounds__(512) void convt4x4_wino_kernel(const WinoTKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int U_FLOATS = NPOS * 2 * (2 * MBW) * 64;       // U fragments of one chunk: [p][c4][cout block][64 lanes]
    constexpr int U_PAD = (U_FLOATS + 255) & ~255;            // pipeline buffers stay 1 KiB aligned
    constexpr int BUF = WCK * RAW_PLANE + U_PAD;
    float* V = lds + 2 * BUF;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    const int grp = blockIdx.y;
    const int ph = (int)blockIdx.z & 3, b = (int)blockIdx.z >> 2;
    const int py = ph >> 1, px = ph & 1;
    const int pt = 1 - py, pl = 1 - px;                       // rows y - pt + {0, 1}, columns x - pl + {0, 1}
    const int y0 = ty_wg * 8, x0 = tx_wg * 32;                // first input position (= parity output) of the workgroup
    const int H = a.H, W = a.W, HW = H * W;

    int voff4[2];                                             // lane l owns the 16-byte groups r = l, l + 64 (< 100): row r / 10, group r % 10
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = lane + 64 * i;
        const int row = r / 10, g4 = r - row * 10;
        const int gy = y0 - pt + row, gx = x0 - 4 + 4 * g4;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff4[i] = r < RAW_ROWS * 10 ? (inb ? (gy * W + gx) * 4 : -1) : -2;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
    const float* wgrp = a.w + (long long)ph * a.wphase_stride + (long long)grp * a.wgroup_stride;

    f32x4 acc[NPOS][MBW];
#pragma unroll
    for (int p = 0; p < NPOS; ++p)
#pragma unroll
        for (int m = 0; m < MBW; ++m) acc[p][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;
    auto issue = [&](int q, int pb) {
        const unsigned buf_addr = lds_base + pb * BUF * 4;
        const unsigned u_addr = buf_addr + WCK * RAW_PLANE * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS;
        constexpr int N1K = U_FLOATS / 256;                   // whole 1 KiB pieces, then 256-byte pieces
        for (int kb = wave; kb < N1K; kb += 8) dma_global_x4(u_addr + kb * 1024, wsrc + kb * 256 + lane * 4);
        for (int fr = N1K * 4 + wave; fr < U_FLOATS / 64; fr += 8) dma_global_x1(u_addr + fr * 256, wsrc + fr * 64 + lane);
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
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF;
        const float* U = raw + WCK * RAW_PLANE;
        dma_wait_all();
        __syncthreads();
        {   // ---- input transform V = B^T d B of the 3x3 patch at rows 2 ty .., columns 2 tx + (4 - pl) .. of the region ------------
            const float* rp = raw + wave * RAW_PLANE + (2 * (lane >> 4)) * RAW_PITCH + 2 * (lane & 15) + 4 - pl;
            float d[3][3], t[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) d[r][c] = rp[r * RAW_PITCH + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                t[0][c] = d[0][c] - d[1][c];
                t[1][c] = d[1][c];
                t[2][c] = d[2][c] - d[1][c];
            }
            float* vp = V + wave * V_PITCH + lane;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                vp[(r * 3 + 0) * WCK * V_PITCH] = t[r][0] - t[r][1];
                vp[(r * 3 + 1) * WCK * V_PITCH] = t[r][1];
                vp[(r * 3 + 2) * WCK * V_PITCH] = t[r][2] - t[r][1];
            }
        }
        if (q + 1 < a.nchunks) issue(q + 1, pb ^ 1);
        __syncthreads();
        {   // ---- sweep: 9 positions x 2 channel quads x MBW cout blocks ----------------------------------------------------------
            const float* vb = V + (lane >> 4) * V_PITCH + tb * 16 + (lane & 15);
            const float* ub = U + (chalf * MBW) * 64 + lane;
#pragma unroll
            for (int p = 0; p < NPOS; ++p)
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
    // ---- output transform Y = A^T M A, epilogue: parity outputs (y, x) land at (2 y + py, 2 x + px) -------------------------------
    const int xo = x0 + 2 * (lane & 15);
    const int yb = y0 + 2 * tb;
    if (xo >= W) return;
    const int OW = 2 * W;
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cout = grp * (32 * MBW) + (chalf * MBW + m) * 16 + (lane >> 4) * 4 + r;
            if (cout >= a.Cout) continue;
            float s0[3], s1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                s0[c] = acc[0 + c][m][r] + acc[3 + c][m][r];
                s1[c] = acc[3 + c][m][r] + acc[6 + c][m][r];
            }
            float y[2][2];
            y[0][0] = s0[0] + s0[1];
            y[0][1] = s0[1] + s0[2];
            y[1][0] = s1[0] + s1[1];
            y[1][1] = s1[1] + s1[2];
            const float bs = a.bias ? a.bias[cout] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yo = yb + i;
                if (yo >= H) continue;
                float* o = a.dst + ((long long)(b * a.Cout + cout) * (2 * H) + (2 * yo + py)) * OW + 2 * xo + px;
                o[0] = act_t(y[i][0] + bs, a.act, a.p0);
                if (xo + 1 < W) o[2] = act_t(y[i][1] + bs, a.act, a.p0);
            }
        }
}

// ---- 16-channel tail of the in-register-transform kernel (variant 2; as wino_rb_tail in conv_wino.hip) ---------------------------------
// out_channels = 32 a + r, 0 < r <= 16 (depth.dec3: 48): the tail channels come from workgroups of 16 x 32 input positions x ONE block of
// 16 channels (8 tile rows, one per wave) instead of a half-empty 32-channel group.  Raw region 18 rows x 40 columns, U of a chunk 4.5 KiB.
constexpr int RAW_ROWS_TT = 18, RAW_PLANE_TT = RAW_PITCH * RAW_ROWS_TT;
constexpr int U_FLOATS_TT = NPOS * 2 * 64;                   // 1152
constexpr int U_PAD_TT = (U_FLOATS_TT + 255) & ~255;         // 1280
constexpr int BUF_TT = WCK * RAW_PLANE_TT + U_PAD_TT;

__device__ __forceinline__ void convt_rb_tail(const WinoTKArgs& a, float* lds) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    if (ty_wg >= a.tiles_y16) return;
    const int ph = (int)blockIdx.z & 3, b = (int)blockIdx.z >> 2;
    const int py = ph >> 1, px = ph & 1;
    const int pt = 1 - py, pl = 1 - px;
    const int y0 = ty_wg * 16, x0 = tx_wg * 32;
    const int H = a.H, W = a.W, HW = H * W;

    int voff4[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = lane + 64 * i;
        const int row = r / 10, g4 = r - row * 10;
        const int gy = y0 - pt + row, gx = x0 - 4 + 4 * g4;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff4[i] = r < RAW_ROWS_TT * 10 ? (inb ? (gy * W + gx) * 4 : -1) : -2;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
    // stream: [phase][a full groups (MBW = 1 layout)][tail group: [chunk][9][2][64]]
    const float* wgrp = a.w + (long long)ph * a.wphase_stride + (long long)a.tail_grp * a.wgroup_stride;

    f32x4 acc[NPOS];
#pragma unroll
    for (int p = 0; p < NPOS; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;
    auto issue = [&](int q, int pb) {
        const unsigned buf_addr = lds_base + pb * BUF_TT * 4;
        const unsigned u_addr = buf_addr + WCK * RAW_PLANE_TT * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS_TT;
        constexpr int N1K = U_FLOATS_TT / 256;               // 4 pieces of 1 KiB, then 2 of 256 bytes
        if (wave < N1K) dma_global_x4(u_addr + wave * 1024, wsrc + wave * 256 + lane * 4);
        for (int fr = N1K * 4 + wave; fr < U_FLOATS_TT / 64; fr += 8) dma_global_x1(u_addr + fr * 256, wsrc + fr * 64 + lane);
        const i32x4 srd = make_srd(a.src[cs], a.src_bytes[cs]);
        const bool cok = cc0 + wave < a.src_c[cs];
        const int so = ((b * a.src_c[cs] + cc0 + (cok ? wave : 0)) * HW) * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (voff4[i] != -2) dma_buffer_x4(buf_addr + wave * (RAW_PLANE_TT * 4) + i * 1024, cok ? voff4[i] : -1, srd, so);
        cc0 += WCK;
        if (cc0 >= a.src_cpad[cs]) { cc0 = 0; ++cs; }
    };

    issue(0, 0);
    const int tb = wave;
    const int patch0 = (lane >> 4) * RAW_PLANE_TT + (2 * tb) * RAW_PITCH + 2 * (lane & 15) + 4 - pl;
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF_TT;
        const float* ub = raw + WCK * RAW_PLANE_TT + lane;
        dma_wait_all();
        __syncthreads();
        if (q + 1 < a.nchunks) issue(q + 1, pb ^ 1);
        float v[2][NPOS];
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const float* rp = raw + patch0 + c4 * 4 * RAW_PLANE_TT;
            float d[3][3], t[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) d[r][c] = rp[r * RAW_PITCH + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                t[0][c] = d[0][c] - d[1][c];
                t[1][c] = d[1][c];
                t[2][c] = d[2][c] - d[1][c];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                v[c4][r * 3 + 0] = t[r][0] - t[r][1];
                v[c4][r * 3 + 1] = t[r][1];
                v[c4][r * 3 + 2] = t[r][2] - t[r][1];
            }
        }
#pragma unroll
        for (int p = 0; p < NPOS; ++p)
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4) {
                const float av = ub[(p * 2 + c4) * 64];
                acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[c4][p], acc[p], 0, 0, 0);
            }
    }
    const int xo = x0 + 2 * (lane & 15);
    const int yb = y0 + 2 * tb;
    if (xo >= W) return;
    const int OW = 2 * W;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cout = a.tail_grp * 32 + (lane >> 4) * 4 + r;
        if (cout >= a.Cout) continue;
        float s0[3], s1[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s0[c] = acc[0 + c][r] + acc[3 + c][r];
            s1[c] = acc[3 + c][r] + acc[6 + c][r];
        }
        float y[2][2];
        y[0][0] = s0[0] + s0[1];
        y[0][1] = s0[1] + s0[2];
        y[1][0] = s1[0] + s1[1];
        y[1][1] = s1[1] + s1[2];
        const float bs = a.bias ? a.bias[cout] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int yo = yb + i;
            if (yo >= H) continue;
            float* o = a.dst + ((long long)(b * a.Cout + cout) * (2 * H) + (2 * yo + py)) * OW + 2 * xo + px;
            o[0] = act_t(y[i][0] + bs, a.act, a.p0);
            if (xo + 1 < W) o[2] = act_t(y[i][1] + bs, a.act, a.p0);
        }
    }
}

// ---- variant with the input transform in registers (mr_wino_desc.variant = 1; same idea as conv3x3_wino_rb_kernel) -----------------
// The B operand of the MFMA for (position p, channel quad c4) is V[p][4 c4 + (lane >> 4)][tile lane & 15]: the lane that needs it
// reads the 3x3 patch of its channel at its tile from the raw region (9 LDS reads per channel quad), transforms it (6 subtractions)
// and holds the 9 values as MFMA operands.  No V buffer - the pipeline buffers of 64 output channels take 62 KB instead of 85, so TWO
// workgroups share a CU and one's chunk fill hides behind the other's sweep - and one barrier per chunk instead of two.  Same
// products in the same order per accumulator as the kernel above: bit-identical outputs.
template <int MBW>
__global__ __launch_bounds__(512) void convt4x4_wino_rb_kernel(const WinoTKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (MBW == 1 && (int)blockIdx.y == a.tail_grp) { convt_rb_tail(a, lds); return; }    // 16-channel tail group: 16-row workgroups
    constexpr int U_FLOATS = NPOS * 2 * (2 * MBW) * 64;
    constexpr int U_PAD = (U_FLOATS + 255) & ~255;
    constexpr int BUF = WCK * RAW_PLANE + U_PAD;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    const int grp = blockIdx.y;
    const int ph = (int)blockIdx.z & 3, b = (int)blockIdx.z >> 2;
    const int py = ph >> 1, px = ph & 1;
    const int pt = 1 - py, pl = 1 - px;
    const int y0 = ty_wg * 8, x0 = tx_wg * 32;
    const int H = a.H, W = a.W, HW = H * W;

    int voff4[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = lane + 64 * i;
        const int row = r / 10, g4 = r - row * 10;
        const int gy = y0 - pt + row, gx = x0 - 4 + 4 * g4;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff4[i] = r < RAW_ROWS * 10 ? (inb ? (gy * W + gx) * 4 : -1) : -2;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
    const float* wgrp = a.w + (long long)ph * a.wphase_stride + (long long)grp * a.wgroup_stride;

    f32x4 acc[NPOS][MBW];
#pragma unroll
    for (int p = 0; p < NPOS; ++p)
#pragma unroll
        for (int m = 0; m < MBW; ++m) acc[p][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;
    auto issue = [&](int q, int pb) {
        const unsigned buf_addr = lds_base + pb * BUF * 4;
        const unsigned u_addr = buf_addr + WCK * RAW_PLANE * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS;
        constexpr int N1K = U_FLOATS / 256;
        for (int kb = wave; kb < N1K; kb += 8) dma_global_x4(u_addr + kb * 1024, wsrc + kb * 256 + lane * 4);
        for (int fr = N1K * 4 + wave; fr < U_FLOATS / 64; fr += 8) dma_global_x1(u_addr + fr * 256, wsrc + fr * 64 + lane);
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
    const int patch0 = (lane >> 4) * RAW_PLANE + (2 * tb) * RAW_PITCH + 2 * (lane & 15) + 4 - pl;   // channel lane >> 4, tile (tb, lane & 15)
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF;
        const float* ub = raw + WCK * RAW_PLANE + (chalf * MBW) * 64 + lane;
        dma_wait_all();
        __syncthreads();                                      // raw + U of chunk q visible; everyone is done with the other buffer
        if (q + 1 < a.nchunks) issue(q + 1, pb ^ 1);
        float v[2][NPOS];                                     // B operands of this lane: V[p] of channels 4 c4 + (lane >> 4)
#pragma unroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const float* rp = raw + patch0 + c4 * 4 * RAW_PLANE;
            float d[3][3], t[3][3];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) d[r][c] = rp[r * RAW_PITCH + c];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                t[0][c] = d[0][c] - d[1][c];
                t[1][c] = d[1][c];
                t[2][c] = d[2][c] - d[1][c];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                v[c4][r * 3 + 0] = t[r][0] - t[r][1];
                v[c4][r * 3 + 1] = t[r][1];
                v[c4][r * 3 + 2] = t[r][2] - t[r][1];
            }
        }
#pragma unroll
        for (int p = 0; p < NPOS; ++p)
#pragma unroll
            for (int c4 = 0; c4 < 2; ++c4)
#pragma unroll
                for (int m = 0; m < MBW; ++m) {
                    const float av = ub[((p * 2 + c4) * (2 * MBW) + m) * 64];
                    acc[p][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[c4][p], acc[p][m], 0, 0, 0);
                }
    }
    // ---- output transform Y = A^T M A, epilogue (as above) -------------------------------------------------------------------------
    const int xo = x0 + 2 * (lane & 15);
    const int yb = y0 + 2 * tb;
    if (xo >= W) return;
    const int OW = 2 * W;
#pragma unroll
    for (int m = 0; m < MBW; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cout = grp * (32 * MBW) + (chalf * MBW + m) * 16 + (lane >> 4) * 4 + r;
            if (cout >= a.Cout) continue;
            float s0[3], s1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                s0[c] = acc[0 + c][m][r] + acc[3 + c][m][r];
                s1[c] = acc[3 + c][m][r] + acc[6 + c][m][r];
            }
            float y[2][2];
            y[0][0] = s0[0] + s0[1];
            y[0][1] = s0[1] + s0[2];
            y[1][0] = s1[0] + s1[1];
            y[1][1] = s1[1] + s1[2];
            const float bs = a.bias ? a.bias[cout] : 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yo = yb + i;
                if (yo >= H) continue;
                float* o = a.dst + ((long long)(b * a.Cout + cout) * (2 * H) + (2 * yo + py)) * OW + 2 * xo + px;
                o[0] = act_t(y[i][0] + bs, a.act, a.p0);
                if (xo + 1 < W) o[2] = act_t(y[i][1] + bs, a.act, a.p0);
            }
        }
}

bool valid_mbw_t(int m) { return m == 1 || m == 2 || m == 4; }
int pad8(int c) { return (c + 7) & ~7; }

struct WinoTDerived {
    WinoTKArgs k;
    dim3 grid;
    size_t lds_bytes;
    int mbw;
    bool regb;
};

int derive_t(const mr_wino_desc* d, WinoTDerived* out) {
    if (!d || d->num_src < 1 || d->num_src > MR_MAX_SOURCES || d->batch < 1 || d->height < 1 || d->width < 4 || !d->dst ||
        !d->packed_weights || d->out_channels < 1 || d->residual)
        return MR_ERR_BAD_ARGUMENT;
    if (d->width % 4) return MR_ERR_UNSUPPORTED;
    if (!valid_mbw_t(d->cout_blocks_per_wave)) return MR_ERR_BAD_ARGUMENT;
    if (d->activation != MR_ACT_NONE && d->activation != MR_ACT_RELU && d->activation != MR_ACT_LEAKY_RELU) return MR_ERR_UNSUPPORTED;
    if (d->activation == MR_ACT_LEAKY_RELU && !(d->act_p0 >= 0.f && d->act_p0 <= 1.f)) return MR_ERR_UNSUPPORTED;   // the epilogue is max(x, x * slope)
    if (d->src_row_pitch || d->src_plane_floats || d->dst_split_columns) return MR_ERR_UNSUPPORTED;      // strided views: mr_conv1d_cooktoom_f32 only
    WinoTKArgs& k = out->k;
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
    k.nsrc = d->num_src;
    k.H = d->height; k.W = d->width;
    k.dst = d->dst; k.bias = d->bias;
    k.act = d->activation; k.p0 = d->act_p0;
    k.Cout = d->out_channels;
    k.tiles_x = (d->width + 31) / 32;
    k.nchunks = nchunks;
    k.w = d->packed_weights;
    const int mbw = d->cout_blocks_per_wave;
    const int ufl = NPOS * 2 * (2 * mbw) * 64;
    const int groups = (d->out_channels + 32 * mbw - 1) / (32 * mbw);
    k.ngroups = groups;
    k.wgroup_stride = (long long)nchunks * ufl;
    k.wphase_stride = (long long)groups * nchunks * ufl;
    if (d->batch * 4 >= 65536 || groups >= 65536) return MR_ERR_UNSUPPORTED;
    out->grid = dim3((unsigned)(k.tiles_x * ((d->height + 7) / 8)), (unsigned)groups, (unsigned)(d->batch * 4));
    if (d->variant < 0 || d->variant > 2) return MR_ERR_BAD_ARGUMENT;
    out->regb = d->variant >= 1;
    out->lds_bytes = (size_t)(2 * (WCK * RAW_PLANE + ((ufl + 255) & ~255)) + (out->regb ? 0 : V_FLOATS)) * 4;
    out->mbw = mbw;
    k.tail_grp = -1;
    k.tiles_y16 = (d->height + 15) / 16;
    if (d->variant == 2) {                                   // 32 a + (1..16) channels: the tail by 16-row workgroups
        const int rem = d->out_channels % 32;
        if (mbw != 1 || rem < 1 || rem > 16) return MR_ERR_BAD_ARGUMENT;
        k.tail_grp = d->out_channels / 32;
        k.wphase_stride = (long long)k.tail_grp * nchunks * ufl + (long long)nchunks * U_FLOATS_TT;
        if ((size_t)2 * BUF_TT * 4 > out->lds_bytes) out->lds_bytes = (size_t)2 * BUF_TT * 4;
    }
    return 0;
}

template <int MBW, bool REGB>
int launch_t(const WinoTDerived& dv, hipStream_t stream) {
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    const void* fn = REGB ? reinterpret_cast<const void*>(&convt4x4_wino_rb_kernel<MBW>) : reinterpret_cast<const void*>(&convt4x4_wino_kernel<MBW>);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    if (REGB) hipLaunchKernelGGL(convt4x4_wino_rb_kernel<MBW>, dv.grid, dim3(512), dv.lds_bytes, stream, dv.k);
    else hipLaunchKernelGGL(convt4x4_wino_kernel<MBW>, dv.grid, dim3(512), dv.lds_bytes, stream, dv.k);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" size_t mr_wino_t_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t mbw) {
    if (!src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_mbw_t(mbw) || out_channels < 1) return 0;
    int nchunks = 0;
    for (int s = 0; s < num_src; ++s) nchunks += pad8(src_channels[s]) / WCK;
    const int groups = (out_channels + 32 * mbw - 1) / (32 * mbw);
    return (size_t)4 * groups * nchunks * (NPOS * 2 * (2 * mbw) * 64);
}

// weight: the nn.ConvTranspose2d tensor (sum(src_channels), out_channels, 4, 4).  Parity (py, px) reads its kernel rows ky = {3, 1}
// (py = 0) or {2, 0} (py = 1) in tap order t = 0, 1 - likewise columns (engine.transposed_phase_weights; model/layers.py:389-397).
// U = G g G^T (G = [1 0; 1 1; 0 1]: sums of two weights, exact in double, rounded once); stream order
// [parity 2 py + px][cout group][chunk][position 3 a + b][channel quad][cout block][64 lanes].
extern "C" int mr_wino_t_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                          int32_t mbw, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_mbw_t(mbw) || out_channels < 1)
        return MR_ERR_BAD_ARGUMENT;
    static const double G[3][2] = {{1.0, 0.0}, {1.0, 1.0}, {0.0, 1.0}};
    static const int taps[2][2] = {{3, 1}, {2, 0}};
    const int groups = (out_channels + 32 * mbw - 1) / (32 * mbw);
    size_t o = 0;
    for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
        for (int g = 0; g < groups; ++g) {
            int cin_off = 0;
            for (int s = 0; s < num_src; ++s) {
                const int cpad = pad8(src_channels[s]);
                for (int c0 = 0; c0 < cpad; c0 += WCK)
                    for (int p = 0; p < NPOS; ++p)
                        for (int c4 = 0; c4 < 2; ++c4)
                            for (int mb = 0; mb < 2 * mbw; ++mb)
                                for (int lane = 0; lane < 64; ++lane) {
                                    const int cout = g * 32 * mbw + mb * 16 + (lane & 15);
                                    const int cl = c0 + c4 * 4 + (lane >> 4);
                                    double u = 0.0;
                                    if (cout < out_channels && cl < src_channels[s]) {
                                        const float* gw = weight + ((size_t)(cin_off + cl) * out_channels + cout) * 16;
                                        const int pa = p / 3, pb = p % 3;
                                        for (int i = 0; i < 2; ++i)
                                            for (int j = 0; j < 2; ++j) u += G[pa][i] * (double)gw[taps[py][i] * 4 + taps[px][j]] * G[pb][j];
                                    }
                                    dst[o++] = (float)u;
                                }
                cin_off += src_channels[s];
            }
        }
    }
    return 0;
}

// Variant 2 (out_channels = 32 a + r, 0 < r <= 16): per parity the a full groups exactly as mr_wino_t_pack_weights_f32(mbw = 1) lays them
// out, then the tail group as [chunk][position][channel quad][64 lanes], lane l = (cout 32 a + (l & 15), channel l >> 4).
extern "C" size_t mr_wino_t_packed_weight_floats_tail(int32_t out_channels, const int32_t* src_channels, int32_t num_src) {
    if (!src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || out_channels < 1) return 0;
    const int rem = out_channels % 32;
    if (rem < 1 || rem > 16) return 0;
    int nchunks = 0;
    for (int s = 0; s < num_src; ++s) nchunks += pad8(src_channels[s]) / WCK;
    return (size_t)4 * ((size_t)(out_channels / 32) * nchunks * (NPOS * 2 * 2 * 64) + (size_t)nchunks * U_FLOATS_TT);
}

extern "C" int mr_wino_t_pack_weights_tail_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src, float* dst) {
    if (!weight || !dst || mr_wino_t_packed_weight_floats_tail(out_channels, src_channels, num_src) == 0) return MR_ERR_BAD_ARGUMENT;
    static const double G[3][2] = {{1.0, 0.0}, {1.0, 1.0}, {0.0, 1.0}};
    static const int taps[2][2] = {{3, 1}, {2, 0}};
    const int full = out_channels / 32;
    size_t o = 0;
    for (int ph = 0; ph < 4; ++ph) {
        const int py = ph >> 1, px = ph & 1;
        for (int g = 0; g <= full; ++g) {                    // g == full: the tail group (one block of 16 channels)
            const int nblk = g < full ? 2 : 1;
            int cin_off = 0;
            for (int s = 0; s < num_src; ++s) {
                const int cpad = pad8(src_channels[s]);
                for (int c0 = 0; c0 < cpad; c0 += WCK)
                    for (int p = 0; p < NPOS; ++p)
                        for (int c4 = 0; c4 < 2; ++c4)
                            for (int mb = 0; mb < nblk; ++mb)
                                for (int lane = 0; lane < 64; ++lane) {
                                    const int cout = g * 32 + mb * 16 + (lane & 15);
                                    const int cl = c0 + c4 * 4 + (lane >> 4);
                                    double u = 0.0;
                                    if (cout < out_channels && cl < src_channels[s]) {
                                        const float* gw = weight + ((size_t)(cin_off + cl) * out_channels + cout) * 16;
                                        const int pa = p / 3, pb = p % 3;
                                        for (int i = 0; i < 2; ++i)
                                            for (int j = 0; j < 2; ++j) u += G[pa][i] * (double)gw[taps[py][i] * 4 + taps[px][j]] * G[pb][j];
                                    }
                                    dst[o++] = (float)u;
                                }
                cin_off += src_channels[s];
            }
        }
    }
    return 0;
}

extern "C" int64_t mr_convt4x4s2_winograd_lds_bytes(const mr_wino_desc* desc) {
    WinoTDerived dv;
    const int rc = derive_t(desc, &dv);
    return rc != 0 ? rc : (int64_t)dv.lds_bytes;
}

extern "C" int mr_convt4x4s2_winograd_f32(const mr_wino_desc* desc, void* stream) {
    WinoTDerived dv;
    const int rc = derive_t(desc, &dv);
    if (rc != 0) return rc;
    if (dv.regb) {
        switch (dv.mbw) {
            case 4: return launch_t<4, true>(dv, (hipStream_t)stream);
            case 2: return launch_t<2, true>(dv, (hipStream_t)stream);
            default: return launch_t<1, true>(dv, (hipStream_t)stream);
        }
    }
    switch (dv.mbw) {
        case 4: return launch_t<4, false>(dv, (hipStream_t)stream);
        case 2: return launch_t<2, false>(dv, (hipStream_t)stream);
        default: return launch_t<1, false>(dv, (hipStream_t)stream);
    }
}
