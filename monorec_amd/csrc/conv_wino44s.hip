// Winograd F(4x4, 3x3) with the 36 positions of a tile split over TWO waves, so that two workgroups share a CU (round 5).
//
// conv_wino44.hip (3x3 stride-1 layers of the MaskModule, reference model/monorec/monorec_model.py:296-343, layers.ConvReLU model/layers.py:317-335)
// holds all 36 accumulator sets of a (tile, 16 output channels) in one wave: 144 of 256 registers, 2 x 76.5 KB of LDS, ONE workgroup per CU - and
// its phases add up instead of overlapping (r04: MFMA 398 + patch reads 371 + input transform 279 + DMA wait 231 + A reads 70 us of 1353 on c3
// mask.enc0.0: the eight waves of the one workgroup go through LDS, VALU and matrix pipe one after the other; ping-pong inside a workgroup lost
// twice).  What breaks the serialisation is a second, independent workgroup on the CU (another barrier domain), and that needs half the registers
// and half the LDS per workgroup:
//   * a wave owns the positions p = 6 i + j of ONE half of the vertical index, i in {3 hf, 3 hf + 1, 3 hf + 2}: 18 accumulator sets = 72 registers;
//     the transforms are linear, so each half forms its own partial A^T M A and the two partial 4x4 output tiles are added through LDS once, after the
//     K loop;
//   * the vertical input transform of a half needs only 5 of the 6 patch rows (B^T of F(4,3): rows 0-2 read d0..d4, rows 3-5 read d1..d5) and is
//     accumulated row by row as the patch rows arrive from LDS (no 36-register patch); the horizontal transform is the generated ct_input_4_3 chain;
//   * K in chunks of ONE channel quad (4 channels): 2 x (12 KB raw region + 24 KB of U) = 72 KB of LDS, ~125 registers: two workgroups per CU,
//     four waves per SIMD from two barrier domains.
// Workgroup = 8 waves = 2 tile rows x 2 blocks of 16 output channels x 2 position halves: 8 x 64 output pixels x 32 channels.  Same arithmetic per
// product as conv_wino44.hip (same U = G g G^T, same B^T d B); the output transform adds its two partial tiles in a fixed order (half 0 + half 1).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>

#include "../../include/monorec_hip.h"
#include "cooktoom_1d.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int SCK = 4;                                   // input channels per chunk: one MFMA quad
constexpr int RH = 8, RW = 64;                           // output pixels per workgroup
constexpr int ROWS = RH + 2, PITCH = RW + 8;             // raw region: rows oy0 - 1 .. oy0 + 8, columns ox0 - 4 .. ox0 + 67
constexpr int G4 = PITCH / 4, NG = ROWS * G4;            // 16-byte groups per channel plane (180)
constexpr int NI = (NG + 63) / 64;                       // DMA instructions per plane (3)
constexpr int PLANE = (ROWS * PITCH + 63) / 64 * 64;     // 720 -> 768 floats = 0 mod 64 banks: the 16-byte patch reads are conflict free (see conv_wino44.hip)
constexpr int U_FLOATS = 2 * 2 * 3 * 2 * 64 * 4;         // U of one chunk: [block][half][ii][jh][64 lanes][4: j = 4 jh + 0..3; j = 6, 7 are zero pads] = 24 KB
constexpr int BUF = SCK * PLANE + U_FLOATS;              // one pipeline buffer (36 KB)
static_assert(PLANE % 64 == 0 && PLANE >= ROWS * PITCH && U_FLOATS % 256 == 0 && 2 * BUF * 4 <= 80 * 1024, "layout: two workgroups per CU");

struct W44SArgs {
    const float* src[MR_MAX_SOURCES];
    int src_bytes[MR_MAX_SOURCES];
    int src_c[MR_MAX_SOURCES];
    int src_cpad[MR_MAX_SOURCES];       // padded to a multiple of SCK
    int nsrc;
    int H, W;
    float* dst;
    const float* bias;
    const float* res;
    int act;
    float p0;
    int Cout, tiles_x, nchunks;
    const float* w;
    long long wgroup_stride;            // packed floats per group of 32 output channels
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

// none / ReLU / LeakyReLU (0 <= p0 <= 1) as max(x, lo), lo = x / 0 / x * p0 (see conv_wino44.hip)
__device__ __forceinline__ float act44s(float v, int act, float p0) {
    const unsigned keep = act == MR_ACT_RELU ? 0u : ~0u;
    const float lo = __uint_as_float(__float_as_uint(v * (act == MR_ACT_LEAKY_RELU ? p0 : 1.f)) & keep);
    return fmaxf(v, lo);
}

// HF = position half of the wave (template: the two halves run different transform code; one kernel, a wave-uniform branch at the top level)
template <int HF>
__device__ __forceinline__ void w44s_body(const W44SArgs& a, float* lds, const int lane, const int wave, const int tb, const int cb) {
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    const int grp = blockIdx.y, b = blockIdx.z;
    const int oy0 = ty_wg * RH, ox0 = tx_wg * RW;
    const int H = a.H, W = a.W, HW = H * W;

    // DMA roles: wave w streams channel plane (w & 3) of the chunk; its 16-byte groups r = lane + 64 i, i = (w >> 2), (w >> 2) + 2, ... (3 instructions
    // per plane: the lower four waves issue two of them, the upper four one) and three of the chunk's 24 U pieces
    const int plane_w = wave & 3;
    int voff4[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = (wave >> 2) + 2 * k;
        const int r = lane + 64 * i;
        const int row = r / G4, g4 = r - row * G4;
        const int gy = oy0 - 1 + row, gx = ox0 - 4 + 4 * g4;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff4[k] = (i < NI && r < NG) ? (inb ? (gy * W + gx) * 4 : -1) : -2;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
    const float* wgrp = a.w + (long long)grp * a.wgroup_stride;

    f32x4 acc[18];                                            // position (i = 3 HF + ii, j): acc[ii * 6 + j]
#pragma unroll
    for (int p = 0; p < 18; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;                                      // chunk cursor: source, first channel
    auto issue = [&](int q, int pb) {
        const unsigned buf_addr = lds_base + pb * BUF * 4;
        const unsigned u_addr = buf_addr + SCK * PLANE * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS;
#pragma unroll
        for (int k = 0; k < 3; ++k) dma_global_x4(u_addr + (wave + 8 * k) * 1024, wsrc + (wave + 8 * k) * 256 + lane * 4);
        const i32x4 srd = make_srd(a.src[cs], a.src_bytes[cs]);
        const bool cok = cc0 + plane_w < a.src_c[cs];         // padded channels read as zero
        const int so = ((b * a.src_c[cs] + cc0 + (cok ? plane_w : 0)) * HW) * 4;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (voff4[k] != -2) dma_buffer_x4(buf_addr + plane_w * (PLANE * 4) + ((wave >> 2) + 2 * k) * 1024, cok ? voff4[k] : -1, srd, so);
        cc0 += SCK;
        if (cc0 >= a.src_cpad[cs]) { cc0 = 0; ++cs; }
    };

    issue(0, 0);
    const int t = lane & 15;
    const bool active = (grp * 2 + cb) * 16 < a.Cout;         // a 16-channel tail group: the waves of its empty block only move data
    // the patch of tile (tb, t), channel lane >> 4: raw rows 4 tb + HF .. 4 tb + HF + 4 (the 5 rows this half's vertical transform reads), raw
    // columns 4 t + 3 .. 4 t + 8, read as the three aligned 16-byte groups from column 4 t on
    const int patch0 = (lane >> 4) * PLANE + (4 * tb + HF) * PITCH + 4 * t;
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF;
        const float* ub = raw + SCK * PLANE + ((cb * 2 + HF) * 6 * 64 + lane) * 4;       // U of a chunk: [block][half][ii][jh][lane][4] (j = 4 jh + e)
        dma_wait_all();
        __syncthreads();                                      // raw + U of chunk q visible; everyone is done with the other buffer
        if (q + 1 < a.nchunks) issue(q + 1, pb ^ 1);
        if (!active) continue;
        // vertical transform of this half, accumulated row by row (B^T of F(4,3), cooktoom_1d.h: ct_input_4_3):
        //   half 0: v0 = 4 d0 - 5 d2 + d4;  e = d4 - 4 d2, o = d3 - 4 d1: v1 = e + o, v2 = e - o          (patch rows d0 .. d4)
        //   half 1: e = d4 - d2, o = 2 d3 - 2 d1: v3 = e + o, v4 = e - o;  v5 = 4 d1 - 5 d3 + d5          (patch rows d1 .. d5 = local rows 0 .. 4)
        float ta[6], te[6], to[6];                            // per raw column c: the lone row (v0 / v5), e, o
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            float x[12];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const f32x4 g = *(const f32x4*)(raw + patch0 + r * PITCH + 4 * j);
                x[4 * j] = g.x; x[4 * j + 1] = g.y; x[4 * j + 2] = g.z; x[4 * j + 3] = g.w;
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const float d = x[3 + c];
                if (HF == 0) {
                    if (r == 0) ta[c] = 4.0f * d;                                             // d0
                    if (r == 1) to[c] = -4.0f * d;                                            // d1
                    if (r == 2) { ta[c] = fmaf(-5.0f, d, ta[c]); te[c] = -4.0f * d; }         // d2
                    if (r == 3) to[c] = to[c] + d;                                            // d3
                    if (r == 4) { ta[c] = ta[c] + d; te[c] = te[c] + d; }                     // d4
                } else {
                    if (r == 0) { to[c] = -2.0f * d; ta[c] = 4.0f * d; }                      // d1
                    if (r == 1) te[c] = -d;                                                   // d2
                    if (r == 2) { to[c] = fmaf(2.0f, d, to[c]); ta[c] = fmaf(-5.0f, d, ta[c]); }   // d3
                    if (r == 3) te[c] = te[c] + d;                                            // d4
                    if (r == 4) ta[c] = ta[c] + d;                                            // d5
                }
            }
        }
        // horizontal transform of the three rows of this half and their MFMAs, one row at a time (6 live B operands)
#pragma unroll
        for (int ii = 0; ii < 3; ++ii) {
            float d[6], v[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                if (HF == 0) d[c] = ii == 0 ? ta[c] : (ii == 1 ? te[c] + to[c] : te[c] - to[c]);
                else d[c] = ii == 0 ? te[c] + to[c] : (ii == 1 ? te[c] - to[c] : ta[c]);
            }
            // the six A operands of this row: two 16-byte reads (lane pitch 16 bytes: conflict free), issued ahead of the transform chain
            const f32x4 a0 = *(const f32x4*)(ub + (ii * 2) * (64 * 4)), a1 = *(const f32x4*)(ub + (ii * 2 + 1) * (64 * 4));
            ct_input_4_3(d, v);
            const float av[6] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y};
#pragma unroll
            for (int j = 0; j < 6; ++j) acc[ii * 6 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], v[j], acc[ii * 6 + j], 0, 0, 0);
        }
    }
    // ---- output transform: this half's partial Y = A^T M A per (cout, tile) in registers ---------------------------------------------------
    // A^T of F(4,3) (ct_output_4_3): y0 = m0 + (m1 + m2) + (m3 + m4), y1 = (m1 - m2) + 2 (m3 - m4), y2 = (m1 + m2) + 4 (m3 + m4),
    // y3 = (m1 - m2) + 8 (m3 - m4) + m5.  Vertical part restricted to this half's rows, then the full horizontal transform of each of the 4 rows.
    __syncthreads();                                          // every wave is done with the pipeline buffers: they become the exchange area
    const int pair = tb * 2 + cb;                             // (tile row, cout block): the two halves of a pair meet here
    float* xch = lds + pair * (64 * 64);                      // [value 0..63 = r * 16 + k * 4 + col][64 lanes]
    // partial output tile of output channel r of this lane: yp[k * 4 + col], 16 values at a time (the accumulators stay live until consumed)
    auto partial = [&](int r, float (&yp)[16]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s[6], y[4];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float m0 = acc[j][r], m1 = acc[6 + j][r], m2 = acc[12 + j][r];
                if (HF == 0) s[j] = k == 0 ? (m0 + (m1 + m2)) : (k == 2 ? (m1 + m2) : (m1 - m2));                     // rows i = 0, 1, 2
                else s[j] = k == 0 ? (m0 + m1) : (k == 1 ? 2.0f * (m0 - m1) : (k == 2 ? 4.0f * (m0 + m1) : fmaf(8.0f, m0 - m1, m2)));   // rows i = 3, 4, 5
            }
            ct_output_4_3(s, y);
#pragma unroll
            for (int c = 0; c < 4; ++c) yp[k * 4 + c] = y[c];
        }
    };
    if (HF == 1 && active) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float yp[16];
            partial(r, yp);
#pragma unroll
            for (int e = 0; e < 16; ++e) xch[(r * 16 + e) * 64 + lane] = yp[e];
        }
    }
    __syncthreads();
    if (HF == 1) return;
    const int ox = ox0 + 4 * t;
    const int oyb = oy0 + 4 * tb;
    if (!active || ox >= W || oyb >= H) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cout = (grp * 2 + cb) * 16 + (lane >> 4) * 4 + r;
        if (cout >= a.Cout) continue;
        const float bs = a.bias ? a.bias[cout] : 0.f;
        float yp[16];
        partial(r, yp);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oy = oyb + k;
            if (oy >= H) continue;
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = (yp[k * 4 + c] + xch[(r * 16 + k * 4 + c) * 64 + lane]) + bs;
            const long long idx = ((long long)(b * a.Cout + cout) * H + oy) * W + ox;      // W % 4 == 0 and ox % 4 == 0: all four columns exist
            if (a.res) o += *(const f32x4*)(a.res + idx);
            o.x = act44s(o.x, a.act, a.p0); o.y = act44s(o.y, a.act, a.p0); o.z = act44s(o.z, a.act, a.p0); o.w = act44s(o.w, a.act, a.p0);
            *(f32x4*)(a.dst + idx) = o;
        }
    }
}

__global__ __launch_bounds__(512, 4) void conv3x3_wino44s_kernel(const W44SArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave = (tile row tb, block of 16 output channels cb, position half hf); waves w and w + 4 share a SIMD: the two halves of a (tb, cb) pair do
    const int tb = wave & 1, cb = (wave >> 1) & 1;
    if ((wave >> 2) == 0) w44s_body<0>(a, lds, lane, wave, tb, cb);
    else w44s_body<1>(a, lds, lane, wave, tb, cb);
}

int pad4(int c) { return (c + 3) & ~3; }

struct W44SDerived {
    W44SArgs k;
    dim3 grid;
    size_t lds_bytes;
};

int derive44s(const mr_wino_desc* d, W44SDerived* out) {
    if (!d || d->num_src < 1 || d->num_src > MR_MAX_SOURCES || d->batch < 1 || d->height < 1 || d->width < 4 || !d->dst ||
        !d->packed_weights || d->out_channels < 1)
        return MR_ERR_BAD_ARGUMENT;
    if (d->width % 4) return MR_ERR_UNSUPPORTED;              // 16-byte groups entirely inside or outside the image
    if (d->activation != MR_ACT_NONE && d->activation != MR_ACT_RELU && d->activation != MR_ACT_LEAKY_RELU) return MR_ERR_UNSUPPORTED;
    if (d->activation == MR_ACT_LEAKY_RELU && !(d->act_p0 >= 0.f && d->act_p0 <= 1.f)) return MR_ERR_UNSUPPORTED;   // the epilogue is max(x, x * slope)
    if (d->src_row_pitch || d->src_plane_floats || d->dst_split_columns) return MR_ERR_UNSUPPORTED;      // strided views: mr_conv1d_cooktoom_f32 only
    W44SArgs& k = out->k;
    memset(&k, 0, sizeof(k));
    int nchunks = 0;
    for (int s = 0; s < d->num_src; ++s) {
        if (!d->src[s] || d->src_channels[s] < 1) return MR_ERR_BAD_ARGUMENT;
        const long long bytes = (long long)d->batch * d->src_channels[s] * d->height * d->width * 4;
        if (bytes >= (1ll << 31)) return MR_ERR_UNSUPPORTED;
        k.src[s] = d->src[s];
        k.src_bytes[s] = (int)bytes;
        k.src_c[s] = d->src_channels[s];
        k.src_cpad[s] = pad4(d->src_channels[s]);
        nchunks += k.src_cpad[s] / SCK;
    }
    if ((long long)d->batch * d->out_channels * d->height * d->width * 4 >= (1ll << 33)) return MR_ERR_UNSUPPORTED;
    k.nsrc = d->num_src;
    k.H = d->height; k.W = d->width;
    k.dst = d->dst; k.bias = d->bias; k.res = d->residual;
    k.act = d->activation; k.p0 = d->act_p0;
    k.Cout = d->out_channels;
    k.tiles_x = (d->width + RW - 1) / RW;
    k.nchunks = nchunks;
    k.w = d->packed_weights;
    k.wgroup_stride = (long long)nchunks * U_FLOATS;
    const int groups = (d->out_channels + 31) / 32;
    if (d->batch >= 65536 || groups >= 65536) return MR_ERR_UNSUPPORTED;
    out->grid = dim3((unsigned)(k.tiles_x * ((d->height + RH - 1) / RH)), (unsigned)groups, (unsigned)d->batch);
    out->lds_bytes = (size_t)(2 * BUF) * 4;
    static_assert(2 * BUF >= 4 * 64 * 64, "the exchange area of the output transform fits the pipeline buffers");
    return 0;
}

}  // namespace

extern "C" size_t mr_wino44s_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src) {
    if (!src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || out_channels < 1) return 0;
    int nchunks = 0;
    for (int s = 0; s < num_src; ++s) nchunks += pad4(src_channels[s]) / SCK;
    return (size_t)((out_channels + 31) / 32) * nchunks * U_FLOATS;
}

// weight: (out_channels, sum(src_channels), 3, 3) fp32, nn.Conv2d layout.  U = G g G^T (6 x 6; G of F(4,3): cooktoom_1d.h) in double, rounded
// once to fp32 - the same values as mr_wino44_pack_weights_f32; stream order [group of 32 output channels][chunk (source-major, 4 channels)]
// [block of 16 channels of the group][position half][ii][jh][64 lanes][4] with element e = position p = 6 (3 half + ii) + (4 jh + e), zero pads
// for 4 jh + e >= 6; lane l = (cout l & 15 of the block, channel l >> 4 of the chunk): a lane reads the six operands of a transform row as two
// 16-byte words.
extern "C" int mr_wino44s_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || out_channels < 1) return MR_ERR_BAD_ARGUMENT;
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    const int groups = (out_channels + 31) / 32;
    size_t o = 0;
    for (int g = 0; g < groups; ++g) {
        int cin_off = 0;
        for (int s = 0; s < num_src; ++s) {
            const int cpad = pad4(src_channels[s]);
            for (int c0 = 0; c0 < cpad; c0 += SCK)
                for (int mb = 0; mb < 2; ++mb)
                    for (int hf = 0; hf < 2; ++hf)
                        for (int ii = 0; ii < 3; ++ii)
                            for (int jh = 0; jh < 2; ++jh)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int e = 0; e < 4; ++e) {
                                    const int pi = 3 * hf + ii, pj = 4 * jh + e;
                                    const int cout = g * 32 + mb * 16 + (lane & 15);
                                    const int cl = c0 + (lane >> 4);
                                    double u = 0.0;
                                    if (pj < 6 && cout < out_channels && cl < src_channels[s]) {
                                        const float* gw = weight + ((size_t)cout * cin_total + (cin_off + cl)) * 9;
                                        for (int i = 0; i < 3; ++i) {
                                            double row = 0.0;
                                            for (int j = 0; j < 3; ++j) row += (double)gw[i * 3 + j] * CT_G_4_3[pj][j];
                                            u += CT_G_4_3[pi][i] * row;
                                        }
                                    }
                                    dst[o++] = (float)u;
                                }
            cin_off += src_channels[s];
        }
    }
    return 0;
}

extern "C" int64_t mr_conv3x3_winograd44s_lds_bytes(const mr_wino_desc* desc) {
    W44SDerived dv;
    const int rc = derive44s(desc, &dv);
    return rc != 0 ? rc : (int64_t)dv.lds_bytes;
}

extern "C" int mr_conv3x3_winograd44s_f32(const mr_wino_desc* desc, void* stream) {
    W44SDerived dv;
    const int rc = derive44s(desc, &dv);
    if (rc != 0) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    static std::atomic<unsigned long long> attr_set{0};          // dynamic-LDS ceiling once per device
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_wino44s_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(conv3x3_wino44s_kernel, dv.grid, dim3(512), dv.lds_bytes, (hipStream_t)stream, dv.k);
    return (int)hipGetLastError();
}
