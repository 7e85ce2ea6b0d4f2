// 3x3 stride-1 convolutions as Winograd F(4x4, 3x3) on the fp32 matrix cores of gfx950 (MI355X).
//
// Next step after conv_wino.hip (F(2x2, 3x3), 16 multiplies per 2x2 outputs = 4 per output): a 4x4 output tile from a 6x6 input patch with 36
// multiplies per (cin, cout) = 2.25 per output (direct: 9) - for the layers of the MaskModule (reference model/monorec/monorec_model.py:296-343,
// layers.ConvReLU model/layers.py:317-335) that fill the chip with 16 x 64-pixel workgroups: the full-resolution stages at batch 1, everything at
// batch 8.
//     Y = A^T [ sum_cin (G g G^T) o (B^T d B) ] A         with the F(4, 3) matrices of cooktoom_1d.h (points 0, +-1, +-2, infinity)
// The transforms have coefficients up to 8 (B^T: 4, 5; A^T: 8) and G has sixths / 24ths, so the effect on the path's outputs was measured BEFORE
// this kernel was written (oracle/numerics_study_winograd.py: every 3x3 stride-1 layer of the mask and depth nets in emulated fp32 F(4x4, 3x3)
// arithmetic moves `result` by 2.4e-7, per layer 5e-6 from fp64; bar 1e-4).
//
// Skeleton = the in-register-transform kernels (conv3x3_wino_rb_kernel, conv1d_ct_kernel): workgroup = 8 waves = 4 tile rows x 2 blocks of 16
// output channels: 16 x 64 output pixels x 32 channels; K in chunks of 8 input channels; the haloed region (18 rows x 72 columns per channel, plane
// pitch 1344 = 0 mod 64 banks - see PLANE -, hardware zero fill = padding) and the chunk's U fragments (36 positions; host-packed G g G^T, formed in double and
// rounded once) arrive by LDS-DMA in one of two pipeline buffers (2 x 78 KB: one workgroup per CU - the 36 accumulator sets need 144 of a wave's
// 256 registers, so two waves per SIMD is what fits anyway); ONE barrier per chunk.  A lane reads the 6x6 patch of its (tile, channel) as 18
// 16-byte LDS reads, transforms rows then columns with the generated F(4,3) chain (one channel quad at a time: 36 live B operands), reads the
// quad's 36 A operands up front and issues 36 MFMAs (a variant with ONE 16-byte read per patch row and the outer columns from the
// neighbouring lanes by DPP moved 2.4x fewer LDS bytes and was no faster - tools/sessions/r04_s24.sh: the reads cost their latency, not
// their bandwidth); it ends up with all 36 positions of its (output channel, tile), so A^T M A, bias, residual, activation run in registers and
// the 4x4 tile leaves as four 16-byte stores.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <type_traits>

#include "../../include/monorec_hip.h"
#include "cooktoom_1d.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int WCK = 8;                                   // input channels per chunk (one per wave in the DMA phase)
constexpr int NP = 36;                                   // positions p = 6 i + j (i: vertical, j: horizontal transform index)
constexpr int RH = 16, RW = 64;                          // output pixels per workgroup
constexpr int ROWS = RH + 2, PITCH = RW + 8;             // raw region: rows oy0 - 1 .. oy0 + 16, columns ox0 - 4 .. ox0 + 67
constexpr int G4 = PITCH / 4, NG = ROWS * G4;            // 16-byte groups per channel plane (324)
constexpr int NI = (NG + 63) / 64;                       // DMA instructions per plane (6)
// Plane pitch: a patch read is a ds_read_b128 at (lane >> 4) * PLANE + 4 (lane & 15) + const: served in four groups of 16 lanes ({0-3, 12-15, 20-27}, ...
// MI355X_MICROARCH.md, LDS) over 64 banks, i.e. conflict free when the 16 lanes of a group - two channels - hit 16 distinct 16-byte slots of a 256-byte
// row: PLANE = 0 mod 64 floats.  (Rounds 3-4 ran the dword rule "16 mod 32" here - 1296 - which puts lanes 12-15 of one channel on the slots of lanes
// 20-23 of the next: every group took two cycles, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 33 % in profiles/r04_c3_pmc_summary.json; tools/lds_banks.py.)
constexpr int PLANE = (ROWS * PITCH + 63) / 64 * 64;     // 1296 -> 1344 floats
constexpr int U_FLOATS = NP * 2 * 2 * 64;                // U fragments of one chunk: [channel quad][cout block][j][64 lanes][i]
constexpr int BUF = WCK * PLANE + U_FLOATS;              // one pipeline buffer
static_assert(PLANE % 64 == 0 && PLANE >= ROWS * PITCH && U_FLOATS % 256 == 0 && 2 * BUF * 4 <= 160 * 1024, "layout");

struct W44KArgs {
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
    long long wgroup_stride;            // packed floats per group of 32 output channels
};

// Ablations (diagnostic library only: python -m monorec_amd.build --timeline, -DMR_W44_ABLATE; MR_W44_DBG picks an instantiation): compile-time,
// because a run-time flag in a kernel at 254 registers changes what is measured (tried: 1352 -> 2896 us with the flag present and zero).
// bits: 1 no input transform, 2 no MFMAs, 4 no patch reads, 8 no A reads, 16 no DMA, 32 every DMA instruction of a chunk in one burst behind the barrier
#define W44_DBG(bit) (DBG & (bit))

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

// none / ReLU / LeakyReLU as ONE branch-free form, max(x, lo) with lo = x (none), 0 (ReLU: -inf -> 0 and no -0.0, like torch.relu),
// x * p0 (LeakyReLU, 0 <= p0 <= 1 - the host side rejects other slopes); lo's selector is wave-uniform: as a
// switch the compiler emitted scalar branches around every stored element of the epilogue (round 4: 200-450 branches per workgroup)
__device__ __forceinline__ float act44(float v, int act, float p0) {
    const unsigned keep = act == MR_ACT_RELU ? 0u : ~0u;          // (an AND, not a select: a uniform select made hipcc clone the store loops)
    const float lo = __uint_as_float(__float_as_uint(v * (act == MR_ACT_LEAKY_RELU ? p0 : 1.f)) & keep);
    return fmaxf(v, lo);
}

// Measured and not kept (tools/sessions/r04_s23.sh - s25.sh, c3 mask.enc0.0, 1353 us): the ablations add up almost exactly - MFMA 398 + patch
// reads 371 + input transform 279 + DMA wait 231 + A reads 70 us - i.e. the eight waves of a workgroup hit the LDS, the VALU and the matrix
// pipe one after the other.  A "ping-pong" loop (the two waves of a SIMD one phase apart - one loads + transforms while the other issues its 36
// MFMAs from registers - a barrier per phase) was correct and 19 % SLOWER (1587 us): a lone wave per SIMD stretches the load + transform phase
// (dependent VALU chains, LDS round trips) beyond what the overlap buys, as round 2 found for F(2x2,3x3).  One 16-byte + one 4-byte LDS read per
// patch row with the outer columns by DPP (2.4x fewer LDS bytes) and 24 % fewer transform instructions changed nothing either.
template <int DBG>
__global__ __launch_bounds__(512) void conv3x3_wino44_kernel(const W44KArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty_wg = (int)blockIdx.x / a.tiles_x, tx_wg = (int)blockIdx.x - ty_wg * a.tiles_x;
    const int grp = blockIdx.y, b = blockIdx.z;
    const int oy0 = ty_wg * RH, ox0 = tx_wg * RW;
    const int H = a.H, W = a.W, HW = H * W;

    int voff4[NI];                                            // lane l owns the 16-byte groups r = l + 64 i of a plane: row r / 18, group r % 18
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = lane + 64 * i;
        const int row = r / G4, g4 = r - row * G4;
        const int gy = oy0 - 1 + row, gx = ox0 - 4 + 4 * g4;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        voff4[i] = r < NG ? (inb ? (gy * W + gx) * 4 : -1) : -2;
    }
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
    const float* wgrp = a.w + (long long)grp * a.wgroup_stride;

    f32x4 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int cs = 0, cc0 = 0;                                      // chunk cursor: source, first channel
    auto issue = [&](int q, int pb) {
        if (W44_DBG(16)) return;
        const unsigned buf_addr = lds_base + pb * BUF * 4;
        const unsigned u_addr = buf_addr + WCK * PLANE * 4;
        const float* wsrc = wgrp + (long long)q * U_FLOATS;
        for (int kb = wave; kb < U_FLOATS / 256; kb += 8) dma_global_x4(u_addr + kb * 1024, wsrc + kb * 256 + lane * 4);
        const i32x4 srd = make_srd(a.src[cs], a.src_bytes[cs]);
        const bool cok = cc0 + wave < a.src_c[cs];            // padded channels read as zero
        const int so = ((b * a.src_c[cs] + cc0 + (cok ? wave : 0)) * HW) * 4;
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if (voff4[i] != -2) dma_buffer_x4(buf_addr + wave * (PLANE * 4) + i * 1024, cok ? voff4[i] : -1, srd, so);
        cc0 += WCK;
        if (cc0 >= a.src_cpad[cs]) { cc0 = 0; ++cs; }
    };

    // The same fetch in pieces (one DMA instruction each), for the chunks after the first: an LDS-DMA instruction holds its wave for a few
    // hundred cycles, and issued in one burst behind the chunk barrier the ~10 of a wave keep BOTH waves of every SIMD off the matrix pipe
    // (ablation: 231 of 1353 us).  Spread over the MFMAs of the chunk's first channel quad, a wave's stall is covered by its SIMD partner's
    // MFMAs; the data still has the whole second quad to land.  Pieces 0..5: the wave's input plane, 6..10: its share of the U block.
    i32x4 nsrd = {0, 0, 0, 0};
    int nso = 0;
    bool ncok = false;
    const float* nwsrc = nullptr;
    unsigned nbuf = 0;
    auto next_begin = [&](int q, int pb) {                    // q = the chunk to fetch, pb = its buffer; advances the chunk cursor
        nbuf = lds_base + pb * BUF * 4;
        nwsrc = wgrp + (long long)q * U_FLOATS;
        nsrd = make_srd(a.src[cs], a.src_bytes[cs]);
        ncok = cc0 + wave < a.src_c[cs];
        nso = ((b * a.src_c[cs] + cc0 + (ncok ? wave : 0)) * HW) * 4;
        cc0 += WCK;
        if (cc0 >= a.src_cpad[cs]) { cc0 = 0; ++cs; }
    };
    auto next_piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if (W44_DBG(16)) return;
        if constexpr (k < NI) {
            if (voff4[k] != -2) dma_buffer_x4(nbuf + wave * (PLANE * 4) + k * 1024, ncok ? voff4[k] : -1, nsrd, nso);
        } else if constexpr (k < NI + 5) {
            const int kb = wave + 8 * (k - NI);
            if (kb < U_FLOATS / 256) dma_global_x4(nbuf + WCK * PLANE * 4 + kb * 1024, nwsrc + kb * 256 + lane * 4);
        }
    };
    constexpr bool SPREAD = !W44_DBG(32);                     // diagnostic instantiation 32: the burst behind the barrier, as before

    issue(0, 0);
    const int tb = wave & 3, cb = wave >> 2;                  // tile row of the workgroup, block of 16 output channels of the group
    const int t = lane & 15;
    const bool active = (grp * 2 + cb) * 16 < a.Cout;         // a 16-channel tail group: the waves of its empty block only move data
    // the 6x6 patch of tile (tb, t), channel lane >> 4 of a quad: raw rows 4 tb .. 4 tb + 5, raw columns 4 t + 3 .. 4 t + 8, read as the three
    // aligned 16-byte groups from column 4 t on
    const int patch0 = (lane >> 4) * PLANE + (4 * tb) * PITCH + 4 * t;
    for (int q = 0; q < a.nchunks; ++q) {
        const int pb = q & 1;
        const float* raw = lds + pb * BUF;
        const float* ub = raw + WCK * PLANE + cb * (6 * 64 * 6) + lane * 6;   // U of a chunk: [quad][block][j][lane][i], position p = 6 i + j
        dma_wait_all();
        __syncthreads();                                      // raw + U of chunk q visible; everyone is done with the other buffer
        const bool more = q + 1 < a.nchunks;
        if (more) {
            if (SPREAD) next_begin(q + 1, pb ^ 1);
            else issue(q + 1, pb ^ 1);
        }
        if (!active) {                                        // the waves of an empty block only move data
            if (SPREAD && more) {
                next_piece(std::integral_constant<int, 0>{}); next_piece(std::integral_constant<int, 1>{}); next_piece(std::integral_constant<int, 2>{});
                next_piece(std::integral_constant<int, 3>{}); next_piece(std::integral_constant<int, 4>{}); next_piece(std::integral_constant<int, 5>{});
                next_piece(std::integral_constant<int, 6>{}); next_piece(std::integral_constant<int, 7>{}); next_piece(std::integral_constant<int, 8>{});
                next_piece(std::integral_constant<int, 9>{}); next_piece(std::integral_constant<int, 10>{});
            }
            continue;
        }
#pragma nounroll
        for (int c4 = 0; c4 < 2; ++c4) {
            const float* rp = raw + patch0 + c4 * 4 * PLANE;
            // the 36 A operands of the quad first, as 18 8-byte reads (lane pitch 24 bytes: conflict free), and nothing may sink them:
            // left to itself hipcc reads every operand right in front of its MFMA (230 registers in use) and the wave waits out an LDS
            // round trip a dozen times per quad
            const float* uq = ub + c4 * (2 * 6 * 64 * 6);
            float2 av[6][3];
#pragma unroll
            for (int j = 0; j < 6; ++j)
#pragma unroll
                for (int k = 0; k < 3; ++k) av[j][k] = W44_DBG(8) ? make_float2((float)(lane + j), (float)k) : *(const float2*)(uq + j * (64 * 6) + 2 * k);
            __builtin_amdgcn_sched_barrier(0);
            float v[NP];                                      // h = d B per patch row first, then B^T h per column, in place
#pragma unroll
            for (int r = 0; r < 6; ++r) {
                float x[12];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const f32x4 g = W44_DBG(4) ? (f32x4){(float)lane, (float)r, (float)j, 1.f} : *(const f32x4*)(rp + r * PITCH + 4 * j);
                    x[4 * j] = g.x; x[4 * j + 1] = g.y; x[4 * j + 2] = g.z; x[4 * j + 3] = g.w;
                }
                float d[6], h[6];
#pragma unroll
                for (int c = 0; c < 6; ++c) d[c] = x[3 + c];
                if (W44_DBG(1)) {
#pragma unroll
                    for (int c = 0; c < 6; ++c) h[c] = d[c];
                } else {
                    ct_input_4_3(d, h);
                }
#pragma unroll
                for (int c = 0; c < 6; ++c) v[r * 6 + c] = h[c];
            }
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                float d[6], h[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) d[r] = v[r * 6 + c];
                if (W44_DBG(1)) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) h[r] = d[r];
                } else {
                    ct_input_4_3(d, h);
                }
#pragma unroll
                for (int r = 0; r < 6; ++r) v[r * 6 + c] = h[r];
            }
            const bool fetch = SPREAD && more && c4 == 0;     // (uniform)
#define W44_COLUMN(j)                                                                                                              \
            _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                        \
                const float av_ = (i & 1) ? av[j][i >> 1].y : av[j][i >> 1].x;                                                     \
                if (W44_DBG(2)) acc[i * 6 + j][0] += av_ * v[i * 6 + j];                                                            \
                else acc[i * 6 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, v[i * 6 + j], acc[i * 6 + j], 0, 0, 0);            \
            }                                                                                                                      \
            if (fetch) {                                                                                                           \
                __builtin_amdgcn_sched_barrier(0);                                                                                 \
                next_piece(std::integral_constant<int, 2 * (j)>{});                                                                \
                next_piece(std::integral_constant<int, 2 * (j) + 1>{});                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                                                 \
            }
            W44_COLUMN(0) W44_COLUMN(1) W44_COLUMN(2) W44_COLUMN(3) W44_COLUMN(4) W44_COLUMN(5)
#undef W44_COLUMN
        }
    }
    // ---- output transform Y = A^T M A per (cout, tile) in registers, epilogue --------------------------------------------------------
    const int ox = ox0 + 4 * t;
    const int oyb = oy0 + 4 * tb;
    if (!active || ox >= W || oyb >= H) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cout = (grp * 2 + cb) * 16 + (lane >> 4) * 4 + r;
        if (cout >= a.Cout) continue;
        float s[4][6];                                        // A^T M: columns of M through the output transform
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float mm[6], y[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) mm[i] = acc[i * 6 + j][r];
            ct_output_4_3(mm, y);
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k][j] = y[k];
        }
        const float bs = a.bias ? a.bias[cout] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int oy = oyb + k;
            if (oy >= H) continue;
            float y[4];
            ct_output_4_3(s[k], y);
            const long long idx = ((long long)(b * a.Cout + cout) * H + oy) * W + ox;      // W % 4 == 0 and ox % 4 == 0: all four columns exist
            f32x4 o = (f32x4){y[0] + bs, y[1] + bs, y[2] + bs, y[3] + bs};
            if (a.res) o += *(const f32x4*)(a.res + idx);
            o.x = act44(o.x, a.act, a.p0); o.y = act44(o.y, a.act, a.p0); o.z = act44(o.z, a.act, a.p0); o.w = act44(o.w, a.act, a.p0);
            *(f32x4*)(a.dst + idx) = o;
        }
    }
}

int pad8(int c) { return (c + 7) & ~7; }

struct W44Derived {
    W44KArgs k;
    dim3 grid;
    size_t lds_bytes;
};

int derive44(const mr_wino_desc* d, W44Derived* out) {
    if (!d || d->num_src < 1 || d->num_src > MR_MAX_SOURCES || d->batch < 1 || d->height < 1 || d->width < 4 || !d->dst ||
        !d->packed_weights || d->out_channels < 1)
        return MR_ERR_BAD_ARGUMENT;
    if (d->width % 4) return MR_ERR_UNSUPPORTED;              // 16-byte groups entirely inside or outside the image
    if (d->activation != MR_ACT_NONE && d->activation != MR_ACT_RELU && d->activation != MR_ACT_LEAKY_RELU) return MR_ERR_UNSUPPORTED;
    if (d->activation == MR_ACT_LEAKY_RELU && !(d->act_p0 >= 0.f && d->act_p0 <= 1.f)) return MR_ERR_UNSUPPORTED;   // the epilogue is max(x, x * slope)
    if (d->src_row_pitch || d->src_plane_floats || d->dst_split_columns) return MR_ERR_UNSUPPORTED;      // strided views: mr_conv1d_cooktoom_f32 only
    W44KArgs& k = out->k;
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
    k.tiles_x = (d->width + RW - 1) / RW;
    k.nchunks = nchunks;
    k.w = d->packed_weights;
    k.wgroup_stride = (long long)nchunks * U_FLOATS;
    const int groups = (d->out_channels + 31) / 32;
    if (d->batch >= 65536 || groups >= 65536) return MR_ERR_UNSUPPORTED;
    out->grid = dim3((unsigned)(k.tiles_x * ((d->height + RH - 1) / RH)), (unsigned)groups, (unsigned)d->batch);
    out->lds_bytes = (size_t)(2 * BUF) * 4;
    return 0;
}

}  // namespace

extern "C" size_t mr_wino44_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src) {
    if (!src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || out_channels < 1) return 0;
    int nchunks = 0;
    for (int s = 0; s < num_src; ++s) nchunks += pad8(src_channels[s]) / WCK;
    return (size_t)((out_channels + 31) / 32) * nchunks * U_FLOATS;
}

// weight: (out_channels, sum(src_channels), 3, 3) fp32, nn.Conv2d layout.  U = G g G^T (6 x 6; G of F(4,3): cooktoom_1d.h) in double, rounded
// once to fp32; stream order [group of 32 output channels][chunk (source-major, 8 channels)][channel quad][block of 16 channels of the
// group][j][64 lanes][i] with position p = 6 i + j, lane l = (cout l & 15 of the block, channel l >> 4 of the quad): a lane reads the six
// operands of a transform column as three 8-byte words.
extern "C" int mr_wino44_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || out_channels < 1) return MR_ERR_BAD_ARGUMENT;
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    const int groups = (out_channels + 31) / 32;
    size_t o = 0;
    for (int g = 0; g < groups; ++g) {
        int cin_off = 0;
        for (int s = 0; s < num_src; ++s) {
            const int cpad = pad8(src_channels[s]);
            for (int c0 = 0; c0 < cpad; c0 += WCK)
                for (int c4 = 0; c4 < 2; ++c4)
                    for (int mb = 0; mb < 2; ++mb)
                        for (int pj_ = 0; pj_ < 6; ++pj_)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int pi_ = 0; pi_ < 6; ++pi_) {
                                const int p = pi_ * 6 + pj_;
                                const int cout = g * 32 + mb * 16 + (lane & 15);
                                const int cl = c0 + c4 * 4 + (lane >> 4);
                                double u = 0.0;
                                if (cout < out_channels && cl < src_channels[s]) {
                                    const float* gw = weight + ((size_t)cout * cin_total + (cin_off + cl)) * 9;
                                    const int pi = p / 6, pj = p % 6;
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

extern "C" int64_t mr_conv3x3_winograd44_lds_bytes(const mr_wino_desc* desc) {
    W44Derived dv;
    const int rc = derive44(desc, &dv);
    return rc != 0 ? rc : (int64_t)dv.lds_bytes;
}

extern "C" int mr_conv3x3_winograd44_f32(const mr_wino_desc* desc, void* stream) {
    W44Derived dv;
    const int rc = derive44(desc, &dv);
    if (rc != 0) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    auto launch = [&](auto kernel, std::atomic<unsigned long long>& attr_set) -> int {      // dynamic-LDS ceiling once per device and instantiation
        if (!(attr_set.load(std::memory_order_acquire) & bit)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_set.fetch_or(bit, std::memory_order_release);
        }
        hipLaunchKernelGGL(kernel, dv.grid, dim3(512), dv.lds_bytes, (hipStream_t)stream, dv.k);
        return 0;
    };
    static std::atomic<unsigned long long> set0{0};
#ifdef MR_W44_ABLATE
    static const int dbg = [] { const char* e = getenv("MR_W44_DBG"); return e ? atoi(e) : 0; }();
    static std::atomic<unsigned long long> setd[8];
    int rc2 = -1000;
    switch (dbg) {
        case 0: break;
        case 1: rc2 = launch(conv3x3_wino44_kernel<1>, setd[0]); break;
        case 2: rc2 = launch(conv3x3_wino44_kernel<2>, setd[1]); break;
        case 4: rc2 = launch(conv3x3_wino44_kernel<4>, setd[2]); break;
        case 8: rc2 = launch(conv3x3_wino44_kernel<8>, setd[3]); break;
        case 16: rc2 = launch(conv3x3_wino44_kernel<16>, setd[4]); break;
        case 5: rc2 = launch(conv3x3_wino44_kernel<5>, setd[5]); break;
        case 13: rc2 = launch(conv3x3_wino44_kernel<13>, setd[6]); break;
        case 29: rc2 = launch(conv3x3_wino44_kernel<29>, setd[7]); break;
        case 32: rc2 = launch(conv3x3_wino44_kernel<32>, setd[0]); break;
        default: return MR_ERR_BAD_ARGUMENT;
    }
    if (rc2 != -1000) return rc2 != 0 ? rc2 : (int)hipGetLastError();
#endif
    const int rc3 = launch(conv3x3_wino44_kernel<0>, set0);
    if (rc3 != 0) return rc3;
    return (int)hipGetLastError();
}
