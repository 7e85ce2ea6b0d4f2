// 3x3 stride-1 convolutions as Winograd F(4x4, 3x3) on the fp32 matrix cores of gfx950 - ONE WAVE PER SIMD, 512 registers (round 6).
//
// Same products, transformed weights (mr_wino44_pack_weights_f32) and results as conv_wino44.hip (reference model/monorec/monorec_model.py:296-343,
// layers.ConvReLU model/layers.py:317-335: the MaskModule's 3x3 layers; the 3x3 stride-1 layers of the ResNet trunk and depth.dec.4.2).  What changes
// is who does what, and it follows from two measurements of this round:
//   * v_mfma_f32_16x16x4_f32 executes on the SIMD's fp32 vector ALUs (tools/probes/mfma_rates.hip: 32.0 cycles per MFMA back to back, 43-55 with one
//     v_add + two SALU per MFMA in between): the input transform of F(4x4,3x3) - 144 VALU instructions per (tile row, channel quad) - does not hide
//     under the MFMAs of another wave, it ADDS to them.  In conv_wino44.hip the two waves that own the two 16-channel blocks of a tile row each run that
//     transform: 167 VALU instructions per 36 MFMAs.
//   * that kernel's phases (patch reads 371, transform 279, MFMA 398, DMA wait 231 of 1353 us at c3 mask.enc0.0, tools/sessions/r04_s23.sh) add up because
//     at 254 registers a wave cannot hold the next quad's patch while it multiplies the current one, and its SIMD partner is in the same phase.
// Here a workgroup is 4 waves (one per SIMD: 512 registers each) on the same 16 x 64 pixels x 32 output channels; a wave owns a tile ROW and BOTH
// 16-channel blocks: 72 accumulator sets (288 registers, most of them AGPRs), ONE input transform per 72 MFMAs, and an explicit software pipeline over
// channel quads: while the 72 MFMAs of quad n issue (B operands from registers, A operands one transform column ahead from LDS), the 18 patch reads of
// quad n + 1 are in flight, then its transform runs - the LDS latency sits behind the MFMAs, the DMA latency behind a ring of FOUR one-quad stages
// (4 x 39 KB) with partial vmcnt waits.
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

constexpr int NP = 36;                                   // positions p = 6 i + j (i: vertical, j: horizontal transform index)
constexpr int RH = 16, RW = 64;                          // output pixels per workgroup
constexpr int ROWS = RH + 2, PITCH = RW + 8;             // raw region: rows oy0 - 1 .. oy0 + 16, columns ox0 - 4 .. ox0 + 67
constexpr int G4 = PITCH / 4, NG = ROWS * G4;            // 16-byte groups per channel plane (324)
constexpr int NI = (NG + 63) / 64;                       // DMA instructions per plane (6)
constexpr int PLANE = (ROWS * PITCH + 63) / 64 * 64;     // 1344 floats: 0 mod 64 banks for the 16-byte patch reads (conv_wino44.hip)
constexpr int UQ_FLOATS = NP * 2 * 64;                   // U fragments of one channel quad: [cout block][j][64 lanes][i]
constexpr int STAGE = 4 * PLANE + UQ_FLOATS;             // one stage: 4 channel planes + the quad's U
constexpr int NSTAGE = 4;
constexpr int UPIECES = UQ_FLOATS / 256;                 // 1 KiB DMA pieces of a quad's U (18)
static_assert(UQ_FLOATS % 256 == 0 && NSTAGE * STAGE * 4 <= 160 * 1024, "layout");

struct W44WArgs {
    const float* src[MR_MAX_SOURCES];
    int src_bytes[MR_MAX_SOURCES];
    int src_c[MR_MAX_SOURCES];
    int src_cpad[MR_MAX_SOURCES];       // padded to a multiple of 8 (the packed stream of conv_wino44.hip)
    int nsrc;
    int H, W;
    float* dst;
    const float* bias;
    const float* res;
    int act;
    float p0;
    int Cout, tiles_x, nquads;
    const float* w;
    long long wgroup_stride;            // packed floats per group of 32 output channels
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
// s_waitcnt vmcnt(N) takes an immediate: wait until at most `n` (wave-uniform) of this wave's VMEM instructions are outstanding (rounding down is safe)
#define MR_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
__device__ __forceinline__ void dma_wait_upto(int n) {
    switch (n < 32 ? n : (n < 48 ? 32 : 48)) {
        MR_VMCNT_CASE(0) MR_VMCNT_CASE(1) MR_VMCNT_CASE(2) MR_VMCNT_CASE(3) MR_VMCNT_CASE(4) MR_VMCNT_CASE(5) MR_VMCNT_CASE(6) MR_VMCNT_CASE(7)
        MR_VMCNT_CASE(8) MR_VMCNT_CASE(9) MR_VMCNT_CASE(10) MR_VMCNT_CASE(11) MR_VMCNT_CASE(12) MR_VMCNT_CASE(13) MR_VMCNT_CASE(14) MR_VMCNT_CASE(15)
        MR_VMCNT_CASE(16) MR_VMCNT_CASE(17) MR_VMCNT_CASE(18) MR_VMCNT_CASE(19) MR_VMCNT_CASE(20) MR_VMCNT_CASE(21) MR_VMCNT_CASE(22) MR_VMCNT_CASE(23)
        MR_VMCNT_CASE(24) MR_VMCNT_CASE(25) MR_VMCNT_CASE(26) MR_VMCNT_CASE(27) MR_VMCNT_CASE(28) MR_VMCNT_CASE(29) MR_VMCNT_CASE(30) MR_VMCNT_CASE(31)
        MR_VMCNT_CASE(32) MR_VMCNT_CASE(48)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ i32x4 make_srd(const void* base, int bytes) {
    const unsigned long long p = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)p);
    r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(p >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000;
    return r;
}

__device__ __forceinline__ float act44w(float v, int act, float p0) {          // none / ReLU / LeakyReLU (0 <= p0 <= 1) as max(x, lo), see conv_wino44.hip
    const unsigned keep = act == MR_ACT_RELU ? 0u : ~0u;
    const float lo = __uint_as_float(__float_as_uint(v * (act == MR_ACT_LEAKY_RELU ? p0 : 1.f)) & keep);
    return fmaxf(v, lo);
}

// 72 accumulator sets are 288 registers and the AGPR file has 256: once a kernel needs AGPRs hipcc gives EVERY builtin MFMA an AGPR accumulator and
// shuttles the sets that do not fit through v_accvgpr_read / _write around each use (104 moves in the first transform column of every quad, measured on
// the ISA).  Eight sets (second block, positions 28 .. 35) therefore live in arch VGPRs for good, their MFMAs written in the VGPR form through inline asm.
// Hazards: the same accumulator is next touched 72 MFMAs later, and read by VALU instructions only in the epilogue.
__device__ __forceinline__ constexpr bool in_vgprs_impl(int nblk, int c, int i, int j) { return nblk == 2 && c == 1 && (i == 5 || (i == 4 && j >= 4)); }
__device__ __forceinline__ void mfma_vgpr_form(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

// NBLK: 16-channel output blocks this workgroup's group really has (2; 1 for a 16-channel tail group)
template <int NBLK>
__device__ __forceinline__ void w44w_body(const W44WArgs& a, float* lds) {
    auto in_vgprs = [](int c, int i, int j) { return in_vgprs_impl(NBLK, c, i, j); };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);              // = tile row of the workgroup
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

    // ---- DMA of one quad = one stage: wave w brings channel plane w (6 instructions) and every fourth 1 KiB piece of the quad's U ----------------
    int cs = 0, cc0 = 0;                                      // issue cursor: source, first channel of the next quad
    int ib = 0, nissued = 0;
    unsigned long long fifo = 0;                              // DMA instructions of the quads in flight, 8 bits each, oldest in the low byte
    int depth = 0, pend = 0;
    i32x4 nsrd = {0, 0, 0, 0};
    int nso = 0;
    bool ncok = false;
    const float* nwsrc = nullptr;
    unsigned nbuf = 0;
    constexpr int NPIECE = NI + (UPIECES + 3) / 4;            // pieces of a wave per quad (11)
    auto next_begin = [&]() {                                 // set up the pieces of quad `nissued` (stage ib) and account for them
        nbuf = lds_base + ib * (STAGE * 4);
        nwsrc = wgrp + (long long)nissued * UQ_FLOATS;
        nsrd = make_srd(a.src[cs], a.src_bytes[cs]);
        ncok = cc0 + wave < a.src_c[cs];                      // padded channels read as zero
        nso = ((b * a.src_c[cs] + cc0 + (ncok ? wave : 0)) * HW) * 4;
        cc0 += 4;
        if (cc0 >= a.src_cpad[cs]) { cc0 = 0; ++cs; }
        const int cnt = NI + (UPIECES - wave + 3) / 4;        // (lane 0 of every input piece is inside the region: all six issue)
        fifo |= (unsigned long long)cnt << (8 * depth);
        ++depth;
        pend += cnt;
        ib = ib + 1 == NSTAGE ? 0 : ib + 1;
        ++nissued;
    };
    auto next_piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < NI) {
            if (voff4[k] != -2) dma_buffer_x4(nbuf + wave * (PLANE * 4) + k * 1024, ncok ? voff4[k] : -1, nsrd, nso);
        } else if constexpr (k < NPIECE) {
            const int kb = wave + 4 * (k - NI);
            if (kb < UPIECES) dma_global_x4(nbuf + 4 * PLANE * 4 + kb * 1024, nwsrc + kb * 256 + lane * 4);
        }
    };
    auto issue_all = [&]() {
        next_begin();
        next_piece(std::integral_constant<int, 0>{}); next_piece(std::integral_constant<int, 1>{}); next_piece(std::integral_constant<int, 2>{});
        next_piece(std::integral_constant<int, 3>{}); next_piece(std::integral_constant<int, 4>{}); next_piece(std::integral_constant<int, 5>{});
        next_piece(std::integral_constant<int, 6>{}); next_piece(std::integral_constant<int, 7>{}); next_piece(std::integral_constant<int, 8>{});
        next_piece(std::integral_constant<int, 9>{}); next_piece(std::integral_constant<int, 10>{});
    };
    auto wait_oldest = [&]() {                                // this wave's share of the oldest quad in flight has landed
        const int cnt = (int)(fifo & 255);
        fifo >>= 8;
        --depth;
        pend -= cnt;
        dma_wait_upto(pend);
    };

    const int nq = a.nquads;
    for (int j = 0; j < NSTAGE && j < nq; ++j) issue_all();

    f32x4 acc[NBLK][NP];
#pragma unroll
    for (int c = 0; c < NBLK; ++c)
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[c][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int t = lane & 15;
    // the 6x6 patch of tile (wave, t), channel lane >> 4 of the quad: raw rows 4 wave .. 4 wave + 5, raw columns 4 t + 3 .. 4 t + 8, read as the three
    // aligned 16-byte groups from column 4 t on
    const int patch0 = (lane >> 4) * PLANE + (4 * wave) * PITCH + 4 * t;

    f32x4 px[18];                                             // raw patch of the NEXT quad (6 rows x 3 groups)
    float2 a0[NBLK][3], a1[NBLK][3];                          // A operands of ONE transform column (block, pairs of i), double buffered: holding all 72 of a
                                                              // quad left hipcc 8 accumulator sets short of registers (v_accvgpr round trips in the MFMA stream)
    float v[NP];                                              // transformed patch of the CURRENT quad
    auto read_patch = [&](const float* stage) {
        const float* rp = stage + patch0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j) px[r * 3 + j] = *(const f32x4*)(rp + r * PITCH + 4 * j);
    };
    auto read_col = [&](float2 (&ac)[NBLK][3], const float* stage, int j) {
        const float* ub = stage + 4 * PLANE + lane * 6 + j * (64 * 6);      // U of a quad: [block][j][lane][i], position p = 6 i + j
#pragma unroll
        for (int c = 0; c < NBLK; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) ac[c][k] = *(const float2*)(ub + c * (6 * 64 * 6) + 2 * k);
    };
    auto transform = [&]() {                                  // v = B^T d B of the patch in px: rows first, then columns, in place
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float x[12];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                x[4 * j] = px[r * 3 + j].x; x[4 * j + 1] = px[r * 3 + j].y; x[4 * j + 2] = px[r * 3 + j].z; x[4 * j + 3] = px[r * 3 + j].w;
            }
            float d[6], h[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] = x[3 + c];
            ct_input_4_3(d, h);
#pragma unroll
            for (int c = 0; c < 6; ++c) v[r * 6 + c] = h[c];
        }
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            float d[6], h[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) d[r] = v[r * 6 + c];
            ct_input_4_3(d, h);
#pragma unroll
            for (int r = 0; r < 6; ++r) v[r * 6 + c] = h[r];
        }
    };

    // ---- quad 0 into registers ---------------------------------------------------------------------------------------------------------------------
    wait_oldest();
    __syncthreads();
    int sb = 0;                                               // stage of the CURRENT quad (its A operands are read column by column during its MFMAs)
    read_patch(lds);
    read_col(a0, lds, 0);
    transform();
    for (int n = 0; n < nq; ++n) {
        const bool more = n + 1 < nq;                         // (uniform)
        const bool fetch = n >= 1 && nissued < nq;            // quad n - 1 + NSTAGE goes into the stage quad n - 1 left
        const float* cur = lds + sb * STAGE;
        const int sn = sb + 1 == NSTAGE ? 0 : sb + 1;
        if (more) {
            wait_oldest();                                    // quad n + 1 has landed (this wave's share) ...
            __syncthreads();                                  // ... everyone's has, and everyone is done with quad n - 1: its stage is free
            if (fetch) next_begin();
            read_patch(lds + sn * STAGE);                     // in flight during the MFMAs below
        }
        __builtin_amdgcn_sched_barrier(0);
#define W44W_COLUMN(j, AC, AN)                                                                                                          \
        if ((j) < 5) read_col(AN, cur, (j) + 1);              /* the next column's A operands land behind this column's MFMAs */       \
        _Pragma("unroll") for (int c = 0; c < NBLK; ++c)                                                                               \
            _Pragma("unroll") for (int i = 0; i < 6; ++i) {                                                                            \
                const float av_ = (i & 1) ? AC[c][i >> 1].y : AC[c][i >> 1].x;                                                          \
                if (in_vgprs(c, i, (j))) mfma_vgpr_form(acc[c][i * 6 + (j)], av_, v[i * 6 + (j)]);                                        \
                else acc[c][i * 6 + (j)] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_, v[i * 6 + (j)], acc[c][i * 6 + (j)], 0, 0, 0);     \
            }                                                                                                                          \
        if (more && fetch) {                                                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            next_piece(std::integral_constant<int, 2 * (j)>{});                                                                        \
            next_piece(std::integral_constant<int, 2 * (j) + 1>{});                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
        }
        W44W_COLUMN(0, a0, a1) W44W_COLUMN(1, a1, a0) W44W_COLUMN(2, a0, a1) W44W_COLUMN(3, a1, a0) W44W_COLUMN(4, a0, a1) W44W_COLUMN(5, a1, a0)
#undef W44W_COLUMN
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            read_col(a0, lds + sn * STAGE, 0);                // the next quad's first A column lands while its patch is transformed
            transform();
            sb = sn;
        }
    }

    // ---- output transform Y = A^T M A per (cout, tile) in registers, epilogue --------------------------------------------------------------------------
    const int ox = ox0 + 4 * t;
    const int oyb = oy0 + 4 * wave;
    if (ox >= W || oyb >= H) return;
#pragma unroll
    for (int c = 0; c < NBLK; ++c) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cout = (grp * 2 + c) * 16 + (lane >> 4) * 4 + r;
            if (cout >= a.Cout) continue;
            float s[4][6];                                    // A^T M: columns of M through the output transform
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float mm[6], y[4];
#pragma unroll
                for (int i = 0; i < 6; ++i) mm[i] = acc[c][i * 6 + j][r];
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
                o.x = act44w(o.x, a.act, a.p0); o.y = act44w(o.y, a.act, a.p0); o.z = act44w(o.z, a.act, a.p0); o.w = act44w(o.w, a.act, a.p0);
                *(f32x4*)(a.dst + idx) = o;
            }
        }
    }
}

__global__ __launch_bounds__(256) void conv3x3_wino44w_kernel(const W44WArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (((int)blockIdx.y * 2 + 1) * 16 < a.Cout) w44w_body<2>(a, lds);      // (uniform) both 16-channel blocks of the group exist
    else w44w_body<1>(a, lds);
}

int pad8(int c) { return (c + 7) & ~7; }

struct W44WDerived {
    W44WArgs k;
    dim3 grid;
    size_t lds_bytes;
};

int derive44w(const mr_wino_desc* d, W44WDerived* out) {
    if (!d || d->num_src < 1 || d->num_src > MR_MAX_SOURCES || d->batch < 1 || d->height < 1 || d->width < 4 || !d->dst ||
        !d->packed_weights || d->out_channels < 1)
        return MR_ERR_BAD_ARGUMENT;
    if (d->width % 4) return MR_ERR_UNSUPPORTED;              // 16-byte groups entirely inside or outside the image
    if (d->activation != MR_ACT_NONE && d->activation != MR_ACT_RELU && d->activation != MR_ACT_LEAKY_RELU) return MR_ERR_UNSUPPORTED;
    if (d->activation == MR_ACT_LEAKY_RELU && !(d->act_p0 >= 0.f && d->act_p0 <= 1.f)) return MR_ERR_UNSUPPORTED;   // the epilogue is max(x, x * slope)
    if (d->src_row_pitch || d->src_plane_floats || d->dst_split_columns) return MR_ERR_UNSUPPORTED;      // strided views: mr_conv1d_cooktoom_f32 only
    W44WArgs& k = out->k;
    memset(&k, 0, sizeof(k));
    int nquads = 0;
    for (int s = 0; s < d->num_src; ++s) {
        if (!d->src[s] || d->src_channels[s] < 1) return MR_ERR_BAD_ARGUMENT;
        const long long bytes = (long long)d->batch * d->src_channels[s] * d->height * d->width * 4;
        if (bytes >= (1ll << 31)) return MR_ERR_UNSUPPORTED;
        k.src[s] = d->src[s];
        k.src_bytes[s] = (int)bytes;
        k.src_c[s] = d->src_channels[s];
        k.src_cpad[s] = pad8(d->src_channels[s]);
        nquads += k.src_cpad[s] / 4;
    }
    if ((long long)d->batch * d->out_channels * d->height * d->width * 4 >= (1ll << 33)) return MR_ERR_UNSUPPORTED;
    k.nsrc = d->num_src;
    k.H = d->height; k.W = d->width;
    k.dst = d->dst; k.bias = d->bias; k.res = d->residual;
    k.act = d->activation; k.p0 = d->act_p0;
    k.Cout = d->out_channels;
    k.tiles_x = (d->width + RW - 1) / RW;
    k.nquads = nquads;
    k.w = d->packed_weights;
    k.wgroup_stride = (long long)nquads * UQ_FLOATS;
    const int groups = (d->out_channels + 31) / 32;
    if (d->batch >= 65536 || groups >= 65536) return MR_ERR_UNSUPPORTED;
    out->grid = dim3((unsigned)(k.tiles_x * ((d->height + RH - 1) / RH)), (unsigned)groups, (unsigned)d->batch);
    out->lds_bytes = (size_t)(NSTAGE * STAGE) * 4;
    return 0;
}

}  // namespace

extern "C" int64_t mr_conv3x3_winograd44w_lds_bytes(const mr_wino_desc* desc) {
    W44WDerived dv;
    const int rc = derive44w(desc, &dv);
    return rc != 0 ? rc : (int64_t)dv.lds_bytes;
}

extern "C" int mr_conv3x3_winograd44w_f32(const mr_wino_desc* desc, void* stream) {
    W44WDerived dv;
    const int rc = derive44w(desc, &dv);
    if (rc != 0) return rc;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    static std::atomic<unsigned long long> attr_set{0};      // dynamic-LDS ceiling once per device
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino44w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(conv3x3_wino44w_kernel, dv.grid, dim3(256), dv.lds_bytes, (hipStream_t)stream, dv.k);
    return (int)hipGetLastError();
}
