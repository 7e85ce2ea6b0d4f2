// Direct convolution on the bf16 matrix cores of gfx950 with CHANNEL-BLOCKED bf16 ACTIVATION STORAGE (BASELINE configs[4]: "bf16 MFMA path").
//
// Stands in for the nn.Conv2d / ConvTranspose2d / Upconv layers of MaskModule and DepthModule (reference model/monorec/monorec_model.py:
// 345-385,526-557; model/layers.py:289-356,380-400) when the model runs with hip_bf16=True.  Round 1-3 kept the activations fp32 in HBM and
// rounded them to bf16 while forming the MFMA operand: at 512x1024 the path then moves 13 GB per keyframe and sits at 7 % of the bf16 peak
// (VERDICT r3, row g).  Here every activation between two convolutions of those nets is stored as "B8":
//     tensor (N, ceil(C / 8), H, W, 8) bf16  -  one 16-byte group = 8 consecutive channels of one pixel, padded channels zero
// so that
//   * a pixel's 8 channels ARE the per-lane operand of v_mfma_f32_16x16x32_bf16 (lane = (pixel l & 15, channel block l >> 4)): the B fragment
//     of a k-step of 32 channels is ONE ds_read_b128 per lane - no conversion, no 16-bit LDS reads;
//   * the haloed input tile goes global -> LDS by LDS-DMA, 16 bytes per lane (buffer_load_dwordx4 ... lds), out-of-range offsets zero-filled
//     by the buffer descriptor (padding, tile halo, padded channel blocks need no branch and no VGPR);
//   * HBM traffic per activation halves.
// Sources may also be fp32 NCHW (what the path's inputs and outputs are: keyframe, cost volumes, image features - and the maps the HBM-bound
// head / classifier kernels read): those chunks are staged through registers (8 coalesced plane loads per pixel, round to nearest even,
// one ds_write_b128).  The destination is B8 or fp32 NCHW.
//
// Workgroup = WV (4 or 8) waves = WV * NB blocks of 16 output pixels (TH rows x 32 columns) x MB blocks of 16 output channels.  K is walked
// in chunks of 32 channels (4 blocks) of ONE source through two LDS buffers: chunk q + 1 streams in while chunk q is swept; per tap and chunk
// a wave reads MB A fragments (lane-linear 16 bytes: packed by mr_b8_pack_weights) and NB B fragments and issues MB * NB MFMAs.
// The four output parities of ConvTranspose2d(4,2) / Upconv run as four phases of one launch (per-phase filter size, padding, output offset).
// fp32 accumulation; bias, activation and the bf16 rounding of a B8 destination in the epilogue.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <type_traits>

#include "../../include/monorec_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define B8_MAX_PPT 2          // tile positions staged per thread (ceil(IH * IW / (64 WV)))

namespace {

struct B8Args {
    const void* src[MR_MAX_SOURCES];
    int src_bytes[MR_MAX_SOURCES];
    int src_c[MR_MAX_SOURCES];       // channels
    int src_cb[MR_MAX_SOURCES];      // blocks of 8 channels
    int src_layout[MR_MAX_SOURCES];
    int src_q0[MR_MAX_SOURCES];      // first K chunk of the source
    int nsrc, nchunks;
    int Hs, Ws;
    int SH, SW, KH, KW;              // KH / KW: maximum over the phases (sizes the input tile)
    int Ho, Wo;
    void* dst;
    int dst_layout, dst_H, dst_W, ostep_h, ostep_w;
    int dst_bytes;                   // B8 destination: whole tensor, < 2 GiB (stored through a buffer descriptor)
    int Cout, CB16;
    const float* bias;
    int act;
    float p0;
    int tiles_x, TH, IH, IW, PLANE;
    int ntiles, tiles_per_wg;        // a workgroup walks tiles_per_wg consecutive tiles
    int nphase, batch;
    const void* w[4];
    long long wgroup_bytes[4];       // packed bytes per cout group
    int KHp[4], KWp[4], PT[4], PL[4], ooff_h[4], ooff_w[4];
    int dbg;                         // diagnostic library only (MR_B8_DBG): 1 skip the sweep, 2 skip the input staging, 4 skip the weight DMA, 8 skip the stores
};

// ablation switches exist only in the diagnostic library (python -m monorec_amd.build --timeline, -DMR_B8_ABLATE): compile-time 0 in the product
#ifdef MR_B8_ABLATE
#define B8_DBG(bit) (a.dbg & (bit))
#else
#define B8_DBG(bit) 0
#endif

__device__ __forceinline__ void dma_buffer_x4(unsigned lds_byte_addr, int voff, i32x4 srd, int soff) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    soff = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte_addr), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_global_x4(unsigned lds_byte_addr, const void* g) {
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

__device__ __forceinline__ float act1(float v, int act, float p0) {
    switch (act) {
        case MR_ACT_RELU: return v > 0.f ? v : 0.f;
        case MR_ACT_LEAKY_RELU: return v > 0.f ? v : v * p0;
        default: return v;
    }
}

__device__ __forceinline__ unsigned pack2(float a, float b) {          // two floats -> two bf16 (round to nearest even), a in the low half
    const bf16x2 h = __builtin_convertvector((f32x2){a, b}, bf16x2);
    return __builtin_bit_cast(unsigned, h);
}

// WRES: the whole weight stream of the workgroup's cout group (all K chunks) is loaded into LDS ONCE and stays there while the workgroup
// walks `tiles_per_wg` consecutive tiles - the layers with few input channels (every full-resolution layer of the two nets) would
// otherwise re-load 27-36 KB of weights per chunk and tile and expose the first chunk's latency in every workgroup (first hardware
// run, r04_s2: mask.enc0.1 at 289 us = 1.4 TB/s).  The input pipeline runs across tile boundaries: chunk 0 of the next tile streams in
// while the last chunk of the current tile is swept.
// element s (0..2, wave-uniform) of a kernel-argument array without dynamic indexing (which would move the array to scratch)
template <typename T>
__device__ __forceinline__ T pick3(const T (&v)[MR_MAX_SOURCES], int s) { return s == 0 ? v[0] : (s == 1 ? v[1] : v[2]); }
template <typename T>
__device__ __forceinline__ T pick4(const T (&v)[4], int p) { return p == 0 ? v[0] : (p == 1 ? v[1] : (p == 2 ? v[2] : v[3])); }

// F32SRC: some source is dense fp32 (register-staged path compiled in; 64 VGPRs of staging registers)
template <int MB, int NB, int WV, bool WRES, bool F32SRC>
__global__ __launch_bounds__(WV * 64) void conv_b8_kernel(const B8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = blockIdx.y;
    int z = blockIdx.z;
    const int ph = a.nphase == 4 ? (z & 3) : 0;
    const int b = a.nphase == 4 ? z >> 2 : z;
    const int KH = pick4(a.KHp, ph), KW = pick4(a.KWp, ph), T = KH * KW;
    const int PT = pick4(a.PT, ph), PL = pick4(a.PL, ph), ooff_h = pick4(a.ooff_h, ph), ooff_w = pick4(a.ooff_w, ph);
    const int PLANE = a.PLANE;
    const int wchunk_bytes = T * MB * 1024;
    const int wres_bytes = WRES ? a.nchunks * a.KH * a.KW * MB * 1024 : 0;           // resident weights in front of the two stages
    const int stage_bytes = 64 * PLANE + (WRES ? 0 : 1024 * a.KH * a.KW * MB);       // [4 blocks][PLANE] x 16 B (+ [taps][MB][64 lanes] x 16 B)
    const int tile_begin = blockIdx.x * a.tiles_per_wg;
    const int tile_end = min(tile_begin + a.tiles_per_wg, a.ntiles);
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned char* wgrp = (const unsigned char*)pick4(a.w, ph) + (long long)grp * pick4(a.wgroup_bytes, ph);
    const int HsWs = a.Hs * a.Ws;

    if (WRES && !B8_DBG(4)) {
        const int pieces = a.nchunks * T * MB;                                         // 1 KiB each, contiguous in the packed stream
        for (int kb = wave; kb < pieces; kb += WV) dma_global_x4(lds_base + kb * 1024, wgrp + (long long)kb * 1024 + lane * 16);
    }

    // ---- load cursor: the next (tile, chunk) to stage; positions p = tid + 64 WV j of the haloed tile -> global pixel of that tile ---------
    int ltile = tile_begin, lq = 0;
    int gpix[B8_MAX_PPT];                // gy * Ws + gx, or -1 outside the image
    bool live[B8_MAX_PPT];
    int piy[B8_MAX_PPT], pix_[B8_MAX_PPT];      // (row, column) of the position inside the tile: the same for every tile (one division per kernel)
#pragma unroll
    for (int j = 0; j < B8_MAX_PPT; ++j) {
        const int p = tid + 64 * WV * j;
        live[j] = p < PLANE;
        piy[j] = p / a.IW;
        pix_[j] = p - piy[j] * a.IW;
    }
    int lty = tile_begin / a.tiles_x, ltx = tile_begin - lty * a.tiles_x;      // (row, column) of the load cursor's tile: one division per kernel
    auto place = [&](int ty, int tx) {
        const int iy_base = ty * a.TH * a.SH - PT, ix_base = tx * 32 * a.SW - PL;
#pragma unroll
        for (int j = 0; j < B8_MAX_PPT; ++j) {
            const int gy = iy_base + piy[j], gx = ix_base + pix_[j];
            gpix[j] = (live[j] && (unsigned)gy < (unsigned)a.Hs && (unsigned)gx < (unsigned)a.Ws) ? gy * a.Ws + gx : -1;
        }
    };
    place(lty, ltx);

    // weights (unless resident) + B8 inputs: LDS-DMA, nothing to wait for until the barrier.  fp32 NCHW inputs: loaded into `stg` here (8
    // channel planes per position), converted and written to LDS by stage_store() AFTER the sweep of the previous chunk - the loads fly
    // during the sweep.
    float stg[F32SRC ? B8_MAX_PPT : 1][4][8];
    auto source_of = [&](int q, int& s, int& blk0) {
        s = (a.nsrc > 2 && q >= a.src_q0[2]) ? 2 : ((a.nsrc > 1 && q >= a.src_q0[1]) ? 1 : 0);
        blk0 = (q - pick3(a.src_q0, s)) * 4;
    };
    auto issue = [&](int q, int pb) {
        const unsigned buf = lds_base + wres_bytes + pb * stage_bytes;
        if (!WRES && !B8_DBG(4)) {
            const unsigned wbuf = buf + 64 * PLANE;
            const unsigned char* wsrc = wgrp + (long long)q * wchunk_bytes;
            for (int kb = wave; kb < T * MB; kb += WV) dma_global_x4(wbuf + kb * 1024, wsrc + kb * 1024 + lane * 16);
        }
        if (B8_DBG(2)) return;
        int s, blk0;
        source_of(q, s, blk0);
        const void* sp = pick3(a.src, s);
        const int sbytes = pick3(a.src_bytes, s);
        if (!F32SRC || pick3(a.src_layout, s) == MR_LAYOUT_BF16_B8) {
            const int scb = pick3(a.src_cb, s);
            const i32x4 srd = make_srd(sp, sbytes);
#pragma unroll
            for (int j = 0; j < B8_MAX_PPT; ++j) {
                if (!live[j]) continue;                                   // EXEC masks lanes beyond the tile
                const unsigned lrow = buf + (wave * 64 + 64 * WV * j) * 16;      // wave-uniform; lane l lands at + 16 l
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool bok = blk0 + k < scb;                      // padded blocks of the chunk read as zero
                    const int so = ((b * scb + (bok ? blk0 + k : 0)) * HsWs) * 16;
                    dma_buffer_x4(lrow + k * PLANE * 16, (bok && gpix[j] >= 0) ? gpix[j] * 16 : -1, srd, so);
                }
            }
        } else if (F32SRC) {
            const int sc = pick3(a.src_c, s);
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sp, 0, sbytes, 0x00020000);
#pragma unroll
            for (int j = 0; j < B8_MAX_PPT; ++j) {
                if (!live[j]) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int c = (blk0 + k) * 8 + e;                 // wave-uniform
                        const bool cok = c < sc;
                        const int so = ((b * sc + (cok ? c : 0)) * HsWs) * 4;
                        stg[F32SRC ? j : 0][k][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (cok && gpix[j] >= 0) ? gpix[j] * 4 : -1, so, 0));
                    }
            }
        }
    };
    auto stage_store = [&](int q, int pb) {
        if (!F32SRC || B8_DBG(2)) return;
        int s, blk0;
        source_of(q, s, blk0);
        if (pick3(a.src_layout, s) == MR_LAYOUT_BF16_B8) return;
        unsigned char* buf = lds + wres_bytes + pb * stage_bytes;
#pragma unroll
        for (int j = 0; j < B8_MAX_PPT; ++j) {
            if (!live[j]) continue;
            const int p = tid + 64 * WV * j;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                i32x4 v;
                const int jj = F32SRC ? j : 0;
                v.x = (int)pack2(stg[jj][k][0], stg[jj][k][1]);
                v.y = (int)pack2(stg[jj][k][2], stg[jj][k][3]);
                v.z = (int)pack2(stg[jj][k][4], stg[jj][k][5]);
                v.w = (int)pack2(stg[jj][k][6], stg[jj][k][7]);
                *(i32x4*)(buf + (k * PLANE + p) * 16) = v;
            }
        }
    };
    auto advance = [&]() {                                                // load cursor to the next (tile, chunk)
        if (++lq == a.nchunks) {
            lq = 0;
            if (++ltx == a.tiles_x) { ltx = 0; ++lty; }
            if (++ltile < tile_end) place(lty, ltx);
        }
    };

    // ---- this wave's NB pixel blocks -----------------------------------------------------------------------------------------------------
    int prow[NB], pcol[NB], lbase[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pb_ = wave * NB + i;
        prow[i] = pb_ >> 1;
        pcol[i] = (pb_ & 1) * 16 + (lane & 15);
        lbase[i] = ((lane >> 4) * PLANE + prow[i] * a.SH * a.IW + pcol[i] * a.SW) * 16;
    }
    const int g4 = (lane >> 4) * 4;
    // activation as ONE branch-free form: max(x, lo) with lo = x (none), 0 (ReLU: -inf -> 0, no -0.0, like torch.relu), x * p0 (LeakyReLU,
    // 0 <= p0 <= 1: derive8 rejects other slopes) - a switch per element compiled into ~190 scalar branches per tile epilogue (r04_s6: 3 us
    // of every 12 us tile)
    const unsigned keep = a.act == MR_ACT_RELU ? 0u : ~0u;   // lo = (x * slope) AND keep: +0 for ReLU (an AND, not a select: hipcc clones the store loops around a uniform select)
    const float slope = a.act == MR_ACT_LEAKY_RELU ? a.p0 : 1.f;
    // bias of this lane's output channels: once per workgroup (a load per tile costs a memory round trip each - 1 us of the 12 us tile)
    float bias[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = (grp * MB + m) * 16 + g4 + r;
            bias[m][r] = (a.bias && co < a.Cout) ? a.bias[co] : 0.f;
        }
    // B8 destination: a tile's results wait, packed, in `pend` and are stored while the NEXT tile's first chunk is swept (r04_s5: stored right
    // behind the last sweep, 8 bytes per lane, they cost 90 of the 218 us of mask.enc0.1 - every chunk barrier waits for vmcnt(0)).  Two
    // pixel blocks at a time: lanes (pixel, channel quad g) and (pixel, g ^ 1) swap halves through the LDS crossbar (ds_swizzle, no LDS
    // memory), so that every lane stores one complete 16-byte group - 256 contiguous bytes per 16 lanes.
    constexpr int NPAIR = NB >= 2 ? NB / 2 : 1;
    i32x4 pend[MB][NPAIR];
    int pend_ty = -1, pend_tx = 0;
    const bool b8_out = a.dst_layout == MR_LAYOUT_BF16_B8;
    const int dcb = (a.Cout + 7) >> 3;
    // Round 6: SQ_INSTS_VALU / SQ_INSTS_MFMA of this kernel was 3.9 (profiles/r06_c5bf16_b8_counters.txt: 837 VALU and 734 SALU instructions per
    // wave and tile against 216 MFMAs, most of them 64-bit address arithmetic of these stores and the exchange selects of the epilogue, issued
    // while the matrix pipe idles - all 8 waves are in their epilogue together).  The stores now go through a buffer descriptor: the lane's
    // offset inside a tile (row, column, channel-block half) is computed ONCE per workgroup, the tile / cout-block part is a scalar.
    const int half = lane >> 5;                               // which 8-channel block of the 16-channel cout block this lane stores
    int svoff[NPAIR], srow[NPAIR], scol[NPAIR];
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) {
        const int ik = NB >= 2 ? 2 * pr + ((lane >> 4) & 1) : 0;          // even quad: the pair's first pixel block, odd quad: its second
        const int pbk = wave * NB + ik;
        srow[pr] = pbk >> 1;
        scol[pr] = (pbk & 1) * 16 + (lane & 15);
        svoff[pr] = ((srow[pr] * a.ostep_h) * a.dst_W + scol[pr] * a.ostep_w + half * a.dst_H * a.dst_W) * 16;
    }
    bool blk_ok[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) blk_ok[m] = (grp * MB + m) * 2 + half < dcb;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(a.dst, 0, b8_out ? a.dst_bytes : 0, 0x00020000);
    auto flush = [&]() {
        if (pend_ty < 0) return;
        const int oy0 = pend_ty * a.TH, ox0 = pend_tx * 32;
        pend_ty = -1;
        if (B8_DBG(8)) return;
        const int rows_left = a.Ho - oy0, cols_left = a.Wo - ox0;                       // wave-uniform
        const int sbase = ((oy0 * a.ostep_h + ooff_h) * a.dst_W + ox0 * a.ostep_w + ooff_w) * 16;
        const int plane16 = a.dst_H * a.dst_W * 16;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int soff = sbase + (b * dcb + (grp * MB + m) * 2) * plane16;          // wave-uniform: the SGPR offset of the store
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) {
                if (NB >= 2) {
                    if (blk_ok[m] && srow[pr] < rows_left && scol[pr] < cols_left) __builtin_amdgcn_raw_buffer_store_b128(pend[m][pr], drs, svoff[pr], soff, 0);
                } else {                                                  // one pixel block per wave: 8 bytes per lane
                    const int blk = ((grp * MB + m) * 16 + g4) >> 3;
                    if (blk >= dcb) continue;
                    const int oy = oy0 + wave / 2, ox = ox0 + (wave & 1) * 16 + (lane & 15);
                    if (oy >= a.Ho || ox >= a.Wo) continue;
                    const int dy = oy * a.ostep_h + ooff_h, dx = ox * a.ostep_w + ooff_w;
                    unsigned long long* o = (unsigned long long*)a.dst + ((((long long)b * dcb + blk) * a.dst_H + dy) * a.dst_W + dx) * 2 + ((g4 >> 2) & 1);
                    *o = (unsigned long long)(unsigned)pend[m][0].x | ((unsigned long long)(unsigned)pend[m][0].y << 32);
                }
            }
        }
    };

    // taps flattened and software-pipelined over two register sets: the fragments of tap t + 1 are on their way from LDS while the
    // MB * NB MFMAs of tap t issue (first hardware run: one set, 20 % of the MFMA rate - every tap waited out its own ds_reads)
    auto sweep = [&](f32x4 (&acc)[MB][NB], const unsigned char* buf, const unsigned char* wl) {
        bf16x8 av0[MB], bv0[NB], av1[MB], bv1[NB];
        auto frags = [&](bf16x8 (&av)[MB], bf16x8 (&bv)[NB], int t, int tapoff) {
#ifdef MR_B8_ABLATE                                      // diagnostic library: bit 16 = no A (weight) LDS reads, bit 32 = no B (input) LDS reads
            if (a.dbg & 16) {
#pragma unroll
                for (int m = 0; m < MB; ++m) av[m] = __builtin_bit_cast(bf16x8, (i32x4){(int)threadIdx.x + m + t, 0x3f803f80, 0x3f803f80, 0x3f803f80});
            } else
#endif
            {
#pragma unroll
                for (int m = 0; m < MB; ++m) av[m] = *(const bf16x8*)(wl + (t * MB + m) * 1024);
            }
#ifdef MR_B8_ABLATE
            if (a.dbg & 32) {
#pragma unroll
                for (int i = 0; i < NB; ++i) bv[i] = __builtin_bit_cast(bf16x8, (i32x4){(int)threadIdx.x + i + tapoff, 0x3f803f80, 0x3f803f80, 0x3f803f80});
            } else
#endif
            {
#pragma unroll
                for (int i = 0; i < NB; ++i) bv[i] = *(const bf16x8*)(buf + lbase[i] + tapoff);
            }
        };
        auto mfmas = [&](const bf16x8 (&av)[MB], const bf16x8 (&bv)[NB]) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int i = 0; i < NB; ++i) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[m], bv[i], acc[m][i], 0, 0, 0);
        };
        int kh = 0, kw = 0;                                   // tap t + 1 as (kh, kw)
        auto next_off = [&]() {
            if (++kw == KW) { kw = 0; ++kh; }
            return (kh * a.IW + kw) * 16;
        };
        // steady state WITHOUT a branch around either load: hipcc merges the LDS counters of the two paths of `if (t + 1 < T) frags(...)`
        // pessimistically and put s_waitcnt lgkmcnt(0) right behind the reads it had just issued (ISA of rounds 4-5: every pair of taps waited
        // out its own LDS latency - 58 % of the bf16 MFMA rate with every LDS read ablated, profiles/r06_c5bf16_b8_ablation.txt); the tail is peeled
        if (B8_DBG(1)) return;
        frags(av0, bv0, 0, 0);
        int t = 0;
        for (; t + 2 < T; t += 2) {
            frags(av1, bv1, t + 1, next_off());
            mfmas(av0, bv0);
            frags(av0, bv0, t + 2, next_off());
            mfmas(av1, bv1);
        }
        if (t + 1 < T) {
            frags(av1, bv1, t + 1, next_off());
            mfmas(av0, bv0);
            mfmas(av1, bv1);
        } else {
            mfmas(av0, bv0);
        }
    };

    // ---- input pipeline: two LDS stages - chunk q + 1 streams in (B8 sources: LDS-DMA; fp32 sources: plane loads into registers, converted and written
    // behind the sweep) while chunk q is swept; one vmcnt(0) + barrier per chunk.  A RING of up to 8 stages with partial vmcnt waits (the conv_mfma.hip
    // pipeline of this round) was built, was bit-identical, and is NOT kept: at two stages its bookkeeping cost 10-15 % per layer (SGPR spills, the load
    // cursor's address arithmetic ahead of the sweep), and deeper rings won 1-5 % on 5 of 46 layer signatures (tools/sessions/r06_s6.sh, r06_s8.sh,
    // r06_s18.sh; profiles/r06_c5bf16_b8_ablation.txt) - the kernel is not short of loads in flight, it is short of instruction issue (837 VALU + 734
    // SALU instructions per 216 MFMAs of a wave and tile, profiles/r06_c5bf16_b8_counters.txt).
    issue(0, 0);
    stage_store(0, 0);
    advance();
    int pb = 0;
    int sty = tile_begin / a.tiles_x, stx = tile_begin - sty * a.tiles_x;      // the sweep cursor's tile
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        f32x4 acc[MB][NB];                                    // starts at the bias: one add per output value less in the epilogue (the sum then rounds as
#pragma unroll                                                // bias + p1 + p2 + ... instead of (p1 + p2 + ...) + bias: inside the bf16 bar by orders of magnitude)
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int i = 0; i < NB; ++i) acc[m][i] = (f32x4){bias[m][0], bias[m][1], bias[m][2], bias[m][3]};
        for (int q = 0; q < a.nchunks; ++q) {
            const unsigned char* buf = lds + wres_bytes + pb * stage_bytes;
            const unsigned char* wl = (WRES ? lds + q * wchunk_bytes : buf + 64 * PLANE) + lane * 16;
            dma_wait_all();
            __syncthreads();                              // this chunk (and the resident weights) visible; everyone is done with the other buffer
            const bool more = ltile < tile_end;           // wave-uniform
            const int nq = lq;
            // (the DMA instructions spread over the taps of the sweep instead of this burst - what gained 4 % on the F(4x4,3x3) kernel, whose MFMA
            // phase runs from registers - was measured here too, tools/sessions/r04_s36.sh: 181 -> 191 us on mask.enc0.1, configs[4] 324-333 ->
            // 311-315 keyframes/s: this sweep reads its operands from LDS and a wave held by a DMA instruction stops feeding them)
            if (more) issue(nq, pb ^ 1);
            if (q == 0) flush();                          // the previous tile's results leave while this chunk is swept
            sweep(acc, buf, wl);
            if (more) {
                stage_store(nq, pb ^ 1);                  // (fp32 sources) the other buffer is free: everyone passed this chunk's barrier
                advance();
            }
            pb ^= 1;
        }

        // ---- epilogue of the tile: D fragment lane l holds pixel (l & 15), couts (l >> 4) * 4 + r ------------------------------------------
        if (b8_out) {
            // activation: max(x, x * slope) (none: slope 1; LeakyReLU with 0 <= slope <= 1) or max(x, +0) (ReLU) - one wave-uniform branch per tile
            auto finish = [&](auto relu_tag) {
                constexpr bool RELU = decltype(relu_tag)::value;
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    unsigned lo[NB], hi[NB];                       // this lane's 4 channels of every pixel block, as bf16
#pragma unroll
                    for (int i = 0; i < NB; ++i) {
                        float v[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float x = acc[m][i][r];          // (bias included; padded channels: zero weights and zero bias give 0)
                            v[r] = RELU ? fmaxf(x, 0.f) : fmaxf(x, x * slope);
                        }
                        lo[i] = pack2(v[0], v[1]);
                        hi[i] = pack2(v[2], v[3]);
                    }
                    if (NB >= 2) {
#pragma unroll
                        for (int pr = 0; pr < NPAIR; ++pr) {
                            // lanes (pixel, channel quad g) and (pixel, g ^ 1) exchange halves: v_permlane16_swap swaps the ODD 16-lane rows of its
                            // first operand with the EVEN rows of its second - afterwards the even quads hold (own, partner's) quads of block 2 pr and the
                            // odd quads (partner's, own) of block 2 pr + 1, both in channel order: no select, no LDS crossbar instruction
                            const auto l = __builtin_amdgcn_permlane16_swap(lo[2 * pr], lo[2 * pr + 1], false, false);
                            const auto h = __builtin_amdgcn_permlane16_swap(hi[2 * pr], hi[2 * pr + 1], false, false);
                            pend[m][pr] = (i32x4){(int)l[0], (int)h[0], (int)l[1], (int)h[1]};
                        }
                    } else {
                        pend[m][0] = (i32x4){(int)lo[0], (int)hi[0], 0, 0};
                    }
                }
            };
            if (a.act == MR_ACT_RELU) finish(std::true_type{});
            else finish(std::false_type{});
            pend_ty = sty;
            pend_tx = stx;
        } else if (!B8_DBG(8)) {
            const int oy0 = sty * a.TH, ox0 = stx * 32;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int cout0 = (grp * MB + m) * 16 + g4;
                if (cout0 >= a.Cout) continue;
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int oy = oy0 + prow[i], ox = ox0 + pcol[i];
                    if (oy >= a.Ho || ox >= a.Wo) continue;
                    const int dy = oy * a.ostep_h + ooff_h, dx = ox * a.ostep_w + ooff_w;
                    float* o = (float*)a.dst + (((long long)b * a.Cout + cout0) * a.dst_H + dy) * a.dst_W + dx;
                    const long long chs = (long long)a.dst_H * a.dst_W;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (cout0 + r < a.Cout) {
                            const float x = acc[m][i][r];
                            o[r * chs] = fmaxf(x, __uint_as_float(__float_as_uint(x * slope) & keep));
                        }
                }
            }
        }
        if (++stx == a.tiles_x) { stx = 0; ++sty; }
    }
    flush();
}

// ---- element-wise companions on B8 tensors ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ unsigned bfmax2(unsigned a, unsigned b) {     // element-wise max of two packed bf16 pairs (exact: max selects)
    const float lo = fmaxf(bf_lo(a), bf_lo(b)), hi = fmaxf(bf_hi(a), bf_hi(b));
    return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}
__device__ __forceinline__ i32x4 bfmax8(i32x4 a, i32x4 b) {
    return (i32x4){(int)bfmax2(a.x, b.x), (int)bfmax2(a.y, b.y), (int)bfmax2(a.z, b.z), (int)bfmax2(a.w, b.w)};
}

// nn.MaxPool2d(2) of the next mask-encoder stage and the maximum over the frames (monorec_model.py:357-365) from one read:
// src (frames, planes, H, W) groups of 16 bytes -> pooled (frames, planes, H/2, W/2), fmax (planes, H, W); planes = batch * channel blocks
__global__ __launch_bounds__(256) void pool2x2_framemax_b8_kernel(const i32x4* __restrict__ src, i32x4* __restrict__ pooled, i32x4* __restrict__ fmax,
                                                                   int frames, int planes, int H, int W) {
    const int Wh = W >> 1, Hh = H >> 1;
    const long long total = (long long)planes * Hh * Wh;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const int x = (int)(i % Wh), y = (int)((i / Wh) % Hh);
    const long long pl = i / ((long long)Wh * Hh);
    i32x4 m00, m01, m10, m11;
    for (int f = 0; f < frames; ++f) {
        const i32x4* s = src + (((long long)f * planes + pl) * H + 2 * y) * W + 2 * x;
        const i32x4 a = s[0], b = s[1], c = s[W], d = s[W + 1];
        pooled[(((long long)f * planes + pl) * Hh + y) * Wh + x] = bfmax8(bfmax8(a, b), bfmax8(c, d));
        if (f == 0) { m00 = a; m01 = b; m10 = c; m11 = d; }
        else { m00 = bfmax8(m00, a); m01 = bfmax8(m01, b); m10 = bfmax8(m10, c); m11 = bfmax8(m11, d); }
    }
    i32x4* o = fmax + (pl * H + 2 * y) * W + 2 * x;
    o[0] = m00; o[1] = m01; o[W] = m10; o[W + 1] = m11;
}

__global__ __launch_bounds__(256) void max_over_frames_b8_kernel(const i32x4* __restrict__ src, i32x4* __restrict__ dst, int frames, long long count) {
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= count) return;
    i32x4 m = src[i];
    for (int f = 1; f < frames; ++f) m = bfmax8(m, src[(long long)f * count + i]);
    dst[i] = m;
}

// layout conversions (tests, and the few places where a B8 map must be handed to an fp32 consumer)
__global__ __launch_bounds__(256) void f32_to_b8_kernel(const float* __restrict__ src, i32x4* __restrict__ dst, int n, int c, long long hw) {
    const int cb = (c + 7) >> 3;
    const long long total = (long long)n * cb * hw;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const long long p = i % hw;
    const int blk = (int)((i / hw) % cb), b = (int)(i / (hw * cb));
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = blk * 8 + e < c ? src[((long long)b * c + blk * 8 + e) * hw + p] : 0.f;
    dst[i] = (i32x4){(int)pack2(v[0], v[1]), (int)pack2(v[2], v[3]), (int)pack2(v[4], v[5]), (int)pack2(v[6], v[7])};
}
__global__ __launch_bounds__(256) void b8_to_f32_kernel(const i32x4* __restrict__ src, float* __restrict__ dst, int n, int c, long long hw) {
    const int cb = (c + 7) >> 3;
    const long long total = (long long)n * cb * hw;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= total) return;
    const long long p = i % hw;
    const int blk = (int)((i / hw) % cb), b = (int)(i / (hw * cb));
    const i32x4 v = src[i];
    const unsigned u[4] = {(unsigned)v.x, (unsigned)v.y, (unsigned)v.z, (unsigned)v.w};
#pragma unroll
    for (int e = 0; e < 8; ++e)
        if (blk * 8 + e < c) dst[((long long)b * c + blk * 8 + e) * hw + p] = (e & 1) ? bf_hi(u[e >> 1]) : bf_lo(u[e >> 1]);
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------------------
struct B8Derived {
    B8Args k;
    dim3 grid;
    size_t lds_bytes;
    int mb, nb, wv;
    bool wres;
};

bool valid_mb8(int mb) { return mb >= 1 && mb <= 4; }

int derive8(const mr_b8_conv_desc* d, B8Derived* out) {
    if (!d || d->num_src < 1 || d->num_src > MR_MAX_SOURCES || d->batch < 1 || d->kh < 1 || d->kw < 1 || d->stride_h < 1 || d->stride_w < 1 ||
        d->stride_h > 2 || d->stride_w > 2 || d->out_h < 1 || d->out_w < 1 || d->out_channels < 1 || !d->dst)
        return MR_ERR_BAD_ARGUMENT;
    const int mb = d->cout_blocks_per_wg, nb = d->pixel_blocks_per_wave, wv = d->waves_per_wg;
    if (!valid_mb8(mb) || !(nb == 1 || nb == 2 || nb == 4) || !(wv == 4 || wv == 8)) return MR_ERR_BAD_ARGUMENT;
    if (d->dst_layout != MR_LAYOUT_F32_NCHW && d->dst_layout != MR_LAYOUT_BF16_B8) return MR_ERR_BAD_ARGUMENT;
    if (d->activation != MR_ACT_NONE && d->activation != MR_ACT_RELU && d->activation != MR_ACT_LEAKY_RELU) return MR_ERR_UNSUPPORTED;
    if (d->activation == MR_ACT_LEAKY_RELU && !(d->act_p0 >= 0.f && d->act_p0 <= 1.f)) return MR_ERR_UNSUPPORTED;   // the epilogue is max(x, x * slope)
    const int nphase = d->num_phases <= 1 ? 1 : d->num_phases;
    if (nphase != 1 && nphase != 4) return MR_ERR_BAD_ARGUMENT;
    B8Args& k = out->k;
    memset(&k, 0, sizeof(k));
    int nchunks = 0;
    for (int s = 0; s < d->num_src; ++s) {
        if (!d->src[s] || d->src_channels[s] < 1) return MR_ERR_BAD_ARGUMENT;
        if (d->src_layout[s] != MR_LAYOUT_F32_NCHW && d->src_layout[s] != MR_LAYOUT_BF16_B8) return MR_ERR_BAD_ARGUMENT;
        k.src[s] = d->src[s];
        k.src_c[s] = d->src_channels[s];
        k.src_cb[s] = (d->src_channels[s] + 7) / 8;
        k.src_layout[s] = d->src_layout[s];
        const long long bytes = d->src_layout[s] == MR_LAYOUT_BF16_B8 ? (long long)d->batch * k.src_cb[s] * d->src_h * d->src_w * 16
                                                                      : (long long)d->batch * d->src_channels[s] * d->src_h * d->src_w * 4;
        if (bytes >= (1ll << 31)) return MR_ERR_UNSUPPORTED;          // 32-bit byte offsets through the buffer descriptor
        k.src_bytes[s] = (int)bytes;
        k.src_q0[s] = nchunks;
        nchunks += (k.src_cb[s] + 3) / 4;
    }
    k.nsrc = d->num_src;
    k.nchunks = nchunks;
    k.Hs = d->src_h; k.Ws = d->src_w;
    k.SH = d->stride_h; k.SW = d->stride_w;
    k.KH = d->kh; k.KW = d->kw;
    k.Ho = d->out_h; k.Wo = d->out_w;
    k.dst = d->dst; k.dst_layout = d->dst_layout;
    if (d->dst_layout == MR_LAYOUT_BF16_B8) {
        const long long db = (long long)d->batch * ((d->out_channels + 7) / 8) * d->dst_plane_h * d->dst_plane_w * 16;
        if (db >= (1ll << 31)) return MR_ERR_UNSUPPORTED;             // 32-bit byte offsets through the buffer descriptor
        k.dst_bytes = (int)db;
    }
    k.dst_H = d->dst_plane_h; k.dst_W = d->dst_plane_w;
    k.ostep_h = d->out_step_h < 1 ? 1 : d->out_step_h;
    k.ostep_w = d->out_step_w < 1 ? 1 : d->out_step_w;
    k.Cout = d->out_channels;
    k.CB16 = (d->out_channels + 15) / 16;
    k.bias = d->bias; k.act = d->activation; k.p0 = d->act_p0;
    k.nphase = nphase; k.batch = d->batch;
    const int ngroups = (k.CB16 + mb - 1) / mb;
    for (int p = 0; p < nphase; ++p) {
        k.w[p] = d->phase_weights[p];
        k.KHp[p] = d->phase_kh[p] > 0 ? d->phase_kh[p] : d->kh;
        k.KWp[p] = d->phase_kw[p] > 0 ? d->phase_kw[p] : d->kw;
        k.PT[p] = d->phase_pad_top[p]; k.PL[p] = d->phase_pad_left[p];
        k.ooff_h[p] = d->phase_out_off_h[p]; k.ooff_w[p] = d->phase_out_off_w[p];
        if (!k.w[p] || k.KHp[p] > d->kh || k.KWp[p] > d->kw || k.ooff_h[p] < 0 || k.ooff_w[p] < 0) return MR_ERR_BAD_ARGUMENT;
        if ((k.Ho - 1) * k.ostep_h + k.ooff_h[p] >= k.dst_H || (k.Wo - 1) * k.ostep_w + k.ooff_w[p] >= k.dst_W) return MR_ERR_BAD_ARGUMENT;
        k.wgroup_bytes[p] = (long long)nchunks * k.KHp[p] * k.KWp[p] * mb * 1024;
    }
    const int blocks = wv * nb;                       // pixel blocks of 16 a workgroup owns: TH rows x 2 blocks
    if (blocks < 2) return MR_ERR_BAD_ARGUMENT;
    k.TH = blocks / 2;
    k.tiles_x = (d->out_w + 31) / 32;
    const int tiles_y = (d->out_h + k.TH - 1) / k.TH;
    k.IH = (k.TH - 1) * k.SH + k.KH;
    k.IW = 31 * k.SW + k.KW;
    // (IH * IW positions between the four channel blocks: by the bank arithmetic of tools/lds_banks.py a pitch of 0 mod 16 - odd at column stride 2 - makes the
    // 16-byte B-fragment reads conflict free, and it measured 0.8 % SLOWER over the configs[4] keyframe (tools/sessions/r05_s14.sh): these reads cost their
    // latency, not LDS cycles.  Not kept.)
    k.PLANE = k.IH * k.IW;
    if ((k.PLANE + 64 * wv - 1) / (64 * wv) > B8_MAX_PPT) return MR_ERR_UNSUPPORTED;
    const long long ntiles = (long long)k.tiles_x * tiles_y;
    if (ntiles >= (1ll << 31) || ngroups >= 65536 || (long long)d->batch * nphase >= 65536) return MR_ERR_UNSUPPORTED;
    k.ntiles = (int)ntiles;
    // resident weights where the whole stream of a cout group fits next to the two input stages (every layer with few input channels)
    const size_t wall = (size_t)nchunks * k.KH * k.KW * mb * 1024;
    const size_t tile2 = 2 * (size_t)64 * k.PLANE;
    out->wres = wall + tile2 <= 160 * 1024 && wall <= 112 * 1024;
    if (out->wres) {
        out->lds_bytes = wall + tile2;
        const long long jobs = ntiles * ngroups * d->batch * nphase;
        long long wgs_target = 1024;                          // ~4 workgroups per CU over the launch, each keeps its weights for tpw tiles
#ifdef MR_B8_ABLATE
        { const char* e = getenv("MR_B8_WGS"); if (e && atoi(e) > 0) wgs_target = atoi(e); }
#endif
        long long tpw = (jobs + wgs_target - 1) / wgs_target;
        if (tpw < 1) tpw = 1;
        if (tpw > 64) tpw = 64;
        k.tiles_per_wg = (int)tpw;
    } else {
        out->lds_bytes = 2 * ((size_t)64 * k.PLANE + (size_t)1024 * k.KH * k.KW * mb);
        k.tiles_per_wg = 1;
    }
    if (out->lds_bytes > 160 * 1024) return MR_ERR_LDS_BUDGET;
    out->grid = dim3((unsigned)((ntiles + k.tiles_per_wg - 1) / k.tiles_per_wg), (unsigned)ngroups, (unsigned)(d->batch * nphase));
    out->mb = mb; out->nb = nb; out->wv = wv;
#ifdef MR_B8_ABLATE
    { const char* e = getenv("MR_B8_DBG"); k.dbg = e ? atoi(e) : 0; }
#endif
    return 0;
}

template <int MB, int NB, int WV, bool WRES, bool F32SRC>
int launch8w(const B8Derived& dv, hipStream_t stream) {
    static std::atomic<unsigned long long> attr_set{0};      // dynamic-LDS ceiling once per instantiation AND device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_b8_kernel<MB, NB, WV, WRES, F32SRC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv_b8_kernel<MB, NB, WV, WRES, F32SRC>), dv.grid, dim3(WV * 64), dv.lds_bytes, stream, dv.k);
    return (int)hipGetLastError();
}

template <int MB, int NB, int WV>
int launch8(const B8Derived& dv, hipStream_t stream) {
    bool f32 = false;
    for (int s = 0; s < dv.k.nsrc; ++s) f32 = f32 || dv.k.src_layout[s] == MR_LAYOUT_F32_NCHW;
    if (f32) return dv.wres ? launch8w<MB, NB, WV, true, true>(dv, stream) : launch8w<MB, NB, WV, false, true>(dv, stream);
    return dv.wres ? launch8w<MB, NB, WV, true, false>(dv, stream) : launch8w<MB, NB, WV, false, false>(dv, stream);
}

template <int MB>
int launch8_mb(const B8Derived& dv, hipStream_t stream) {
    if (dv.wv == 8) {
        switch (dv.nb) {
            case 1: return launch8<MB, 1, 8>(dv, stream);
            case 2: return launch8<MB, 2, 8>(dv, stream);
            default: return launch8<MB, 4, 8>(dv, stream);
        }
    }
    switch (dv.nb) {
        case 1: return launch8<MB, 1, 4>(dv, stream);
        case 2: return launch8<MB, 2, 4>(dv, stream);
        default: return launch8<MB, 4, 4>(dv, stream);
    }
}

uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu)) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

}  // namespace

extern "C" size_t mr_b8_packed_weight_bytes(int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t kh, int32_t kw, int32_t mb) {
    if (!src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_mb8(mb) || out_channels < 1 || kh < 1 || kw < 1) return 0;
    int nchunks = 0;
    for (int s = 0; s < num_src; ++s) nchunks += ((src_channels[s] + 7) / 8 + 3) / 4;
    const int groups = ((out_channels + 15) / 16 + mb - 1) / mb;
    return (size_t)groups * nchunks * kh * kw * mb * 1024;
}

// weight: (out_channels, sum(src_channels), kh, kw) fp32, nn.Conv2d layout.  Stream: [cout group of 16 mb][chunk of 32 channels, source-major]
// [tap][cout block][64 lanes][8 bf16]; lane l = (cout l & 15 of the block, channel block l >> 4 of the chunk), element e = channel 8 (l >> 4) + e
// of the chunk.  Padded channels / output channels are zero.
extern "C" int mr_b8_pack_weights(const float* weight, int32_t out_channels, const int32_t* src_channels, int32_t num_src, int32_t kh, int32_t kw,
                                  int32_t mb, void* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES || !valid_mb8(mb) || out_channels < 1 || kh < 1 || kw < 1)
        return MR_ERR_BAD_ARGUMENT;
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    const int taps = kh * kw;
    const int groups = ((out_channels + 15) / 16 + mb - 1) / mb;
    uint16_t* o = (uint16_t*)dst;
    for (int g = 0; g < groups; ++g) {
        int cin_off = 0;
        for (int s = 0; s < num_src; ++s) {
            const int nch = ((src_channels[s] + 7) / 8 + 3) / 4;
            for (int q = 0; q < nch; ++q)
                for (int tap = 0; tap < taps; ++tap)
                    for (int m = 0; m < mb; ++m)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 8; ++e) {
                                const int cout = (g * mb + m) * 16 + (lane & 15);
                                const int cl = q * 32 + (lane >> 4) * 8 + e;
                                float v = 0.f;
                                if (cout < out_channels && cl < src_channels[s]) v = weight[((size_t)cout * cin_total + (cin_off + cl)) * taps + tap];
                                *o++ = bf16_rne(v);
                            }
            cin_off += src_channels[s];
        }
    }
    return 0;
}

extern "C" int64_t mr_conv2d_b8_lds_bytes(const mr_b8_conv_desc* desc) {
    B8Derived dv;
    const int rc = derive8(desc, &dv);
    return rc != 0 ? rc : (int64_t)dv.lds_bytes;
}

extern "C" int mr_conv2d_b8(const mr_b8_conv_desc* desc, void* stream) {
    B8Derived dv;
    const int rc = derive8(desc, &dv);
    if (rc != 0) return rc;
    switch (dv.mb) {
        case 1: return launch8_mb<1>(dv, (hipStream_t)stream);
        case 2: return launch8_mb<2>(dv, (hipStream_t)stream);
        case 3: return launch8_mb<3>(dv, (hipStream_t)stream);
        default: return launch8_mb<4>(dv, (hipStream_t)stream);
    }
}

extern "C" int mr_pool2x2_framemax_b8(const void* src, void* pooled, void* fmax, int32_t frames, int64_t planes, int32_t h, int32_t w, void* stream) {
    if (!src || !pooled || !fmax || frames < 1 || planes < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || planes >= (1ll << 31)) return MR_ERR_BAD_ARGUMENT;
    const long long total = planes * (h / 2) * (w / 2);
    hipLaunchKernelGGL(pool2x2_framemax_b8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const i32x4*)src,
                       (i32x4*)pooled, (i32x4*)fmax, frames, (int)planes, h, w);
    return (int)hipGetLastError();
}

extern "C" int mr_max_over_frames_b8(const void* src, void* dst, int32_t frames, int64_t groups16, void* stream) {
    if (!src || !dst || frames < 1 || groups16 < 1) return MR_ERR_BAD_ARGUMENT;
    hipLaunchKernelGGL(max_over_frames_b8_kernel, dim3((unsigned)((groups16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const i32x4*)src,
                       (i32x4*)dst, frames, (long long)groups16);
    return (int)hipGetLastError();
}

extern "C" int mr_f32_nchw_to_b8(const float* src, void* dst, int32_t n, int32_t c, int64_t hw, void* stream) {
    if (!src || !dst || n < 1 || c < 1 || hw < 1) return MR_ERR_BAD_ARGUMENT;
    const long long total = (long long)n * ((c + 7) / 8) * hw;
    hipLaunchKernelGGL(f32_to_b8_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (i32x4*)dst, n, c, (long long)hw);
    return (int)hipGetLastError();
}

extern "C" int mr_b8_to_f32_nchw(const void* src, float* dst, int32_t n, int32_t c, int64_t hw, void* stream) {
    if (!src || !dst || n < 1 || c < 1 || hw < 1) return MR_ERR_BAD_ARGUMENT;
    const long long total = (long long)n * ((c + 7) / 8) * hw;
    hipLaunchKernelGGL(b8_to_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const i32x4*)src, dst, n, c, (long long)hw);
    return (int)hipGetLastError();
}
