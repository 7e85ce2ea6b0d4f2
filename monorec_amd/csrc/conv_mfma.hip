// Direct convolution on the fp32 matrix cores of gfx950 (MI355X).
//
// Stands in for every nn.Conv2d / ConvTranspose2d on the MonoRec inference path
// (reference: model/layers.py:241-252,289-356,380-400; model/monorec/monorec_model.py:118-129,
// 345-385,526-557).  See conv_layout.h for the GEMM view and the packed weight stream.
//
// Workgroup = WV (4 or 8) waves.  It owns a TH x (TWB*16) tile of one output plane (WV*NB blocks of 16 pixels) and MB
// consecutive 16-channel output blocks.  K is walked in chunks of <= CK input channels of one concat source through two
// LDS buffers (one when a workgroup only ever sees one chunk); per chunk:
//   * the haloed input tile goes global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: 16-byte groups of a 4-aligned
//     tile row; dword DMA for the nearest-upsample read): out-of-range SRD offsets make the hardware write zeros, so
//     padding, padded channels and the halo need no branch and no VGPR.  A second instantiation stages through
//     registers instead (2x2 max-pool / ResNet input normalisation while loading);
//   * the chunk's A fragments (one contiguous block of the packed stream) follow with global_load_lds_dwordx4;
//   * after one barrier every wave sweeps taps x channel-quads reading A (lane-linear) and B (plane stride = 16 mod 32,
//     conflict free) from LDS and issues MB*NB v_mfma_f32_16x16x4_f32 per k-step (or, MR_COMPUTE_BF16, one
//     v_mfma_f32_16x16x16_bf16 per 16 channels with the B fragment rounded to bf16 on the way) - the sweep contains no
//     global memory access at all, and the next chunk streams into the other buffer meanwhile.
// The schedule (MB, NB, split_k, CK, WV) per layer shape is measured, not modelled (tools/tune_conv.py).  fp32 MFMA is an
// exact fmaf chain, so results differ from the oneDNN CPU reference only by summation order.
//
// Epilogue: bias (eval-BatchNorm folded by the host), residual add, activation, scatter with an
// output step/offset into a channel slice of the destination.  The four output parities of
// ConvTranspose2d(k4,s2) run as four "phases" of ONE launch.  split_k > 1 writes raw partial sums to a
// workspace; splitk_epilogue_kernel finishes them in a fixed order (deterministic, no atomics).  (Finishing
// inside the launch - last workgroup of a tile to arrive sums the slices - was built and measured: the
// device-scope release/acquire fences it needs write back / invalidate a whole XCD L2 each, +20 us per layer.)
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <type_traits>

#include "../../include/monorec_hip.h"
#include "conv_layout.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MR_MAX_PPT 6   // tile positions staged per thread (ceil(IH*IW / 256))
#define MR_MAX_G4 4    // 16-byte groups per lane per channel plane in the dwordx4 DMA path

struct ConvKArgs {
    const float* src[MR_MAX_SOURCES];
    int src_bytes[MR_MAX_SOURCES];   // whole source tensor, < 2 GiB (buffer descriptor range)
    int src_c[MR_MAX_SOURCES];
    int src_cpad[MR_MAX_SOURCES];
    int nsrc;
    int Hs, Ws, Hin, Win;
    int in_mode, in_tf;
    int KH, KW, SH, SW;
    int Ho, Wo;
    float* dst;
    long long dst_bstride;
    int dst_H, dst_W, ch_off;
    int ostep_h, ostep_w;
    int Cout, CB;
    const float* bias;
    const float* res;
    int act;
    float p0, p1;
    int tiles_x, TH, TWB;
    int IH, IW, PLANE, CK, ppt;
    int wmax_floats;           // A-fragment floats of the largest chunk (per pipeline buffer)
    int dma_in;                // input tile staged by buffer_load ... lds (direct / upsample reads)
    int dma_x4;                // ... as aligned 16-byte groups (direct reads, source width % 4 == 0)
    int IWa;                   // LDS row pitch of the tile (IW, or the 4-aligned superset for dma_x4)
    int g4pt;                  // 16-byte groups per lane per channel plane (dma_x4)
    int xsh[4];                // per phase: columns between the aligned tile origin and the first tap column
    unsigned gpr_magic;        // floor(r / (IWa/4)) == (r * gpr_magic) >> 16 for r < 256 (dma_x4 prologue)
    unsigned tiles_x_magic;    // floor(t / tiles_x) == (t * tiles_x_magic) >> 32 for t < 2^16
    int dbg;                   // ablation bits (env MR_CONV_DBG): 1 skip sweep, 2 skip input DMA, 4 skip weight DMA, 8 skip stores,
                               // 16 per-workgroup timestamps (tools/wg_timeline.py)
    int ksplit, nchunks, batch, nphase;
    int kws;                   // 1: the waves of a workgroup split K (all work on the same NB pixel blocks, reduced through LDS)
    int bf16;                  // 1: bf16 MFMA mode (weights bf16, activations rounded to bf16 in the B fragment): half-size weight blocks
    int tiles_y, ngroups, ks_shift;   // ks_shift: log2(ksplit) or -1
    long long wgroup_stride[4];   // packed floats per cout group, per phase
    int KHp[4], KWp[4];        // taps swept by each phase (<= KH, KW; the common KH / KW size the input tile)
    float* ws;
    const float* w[4];         // per phase
    int PT[4], PL[4], ooff_h[4], ooff_w[4];
};

__device__ __forceinline__ float mr_activate(float v, int act, float p0, float p1) {
    switch (act) {
        case MR_ACT_RELU: return v > 0.f ? v : 0.f;
        case MR_ACT_LEAKY_RELU: return v > 0.f ? v : v * p0;
        case MR_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case MR_ACT_ABS_TANH_AFFINE: {
            float t = fabsf(tanhf(v));
            return (1.f - t) * p0 + t * p1;   // monorec_model.py:717
        }
        default: return v;
    }
}

// Stage 16 channel planes of one tile position through a buffer descriptor: an offset of -1 is out of
// range for the SRD, so the hardware returns 0 - zero padding, padded channels and inactive lanes need
// no branch, and the 16 buffer_load_dword are issued back to back (one memory latency per batch).
template <bool POOL, bool NORM>
__device__ __forceinline__ void stage_position(float* __restrict__ ldsI, __amdgpu_buffer_rsrc_t rsrc, int voff, int sbase_bytes,
                                               int cs, int ck, int creal, int HsWs, int Ws, int PLANE, int lo) {
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const bool cok = cs + c < creal;                          // wave-uniform
        const int vo = cok ? voff : -1;
        const int so = sbase_bytes + (cs + c) * HsWs * 4;         // wave-uniform -> SGPR soffset
        float x;
        if (POOL) {
            const int v1 = vo < 0 ? -1 : vo + 4, v2 = vo < 0 ? -1 : vo + Ws * 4, v3 = vo < 0 ? -1 : vo + Ws * 4 + 4;
            const float x0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, so, 0));
            const float x1 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, v1, so, 0));
            const float x2 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, v2, so, 0));
            const float x3 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, v3, so, 0));
            x = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
        } else {
            x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, vo, so, 0));
        }
        // out-of-range samples loaded as 0 must stay 0 after the normalisation: multiply by a 0/1 lane mask
        // (arithmetic, so the compiler cannot sink the load under a branch; the operand is always finite)
        if (NORM) x = (((x + 0.5f) - 0.45f) / 0.225f) * (vo < 0 ? 0.f : 1.f);
        v[c] = x;
    }
#pragma unroll
    for (int c = 0; c < 16; c += 4)
        if (cs + c < ck) {                                        // ck is a multiple of 4
#pragma unroll
            for (int u = 0; u < 4; ++u) ldsI[(cs + c + u) * PLANE + lo] = v[c + u];
        }
}

template <int MB, int NB>
__device__ __forceinline__ void kstep(f32x4 (&acc)[MB][NB], const float* __restrict__ wt, const float* __restrict__ ldsI,
                                      const int (&lbase)[NB], int c4, int off) {
    float av[MB], bv[NB];
#pragma unroll
    for (int m = 0; m < MB; ++m) av[m] = wt[(c4 * MB + m) * 64];
#pragma unroll
    for (int i = 0; i < NB; ++i) bv[i] = ldsI[lbase[i] + off];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[i], acc[m][i], 0, 0, 0);
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// ---- LDS-DMA, issued through inline asm ------------------------------------------------------------------
// hipcc treats an LDS-DMA builtin as a store that may alias every later ds_read and drains it (vmcnt(0))
// before the MFMA sweep of the *other* pipeline buffer, which serialises load and compute.  Inline asm is
// invisible to that pass; the kernel waits itself (dma_wait_all) right before the barrier that publishes the
// buffer.  M0 (LDS base of the DMA) is saved/restored inside the statement; `s_nop 4` covers the
// SALU/VALU-write -> VMEM-SGPR-read hazard the compiler does not pad for asm operands.
__device__ __forceinline__ void dma_buffer_dword(unsigned lds_byte_addr, int voff, i32x4 srd, int soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dword %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte_addr), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_buffer_x4(unsigned lds_byte_addr, int voff, i32x4 srd, int soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte_addr), "v"(voff), "s"(srd), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma_global_x4(unsigned lds_byte_addr, const float* g) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte_addr), "v"(g) : "memory");
}
__device__ __forceinline__ void dma_global_x1(unsigned lds_byte_addr, const float* g) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_byte_addr), "v"(g) : "memory");
}
// Ablation / timeline switches exist only in the diagnostic library (python -m monorec_amd.build --timeline, -DMR_CONV_TIMELINE):
// in the product kernel MR_DBG(bit) is a compile-time 0 - carried as run-time tests they cost ~3 % keyframes/s.
#ifdef MR_CONV_TIMELINE
#define MR_DBG(bit) (a.dbg & (bit))
#else
#define MR_DBG(bit) 0
#endif

// MR_CONV_DBG bit 16 (tools/wg_timeline.py, library built with -DMR_CONV_TIMELINE): thread 0 of every workgroup drops
// 100 MHz timestamps into the workspace
__device__ __forceinline__ void dbg_stamp(const ConvKArgs& a, int k) {
#ifndef MR_CONV_TIMELINE      // the stamps cost ~3 % keyframes/s even when switched off (SGPR pressure): opt-in at compile time
    (void)a; (void)k;
#else
    if ((a.dbg & 16) && threadIdx.x == 0) {
        const long long wg = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z);
        // split-K launches keep their slabs at the head of the workspace: the stamps go behind them (tools/wg_timeline.py sizes the buffer)
        float* base = a.ws + (a.ksplit > 1 ? (long long)a.ksplit * a.nphase * a.batch * (a.CB * 16) * a.Ho * a.Wo : 0);
        ((unsigned long long*)base)[wg * 12 + k] = (k == 9 || k == 10) ? clock64() : wall_clock64();
    }
#endif
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

// K-chunk cursor over the concatenated sources (chunks never straddle two sources).
struct ChunkCursor {
    int s, c0;
    long long woff;
};

template <int MB>
__device__ __forceinline__ void cursor_advance(const ConvKArgs& a, ChunkCursor& c, int T) {
    const int ck = min(a.CK, a.src_cpad[c.s] - c.c0);
    c.woff += (long long)T * ck * MB * (a.bf16 ? 8 : 16);
    c.c0 += a.CK;
    if (c.c0 >= a.src_cpad[c.s]) { c.c0 = 0; ++c.s; }
}

// Issue everything chunk `c` needs into LDS buffer (ldsI, ldsW):
//   * A fragments: one contiguous block, global_load_lds_dwordx4 (1 KiB per wave instruction);
//   * input tile: DMA path (direct / upsample reads): buffer_load_dword ... lds, one 256 B row of the tile per
//     wave instruction, hardware zero fill for out-of-range offsets; no VGPRs, nothing to wait for here;
//     register path (2x2 max-pool or input normalisation): SRD loads -> VALU -> ds_write.
template <int MB, bool DMA_IN>
__device__ __forceinline__ void issue_chunk(const ConvKArgs& a, const ChunkCursor& c, float* ldsI, float* ldsW,
                                             unsigned ldsI_addr, unsigned ldsW_addr,
                                             const float* wgrp, int b, int T, int lane, int wave, int nwave, int HsWs,
                                             const int (&goff)[MR_MAX_PPT], const int (&loff)[MR_MAX_PPT],
                                             const int (&voff4)[MR_MAX_G4]) {
    const int ck = min(a.CK, a.src_cpad[c.s] - c.c0);
    const int wfloats = T * ck * MB * (a.bf16 ? 8 : 16);
    const float* wsrc = wgrp + c.woff;
    const int n1k = MR_DBG(4) ? 0 : wfloats >> 8;   // 1 KiB pieces (64 lanes x 16 B)
    for (int kb = wave; kb < n1k; kb += nwave) dma_global_x4(ldsW_addr + kb * 1024, wsrc + kb * 256 + lane * 4);
    const int nfrag = MR_DBG(4) ? 0 : wfloats >> 6; // tail: 256 B pieces (64 lanes x 4 B)
    for (int fr = (n1k << 2) + wave; fr < nfrag; fr += nwave) dma_global_x1(ldsW_addr + fr * 256, wsrc + fr * 64 + lane);

    const int creal = a.src_c[c.s] - c.c0;                            // real (unpadded) channels left
    const int sbase_bytes = (b * a.src_c[c.s] + c.c0) * HsWs * 4;     // byte offset of channel c0 of sample b
    if (MR_DBG(2)) return;
    if (DMA_IN && a.dma_x4) {
        // one buffer_load_dwordx4 ... lds = 64 lanes x 16 B = up to 256 consecutive floats of one channel plane;
        // wave w streams channels w, w+4, ... of the chunk
        const i32x4 srd = make_srd(a.src[c.s], a.src_bytes[c.s]);
        for (int cc = wave; cc < ck; cc += nwave) {
            const unsigned lplane = ldsI_addr + cc * a.PLANE * 4;
            const int so = sbase_bytes + cc * HsWs * 4;
            const bool cok = cc < creal;
#pragma unroll
            for (int i = 0; i < MR_MAX_G4; ++i) {
                if (i < a.g4pt && voff4[i] != -2)                      // -2: lane beyond the tile rows (EXEC off)
                    dma_buffer_x4(lplane + i * 1024, cok ? voff4[i] : -1, srd, so);
            }
        }
    } else if (DMA_IN) {
        const i32x4 srd = make_srd(a.src[c.s], a.src_bytes[c.s]);
#pragma unroll
        for (int j = 0; j < MR_MAX_PPT; ++j) {
            if (j < a.ppt && loff[j] >= 0) {                          // EXEC masks lanes beyond the tile
                const int voff = goff[j] >= 0 ? goff[j] * 4 : -1;     // -1: out of range -> hardware writes 0
                const unsigned lrow = ldsI_addr + (wave * 64 + 256 * j) * 4;   // wave-uniform; lane l lands at +4l
                for (int cc = 0; cc < ck; ++cc) {
                    const int vo = cc < creal ? voff : -1;            // padded channels read as zero
                    dma_buffer_dword(lrow + cc * a.PLANE * 4, vo, srd, sbase_bytes + cc * HsWs * 4);
                }
            }
        }
    } else {
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.src[c.s], 0, a.src_bytes[c.s], 0x00020000);
        for (int cs = 0; cs < ck; cs += 16) {
#pragma unroll
            for (int j = 0; j < MR_MAX_PPT; ++j) {
                if (j < a.ppt && loff[j] >= 0) {
                    const int voff = goff[j] >= 0 ? goff[j] * 4 : -1;
                    if (a.in_mode == MR_IN_MAXPOOL2)
                        stage_position<true, false>(ldsI, rsrc, voff, sbase_bytes, cs, ck, creal, HsWs, a.Ws, a.PLANE, loff[j]);
                    else if (a.in_tf == MR_TF_RESNET_NORM)
                        stage_position<false, true>(ldsI, rsrc, voff, sbase_bytes, cs, ck, creal, HsWs, a.Ws, a.PLANE, loff[j]);
                    else
                        stage_position<false, false>(ldsI, rsrc, voff, sbase_bytes, cs, ck, creal, HsWs, a.Ws, a.PLANE, loff[j]);
                }
            }
        }
    }
}

template <int MB, int NB>
__device__ __forceinline__ void sweep_chunk(const ConvKArgs& a, f32x4 (&acc)[MB][NB], const float* ldsI, const float* ldsW,
                                            const int (&lbase)[NB], int ck4, int lane, int KH, int KW) {
    const float* wl = ldsW + lane;
    // MB = NB = 1 has a single accumulator: every MFMA waits for the previous one to retire.  Two partial sums
    // (even / odd channel quads) keep two MFMAs in flight; they are added once per chunk.
    constexpr bool DUAL = MB * NB == 1;
    f32x4 acc2[MB][NB];
    if (DUAL) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int i = 0; i < NB; ++i) acc2[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    for (int kh = 0; kh < KH; ++kh) {
        for (int kw = 0; kw < KW; ++kw) {
            const int tapoff = kh * a.IWa + kw;
            const float* wt = wl + (kh * KW + kw) * ck4 * (MB * 64);
            int c4 = 0;
            for (; c4 + 4 <= ck4; c4 += 4) {          // manual 4x unroll: LDS reads of 4 k-steps overlap
                kstep<MB, NB>(acc, wt, ldsI, lbase, c4, c4 * 4 * a.PLANE + tapoff);
                kstep<MB, NB>(DUAL ? acc2 : acc, wt, ldsI, lbase, c4 + 1, (c4 + 1) * 4 * a.PLANE + tapoff);
                kstep<MB, NB>(acc, wt, ldsI, lbase, c4 + 2, (c4 + 2) * 4 * a.PLANE + tapoff);
                kstep<MB, NB>(DUAL ? acc2 : acc, wt, ldsI, lbase, c4 + 3, (c4 + 3) * 4 * a.PLANE + tapoff);
            }
            for (; c4 < ck4; ++c4) kstep<MB, NB>(acc, wt, ldsI, lbase, c4, c4 * 4 * a.PLANE + tapoff);
        }
    }
    if (DUAL) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int i = 0; i < NB; ++i) acc[m][i] += acc2[m][i];
    }
}

// ---- software-pipelined fp32 sweep (round 6) -----------------------------------------------------------------------
// The sweep above compiles to "issue the LDS reads of 4 k-steps / wait for them / 4 MFMAs" per iteration: nothing of iteration i + 1 is
// in flight while the MFMAs of iteration i run (profiles/r05_c2_wg_timeline.json: the matrix pipe 54 % busy inside the sweep).  Here the
// k-steps of a chunk are walked in GROUPS of 4 channel quads of one tap, with an explicit register double buffer: the A / B fragments of
// group g + 1 are requested before the MFMAs of group g issue, so that every s_waitcnt lgkmcnt(N) of the loop counts only reads that are
// one group (>= 4 MFMAs = 128 matrix-pipe cycles) old.  Per accumulator the k order is the one of sweep_chunk - tap-major, channel quads
// ascending, even / odd quads on the two partial sums of the 1 x 1 register tile - so the results are bit-identical.
// Needs ck4 % 4 == 0 (chunks of 16 k channels); other chunk sizes take sweep_chunk.
template <int MB, int NB>
struct KGroup {
    float a[4][MB];
    float b[4][NB];
};

template <int MB, int NB, int PL>
__device__ __forceinline__ void kgroup_load(KGroup<MB, NB>& f, const float* __restrict__ wq,
                                            const float* const (&pq)[NB], int plane4, int dbg) {
    (void)dbg;
#ifdef MR_CONV_TIMELINE                              // ablation (diagnostic library): bit 32 = no A reads, bit 64 = no B reads (operands = lane constants)
    if (dbg & 32) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int m = 0; m < MB; ++m) f.a[g][m] = __int_as_float(0x3f800000 + g + m + (int)threadIdx.x);
    } else
#endif
    {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int m = 0; m < MB; ++m) f.a[g][m] = wq[(g * MB + m) * 64];
    }
#ifdef MR_CONV_TIMELINE
    if (dbg & 64) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < NB; ++i) f.b[g][i] = __int_as_float(0x3f000000 + g + i + (int)threadIdx.x);
    } else
#endif
    {
        // PL > 0: the plane pitch is a compile-time constant - the four channel quads of the group are IMMEDIATE offsets of one address
        // register per pixel block (ds_read_b32 v, vq offset:g*16*PL); PL = 0: run-time pitch, one v_add per read
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int i = 0; i < NB; ++i) f.b[g][i] = pq[i][g * (PL ? 4 * PL : plane4)];
    }
}

template <int MB, int NB, bool DUAL>
__device__ __forceinline__ void kgroup_mma(f32x4 (&acc)[MB][NB], f32x4 (&acc2)[MB][NB], const KGroup<MB, NB>& f) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (DUAL && (g & 1)) acc2[m][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[g][m], f.b[g][i], acc2[m][i], 0, 0, 0);
                else acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[g][m], f.b[g][i], acc[m][i], 0, 0, 0);
            }
}

// cursor over the groups of a chunk in sweep order: tap (kh, kw) outer, group of 4 channel quads inner
struct GroupCursor {
    int q, kw;                // quad group inside the tap, tap column
    const float* wq;          // A fragments of the group (lane offset included)
};

// The fp32 MFMA executes on the SIMD's fp32 vector ALUs (that is why its peak IS the vector peak): every VALU instruction between two MFMAs
// takes its issue time out of the matrix stream - tools/probes/mfma_rates.hip: 32.0 cycles per v_mfma_f32_16x16x4_f32 back to back,
// 43-55 with one v_add_u32 + two SALU in between; profiles/r06_c2_sweep_ablation.json: the sweep of a ResNet layer1 workgroup takes
// 6.5 us with EVERY LDS read and DMA removed, 7.0 with them, 4.1 at 32 cycles per MFMA.  So the sweep must not do address arithmetic
// per operand read.  PL > 0 instantiates the sweep for ONE plane pitch: a read is `ds_read_b32 v, vq offset:imm`, and the only VALU work
// left is one v_add per pixel block and group (the cursor) + one for the A pointer.
template <int MB, int NB, int PL>
__device__ __forceinline__ void sweep_chunk_pipe_pl(const ConvKArgs& a, f32x4 (&acc)[MB][NB], const float* ldsI, const float* ldsW,
                                                    const int (&lbase)[NB], int ck4, int lane, int KH, int KW) {
    constexpr bool DUAL = MB * NB == 1;
    f32x4 acc2[MB][NB];
    if (DUAL) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int i = 0; i < NB; ++i) acc2[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int nq = ck4 >> 2;
    const int ngroups = KH * KW * nq;
    const int plane4 = PL ? 4 * PL : 4 * a.PLANE, plane16 = 4 * plane4;
    const int next_tap = 1 - nq * plane16 + plane16, next_row = a.IWa - KW;       // (scalar: the cursor's steps at a tap / tap-row boundary)
    const float* pq[NB];                               // B operand of the group's first quad, per pixel block (a byte address: one v_add per group)
#pragma unroll
    for (int i = 0; i < NB; ++i) pq[i] = ldsI + lbase[i];
    GroupCursor c = {0, 0, ldsW + lane};
    auto next = [&](KGroup<MB, NB>& f) {
        kgroup_load<MB, NB, PL>(f, c.wq, pq, plane4, a.dbg);
        c.wq += 4 * MB * 64;
        int step = plane16;                            // wave-uniform: SALU
        if (++c.q == nq) {                             // next tap: back to the first quad, one column on - or to the head of the next tap row
            c.q = 0;
            step = next_tap;
            if (++c.kw == KW) { c.kw = 0; step += next_row; }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) pq[i] += step;
    };
    KGroup<MB, NB> f0, f1;
    next(f0);
    int g = 0;
    for (; g + 2 < ngroups; g += 2) {                  // steady state: both loads of the body are in range, no branch around either
        next(f1);
        kgroup_mma<MB, NB, DUAL>(acc, acc2, f0);
        next(f0);
        kgroup_mma<MB, NB, DUAL>(acc, acc2, f1);
    }
    if (g + 1 < ngroups) {
        next(f1);
        kgroup_mma<MB, NB, DUAL>(acc, acc2, f0);
        kgroup_mma<MB, NB, DUAL>(acc, acc2, f1);
    } else {
        kgroup_mma<MB, NB, DUAL>(acc, acc2, f0);
    }
    if (DUAL) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int i = 0; i < NB; ++i) acc[m][i] += acc2[m][i];
    }
}

// plane pitches with their own instantiation of the sweep: the pitches of the c2 / c3 tables' direct-kernel schedules by multiply-adds
// (tools/isa_loops.py --planes lists them); any other pitch takes the run-time form (PL = 0)
#define MR_PLANE_MENU(X) X(240) X(176) X(208) X(400) X(612)
template <int MB, int NB>
__device__ __forceinline__ void sweep_chunk_pipe(const ConvKArgs& a, f32x4 (&acc)[MB][NB], const float* ldsI, const float* ldsW,
                                                 const int (&lbase)[NB], int ck4, int lane, int KH, int KW) {
    switch (a.PLANE) {
#define MR_PLANE_CASE(P) case P: sweep_chunk_pipe_pl<MB, NB, P>(a, acc, ldsI, ldsW, lbase, ck4, lane, KH, KW); break;
        MR_PLANE_MENU(MR_PLANE_CASE)
#undef MR_PLANE_CASE
        default: sweep_chunk_pipe_pl<MB, NB, 0>(a, acc, ldsI, ldsW, lbase, ck4, lane, KH, KW); break;
    }
}

// K split across the waves of a workgroup (a.kws): every wave sweeps the k-steps t = wave, wave + WV, ... of the chunk's
// taps x channel-quads (flattened, so that chunks with fewer quads than waves still spread) for the SAME NB pixel blocks; the
// partial accumulators meet in LDS after the K loop.  Small layers (a few hundred output pixels per image) get WV x more
// workgroups this way without the workspace round trip and the finishing launch of split_k.
template <int MB, int NB, int WV>
__device__ __forceinline__ void sweep_chunk_kws(const ConvKArgs& a, f32x4 (&acc)[MB][NB], f32x4 (&acc2)[MB][NB], const float* ldsI,
                                                const float* ldsW, const int (&lbase)[NB], int ck4, int lane, int wave, int KH, int KW) {
    const float* wl = ldsW + lane;
    const int nsteps = KH * KW * ck4;
    bool odd = false;
    for (int t = wave; t < nsteps; t += WV) {
        const int tap = t / ck4, c4 = t - tap * ck4;          // wave-uniform
        const int kh = tap / KW, kw = tap - kh * KW;
        const float* wt = wl + tap * ck4 * (MB * 64);
        if (odd) kstep<MB, NB>(acc2, wt, ldsI, lbase, c4, c4 * 4 * a.PLANE + kh * a.IWa + kw);
        else kstep<MB, NB>(acc, wt, ldsI, lbase, c4, c4 * 4 * a.PLANE + kh * a.IWa + kw);
        odd = !odd;
    }
}

// ---- bf16 MFMA sweep: one v_mfma_f32_16x16x16_bf16 per (cout block, pixel block) and 16 input channels of a tap -------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ s16x4 pack_bf16x4(float x0, float x1, float x2, float x3) {   // round to nearest even
    const bf16x2 lo = __builtin_convertvector((f32x2){x0, x1}, bf16x2);
    const bf16x2 hi = __builtin_convertvector((f32x2){x2, x3}, bf16x2);
    union { bf16x2 h[2]; s16x4 v; } u;
    u.h[0] = lo; u.h[1] = hi;
    return u.v;
}

template <int MB, int NB>
__device__ __forceinline__ void kstep_bf16(f32x4 (&acc)[MB][NB], const float* __restrict__ wt, const float* __restrict__ ldsI,
                                           const int (&lbase)[NB], int c16, int off, int plane4) {
    s16x4 av[MB], bv[NB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const f32x2 raw = *(const f32x2*)(wt + (c16 * MB + m) * 128);       // wt already includes lane * 2
        union { f32x2 f; s16x4 v; } u;
        u.f = raw;
        av[m] = u.v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const float* p = ldsI + lbase[i] + off;                            // channel (lane >> 4) of this 16-channel step
        bv[i] = pack_bf16x4(p[0], p[plane4], p[2 * plane4], p[3 * plane4]);   // channels g, 4+g, 8+g, 12+g
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < NB; ++i) acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(av[m], bv[i], acc[m][i], 0, 0, 0);
}

// Two 16-channel steps of a tap in ONE v_mfma_f32_16x16x32_bf16 (gfx950: K = 32, the same issue time as the K = 16 form): lane
// (cout l&15, group g = l>>4) holds 8 bf16 - elements 0..3 = channels 4j+g of step c16, 4..7 = the same of step c16 + 1; the
// packed weight stream and the LDS planes are those of the K = 16 path, read as two halves.
typedef short s16x8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MB, int NB>
__device__ __forceinline__ void kstep_bf16_k32(f32x4 (&acc)[MB][NB], const float* __restrict__ wt, const float* __restrict__ ldsI,
                                               const int (&lbase)[NB], int c16, int off, int plane4) {
    s16x8v av[MB], bv[NB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        union { f32x2 f; s16x4 v; } lo, hi;
        lo.f = *(const f32x2*)(wt + (c16 * MB + m) * 128);                  // wt already includes lane * 2
        hi.f = *(const f32x2*)(wt + ((c16 + 1) * MB + m) * 128);
        av[m] = (s16x8v){lo.v[0], lo.v[1], lo.v[2], lo.v[3], hi.v[0], hi.v[1], hi.v[2], hi.v[3]};
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const float* p = ldsI + lbase[i] + off;                            // channel (lane >> 4) of step c16; step c16 + 1 is 16 planes on
        const s16x4 lo = pack_bf16x4(p[0], p[plane4], p[2 * plane4], p[3 * plane4]);
        const s16x4 hi = pack_bf16x4(p[4 * plane4], p[5 * plane4], p[6 * plane4], p[7 * plane4]);
        bv[i] = (s16x8v){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < NB; ++i)
            acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[m]), __builtin_bit_cast(bf16x8, bv[i]), acc[m][i], 0, 0, 0);
}

template <int MB, int NB>
__device__ __forceinline__ void sweep_chunk_bf16(const ConvKArgs& a, f32x4 (&acc)[MB][NB], const float* ldsI, const float* ldsW,
                                                 const int (&lbase)[NB], int ck16, int lane, int KH, int KW) {
    const float* wl = ldsW + lane * 2;
    const int plane4 = 4 * a.PLANE;
    for (int kh = 0; kh < KH; ++kh) {
        for (int kw = 0; kw < KW; ++kw) {
            const int tapoff = kh * a.IWa + kw;
            const float* wt = wl + (kh * KW + kw) * ck16 * (MB * 128);
            int c16 = 0;
            for (; c16 + 2 <= ck16; c16 += 2)       // pairs of 16-channel steps on the K = 32 instruction
                kstep_bf16_k32<MB, NB>(acc, wt, ldsI, lbase, c16, c16 * 16 * a.PLANE + tapoff, plane4);
            for (; c16 < ck16; ++c16) kstep_bf16<MB, NB>(acc, wt, ldsI, lbase, c16, c16 * 16 * a.PLANE + tapoff, plane4);
        }
    }
}

// ---- bf16x3 sweep (MR_COMPUTE_BF16X3): fp32-class accuracy on the bf16 matrix cores ------------------------------------------
// x = hi + lo with hi = bf16(x), lo = bf16(x - hi) carries 16 mantissa bits; a*b ~ a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (the dropped
// lo*lo term and the two roundings are ~2^-16 relative), products exact, fp32 accumulate.  The fp32 MFMA of gfx950 runs at 1/16 of
// the bf16 rate, so three bf16 MFMAs per 16 channels replace four fp32 MFMAs (16x16x4) at 3/16 of their matrix-core time.
// Weights arrive pre-split (mr_conv_pack_weights_bf16x3: per lane 4 hi + 4 lo bf16 = 16 bytes, one ds_read_b128), activations
// stay fp32 in HBM / LDS and are split while the B fragment is formed.
typedef short s16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_bits_to_f32(short h) {
    return __builtin_bit_cast(float, (unsigned)(unsigned short)h << 16);
}

template <int MB, int NB>
__device__ __forceinline__ void kstep_bf16x3(f32x4 (&acc)[MB][NB], const float* __restrict__ wt, const float* __restrict__ ldsI,
                                             const int (&lbase)[NB], int c16, int off, int plane4) {
    s16x4 ah[MB], al[MB], bh[NB], bl[NB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        union { f32x4 f; s16x8 v; } u;
        u.f = *(const f32x4*)(wt + (c16 * MB + m) * 256);                   // wt already includes lane * 4
        ah[m] = (s16x4){u.v[0], u.v[1], u.v[2], u.v[3]};
        al[m] = (s16x4){u.v[4], u.v[5], u.v[6], u.v[7]};
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const float* p = ldsI + lbase[i] + off;
        const float x0 = p[0], x1 = p[plane4], x2 = p[2 * plane4], x3 = p[3 * plane4];
        bh[i] = pack_bf16x4(x0, x1, x2, x3);
        bl[i] = pack_bf16x4(x0 - bf16_bits_to_f32(bh[i][0]), x1 - bf16_bits_to_f32(bh[i][1]),
                            x2 - bf16_bits_to_f32(bh[i][2]), x3 - bf16_bits_to_f32(bh[i][3]));
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < NB; ++i) {                                     // small terms first
            acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al[m], bh[i], acc[m][i], 0, 0, 0);
            acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[m], bl[i], acc[m][i], 0, 0, 0);
            acc[m][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah[m], bh[i], acc[m][i], 0, 0, 0);
        }
}

template <int MB, int NB>
__device__ __forceinline__ void sweep_chunk_bf16x3(const ConvKArgs& a, f32x4 (&acc)[MB][NB], const float* ldsI, const float* ldsW,
                                                   const int (&lbase)[NB], int ck16, int lane, int KH, int KW) {
    const float* wl = ldsW + lane * 4;
    const int plane4 = 4 * a.PLANE;
    for (int kh = 0; kh < KH; ++kh) {
        for (int kw = 0; kw < KW; ++kw) {
            const int tapoff = kh * a.IWa + kw;
            const float* wt = wl + (kh * KW + kw) * ck16 * (MB * 256);
            for (int c16 = 0; c16 < ck16; ++c16) kstep_bf16x3<MB, NB>(acc, wt, ldsI, lbase, c16, c16 * 16 * a.PLANE + tapoff, plane4);
        }
    }
}

// DMA_IN: input tile staged by LDS-DMA (direct / upsample reads).  false: register-staged variant for the 2x2
// max-pool and input-normalisation reads (kept out of the DMA kernel: the compiler-visible loads of that path
// make hipcc drain vmcnt before every sweep and spill SGPRs).
template <int MB, int NB, bool DMA_IN, int WV, int BF16, bool KWS = false>
__global__ __launch_bounds__(WV * 64) void conv_mfma_kernel(const ConvKArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    dbg_stamp(a, 0);
    // two pipeline buffers, each [CK][PLANE] input tile + [taps][ck4][MB][64] A fragments
    const int ioff = a.CK * a.PLANE;
    const int bufsz = ioff + a.wmax_floats;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The ~480-byte argument block spans eight 64-byte lines that the compiler fetches piecemeal, each first touch a
    // miss all the way to memory (the L2 is cold at kernel start): touch every line once, up front, together.
    {
        const int* ka = (const int*)__builtin_amdgcn_kernarg_segment_ptr();
        int t = 0;
#pragma unroll
        for (int i = 0; i < (int)(sizeof(ConvKArgs) / 64); ++i) t |= __builtin_nontemporal_load(ka + 16 * i + 15);
        if (t == 0x7fffffff && a.dbg == 0x7fffffff) return;           // never true; keeps the loads alive
    }
    // grid (tile, cout group, z): consecutive workgroup ids (tiles) go round-robin to the 8 XCDs, so the cout groups
    // of one tile (ids a multiple of the tile count apart) mostly meet on one XCD and share the input tile in its L2.
    // (Renumbering XCD-major so that weight blocks are shared instead was measured: no change - the fills are
    // latency bound, ~1.8 us per chunk from a cold L2, not fabric-bandwidth bound.)
    const int tile = blockIdx.x;
    const int grp = blockIdx.y;
    const int ty = a.tiles_x == 1 ? tile : (int)(((unsigned long long)(unsigned)tile * a.tiles_x_magic) >> 32);   // tile / tiles_x
    const int tx = tile - ty * a.tiles_x;
    const int cb0 = grp * MB;
    int z = blockIdx.z;                                                // ((b * ksplit) + ks) * nphase + ph
    const int ph = a.nphase == 4 ? (z & 3) : 0;
    z = a.nphase == 4 ? z >> 2 : z;
    const int ks = a.ks_shift >= 0 ? (z & (a.ksplit - 1)) : z % a.ksplit;
    const int b = a.ks_shift >= 0 ? (z >> a.ks_shift) : z / a.ksplit;
    const int oy0 = ty * a.TH, ox0 = tx * a.TWB * 16;
    const int iy_base = oy0 * a.SH - a.PT[ph], ix_base = ox0 * a.SW - a.PL[ph];
    const int HsWs = a.Hs * a.Ws;

    // ---- fixed staging positions of this thread: p = tid + 256*j  ->  (iy, ix) of the haloed tile ----
    int goff[MR_MAX_PPT], loff[MR_MAX_PPT];
    const int P = a.IH * a.IWa;
    const int xsh = a.xsh[ph];
#pragma unroll
    for (int j = 0; j < MR_MAX_PPT; ++j) { goff[j] = -1; loff[j] = -1; }
    if (!a.dma_x4) {                                  // dword-DMA / register staging only
#pragma unroll
        for (int j = 0; j < MR_MAX_PPT; ++j) {
            const int p = tid + 256 * j;
            const int iy = p / a.IWa, ix = p - iy * a.IWa;
            const int gy = iy_base + iy, gx = ix_base - xsh + ix;
            loff[j] = p < P ? p : -1;
            bool inb = p < P && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
            int g = 0;
            if (a.in_mode == MR_IN_DIRECT) g = gy * a.Ws + gx;
            else if (a.in_mode == MR_IN_UPSAMPLE2) g = (gy >> 1) * a.Ws + (gx >> 1);
            else g = (2 * gy) * a.Ws + 2 * gx;
            goff[j] = inb ? g : -1;
        }
    }

    // dwordx4 DMA path: lane l owns the 16-byte groups r = l + 64*i of every channel plane
    int voff4[MR_MAX_G4];
    {
        const int gpr = a.IWa >> 2;                       // groups per tile row
#pragma unroll
        for (int i = 0; i < MR_MAX_G4; ++i) {
            const int r = lane + 64 * i;
            const int iy = (int)(((unsigned)r * a.gpr_magic) >> 16), ix4 = r - iy * gpr;   // r / gpr, r < 256
            const int gy = iy_base + iy, gx = ix_base - xsh + 4 * ix4;
            const bool inb = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
            voff4[i] = iy < a.IH ? (inb ? (gy * a.Ws + gx) * 4 : -1) : -2;
        }
    }

    int prow[NB], pcol[NB], lbase[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int pb = KWS ? i : wave * NB + i;           // KWS: every wave works on the same NB blocks
        prow[i] = a.TWB == 2 ? pb >> 1 : pb;
        pcol[i] = (a.TWB == 2 ? (pb & 1) : 0) * 16 + (lane & 15);
        lbase[i] = (lane >> 4) * a.PLANE + prow[i] * a.SH * a.IWa + pcol[i] * a.SW + xsh;
    }

    f32x4 acc[MB][NB];
    f32x4 acck[MB][NB];                                // KWS: second partial sum (alternating k-steps keep two MFMA chains in flight)
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int i = 0; i < NB; ++i) { acc[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; acck[m][i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int q_lo = a.ks_shift >= 0 ? (ks * a.nchunks) >> a.ks_shift : (ks * a.nchunks) / a.ksplit;
    const int q_hi = a.ks_shift >= 0 ? ((ks + 1) * a.nchunks) >> a.ks_shift : ((ks + 1) * a.nchunks) / a.ksplit;
    const int KH = a.KHp[ph], KW = a.KWp[ph];
    const int T = KH * KW;
    const float* wgrp = a.w[ph] + (long long)grp * a.wgroup_stride[ph];

    // ---- software pipeline over K chunks: chunk q+1 streams into the other buffer while q is swept ----------
    ChunkCursor cur = {0, 0, 0};
    for (int q = 0; q < q_lo; ++q) cursor_advance<MB>(a, cur, T);
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds;
    dbg_stamp(a, 11);                                  // setup done, first DMA goes out
    issue_chunk<MB, DMA_IN>(a, cur, lds, lds + ioff, lds_base, lds_base + ioff * 4, wgrp, b, T, lane, wave, WV, HsWs, goff, loff, voff4);
    dma_wait_all();
    __syncthreads();
    dbg_stamp(a, 1);
    int pb = 0;
    for (int q = q_lo; q < q_hi; ++q) {
        const int ck4 = min(a.CK, a.src_cpad[cur.s] - cur.c0) >> 2;
        float* bcur = lds + pb * bufsz;
        float* bnxt = lds + (pb ^ 1) * bufsz;
        cursor_advance<MB>(a, cur, T);
        const bool stamp = q == q_lo + 1 || (q == q_lo && q_hi == q_lo + 1);
        if (stamp) { dbg_stamp(a, 4); dbg_stamp(a, 9); }
        if (q + 1 < q_hi) {
            const unsigned nb_addr = lds_base + (pb ^ 1) * bufsz * 4;
            issue_chunk<MB, DMA_IN>(a, cur, bnxt, bnxt + ioff, nb_addr, nb_addr + ioff * 4, wgrp, b, T, lane, wave, WV, HsWs, goff, loff, voff4);
        }
        if (stamp) dbg_stamp(a, 5);
        if (!MR_DBG(1)) {
            if (KWS) sweep_chunk_kws<MB, NB, WV>(a, acc, acck, bcur, bcur + ioff, lbase, ck4, lane, wave, KH, KW);
            else if (BF16 == 2) sweep_chunk_bf16x3<MB, NB>(a, acc, bcur, bcur + ioff, lbase, ck4 >> 2, lane, KH, KW);
            else if (BF16 == 1) sweep_chunk_bf16<MB, NB>(a, acc, bcur, bcur + ioff, lbase, ck4 >> 2, lane, KH, KW);
            else if ((ck4 & 3) == 0) sweep_chunk_pipe<MB, NB>(a, acc, bcur, bcur + ioff, lbase, ck4, lane, KH, KW);
            else sweep_chunk<MB, NB>(a, acc, bcur, bcur + ioff, lbase, ck4, lane, KH, KW);
        }
        if (stamp) dbg_stamp(a, 6);
        dma_wait_all();                                // this wave's share of the next chunk has landed
        if (stamp) dbg_stamp(a, 7);
        __syncthreads();                               // ... everyone's has, and everyone is done with this buffer
        if (stamp) { dbg_stamp(a, 8); dbg_stamp(a, 10); }
        pb ^= 1;
    }
    // ---- KWS: the WV partial accumulators of every (cout block, pixel block) meet in LDS (the pipeline buffers are free behind
    //      one more barrier); block j = m * NB + i is summed - in wave order, deterministic - and finished by wave j % WV
    if (KWS) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                acc[m][i] += acck[m][i];
#pragma unroll
                for (int r = 0; r < 4; ++r) lds[((wave * (MB * NB) + m * NB + i) * 4 + r) * 64 + lane] = acc[m][i][r];
            }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (((m * NB + i) % WV) != wave) continue;
                f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
                for (int w = 0; w < WV; ++w)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sum[r] += lds[((w * (MB * NB) + m * NB + i) * 4 + r) * 64 + lane];
                acc[m][i] = sum;
            }
    }
    // ---- epilogue: D fragment lane l holds pixel (l&15), couts (l>>4)*4 + r ---------------------
    const int CB16 = a.CB * 16;
    dbg_stamp(a, 2);
    if (MR_DBG(16)) {                                  // stamp 3 = after this thread's stores were accepted
        if (a.ksplit > 1) return;                      // (the workspace is the stamp buffer here)
    }
    if (MR_DBG(8)) { if (acc[0][0][0] != 123.456f) return; }
    const int lq4 = (lane >> 4) * 4;
    if (a.ksplit > 1) {                                // raw partial sums; splitk_epilogue_kernel finishes them
        const long long plane = (long long)a.Ho * a.Wo;
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int cout0 = (cb0 + m) * 16 + lq4;
            if (cout0 >= CB16) continue;
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const int oy = oy0 + prow[i], ox = ox0 + pcol[i];
                if (oy >= a.Ho || ox >= a.Wo) continue;
                float* w = a.ws + ((((long long)ks * a.nphase + ph) * a.batch + b) * CB16 + cout0) * plane + (long long)oy * a.Wo + ox;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r * plane] = acc[m][i][r];
            }
        }
        return;
    }
    {
        // bias / residual operands are fetched in batches ahead of the stores that need them: a load issued
        // between two stores would wait out a full memory round trip per output element.
        // The activation of the common layers (none / ReLU / LeakyReLU with a slope in [0, 1]: derive() rejects others) is ONE branch-free form, max(x, lo) with lo = x /
        // 0 / x * p0 (ReLU as max(x, 0): -inf -> 0 and no -0.0, like torch.relu - ADVICE r4; `x > 0 ? x : x * 0` gave NaN / -0.0 there):
        // as a switch per element it compiled into two scalar branches and an inlined tanh / exp body per stored value (r04: 180
        // instructions per store in the <1,1> kernel, 59 branches for its 4 stores); the sigmoid / |tanh| layers keep the general form
        // behind ONE uniform branch around the whole store loop.
        const long long chs = (long long)a.dst_H * a.dst_W;
        float bias[MB][4];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cout = (cb0 + m) * 16 + lq4 + r;
                bias[m][r] = (a.bias && cout < a.Cout) ? a.bias[cout] : 0.f;
            }
        const long long bbase = (long long)b * a.dst_bstride;
        const unsigned keep = a.act == MR_ACT_RELU ? 0u : ~0u;   // lo = (x * slope) AND keep: +0 for ReLU (an AND, not a select: hipcc clones the store loops around a uniform select)
        const float slope = a.act == MR_ACT_LEAKY_RELU ? a.p0 : 1.f;
        auto store_all = [&](auto simple_tag) {
            constexpr bool SIMPLE = decltype(simple_tag)::value;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                const int cout0 = (cb0 + m) * 16 + lq4;
                long long idx0[NB];
                float rv[NB][4];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int oy = oy0 + prow[i], ox = ox0 + pcol[i];
                    const bool ok = oy < a.Ho && ox < a.Wo && (!KWS || ((m * NB + i) % WV) == wave);
                    idx0[i] = ok ? bbase + ((long long)(a.ch_off + cout0) * a.dst_H + (oy * a.ostep_h + a.ooff_h[ph])) * a.dst_W +
                                       (ox * a.ostep_w + a.ooff_w[ph])
                                 : -1;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        rv[i][r] = (a.res && ok && cout0 + r < a.Cout) ? a.res[idx0[i] + r * chs] : 0.f;
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    if (idx0[i] < 0) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (cout0 + r < a.Cout) {
                            float v = acc[m][i][r] + bias[m][r];              // (no bias: + 0)
                            v += rv[i][r];                                    // (no residual: + 0)
                            a.dst[idx0[i] + r * chs] = SIMPLE ? fmaxf(v, __uint_as_float(__float_as_uint(v * slope) & keep)) : mr_activate(v, a.act, a.p0, a.p1);
                        }
                }
            }
        };
        if (a.act <= MR_ACT_LEAKY_RELU) store_all(std::true_type{});
        else store_all(std::false_type{});
    }
    if (MR_DBG(16)) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); dbg_stamp(a, 3); }
}

__device__ __forceinline__ void mr_store_out(const ConvKArgs& a, int ph, int b, int cout, int oy, int ox, float v) {
    if (a.bias) v += a.bias[cout];
    const long long idx = (long long)b * a.dst_bstride +
                          ((long long)(a.ch_off + cout) * a.dst_H + (oy * a.ostep_h + a.ooff_h[ph])) * a.dst_W +
                          (ox * a.ostep_w + a.ooff_w[ph]);
    if (a.res) v += a.res[idx];
    a.dst[idx] = mr_activate(v, a.act, a.p0, a.p1);
}

__global__ __launch_bounds__(256) void splitk_epilogue_kernel(const ConvKArgs a) {
    const long long plane = (long long)a.Ho * a.Wo;
    const long long total = (long long)a.nphase * a.batch * a.Cout * plane;
    const int CB16 = a.CB * 16;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % a.Wo);
        const int oy = (int)((i / a.Wo) % a.Ho);
        const int cout = (int)((i / plane) % a.Cout);
        const int b = (int)((i / (plane * a.Cout)) % a.batch);
        const int ph = (int)(i / (plane * a.Cout * a.batch));
        float v = 0.f;
        for (int ks = 0; ks < a.ksplit; ++ks)
            v += a.ws[(((((long long)ks * a.nphase + ph) * a.batch + b) * CB16 + cout) * a.Ho + oy) * a.Wo + ox];
        mr_store_out(a, ph, b, cout, oy, ox, v);
    }
}

// Fast finishing kernel for the common case (one phase, dense destination planes, plane size % 4 == 0): one thread per 4 consecutive
// pixels of one (sample, cout) plane; all ksplit 16-byte partial loads (and the residual) are issued before the first add - the generic
// kernel above walks the slices with dependent scalar loads and 64-bit div / mod per element (5.6 us per launch at c2, 15 launches per
// keyframe).  Same summation order (slice 0, 1, ...), same epilogue arithmetic: bit-identical results.
template <int KS>
__global__ __launch_bounds__(256) void splitk_epilogue4_kernel(const ConvKArgs a) {
    const int plane = a.Ho * a.Wo, q_per_plane = plane >> 2;
    const int total = a.batch * a.Cout * q_per_plane;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int pix = (i % q_per_plane) << 2;
    const int t = i / q_per_plane;
    const int cout = t % a.Cout, b = t / a.Cout;
    const int CB16 = a.CB * 16;
    const long long slab = (long long)a.batch * CB16 * plane;                    // floats per k slice
    const float* w0 = a.ws + ((long long)b * CB16 + cout) * plane + pix;
    f32x4 part[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) part[ks] = *(const f32x4*)(w0 + ks * slab);
    const long long idx = (long long)b * a.dst_bstride + (long long)(a.ch_off + cout) * plane + pix;
    f32x4 rv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.res) rv = *(const f32x4*)(a.res + idx);
    const float bias = a.bias ? a.bias[cout] : 0.f;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) v += part[ks];
    f32x4 o;
    const unsigned keep = a.act == MR_ACT_RELU ? 0u : ~0u;   // lo = (x * slope) AND keep: +0 for ReLU (an AND, not a select: hipcc clones the store loops around a uniform select)
    const float slope = a.act == MR_ACT_LEAKY_RELU ? a.p0 : 1.f;
    if (a.act <= MR_ACT_LEAKY_RELU) {                 // branch-free form of none / ReLU / LeakyReLU (see conv_mfma_kernel's epilogue)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = (v[r] + bias) + rv[r];
            o[r] = fmaxf(x, __uint_as_float(__float_as_uint(x * slope) & keep));
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = mr_activate((v[r] + bias) + rv[r], a.act, a.p0, a.p1);
    }
    *(f32x4*)(a.dst + idx) = o;
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
namespace {

struct Derived {
    ConvKArgs k;
    size_t lds_bytes;
    int mb, nb, wv;
    int mode;                  // MR_COMPUTE_*: 0 fp32 MFMA, 1 bf16, 2 bf16x3 split
    dim3 grid;
};

bool valid_mb(int mb) { return mb == 1 || mb == 2 || mb == 3 || mb == 4 || mb == 6; }
bool valid_ck(int ck) { return ck == 8 || ck == 16 || ck == 32 || ck == 64 || ck == 128; }

int derive(const mr_conv_desc* d, Derived* out) {
    if (!d || d->num_src < 1 || d->num_src > MR_MAX_SOURCES) return MR_ERR_BAD_ARGUMENT;
    if (d->batch < 1 || d->kh < 1 || d->kw < 1 || d->stride_h < 1 || d->stride_w < 1) return MR_ERR_BAD_ARGUMENT;
    if (d->stride_w > 2) return MR_ERR_UNSUPPORTED;
    if (d->out_h < 1 || d->out_w < 1 || d->out_channels < 1 || !d->dst) return MR_ERR_BAD_ARGUMENT;
    const int mb = d->cout_blocks_per_wg, nb = d->pixel_blocks_per_wave;
    const int wv = d->waves_per_wg == 0 ? 4 : d->waves_per_wg;
    if (wv != 4 && wv != 8) return MR_ERR_BAD_ARGUMENT;
    const int kws = d->k_split_waves ? 1 : 0;
    if (kws && (d->split_k != 1 || d->compute_dtype != MR_COMPUTE_F32)) return MR_ERR_UNSUPPORTED;
    if (!valid_mb(mb) || !(nb == 1 || nb == 2 || nb == 4) || !valid_ck(d->chunk_channels)) return MR_ERR_BAD_ARGUMENT;
    if (d->split_k < 1) return MR_ERR_BAD_ARGUMENT;
    if (d->activation == MR_ACT_LEAKY_RELU && !(d->act_p0 >= 0.f && d->act_p0 <= 1.f)) return MR_ERR_UNSUPPORTED;   // the epilogue is max(x, x * slope)
    const int nphase = d->num_phases <= 1 ? 1 : d->num_phases;
    if (nphase != 1 && nphase != 4) return MR_ERR_BAD_ARGUMENT;
    ConvKArgs& k = out->k;
    memset(&k, 0, sizeof(k));
    k.nsrc = d->num_src;
    k.Hs = d->src_h;
    k.Ws = d->src_w;
    switch (d->in_mode) {
        case MR_IN_DIRECT: k.Hin = k.Hs; k.Win = k.Ws; break;
        case MR_IN_UPSAMPLE2: k.Hin = 2 * k.Hs; k.Win = 2 * k.Ws; break;
        case MR_IN_MAXPOOL2: k.Hin = k.Hs / 2; k.Win = k.Ws / 2; break;
        default: return MR_ERR_BAD_ARGUMENT;
    }
    k.in_mode = d->in_mode;
    k.in_tf = d->in_transform;
    k.CK = d->chunk_channels;
    const int bf16 = d->compute_dtype == MR_COMPUTE_BF16 ? 1 : d->compute_dtype == MR_COMPUTE_BF16X3 ? 2 : 0;
    if (d->compute_dtype != MR_COMPUTE_F32 && !bf16) return MR_ERR_BAD_ARGUMENT;
    if (bf16 && k.CK < 16) return MR_ERR_BAD_ARGUMENT;
    k.bf16 = bf16 == 1 ? 1 : 0;       // device side: size of a weight block only (bf16x3 blocks are as large as fp32 ones)
    out->mode = bf16;
    int nchunks = 0, cpad_total = 0;
    for (int s = 0; s < d->num_src; ++s) {
        if (!d->src[s] || d->src_channels[s] < 1) return MR_ERR_BAD_ARGUMENT;
        k.src[s] = d->src[s];
        k.src_c[s] = d->src_channels[s];
        k.src_cpad[s] = mr_pad_channels(d->src_channels[s], bf16);
        const long long sbytes = (long long)d->batch * d->src_channels[s] * d->src_h * d->src_w * 4;
        if (sbytes >= (1ll << 31)) return MR_ERR_UNSUPPORTED;   // 32-bit byte offsets in the SRD path
        k.src_bytes[s] = (int)sbytes;
        nchunks += mr_ceil_div(k.src_cpad[s], k.CK);
        cpad_total += k.src_cpad[s];
    }
    if (d->split_k > nchunks) return MR_ERR_BAD_ARGUMENT;
    k.KH = d->kh; k.KW = d->kw; k.SH = d->stride_h; k.SW = d->stride_w;
    k.Ho = d->out_h; k.Wo = d->out_w;
    k.dst = d->dst;
    k.dst_H = d->dst_plane_h; k.dst_W = d->dst_plane_w;
    k.dst_bstride = (long long)d->dst_total_channels * d->dst_plane_h * d->dst_plane_w;
    k.ch_off = d->dst_channel_offset;
    k.ostep_h = d->out_step_h; k.ostep_w = d->out_step_w;
    if (k.ostep_h < 1 || k.ostep_w < 1) return MR_ERR_BAD_ARGUMENT;
    k.nphase = nphase;
    for (int p = 0; p < nphase; ++p) {
        if (nphase == 1) {
            k.w[0] = d->packed_weights; k.PT[0] = d->pad_top; k.PL[0] = d->pad_left;
            k.ooff_h[0] = d->out_off_h; k.ooff_w[0] = d->out_off_w;
        } else {
            k.w[p] = d->phase_weights[p]; k.PT[p] = d->phase_pad_top[p]; k.PL[p] = d->phase_pad_left[p];
            k.ooff_h[p] = d->phase_out_off_h[p]; k.ooff_w[p] = d->phase_out_off_w[p];
        }
        k.KHp[p] = (nphase == 4 && d->phase_kh[p] > 0) ? d->phase_kh[p] : d->kh;
        k.KWp[p] = (nphase == 4 && d->phase_kw[p] > 0) ? d->phase_kw[p] : d->kw;
        if (k.KHp[p] > d->kh || k.KWp[p] > d->kw) return MR_ERR_BAD_ARGUMENT;      // kh / kw size the input tile: the maximum
        if (!k.w[p] || k.ooff_h[p] < 0 || k.ooff_w[p] < 0) return MR_ERR_BAD_ARGUMENT;
        if ((k.Ho - 1) * k.ostep_h + k.ooff_h[p] >= k.dst_H || (k.Wo - 1) * k.ostep_w + k.ooff_w[p] >= k.dst_W)
            return MR_ERR_BAD_ARGUMENT;
    }
    if (d->dst_channel_offset < 0 || d->dst_channel_offset + d->out_channels > d->dst_total_channels) return MR_ERR_BAD_ARGUMENT;
    k.Cout = d->out_channels;
    k.CB = mr_ceil_div(d->out_channels, 16);
    k.bias = d->bias; k.res = d->residual;
    k.act = d->activation; k.p0 = d->act_p0; k.p1 = d->act_p1;
    k.kws = kws;
    const int blocks_per_wg = kws ? nb : wv * nb;      // pixel blocks of 16 a workgroup owns
    k.TWB = (d->out_w >= 32 && blocks_per_wg >= 2) ? 2 : 1;
    k.TH = blocks_per_wg / k.TWB;
    k.tiles_x = mr_ceil_div(d->out_w, k.TWB * 16);
    const int tiles_y = mr_ceil_div(d->out_h, k.TH);
    k.IH = (k.TH - 1) * k.SH + k.KH;
    k.IW = (k.TWB * 16 - 1) * k.SW + k.KW;
    k.dma_in = (d->in_mode != MR_IN_MAXPOOL2 && d->in_transform == MR_TF_NONE) ? 1 : 0;
    // dwordx4 DMA needs 16-byte groups that are entirely inside or outside the image: tile origin rounded
    // down to a multiple of 4 columns (tile x0 * stride is a multiple of 16), source width % 4 == 0
    k.dma_x4 = (k.dma_in && d->in_mode == MR_IN_DIRECT && (k.Ws & 3) == 0) ? 1 : 0;
    int xsh_max = 0;
    for (int p = 0; p < nphase; ++p) {
        k.xsh[p] = k.dma_x4 ? ((4 - (k.PL[p] & 3)) & 3) : 0;
        if (k.xsh[p] > xsh_max) xsh_max = k.xsh[p];
    }
    k.IWa = k.dma_x4 ? ((xsh_max + k.IW + 3) & ~3) : k.IW;
    int plane = k.IH * k.IWa;
    if (k.SW == 1) { while ((plane & 31) != 16) ++plane; }
    else if (k.dma_x4) { plane = (plane + 3) & ~3; }          // 16-byte rows win over the odd-stride bank trick
    else { plane |= 1; }
    k.PLANE = plane;
    k.ppt = mr_ceil_div(k.IH * k.IWa, 256);
    k.g4pt = mr_ceil_div(k.IH * (k.IWa >> 2), 64);
    if (!(k.dma_x4 && k.g4pt <= MR_MAX_G4) && k.ppt > MR_MAX_PPT) return MR_ERR_UNSUPPORTED;   // per-position staging only
    if (k.dma_x4 && k.g4pt > MR_MAX_G4) { k.dma_x4 = 0; k.IWa = k.IW; for (int p = 0; p < 4; ++p) k.xsh[p] = 0;
        plane = k.IH * k.IW; if (k.SW == 1) { while ((plane & 31) != 16) ++plane; } else { plane |= 1; }
        k.PLANE = plane; k.ppt = mr_ceil_div(k.IH * k.IW, 256); if (k.ppt > MR_MAX_PPT) return MR_ERR_UNSUPPORTED; }
    k.ksplit = d->split_k; k.nchunks = nchunks; k.batch = d->batch; k.ws = d->workspace;
    const int taps = k.KH * k.KW;
    for (int p = 0; p < nphase; ++p) k.wgroup_stride[p] = (long long)k.KHp[p] * k.KWp[p] * cpad_total * mb * (bf16 == 1 ? 8 : 16);
    int ck_max = 0;                                              // largest chunk of any source
    for (int s = 0; s < d->num_src; ++s) {
        const int c = k.src_cpad[s] < k.CK ? k.src_cpad[s] : k.CK;
        if (c > ck_max) ck_max = c;
    }
    k.wmax_floats = taps * ck_max * mb * (bf16 == 1 ? 8 : 16);
    // two pipeline buffers - one when no workgroup ever streams a second chunk.  (Round 6 measured a ring of up to 8 buffers with partial
    // vmcnt waits here: filling more than one buffer ahead delays the FIRST chunk - the launch's fills compete for the same fabric - and the
    // ring's bookkeeping costs the one-chunk layers 1 %: the two-buffer loop is 4 % faster over the direct launches of a c2 keyframe, also on
    // the layers whose table entries had chosen 3 buffers; tools/sessions/r06_s4.sh, r06_s19.sh.)
    const int nbuf = mr_ceil_div(nchunks, d->split_k) > 1 ? 2 : 1;
    out->lds_bytes = nbuf * ((size_t)k.CK * plane + (size_t)k.wmax_floats) * sizeof(float);
    if (kws && out->lds_bytes < (size_t)wv * mb * nb * 1024) out->lds_bytes = (size_t)wv * mb * nb * 1024;   // reduction scratch
    if (out->lds_bytes > 160 * 1024) return MR_ERR_LDS_BUDGET;
    k.gpr_magic = 65536u / (unsigned)(k.IWa >> 2 > 0 ? k.IWa >> 2 : 1) + 1u;
    k.tiles_x_magic = k.tiles_x == 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)k.tiles_x) + 1u;   // (1 would overflow)
    if ((long long)k.tiles_x * tiles_y >= 65536) return MR_ERR_UNSUPPORTED;
#ifdef MR_CONV_TIMELINE      // ablation / timestamp switches: diagnostic library only - the launch path of the product reads no environment
    { static const int dbg = [] { const char* e = getenv("MR_CONV_DBG"); return e ? atoi(e) : 0; }(); k.dbg = dbg; }
#endif
    out->mb = mb; out->nb = nb; out->wv = wv;
    if (wv == 8 && !k.dma_x4) return MR_ERR_UNSUPPORTED;
    if (kws && !k.dma_in) return MR_ERR_UNSUPPORTED;
    if (bf16 && !k.dma_in) return MR_ERR_UNSUPPORTED;                 // bf16 mode: LDS-DMA staged inputs only
    k.tiles_y = tiles_y;
    k.ngroups = mr_ceil_div(k.CB, mb);
    k.ks_shift = -1;
    for (int sft = 0; sft < 16; ++sft) if ((1 << sft) == d->split_k) k.ks_shift = sft;
    if ((long long)d->batch * d->split_k * nphase >= 65536) return MR_ERR_UNSUPPORTED;
    out->grid = dim3((unsigned)(k.tiles_x * tiles_y), (unsigned)k.ngroups, (unsigned)(d->batch * d->split_k * nphase));
    return 0;
}

template <int MB, int NB, bool DMA_IN, int WV, int BF16, bool KWS = false>
int launch(const Derived& dv, hipStream_t stream) {
    // raise the dynamic-LDS ceiling once per instantiation AND device (the attribute lives in the device's code object:
    // a process that drives several GPUs - nn.DataParallel replicas - must set it on each)
    static std::atomic<unsigned long long> attr_set{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(attr_set.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<MB, NB, DMA_IN, WV, BF16, KWS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, DMA_IN, WV, BF16, KWS>), dv.grid, dim3(WV * 64), dv.lds_bytes, stream, dv.k);
    return (int)hipGetLastError();
}

template <int MB, int NB>
int launch_variant(const Derived& dv, hipStream_t stream) {
    if (dv.mode == 2) {                                // LDS-DMA staging only (derive() checked)
        if (dv.wv == 8) return launch<MB, NB, true, 8, 2>(dv, stream);
        return launch<MB, NB, true, 4, 2>(dv, stream);
    }
    if (dv.mode == 1) {
        if (dv.wv == 8) return launch<MB, NB, true, 8, 1>(dv, stream);
        return launch<MB, NB, true, 4, 1>(dv, stream);
    }
    if (dv.k.kws) {                                    // K split across the waves: LDS-DMA staged fp32 launches only (derive() checked)
        if (dv.wv == 8) return launch<MB, NB, true, 8, 0, true>(dv, stream);
        return launch<MB, NB, true, 4, 0, true>(dv, stream);
    }
    if (dv.wv == 8) return launch<MB, NB, true, 8, 0>(dv, stream);         // dwordx4 DMA path only (derive() checked)
    if (dv.k.dma_in) return launch<MB, NB, true, 4, 0>(dv, stream);
    return launch<MB, NB, false, 4, 0>(dv, stream);
}

template <int MB>
int launch_nb(const Derived& dv, hipStream_t stream) {
    switch (dv.nb) {
        case 1: return launch_variant<MB, 1>(dv, stream);
        case 2: return launch_variant<MB, 2>(dv, stream);
        default: return launch_variant<MB, 4>(dv, stream);
    }
}

}  // namespace

extern "C" size_t mr_conv_packed_weight_floats(int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                               int32_t kh, int32_t kw, int32_t mb, int32_t ck) {
    if (!src_channels || num_src < 1 || !valid_mb(mb) || !valid_ck(ck)) return 0;
    const int groups = mr_ceil_div(mr_ceil_div(out_channels, 16), mb);
    int cpad_total = 0;
    for (int s = 0; s < num_src; ++s) cpad_total += mr_pad4(src_channels[s]);
    return (size_t)groups * kh * kw * (cpad_total / 4) * mb * 64;
}

extern "C" int mr_conv_pack_weights_f32(const float* weight, int32_t out_channels, const int32_t* src_channels,
                                        int32_t num_src, int32_t kh, int32_t kw, int32_t mb, int32_t ck, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES) return MR_ERR_BAD_ARGUMENT;
    if (!valid_mb(mb) || !valid_ck(ck)) return MR_ERR_BAD_ARGUMENT;
    const int groups = mr_ceil_div(mr_ceil_div(out_channels, 16), mb);
    const int taps = kh * kw;
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    size_t o = 0;
    for (int g = 0; g < groups; ++g) {
        int cin_off = 0;
        for (int s = 0; s < num_src; ++s) {
            const int cpad = mr_pad4(src_channels[s]);
            for (int c0 = 0; c0 < cpad; c0 += ck) {
                const int ckq = cpad - c0 < ck ? cpad - c0 : ck;
                for (int tap = 0; tap < taps; ++tap)
                    for (int c4 = 0; c4 < ckq / 4; ++c4)
                        for (int m = 0; m < mb; ++m)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int cout = (g * mb + m) * 16 + (lane & 15);
                                const int cl = c0 + c4 * 4 + (lane >> 4);
                                float v = 0.f;
                                if (cout < out_channels && cl < src_channels[s])
                                    v = weight[((size_t)cout * cin_total + (cin_off + cl)) * taps + tap];
                                dst[o++] = v;
                            }
            }
            cin_off += src_channels[s];
        }
    }
    return 0;
}

namespace {
uint16_t bf16_rne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu)) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
}  // namespace

extern "C" size_t mr_conv_packed_weight_floats_bf16(int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                                    int32_t kh, int32_t kw, int32_t mb, int32_t ck) {
    if (!src_channels || num_src < 1 || !valid_mb(mb) || !valid_ck(ck) || ck < 16) return 0;
    const int groups = mr_ceil_div(mr_ceil_div(out_channels, 16), mb);
    int cpad_total = 0;
    for (int s = 0; s < num_src; ++s) cpad_total += mr_pad16(src_channels[s]);
    return (size_t)groups * kh * kw * (cpad_total / 16) * mb * 128;
}

extern "C" int mr_conv_pack_weights_bf16(const float* weight, int32_t out_channels, const int32_t* src_channels,
                                         int32_t num_src, int32_t kh, int32_t kw, int32_t mb, int32_t ck, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES) return MR_ERR_BAD_ARGUMENT;
    if (!valid_mb(mb) || !valid_ck(ck) || ck < 16) return MR_ERR_BAD_ARGUMENT;
    const int groups = mr_ceil_div(mr_ceil_div(out_channels, 16), mb);
    const int taps = kh * kw;
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    uint16_t* o = (uint16_t*)dst;
    for (int g = 0; g < groups; ++g) {
        int cin_off = 0;
        for (int s = 0; s < num_src; ++s) {
            const int cpad = mr_pad16(src_channels[s]);
            for (int c0 = 0; c0 < cpad; c0 += ck) {
                const int ckq = cpad - c0 < ck ? cpad - c0 : ck;
                for (int tap = 0; tap < taps; ++tap)
                    for (int c16 = 0; c16 < ckq / 16; ++c16)
                        for (int m = 0; m < mb; ++m)
                            for (int lane = 0; lane < 64; ++lane)
                                for (int j = 0; j < 4; ++j) {
                                    const int cout = (g * mb + m) * 16 + (lane & 15);
                                    const int cl = c0 + c16 * 16 + 4 * j + (lane >> 4);     // element j of lane group g: channel 4j+g
                                    float v = 0.f;
                                    if (cout < out_channels && cl < src_channels[s])
                                        v = weight[((size_t)cout * cin_total + (cin_off + cl)) * taps + tap];
                                    *o++ = bf16_rne(v);
                                }
            }
            cin_off += src_channels[s];
        }
    }
    return 0;
}

extern "C" size_t mr_conv_packed_weight_floats_bf16x3(int32_t out_channels, const int32_t* src_channels, int32_t num_src,
                                                      int32_t kh, int32_t kw, int32_t mb, int32_t ck) {
    return 2 * mr_conv_packed_weight_floats_bf16(out_channels, src_channels, num_src, kh, kw, mb, ck);
}

extern "C" int mr_conv_pack_weights_bf16x3(const float* weight, int32_t out_channels, const int32_t* src_channels,
                                           int32_t num_src, int32_t kh, int32_t kw, int32_t mb, int32_t ck, float* dst) {
    if (!weight || !dst || !src_channels || num_src < 1 || num_src > MR_MAX_SOURCES) return MR_ERR_BAD_ARGUMENT;
    if (!valid_mb(mb) || !valid_ck(ck) || ck < 16) return MR_ERR_BAD_ARGUMENT;
    const int groups = mr_ceil_div(mr_ceil_div(out_channels, 16), mb);
    const int taps = kh * kw;
    int cin_total = 0;
    for (int s = 0; s < num_src; ++s) cin_total += src_channels[s];
    uint16_t* o = (uint16_t*)dst;
    for (int g = 0; g < groups; ++g) {
        int cin_off = 0;
        for (int s = 0; s < num_src; ++s) {
            const int cpad = mr_pad16(src_channels[s]);
            for (int c0 = 0; c0 < cpad; c0 += ck) {
                const int ckq = cpad - c0 < ck ? cpad - c0 : ck;
                for (int tap = 0; tap < taps; ++tap)
                    for (int c16 = 0; c16 < ckq / 16; ++c16)
                        for (int m = 0; m < mb; ++m)
                            for (int lane = 0; lane < 64; ++lane) {
                                uint16_t hi[4], lo[4];                       // same element order as the bf16 layout
                                for (int j = 0; j < 4; ++j) {
                                    const int cout = (g * mb + m) * 16 + (lane & 15);
                                    const int cl = c0 + c16 * 16 + 4 * j + (lane >> 4);
                                    float v = 0.f;
                                    if (cout < out_channels && cl < src_channels[s])
                                        v = weight[((size_t)cout * cin_total + (cin_off + cl)) * taps + tap];
                                    hi[j] = bf16_rne(v);
                                    const uint32_t hb = (uint32_t)hi[j] << 16;
                                    float hf;
                                    memcpy(&hf, &hb, 4);
                                    lo[j] = bf16_rne(v - hf);
                                }
                                for (int j = 0; j < 4; ++j) *o++ = hi[j];
                                for (int j = 0; j < 4; ++j) *o++ = lo[j];
                            }
            }
            cin_off += src_channels[s];
        }
    }
    return 0;
}

extern "C" int64_t mr_conv2d_lds_bytes(const mr_conv_desc* desc) {
    Derived dv;
    const int rc = derive(desc, &dv);
    if (rc != 0) return rc;
    return (int64_t)dv.lds_bytes;
}

extern "C" int mr_conv2d_f32(const mr_conv_desc* desc, void* stream_) {
    Derived dv;
    int rc = derive(desc, &dv);
    if (rc != 0) return rc;
    if (desc->split_k > 1 && !desc->workspace) return MR_ERR_BAD_ARGUMENT;
    hipStream_t stream = (hipStream_t)stream_;
#ifdef MR_CONV_FAST_COMPILE                            // ISA inspection only (tools/isa_loops.py): one register tile instead of fifteen
    rc = launch_variant<MR_CONV_FAST_COMPILE_MB, MR_CONV_FAST_COMPILE_NB>(dv, stream);
#else
    switch (dv.mb) {
        case 1: rc = launch_nb<1>(dv, stream); break;
        case 2: rc = launch_nb<2>(dv, stream); break;
        case 3: rc = launch_nb<3>(dv, stream); break;
        case 4: rc = launch_nb<4>(dv, stream); break;
        default: rc = launch_nb<6>(dv, stream); break;
    }
#endif
    if (rc != 0) return rc;
    if (dv.k.ksplit > 1) {
        const ConvKArgs& k = dv.k;
        const long long total = (long long)k.nphase * k.batch * k.Cout * k.Ho * k.Wo;
        const bool dense = k.nphase == 1 && k.ostep_h == 1 && k.ostep_w == 1 && k.ooff_h[0] == 0 && k.ooff_w[0] == 0 && k.dst_H == k.Ho &&
                           k.dst_W == k.Wo && ((k.Ho * k.Wo) & 3) == 0 && total < (1ll << 31) &&
                           (((unsigned long long)k.dst | (unsigned long long)k.ws | (unsigned long long)k.res) & 15) == 0;
        const unsigned blocks4 = (unsigned)((total / 4 + 255) / 256);
        if (dense && k.ksplit == 2) hipLaunchKernelGGL((splitk_epilogue4_kernel<2>), dim3(blocks4), dim3(256), 0, stream, k);
        else if (dense && k.ksplit == 4) hipLaunchKernelGGL((splitk_epilogue4_kernel<4>), dim3(blocks4), dim3(256), 0, stream, k);
        else if (dense && k.ksplit == 8) hipLaunchKernelGGL((splitk_epilogue4_kernel<8>), dim3(blocks4), dim3(256), 0, stream, k);
        else if (dense && k.ksplit == 16) hipLaunchKernelGGL((splitk_epilogue4_kernel<16>), dim3(blocks4), dim3(256), 0, stream, k);
        else {
            const unsigned blocks = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
            hipLaunchKernelGGL(splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, stream, k);
        }
        rc = (int)hipGetLastError();
    }
    return rc;
}
