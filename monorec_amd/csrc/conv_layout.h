// Packed-weight / K-loop layout shared by the host packer and the gfx950 kernel (conv_mfma.hip).
//
// GEMM view of one convolution (MI355X-first, not a cuDNN-style im2col):
//   D[cout][pixel] += sum_k  A[cout][k] * B[k][pixel]
//   A = weights, B = input samples, one v_mfma_f32_16x16x4_f32 per (16 couts) x (16 pixels of one
//   output row) x (4 consecutive input channels of one filter tap).  Tensors stay NCHW, so the B
//   fragment of a wave (lane l: pixel l&15, channel l>>4) is 16 consecutive floats of 4 channel
//   planes, and the D fragment stores 64-byte row segments per output channel - no layout transforms.
//
// K is walked as:  for chunk (<= CK channels of one concat source, padded to a multiple of 4)
//                    for tap (ky,kx row-major)
//                      for c4 (4-channel group inside the chunk)       <- one MFMA k-step
// A workgroup owns MB consecutive 16-channel output blocks ("cout group" g).  The packed weights are
// the A fragments in exactly the order that workgroup consumes them:
//   [g][chunk][tap][c4][m]  ->  64 floats:  lane l = W[(g*MB+m)*16 + (l&15)][chunk.c0 + c4*4 + (l>>4)][tap]
// so the weights of one (group, chunk) are ONE contiguous block that is DMA'd (global_load_lds) into
// LDS once per chunk and read lane-linearly (bank-conflict free) by every wave.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MR_HD __host__ __device__
#else
#define MR_HD
#endif

MR_HD static inline int mr_pad4(int c) { return (c + 3) & ~3; }
MR_HD static inline int mr_ceil_div(int a, int b) { return (a + b - 1) / b; }

// number of K chunks for a source with `c` real channels when chunks hold `ck` channels
MR_HD static inline int mr_chunks_of(int c, int ck) { return mr_ceil_div(mr_pad4(c), ck); }

// floats of packed weights of one (cout group, chunk): `ckq` = channels of this chunk (multiple of 4)
MR_HD static inline int64_t mr_chunk_weight_floats(int ckq, int taps, int mb) {
    return (int64_t)taps * (ckq / 4) * mb * 64;
}

// ---- bf16 MFMA mode (v_mfma_f32_16x16x16_bf16: weights stored as bf16, activations rounded to bf16 when the B fragment is
// formed, fp32 accumulate).  One k-step = 16 input channels of one tap; sources are padded to multiples of 16.  Lane
// (l&15 = cout, g = l>>4) holds, as its 4 consecutive k elements j = 0..3, channels c0 + 16*c16 + 4*j + g - element j of
// lane group g is channel 4j+g, not 4g+j, so that the four lane groups of a B read touch neighbouring channel planes
// (plane stride = 16 mod 32 banks: conflict free) exactly like the fp32 path.  Packed order
//   [g][chunk][tap][c16][m] -> 64 lanes x 4 bf16 (8 bytes per lane) = 128 floats.
MR_HD static inline int mr_pad16(int c) { return (c + 15) & ~15; }
MR_HD static inline int mr_pad_channels(int c, int bf16) { return bf16 ? mr_pad16(c) : mr_pad4(c); }
