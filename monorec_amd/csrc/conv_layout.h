// Packed-weight / K-loop layout shared by the host packer and the gfx950 kernel (conv_mfma.hip).
//
// GEMM view of one convolution (MI355X-first, not a cuDNN-style im2col):
//   D[cout][pixel] += sum_k  A[cout][k] * B[k][pixel]
//   A = weights, B = input samples, one v_mfma_f32_16x16x4_f32 per (16 couts) x (16 pixels of one
//   output row) x (4 consecutive input channels of one filter tap).  Tensors stay NCHW, so the B
//   fragment of a wave (lane l: pixel l&15, channel l>>4) is 16 consecutive floats of 4 channel
//   planes, and the D fragment stores 64-byte row segments per output channel - no layout transforms.
//
// K is walked as:  for chunk (<=16 channels of one concat source, padded to a multiple of 4)
//                    for tap (ky,kx row-major)
//                      for c4 (4-channel group inside the chunk)       <- one MFMA k-step
// and the packed weights are the A fragments in exactly that order:
//   frag(chunk, tap, c4, cb)[lane] = W[cb*16 + (lane&15)][chunk.c0 + c4*4 + (lane>>4)][tap]
// stored as 64 consecutive floats; fragments of one k-step for all cout blocks cb are adjacent, so a
// workgroup owning MB consecutive cout blocks reads MB*256 contiguous bytes per k-step.
#pragma once
#include <stdint.h>

#define MR_CHUNK_CHANNELS 16

#if defined(__HIPCC__)
#define MR_HD __host__ __device__
#else
#define MR_HD
#endif

MR_HD static inline int mr_pad4(int c) { return (c + 3) & ~3; }
MR_HD static inline int mr_ceil_div(int a, int b) { return (a + b - 1) / b; }

// number of K chunks for a source with `c` real channels
MR_HD static inline int mr_chunks_of(int c) { return mr_ceil_div(mr_pad4(c), MR_CHUNK_CHANNELS); }

// floats of packed weights contributed by one chunk of `ck` (multiple of 4) channels
MR_HD static inline int64_t mr_chunk_weight_floats(int ck, int taps, int cout_blocks) {
    return (int64_t)taps * (ck / 4) * cout_blocks * 64;
}
