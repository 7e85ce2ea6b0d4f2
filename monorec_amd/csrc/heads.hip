// One-channel output layers of the MonoRec inference path as HBM-bound kernels (gfx950).
//
// A convolution with ONE output channel is a per-pixel dot product: on the MFMA kernel it fills one row of a 16-row tile
// (6 % of the matrix work useful) and still pays the launch + LDS-fill latency of a full layer - 15-20 us each at batch 1 for
// 5-28 MMAC, five launches (+ two split-K finishing launches) per keyframe.  Here:
//   * depth_heads_kernel: the four DepthModule.predictors (PadSameConv2d(3) + Conv2d(C, 1, 3), then abs(tanh) and the
//     inverse-depth affine; reference model/monorec/monorec_model.py:520-523,554-557,716-717) in ONE launch at the end of the
//     decoder - their inputs stay resident, nothing downstream reads their outputs;
//   * mask_classifier_kernel: MaskModule.classifier (Conv2d(C, 1, 1) + Sigmoid, :340-343,383) fused with the mask multiply
//     cost_volume = (1 - cv_mask) * cost_volume (:713) that follows it - the mask never makes a round trip through HBM.
// fp32 FMA chains, fixed summation order (deterministic); they differ from the CPU reference by summation order only.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

#include "../../include/monorec_hip.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct HeadK {
    const float* src;
    const float* w;
    const float* bias;
    float* dst;
    int B, C, H, W;
    int src_bytes;
    int quad;          // 1: four pixels per lane, channels split over the 4 waves; 0: one pixel per lane, 16-way channel split
    int first_block;   // first workgroup of this head in the launch
    int pad_;
};

struct HeadsArgs {
    HeadK h[MR_MAX_HEADS];
    int n;
    float p0, p1;
};

__device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// |tanh(v)| -> (1 - t) * p0 + t * p1: the expression of MR_ACT_ABS_TANH_AFFINE in conv_mfma.hip (monorec_model.py:556,717)
__device__ __forceinline__ float head_activate(float v, float p0, float p1) {
    const float t = fabsf(tanhf(v));
    return (1.f - t) * p0 + t * p1;
}

// Workgroup = 4 waves.  quad mode: lane = 4 consecutive pixels of a row (one 16-byte load per input row and channel + the two
// neighbours), wave = a quarter of the channels.  pixel mode (small maps: too few pixels to fill the chip otherwise): lane =
// (pixel 0..15, channel sub-split 0..3), wave = a quarter again - 16 partial sums per pixel.  The partial sums meet in LDS and
// are added in a fixed order.  Zero padding = buffer loads with an out-of-range offset (the hardware returns 0).
__global__ __launch_bounds__(256) void depth_heads_kernel(const HeadsArgs a) {
    __shared__ float part[4][64][4];
    HeadK h = a.h[0];
#pragma unroll
    for (int i = 1; i < MR_MAX_HEADS; ++i)
        if (i < a.n && (int)blockIdx.x >= a.h[i].first_block) h = a.h[i];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = h.H, W = h.W, C = h.C;
    const int HW = H * W;
    const long long total = (long long)h.B * HW;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)h.src, 0, h.src_bytes, 0x00020000);
    const int blk = (int)blockIdx.x - h.first_block;
    if (h.quad) {
        const int per = (C + 3) >> 2;
        const int c_lo = wave * per, c_hi = min(C, c_lo + per);
        const long long p = ((long long)blk * 64 + lane) * 4;
        const bool ok = p < total;
        const int b = ok ? (int)(p / HW) : 0;
        const int r = ok ? (int)(p - (long long)b * HW) : 0;
        const int y = r / W, x = r - y * W;
        const int base = (b * C * HW + y * W + x) * 4;
        int vo[3], vl[3], vr[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            const bool rok = ok && yy >= 0 && yy < H;
            vo[dy] = rok ? base + (dy - 1) * W * 4 : -1;
            vl[dy] = (rok && x > 0) ? vo[dy] - 4 : -1;
            vr[dy] = (rok && x + 4 < W) ? vo[dy] + 16 : -1;
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 3
        for (int c = c_lo; c < c_hi; ++c) {
            const int so = c * HW * 4;
            const float* wc = h.w + c * 9;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const u32x4 mraw = __builtin_amdgcn_raw_buffer_load_b128(rs, vo[dy], so, 0);
                const float l = ld1(rs, vl[dy], so), rr = ld1(rs, vr[dy], so);
                const float m0 = __uint_as_float(mraw.x), m1 = __uint_as_float(mraw.y), m2 = __uint_as_float(mraw.z), m3 = __uint_as_float(mraw.w);
                const float w0 = wc[dy * 3], w1 = wc[dy * 3 + 1], w2 = wc[dy * 3 + 2];
                acc[0] = fmaf(w2, m1, fmaf(w1, m0, fmaf(w0, l, acc[0])));
                acc[1] = fmaf(w2, m2, fmaf(w1, m1, fmaf(w0, m0, acc[1])));
                acc[2] = fmaf(w2, m3, fmaf(w1, m2, fmaf(w0, m1, acc[2])));
                acc[3] = fmaf(w2, rr, fmaf(w1, m3, fmaf(w0, m2, acc[3])));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) part[wave][lane][j] = acc[j];
        __syncthreads();
        if (wave == 0 && ok) {
            const float bias = h.bias[0];
            float4 o;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = head_activate((((part[0][lane][j] + part[1][lane][j]) + part[2][lane][j]) + part[3][lane][j]) + bias, a.p0, a.p1);
            o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = v[3];
            *(float4*)(h.dst + p) = o;
        }
    } else {
        const int pix = lane & 15, sub = lane >> 4;
        const int s = wave * 4 + sub;                         // channel split 0..15
        const int per = (C + 15) >> 4;
        const int c_lo = s * per, c_hi = min(C, c_lo + per);
        const long long p = (long long)blk * 16 + pix;
        const bool ok = p < total;
        const int b = ok ? (int)(p / HW) : 0;
        const int r = ok ? (int)(p - (long long)b * HW) : 0;
        const int y = r / W, x = r - y * W;
        int vo[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            vo[t] = (ok && yy >= 0 && yy < H && xx >= 0 && xx < W) ? (b * C * HW + yy * W + xx) * 4 : -1;
        }
        // (staging the weights in LDS first was measured: slower, 25.1 vs 22.5 us for the four c2 heads - one more round trip
        // and a barrier in front of a kernel that is one latency chain anyway)
        float acc = 0.f;
#pragma unroll 4
        for (int c = c_lo; c < c_hi; ++c) {
            const int co = c * HW * 4;
            const float* wc = h.w + c * 9;
            float xv[9], wv[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                xv[t] = ld1(rs, vo[t] < 0 ? -1 : vo[t] + co, 0);
                wv[t] = wc[t];
            }
#pragma unroll
            for (int t = 0; t < 9; ++t) acc = fmaf(wv[t], xv[t], acc);
        }
        part[wave][lane][0] = acc;
        __syncthreads();
        if (wave == 0 && sub == 0 && ok) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) v += part[q >> 2][(q & 3) * 16 + pix][0];
            h.dst[p] = head_activate(v + h.bias[0], a.p0, a.p1);
        }
    }
}

// Thread = VEC (1 or 2) neighbouring pixels of one sample: C-term dot product, sigmoid, mask store, then the D planes of the cost
// volume scaled in place.  Every access is a coalesced stream over a plane; the loops run in batches of 16 independent loads (the
// kernel is bound by how many bytes it keeps in flight: ~60 MB per keyframe at c2 behind ~2 us of memory latency).
template <int VEC>
struct VecF;
template <>
struct VecF<1> { typedef float type; };
template <>
struct VecF<2> { typedef float2 type; };

__device__ __forceinline__ float vget(float v, int) { return v; }
__device__ __forceinline__ float vget(float2 v, int i) { return i ? v.y : v.x; }
__device__ __forceinline__ void vset(float& v, int, float x) { v = x; }
__device__ __forceinline__ void vset(float2& v, int i, float x) { if (i) v.y = x; else v.x = x; }

__device__ __forceinline__ unsigned hd_pack_bf16x2(float a, float b) {      // round to nearest even, a in the low half
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, h);
}

// cvb8 (optional; D % 16 == 0): the masked volume once more in the channel-blocked bf16 layout of csrc/conv_b8.hip - what the first
// layer of the depth net reads in the bf16 MFMA mode instead of staging the fp32 volume through registers
template <int VEC>
__global__ __launch_bounds__(256) void mask_classifier_kernel(const float* __restrict__ feat_, const float* __restrict__ w,
                                                              const float* __restrict__ bias, int C, long long planev, long long totalv,
                                                              float* __restrict__ mask_, float* cv_, int D, u32x4* __restrict__ cvb8) {
    typedef typename VecF<VEC>::type V;
    const long long i = blockIdx.x * 256ll + threadIdx.x;
    if (i >= totalv) return;
    const long long b = i / planev, p = i - b * planev;
    const V* f = (const V*)feat_ + b * C * planev + p;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    int c = 0;
    for (; c + 16 <= C; c += 16) {
        V x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = f[(long long)(c + u) * planev];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float wc = w[c + u];
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] = fmaf(wc, vget(x[u], j), acc[j]);
        }
    }
    for (; c < C; ++c) {
        const V x = f[(long long)c * planev];
        const float wc = w[c];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = fmaf(wc, vget(x, j), acc[j]);
    }
    const float bs = bias[0];
    V m;
    float keep[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float mj = 1.f / (1.f + expf(-(acc[j] + bs)));                    // MR_ACT_SIGMOID of conv_mfma.hip
        vset(m, j, mj);
        keep[j] = 1.0f - mj;
    }
    ((V*)mask_)[i] = m;
    if (cv_) {                                                                  // monorec_model.py:713
        V* v = (V*)cv_ + b * D * planev + p;
        int d = 0;
        for (; d + 16 <= D; d += 16) {
            V t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = v[(long long)(d + u) * planev];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) vset(t[u], j, keep[j] * vget(t[u], j));
                v[(long long)(d + u) * planev] = t[u];
            }
            if (cvb8) {
#pragma unroll
                for (int h8 = 0; h8 < 2; ++h8)
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const int u0 = h8 * 8;
                        cvb8[((b * (D >> 3) + ((d >> 3) + h8)) * planev + p) * VEC + j] =
                            (u32x4){hd_pack_bf16x2(vget(t[u0], j), vget(t[u0 + 1], j)), hd_pack_bf16x2(vget(t[u0 + 2], j), vget(t[u0 + 3], j)),
                                    hd_pack_bf16x2(vget(t[u0 + 4], j), vget(t[u0 + 5], j)), hd_pack_bf16x2(vget(t[u0 + 6], j), vget(t[u0 + 7], j))};
                    }
            }
        }
        for (; d < D; ++d) {
            V t = v[(long long)d * planev];
#pragma unroll
            for (int j = 0; j < VEC; ++j) vset(t, j, keep[j] * vget(t, j));
            v[(long long)d * planev] = t;
        }
    }
}

}  // namespace

extern "C" int mr_depth_heads_f32(const mr_head_desc* heads, int32_t num_heads, float act_p0, float act_p1, void* stream) {
    if (!heads || num_heads < 1 || num_heads > MR_MAX_HEADS) return MR_ERR_BAD_ARGUMENT;
    // pixels from which a head runs in quad mode (MR_HEADS_QUAD_MIN: tuning aid, read once per process).  Measured at the c2 decoder
    // sizes (2 048 / 8 192 / 32 768 / 131 072 pixels): quad mode wins from 32 768 pixels up, pixel mode below (tools/bench_heads.py)
#ifdef MR_TUNING_ENV         // tuning aid: diagnostic library only; the product reads no environment
    static const long long quad_min = [] { const char* e = getenv("MR_HEADS_QUAD_MIN"); return e ? atoll(e) : 16384ll; }();
#else
    const long long quad_min = 16384ll;    // measured (tools/bench_heads.py): quad mode from 16 384 pixels up
#endif
    HeadsArgs a;
    a.n = num_heads;
    a.p0 = act_p0;
    a.p1 = act_p1;
    long long blocks = 0;
    for (int i = 0; i < MR_MAX_HEADS; ++i) {
        const mr_head_desc& d = heads[i < num_heads ? i : 0];
        if (!d.src || !d.weight || !d.bias || !d.dst || d.batch < 1 || d.channels < 1 || d.height < 1 || d.width < 1)
            return MR_ERR_BAD_ARGUMENT;
        const long long bytes = (long long)d.batch * d.channels * d.height * d.width * 4;
        if (bytes >= (1ll << 31)) return MR_ERR_UNSUPPORTED;              // 32-bit byte offsets in the buffer descriptor
        const long long pixels = (long long)d.batch * d.height * d.width;
        HeadK& k = a.h[i];
        k.src = d.src; k.w = d.weight; k.bias = d.bias; k.dst = d.dst;
        k.B = d.batch; k.C = d.channels; k.H = d.height; k.W = d.width;
        k.src_bytes = (int)bytes;
        k.quad = (d.width % 4 == 0 && pixels >= quad_min) ? 1 : 0;
        k.first_block = (int)blocks;
        k.pad_ = 0;
        if (i < num_heads) blocks += k.quad ? (pixels / 4 + 63) / 64 : (pixels + 15) / 16;
        if (blocks >= (1ll << 31)) return MR_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(depth_heads_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

namespace {
int mask_classifier_launch(const float* features, const float* weight, const float* bias, int32_t batch, int32_t channels, int64_t plane,
                           float* cv_mask, float* cost_volume, int32_t num_depths, void* cv_b8, void* stream);
}

extern "C" int mr_mask_classifier_f32(const float* features, const float* weight, const float* bias, int32_t batch,
                                      int32_t channels, int64_t plane, float* cv_mask, float* cost_volume, int32_t num_depths,
                                      void* stream) {
    return mask_classifier_launch(features, weight, bias, batch, channels, plane, cv_mask, cost_volume, num_depths, nullptr, stream);
}

// mr_mask_classifier_f32 that ALSO writes the masked cost volume in the channel-blocked bf16 layout of mr_conv2d_b8:
// cost_volume_b8 = (batch, num_depths / 8, plane, 8) bf16; num_depths % 16 == 0, cost_volume required.
extern "C" int mr_mask_classifier_b8_f32(const float* features, const float* weight, const float* bias, int32_t batch,
                                         int32_t channels, int64_t plane, float* cv_mask, float* cost_volume, int32_t num_depths,
                                         void* cost_volume_b8, void* stream) {
    if (!cost_volume || !cost_volume_b8 || (num_depths & 15)) return MR_ERR_BAD_ARGUMENT;
    return mask_classifier_launch(features, weight, bias, batch, channels, plane, cv_mask, cost_volume, num_depths, cost_volume_b8, stream);
}

namespace {
int mask_classifier_launch(const float* features, const float* weight, const float* bias, int32_t batch, int32_t channels, int64_t plane,
                           float* cv_mask, float* cost_volume, int32_t num_depths, void* cv_b8, void* stream) {
    if (!features || !weight || !bias || !cv_mask || batch < 1 || channels < 1 || plane < 2 || (plane & 1)) return MR_ERR_BAD_ARGUMENT;
    if (cost_volume && num_depths < 1) return MR_ERR_BAD_ARGUMENT;
    // one pixel per thread while that still leaves the chip short of waves (<= 2 workgroups of 256 per CU), two beyond
    const int vec = (long long)batch * plane <= 256ll * 256 * 2 ? 1 : 2;
    const long long planev = plane / vec, totalv = (long long)batch * planev;
    const long long blocks = (totalv + 255) / 256;
    if (blocks >= (1ll << 31)) return MR_ERR_UNSUPPORTED;
    if (vec == 1)
        hipLaunchKernelGGL(mask_classifier_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, features, weight, bias,
                           channels, planev, totalv, cv_mask, cost_volume, num_depths, (u32x4*)cv_b8);
    else
        hipLaunchKernelGGL(mask_classifier_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, features, weight, bias,
                           channels, planev, totalv, cv_mask, cost_volume, num_depths, (u32x4*)cv_b8);
    return (int)hipGetLastError();
}
}  // namespace
