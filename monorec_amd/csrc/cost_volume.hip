// Plane-sweep cost volume for gfx950 (MI355X): replaces CostVolumeModule.forward's per-pixel work,
// reference model/monorec/monorec_model.py:193-271 (+ model/layers.py:63-71 point_projection,
// :119-137 SSIM, F.grid_sample x2, F.conv3d) - about 60 full-tensor ATen passes over (D*F,3,H,W)
// temporaries in the reference.  Two launches:
//
//  A  cv_sad_kernel<TX,TY>   one workgroup per (keyframe tile, source frame, depth chunk); one thread per pixel.
//     For two depth hypotheses at a time it (a) projects the tile + 2 px halo into the source frame and gathers
//     RGB bilinearly (the 1.5 MB source image is L2 resident; the footprint of a tile is data dependent, so it
//     is gathered, not LDS-windowed) -> warped planes in LDS, (b) evaluates the 3x3 SSIM distance on tile + 1 px
//     (reflection at image borders), channel weighted -> LDS, (c) 3x3 box sum (zero padded) -> sad(f,d,pixel),
//     stored straight into the single-frame-volume buffer.  Only ~31 KB of LDS per workgroup, so several
//     workgroups share a CU and hide each other's dependent VALU chains and gather latency (the first, fully
//     fused version kept every sad of a tile in LDS - 150 KB - ran one workgroup per CU and was 2-5x slower).
//     The all-depth AND of the warped border mask (validity, :218-219) travels in the sign bit of the last plane
//     each chunk writes (sad >= 0, so the sign is free).
//  B  cv_fuse_kernel         one thread per pixel: validity, soft-min frame weights (:257-260), in-place
//     sfcv = (1 - 2 sad) * valid (:251) and the fused volume (:262-269).  HBM-bound (reads F*D twice - the second
//     time from L2 - writes F*D + D).
//
//  A' cv_sad_patch_kernel    the generic variant of A for cv_patch_size != 3 (P x P box, border radius P / 2 + 1): same
//     arithmetic, strided loops instead of the hand-scheduled 3x3 tile; B is shared.
//
// Arithmetic follows the reference operation by operation with contraction disabled
// (-ffp-contract=off) and explicit fmaf where the CPU reference fuses (MKL sgemm k-ascending FMA
// chain; ATen grid_sampler unnormalise + bilinear FMA chain; see oracle/make_golden.py for the
// bitwise pinning), so the projection, the warped samples and the validity mask reproduce the CPU bits.  Two places are within an ulp
// instead (round 3): the SSIM ratio (v_rcp_f32 + multiply, ssim_ratio below: sad moves by <= 1e-7; single-frame volumes stay
// <= 2e-6 from the reference fixtures) and nothing else.  A non-finite u / v (pz + 1e-7 == 0) goes through div_const's fast path as NaN
// and leaves the clamp as -2 where the reference's +-inf leaves it as +-2: both are outside the image on the same side of the validity
// test (every tap out of range, bilinear 0, border-mask sample 0), so the outputs are the same.
#include <hip/hip_runtime.h>
#include <map>
#include <mutex>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/monorec_hip.h"

namespace {

struct CvArgs {
    const float* keyframe;
    const float* frames[MR_MAX_FRAMES];
    float* sfcv[MR_MAX_FRAMES];
    const float* kinv;    // B x 9
    const float* proj;    // B x F x 12
    const float* depths;  // D
    const float* pix_depths;  // B x D x H x W per-pixel hypotheses (data_dict['cv_depths'], monorec_model.py:181-182) or null
    float* cv;
    int F, B, D, H, W;
    int tiles_x, nchunk, dchunk;
    float alpha;
    float cw[3];          // channel_weight / 9 (fp32 division, monorec_model.py:141)
    float inv_dm1;        // fp32(1/(D-1)) (python double division, then cast; :258)
    int border;           // border_radius = patch_size / 2 + 1 (monorec_model.py:139); 2 for the default 3x3 patch
    float wm1, hm1;       // fp32(W - 1), fp32(H - 1): the divisors of layers.py:67-68
    float rwm1, rhm1;     // fp32(1 / wm1), fp32(1 / hm1)
    int relaxed_sums;     // 1 (mr_cost_volume_b8_f32 = the bf16 configuration only): separable 3x3 window sums and x * fp32(1/9) - see march_finish
    int fast_w, fast_h;   // 1: the 3-instruction sequence div_const() equals the correctly rounded quotient for EVERY fp32 dividend
                          // (checked exhaustively on the host, mr_exact_const_division); 0: IEEE division
    int lean;             // 1 (mr_cost_volume_b8_lean_f32): the fusion kernel does NOT finalise the dense fp32 single-frame volumes (sfcv stays scratch)
    void* sfcv_b8[MR_MAX_FRAMES];   // optional second copy of the single-frame volumes in the channel-blocked bf16 layout of csrc/conv_b8.hip
                                    // ((B, D / 8, H, W, 8) bf16 per frame; the bf16 MFMA mode's mask encoder reads it), or null
};

// a / 9.0f in 3 instructions instead of the ~10 of the IEEE division sequence: q0 = a*y, r = fma(-9, q0, a),
// q = fma(r, y, q0) with y = fp32(1/9).  Bit-identical to the correctly rounded quotient for every fp32 mantissa
// (checked exhaustively, scale invariant above 2^-100; tools/probes/div9_check.py) - the reference divides
// (AvgPool2d), it does not multiply by a reciprocal, and SSIM amplifies 1-ulp differences.
__device__ __forceinline__ float div9(float a) {
    const float y = 0x1.c71c72p-4f;
    const float q0 = a * y;
    const float r = fmaf(-9.0f, q0, a);
    return fmaf(r, y, q0);
}

// a / d for the constant divisors W - 1 / H - 1 of layers.py:67-68 by the same 3-instruction sequence (y = fp32(1/d)) - used only
// where the host has checked the sequence against the correctly rounded quotient for all 2^24 mantissas of the dividend (it is
// scale invariant, cannot overflow because |y| < 1, and a dividend small enough to underflow vanishes in `u - 0.5` anyway);
// `fast` is a kernel argument, i.e. wave-uniform.  The pixel coordinate decides validity: this division stays bit-exact.
__device__ __forceinline__ float div_const(float a, float d, float y, int fast) {
    if (fast) {
        const float q0 = a * y;
        const float r = fmaf(-d, q0, a);
        return fmaf(r, y, q0);
    }
    return a / d;
}

// The SSIM ratio of layers.py:137.  The reference divides; here v_rcp_f32 (1 ulp) and one multiply: the quotient lies in [-1, 1],
// so the result moves by <= 2e-7 and the sad by <= 1e-7 - it feeds nothing index-like (round 3: -8 VALU instructions per channel).
// sd >= C1 * C2 = 9e-8: a normal number, never 0.
__device__ __forceinline__ float ssim_ratio(float sn, float sd) { return sn * __builtin_amdgcn_rcpf(sd); }

__device__ __forceinline__ int reflect_idx(int i, int n) {  // nn.ReflectionPad2d(1), layers.py:112
    return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i);
}

// One bilinear sample position of frame f at depth `depth` for keyframe pixel ray (r0,r1,r2).
struct Sample {
    int x0, y0;
    float nw, ne, sw, se;
};

__device__ __forceinline__ Sample project(float r0, float r1, float r2, float depth, const float* P, int H, int W, const CvArgs& a) {
    // cam point = depth * ray (monorec_model.py:200); pc = P[:, :3] X + P[:, 3] as MKL's k-ascending FMA chain
    const float X0 = depth * r0, X1 = depth * r1, X2 = depth * r2;
    const float pcx = fmaf(P[3], 1.0f, fmaf(P[2], X2, fmaf(P[1], X1, P[0] * X0)));
    const float pcy = fmaf(P[7], 1.0f, fmaf(P[6], X2, fmaf(P[5], X1, P[4] * X0)));
    const float pcz = fmaf(P[11], 1.0f, fmaf(P[10], X2, fmaf(P[9], X1, P[8] * X0)));
    const float z = pcz + 1e-7f;                       // layers.py:66
    float u = pcx / z, v = pcy / z;
    u = div_const(u, a.wm1, a.rwm1, a.fast_w);         // layers.py:67
    v = div_const(v, a.hm1, a.rhm1, a.fast_h);         // layers.py:68
    u = fminf(fmaxf((u - 0.5f) * 2.0f, -2.0f), 2.0f);  // layers.py:69 + clamp(-2,2) monorec_model.py:208
    v = fminf(fmaxf((v - 0.5f) * 2.0f, -2.0f), 2.0f);
    // grid_sample(align_corners=False): unnormalise as fma(g + 1, size/2, -0.5) (ATen GridSamplerKernel)
    const float sx = fmaf(u + 1.0f, (float)W * 0.5f, -0.5f);
    const float sy = fmaf(v + 1.0f, (float)H * 0.5f, -0.5f);
    const float fx = floorf(sx), fy = floorf(sy);
    const float w = sx - fx, e = 1.0f - w, n = sy - fy, s = 1.0f - n;
    Sample o;
    o.x0 = (int)fx;
    o.y0 = (int)fy;
    o.nw = s * e; o.ne = s * w; o.sw = n * e; o.se = n * w;
    return o;
}

// Byte offsets of the four bilinear taps inside one image plane, -1 (out of range for the buffer descriptor,
// the hardware then returns 0 = grid_sample's padding_mode='zeros') for taps outside the image: no branches.
struct Taps { int a, b, c, d; };

__device__ __forceinline__ Taps tap_offsets(const Sample& sp, int H, int W) {
    const bool xl = sp.x0 >= 0 && sp.x0 < W, xr = sp.x0 + 1 >= 0 && sp.x0 + 1 < W;
    const bool yt = sp.y0 >= 0 && sp.y0 < H, yb = sp.y0 + 1 >= 0 && sp.y0 + 1 < H;
    const int o = (sp.y0 * W + sp.x0) * 4;
    Taps t;
    t.a = (xl && yt) ? o : -1;
    t.b = (xr && yt) ? o + 4 : -1;
    t.c = (xl && yb) ? o + W * 4 : -1;
    t.d = (xr && yb) ? o + W * 4 + 4 : -1;
    return t;
}

__device__ __forceinline__ float bilinear(__amdgpu_buffer_rsrc_t img, int plane_bytes, const Taps& t, const Sample& sp) {
    const float a = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t.a, plane_bytes, 0));
    const float b = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t.b, plane_bytes, 0));
    const float c = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t.c, plane_bytes, 0));
    const float d = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t.d, plane_bytes, 0));
    return fmaf(d, sp.se, fmaf(c, sp.sw, fmaf(b, sp.ne, a * sp.nw)));
}

// bilinear sample of the border mask (ones with a 2 px zero frame, monorec_model.py:282-284) != 0
__device__ __forceinline__ bool mask_hit(const Sample& sp, int H, int W) {
    const float ml = (sp.x0 >= 2 && sp.x0 < W - 2) ? 1.f : 0.f, mr = (sp.x0 + 1 >= 2 && sp.x0 + 1 < W - 2) ? 1.f : 0.f;
    const float mt = (sp.y0 >= 2 && sp.y0 < H - 2) ? 1.f : 0.f, mb = (sp.y0 + 1 >= 2 && sp.y0 + 1 < H - 2) ? 1.f : 0.f;
    const float m = fmaf(mr * mb, sp.se, fmaf(ml * mb, sp.sw, fmaf(mr * mt, sp.ne, (ml * mt) * sp.nw)));
    return m != 0.f;
}

// MODE = the photometric term of monorec_model.py:227-243 (use_ssim): 1 SSIM distance (default), 0 absolute difference,
// 2 the 0.85 / 0.15 mix of both, 3 absolute difference averaged over 3x3 (zero padded, avg_pool2d).
// OPT bit 0: per-pixel depth hypotheses (cv_depths).  OPT bit 1: sfcv_mult_mask=False (monorec_model.py:252-253) - the single-frame
// volumes are masked per depth plane by (any channel of the warped pixel != 0) | (warped pixel == keyframe pixel) instead of by
// the all-depth validity; that flag rides in the sign bit of every raw sad plane and the validity moves into the (not yet
// written) cost-volume buffer, plane f of sample b, cleared with an atomic AND.
template <int TX, int TY, int MODE, int OPT>
__global__ __launch_bounds__(TX * TY) void cv_sad_kernel(const CvArgs a) {
    constexpr bool PIXD = OPT & 1, PFLAG = (OPT & 2) != 0;
    constexpr int NT = TX * TY;
    constexpr int HX = TX + 4, HY = TY + 4;   // warped / keyframe tile with 2 px halo
    constexpr int SX = TX + 2, SY = TY + 2;   // SSIM tile with 1 px halo
    constexpr int NHALO = HX * HY - NT;       // halo positions warped by the first NHALO threads
    constexpr int NSS = (SX * SY + NT - 1) / NT;  // SSIM positions per thread (2)
    constexpr int DPI = 2;                    // depth hypotheses per iteration
    static_assert(NHALO <= NT, "halo must fit one extra round");
    static_assert(NSS == 2, "two SSIM rounds expected");

    __shared__ float kf[3 * HY * HX];          // keyframe + 0.5
    __shared__ float wr[DPI * 3 * HY * HX];    // warped + 0.5
    __shared__ float es[DPI * SY * SX];        // channel-weighted SSIM distance

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, D = a.D;
    const int b = blockIdx.z;
    const int f = blockIdx.y / a.nchunk, chunk = blockIdx.y % a.nchunk;
    const int d_lo = chunk * a.dchunk, d_hi = min(D, d_lo + a.dchunk);
    const int ty0 = (blockIdx.x / a.tiles_x) * TY, tx0 = (blockIdx.x % a.tiles_x) * TX;
    const int HWp = H * W;
    const float* kimg = a.keyframe + (long long)b * 3 * HWp;

    // ---- keyframe tile (+0.5) ---------------------------------------------------------------------
    for (int i = tid; i < 3 * HY * HX; i += NT) {
        const int c = i / (HY * HX), r = i % (HY * HX);
        const int gy = ty0 - 2 + r / HX, gx = tx0 - 2 + r % HX;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = kimg[c * HWp + gy * W + gx] + 0.5f;
        kf[i] = v;
    }

    // ---- positions owned by this thread ---------------------------------------------------------------
    const int oly = tid / TX, olx = tid % TX;
    const int opy = ty0 + oly, opx = tx0 + olx;
    const bool own_in = opy < H && opx < W;
    // halo position (threads < NHALO): top 2 rows, bottom 2 rows, then 2+2 side columns
    int hly = 0, hlx = 0;
    bool has_halo = tid < NHALO;
    if (has_halo) {
        if (tid < 2 * HX) { hly = tid / HX; hlx = tid % HX; }
        else if (tid < 4 * HX) { const int j = tid - 2 * HX; hly = TY + 2 + j / HX; hlx = j % HX; }
        else { const int j = tid - 4 * HX; hly = 2 + (j >> 2); const int k = j & 3; hlx = k < 2 ? k : TX + k; }
    }
    const int hpy = ty0 - 2 + hly, hpx = tx0 - 2 + hlx;
    has_halo = has_halo && hpy >= 0 && hpy < H && hpx >= 0 && hpx < W;

    // pixel rays Kinv[:3,:3] @ [x,y,1] (monorec_model.py:199), MKL k-ascending FMA chain
    const float* Ki = a.kinv + b * 9;
    float ro[3], rh[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ro[i] = fmaf(Ki[3 * i + 2], 1.0f, fmaf(Ki[3 * i + 1], (float)opy, Ki[3 * i] * (float)opx));
        rh[i] = fmaf(Ki[3 * i + 2], 1.0f, fmaf(Ki[3 * i + 1], (float)hpy, Ki[3 * i] * (float)hpx));
    }

    __syncthreads();

    // ---- keyframe SSIM statistics of this thread's SSIM positions (registers) -----------------------------
    int sly[NSS], slx[NSS];
    bool s_in[NSS];
    float kmu[NSS][3], ksg[NSS][3];
#pragma unroll
    for (int r = 0; r < NSS; ++r) {
        const int idx = tid + r * NT;
        sly[r] = idx / SX;
        slx[r] = idx % SX;
        const int qy = ty0 - 1 + sly[r], qx = tx0 - 1 + slx[r];
        s_in[r] = idx < SX * SY && qy >= 0 && qy < H && qx >= 0 && qx < W;
#pragma unroll
        for (int c = 0; c < 3; ++c) { kmu[r][c] = 0.f; ksg[r][c] = 0.f; }
        if (s_in[r]) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s1 = 0.f, s2 = 0.f;
                bool first = true;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        const int ly = reflect_idx(qy + dy, H) - (ty0 - 2), lx = reflect_idx(qx + dx, W) - (tx0 - 2);
                        const float k = kf[(c * HY + ly) * HX + lx];
                        const float kk = k * k;
                        if (first) { s1 = k; s2 = kk; first = false; } else { s1 = s1 + k; s2 = s2 + kk; }
                    }
                const float mu = s1 / 9.0f;                     // AvgPool2d(3,1): sequential sum / 9
                kmu[r][c] = mu;
                ksg[r][c] = s2 / 9.0f - mu * mu;                // layers.py:130
            }
        }
    }

    const float C1 = 0x1.a36e2ep-14f, C2 = 0x1.d7dbf4p-11f;   // fp32(0.01**2), fp32(0.03**2)  layers.py:116-117
    const float* P = a.proj + ((long long)b * a.F + f) * 12;
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc((void*)(a.frames[f] + (long long)b * 3 * HWp), 0, 3 * HWp * 4, 0x00020000);
    float* sad_out = a.sfcv[f] + (long long)b * D * HWp + opy * W + opx;
    bool hit_all = true;       // all depth planes of this chunk sample the border mask != 0
    float kraw[3] = {0.f, 0.f, 0.f};
    if (PFLAG && own_in) {
#pragma unroll
        for (int c = 0; c < 3; ++c) kraw[c] = kimg[c * HWp + opy * W + opx];
    }
    bool pflag[DPI];

    for (int d = d_lo; d < d_hi; d += DPI) {
        // ---- (a) warp own pixel + one halo position, DPI depth planes ---------------------------
#pragma unroll
        for (int u = 0; u < DPI; ++u) {
            const int du = min(d + u, D - 1);      // odd D: the second plane of the last pair repeats the last hypothesis and is not stored
            const float depth = PIXD ? 0.f : a.depths[du];
            const float* pd = PIXD ? a.pix_depths + ((long long)b * D + du) * HWp : nullptr;
            float* wru = wr + u * 3 * HY * HX;
            if (own_in) {
                const Sample sp = project(ro[0], ro[1], ro[2], PIXD ? pd[opy * W + opx] : depth, P, H, W, a);
                hit_all = hit_all && mask_hit(sp, H, W);                   // monorec_model.py:218-219
                const Taps tp = tap_offsets(sp, H, W);
                bool any_nz = false, all_eq = true;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float wv = bilinear(img, c * HWp * 4, tp, sp);
                    if (PFLAG) { any_nz = any_nz || wv != 0.f; all_eq = all_eq && wv == kraw[c]; }
                    wru[(c * HY + oly + 2) * HX + olx + 2] = wv + 0.5f;
                }
                pflag[u] = any_nz || all_eq;                                     // :253
            }
            if (has_halo) {
                const Sample sp = project(rh[0], rh[1], rh[2], PIXD ? pd[hpy * W + hpx] : depth, P, H, W, a);
                const Taps tp = tap_offsets(sp, H, W);
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    wru[(c * HY + hly) * HX + hlx] = bilinear(img, c * HWp * 4, tp, sp) + 0.5f;
            }
        }
        __syncthreads();
        // ---- (b) SSIM distance on tile + 1 px halo -------------------------------------------
#pragma unroll
        for (int r = 0; r < NSS; ++r) {
            if (tid + r * NT < SX * SY) {
                int lyy[3], lxx[3];
                if (s_in[r]) {
                    const int qy = ty0 - 1 + sly[r], qx = tx0 - 1 + slx[r];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        lyy[t] = reflect_idx(qy + t - 1, H) - (ty0 - 2);
                        lxx[t] = reflect_idx(qx + t - 1, W) - (tx0 - 2);
                    }
                }
#pragma unroll
                for (int u = 0; u < DPI; ++u) {
                    const float* wru = wr + u * 3 * HY * HX;
                    float e = 0.f;   // zero padding of the 3x3 box (conv3d padding, monorec_model.py:247)
                    if (s_in[r]) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            float sv = 0.f;
                            if (MODE == 1 || MODE == 2) {
                                float sx1 = 0.f, sx2 = 0.f, sxy = 0.f;
#pragma unroll
                                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                                    for (int dx = 0; dx < 3; ++dx) {
                                        const int li = (c * HY + lyy[dy]) * HX + lxx[dx];
                                        const float x = wru[li], k = kf[li];
                                        const float xx = x * x, xk = x * k;
                                        if (dy == 0 && dx == 0) { sx1 = x; sx2 = xx; sxy = xk; }
                                        else { sx1 = sx1 + x; sx2 = sx2 + xx; sxy = sxy + xk; }
                                    }
                                const float mu_x = div9(sx1), mu_y = kmu[r][c];
                                const float mu_x_sq = mu_x * mu_x, mu_y_sq = mu_y * mu_y, mu_xy = mu_x * mu_y;
                                const float sig_x = div9(sx2) - mu_x_sq;
                                const float sig_xy = div9(sxy) - mu_xy;
                                const float sn = (2.0f * mu_xy + C1) * (2.0f * sig_xy + C2);          // layers.py:133
                                const float sd = (mu_x_sq + mu_y_sq + C1) * (sig_x + ksg[r][c] + C2); // layers.py:134
                                sv = fminf(fmaxf((1.0f - ssim_ratio(sn, sd)) / 2.0f, 0.0f), 1.0f);               // layers.py:137
                            }
                            if (MODE == 0 || MODE == 2) {                                             // |warped - keyframe|, :228,239
                                const int li = (c * HY + sly[r] + 1) * HX + slx[r] + 1;
                                const float ad = fabsf(wru[li] - kf[li]);
                                sv = MODE == 0 ? ad : 0.85f * sv + 0.15f * ad;
                            }
                            if (MODE == 3) {                                                          // avg_pool2d(|.|, 3, 1, padding=1), :241
                                const int qy = ty0 - 1 + sly[r], qx = tx0 - 1 + slx[r];
                                float acc = 0.f;
                                bool first = true;
#pragma unroll
                                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                                    for (int dx = -1; dx <= 1; ++dx) {
                                        if (qy + dy < 0 || qy + dy >= H || qx + dx < 0 || qx + dx >= W) continue;   // zero padding
                                        const int li = (c * HY + sly[r] + 1 + dy) * HX + slx[r] + 1 + dx;
                                        const float ad = fabsf(wru[li] - kf[li]);
                                        acc = first ? ad : acc + ad;
                                        first = false;
                                    }
                                sv = div9(acc);
                            }
                            e = (c == 0) ? sv * a.cw[0] : fmaf(sv, a.cw[c], e);
                        }
                    }
                    es[u * SY * SX + sly[r] * SX + slx[r]] = e;
                }
            }
        }
        __syncthreads();
        // ---- (c) 3x3 box sum -> sad, stored raw; the chunk's last plane carries the validity in its sign ----
        if (own_in) {
#pragma unroll
            for (int u = 0; u < DPI; ++u) {
                const float* esu = es + u * SY * SX;
                float s = 0.f;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float v = esu[(oly + dy) * SX + olx + dx];
                        s = (dy == 0 && dx == 0) ? v : s + v;
                    }
                if (PFLAG) { if (!pflag[u]) s = -s; }           // per-plane flag in the sign bit
                else if (d + u == d_hi - 1 && !hit_all) s = -s; // sad >= 0: the sign bit is free (-0.0 keeps it)
                if (d + u < d_hi) sad_out[(long long)(d + u) * HWp] = s;
            }
        }
    }
    if (PFLAG && own_in && !hit_all)
        atomicAnd((unsigned int*)a.cv + ((long long)b * D + f) * HWp + opy * W + opx, 0u);
}

// ---- A2  cv_sad_march_kernel: the default configuration (SSIM, 3x3 patch, all-depth validity) without LDS ------------------
// One WAVE owns a strip of 64 image columns (60 of them produce output, 2 + 2 are halo), a segment of TY rows, one source frame
// and DP depth planes, and marches down the rows.  Lane l holds the warped pixel of column x0 - 2 + l of the current row in
// registers; the left / right neighbours of a 3x3 window come from the adjacent lanes through DPP wave shifts
// (v_add_f32_dpp wave_shr:1 / wave_shl:1 - the shift rides on the add, no LDS, no barrier), the rows above from two rows of
// register state per quantity: the horizontal sum of the window's top row and the raw values of its middle row.  The additions
// are issued in the reference's order (row-major over the window, AvgPool2d / conv3d), so every sum - and with it every SSIM
// value and sad - is bit-identical to cv_sad_kernel's and to the CPU reference; only the data movement differs:
//   * cv_sad_kernel re-reads 2 x 9 LDS words per channel, plane and SSIM position (39 % bank-conflict cycles at pitch 36/34),
//     evaluates 612 SSIM positions with 512 threads in two unbalanced rounds and synchronises twice per pair of planes;
//   * here a window costs 8 VALU adds per quantity, every (pixel, plane) is warped once per strip (halo overhead 64/60 x
//     (TY+4)/TY), and nothing waits on anything but its own gathers.
// Image borders: the reflection padding of the SSIM windows (layers.py:112) is realised by warping "virtual" row -1 / H and
// column -1 / W from the reflected coordinate (1 / H-2, 1 / W-2): the neighbour lane / previous row then IS the reflected tap.
// Validity (monorec_model.py:218-219): the sign bit of every raw sad plane carries "this plane's border-mask sample was 0"
// for its own pixel; cv_fuse ANDs the signs of all planes, which is the reference's product over the depth axis.
// TAG only keeps the compiler from merging two reads of the same neighbour (the middle row's neighbours feed both the window
// and the next top-row sum): merged, they become one v_mov_b32_dpp plus two plain adds; apart, each rides on its v_add_f32_dpp.
// Lanes without a neighbour read 0 either way (bound_ctrl with a zero `old`, or `old` = 0 kept).
template <bool TAG>
__device__ __forceinline__ float dpp_left(float v) {    // value of lane - 1 (0 for lane 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, TAG));
}
template <bool TAG>
__device__ __forceinline__ float dpp_right(float v) {   // value of lane + 1 (0 for lane 63)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, TAG));
}
// The window sums of one step are NQ independent chains of dependent adds (a dependent VALU instruction issues ~8 cycles after
// its producer on this chip, an independent one ~4.6 cycles after the previous instruction of the wave): all chains advance one
// tap at a time, level by level, so that neighbouring instructions are independent.
//   s = ((l + c) + r) of `m`: the horizontal part of a row-major 3x3 sum
template <int N>
__device__ __forceinline__ void hsum3_batch(float (&out)[N], const float (&m)[N]) {
    float t[N];
#pragma unroll
    for (int i = 0; i < N; ++i) t[i] = dpp_left<true>(m[i]) + m[i];
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = t[i] + dpp_right<true>(m[i]);
}
//   top-row sum, then middle and bottom row tap by tap, in the reference's order (AvgPool2d / conv3d: row-major)
template <int N>
__device__ __forceinline__ void win9_batch(float (&s)[N], const float (&hT)[N], const float (&m)[N], const float (&b)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = hT[i] + dpp_left<false>(m[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = s[i] + m[i];
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = s[i] + dpp_right<false>(m[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = s[i] + dpp_left<false>(b[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = s[i] + b[i];
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = s[i] + dpp_right<false>(b[i]);
}
//   a / 9 for N values, stage by stage (see div9)
template <int N>
__device__ __forceinline__ void div9_batch(float (&a)[N]) {
    const float y = 0x1.c71c72p-4f;
    float q0[N], r[N];
#pragma unroll
    for (int i = 0; i < N; ++i) q0[i] = a[i] * y;
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = fmaf(-9.0f, q0[i], a[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) a[i] = fmaf(r[i], y, q0[i]);
}

// project() for N depth hypotheses of one pixel ray, stage by stage (same operations, same order per hypothesis)
// hit[i]: the bilinear sample of the border mask (ones inside a 2-pixel frame, monorec_model.py:282-284) at the sample position is
// != 0 (:218-219), evaluated as logic instead of the reference's 4-term FMA chain: all terms are >= 0, so the chain is 0 iff every
// term is, and a term is mask(tap) * wy * wx with wy in {s, n}, wx in {e, w} - the weights of a tap inside the frame (coordinate
// >= 2, hence sx, sy >= 1) are 0 or >= 2^-24, their product never underflows.  Hence hit = ((top in frame & s != 0) | (bottom in
// frame & n != 0)) & ((left in frame & e != 0) | (right in frame & w != 0)) - exactly the reference's result, ~12 instead of ~25
// VALU instructions.
// FD (compile time - a branch inside the marching step, even a uniform one, splits the basic block and the DPP shifts stop folding
// into their adds): both constant divisions by the exact 3-instruction sequence (the host checked both divisors), else IEEE.
template <int N, bool FD>
__device__ __forceinline__ void project_batch(Sample (&o)[N], bool (&hit)[N], float r0, float r1, float r2, const float (&depth)[N], const float* P,
                                              int H, int W, const CvArgs& a) {
    float X0[N], X1[N], X2[N], px[N], py[N], pz[N], u[N], v[N];
#pragma unroll
    for (int i = 0; i < N; ++i) { X0[i] = depth[i] * r0; X1[i] = depth[i] * r1; X2[i] = depth[i] * r2; }
#pragma unroll
    for (int i = 0; i < N; ++i) { px[i] = P[0] * X0[i]; py[i] = P[4] * X0[i]; pz[i] = P[8] * X0[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) { px[i] = fmaf(P[1], X1[i], px[i]); py[i] = fmaf(P[5], X1[i], py[i]); pz[i] = fmaf(P[9], X1[i], pz[i]); }
#pragma unroll
    for (int i = 0; i < N; ++i) { px[i] = fmaf(P[2], X2[i], px[i]); py[i] = fmaf(P[6], X2[i], py[i]); pz[i] = fmaf(P[10], X2[i], pz[i]); }
#pragma unroll
    for (int i = 0; i < N; ++i) { px[i] = fmaf(P[3], 1.0f, px[i]); py[i] = fmaf(P[7], 1.0f, py[i]); pz[i] = fmaf(P[11], 1.0f, pz[i]); }
#pragma unroll
    for (int i = 0; i < N; ++i) pz[i] = pz[i] + 1e-7f;                  // layers.py:66
#pragma unroll
    for (int i = 0; i < N; ++i) { u[i] = px[i] / pz[i]; v[i] = py[i] / pz[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) { u[i] = div_const(u[i], a.wm1, a.rwm1, FD ? 1 : 0); v[i] = div_const(v[i], a.hm1, a.rhm1, FD ? 1 : 0); }   // layers.py:67-68
#pragma unroll
    for (int i = 0; i < N; ++i) {                                       // layers.py:69 + clamp(-2,2) monorec_model.py:208
        u[i] = fminf(fmaxf((u[i] - 0.5f) * 2.0f, -2.0f), 2.0f);
        v[i] = fminf(fmaxf((v[i] - 0.5f) * 2.0f, -2.0f), 2.0f);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {                                       // grid_sample(align_corners=False), see project()
        const float sx = fmaf(u[i] + 1.0f, (float)W * 0.5f, -0.5f);
        const float sy = fmaf(v[i] + 1.0f, (float)H * 0.5f, -0.5f);
        const float fx = floorf(sx), fy = floorf(sy);
        const float w = sx - fx, e = 1.0f - w, n = sy - fy, s_ = 1.0f - n;
        o[i].x0 = (int)fx;
        o[i].y0 = (int)fy;
        o[i].nw = s_ * e; o[i].ne = s_ * w; o[i].sw = n * e; o[i].se = n * w;
        const unsigned fw = (unsigned)(W - 4), fh = (unsigned)(H - 4);          // columns / rows inside the 2-pixel frame
        const bool lft = (unsigned)(o[i].x0 - 2) < fw && e != 0.f, rgt = (unsigned)(o[i].x0 - 1) < fw && w != 0.f;
        const bool top = (unsigned)(o[i].y0 - 2) < fh && s_ != 0.f, bot = (unsigned)(o[i].y0 - 1) < fh && n != 0.f;
        hit[i] = (lft || rgt) && (top || bot);
    }
}

// bilinear() for N planes x 3 channels: all 12 N gathers first, then the N * 3 FMA chains level by level
template <int N>
__device__ __forceinline__ void bilinear_batch(float (&out)[N * 3], __amdgpu_buffer_rsrc_t img, int plane_bytes, const Taps (&t)[N],
                                               const Sample (&sp)[N]) {
    float a[N * 3], b[N * 3], c[N * 3], d[N * 3];
#pragma unroll
    for (int i = 0; i < N * 3; ++i) {
        const int so = (i % 3) * plane_bytes;
        a[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t[i / 3].a, so, 0));
        b[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t[i / 3].b, so, 0));
        c[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t[i / 3].c, so, 0));
        d[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t[i / 3].d, so, 0));
    }
#pragma unroll
    for (int i = 0; i < N * 3; ++i) out[i] = a[i] * sp[i / 3].nw;
#pragma unroll
    for (int i = 0; i < N * 3; ++i) out[i] = fmaf(b[i], sp[i / 3].ne, out[i]);
#pragma unroll
    for (int i = 0; i < N * 3; ++i) out[i] = fmaf(c[i], sp[i / 3].sw, out[i]);
#pragma unroll
    for (int i = 0; i < N * 3; ++i) out[i] = fmaf(d[i], sp[i / 3].se, out[i]);
}

// The same in two halves for the prefetching marching step: the 12 N gathers of a row are issued one step ahead of their use.
template <int N>
struct Gathered {
    float a[N * 3], b[N * 3], c[N * 3], d[N * 3];     // the four taps of every (plane, channel)
    float nw[N], ne[N], sw[N], se[N];                 // bilinear weights
    bool hit[N];                                      // border-mask sample != 0
    float K[3];                                       // keyframe pixel + 0.5
    float kmu[3], ksg[3];                             // keyframe window statistics of SSIM position (r - 1, x) (prepass)
};

template <int N>
__device__ __forceinline__ void bilinear_issue(Gathered<N>& g, __amdgpu_buffer_rsrc_t img, int plane_bytes, const Taps (&t)[N]) {
#pragma unroll
    for (int i = 0; i < N * 3; ++i) {
        const int so = (i % 3) * plane_bytes;
        g.a[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t[i / 3].a, so, 0));
        g.b[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t[i / 3].b, so, 0));
        g.c[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t[i / 3].c, so, 0));
        g.d[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(img, t[i / 3].d, so, 0));
    }
}

template <int N>
__device__ __forceinline__ void bilinear_combine(float (&out)[N * 3], const Gathered<N>& g) {
#pragma unroll
    for (int i = 0; i < N * 3; ++i) out[i] = g.a[i] * g.nw[i / 3];
#pragma unroll
    for (int i = 0; i < N * 3; ++i) out[i] = fmaf(g.b[i], g.ne[i / 3], out[i]);
#pragma unroll
    for (int i = 0; i < N * 3; ++i) out[i] = fmaf(g.c[i], g.sw[i / 3], out[i]);
#pragma unroll
    for (int i = 0; i < N * 3; ++i) out[i] = fmaf(g.d[i], g.se[i / 3], out[i]);
}

struct MarchGeom {
    int strips, pitch, TY, ysegs, npairs;
};

// Quantities a wave tracks per image row (one register each): for plane u, channel ch: x, x^2, x*k at (u*3 + ch)*3 + {0,1,2};
// and - unless the keyframe statistics come from the prepass (KFS) - keyframe k, k^2 of channel ch at DP*9 + ch*2 + {0,1}.
// MarchRow holds the raw values of a row or (as `top`) the horizontal sums ((l + c) + r) of the row two steps back; e = the
// channel-weighted SSIM distance.
template <int DP, bool KFS>
struct MarchRow {
    float q[DP * 9 + (KFS ? 0 : 6)];
    float e[DP];
};

template <int DP>
struct MarchCtx {
    const CvArgs& a;
    const float* kimg;
    const float* Ki;
    const float* P;
    __amdgpu_buffer_rsrc_t img;
    float kix[3];
    float depth[DP];
    const float* pixd;      // per-pixel depths of plane d0 of this sample, or null
    const float* kstats;    // prepass output of this sample: 3 planes of the keyframe's 3x3 mean, 3 of its variance (KFS), or null
    __amdgpu_buffer_rsrc_t outr;   // raw sad planes d0 .. d0 + DP - 1 of frame f, sample b
    int cx, vx, y0, y1;
    bool col_in, out_lane;
    int planes;             // planes of this wave that exist: DP, or 1 for the last wave of an odd D
};

// One marching step: warp virtual row r into `cur`, emit the SSIM row r - 1 (windows over top / mid / cur) as cur.e, emit the
// sad of output row r - 2 (box over the e rows), then turn `mid` into the next top sums.  The caller alternates two MarchRow
// objects as mid / cur, so the raw rows never move between registers.
// First half of a marching step: project virtual row r for DP planes and ISSUE every global load the step needs (12 DP gathers, the
// keyframe pixel, the prepass statistics of SSIM row r - 1).  Nothing here waits for a load.
template <int DP, bool PIXD, bool KFS, bool FD>
__device__ __forceinline__ void march_issue(const MarchCtx<DP>& c, int r, Gathered<DP>& g) {
    const CvArgs& a = c.a;
    const int H = a.H, W = a.W;
    const int HWp = H * W;
    int wr = r < 0 ? -r : (r >= H ? 2 * H - 2 - r : r);
    wr = min(max(wr, 0), H - 1);
    const int pix = wr * W + c.cx;
    float ray[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) ray[i] = fmaf(c.Ki[3 * i + 2], 1.0f, fmaf(c.Ki[3 * i + 1], (float)wr, c.kix[i]));
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) g.K[ch] = c.kimg[ch * HWp + pix] + 0.5f;
    if (KFS) {                                                              // keyframe statistics of SSIM position (r - 1, x): prepass
        const int sp_ = min(max(r - 1, 0), H - 1) * W + c.cx;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { g.kmu[ch] = c.kstats[ch * HWp + sp_]; g.ksg[ch] = c.kstats[(3 + ch) * HWp + sp_]; }
    }
    Sample sp[DP];
    Taps tp[DP];
    float dep[DP];
#pragma unroll
    for (int u = 0; u < DP; ++u) dep[u] = PIXD ? c.pixd[(long long)(u < c.planes ? u : 0) * HWp + pix] : c.depth[u];
    project_batch<DP, FD>(sp, g.hit, ray[0], ray[1], ray[2], dep, c.P, H, W, a);
#pragma unroll
    for (int u = 0; u < DP; ++u) {
        tp[u] = tap_offsets(sp[u], H, W);
        g.nw[u] = sp[u].nw; g.ne[u] = sp[u].ne; g.sw[u] = sp[u].sw; g.se[u] = sp[u].se;
    }
    bilinear_issue<DP>(g, c.img, HWp * 4, tp);
}

// One marching step: warp virtual row r into `cur`, emit the SSIM row r - 1 (windows over top / mid / cur) as cur.e, emit the
// sad of output row r - 2 (box over the e rows), then turn `mid` into the next top sums.  The caller alternates two MarchRow
// objects as mid / cur, so the raw rows never move between registers.  `g`: what march_issue() gathered for row r.
// RELAXED (the bf16 configuration, whose bar is 1e-2 / 1e-3 on the depth): the 3x3 sums are formed separably - every row keeps its horizontal
// sums ((l + c) + r), a window is (top + mid) + cur - and x / 9 is x * fp32(1/9): 4 instead of 8 additions per quantity and step, 1 instead
// of 3 instructions per division (-18 % instructions).  The volumes move by <= 7e-5 (an ulp of a 9-term sum against C2 = 9e-4 where the
// variance cancels; priced on the oracle) - which is why the fp32 path keeps the reference's order.  The caller rotates THREE row
// buffers (top / mid / cur all hold horizontal sums); validity, projection and gathers are unchanged.
template <int DP, bool PIXD, bool KFS, bool FD, bool RELAXED = false>
__device__ __forceinline__ void march_finish(const MarchCtx<DP>& c, int r, const Gathered<DP>& g, MarchRow<DP, KFS>& top, MarchRow<DP, KFS>& mid,
                                             MarchRow<DP, KFS>& cur, unsigned (&hits)[DP]) {
    constexpr int NQ = DP * 9 + (KFS ? 0 : 6), KQ = DP * 9;
    const CvArgs& a = c.a;
    const int H = a.H, W = a.W;
    const int HWp = H * W;
    const float C1 = 0x1.a36e2ep-14f, C2 = 0x1.d7dbf4p-11f;   // fp32(0.01**2), fp32(0.03**2)  layers.py:116-117
    float K[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) K[ch] = g.K[ch];
#pragma unroll
    for (int u = 0; u < DP; ++u) hits[u] = (hits[u] << 1) | (g.hit[u] ? 1u : 0u);                    // monorec_model.py:218-219
    float xw[DP * 3];
    bilinear_combine<DP>(xw, g);
#pragma unroll
    for (int i = 0; i < DP * 3; ++i) cur.q[i * 3] = xw[i] + 0.5f;
    if (!KFS) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { cur.q[KQ + ch * 2] = K[ch]; cur.q[KQ + ch * 2 + 1] = K[ch] * K[ch]; }
    }
#pragma unroll
    for (int u = 0; u < DP; ++u)
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float x = cur.q[(u * 3 + ch) * 3];
            cur.q[(u * 3 + ch) * 3 + 1] = x * x;
            cur.q[(u * 3 + ch) * 3 + 2] = x * K[ch];
        }
    // ---- 3x3 sums of every quantity over virtual rows r-2, r-1, r; / 9 (AvgPool2d(3,1): row-major sum, then the division) ----
    float s[NQ];
    if (RELAXED) {
        hsum3_batch<NQ>(cur.q, cur.q);                                      // the raw values are not needed again
#pragma unroll
        for (int i = 0; i < NQ; ++i) s[i] = (top.q[i] + mid.q[i]) + cur.q[i];
#pragma unroll
        for (int i = 0; i < NQ; ++i) s[i] = s[i] * 0x1.c71c72p-4f;
    } else {
        win9_batch<NQ>(s, top.q, mid.q, cur.q);
        div9_batch<NQ>(s);
    }
    // ---- SSIM row q = r - 1, channel-weighted -> e (layers.py:119-137, monorec_model.py:133,141) ----------------------------
    const int q = r - 1;
    const bool row_in = q >= 0 && q < H;                                    // wave-uniform
    float kmu[3], kmu2[3], ksg[3];
    if (KFS) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { kmu[ch] = g.kmu[ch]; ksg[ch] = g.ksg[ch]; }
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) kmu2[ch] = kmu[ch] * kmu[ch];
    } else {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) kmu[ch] = s[KQ + ch * 2];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) kmu2[ch] = kmu[ch] * kmu[ch];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) ksg[ch] = s[KQ + ch * 2 + 1] - kmu2[ch];                   // layers.py:130
    }
    constexpr int NS = DP * 3;
    float mu_x_sq[NS], mu_xy[NS], sig_x[NS], sig_xy[NS], sn[NS], sd[NS], sv[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) mu_x_sq[i] = s[i * 3] * s[i * 3];
#pragma unroll
    for (int i = 0; i < NS; ++i) mu_xy[i] = s[i * 3] * kmu[i % 3];
#pragma unroll
    for (int i = 0; i < NS; ++i) sig_x[i] = s[i * 3 + 1] - mu_x_sq[i];
#pragma unroll
    for (int i = 0; i < NS; ++i) sig_xy[i] = s[i * 3 + 2] - mu_xy[i];
#pragma unroll
    for (int i = 0; i < NS; ++i) sn[i] = (2.0f * mu_xy[i] + C1) * (2.0f * sig_xy[i] + C2);                       // layers.py:133
#pragma unroll
    for (int i = 0; i < NS; ++i) sd[i] = (mu_x_sq[i] + kmu2[i % 3] + C1) * (sig_x[i] + ksg[i % 3] + C2);         // layers.py:134
#pragma unroll
    for (int i = 0; i < NS; ++i) sv[i] = fminf(fmaxf((1.0f - ssim_ratio(sn[i], sd[i])) / 2.0f, 0.0f), 1.0f);                // layers.py:137
#pragma unroll
    for (int u = 0; u < DP; ++u) {
        const float ev = fmaf(sv[u * 3 + 2], a.cw[2], fmaf(sv[u * 3 + 1], a.cw[1], sv[u * 3] * a.cw[0]));
        cur.e[u] = (row_in && c.col_in) ? ev : 0.f;                         // zero padding of the 3x3 box (:247)
    }
    // ---- 3x3 box over e rows r-3, r-2, r-1 -> sad of output row y = r - 2 ---------------------------------------------
    const int y = r - 2;
    {   // no branch: a row outside the segment / a halo lane stores through an out-of-range offset (dropped by the descriptor) - one
        // basic block per step keeps every DPP shift next to the add it folds into
        float sad[DP];
        if (RELAXED) {
            hsum3_batch<DP>(cur.e, cur.e);
#pragma unroll
            for (int u = 0; u < DP; ++u) sad[u] = (top.e[u] + mid.e[u]) + cur.e[u];
        } else {
            win9_batch<DP>(sad, top.e, mid.e, cur.e);
        }
        const bool st = (y >= c.y0) & (y < c.y1) & c.out_lane;
        const int voff = st ? (y * W + c.vx) * 4 : -1;
#pragma unroll
        for (int u = 0; u < DP; ++u) {
            if (!(hits[u] & 4u)) sad[u] = -sad[u];                          // sad >= 0: the sign bit is free (-0.0 keeps it)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(sad[u]), c.outr, voff, u * HWp * 4, 0);
        }
    }
    // ---- the middle row becomes the top row of the next step (as horizontal sums) -----------------------------------------
    if (!RELAXED) {
        hsum3_batch<NQ>(top.q, mid.q);
        hsum3_batch<DP>(top.e, mid.e);
    }
}

// issue + finish back to back: the step as rounds 2-3 ran it (every wave waits out its own gathers; the other waves of the SIMD cover)
template <int DP, bool PIXD, bool KFS, bool FD, bool RELAXED = false>
__device__ __forceinline__ void march_step(const MarchCtx<DP>& c, int r, MarchRow<DP, KFS>& top, MarchRow<DP, KFS>& mid, MarchRow<DP, KFS>& cur,
                                           unsigned (&hits)[DP]) {
    Gathered<DP> g;
    march_issue<DP, PIXD, KFS, FD>(c, r, g);
    march_finish<DP, PIXD, KFS, FD, RELAXED>(c, r, g, top, mid, cur, hits);
}

// Keyframe statistics of the SSIM windows, once per keyframe instead of once per (frame, plane pair, row) in every wave: 3x3
// reflection-padded mean and variance of keyframe + 0.5 per channel (layers.py:120-130; row-major sum / 9 like AvgPool2d).
// Written into planes 0..5 of the sample's (not yet written) cost-volume buffer, which cv_fuse overwrites afterwards.
__global__ __launch_bounds__(256) void cv_kf_stats_kernel(const CvArgs a) {
    const int HWp = a.H * a.W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (p >= HWp) return;
    const int y = p / a.W, x = p - y * a.W;
    const float* kimg = a.keyframe + (long long)b * 3 * HWp;
    float* st = a.cv + (long long)b * a.D * HWp;
    int yy[3], xx[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) { yy[t] = reflect_idx(y + t - 1, a.H) * a.W; xx[t] = reflect_idx(x + t - 1, a.W); }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float k = kimg[c * HWp + yy[dy] + xx[dx]] + 0.5f;
                const float kk = k * k;
                if (dy == 0 && dx == 0) { s1 = k; s2 = kk; } else { s1 = s1 + k; s2 = s2 + kk; }
            }
        const float mu = div9(s1);
        st[c * HWp + p] = mu;
        st[(3 + c) * HWp + p] = div9(s2) - mu * mu;
    }
}

// (Software prefetch - the gathers of row r + 1 issued before the arithmetic of row r, two Gathered sets alternating - was measured in
// tools/sessions/r04_s12.sh: 120.3 -> 119.7 us at c2 with one plane per wave (99 instead of 63 registers), slower at 512x1024, and at
// two planes per wave the second set does not fit the register file (256 VGPRs).  The other waves of the SIMD cover the gathers.)
template <int DP, bool PIXD, bool KFS, bool FD, bool RELAXED = false>
__global__ __launch_bounds__(256) void cv_sad_march_kernel(const CvArgs a, const MarchGeom g) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int H = a.H, W = a.W, D = a.D;
    const int HWp = H * W;
    const int strip = blockIdx.x % g.strips, yseg = blockIdx.x / g.strips;
    const int gy = (g.npairs + 3) >> 2;                       // groups of 4 plane sets (one per wave) per frame
    const int f = blockIdx.y / gy;
    const int pi = (blockIdx.y - f * gy) * 4 + wave;          // plane set of this wave
    const int b = blockIdx.z;
    if (pi >= g.npairs) return;
    const int d0 = pi * DP;

    const int vx = strip * g.pitch + lane - 2;                // virtual column of this lane
    int cx = vx < 0 ? -vx : (vx >= W ? 2 * W - 2 - vx : vx);  // reflected (layers.py:112) ...
    cx = min(max(cx, 0), W - 1);                              // ... and kept inside the image for the lanes nothing reads
    const int y0 = yseg * g.TY;
    MarchCtx<DP> c = {a,
                      a.keyframe + (long long)b * 3 * HWp,
                      a.kinv + b * 9,
                      a.proj + ((long long)b * a.F + f) * 12,
                      __builtin_amdgcn_make_buffer_rsrc((void*)(a.frames[f] + (long long)b * 3 * HWp), 0, 3 * HWp * 4, 0x00020000),
                      {0.f, 0.f, 0.f}, {}, PIXD ? a.pix_depths + ((long long)b * D + d0) * HWp : nullptr,
                      KFS ? a.cv + (long long)b * D * HWp : nullptr,
                      // (odd D: the last wave's second plane repeats the last hypothesis; its stores fall outside the descriptor and are dropped)
                      __builtin_amdgcn_make_buffer_rsrc((void*)(a.sfcv[f] + ((long long)b * D + d0) * HWp), 0, min(DP, D - d0) * HWp * 4, 0x00020000),
                      cx, vx, y0, min(y0 + g.TY, H),
                      vx >= 0 && vx < W,                      // SSIM positions outside the image contribute 0 to the box (:247)
                      lane >= 2 && lane < 2 + g.pitch && vx < W,
                      min(DP, D - d0)};
#pragma unroll
    for (int i = 0; i < 3; ++i) c.kix[i] = c.Ki[3 * i] * (float)cx;   // Kinv[:, 0] * x, the first product of the ray's FMA chain (:199)
#pragma unroll
    for (int u = 0; u < DP; ++u) c.depth[u] = PIXD ? 0.f : a.depths[min(d0 + u, D - 1)];

    MarchRow<DP, KFS> top = {}, rowA = {}, rowB = {};
    unsigned hits[DP];                                        // bit j: border-mask sample of the row warped j steps ago != 0
#pragma unroll
    for (int u = 0; u < DP; ++u) hits[u] = 0u;
    const int r_last = c.y1 + 1;
    if (RELAXED) {                                            // three buffers of horizontal sums rotate as top / mid / cur
        for (int r = y0 - 2; r <= r_last; r += 3) {
            march_step<DP, PIXD, KFS, FD, true>(c, r, top, rowA, rowB, hits);
            if (r + 1 <= r_last) march_step<DP, PIXD, KFS, FD, true>(c, r + 1, rowA, rowB, top, hits);
            if (r + 2 <= r_last) march_step<DP, PIXD, KFS, FD, true>(c, r + 2, rowB, top, rowA, hits);
        }
        return;
    }
    for (int r = y0 - 2; r <= r_last; r += 2) {
        march_step<DP, PIXD, KFS, FD>(c, r, top, rowA, rowB, hits);
        if (r + 1 <= r_last) march_step<DP, PIXD, KFS, FD>(c, r + 1, top, rowB, rowA, hits);
    }
}

// Per-pixel frame fusion with the raw sads of a pixel held in registers: every sad is read once, every output written once
// (cv_fuse_kernel reads the F*D raw values three times - twice from L2).  Same arithmetic, same order.
__device__ __forceinline__ unsigned cv_pack_bf16x2(float a, float b) {      // round to nearest even, a in the low half
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
    return __builtin_bit_cast(unsigned, h);
}

// B8OUT: besides the dense fp32 single-frame volumes (the path's outputs, written exactly as before) every pixel's D values of a frame
// also leave as D / 8 groups of 8 bf16 (16 bytes each) - the layout the bf16 MFMA mode's first mask-encoder layer consumes without
// the register-staged fp32 reads (r04_s5: 170 of that layer's 470 us at 512x1024).  The kernel is VALU-bound; the extra stores are free.
template <int DD, bool B8OUT>
__global__ __launch_bounds__(256) void cv_fuse_reg_kernel(const CvArgs a) {
    const int HWp = a.H * a.W;
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int py = p / a.W, px = p - py * a.W;
    const bool border = py >= a.border && py < a.H - a.border && px >= a.border && px < a.W - a.border;     // mask_to_warp[0], :219
    // plane d of a volume = buffer offset p*4 (per lane) + d*HW*4 (scalar): no 64-bit per-plane addresses in VGPRs; lanes beyond
    // the image read 0 / write nothing through the descriptor's range check
    const int voff = p < HWp ? p * 4 : -1;
    const int voff_sf = (B8OUT && a.lean) ? -1 : voff;     // lean: the fp32 single-frame stores fall outside the descriptor's range - dropped by the hardware, no branch
    const int vol_bytes = DD * HWp * 4;
    float num[DD];
    float wsum = 0.f;
    for (int f = 0; f < a.F; ++f) {
        const __amdgpu_buffer_rsrc_t sf = __builtin_amdgcn_make_buffer_rsrc((void*)(a.sfcv[f] + (long long)b * DD * HWp), 0, vol_bytes, 0x00020000);
        float v[DD];
#pragma unroll
        for (int d = 0; d < DD; ++d) v[d] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(sf, voff, d * HWp * 4, 0));
        bool valid = border;
        float smin = INFINITY;
#pragma unroll
        for (int d = 0; d < DD; ++d) {
            valid = valid && !(__float_as_uint(v[d]) & 0x80000000u);
            v[d] = fabsf(v[d]);
            smin = fminf(smin, v[d]);
        }
        float se = 0.f;
#pragma unroll
        for (int d = 0; d < DD; ++d) {
            const float df = v[d] - smin;
            const float ev = expf(-a.alpha * (df * df));                 // :257
            se = d == 0 ? ev : se + ev;
        }
        const float vm = valid ? 1.f : 0.f;
        float w = 1.0f - a.inv_dm1 * (se - 1.0f);                        // :258
        w = w * vm;                                                      // :260
        wsum = f == 0 ? w : wsum + w;                                    // :264
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t* b8 = B8OUT ? (u32x4_t*)a.sfcv_b8[f] + (long long)b * (DD / 8) * HWp : nullptr;
#pragma unroll
        for (int d0 = 0; d0 < DD; d0 += 8) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = d0 + j;
                o[j] = (1.0f - v[d] * 2.0f) * vm;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[j]), sf, voff_sf, d * HWp * 4, 0);   // :251
                const float t = v[d] * w;                                // :262
                num[d] = f == 0 ? t : num[d] + t;
            }
            if (B8OUT && p < HWp)
                b8[(long long)(d0 / 8) * HWp + p] = (u32x4_t){cv_pack_bf16x2(o[0], o[1]), cv_pack_bf16x2(o[2], o[3]), cv_pack_bf16x2(o[4], o[5]), cv_pack_bf16x2(o[6], o[7])};
        }
    }
    const bool nz = wsum != 0.f;
    const __amdgpu_buffer_rsrc_t cvr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.cv + (long long)b * DD * HWp), 0, vol_bytes, 0x00020000);
#pragma unroll
    for (int d = 0; d < DD; ++d)
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(nz ? 1.0f - 2.0f * (num[d] / wsum) : 0.f), cvr, voff, d * HWp * 4, 0);   // :266-269
}

// ---- generic patch size (cv_patch_size != 3; monorec_model.py:138-142,247) -----------------------------------------------
// P x P zero-padded box of the photometric term instead of 3x3, border radius P / 2 + 1.  Rarely used (no reference config sets
// it), so this variant trades speed for simplicity: 32x8 tile, one depth plane per iteration, every stage a strided loop over its
// tile (warped planes on tile + R + 1, photometric term on tile + R, R = P / 2), keyframe SSIM statistics kept in LDS.
// Same conventions as cv_sad_kernel for the raw sad planes (validity / per-plane flag in the sign bit).
__device__ __forceinline__ bool mask_hit_r(const Sample& sp, int H, int W, int br) {
    const float ml = (sp.x0 >= br && sp.x0 < W - br) ? 1.f : 0.f, mr = (sp.x0 + 1 >= br && sp.x0 + 1 < W - br) ? 1.f : 0.f;
    const float mt = (sp.y0 >= br && sp.y0 < H - br) ? 1.f : 0.f, mb = (sp.y0 + 1 >= br && sp.y0 + 1 < H - br) ? 1.f : 0.f;
    const float m = fmaf(mr * mb, sp.se, fmaf(ml * mb, sp.sw, fmaf(mr * mt, sp.ne, (ml * mt) * sp.nw)));
    return m != 0.f;
}

template <int MODE, int OPT>
__global__ __launch_bounds__(256) void cv_sad_patch_kernel(const CvArgs a, const int R) {
    constexpr bool PIXD = OPT & 1, PFLAG = (OPT & 2) != 0;
    constexpr int TX = 32, TY = 8, NT = TX * TY;
    const int HALO = R + 1;
    const int HX = TX + 2 * HALO, HY = TY + 2 * HALO;   // warped / keyframe tile
    const int SX = TX + 2 * R, SY = TY + 2 * R;         // photometric-term tile
    extern __shared__ float patch_lds[];
    float* kf = patch_lds;                  // 3 * HY * HX   keyframe + 0.5
    float* wr = kf + 3 * HY * HX;           // 3 * HY * HX   warped + 0.5
    float* es = wr + 3 * HY * HX;           // SY * SX       channel-weighted photometric term
    float* kmu = es + SY * SX;              // 3 * SY * SX   keyframe 3x3 mean
    float* ksg = kmu + 3 * SY * SX;         // 3 * SY * SX   keyframe 3x3 variance

    const int tid = threadIdx.x;
    const int H = a.H, W = a.W, D = a.D;
    const int b = blockIdx.z;
    const int f = blockIdx.y / a.nchunk, chunk = blockIdx.y % a.nchunk;
    const int d_lo = chunk * a.dchunk, d_hi = min(D, d_lo + a.dchunk);
    const int ty0 = (blockIdx.x / a.tiles_x) * TY, tx0 = (blockIdx.x % a.tiles_x) * TX;
    const int HWp = H * W;
    const float* kimg = a.keyframe + (long long)b * 3 * HWp;

    for (int i = tid; i < 3 * HY * HX; i += NT) {
        const int c = i / (HY * HX), r = i % (HY * HX);
        const int gy = ty0 - HALO + r / HX, gx = tx0 - HALO + r % HX;
        float v = 0.f;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = kimg[c * HWp + gy * W + gx] + 0.5f;
        kf[i] = v;
    }
    __syncthreads();
    if (MODE == 1 || MODE == 2) {
        for (int i = tid; i < SX * SY; i += NT) {
            const int qy = ty0 - R + i / SX, qx = tx0 - R + i % SX;
            const bool in = qy >= 0 && qy < H && qx >= 0 && qx < W;
            for (int c = 0; c < 3; ++c) {
                float s1 = 0.f, s2 = 0.f;
                if (in) {
                    bool first = true;
                    for (int dy = -1; dy <= 1; ++dy)
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int ly = reflect_idx(qy + dy, H) - (ty0 - HALO), lx = reflect_idx(qx + dx, W) - (tx0 - HALO);
                            const float k = kf[(c * HY + ly) * HX + lx];
                            const float kk = k * k;
                            if (first) { s1 = k; s2 = kk; first = false; } else { s1 = s1 + k; s2 = s2 + kk; }
                        }
                }
                const float mu = s1 / 9.0f;
                kmu[c * SY * SX + i] = mu;
                ksg[c * SY * SX + i] = s2 / 9.0f - mu * mu;
            }
        }
    }

    const int oly = tid / TX, olx = tid % TX;
    const int opy = ty0 + oly, opx = tx0 + olx;
    const bool own_in = opy < H && opx < W;
    const float* Ki = a.kinv + b * 9;
    const float C1 = 0x1.a36e2ep-14f, C2 = 0x1.d7dbf4p-11f;
    const float* P = a.proj + ((long long)b * a.F + f) * 12;
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc((void*)(a.frames[f] + (long long)b * 3 * HWp), 0, 3 * HWp * 4, 0x00020000);
    float* sad_out = a.sfcv[f] + (long long)b * D * HWp + opy * W + opx;
    bool hit_all = true;
    float kraw[3] = {0.f, 0.f, 0.f};
    if (PFLAG && own_in) {
#pragma unroll
        for (int c = 0; c < 3; ++c) kraw[c] = kimg[c * HWp + opy * W + opx];
    }

    for (int d = d_lo; d < d_hi; ++d) {
        const float plane_depth = PIXD ? 0.f : a.depths[d];
        const float* pd = PIXD ? a.pix_depths + ((long long)b * D + d) * HWp : nullptr;
        bool pflag = true;
        // ---- (a) warp every in-image position of the tile + halo; the thread of an interior pixel also keeps its flags ----
        for (int i = tid; i < HX * HY; i += NT) {
            const int ly = i / HX, lx = i % HX;
            const int gy = ty0 - HALO + ly, gx = tx0 - HALO + lx;
            if (gy < 0 || gy >= H || gx < 0 || gx >= W) continue;
            const bool interior = ly >= HALO && ly < HALO + TY && lx >= HALO && lx < HALO + TX;
            if (interior) continue;                                     // interior pixels: below, by their own thread
            float ray[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) ray[k] = fmaf(Ki[3 * k + 2], 1.0f, fmaf(Ki[3 * k + 1], (float)gy, Ki[3 * k] * (float)gx));
            const Sample sp = project(ray[0], ray[1], ray[2], PIXD ? pd[gy * W + gx] : plane_depth, P, H, W, a);
            const Taps tp = tap_offsets(sp, H, W);
#pragma unroll
            for (int c = 0; c < 3; ++c) wr[(c * HY + ly) * HX + lx] = bilinear(img, c * HWp * 4, tp, sp) + 0.5f;
        }
        if (own_in) {
            float ray[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) ray[k] = fmaf(Ki[3 * k + 2], 1.0f, fmaf(Ki[3 * k + 1], (float)opy, Ki[3 * k] * (float)opx));
            const Sample sp = project(ray[0], ray[1], ray[2], PIXD ? pd[opy * W + opx] : plane_depth, P, H, W, a);
            hit_all = hit_all && mask_hit_r(sp, H, W, a.border);          // monorec_model.py:218-219
            const Taps tp = tap_offsets(sp, H, W);
            bool any_nz = false, all_eq = true;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float wv = bilinear(img, c * HWp * 4, tp, sp);
                if (PFLAG) { any_nz = any_nz || wv != 0.f; all_eq = all_eq && wv == kraw[c]; }
                wr[(c * HY + oly + HALO) * HX + olx + HALO] = wv + 0.5f;
            }
            pflag = any_nz || all_eq;                                    // :253
        }
        __syncthreads();
        // ---- (b) photometric term on tile + R ------------------------------------------------------------------
        for (int i = tid; i < SX * SY; i += NT) {
            const int sy = i / SX, sx = i % SX;
            const int qy = ty0 - R + sy, qx = tx0 - R + sx;
            float e = 0.f;                                              // zero padding of the P x P box (:247)
            if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
                int lyy[3], lxx[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    lyy[t] = reflect_idx(qy + t - 1, H) - (ty0 - HALO);
                    lxx[t] = reflect_idx(qx + t - 1, W) - (tx0 - HALO);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float sv = 0.f;
                    if (MODE == 1 || MODE == 2) {
                        float sx1 = 0.f, sx2 = 0.f, sxy = 0.f;
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                            for (int dx = 0; dx < 3; ++dx) {
                                const int li = (c * HY + lyy[dy]) * HX + lxx[dx];
                                const float x = wr[li], k = kf[li];
                                const float xx = x * x, xk = x * k;
                                if (dy == 0 && dx == 0) { sx1 = x; sx2 = xx; sxy = xk; }
                                else { sx1 = sx1 + x; sx2 = sx2 + xx; sxy = sxy + xk; }
                            }
                        const float mu_x = div9(sx1), mu_y = kmu[c * SY * SX + i];
                        const float mu_x_sq = mu_x * mu_x, mu_y_sq = mu_y * mu_y, mu_xy = mu_x * mu_y;
                        const float sig_x = div9(sx2) - mu_x_sq;
                        const float sig_xy = div9(sxy) - mu_xy;
                        const float sn = (2.0f * mu_xy + C1) * (2.0f * sig_xy + C2);                      // layers.py:133
                        const float sd = (mu_x_sq + mu_y_sq + C1) * (sig_x + ksg[c * SY * SX + i] + C2);  // layers.py:134
                        sv = fminf(fmaxf((1.0f - ssim_ratio(sn, sd)) / 2.0f, 0.0f), 1.0f);                           // layers.py:137
                    }
                    if (MODE == 0 || MODE == 2) {
                        const int li = (c * HY + sy + 1) * HX + sx + 1;
                        const float ad = fabsf(wr[li] - kf[li]);
                        sv = MODE == 0 ? ad : 0.85f * sv + 0.15f * ad;
                    }
                    if (MODE == 3) {
                        float acc = 0.f;
                        bool first = true;
                        for (int dy = -1; dy <= 1; ++dy)
                            for (int dx = -1; dx <= 1; ++dx) {
                                if (qy + dy < 0 || qy + dy >= H || qx + dx < 0 || qx + dx >= W) continue;   // zero padding
                                const int li = (c * HY + sy + 1 + dy) * HX + sx + 1 + dx;
                                const float ad = fabsf(wr[li] - kf[li]);
                                acc = first ? ad : acc + ad;
                                first = false;
                            }
                        sv = div9(acc);
                    }
                    e = (c == 0) ? sv * a.cw[0] : fmaf(sv, a.cw[c], e);
                }
            }
            es[i] = e;
        }
        __syncthreads();
        // ---- (c) P x P box sum -> sad ----------------------------------------------------------------------------
        if (own_in) {
            float s = 0.f;
            for (int dy = 0; dy <= 2 * R; ++dy)
                for (int dx = 0; dx <= 2 * R; ++dx) {
                    const float v = es[(oly + dy) * SX + olx + dx];
                    s = (dy == 0 && dx == 0) ? v : s + v;
                }
            if (PFLAG) { if (!pflag) s = -s; }
            else if (d == d_hi - 1 && !hit_all) s = -s;
            sad_out[(long long)d * HWp] = s;
        }
    }
    if (PFLAG && own_in && !hit_all)
        atomicAnd((unsigned int*)a.cv + ((long long)b * D + f) * HWp + opy * W + opx, 0u);
}

// Per-pixel frame fusion (monorec_model.py:251-269) over the raw sad values kernel A left in the sfcv buffers.
template <bool PFLAG>
__global__ __launch_bounds__(256) void cv_fuse_kernel(const CvArgs a) {
    const int HWp = a.H * a.W;
    const int D = a.D;
    const long long total = (long long)a.B * HWp;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int b = (int)(i / HWp), p = (int)(i % HWp);
        const int py = p / a.W, px = p % a.W;
        const bool border = py >= a.border && py < a.H - a.border && px >= a.border && px < a.W - a.border;     // mask_to_warp[0], :219
        float wgt[MR_MAX_FRAMES], vmask[MR_MAX_FRAMES];
        float wsum = 0.f;
#pragma unroll
        for (int f = 0; f < MR_MAX_FRAMES; ++f) {
            wgt[f] = 0.f; vmask[f] = 0.f;
            if (f < a.F) {
                const float* sf = a.sfcv[f] + (long long)b * D * HWp + p;
                bool valid = border;
                if (PFLAG) valid = valid && ((const unsigned int*)a.cv)[((long long)b * D + f) * HWp + p] != 0u;
                float smin = INFINITY;
#pragma unroll 8
                for (int d = 0; d < D; ++d) {
                    const float v = sf[(long long)d * HWp];
                    if (!PFLAG) valid = valid && !(__float_as_uint(v) & 0x80000000u);
                    smin = fminf(smin, fabsf(v));
                }
                float se = 0.f;
#pragma unroll 8
                for (int d = 0; d < D; ++d) {
                    const float df = fabsf(sf[(long long)d * HWp]) - smin;
                    const float ev = expf(-a.alpha * (df * df));                 // :257
                    se = d == 0 ? ev : se + ev;
                }
                const float vm = valid ? 1.f : 0.f;
                float w = 1.0f - a.inv_dm1 * (se - 1.0f);                        // :258
                w = w * vm;                                                      // :260
                wgt[f] = w; vmask[f] = vm;
                wsum = f == 0 ? w : wsum + w;                                    // :264
            }
        }
        const bool nz = wsum != 0.f;
        float* cvp = a.cv + (long long)b * D * HWp + p;
#pragma unroll 4
        for (int d = 0; d < D; ++d) {
            float num = 0.f;
#pragma unroll
            for (int f = 0; f < MR_MAX_FRAMES; ++f) {
                if (f < a.F) {
                    float* sf = a.sfcv[f] + (long long)b * D * HWp + p + (long long)d * HWp;
                    const float raw = *sf;
                    const float s = fabsf(raw);
                    const float keep = PFLAG ? ((__float_as_uint(raw) & 0x80000000u) ? 0.f : 1.f) : vmask[f];
                    *sf = (1.0f - s * 2.0f) * keep;                              // :251 / :253
                    const float t = s * wgt[f];                                  // :262
                    num = f == 0 ? t : num + t;
                }
            }
            float v = 0.f;                                                       // :269
            if (nz) v = 1.0f - 2.0f * (num / wsum);                              // :266,268
            cvp[(long long)d * HWp] = v;
        }
    }
}

template <int TX, int TY, int MODE, int OPT>
void launch_sad(const CvArgs& k, dim3 grid, hipStream_t stream) {
    hipLaunchKernelGGL((cv_sad_kernel<TX, TY, MODE, OPT>), grid, dim3(TX * TY), 0, stream, k);
}

template <int TX, int TY, int MODE>
void launch_sad_opt(const CvArgs& k, int opt, dim3 grid, hipStream_t stream) {
    switch (opt) {
        case 1: launch_sad<TX, TY, MODE, 1>(k, grid, stream); break;
        case 2: launch_sad<TX, TY, MODE, 2>(k, grid, stream); break;
        case 3: launch_sad<TX, TY, MODE, 3>(k, grid, stream); break;
        default: launch_sad<TX, TY, MODE, 0>(k, grid, stream); break;
    }
}

void launch_fuse(const CvArgs& k, bool plane_flags, bool tiled, hipStream_t stream) {
    const long long total = (long long)k.B * k.H * k.W;
    if (!plane_flags && !tiled && (k.D == 32 || k.D == 48 || k.D == 64)) {
        const dim3 grid((unsigned)(((long long)k.H * k.W + 255) / 256), (unsigned)k.B);
        const bool b8 = k.sfcv_b8[0] != nullptr;
        if (k.D == 32) { if (b8) hipLaunchKernelGGL((cv_fuse_reg_kernel<32, true>), grid, dim3(256), 0, stream, k); else hipLaunchKernelGGL((cv_fuse_reg_kernel<32, false>), grid, dim3(256), 0, stream, k); }
        else if (k.D == 48) { if (b8) hipLaunchKernelGGL((cv_fuse_reg_kernel<48, true>), grid, dim3(256), 0, stream, k); else hipLaunchKernelGGL((cv_fuse_reg_kernel<48, false>), grid, dim3(256), 0, stream, k); }
        else { if (b8) hipLaunchKernelGGL((cv_fuse_reg_kernel<64, true>), grid, dim3(256), 0, stream, k); else hipLaunchKernelGGL((cv_fuse_reg_kernel<64, false>), grid, dim3(256), 0, stream, k); }
        return;
    }
    const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (plane_flags) hipLaunchKernelGGL(cv_fuse_kernel<true>, dim3(blocks), dim3(256), 0, stream, k);
    else hipLaunchKernelGGL(cv_fuse_kernel<false>, dim3(blocks), dim3(256), 0, stream, k);
}

// Marching kernel geometry: strips of <= 60 output columns (equal pitch, so that every strip carries the same load) and row
// segments of TY rows.  A wave is the unit of work (it never migrates), so the makespan is ceil(waves / SIMDs) wave-lengths: TY
// trades the 4 halo rows every segment warps twice against how evenly the waves divide over the chip's 1024 SIMDs
// (c2: TY 32 -> 2304 waves = 3 rounds at 2.25 average; TY 37 -> 2016 waves = 2 full rounds).

MarchGeom march_geometry(const CvArgs& a, int dp) {
    MarchGeom g;
    g.strips = (a.W + 59) / 60;
    g.pitch = (a.W + g.strips - 1) / g.strips;
    g.npairs = (a.D + dp - 1) / dp;
#ifdef MR_TUNING_ENV         // tuning aids: diagnostic library only (python -m monorec_amd.build --timeline); the product reads no environment
    static const int forced = [] { const char* e = getenv("MR_CV_MARCH_TY"); return e ? atoi(e) : 0; }();
#else
    const int forced = 0;
#endif
    int best_ty = 64;
    if (forced >= 4) best_ty = forced;
    else {
        const double simds = 1024.0;
        const long long per_seg = (long long)g.strips * a.F * a.B * g.npairs;
        double best = -1.0;
        for (int ty = 8; ty <= 64; ++ty) {
            const int segs = (a.H + ty - 1) / ty;
            const double waves = (double)per_seg * segs;
            const double rounds = waves / simds;
            const double balance = rounds >= 8.0 ? 1.0 : rounds / (double)(long long)(rounds + 0.999999);
            const double rows = (double)a.H / ((double)segs * (ty + 4));      // useful rows / warped rows
            const double score = balance * rows;
            if (score > best + 1e-9) { best = score; best_ty = ty; }
        }
    }
    g.TY = best_ty;
    g.ysegs = (a.H + best_ty - 1) / best_ty;
    return g;
}

template <int TX, int TY>
int launch_cv(const CvArgs& a, int mode, bool plane_flags, bool tiled, hipStream_t stream) {
    CvArgs k = a;
    if (mode == 1 && !plane_flags && !tiled) {
        // default configuration: LDS-free marching kernel, two depth planes per wave
        // One plane per wave instead of two when two would leave the 1024 SIMDs with fewer than 4 waves each (a wave issues one
        // VALU instruction per ~4.6 cycles on its own, a SIMD takes one per ~1.6 from 4 waves): c2 2016 -> 4032 waves, 143 -> 133 us;
        // no difference once the chip is full (c3).  MR_CV_MARCH_DP=1 / 2 forces either (tuning aid, read once).
#ifdef MR_TUNING_ENV
        static const int dp_env = [] { const char* e = getenv("MR_CV_MARCH_DP"); return e ? atoi(e) : 0; }();
#else
        const int dp_env = 0;
#endif
        bool dp1 = false;
        if (!a.pix_depths && a.D >= 6) {
            const MarchGeom g2 = march_geometry(a, 2);
            const long long waves2 = (long long)g2.strips * g2.ysegs * a.F * a.B * g2.npairs;
            dp1 = dp_env == 1 || (dp_env != 2 && waves2 < 4096);
        }
        const MarchGeom g = march_geometry(a, dp1 ? 1 : 2);
        const dim3 grid((unsigned)(g.strips * g.ysegs), (unsigned)(a.F * ((g.npairs + 3) / 4)), (unsigned)a.B);
#ifdef MR_TUNING_ENV
        static const bool no_prepass = getenv("MR_CV_NO_KF_PREPASS") != nullptr;           // A/B aid
#else
        const bool no_prepass = false;
#endif
        const bool fd = a.fast_w && a.fast_h;  // both constant divisions by the exact 3-instruction sequence (checked on the host)
        const bool kfs = dp1 || (a.D >= 6 && !no_prepass);
        if (kfs)                             // keyframe window statistics once, into planes 0..5 of the cost-volume buffer
            hipLaunchKernelGGL(cv_kf_stats_kernel, dim3((unsigned)((a.H * a.W + 255) / 256), (unsigned)a.B), dim3(256), 0, stream, k);
#define MR_MARCH(DP_, PIXD_, KFS_)                                                                                              \
    do {                                                                                                                        \
        if (fd) hipLaunchKernelGGL((cv_sad_march_kernel<DP_, PIXD_, KFS_, true>), grid, dim3(256), 0, stream, k, g);          \
        else hipLaunchKernelGGL((cv_sad_march_kernel<DP_, PIXD_, KFS_, false>), grid, dim3(256), 0, stream, k, g);                 \
    } while (0)
        if (a.relaxed_sums && fd && kfs && !a.pix_depths) {      // the bf16 configuration (mr_cost_volume_b8_f32): separable sums, see march_finish
            if (dp1) hipLaunchKernelGGL((cv_sad_march_kernel<1, false, true, true, true>), grid, dim3(256), 0, stream, k, g);
            else hipLaunchKernelGGL((cv_sad_march_kernel<2, false, true, true, true>), grid, dim3(256), 0, stream, k, g);
        } else
        if (dp1) MR_MARCH(1, false, true);   // twice the waves, each with one plane: for shapes that leave the SIMDs short of waves
        else if (kfs && a.pix_depths) MR_MARCH(2, true, true);
        else if (kfs) MR_MARCH(2, false, true);
        else if (a.pix_depths) MR_MARCH(2, true, false);
        else MR_MARCH(2, false, false);
#undef MR_MARCH
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return (int)e;
        launch_fuse(k, false, false, stream);
        return (int)hipGetLastError();
    }
    k.tiles_x = (a.W + TX - 1) / TX;
    const int tiles = k.tiles_x * ((a.H + TY - 1) / TY);
    // depth chunks: enough workgroups to put >= 4 on every CU, chunks of an even number of planes
    int nchunk = 1;
    while ((long long)tiles * a.F * a.B * nchunk < 1024 && (a.D / (nchunk * 2)) >= 4 && (a.D % (nchunk * 4)) == 0) nchunk *= 2;
    k.nchunk = nchunk;
    k.dchunk = a.D / nchunk;
    const dim3 grid(tiles, a.F * nchunk, a.B);
    const int opt = (a.pix_depths ? 1 : 0) | (plane_flags ? 2 : 0);
    if (plane_flags) {     // validity words (plane f of every sample of the cost-volume buffer) start as all ones
        hipError_t e = hipMemsetAsync(a.cv, 0xff, (size_t)a.B * a.D * a.H * a.W * sizeof(float), stream);
        if (e != hipSuccess) return (int)e;
    }
    switch (mode) {
        case 0: launch_sad_opt<TX, TY, 0>(k, opt, grid, stream); break;
        case 2: launch_sad_opt<TX, TY, 2>(k, opt, grid, stream); break;
        case 3: launch_sad_opt<TX, TY, 3>(k, opt, grid, stream); break;
        default: launch_sad_opt<TX, TY, 1>(k, opt, grid, stream); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    launch_fuse(k, plane_flags, tiled, stream);
    return (int)hipGetLastError();
}

template <int MODE>
void launch_sad_patch(const CvArgs& k, int opt, int R, dim3 grid, size_t lds, hipStream_t stream) {
    switch (opt) {
        case 1: hipLaunchKernelGGL((cv_sad_patch_kernel<MODE, 1>), grid, dim3(256), lds, stream, k, R); break;
        case 2: hipLaunchKernelGGL((cv_sad_patch_kernel<MODE, 2>), grid, dim3(256), lds, stream, k, R); break;
        case 3: hipLaunchKernelGGL((cv_sad_patch_kernel<MODE, 3>), grid, dim3(256), lds, stream, k, R); break;
        default: hipLaunchKernelGGL((cv_sad_patch_kernel<MODE, 0>), grid, dim3(256), lds, stream, k, R); break;
    }
}

int launch_cv_patch(const CvArgs& a, int mode, bool plane_flags, int R, hipStream_t stream) {
    constexpr int TX = 32, TY = 8;
    CvArgs k = a;
    k.tiles_x = (a.W + TX - 1) / TX;
    const int tiles = k.tiles_x * ((a.H + TY - 1) / TY);
    int nchunk = 1;
    while ((long long)tiles * a.F * a.B * nchunk < 1024 && (a.D / (nchunk * 2)) >= 4 && (a.D % (nchunk * 2)) == 0) nchunk *= 2;
    k.nchunk = nchunk;
    k.dchunk = a.D / nchunk;
    const dim3 grid(tiles, a.F * nchunk, a.B);
    const int opt = (a.pix_depths ? 1 : 0) | (plane_flags ? 2 : 0);
    const int hx = TX + 2 * (R + 1), hy = TY + 2 * (R + 1), sx = TX + 2 * R, sy = TY + 2 * R;
    const size_t lds = sizeof(float) * (size_t)(6 * hy * hx + 7 * sy * sx);
    if (plane_flags) {
        hipError_t e = hipMemsetAsync(a.cv, 0xff, (size_t)a.B * a.D * a.H * a.W * sizeof(float), stream);
        if (e != hipSuccess) return (int)e;
    }
    switch (mode) {
        case 0: launch_sad_patch<0>(k, opt, R, grid, lds, stream); break;
        case 2: launch_sad_patch<2>(k, opt, R, grid, lds, stream); break;
        case 3: launch_sad_patch<3>(k, opt, R, grid, lds, stream); break;
        default: launch_sad_patch<1>(k, opt, R, grid, lds, stream); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    launch_fuse(k, plane_flags, false, stream);
    return (int)hipGetLastError();
}

int cost_volume_entry(const float* keyframe, const float* const* frames, int32_t num_frames,
                      const float* kinv, const float* proj, const float* depths,
                      int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                      float alpha, const float* channel_weights, int32_t use_ssim,
                      const float* pixel_depths, int32_t sfcv_mult_mask, int32_t patch_size, bool tiled,
                      float* cost_volume, float* const* sfcv, void* stream, void* const* sfcv_b8 = nullptr, bool relaxed = false, bool lean = false) {
    if (use_ssim < 0 || use_ssim > 3) return MR_ERR_BAD_ARGUMENT;
    if (patch_size < 1 || patch_size > 7 || !(patch_size & 1)) return MR_ERR_UNSUPPORTED;
    if (!sfcv_mult_mask && num_depths < num_frames) return MR_ERR_UNSUPPORTED;      // validity words live in planes 0..F-1
    if (!keyframe || !frames || !kinv || !proj || (!depths && !pixel_depths) || !cost_volume || !sfcv || !channel_weights)
        return MR_ERR_BAD_ARGUMENT;
    if (num_frames < 1 || num_frames > MR_MAX_FRAMES || batch < 1 || height < 5 || width < 5) return MR_ERR_BAD_ARGUMENT;
    if (num_depths < 2) return MR_ERR_BAD_ARGUMENT;                       // (:258 divides by num_depths - 1)
    CvArgs a;
    a.keyframe = keyframe;
    for (int f = 0; f < MR_MAX_FRAMES; ++f) {
        a.frames[f] = f < num_frames ? frames[f] : nullptr;
        a.sfcv[f] = f < num_frames ? sfcv[f] : nullptr;
        a.sfcv_b8[f] = (sfcv_b8 && f < num_frames) ? sfcv_b8[f] : nullptr;
        if (f < num_frames && (!a.frames[f] || !a.sfcv[f] || (sfcv_b8 && !a.sfcv_b8[f]))) return MR_ERR_BAD_ARGUMENT;
    }
    // the bf16 copy is written by the register-held fusion kernel of the default configuration only
    if (sfcv_b8 && !(patch_size == 3 && sfcv_mult_mask && !tiled && (num_depths == 32 || num_depths == 48 || num_depths == 64))) return MR_ERR_UNSUPPORTED;
    a.kinv = kinv; a.proj = proj; a.depths = depths; a.pix_depths = pixel_depths; a.cv = cost_volume;
    a.F = num_frames; a.B = batch; a.D = num_depths; a.H = height; a.W = width;
    a.tiles_x = 0; a.nchunk = 1; a.dchunk = num_depths;
    a.relaxed_sums = (sfcv_b8 != nullptr || relaxed) ? 1 : 0;
    a.lean = (lean && sfcv_b8 != nullptr) ? 1 : 0;
    a.alpha = alpha;
    for (int c = 0; c < 3; ++c) a.cw[c] = channel_weights[c] / (float)(patch_size * patch_size);      // :141
    a.inv_dm1 = (float)(1.0 / (double)(num_depths - 1));
    a.border = patch_size / 2 + 1;                                                                   // :139
    a.wm1 = (float)(width - 1); a.hm1 = (float)(height - 1);
    a.rwm1 = 1.0f / a.wm1; a.rhm1 = 1.0f / a.hm1;
    a.fast_w = mr_exact_const_division(a.wm1); a.fast_h = mr_exact_const_division(a.hm1);
    if (height < 2 * a.border + 1 || width < 2 * a.border + 1) return MR_ERR_BAD_ARGUMENT;
    if (patch_size == 3) return launch_cv<32, 16>(a, use_ssim, !sfcv_mult_mask, tiled, (hipStream_t)stream);
    return launch_cv_patch(a, use_ssim, !sfcv_mult_mask, patch_size / 2, (hipStream_t)stream);
}

}  // namespace

// 1 when q = fma(r, y, q0), q0 = a * y, r = fma(-d, q0, a), y = fp32(1 / d) equals the correctly rounded fp32 quotient a / d for EVERY
// dividend: all 2^23 mantissas of one binade are tried (the sequence is scale invariant; |y| < 1 excludes overflow), once per divisor
// and process (~40 ms), the verdict is cached.  The cost-volume kernels then divide by W - 1 / H - 1 (layers.py:67-68) in 3
// instructions instead of the ~10 of the IEEE sequence without giving up a bit (Markstein's correction step; every divisor tried so far
// passes - 511, 255 (c2), 1023 (configs[4]), 95, 63 - the check is what makes that a fact per divisor rather than a belief).
extern "C" int mr_exact_const_division(float d) {
    if (!(d > 1.0f) || !(d < 1e30f)) return 0;
    static std::mutex mu;
    static std::map<uint32_t, int> verdict;
    uint32_t key;
    memcpy(&key, &d, 4);
    std::lock_guard<std::mutex> lock(mu);
    const auto it = verdict.find(key);
    if (it != verdict.end()) return it->second;
    const float y = 1.0f / d;
    int ok = 1;
    for (uint32_t m = 0x4b000000u; m < 0x4b800000u && ok; ++m) {       // [2^23, 2^24)
        float a;
        memcpy(&a, &m, 4);
        const float q0 = a * y;
        const float r = fmaf(-d, q0, a);
        const float q = fmaf(r, y, q0);
        if (q != a / d) ok = 0;
    }
    verdict[key] = ok;
    return ok;
}

extern "C" int mr_cost_volume_patch_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                                       const float* kinv, const float* proj, const float* depths,
                                       int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                                       float alpha, const float* channel_weights, int32_t use_ssim,
                                       const float* pixel_depths, int32_t sfcv_mult_mask, int32_t patch_size,
                                       float* cost_volume, float* const* sfcv, void* stream) {
    return cost_volume_entry(keyframe, frames, num_frames, kinv, proj, depths, batch, num_depths, height, width, alpha,
                             channel_weights, use_ssim, pixel_depths, sfcv_mult_mask, patch_size, false, cost_volume, sfcv, stream);
}

extern "C" int mr_cost_volume_tiled_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                                       const float* kinv, const float* proj, const float* depths,
                                       int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                                       float alpha, const float* channel_weights, int32_t use_ssim,
                                       const float* pixel_depths, int32_t sfcv_mult_mask,
                                       float* cost_volume, float* const* sfcv, void* stream) {
    return cost_volume_entry(keyframe, frames, num_frames, kinv, proj, depths, batch, num_depths, height, width, alpha,
                             channel_weights, use_ssim, pixel_depths, sfcv_mult_mask, 3, true, cost_volume, sfcv, stream);
}

extern "C" int mr_cost_volume_mode_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                                       const float* kinv, const float* proj, const float* depths,
                                       int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                                       float alpha, const float* channel_weights, int32_t use_ssim,
                                       const float* pixel_depths, int32_t sfcv_mult_mask,
                                       float* cost_volume, float* const* sfcv, void* stream) {
    return mr_cost_volume_patch_f32(keyframe, frames, num_frames, kinv, proj, depths, batch, num_depths, height, width, alpha,
                                    channel_weights, use_ssim, pixel_depths, sfcv_mult_mask, 3, cost_volume, sfcv, stream);
}

// mr_cost_volume_mode_f32 of the default configuration (3x3 patch, sfcv * mask, 32 / 48 / 64 depth steps) that ALSO writes every
// single-frame volume in the channel-blocked bf16 layout of mr_conv2d_b8: sfcv_b8[f] = (batch, num_depths / 8, height, width, 8) bf16.
extern "C" int mr_cost_volume_b8_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                                     const float* kinv, const float* proj, const float* depths,
                                     int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                                     float alpha, const float* channel_weights, int32_t use_ssim, const float* pixel_depths,
                                     float* cost_volume, float* const* sfcv, void* const* sfcv_b8, void* stream) {
    if (!sfcv_b8) return MR_ERR_BAD_ARGUMENT;
    return cost_volume_entry(keyframe, frames, num_frames, kinv, proj, depths, batch, num_depths, height, width, alpha,
                             channel_weights, use_ssim, pixel_depths, 1, 3, false, cost_volume, sfcv, stream, sfcv_b8);
}

// mr_cost_volume_b8_f32 without the dense fp32 single-frame volumes: `sfcv` is scratch (raw per-frame costs), only the fused volume and the B8
// copies are outputs - for callers of the bf16 configuration that do not hand out `single_frame_cvs` (MonoRecModel(hip_lean_outputs=True)).
extern "C" int mr_cost_volume_b8_lean_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                                          const float* kinv, const float* proj, const float* depths,
                                          int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                                          float alpha, const float* channel_weights, int32_t use_ssim, const float* pixel_depths,
                                          float* cost_volume, float* const* sfcv_scratch, void* const* sfcv_b8, void* stream) {
    if (!sfcv_b8) return MR_ERR_BAD_ARGUMENT;
    return cost_volume_entry(keyframe, frames, num_frames, kinv, proj, depths, batch, num_depths, height, width, alpha,
                             channel_weights, use_ssim, pixel_depths, 1, 3, false, cost_volume, sfcv_scratch, stream, sfcv_b8, false, true);
}

// mr_cost_volume_mode_f32 of the default configuration with the RELAXED window sums of mr_cost_volume_b8_f32 (separable 3x3 sums, x * fp32(1/9))
// and dense fp32 outputs only: the opt-in of the fp32 path (MonoRecModel(hip_cv_separable=True)).  Validity (the zeros of
// the volumes) is exactly that of mr_cost_volume_f32; the values differ by the rounding of another summation order of the same nine terms
// (single-frame volumes <= 1e-4, depth <= 2e-6 on the fixtures: tests/test_gpu_kernels.py).  Per-pixel depths and sizes whose constant
// divisions fail mr_exact_const_division run the exact kernels (the relaxed instantiation exists for the common case only).
extern "C" int mr_cost_volume_relaxed_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                                          const float* kinv, const float* proj, const float* depths,
                                          int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                                          float alpha, const float* channel_weights, int32_t use_ssim, const float* pixel_depths,
                                          float* cost_volume, float* const* sfcv, void* stream) {
    return cost_volume_entry(keyframe, frames, num_frames, kinv, proj, depths, batch, num_depths, height, width, alpha,
                             channel_weights, use_ssim, pixel_depths, 1, 3, false, cost_volume, sfcv, stream, nullptr, true);
}

extern "C" int mr_cost_volume_f32(const float* keyframe, const float* const* frames, int32_t num_frames,
                                  const float* kinv, const float* proj, const float* depths,
                                  int32_t batch, int32_t num_depths, int32_t height, int32_t width,
                                  float alpha, const float* channel_weights,
                                  float* cost_volume, float* const* sfcv, void* stream) {
    return mr_cost_volume_mode_f32(keyframe, frames, num_frames, kinv, proj, depths, batch, num_depths, height, width, alpha,
                                   channel_weights, 1, nullptr, 1, cost_volume, sfcv, stream);
}
