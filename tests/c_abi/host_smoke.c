/* Plain C (C99) consumer of include/monorec_hip.h: proves that the boundary is a C ABI (no C++ / torch types in the
 * signatures) and exercises the entry points that run on the host.  Built and run by tests/test_capi_and_host.py with gcc. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "monorec_hip.h"

int main(void) {
    if (mr_abi_version() != MR_ABI_VERSION) { printf("abi %d != header %d\n", mr_abi_version(), MR_ABI_VERSION); return 1; }
    if (strcmp(mr_error_string(0), "success") != 0 || !strstr(mr_error_string(MR_ERR_LDS_BUDGET), "LDS")) return 2;

    /* weight repack for one 3x3 conv, 20 + 5 concatenated input channels -> 40 output channels, 2 cout blocks per workgroup */
    const int32_t srcs[2] = {20, 5};
    const int cout = 40, cin = 25, kh = 3, kw = 3, mb = 2, ck = 16;
    const size_t n = mr_conv_packed_weight_floats(cout, srcs, 2, kh, kw, mb, ck);
    if (n == 0) return 3;
    float* w = (float*)malloc(sizeof(float) * cout * cin * kh * kw);
    float* packed = (float*)malloc(sizeof(float) * n);
    double sum = 0.0, packed_sum = 0.0;
    for (int i = 0; i < cout * cin * kh * kw; ++i) { w[i] = (float)((i * 37 % 101) - 50) / 64.0f; sum += w[i]; }
    if (mr_conv_pack_weights_f32(w, cout, srcs, 2, kh, kw, mb, ck, packed) != 0) return 4;
    for (size_t i = 0; i < n; ++i) packed_sum += packed[i];
    if (fabs(sum - packed_sum) > 1e-6) { printf("repack lost weights: %g vs %g\n", sum, packed_sum); return 5; }   /* padding is zero */

    /* launch planning without a device: LDS bytes of that conv on a 64x96 map, and the rejection of an oversized chunk */
    mr_conv_desc d;
    memset(&d, 0, sizeof d);
    d.src[0] = d.src[1] = (const float*)16; d.src_channels[0] = 20; d.src_channels[1] = 5; d.num_src = 2;
    d.batch = 1; d.src_h = 64; d.src_w = 96; d.in_mode = MR_IN_DIRECT;
    d.kh = kh; d.kw = kw; d.stride_h = d.stride_w = 1; d.pad_top = d.pad_left = 1;
    d.out_h = 64; d.out_w = 96; d.dst = (float*)16; d.out_channels = cout; d.dst_total_channels = cout;
    d.dst_plane_h = 64; d.dst_plane_w = 96; d.out_step_h = d.out_step_w = 1;
    d.packed_weights = packed; d.num_phases = 1;
    d.cout_blocks_per_wg = mb; d.pixel_blocks_per_wave = 2; d.split_k = 1; d.chunk_channels = ck; d.waves_per_wg = 4;
    const long long lds = (long long)mr_conv2d_lds_bytes(&d);
    if (lds <= 0 || lds > 160 * 1024) { printf("lds %lld\n", lds); return 6; }
    d.chunk_channels = 12;                                     /* not one of 8/16/32/64/128 */
    if (mr_conv2d_lds_bytes(&d) >= 0) return 7;

    /* Pillow-exact resize tables: 1226 -> 512 columns of the KITTI crop */
    const int32_t ks = mr_resample_ksize_bilinear(243, 983, 512);
    if (ks < 2) return 8;
    int32_t* bounds = (int32_t*)malloc(sizeof(int32_t) * 512 * 2);
    int32_t* coeffs = (int32_t*)malloc(sizeof(int32_t) * 512 * ks);
    if (mr_resample_coeffs_bilinear(1226, 243, 983, 512, bounds, coeffs) != 0) return 9;
    for (int x = 0; x < 512; ++x) {
        long long s = 0;
        for (int k = 0; k < bounds[2 * x + 1]; ++k) s += coeffs[x * ks + k];
        if (llabs(s - (1ll << 22)) > ks) { printf("column %d: weights sum to %lld\n", x, s); return 10; }
    }
    printf("c abi ok: %zu packed floats, %lld LDS bytes, resize kernel size %d\n", n, lds, (int)ks);
    free(w); free(packed); free(bounds); free(coeffs);
    return 0;
}
