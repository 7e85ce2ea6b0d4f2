// Host harness of the GENERATED Cook-Toom transform code (monorec_amd/csrc/cooktoom_1d.h): the same straight-line functions the
// gfx950 kernel inlines, compiled for the CPU (`__device__` defined away), evaluated in fp32 on seeded data and compared with the
// direct r-tap correlation in double.  Prints one line per form: "m r max_abs_err max_abs_y".  (tests/test_capi_and_host.py)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#define __device__
#define __forceinline__ inline
#include "../../monorec_amd/csrc/cooktoom_1d.h"

static unsigned long long s = 88172645463325252ull;
static double rnd() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0; }

template <int M, int R, typename FI, typename FO>
static void run(const double (*G)[R], FI input, FO output) {
    constexpr int N = M + R - 1;
    double worst = 0, scale = 0;
    for (int trial = 0; trial < 2000; ++trial) {
        float d[N], v[N], mm[N], y[M];
        double g[R];
        for (int i = 0; i < N; ++i) d[i] = (float)rnd();
        for (int j = 0; j < R; ++j) g[j] = (float)rnd();
        input(d, v);
        for (int i = 0; i < N; ++i) {
            double u = 0;
            for (int j = 0; j < R; ++j) u += G[i][j] * g[j];
            mm[i] = (float)u * v[i];
        }
        output(mm, y);
        for (int k = 0; k < M; ++k) {
            double ref = 0;
            for (int j = 0; j < R; ++j) ref += g[j] * (double)d[k + j];
            worst = fmax(worst, fabs(ref - (double)y[k]));
            scale = fmax(scale, fabs(ref));
        }
    }
    printf("%d %d %.3e %.3e\n", M, R, worst, scale);
}

int main() {
    run<4, 3>(CT_G_4_3, ct_input_4_3, ct_output_4_3);
    run<2, 7>(CT_G_2_7, ct_input_2_7, ct_output_2_7);
    run<4, 7>(CT_G_4_7, ct_input_4_7, ct_output_4_7);
    run<4, 4>(CT_G_4_4, ct_input_4_4, ct_output_4_4);
    return 0;
}
