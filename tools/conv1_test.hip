// Development check: conv1 matrix-core kernel (conv1.hip) against the VALU kernel (misc_kernels.hip).
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -I pix2pose_amd/csrc tools/conv1_test.hip pix2pose_amd/csrc/conv1.o pix2pose_amd/csrc/misc_kernels.o -o tools/conv1_test
#include "kernels.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace p2p;
int main()
{
    const int N = 3;
    std::vector<float> x((size_t)N * 128 * 128 * 3), w(147 * 64), sc(64, 1.f), sh(64, 0.f);
    for (auto& v : x) v = (float)rand() / RAND_MAX * 2 - 1;
    for (auto& v : w) v = ((float)rand() / RAND_MAX * 2 - 1) * 0.1f;
    const char* dl = getenv("DELTA");
    int dkh = -1, dkw = 0, dc = 0;
    if (dl) { sscanf(dl, "%d,%d,%d", &dkh, &dkw, &dc); for (auto& v : w) v = 0.f; w[(size_t)((dkh * 7 + dkw) * 3 + dc) * 64 + 5] = 1.f; }
    std::vector<float> panel(conv1_f16x3_panel_floats(), 0.f);
    _Float16* o = reinterpret_cast<_Float16*>(panel.data());
    for (int kh = 0; kh < 7; ++kh) for (int kw = 0; kw < 7; ++kw) for (int c = 0; c < 3; ++c) for (int co = 0; co < 64; ++co) {
        const float v = w[(size_t)((kh * 7 + kw) * 3 + c) * 64 + co];
        const _Float16 hi = (_Float16)v;
        o[conv1_f16x3_panel_index(kh, 0, kw, c, co)] = hi;
        o[conv1_f16x3_panel_index(kh, 1, kw, c, co)] = (_Float16)(v - (float)hi);
    }
    float *dx, *dw, *dp, *dsc, *dsh, *o1, *o2;
    const size_t on = (size_t)N * 64 * 64 * 64;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dw, w.size() * 4); hipMalloc(&dp, panel.size() * 4);
    hipMalloc(&dsc, 256); hipMalloc(&dsh, 256); hipMalloc(&o1, on * 4); hipMalloc(&o2, on * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dp, panel.data(), panel.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dsc, sc.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsh, sh.data(), 256, hipMemcpyHostToDevice);
    hipMemset(o2, 0xFF, on * 4);
    launch_conv_first(dx, N, 128, 128, dw, 7, 2, 3, 64, dsc, dsh, ACT_NONE, 0.3f, o1, 64, 64, 0);
    Conv1Groups G; G.n_groups = 1; G.start[0] = 0; G.start[1] = N; G.w[0] = dp; G.scale[0] = dsc; G.shift[0] = dsh;
    hipError_t e = launch_conv1_f16x3(dx, N, G, ACT_NONE, 0.3f, o2, 0);
    hipDeviceSynchronize();
    printf("launch: %s / %s\n", hipGetErrorString(e), hipGetErrorString(hipGetLastError()));
    std::vector<float> a(on), b(on);
    hipMemcpy(a.data(), o1, on * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, on * 4, hipMemcpyDeviceToHost);
    double md = 0; size_t bad = 0, first = (size_t)-1;
    for (size_t i = 0; i < on; ++i) { double d = std::fabs(a[i] - b[i]); if (!(d < 1e-4)) { ++bad; if (first == (size_t)-1) first = i; } if (d > md) md = d; }
    printf("max diff %.3g, bad %zu of %zu\n", md, bad, on);
    for (size_t i = first, k = 0; first != (size_t)-1 && k < 8 && i < on; i += 1, ++k) {
        size_t co = i % 64, px = (i / 64) % 64, py = (i / 4096) % 64, n = i / (4096 * 64);
        printf("  n %zu y %zu x %zu c %zu: ref %g got %g\n", n, py, px, co, a[i], b[i]);
    }
    if (dl) {
        for (int py : {10, 11}) for (int px : {10, 11, 40}) {
            const float got = b[(((size_t)0 * 64 + py) * 64 + px) * 64 + 5], ref = a[(((size_t)0 * 64 + py) * 64 + px) * 64 + 5];
            printf("out(%d,%d) c5: ref %g got %g; got matches input at:", py, px, ref, got);
            for (size_t i = 0; i < (size_t)128 * 128 * 3; ++i) if (std::fabs(x[i] - got) < 2e-6) printf(" (iy %zu ix %zu ch %zu)", i / 384, (i / 3) % 128, i % 3);
            printf("  expected (iy %d ix %d ch %d)\n", 2 * py - 3 + dkh, 2 * px - 3 + dkw, dc);
        }
        int nz = 0; for (size_t i = 0; i < on; ++i) if (i % 64 != 5 && b[i] != 0.f) ++nz;
        printf("nonzero outputs in other channels: %d\n", nz);
    }
    // which (y, x, c) are wrong: histogram over x, y, c
    int hx[64] = {0}, hy[64] = {0}, hc[64] = {0};
    for (size_t i = 0; i < on; ++i) if (!(std::fabs(a[i] - b[i]) < 1e-4)) { hc[i % 64]++; hx[(i / 64) % 64]++; hy[(i / 4096) % 64]++; }
    printf("bad by x:"); for (int i = 0; i < 64; ++i) printf(" %d", hx[i]); printf("\nbad by y:"); for (int i = 0; i < 64; ++i) printf(" %d", hy[i]);
    printf("\nbad by c:"); for (int i = 0; i < 64; ++i) printf(" %d", hc[i]); printf("\n");
    return 0;
}
