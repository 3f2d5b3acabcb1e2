// Host side of the generator forward pass: weight packing (Keras layouts -> MFMA implicit-GEMM
// panels, BatchNorm folded into scale/shift), activation workspace, layer sequencing, and the
// C ABI of include/p2p_mi355.h for the network call.
//
// Reference graphs: pix2pose_model/ae_model.py:70-150 (paper), :175-240 (resnet50),
// pix2pose_model/resnet50_mod.py:40-118,200-213.  Call sites replaced: recognition.py:21-26
// (construction + load_weights) and recognition.py:84,129 (generator_train.predict).
#include "model.h"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <thread>

namespace p2p {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return P2P_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

constexpr double BN_EPS = 1e-3;     // Keras BatchNormalization default
constexpr float LEAKY = 0.3f;       // keras.layers.LeakyReLU() default

// ------------------------------------------------------------------------------------------
// weight lookup / packing (host)
// ------------------------------------------------------------------------------------------
struct TensorMap {
    std::unordered_map<std::string, std::pair<const float*, int64_t>> m;
    const float* get(const std::string& name, int64_t numel) const
    {
        auto it = m.find(name);
        if (it == m.end()) {
            set_error("weight tensor '%s' missing", name.c_str());
            return nullptr;
        }
        if (it->second.second != numel) {
            set_error("weight tensor '%s' has %lld elements, expected %lld", name.c_str(),
                      (long long)it->second.second, (long long)numel);
            return nullptr;
        }
        return it->second.first;
    }
};

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static int upload(const std::vector<float>& h, float** d)
{
    HIP_TRY(hipMalloc((void**)d, h.size() * sizeof(float)));
    HIP_TRY(hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return P2P_OK;
}

// scale = gamma / sqrt(var + eps); shift = (bias - mean) * scale + beta  (per output channel)
static int fold_bn(const TensorMap& T, const std::string& name, int C, bool has_bn, std::vector<float>& scale,
                   std::vector<float>& shift)
{
    const float* bias = T.get(name + ".bias", C);
    if (!bias) return P2P_ERR_WEIGHTS;
    if (!has_bn) {
        for (int c = 0; c < C; ++c) { scale.push_back(1.f); shift.push_back(bias[c]); }
        return P2P_OK;
    }
    const float* g = T.get(name + ".gamma", C);
    const float* b = T.get(name + ".beta", C);
    const float* mu = T.get(name + ".mean", C);
    const float* var = T.get(name + ".var", C);
    if (!g || !b || !mu || !var) return P2P_ERR_WEIGHTS;
    for (int c = 0; c < C; ++c) {
        const double s = (double)g[c] / std::sqrt((double)var[c] + BN_EPS);
        scale.push_back((float)s);
        shift.push_back((float)(((double)bias[c] - (double)mu[c]) * s + (double)b[c]));
    }
    return P2P_OK;
}

// IEEE binary16 <-> binary32 on the host (round to nearest even), for the offline weight split
static uint16_t f32_to_f16(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7F800000u ? 0x7E00u : 0x7C00u));   // overflow -> inf, NaN
    if (x < 0x38800000u) {                       // subnormal half (or zero)
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 113 - (int)(x >> 23);
        uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
        const uint32_t q = m >> (shift + 13), rem = m & ((1u << (shift + 13)) - 1), half = 1u << (shift + 12);
        return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1))) ? 1u : 0u)));
    }
    const uint32_t q = (x - 0x38000000u) >> 13, rem = x & 0x1FFFu;
    return (uint16_t)(sign | (q + ((rem > 0x1000u || (rem == 0x1000u && (q & 1))) ? 1u : 0u)));
}
static float f16_to_f32(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31, m = h & 0x3FFu;
    uint32_t x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { float v = (float)m * 5.9604644775390625e-08f; memcpy(&x, &v, 4); x |= sign; }   // m * 2^-24
    } else if (e == 31) x = sign | 0x7F800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// Development switch: P2P_NO_HALO=1 routes every layer through the generic kernels (igemm.hip, the VALU first layer);
// tests/test_halo_gpu.py compares the two paths.
static bool specialised_kernels() { static const bool on = dev_env("P2P_NO_HALO") == nullptr; return on; }

// Launches whose batched kernel would run on at most this many workgroups take the streaming kernel (igemm_stream.hip) instead.
// P2P_STREAM_WGS=0 switches the route off (tests compare the two routes bit for bit).
static int stream_max_wgs()
{
    static const int v = dev_env("P2P_STREAM_WGS") ? atoi(dev_env("P2P_STREAM_WGS")) : 64;
    return specialised_kernels() ? v : 0;
}

static thread_local int g_pack_prec = PREC_F32;   // precision of the model being packed (build_model)

// Model construction is host work over 28 M weights (transposes into GEMM panels, the f16 split): rows are independent, so the big
// loops run on a few threads (0.33 s per object on one thread; a 30-object T-LESS set is built once per process).
template <typename F>
static void parallel_rows(size_t n, size_t work_per_row, F fn)
{
    unsigned nt = std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
    if (n * work_per_row < (size_t)1 << 18 || nt == 1 || n < 2) { fn((size_t)0, n); return; }
    nt = (unsigned)std::min<size_t>(nt, n);
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t a = t * per, b = std::min(n, a + per);
        if (a >= b) break;
        th.emplace_back([=] { fn(a, b); });
    }
    for (auto& x : th) x.join();
}
// Power-of-two pre-scale of ONE OUTPUT CHANNEL's weights for the split: the largest |w| of the row lands in
// [2^13, 2^14) (f16 max is 65504), so the lo parts of all but vanishing weights stay in the f16 normal range.
// Per row, not per layer: in a trained network the rows of one layer differ by orders of magnitude (BatchNorm
// variances spread over decades, and the folded BN scale then multiplies a small row's rounding error back up);
// the epilogue scale is per output channel anyway and carries the inverse.
static float f16x3_row_scale(const float* w, size_t n, size_t stride = 1)
{
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) m = std::max(m, std::fabs(w[i * stride]));
    if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
    int e = 0;
    std::frexp(m, &e);                 // m = f * 2^e, f in [0.5, 1)
    return std::ldexp(1.f, std::min(std::max(14 - e, -14), 24));
}

// fp32 panel [rows][K] -> split-f16 panel of the same byte size: per row and 32-wide K-step the image
// [hi x32 | lo x32], hi = f16(w * s_row), lo = f16(w * s_row - hi), s_row a power of two (row_scale[r]).
// The epilogue scale carries 1/s_row.
static std::vector<float> split_panel(const std::vector<float>& w, int K, std::vector<float>& row_scale)
{
    std::vector<float> out(w.size());
    uint16_t* o = reinterpret_cast<uint16_t*>(out.data());
    const size_t rows = w.size() / K;
    row_scale.assign(rows, 1.f);
    parallel_rows(rows, (size_t)K, [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; ++r) {
            const float sc = row_scale[r] = f16x3_row_scale(w.data() + r * K, (size_t)K);
            for (int k0 = 0; k0 < K; k0 += 32) {
                uint16_t* blk = o + (r * K + k0) * 2;
                for (int k = 0; k < 32; ++k) {
                    const float v = w[r * K + k0 + k] * sc;
                    const uint16_t hi = f32_to_f16(v);
                    blk[k] = hi;
                    blk[32 + k] = f32_to_f16(v - f16_to_f32(hi));
                }
            }
        }
    });
    return out;
}

static int finish_layer(ConvLayer& L, const std::vector<float>& w, const std::vector<float>& scale,
                        const std::vector<float>& shift)
{
    int rc;
    if (L.prec == PREC_F16X3) {
        std::vector<float> rs;
        std::vector<float> panel = split_panel(w, L.K, rs);
        std::vector<float> sc(scale);
        for (size_t c = 0; c < sc.size(); ++c) sc[c] *= 1.f / rs[c];          // exact: a power of two
        if ((rc = upload(panel, &L.w))) return rc;
        if ((rc = upload(sc, &L.scale))) return rc;
    } else {
        if ((rc = upload(w, &L.w))) return rc;
        if ((rc = upload(scale, &L.scale))) return rc;
    }
    if ((rc = upload(shift, &L.shift))) return rc;
    return P2P_OK;
}

// Conv2D kernels (kh,kw,Cin,Cout) of `names` (concatenated along Cout: parallel branches on one
// input, e.g. conv4_1 || conv4_2) -> W[Cout_pad][kh*kw*Cin], tap t = kh*KW+kw at (kh-pad, kw-pad).
static int pack_conv(const TensorMap& T, const std::vector<std::string>& names, int KH, int Cin, int cout_each,
                     int pad, bool bn, ConvLayer& L)
{
    const int nb = (int)names.size();
    L.prec = g_pack_prec;
    L.Cout = cout_each * nb;
    L.ntaps = KH * KH;
    L.K = L.ntaps * Cin;
    if (L.ntaps > IGEMM_MAX_TAPS || Cin % IGEMM_BK) {
        set_error("pack_conv(%s): unsupported shape", names[0].c_str());
        return P2P_ERR_INVALID_ARG;
    }
    std::vector<float> w((size_t)round_up(L.Cout, 128) * L.K, 0.f), scale, shift;
    for (int b = 0; b < nb; ++b) {
        const float* k = T.get(names[b] + ".kernel", (int64_t)KH * KH * Cin * cout_each);
        if (!k) return P2P_ERR_WEIGHTS;
        parallel_rows((size_t)cout_each, (size_t)L.K, [&](size_t c0, size_t c1) {
            for (int t = 0; t < L.ntaps; ++t)
                for (int ci = 0; ci < Cin; ++ci)
                    for (size_t co = c0; co < c1; ++co)
                        w[(size_t)(b * cout_each + co) * L.K + (size_t)t * Cin + ci] = k[((size_t)t * Cin + ci) * cout_each + co];
        });
        int rc = fold_bn(T, names[b], cout_each, bn, scale, shift);
        if (rc) return rc;
    }
    for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < KH; ++kw) {
            L.dy[kh * KH + kw] = (int8_t)(kh - pad);
            L.dx[kh * KH + kw] = (int8_t)(kw - pad);
        }
    return finish_layer(L, w, scale, shift);
}

// Winograd F(4,5) panel of a Conv2D 5x5 stride-1 layer (wino.hip), beside the direct panel pack_conv built: along the row axis
//   U_j[ky][ci][co] = sum_kx G[j][kx] w[ky][kx][ci][co],   G = the Cook-Toom filter matrix at the points {0, 1, -1, 2, -2, 1/2, -1/2, inf}
// computed in double, rounded to fp32, pre-scaled per output channel by a power of two (the epilogue scale carries the inverse, like
// split_panel), split hi / lo and stored in the order the GEMM kernel's waves stream it:
//   [Cout / 64][position 8][Cin / 16][ky 5][fragment: (tile 0 hi, tile 0 lo, tile 1 hi, tile 1 lo)][lane 64][8 halves]
// lane (li, lk) of a fragment = output channel 32 tile + li, input channels 16 slice + 8 lk + (0..7): the A operand of v_mfma_f32_32x32x16_f16.
static const double kWinoG[8][5] = {
    {-1.0, 0.0, 0.0, 0.0, 0.0},
    {-2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9, -2.0 / 9},
    {-2.0 / 9, 2.0 / 9, -2.0 / 9, 2.0 / 9, -2.0 / 9},
    {1.0 / 90, 2.0 / 90, 4.0 / 90, 8.0 / 90, 16.0 / 90},
    {1.0 / 90, -2.0 / 90, 4.0 / 90, -8.0 / 90, 16.0 / 90},
    {32.0 / 45, 16.0 / 45, 8.0 / 45, 4.0 / 45, 2.0 / 45},
    {32.0 / 45, -16.0 / 45, 8.0 / 45, -4.0 / 45, 2.0 / 45},
    {0.0, 0.0, 0.0, 0.0, 1.0}};

static int pack_wino(const TensorMap& T, const std::string& name, int Cin, int Cout, ConvLayer& L)
{
    if (L.prec != PREC_F16X3 || Cin % 32 || Cout % 64) return P2P_OK;       // strict-fp32 models keep the direct form only
    const float* k = T.get(name + ".kernel", (int64_t)25 * Cin * Cout);
    if (!k) return P2P_ERR_WEIGHTS;
    const int S = Cin / 16;
    // U as [co][j][ky][ci] floats
    const size_t per_co = (size_t)8 * 5 * Cin;
    std::vector<float> U((size_t)Cout * per_co);
    std::vector<float> rs((size_t)Cout, 1.f);
    parallel_rows((size_t)Cout, per_co * 5, [&](size_t c0, size_t c1) {
        for (size_t co = c0; co < c1; ++co) {
            float* u = U.data() + co * per_co;
            for (int j = 0; j < 8; ++j)
                for (int ky = 0; ky < 5; ++ky)
                    for (int ci = 0; ci < Cin; ++ci) {
                        double a = 0.0;
                        for (int kx = 0; kx < 5; ++kx) a += kWinoG[j][kx] * (double)k[((size_t)(ky * 5 + kx) * Cin + ci) * Cout + co];
                        u[((size_t)j * 5 + ky) * Cin + ci] = (float)a;
                    }
            rs[co] = f16x3_row_scale(u, per_co);
        }
    });
    const size_t halves = (size_t)(Cout / 64) * 8 * S * 5 * 4 * 512 + 4 * 512;      // + one K-step of padding (wino_gemm_kernel loads one ahead)
    std::vector<float> panel((halves + 1) / 2, 0.f);
    uint16_t* o = reinterpret_cast<uint16_t*>(panel.data());
    parallel_rows((size_t)(Cout / 64) * 8, (size_t)S * 5 * 4 * 512, [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; ++r) {
            const int nt = (int)(r / 8), j = (int)(r % 8);
            for (int s = 0; s < S; ++s)
                for (int ky = 0; ky < 5; ++ky)
                    for (int f = 0; f < 4; ++f)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = nt * 64 + (f >> 1) * 32 + (lane & 31);
                            const float sc = rs[co];
                            const float* u = U.data() + (size_t)co * per_co + ((size_t)j * 5 + ky) * Cin + s * 16 + (lane >> 5) * 8;
                            uint16_t* dst = o + ((((r * S + s) * 5 + ky) * 4 + f) * 64 + lane) * 8;
                            for (int e = 0; e < 8; ++e) {
                                const float v = u[e] * sc;
                                const uint16_t hi = f32_to_f16(v);
                                dst[e] = (f & 1) ? f32_to_f16(v - f16_to_f32(hi)) : hi;
                            }
                        }
        }
    });
    std::vector<float> scale, shift;
    int rc = fold_bn(T, name, Cout, true, scale, shift);
    if (rc) return rc;
    for (int c = 0; c < Cout; ++c) scale[c] *= 1.f / rs[c];          // exact: a power of two
    if ((rc = upload(panel, &L.wino_u))) return rc;
    if ((rc = upload(scale, &L.wino_scale))) return rc;
    L.wino_bytes = panel.size() * sizeof(float);
    return P2P_OK;
}

// Winograd F(4,3) panel of a Conv2DTranspose 5x5 stride-2 layer (wino3.hip), beside the four direct phase panels: for phase (py, px)
//   U_j[ky][ci][co] = sum_dx G[j][dx + 1] k[py + 1 - 2 (ky - 1)][px + 1 - 2 dx][co][ci]      (taps outside the 5x5 kernel are zero)
// computed in double, rounded to fp32, pre-scaled per (phase, output channel) by a power of two, split hi / lo, stored in the order the
// GEMM kernel's waves stream it: [py][px][Cout / 64][position 6][Cin / 16][ky 2 + py][fragment 4][lane 64][8 halves].
static const double kWino3G[6][3] = {
    {1.0 / 4, 0.0, 0.0},
    {-1.0 / 6, -1.0 / 6, -1.0 / 6},
    {-1.0 / 6, 1.0 / 6, -1.0 / 6},
    {1.0 / 24, 1.0 / 12, 1.0 / 6},
    {1.0 / 24, -1.0 / 12, 1.0 / 6},
    {0.0, 0.0, 1.0}};

static int pack_wino3(const TensorMap& T, const std::string& name, int Cin, int Cout, ConvLayer& L)
{
    if (L.prec != PREC_F16X3 || Cin % 32 || Cout % 64) return P2P_OK;       // strict-fp32 models keep the direct phases only
    const float* k = T.get(name + ".kernel", (int64_t)25 * Cin * Cout);      // (kh, kw, Cout, Cin)
    if (!k) return P2P_ERR_WEIGHTS;
    const int S = Cin / 16, NT = Cout / 64;
    const size_t per_co = (size_t)6 * 3 * Cin;                               // [j][ky][ci]; ky >= 2 + py stays zero
    std::vector<float> U((size_t)4 * Cout * per_co, 0.f);
    std::vector<float> rs((size_t)4 * Cout, 1.f);
    parallel_rows((size_t)4 * Cout, per_co * 3, [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; ++r) {
            const int ph = (int)(r / Cout), co = (int)(r % Cout), py = ph >> 1, px = ph & 1;
            float* u = U.data() + r * per_co;
            for (int j = 0; j < 6; ++j)
                for (int ky = 0; ky < 2 + py; ++ky) {
                    const int kh = py + 1 - 2 * (ky - 1);
                    for (int ci = 0; ci < Cin; ++ci) {
                        double a = 0.0;
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int kw = px + 1 - 2 * dx;
                            if (kw >= 0 && kw < 5) a += kWino3G[j][dx + 1] * (double)k[(((size_t)kh * 5 + kw) * Cout + co) * Cin + ci];
                        }
                        u[((size_t)j * 3 + ky) * Cin + ci] = (float)a;
                    }
                }
            rs[r] = f16x3_row_scale(u, per_co);
        }
    });
    const size_t stream0 = (size_t)S * 2 * 4 * 512, stream1 = (size_t)S * 3 * 4 * 512;      // halves of one (phase, channel tile, position) stream
    const size_t halves = (size_t)2 * NT * 6 * (stream0 + stream1) + 4 * 512;                // + one K-step of padding (the kernel loads one ahead)
    std::vector<float> panel((halves + 1) / 2, 0.f);
    uint16_t* o = reinterpret_cast<uint16_t*>(panel.data());
    parallel_rows((size_t)4 * NT * 6, stream1, [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; ++r) {
            const int j = (int)(r % 6), nt = (int)((r / 6) % NT), ph = (int)(r / 6 / NT), py = ph >> 1, px = ph & 1, nky = 2 + py;
            uint16_t* base = o + (py ? (size_t)2 * NT * 6 * stream0 : 0) + (size_t)((px * NT + nt) * 6 + j) * (py ? stream1 : stream0);
            for (int s = 0; s < S; ++s)
                for (int ky = 0; ky < nky; ++ky)
                    for (int f = 0; f < 4; ++f)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = nt * 64 + (f >> 1) * 32 + (lane & 31);
                            const float sc = rs[(size_t)ph * Cout + co];
                            const float* u = U.data() + ((size_t)ph * Cout + co) * per_co + ((size_t)j * 3 + ky) * Cin + s * 16 + (lane >> 5) * 8;
                            uint16_t* dst = base + ((((size_t)s * nky + ky) * 4 + f) * 64 + lane) * 8;
                            for (int e = 0; e < 8; ++e) {
                                const float v = u[e] * sc;
                                const uint16_t hi = f32_to_f16(v);
                                dst[e] = (f & 1) ? f32_to_f16(v - f16_to_f32(hi)) : hi;
                            }
                        }
        }
    });
    std::vector<float> scale, shift, scale4((size_t)4 * Cout);
    int rc = fold_bn(T, name, Cout, true, scale, shift);
    if (rc) return rc;
    for (int ph = 0; ph < 4; ++ph)
        for (int c = 0; c < Cout; ++c) scale4[(size_t)ph * Cout + c] = scale[c] * (1.f / rs[(size_t)ph * Cout + c]);      // exact: a power of two
    if ((rc = upload(panel, &L.wino_u))) return rc;
    if ((rc = upload(scale4, &L.wino_scale))) return rc;
    L.wino_bytes = panel.size() * sizeof(float);
    return P2P_OK;
}

// Winograd F(4,3) panel of a Conv2D 5x5 stride-2 'SAME' layer on an 8x8 output grid (wino3o.hip, mode 1), beside the direct panel: on the
// parity plane (a, b) of the input the layer is a (2 + a) x (2 + b)-tap correlation (odd planes: kernel index 2 (d + 1) at offset d = -1, 0, 1;
// even planes: 2 d + 1 at d = 0, 1), so
//   U_j[a][b][ky][ci][co] = sum_dx G[j][dx + 1] k[kh(a, ky)][kw(b, dx)][ci][co],   kh(1, ky) = 2 ky, kh(0, ky) = 2 ky + 1, kw(1, dx) = 2 dx + 2, kw(0, dx) = 2 dx + 1
// stored [Cout / 64][position 6][a = 0: (b, Cin / 16, ky 2) | a = 1: (b, Cin / 16, ky 3)][fragment 4][lane 64][8 halves].  Branches are concatenated along Cout.
static int pack_wino3_s2(const TensorMap& T, const std::vector<std::string>& names, int Cin, int cout_each, ConvLayer& L)
{
    const int nb = (int)names.size(), Cout = cout_each * nb;
    if (L.prec != PREC_F16X3 || Cin % 32 || Cout % 64) return P2P_OK;
    const int S = Cin / 16, NT = Cout / 64;
    const size_t per_co = (size_t)6 * 10 * Cin;                              // [j][(a, b, ky): 10 combinations][ci]
    std::vector<float> U((size_t)Cout * per_co, 0.f);
    std::vector<float> rs((size_t)Cout, 1.f);
    static const int comb_a[10] = {0, 0, 0, 0, 1, 1, 1, 1, 1, 1}, comb_b[10] = {0, 0, 1, 1, 0, 0, 0, 1, 1, 1}, comb_ky[10] = {0, 1, 0, 1, 0, 1, 2, 0, 1, 2};
    for (int br = 0; br < nb; ++br) {
        const float* k = T.get(names[br] + ".kernel", (int64_t)25 * Cin * cout_each);      // (kh, kw, Cin, Cout)
        if (!k) return P2P_ERR_WEIGHTS;
        parallel_rows((size_t)cout_each, per_co * 3, [&](size_t c0, size_t c1) {
            for (size_t c = c0; c < c1; ++c) {
                float* u = U.data() + ((size_t)br * cout_each + c) * per_co;
                for (int j = 0; j < 6; ++j)
                    for (int q = 0; q < 10; ++q) {
                        const int a = comb_a[q], b = comb_b[q], ky = comb_ky[q];
                        const int kh = a ? 2 * ky : 2 * ky + 1;
                        for (int ci = 0; ci < Cin; ++ci) {
                            double acc = 0.0;
                            for (int dx = -1; dx <= 1; ++dx) {
                                const int kw = b ? 2 * dx + 2 : 2 * dx + 1;
                                if (kw >= 0 && kw < 5) acc += kWino3G[j][dx + 1] * (double)k[(((size_t)kh * 5 + kw) * Cin + ci) * cout_each + c];
                            }
                            u[((size_t)j * 10 + q) * Cin + ci] = (float)acc;
                        }
                    }
                rs[(size_t)br * cout_each + c] = f16x3_row_scale(u, per_co);
            }
        });
    }
    const size_t stream = (size_t)10 * S * 4 * 512;                          // halves of one (channel tile, position) stream
    const size_t halves = (size_t)NT * 6 * stream + 4 * 512;                 // + one K-step of padding
    std::vector<float> panel((halves + 1) / 2, 0.f);
    uint16_t* o = reinterpret_cast<uint16_t*>(panel.data());
    parallel_rows((size_t)NT * 6, stream, [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; ++r) {
            const int j = (int)(r % 6), nt = (int)(r / 6);
            uint16_t* base = o + r * stream;
            size_t kb = 0;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b)
                    for (int sl = 0; sl < S; ++sl)
                        for (int ky = 0; ky < 2 + a; ++ky, ++kb) {
                            int q = 0;
                            while (comb_a[q] != a || comb_b[q] != b || comb_ky[q] != ky) ++q;
                            for (int f = 0; f < 4; ++f)
                                for (int lane = 0; lane < 64; ++lane) {
                                    const int co = nt * 64 + (f >> 1) * 32 + (lane & 31);
                                    const float sc = rs[co];
                                    const float* u = U.data() + (size_t)co * per_co + ((size_t)j * 10 + q) * Cin + sl * 16 + (lane >> 5) * 8;
                                    uint16_t* dst = base + ((kb * 4 + f) * 64 + lane) * 8;
                                    for (int e = 0; e < 8; ++e) {
                                        const float v = u[e] * sc;
                                        const uint16_t hi = f32_to_f16(v);
                                        dst[e] = (f & 1) ? f32_to_f16(v - f16_to_f32(hi)) : hi;
                                    }
                                }
                        }
        }
    });
    std::vector<float> scale, shift;
    for (int br = 0; br < nb; ++br) {
        int rc = fold_bn(T, names[br], cout_each, true, scale, shift);
        if (rc) return rc;
    }
    for (int c = 0; c < Cout; ++c) scale[c] *= 1.f / rs[c];          // exact: a power of two
    int rc;
    if ((rc = upload(panel, &L.wino_u))) return rc;
    if ((rc = upload(scale, &L.wino_scale))) return rc;
    L.wino_bytes = panel.size() * sizeof(float);
    return P2P_OK;
}

// First-layer (Cin=3) direct-conv panel: [kh*kw*3][Cout] (branches concatenated along Cout).
static int pack_conv_first(const TensorMap& T, const std::vector<std::string>& names, int KH, int cout_each,
                           ConvLayer& L)
{
    const int nb = (int)names.size();
    L.prec = PREC_F32;                 // VALU direct convolution: always fp32
    L.Cout = cout_each * nb;
    L.K = KH * KH * 3;
    std::vector<float> w((size_t)L.K * L.Cout), scale, shift;
    for (int b = 0; b < nb; ++b) {
        const float* k = T.get(names[b] + ".kernel", (int64_t)L.K * cout_each);
        if (!k) return P2P_ERR_WEIGHTS;
        for (int r = 0; r < L.K; ++r)
            for (int co = 0; co < cout_each; ++co) w[(size_t)r * L.Cout + b * cout_each + co] = k[(size_t)r * cout_each + co];
        int rc = fold_bn(T, names[b], cout_each, true, scale, shift);
        if (rc) return rc;
    }
    if (g_pack_prec == PREC_F16X3 && conv1_f16x3_supported(KH, L.Cout) && specialised_kernels()) {
        // matrix-core variant (conv1.hip): split-f16 panel in that kernel's own layout; L.prec marks it
        const int Cout = L.Cout;
        std::vector<float> panel(conv1_f16x3_panel_floats(KH, Cout), 0.f), sc(scale), ws(Cout);
        for (int co = 0; co < Cout; ++co) ws[co] = f16x3_row_scale(w.data() + co, (size_t)L.K, (size_t)Cout);   // per output channel
        uint16_t* o = reinterpret_cast<uint16_t*>(panel.data());
        for (int kh = 0; kh < KH; ++kh)
            for (int kw = 0; kw < KH; ++kw)
                for (int c = 0; c < 3; ++c)
                    for (int co = 0; co < Cout; ++co) {
                        const float v = w[(size_t)((kh * KH + kw) * 3 + c) * Cout + co] * ws[co];
                        const uint16_t hi = f32_to_f16(v);
                        o[conv1_f16x3_panel_index(KH, kh, 0, kw, c, co)] = hi;
                        o[conv1_f16x3_panel_index(KH, kh, 1, kw, c, co)] = f32_to_f16(v - f16_to_f32(hi));
                    }
        for (int co = 0; co < Cout; ++co) sc[co] *= 1.f / ws[co];
        L.prec = PREC_F16X3;
        int rc;
        if ((rc = upload(panel, &L.w))) return rc;
        if ((rc = upload(sc, &L.scale))) return rc;
        return upload(shift, &L.shift);
    }
    return finish_layer(L, w, scale, shift);
}

// Projection-shortcut block tail as ONE 1x1 convolution over the channel concatenation [t (f1) || x (cin)]:
//   relu(BN2c(W2c t) + BN1(W1 x)) = relu([s2c W2c | s1 W1] [t ; x] + (shift2c + shift1))
// (resnet50_mod.py:81-91: conv 2c + BN, shortcut conv + BN, add, relu).  Saves writing the shortcut
// tensor and reading it back as a residual.  x may live on a finer grid (stride-2 blocks: IgemmParams::seg1_stride).
static int pack_merged_shortcut(const TensorMap& T, const std::string& n, int f1, int cin, int f3, ConvLayer& L)
{
    const float* k2 = T.get(n + "_2c.kernel", (int64_t)f1 * f3);
    const float* k1 = T.get(n + "_1.kernel", (int64_t)cin * f3);
    if (!k2 || !k1) return P2P_ERR_WEIGHTS;
    std::vector<float> s2, h2, s1, h1;
    int rc;
    if ((rc = fold_bn(T, n + "_2c", f3, true, s2, h2))) return rc;
    if ((rc = fold_bn(T, n + "_1", f3, true, s1, h1))) return rc;
    L.prec = g_pack_prec;
    L.Cout = f3;
    L.ntaps = 1;
    L.K = f1 + cin;
    L.dy[0] = L.dx[0] = 0;
    if (f1 % IGEMM_BK || cin % IGEMM_BK) { set_error("pack_merged_shortcut(%s): unsupported shape", n.c_str()); return P2P_ERR_INVALID_ARG; }
    std::vector<float> w((size_t)round_up(f3, 128) * L.K, 0.f), scale(f3, 1.f), shift(f3);
    for (int co = 0; co < f3; ++co) {
        for (int k = 0; k < f1; ++k) w[(size_t)co * L.K + k] = (float)((double)s2[co] * (double)k2[(size_t)k * f3 + co]);
        for (int k = 0; k < cin; ++k) w[(size_t)co * L.K + f1 + k] = (float)((double)s1[co] * (double)k1[(size_t)k * f3 + co]);
        shift[co] = h2[co] + h1[co];
    }
    return finish_layer(L, w, scale, shift);
}

// Conv2DTranspose 5x5 stride 2 'SAME' (kernel (kh,kw,Cout,Cin)); y[o] = sum x[i] w[k] with
// o = 2i + k - 1.  Output phase (py,px): o = 2m+p uses k = p+1-2d at i = m+d,
//   p=0: (d,k) in {(0,1), (-1,3)};  p=1: (d,k) in {(+1,0), (0,2), (-1,4)}   (SURVEY 8a-N5).
static int pack_deconv_phase(const TensorMap& T, const std::string& name, int Cin, int Cout, int py, int px,
                             ConvLayer& L, bool with_epilogue)
{
    const float* k = T.get(name + ".kernel", (int64_t)25 * Cin * Cout);
    if (!k) return P2P_ERR_WEIGHTS;
    L.prec = g_pack_prec;
    L.Cout = Cout;
    L.ntaps = 0;
    int khs[3], kws[3], nkh = 0, nkw = 0;
    for (int d = 1; d >= -1; --d) {
        const int kh = py + 1 - 2 * d;
        if (kh >= 0 && kh < 5) khs[nkh++] = kh;
        const int kw = px + 1 - 2 * d;
        if (kw >= 0 && kw < 5) kws[nkw++] = kw;
    }
    L.K = nkh * nkw * Cin;
    std::vector<float> w((size_t)round_up(Cout, 128) * L.K, 0.f);
    for (int a = 0; a < nkh; ++a)
        for (int b = 0; b < nkw; ++b) {
            const int kh = khs[a], kw = kws[b], t = L.ntaps++;
            L.dy[t] = (int8_t)((py + 1 - kh) / 2);
            L.dx[t] = (int8_t)((px + 1 - kw) / 2);
            for (int co = 0; co < Cout; ++co)
                for (int ci = 0; ci < Cin; ++ci)
                    w[(size_t)co * L.K + (size_t)t * Cin + ci] = k[(((size_t)kh * 5 + kw) * Cout + co) * Cin + ci];
        }
    std::vector<float> scale, shift;
    int rc = fold_bn(T, name, Cout, true, scale, shift);
    if (rc) return rc;
    (void)with_epilogue;
    return finish_layer(L, w, scale, shift);
}

// Both heads (Conv2DTranspose 128->3 tanh, 128->1 sigmoid; ae_model.py:233-236) as ONE 3x3-tap
// convolution over the 64x64 input grid with 16 "channels" = 4 output phases x (x,y,z,prob);
// taps a phase does not use are zero (25 of 36 tap-phase pairs are live).
static int pack_heads(const TensorMap& T, ConvLayer& L)
{
    const int Cin = 128;
    const float* kx = T.get("head_xyz.kernel", (int64_t)25 * 3 * Cin);
    const float* kp = T.get("head_prob.kernel", (int64_t)25 * 1 * Cin);
    const float* bx = T.get("head_xyz.bias", 3);
    const float* bp = T.get("head_prob.bias", 1);
    if (!kx || !kp || !bx || !bp) return P2P_ERR_WEIGHTS;
    L.prec = g_pack_prec;
    L.Cout = 16;
    L.ntaps = 9;
    L.K = 9 * Cin;
    std::vector<float> w((size_t)128 * L.K, 0.f), scale(16, 1.f), shift(16);
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int t = (dy + 1) * 3 + (dx + 1);
            L.dy[t] = (int8_t)dy;
            L.dx[t] = (int8_t)dx;
            for (int py = 0; py < 2; ++py)
                for (int px = 0; px < 2; ++px) {
                    const int kh = py + 1 - 2 * dy, kw = px + 1 - 2 * dx;
                    if (kh < 0 || kh > 4 || kw < 0 || kw > 4) continue;
                    for (int ch = 0; ch < 4; ++ch) {
                        const int j = (py * 2 + px) * 4 + ch;
                        for (int ci = 0; ci < Cin; ++ci)
                            w[(size_t)j * L.K + (size_t)t * Cin + ci] =
                                ch < 3 ? kx[(((size_t)kh * 5 + kw) * 3 + ch) * Cin + ci] : kp[((size_t)kh * 5 + kw) * Cin + ci];
                    }
                }
        }
    for (int j = 0; j < 16; ++j) shift[j] = (j & 3) < 3 ? bx[j & 3] : bp[0];
    return finish_layer(L, w, scale, shift);
}

// Dense kernel (in,out) -> W[out_pad][in]; bias in shift.
static int pack_dense(const TensorMap& T, const std::string& name, int In, int Out, ConvLayer& L)
{
    const float* k = T.get(name + ".kernel", (int64_t)In * Out);
    if (!k) return P2P_ERR_WEIGHTS;
    L.prec = g_pack_prec;
    L.Cout = Out;
    L.ntaps = 1;
    L.K = In;
    L.dy[0] = L.dx[0] = 0;
    std::vector<float> w((size_t)round_up(Out, 128) * In, 0.f), scale, shift;
    parallel_rows((size_t)Out, (size_t)In, [&](size_t o0, size_t o1) {
        for (int i = 0; i < In; ++i)
            for (size_t o = o0; o < o1; ++o) w[o * In + i] = k[(size_t)i * Out + o];
    });
    int rc = fold_bn(T, name, Out, false, scale, shift);
    if (rc) return rc;
    return finish_layer(L, w, scale, shift);
}

static void free_layer(ConvLayer& L)
{
    if (L.w) hipFree(L.w);
    if (L.scale) hipFree(L.scale);
    if (L.shift) hipFree(L.shift);
    if (L.wino_u) hipFree(L.wino_u);
    if (L.w_frag) hipFree(L.w_frag);
    if (L.wino_scale) hipFree(L.wino_scale);
    L.w = L.scale = L.shift = L.wino_u = L.wino_scale = L.w_frag = nullptr;
}

// ------------------------------------------------------------------------------------------
// model construction
// ------------------------------------------------------------------------------------------
static int build_decoder(const TensorMap& T, Model& M, int skip3, int skip2, int skip1)
{
    int rc;
    if ((rc = pack_dense(T, "dense_enc", 32768, 256, M.L["dense_enc"]))) return rc;
    if ((rc = pack_dense(T, "dense_dec", 256, 16384, M.L["dense_dec"]))) return rc;
    const struct { const char* n; int cin, cout; } ups[3] = {{"up1", 256, 256}, {"up2", 256, 128}, {"up3", 256, 64}};
    for (const auto& u : ups)
        for (int ph = 0; ph < 4; ++ph) {
            ConvLayer& L = M.L[std::string(u.n) + "_p" + std::to_string(ph)];
            if ((rc = pack_deconv_phase(T, u.n, u.cin, u.cout, ph >> 1, ph & 1, L, true))) return rc;
        }
    for (int i = 0; i < 3; ++i)          // Winograd F(4,3) panel of the layer, kept with its phase-0 entry (wino3o.hip serves the 8x8 input grid, wino3.hip 16x16 and 32x32)
        if ((rc = pack_wino3(T, ups[i].n, ups[i].cin, ups[i].cout, M.L[std::string(ups[i].n) + "_p0"]))) return rc;
    if ((rc = pack_conv(T, {"deconv1"}, 5, 256 + skip3, 256, 2, true, M.L["deconv1"]))) return rc;
    if ((rc = pack_conv(T, {"deconv2"}, 5, 128 + skip2, 256, 2, true, M.L["deconv2"]))) return rc;
    if ((rc = pack_conv(T, {"deconv3"}, 5, 64 + skip1, 128, 2, true, M.L["deconv3"]))) return rc;
    if ((rc = pack_wino(T, "deconv1", 256 + skip3, 256, M.L["deconv1"]))) return rc;
    if ((rc = pack_wino(T, "deconv2", 128 + skip2, 256, M.L["deconv2"]))) return rc;
    if ((rc = pack_wino(T, "deconv3", 64 + skip1, 128, M.L["deconv3"]))) return rc;
    return pack_heads(T, M.L["heads"]);
}

static int build_model(const TensorMap& T, Model& M)
{
    int rc;
    g_pack_prec = M.prec;      // the igemm panels packed below take the model's precision
    if (M.backbone == P2P_BACKBONE_RESNET50) {
        if ((rc = pack_conv_first(T, {"conv1"}, 7, 64, M.L["conv1"]))) return rc;
        const struct { const char* n; int cin, f1, f3; bool sc; } blocks[7] = {
            {"res2a", 64, 64, 256, true},    {"res2b", 256, 64, 256, false},  {"res2c", 256, 64, 256, false},
            {"res3a", 256, 128, 512, true},  {"res3b", 512, 128, 512, false}, {"res3c", 512, 128, 512, false},
            {"res3d", 512, 128, 512, false}};
        for (const auto& b : blocks) {
            const std::string n = b.n;
            if ((rc = pack_conv(T, {n + "_2a"}, 1, b.cin, b.f1, 0, true, M.L[n + "_2a"]))) return rc;
            if ((rc = pack_conv(T, {n + "_2b"}, 3, b.f1, b.f1, 1, true, M.L[n + "_2b"]))) return rc;
            if (b.sc) {                          // projection shortcut: folded into the block's last convolution
                if ((rc = pack_merged_shortcut(T, n, b.f1, b.cin, b.f3, M.L[n + "_2c1"]))) return rc;
                continue;
            }
            if ((rc = pack_conv(T, {n + "_2c"}, 1, b.f1, b.f3, 0, true, M.L[n + "_2c"]))) return rc;
            if (b.sc && (rc = pack_conv(T, {n + "_1"}, 1, b.cin, b.f3, 0, true, M.L[n + "_1"]))) return rc;
        }
        if ((rc = pack_conv(T, {"conv4_1", "conv4_2"}, 5, 512, 256, 1, true, M.L["conv4"]))) return rc;
        if ((rc = pack_wino3_s2(T, {"conv4_1", "conv4_2"}, 512, 256, M.L["conv4"]))) return rc;
        return build_decoder(T, M, 128, 128, 32);
    }
    if (M.backbone == P2P_BACKBONE_PAPER) {
        if ((rc = pack_conv_first(T, {"conv1_1", "conv1_2"}, 5, 64, M.L["conv1"]))) return rc;
        if ((rc = pack_conv(T, {"conv2_1", "conv2_2"}, 5, 128, 128, 1, true, M.L["conv2"]))) return rc;
        if ((rc = pack_conv(T, {"conv3_1", "conv3_2"}, 5, 256, 128, 1, true, M.L["conv3"]))) return rc;
        if ((rc = pack_conv(T, {"conv4_1", "conv4_2"}, 5, 256, 256, 1, true, M.L["conv4"]))) return rc;
        if ((rc = pack_wino3_s2(T, {"conv4_1", "conv4_2"}, 256, 256, M.L["conv4"]))) return rc;
        return build_decoder(T, M, 128, 128, 64);
    }
    set_error("unknown backbone %d", M.backbone);
    return P2P_ERR_INVALID_ARG;
}

// Folded BatchNorm of an identity bottleneck block's three layers in ONE device array, the layout resblock.hip reads:
// [scale 2a | shift 2a | scale 2b | shift 2b | scale 2c | shift 2c] (F1, F1, F1, F1, 4 F1, 4 F1 floats).  Copied device to device
// from the layers' own arrays, so both routes multiply by the same numbers.
static int pack_block_ss(Model& M)
{
    if (M.backbone != P2P_BACKBONE_RESNET50 || M.prec != PREC_F16X3) return P2P_OK;
    for (const char* nm : {"res2a", "res2b", "res2c", "res3a", "res3b", "res3c", "res3d"}) {
        const std::string n = nm;
        if (!M.L.count(n + "_2c") && !M.L.count(n + "_2c1")) continue;
        // (projection blocks: the last convolution is the merged [2c | shortcut] layer, pack_merged_shortcut)
        const ConvLayer &a = M.L.at(n + "_2a"), &b = M.L.at(n + "_2b"), &c = M.L.count(n + "_2c") ? M.L.at(n + "_2c") : M.L.at(n + "_2c1");
        const int F1 = a.Cout, C = c.Cout;
        float* d = nullptr;
        HIP_TRY(hipMalloc((void**)&d, (size_t)(4 * F1 + 2 * C) * sizeof(float)));
        M.block_ss[n] = d;
        if (!M.L.at(n + "_2b").w_frag) {
            // fragment-ordered copy of the 2b panel: the same halves, regrouped per (32-row tile, K-step, k half, hi / lo) into 1 KB lane images
            ConvLayer& Lb = M.L.at(n + "_2b");
            const int K = Lb.K, KS = K / 32, NTL = Lb.Cout / 32;
            std::vector<float> host((size_t)round_up(Lb.Cout, 128) * K), frag((size_t)Lb.Cout * K);
            HIP_TRY(hipMemcpy(host.data(), Lb.w, host.size() * sizeof(float), hipMemcpyDeviceToHost));
            const uint16_t* sh = reinterpret_cast<const uint16_t*>(host.data());
            uint16_t* dh = reinterpret_cast<uint16_t*>(frag.data());
            for (int nt = 0; nt < NTL; ++nt)
                for (int ks = 0; ks < KS; ++ks)
                    for (int f = 0; f < 4; ++f)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int row = nt * 32 + (lane & 31), kb = f >> 1, hf = f & 1;
                            const uint16_t* sp = sh + ((size_t)row * K + (size_t)ks * 32) * 2 + hf * 32 + kb * 16 + (lane >> 5) * 8;
                            uint16_t* dp = dh + ((((size_t)nt * KS + ks) * 4 + f) * 64 + lane) * 8;
                            for (int e = 0; e < 8; ++e) dp[e] = sp[e];
                        }
            int rcu = upload(frag, &Lb.w_frag);
            if (rcu) return rcu;
        }
        const float* src[6] = {a.scale, a.shift, b.scale, b.shift, c.scale, c.shift};
        const int len[6] = {F1, F1, F1, F1, C, C};
        size_t off = 0;
        for (int k = 0; k < 6; ++k) {
            HIP_TRY(hipMemcpy(d + off, src[k], (size_t)len[k] * sizeof(float), hipMemcpyDeviceToDevice));
            off += len[k];
        }
    }
    return P2P_OK;
}

// ------------------------------------------------------------------------------------------
// activation workspace
// ------------------------------------------------------------------------------------------
constexpr size_t WINO_PARTIAL_FLOATS = (size_t)256 * 16 * 32 * 64;      // 256 (tile, K range) pairs x the 16 x 32 pixels x 64 channels of a tile
static const struct { const char* name; size_t per_sample; } kBuffers[] = {
    // shared
    {"f4", 8 * 8 * 512}, {"enc", 256}, {"dd", 8 * 8 * 256}, {"u1", 16 * 16 * 256}, {"c1", 16 * 16 * 256},
    {"u2", 32 * 32 * 128}, {"c2", 32 * 32 * 256}, {"u3", 64 * 64 * 64}, {"c3", 64 * 64 * 128},
    {"part", 32 * 256},
    {"wv", 64 * 64 * 128 * 2},      // Winograd-transformed input of the largest 5x5 layer (deconv3 of the paper backbone: 8 bytes per input element)
    // front (resnet50 sizes dominate the paper ones)
    {"f1", 64 * 64 * 128}, {"p1", 32 * 32 * 64}, {"t_a", 32 * 32 * 64}, {"t_b", 32 * 32 * 64},
    {"sc", 32 * 32 * 256}, {"o_a", 32 * 32 * 256}, {"o_b", 32 * 32 * 256}, {"f2", 32 * 32 * 256},
    {"f3", 16 * 16 * 512},
};

int Ctx::ensure_lane(int i)
{
    Lane& ln = lane[i];
    if (!ln.act.empty()) return P2P_OK;
    if (!ln.stream) {
        if (i == 0) ln.stream = stream;
        else HIP_TRY(hipStreamCreate(&ln.stream));
    }
    if (!ln.done) HIP_TRY(hipEventCreateWithFlags(&ln.done, hipEventDisableTiming));
    if (!fork) HIP_TRY(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (const auto& b : kBuffers) {
        float* d = nullptr;
        HIP_TRY(hipMalloc((void**)&d, b.per_sample * (size_t)max_batch * sizeof(float)));
        ln.act[b.name] = d;
    }
    {   // raw partial sums of a split-K Winograd launch (try_wino): launches under one workgroup per CU only, so the size does not grow with max_batch
        float* d = nullptr;
        HIP_TRY(hipMalloc((void**)&d, WINO_PARTIAL_FLOATS * sizeof(float)));
        ln.act["wp"] = d;
    }
    return P2P_OK;
}

int Ctx::range_read(int word, float* out)
{
    unsigned bits = 0;
    *out = 0.f;
    if (!range_words) return P2P_OK;
    HIP_TRY(hipMemcpy(&bits, range_words + word, sizeof(bits), hipMemcpyDeviceToHost));
    if (bits) {
        HIP_TRY(hipMemset(range_words + word, 0, sizeof(bits)));
        memcpy(out, &bits, sizeof(bits));
    }
    return P2P_OK;
}

int Ctx::ensure_workspace()
{
    if (x_stage) return P2P_OK;
    int rc = ensure_lane(0);
    if (rc) return rc;
    HIP_TRY(hipMalloc((void**)&x_stage, (size_t)max_batch * 128 * 128 * 3 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&xyzp_stage, (size_t)max_batch * 128 * 128 * 4 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&xyz_stage, (size_t)max_batch * 128 * 128 * 3 * sizeof(float)));
    HIP_TRY(hipMalloc((void**)&prob_stage, (size_t)max_batch * 128 * 128 * sizeof(float)));
    return P2P_OK;
}

// ------------------------------------------------------------------------------------------
// layer launch helpers
// ------------------------------------------------------------------------------------------
struct Src {
    const float* ptr; int C; int cstride; int coff;
};

struct ConvCall {
    Src s0{nullptr, 0, 0, 0}, s1{nullptr, 0, 0, 0};
    int N = 0, Hin = 0, Win = 0, Hg = 0, Wg = 0, in_stride = 1;
    float* out = nullptr;
    int Hout = 0, Wout = 0, os = 1, oy = 0, ox = 0, out_cstride = 0, out_coff = 0;
    int act = ACT_NONE;
    const float* residual = nullptr;
    int res_cstride = 0;
    int mode = EPI_NORMAL;
    int ksplit = 1;
    float* partial = nullptr;
    double algo_macs = -1;   // algorithmic MACs of this launch; < 0 => M * Cout * K (no padded work)
    int s1_Hin = 0, s1_Win = 0, s1_stride = 0;   // segment 1 on its own (strided) grid; 1-tap layers only
};

// A launch ready to go: parameter block, route and (streaming route) K-step order.
struct PreparedConv {
    IgemmParams p;
    StreamOrder so;
    bool stream = false, halo = false, halo_conv = false, halo8 = false, halo_s2 = false;
    int cfg = 0, prof_slot = 0;
    double flops = 0, bytes = 0;
};

static int prepare_conv(Ctx& X, const ConvLayer& L, const ConvCall& c, PreparedConv& pc)
{
    IgemmParams& p = pc.p;
    memset(&p, 0, sizeof(p));
    p.seg[0] = {c.s0.ptr, c.s0.C, c.s0.cstride, c.s0.coff};
    p.seg[1] = {c.s1.ptr, c.s1.C, c.s1.cstride, c.s1.coff};
    const int cin = c.s0.C + c.s1.C;
    if (c.s0.C % IGEMM_BK || c.s1.C % IGEMM_BK || cin * L.ntaps != L.K) {
        set_error("run_conv: channel segments (%d+%d) x %d taps do not match packed K=%d", c.s0.C, c.s1.C, L.ntaps, L.K);
        return P2P_ERR_INVALID_ARG;
    }
    p.seg1_Hin = c.s1_Hin; p.seg1_Win = c.s1_Win; p.seg1_stride = c.s1_stride;
    if (c.s1_stride && L.ntaps != 1) { set_error("run_conv: a strided second segment needs a 1-tap layer"); return P2P_ERR_INVALID_ARG; }
    p.seg0_chunks = c.s0.C / IGEMM_BK;
    p.chunks_per_tap = cin / IGEMM_BK;
    p.N = c.N; p.Hin = c.Hin; p.Win = c.Win; p.Hg = c.Hg; p.Wg = c.Wg;
    p.M = c.N * c.Hg * c.Wg;
    p.in_stride = c.in_stride;
    p.ntaps = L.ntaps;
    memcpy(p.dy, L.dy, sizeof(L.dy));
    memcpy(p.dx, L.dx, sizeof(L.dx));
    {
        // buffer-descriptor ranges (32-bit): tensors are < 4 GB for max_batch <= 1024
        const size_t px = (size_t)c.N * c.Hin * c.Win;
        const size_t px1 = c.s1_stride ? (size_t)c.N * c.s1_Hin * c.s1_Win : px;
        const size_t b0 = px * c.s0.cstride * sizeof(float), b1 = px1 * c.s1.cstride * sizeof(float);
        const size_t bw = (size_t)((L.Cout + 127) / 128 * 128) * L.K * sizeof(float);
        if (b0 >= 0xFFFFFFF0ull || b1 >= 0xFFFFFFF0ull || bw >= 0xFFFFFFF0ull) {
            set_error("run_conv: tensor exceeds the 4 GB buffer-descriptor range (lower max_batch)");
            return P2P_ERR_CAPACITY;
        }
        p.seg_bytes[0] = (unsigned)b0; p.seg_bytes[1] = (unsigned)(c.s1.ptr ? b1 : b0); p.w_bytes = (unsigned)bw;
    }
    p.w = L.w; p.K = L.K; p.Cout = L.Cout;
    p.ksteps = L.K / IGEMM_BK;
    p.ksplit = c.ksplit; p.partial = c.partial;
    p.scale = L.scale; p.shift = L.shift;
    p.residual = c.residual; p.res_cstride = c.res_cstride;
    p.act = c.act; p.alpha = LEAKY;
    p.out = c.out; p.Hout = c.Hout; p.Wout = c.Wout; p.os = c.os; p.oy = c.oy; p.ox = c.ox;
    p.out_cstride = c.out_cstride; p.out_coff = c.out_coff;
    p.mode = c.mode;
    p.prec = L.prec;
    p.range_acc = (L.prec == PREC_F16X3 && c.mode == EPI_NORMAL) ? X.range_cur : nullptr;      // the heads' outputs are bounded (tanh / sigmoid)
    const int cfg = L.Cout > 64 ? 0 : (L.Cout > 32 ? 1 : 2);      // 128x128 / 128x64 / 128x32 tiles
    if (X.grp && X.grp->models.size() > 1) {
        const GroupCtx& G = *X.grp;
        const int ng = (int)G.models.size(), BM = igemm_tile_m(cfg), rows_per_sample = c.Hg * c.Wg;
        if (ng > IGEMM_MAX_GROUPS) { set_error("run_conv: %d object groups exceed IGEMM_MAX_GROUPS", ng); return P2P_ERR_CAPACITY; }
        int tile0 = 0;
        for (int g = 0; g < ng; ++g) {
            const ConvLayer& Lg = G.models[g]->L.at(L.name);
            if (Lg.prec != L.prec) { set_error("run_conv: objects of one grouped pass must share a precision"); return P2P_ERR_INVALID_ARG; }
            p.grp[g] = {Lg.w, Lg.scale, Lg.shift, G.start[g] * rows_per_sample, tile0};
            tile0 += ((G.start[g + 1] - G.start[g]) * rows_per_sample + BM - 1) / BM;
        }
        p.grp[ng] = {nullptr, nullptr, nullptr, G.start[ng] * rows_per_sample, tile0};
        p.n_groups = ng;
    }
    const bool halo = specialised_kernels() && heads_halo_supported(p);            // the output heads have their own kernel (heads.hip)
    const bool halo_conv = !halo && specialised_kernels() && igemm_halo_supported(p);   // stride-1 multi-tap layers (igemm_halo.hip)
    const bool halo8 = !halo && !halo_conv && specialised_kernels() && igemm_halo8_mode(p) != 0;   // 8x8-grid layers (igemm_halo8.hip)
    const bool halo_s2 = !halo && !halo_conv && !halo8 && specialised_kernels() && igemm_halo_s2_supported(p);   // 5x5 stride 2 on larger grids (igemm_halo_s2.hip)
    // Small launches (one detection at a time: the reference's own caller) go to the streaming kernel, which gives every 32x32 output
    // tile its own wave and walks K in the SAME order as the batched kernel chosen above -- so the bits do not depend on the route.
    StreamOrder& so = pc.so;
    bool stream = false;
    memset(&so, 0, sizeof(so));
    if (!halo && stream_max_wgs() > 0 && igemm_stream_supported(p)) {
        int grid;         // workgroups of the batched kernel
        if (halo_conv) {
            const int m_tiles = c.N * (c.Hg / 8) * (c.Wg / 16);
            grid = L.Cout % 128 == 0 ? m_tiles * (L.Cout / 128) : (c.Hg % 16 == 0 ? m_tiles / 2 : m_tiles) * (L.Cout / 64);
        } else if (halo8) grid = ((p.M + 127) / 128) * (L.Cout / 128);
        else if (halo_s2) grid = c.N * (c.Hg / 8) * (c.Wg / 16) * (L.Cout / 128);
        else grid = ((p.M + 127) / 128) * ((L.Cout + (cfg == 0 ? 127 : cfg == 1 ? 63 : 31)) / (cfg == 0 ? 128 : cfg == 1 ? 64 : 32)) * std::max(1, c.ksplit);
        // split-K launches keep their K partition (it is a property of the layer, so the partial sums and their reduction order
        // are the same on both routes); the streaming route only changes who computes a partial tile
        if (grid <= stream_max_wgs()) {
            stream = true;
            if (halo_conv || (halo8 && igemm_halo8_mode(p) == 1)) {          // (slice, tap)
                so.n_groups = 1;
                so.gstart[1] = (int8_t)L.ntaps;
                for (int t = 0; t < L.ntaps; ++t) so.tap[t] = (int8_t)t;
            } else if (halo8 || halo_s2) {                                    // (parity plane, slice, tap of the plane)
                so.n_groups = 4;
                int k = 0;
                for (int g = 0; g < 4; ++g) {
                    const int py = g >> 1, px = g & 1, ny = py ? 3 : 2, nx = px ? 3 : 2;
                    so.gstart[g] = (int8_t)k;
                    for (int iy = 0; iy < ny; ++iy)
                        for (int ix = 0; ix < nx; ++ix) so.tap[k++] = (int8_t)((py ? 2 * iy : 2 * iy + 1) * 5 + (px ? 2 * ix : 2 * ix + 1));
                }
                so.gstart[4] = (int8_t)k;
            } else {                                                          // (tap, slice)
                so.n_groups = L.ntaps;
                for (int t = 0; t <= L.ntaps; ++t) { so.gstart[t] = (int8_t)t; so.tap[t] = (int8_t)t; }
            }
        }
    }
    pc.stream = stream; pc.halo = halo; pc.halo_conv = halo_conv; pc.halo8 = halo8; pc.halo_s2 = halo_s2; pc.cfg = cfg;
    pc.prof_slot = stream ? 8 : halo ? 5 : halo_conv ? (L.Cout % 128 == 0 ? 3 : 4) : halo8 ? 6 : halo_s2 ? 7 : cfg;
    pc.flops = 2.0 * (c.algo_macs >= 0 ? c.algo_macs : (double)p.M * L.Cout * L.K);
    {   // compulsory bytes: the channels of each input segment the layer reads (once), the residual, the weight panel, the output
        const double in_px = (double)c.N * c.Hin * c.Win / (L.ntaps == 1 ? (double)c.in_stride * c.in_stride : 1.0);      // a strided 1x1 reads every s-th pixel
        const double in1_px = c.s1_stride ? (double)c.N * c.Hg * c.Wg : in_px;
        const double out_el = (double)p.M * L.Cout;
        pc.bytes = 4.0 * (in_px * c.s0.C + in1_px * c.s1.C + (c.residual ? out_el : 0.0) + (double)L.K * L.Cout +
                          (c.mode == EPI_HEAD ? (double)p.M * 16 : out_el));
    }
    return P2P_OK;
}

// n == 1: the prepared launch; n > 1 (streaming route only): the launches share input, grid and epilogue -- one merged launch
static int launch_prepared(Ctx& X, const PreparedConv* pc, int n)
{
    hipStream_t st = X.cur->stream;
    const PreparedConv& c0 = pc[0];
    StreamMulti mp;
    if (c0.stream) {
        mp.n = n;
        for (int i = 0; i < n; ++i) stream_phase_of(pc[i].p, pc[i].so, &mp.ph[i]);
    } else if (n != 1) { set_error("launch_prepared: only streaming launches merge"); return P2P_ERR_INVALID_ARG; }
    const IgemmParams& p = c0.p;
    auto launch = [&]() { return c0.stream ? launch_igemm_stream(p, mp, st) : c0.halo ? launch_heads_halo(p, st) : c0.halo_conv ? launch_igemm_halo(p, st) :
                                 c0.halo8 ? launch_igemm_halo8(p, st) : c0.halo_s2 ? launch_igemm_halo_s2(p, st) : launch_igemm(p, c0.cfg, st); };
    if (X.profiling) {
        double fl = 0, by = 0;
        for (int i = 0; i < n; ++i) { fl += pc[i].flops; by += pc[i].bytes; }
        Ctx::ProfEvent ev{X.prof_get_event(), X.prof_get_event(), c0.prof_slot, fl, by};
        if (!ev.a || !ev.b) return P2P_ERR_HIP;
        HIP_TRY(hipEventRecord(ev.a, st));
        HIP_TRY(launch());
        HIP_TRY(hipEventRecord(ev.b, st));
        X.prof_pending.push_back(ev);
        return P2P_OK;
    }
    HIP_TRY(launch());
    return P2P_OK;
}

static bool small_split_route() { static const bool on = dev_env("P2P_NO_SMALL_SPLIT") == nullptr; return on; }     // development builds: A/B against the single chain

static int run_conv(Ctx& X, const ConvLayer& L, const ConvCall& c)
{
    PreparedConv pc;
    int rc = prepare_conv(X, L, c, pc);
    if (rc) return rc;
    return launch_prepared(X, &pc, 1);
}

// The long-K layers of a pass over a few inputs (one detection at a time: conv4 walks 400 K-steps on 16 waves, deconv1 300 on 64 -- a
// quarter of such a pass while the chip idles): under "auto" -- the mode whose form already depends on the pass size -- the streaming launch
// splits K over enough waves to occupy the chip (raw partial sums into the Winograd workspace, which such a pass does not use for this layer)
// and splitk_reduce_kernel applies the epilogue.  The partial sums are added in a fixed order: the bits depend on the pass size (as the
// Winograd forms' do), not on the run.  "off" / "always" keep the single chain (size-independent bits).
static int run_conv_small(Ctx& X, const ConvLayer& L, const ConvCall& c)
{
    PreparedConv pc;
    int rc = prepare_conv(X, L, c, pc);
    if (rc) return rc;
    const IgemmParams& p = pc.p;
    if (pc.stream && X.wino_mode == P2P_WINOGRAD_AUTO && small_split_route() && c.ksplit == 1 && !c.residual && c.mode == EPI_NORMAL && c.os == 1 &&
        c.out_coff == 0 && c.out_cstride == L.Cout && c.Hout == c.Hg && c.Wout == c.Wg && p.n_groups <= 1) {
        const int tiles = igemm_stream_waves(p, 1);
        // (measured per pass size, tools/time_small.py with the development switches below: a launch of more waves than these is bound by what
        // its waves stream from L2 -- every wave loads its own operands -- and a split adds nothing: deconv2 at one input 55 us either way)
        static const int max_s1 = dev_env("P2P_SMALL_SPLIT_S1") ? atoi(dev_env("P2P_SMALL_SPLIT_S1")) : 128;      // development builds: largest launch (waves) that still splits
        static const int max_s2 = dev_env("P2P_SMALL_SPLIT_S2") ? atoi(dev_env("P2P_SMALL_SPLIT_S2")) : 256;     // stride-2 gathers stream at half the rate per wave
        int S = tiles <= (c.in_stride == 2 ? max_s2 : max_s1) ? std::min(std::min(16, 768 / std::max(1, tiles)), p.ksteps / 12) : 1;
        while (S >= 2 && (size_t)S * p.M * L.Cout > (size_t)X.max_batch * 64 * 64 * 128 * 2) --S;      // the "wv" workspace (kBuffers)
        if (S >= 2) {
            ConvCall c2 = c;
            c2.ksplit = S; c2.partial = X.cur->act["wv"];
            PreparedConv pc2;
            if ((rc = prepare_conv(X, L, c2, pc2))) return rc;
            if (pc2.stream) {
                if ((rc = launch_prepared(X, &pc2, 1))) return rc;
                HIP_TRY(launch_splitk_reduce(c2.partial, S, p.M, L.Cout, L.scale, L.shift, c.act, LEAKY, c.out, L.prec == PREC_F16X3 ? X.range_cur : nullptr, X.cur->stream));
                return P2P_OK;
            }
        }
    }
    return launch_prepared(X, &pc, 1);
}

// stride-1/2 Conv2D on a full tensor `in` [N,H,W,C] -> out [N,H/s,W/s,Cout]
static int conv_layer(Ctx& X, const ConvLayer& L, const float* in, int N, int H, int W, int C, int stride,
                      float* out, int act, const float* residual = nullptr)
{
    ConvCall c;
    c.s0 = {in, C, C, 0};
    c.N = N; c.Hin = H; c.Win = W; c.Hg = H / stride; c.Wg = W / stride; c.in_stride = stride;
    c.out = out; c.Hout = c.Hg; c.Wout = c.Wg; c.out_cstride = L.Cout;
    c.act = act; c.residual = residual; c.res_cstride = L.Cout;
    return L.ntaps == 25 ? run_conv_small(X, L, c) : run_conv(X, L, c);
}

// The 5x5 stride-1 layers of split-f16 models in Winograd form (wino.hip: 2.5x fewer MFMA products).  Measured per generator pass
// (profiles/r06_wino_small_batches.txt): faster than the direct kernels from 3 inputs up (703 vs 772 us; 8: 841 vs 971; 64: 2100 vs 2613;
// 256: 5490 vs 7740), 5 % slower at ONE input (651 vs 622 us: three launches of 4-16 workgroups walking their whole K range) -- until such
// launches split K over ranges of channel slices (try_wino: one input 0.510 -> 0.462 ms against the direct kernels with THEIR split, two
// inputs 0.607 -> 0.492, three 0.650 -> 0.539): "auto" now takes the Winograd form at every size.  Unlike the other route pairs of this file the two forms do NOT compute the same
// bits: both sit within the generator's error bar of the oracle (tests/test_wino_gpu.py), 3e-5 apart -- so the choice is the caller's
// (p2p_ctx_set_winograd: off / auto / always), not a development switch.
constexpr int WINO_MIN_INPUTS = 1;

static int timed_launch(Ctx& X, int slot, double flops, double bytes, const std::function<hipError_t()>& launch)
{
    hipStream_t st = X.cur->stream;
    if (X.profiling) {
        Ctx::ProfEvent ev{X.prof_get_event(), X.prof_get_event(), slot, flops, bytes};
        if (!ev.a || !ev.b) return P2P_ERR_HIP;
        HIP_TRY(hipEventRecord(ev.a, st));
        HIP_TRY(launch());
        HIP_TRY(hipEventRecord(ev.b, st));
        X.prof_pending.push_back(ev);
        return P2P_OK;
    }
    HIP_TRY(launch());
    return P2P_OK;
}

static bool wino_split_route() { static const bool on = dev_env("P2P_NO_WINO_SPLIT") == nullptr; return on; }     // development builds: A/B against the single chain

// 0 = not for this route (the caller falls through to the direct kernels), 1 = done, < 0 = error
static int try_wino(Ctx& X, const ConvLayer& L, const float* a, int Ca, const float* b, int Cb, int cb_stride, int cb_off, int N, int H, float* out)
{
    if (!L.wino_u || L.prec != PREC_F16X3 || X.wino_mode == P2P_WINOGRAD_OFF || !specialised_kernels() || !wino_supported(H, H, Ca, Cb, L.Cout)) return 0;
    WinoParams p;
    memset(&p, 0, sizeof(p));
    p.seg[0] = {a, Ca, Ca, 0};
    p.seg[1] = {b, Cb, cb_stride, cb_off};
    const size_t px = (size_t)N * H * H;
    const size_t b0 = px * Ca * sizeof(float), b1 = px * cb_stride * sizeof(float);
    if (b0 >= 0xFFFFFFF0ull || b1 >= 0xFFFFFFF0ull || L.wino_bytes >= 0xFFFFFFF0ull) return 0;
    p.seg_bytes[0] = (unsigned)b0; p.seg_bytes[1] = (unsigned)b1;
    p.seg0_groups = Ca / 32;
    p.N = N; p.H = H; p.W = H; p.Cin = Ca + Cb; p.Cout = L.Cout;
    p.V = X.cur->act["wv"];
    p.U = L.wino_u; p.scale = L.wino_scale; p.shift = L.shift;
    p.act = ACT_LEAKY; p.alpha = LEAKY;
    p.out = out; p.out_cstride = L.Cout; p.out_coff = 0;
    p.range_acc = X.range_cur;
    if (X.grp && X.grp->models.size() > 1) {
        const GroupCtx& G = *X.grp;
        const int ng = (int)G.models.size();
        if (ng > IGEMM_MAX_GROUPS) { set_error("run_conv: %d object groups exceed IGEMM_MAX_GROUPS", ng); return P2P_ERR_CAPACITY; }
        int unit0 = 0;                  // 16x16 grids: two samples per workgroup, every object paired up on its own
        for (int g = 0; g < ng; ++g) {
            const ConvLayer& Lg = G.models[g]->L.at(L.name);
            if (!Lg.wino_u || Lg.prec != PREC_F16X3) { set_error("run_conv: objects of one grouped pass must share a precision"); return P2P_ERR_INVALID_ARG; }
            p.grp[g] = {Lg.wino_u, Lg.wino_scale, Lg.shift, G.start[g], unit0};
            unit0 += (G.start[g + 1] - G.start[g] + 1) / 2;
        }
        p.grp[ng] = {nullptr, nullptr, nullptr, G.start[ng], unit0};
        p.n_groups = ng;
    }
    static const int wino_min = dev_env("P2P_WINO_MIN") ? atoi(dev_env("P2P_WINO_MIN")) : WINO_MIN_INPUTS;
    if (X.wino_mode != P2P_WINOGRAD_ALWAYS && N < wino_min) return 0;
    // Launches under half a workgroup per CU (a pass over a few inputs; deconv1 up to 64): under "auto" the channel slices are cut into ranges,
    // one workgroup each, and the raw sums meet in splitk_reduce_kernel (the inverse transform is linear).  Like every choice "auto" makes
    // the split depends on the pass size only; "always" keeps one chain (size-independent bits).
    if (X.wino_mode == P2P_WINOGRAD_AUTO && p.n_groups <= 1 && wino_split_route()) {
        const int tiles = wino_gemm_grid(p), S = p.Cin / 16;
        for (int k : {6, 4, 3, 2})
            if (tiles * k <= 256 && S % (2 * k) == 0 && (size_t)k * px * L.Cout <= WINO_PARTIAL_FLOATS) { p.ksplit = k; p.partial = X.cur->act["wp"]; break; }
    }
    hipStream_t st = X.cur->stream;
    const double in_el = (double)px * p.Cin, out_el = (double)px * L.Cout;
    int rc = timed_launch(X, 11, 0.0, 4.0 * in_el + 8.0 * in_el, [&]() { return launch_wino_input(p, st); });
    if (rc) return rc;
    // algorithmic work of the LAYER (the direct form's MACs, like every other slot); compulsory bytes: V once, the panel once, the output
    rc = timed_launch(X, 10, 2.0 * out_el * 25.0 * p.Cin, 8.0 * in_el + (double)L.wino_bytes + 4.0 * out_el, [&]() { return launch_wino_gemm(p, st); });
    if (rc) return rc;
    if (p.ksplit > 1) HIP_TRY(launch_splitk_reduce(p.partial, p.ksplit, (int)px, L.Cout, L.wino_scale, L.shift, ACT_LEAKY, LEAKY, out, X.range_cur, st));
    return 1;
}

// 5x5 stride-1 'SAME' conv over the concatenation [a (Ca ch) || b[..., :Cb] (pixel stride cb_stride)]
static int concat_conv(Ctx& X, const ConvLayer& L, const float* a, int Ca, const float* b, int Cb,
                       int cb_stride, int cb_off, int N, int H, float* out)
{
    {
        const int rc = try_wino(X, L, a, Ca, b, Cb, cb_stride, cb_off, N, H, out);
        if (rc < 0) return rc;          // (error codes of this library are negative)
        if (rc == 1) return P2P_OK;
    }
    ConvCall c;
    c.s0 = {a, Ca, Ca, 0};
    c.s1 = {b, Cb, cb_stride, cb_off};
    c.N = N; c.Hin = H; c.Win = H; c.Hg = H; c.Wg = H;
    c.out = out; c.Hout = H; c.Wout = H; c.out_cstride = L.Cout;
    c.act = ACT_LEAKY;
    return run_conv_small(X, L, c);
}

// Conv2DTranspose 5x5/2 + BN + LeakyReLU as four phase convolutions
// The transposed convolutions of split-f16 models on 16x16 / 32x32 input grids in Winograd F(4,3) form (wino3.hip), same switch as the
// stride-1 layers (p2p_ctx_set_winograd).  0 = not for this route, 1 = done, < 0 = error.
// (thresholds re-measured after the K splits of the small launches, tools/time_small.py with P2P_WINO3_MIN / P2P_WINO3O_MIN: per pass at
// 5 / 6 inputs 660 / 669 us against 678 / 685 with up2 / up3 direct; at 9 / 12 / 15 inputs 764 / 802 / 856 us against 833 / 875 / 939 with
// conv4 / up1 direct, at 8 inputs 782 against 760)
constexpr int WINO3_MIN_INPUTS = 5;       // up2 / up3
constexpr int WINO3O_MIN_INPUTS = 9;      // up1 / conv4, eight samples per workgroup
static bool wino3_route() { static const bool on = dev_env("P2P_NO_WINO3") == nullptr; return on; }     // development builds: A/B against the direct phases

// The 8x8-grid layers (wino3o.hip): mode 0 = the transposed convolution up1, mode 1 = the stride-2 convolution conv4.
static int try_wino3o(Ctx& X, const ConvLayer& L, int mode, const float* in, int N, int C, float* out)
{
    if (!L.wino_u || L.prec != PREC_F16X3 || X.wino_mode == P2P_WINOGRAD_OFF || !specialised_kernels() || !wino3_route() || !wino3o_supported(mode, C, L.Cout)) return 0;
    static const int wino3o_min = dev_env("P2P_WINO3O_MIN") ? atoi(dev_env("P2P_WINO3O_MIN")) : WINO3O_MIN_INPUTS;
    if (X.wino_mode != P2P_WINOGRAD_ALWAYS && N < wino3o_min) return 0;
    const size_t b0 = (size_t)N * (mode ? 256 : 64) * C * sizeof(float);
    if (b0 >= 0xFFFFFFF0ull || L.wino_bytes >= 0xFFFFFFF0ull) return 0;
    Wino3oParams p;
    memset(&p, 0, sizeof(p));
    p.in = in; p.in_bytes = (unsigned)b0;
    p.N = N; p.Cin = C; p.Cout = L.Cout;
    p.V = X.cur->act["wv"];
    p.U = L.wino_u; p.scale = L.wino_scale; p.shift = L.shift;
    p.act = ACT_LEAKY; p.alpha = LEAKY;
    p.out = out; p.out_cstride = L.Cout; p.out_coff = 0;
    p.ksplit = 1;
    p.range_acc = X.range_cur;
    if (X.grp && X.grp->models.size() > 1) {
        const GroupCtx& G = *X.grp;
        const int ng = (int)G.models.size();
        if (ng > IGEMM_MAX_GROUPS) { set_error("wino3o: %d object groups exceed IGEMM_MAX_GROUPS", ng); return P2P_ERR_CAPACITY; }
        int unit0 = 0;                  // eight samples per workgroup, every object grouped on its own
        for (int g = 0; g < ng; ++g) {
            const ConvLayer& Lg = G.models[g]->L.at(L.name);
            if (!Lg.wino_u || Lg.prec != PREC_F16X3) { set_error("wino3o: objects of one grouped pass must share a precision"); return P2P_ERR_INVALID_ARG; }
            p.grp[g] = {Lg.wino_u, Lg.wino_scale, Lg.shift, G.start[g], unit0};
            unit0 += (G.start[g + 1] - G.start[g] + 7) / 8;
        }
        p.grp[ng] = {nullptr, nullptr, nullptr, G.start[ng], unit0};
        p.n_groups = ng;
    }
    const int units = wino3o_units(p);
    if (wino3o_v_bytes(mode, units, C) > (size_t)X.max_batch * 64 * 64 * 128 * 2 * sizeof(float)) return 0;      // (many objects of a few detections each: the padded units outgrow the V workspace)
    // conv4 on less than one workgroup per CU: K split over the four parity planes, the four partial sums added after the inverse transform.
    // The split changes the order of the additions, so "always" -- the mode whose bits do not depend on the pass -- splits at every size.
    if (mode == 1 && (X.wino_mode == P2P_WINOGRAD_ALWAYS || units * (L.Cout / 64) < 256)) { p.ksplit = 4; p.partial = X.cur->act["c3"]; }
    hipStream_t st = X.cur->stream;
    const double in_el = (double)N * (mode ? 256 : 64) * C, out_el = (double)N * (mode ? 64 : 256) * L.Cout;
    int rc = timed_launch(X, 21, 0.0, 4.0 * in_el + 6.0 * in_el, [&]() { return launch_wino3o_input(p, mode, st); });
    if (rc) return rc;
    // algorithmic work of the LAYER: conv4 25 MACs per output element and input channel; up1 25 taps over its four phases = 6.25
    rc = timed_launch(X, 20, 2.0 * out_el * (mode ? 25.0 : 6.25) * C, 6.0 * in_el + (double)L.wino_bytes + 4.0 * out_el, [&]() { return launch_wino3o_gemm(p, mode, st); });
    if (rc) return rc;
    if (p.ksplit > 1 && p.n_groups > 1) {
        Conv1Groups G;
        G.n_groups = p.n_groups;
        for (int g = 0; g < p.n_groups; ++g) { G.start[g] = p.grp[g].sample0 * 64; G.w[g] = nullptr; G.scale[g] = p.grp[g].scale; G.shift[g] = p.grp[g].shift; }
        G.start[p.n_groups] = N * 64;
        HIP_TRY(launch_splitk_reduce_groups(p.partial, p.ksplit, N * 64, L.Cout, G, ACT_LEAKY, LEAKY, out, X.range_cur, st));
    } else if (p.ksplit > 1)
        HIP_TRY(launch_splitk_reduce(p.partial, p.ksplit, N * 64, L.Cout, L.wino_scale, L.shift, ACT_LEAKY, LEAKY, out, X.range_cur, st));
    return 1;
}

static int try_wino3(Ctx& X, const ConvLayer& L, const float* in, int N, int H, int C, float* out)
{
    if (H == 8) return try_wino3o(X, L, 0, in, N, C, out);
    if (!L.wino_u || L.prec != PREC_F16X3 || X.wino_mode == P2P_WINOGRAD_OFF || !specialised_kernels() || !wino3_route() || !wino3_supported(H, H, C, L.Cout)) return 0;
    static const int wino3_min = dev_env("P2P_WINO3_MIN") ? atoi(dev_env("P2P_WINO3_MIN")) : WINO3_MIN_INPUTS;
    if (X.wino_mode != P2P_WINOGRAD_ALWAYS && N < wino3_min) return 0;
    const size_t px = (size_t)N * H * H;
    const size_t b0 = px * C * sizeof(float);
    if (b0 >= 0xFFFFFFF0ull || L.wino_bytes >= 0xFFFFFFF0ull) return 0;
    Wino3Params p;
    memset(&p, 0, sizeof(p));
    p.in = in; p.in_bytes = (unsigned)b0;
    p.N = N; p.H = H; p.W = H; p.Cin = C; p.Cout = L.Cout;
    p.V = X.cur->act["wv"];
    p.U = L.wino_u; p.scale = L.wino_scale; p.shift = L.shift;
    p.act = ACT_LEAKY; p.alpha = LEAKY;
    p.out = out; p.out_cstride = L.Cout; p.out_coff = 0;
    p.range_acc = X.range_cur;
    if (X.grp && X.grp->models.size() > 1) {
        const GroupCtx& G = *X.grp;
        const int ng = (int)G.models.size();
        if (ng > IGEMM_MAX_GROUPS) { set_error("deconv_layer: %d object groups exceed IGEMM_MAX_GROUPS", ng); return P2P_ERR_CAPACITY; }
        int unit0 = 0;                  // 16x16 grids: two samples per workgroup, every object paired up on its own
        for (int g = 0; g < ng; ++g) {
            const ConvLayer& Lg = G.models[g]->L.at(L.name);
            if (!Lg.wino_u || Lg.prec != PREC_F16X3) { set_error("deconv_layer: objects of one grouped pass must share a precision"); return P2P_ERR_INVALID_ARG; }
            p.grp[g] = {Lg.wino_u, Lg.wino_scale, Lg.shift, G.start[g], unit0};
            unit0 += (G.start[g + 1] - G.start[g] + 1) / 2;
        }
        p.grp[ng] = {nullptr, nullptr, nullptr, G.start[ng], unit0};
        p.n_groups = ng;
    }
    hipStream_t st = X.cur->stream;
    const double in_el = (double)px * C, out_el = 4.0 * (double)px * L.Cout;
    int rc = timed_launch(X, 21, 0.0, 4.0 * in_el + 6.0 * in_el, [&]() { return launch_wino3_input(p, st); });
    if (rc) return rc;
    // algorithmic work of the LAYER: 25 taps over the four phases = 6.25 MACs per output element and input channel
    rc = timed_launch(X, 20, 2.0 * out_el * 6.25 * C, 6.0 * in_el + (double)L.wino_bytes + 4.0 * out_el, [&]() { return launch_wino3_gemm(p, st); });
    return rc ? rc : 1;
}

static int deconv_layer(Ctx& X, const Model& M, const char* name, const float* in, int N, int H, int C,
                        float* out)
{
    {
        const int rc = try_wino3(X, M.L.at(std::string(name) + "_p0"), in, N, H, C, out);
        if (rc < 0) return rc;
        if (rc == 1) return P2P_OK;
    }
    PreparedConv pc[4];
    bool all_stream = true;
    for (int ph = 0; ph < 4; ++ph) {
        const ConvLayer& L = M.L.at(std::string(name) + "_p" + std::to_string(ph));
        ConvCall c;
        c.s0 = {in, C, C, 0};
        c.N = N; c.Hin = H; c.Win = H; c.Hg = H; c.Wg = H;
        c.out = out; c.Hout = 2 * H; c.Wout = 2 * H; c.os = 2; c.oy = ph >> 1; c.ox = ph & 1;
        c.out_cstride = L.Cout;
        c.act = ACT_LEAKY;
        int rc = prepare_conv(X, L, c, pc[ph]);
        if (rc) return rc;
        all_stream = all_stream && pc[ph].stream;
    }
    if (all_stream) return launch_prepared(X, pc, 4);          // small launches: the four phases in ONE launch (blockIdx.y = phase)
    for (int ph = 0; ph < 4; ++ph) {
        int rc = launch_prepared(X, &pc[ph], 1);
        if (rc) return rc;
    }
    return P2P_OK;
}

// Identity blocks of split-f16 models in one kernel (resblock.hip) when the launch is large enough to fill the chip; small launches (one
// detection at a time) keep the three streaming launches.  The route depends on the batch size, the bits do not (tests/test_resblock_gpu.py).
// Development builds: P2P_NO_FUSED_BLOCK=1 keeps the three-launch route.
static bool fused_blocks() { static const bool on = dev_env("P2P_NO_FUSED_BLOCK") == nullptr; return on && specialised_kernels(); }
// smallest fused launch (workgroups): below it the three streaming launches run (development builds: P2P_FUSED_MIN_WGS)
static int fused_min_wgs() { static const int v = dev_env("P2P_FUSED_MIN_WGS") ? atoi(dev_env("P2P_FUSED_MIN_WGS")) : -1; return v >= 0 ? v : stream_max_wgs() + 1; }

static int run_resblock(const Model& M, Ctx& X, const std::string& n, const float* in, int N, int H, int f1, float* out)
{
    const ConvLayer &a = M.L.at(n + "_2a"), &b = M.L.at(n + "_2b"), &c = M.L.at(n + "_2c");
    const int C = 4 * f1;
    ResBlockParams p;
    memset(&p, 0, sizeof(p));
    p.x = in; p.out = out; p.N = N; p.H = H; p.W = H;
    const size_t xb = (size_t)N * H * H * C * sizeof(float);
    if (xb >= 0xFFFFFFF0ull) { set_error("run_resblock: tensor exceeds the 4 GB buffer-descriptor range (lower max_batch)"); return P2P_ERR_CAPACITY; }
    p.x_bytes = (unsigned)xb;
    p.wa_bytes = (unsigned)((size_t)round_up(a.Cout, 128) * a.K * sizeof(float));
    p.wb_bytes = (unsigned)((size_t)round_up(b.Cout, 128) * b.K * sizeof(float));
    p.wc_bytes = (unsigned)((size_t)round_up(c.Cout, 128) * c.K * sizeof(float));
    p.range_acc = X.range_cur;
    if (X.grp && X.grp->models.size() > 1) {
        const GroupCtx& G = *X.grp;
        const int ng = (int)G.models.size();
        if (ng > IGEMM_MAX_GROUPS) { set_error("run_resblock: %d object groups exceed IGEMM_MAX_GROUPS", ng); return P2P_ERR_CAPACITY; }
        for (int g = 0; g < ng; ++g) {
            const Model& Mg = *G.models[g];
            if (Mg.prec != PREC_F16X3 || !Mg.block_ss.count(n)) { set_error("run_resblock: objects of one grouped pass must share a precision"); return P2P_ERR_INVALID_ARG; }
            p.grp[g] = {Mg.L.at(n + "_2a").w, Mg.L.at(n + "_2b").w, Mg.L.at(n + "_2c").w, Mg.block_ss.at(n), G.start[g], 0, Mg.L.at(n + "_2b").w_frag};
        }
        p.grp[ng] = {nullptr, nullptr, nullptr, nullptr, G.start[ng], 0, nullptr};
        p.n_groups = ng;
    } else {
        p.grp[0] = {a.w, b.w, c.w, M.block_ss.at(n), 0, 0, b.w_frag};
        p.grp[1] = {nullptr, nullptr, nullptr, nullptr, N, 0, nullptr};
        p.n_groups = 1;
    }
    hipStream_t st = X.cur->stream;
    if (X.profiling) {
        const double px = (double)N * H * H;
        Ctx::ProfEvent ev{X.prof_get_event(), X.prof_get_event(), 9, 2.0 * px * ((double)C * f1 + 9.0 * f1 * f1 + (double)f1 * C),
                          4.0 * (2.0 * px * C + (double)C * f1 * 2 + 9.0 * f1 * f1)};      // input once (it is also the residual), output once, the three panels
        if (!ev.a || !ev.b) return P2P_ERR_HIP;
        HIP_TRY(hipEventRecord(ev.a, st));
        HIP_TRY(launch_resblock(p, f1, st));
        HIP_TRY(hipEventRecord(ev.b, st));
        X.prof_pending.push_back(ev);
        return P2P_OK;
    }
    HIP_TRY(launch_resblock(p, f1, st));
    return P2P_OK;
}

// The projection blocks res2a / res3a in the same kernel (resblock.hip, PROJ): H = input grid, Ho = H / stride the output grid.
static int run_resproj(const Model& M, Ctx& X, const std::string& n, const float* in, int N, int H, int Cin, int f1, int stride, float* out)
{
    const ConvLayer &a = M.L.at(n + "_2a"), &b = M.L.at(n + "_2b"), &c = M.L.at(n + "_2c1");
    const int C = 4 * f1, Ho = H / stride;
    ResBlockParams p;
    memset(&p, 0, sizeof(p));
    p.x = in; p.out = out; p.N = N; p.H = Ho; p.W = Ho;
    const size_t xb = (size_t)N * H * H * Cin * sizeof(float), ob = (size_t)N * Ho * Ho * C * sizeof(float);
    if (xb >= 0xFFFFFFF0ull || ob >= 0xFFFFFFF0ull) { set_error("run_resproj: tensor exceeds the 4 GB buffer-descriptor range (lower max_batch)"); return P2P_ERR_CAPACITY; }
    p.x_bytes = (unsigned)xb;
    p.wa_bytes = (unsigned)((size_t)round_up(a.Cout, 128) * a.K * sizeof(float));
    p.wb_bytes = (unsigned)((size_t)round_up(b.Cout, 128) * b.K * sizeof(float));
    p.wc_bytes = (unsigned)((size_t)round_up(c.Cout, 128) * c.K * sizeof(float));
    p.range_acc = X.range_cur;
    if (X.grp && X.grp->models.size() > 1) {
        const GroupCtx& G = *X.grp;
        const int ng = (int)G.models.size();
        if (ng > IGEMM_MAX_GROUPS) { set_error("run_resproj: %d object groups exceed IGEMM_MAX_GROUPS", ng); return P2P_ERR_CAPACITY; }
        for (int g = 0; g < ng; ++g) {
            const Model& Mg = *G.models[g];
            if (Mg.prec != PREC_F16X3 || !Mg.block_ss.count(n)) { set_error("run_resproj: objects of one grouped pass must share a precision"); return P2P_ERR_INVALID_ARG; }
            p.grp[g] = {Mg.L.at(n + "_2a").w, Mg.L.at(n + "_2b").w, Mg.L.at(n + "_2c1").w, Mg.block_ss.at(n), G.start[g], 0, Mg.L.at(n + "_2b").w_frag};
        }
        p.grp[ng] = {nullptr, nullptr, nullptr, nullptr, G.start[ng], 0, nullptr};
        p.n_groups = ng;
    } else {
        p.grp[0] = {a.w, b.w, c.w, M.block_ss.at(n), 0, 0, b.w_frag};
        p.grp[1] = {nullptr, nullptr, nullptr, nullptr, N, 0, nullptr};
        p.n_groups = 1;
    }
    hipStream_t st = X.cur->stream;
    if (X.profiling) {
        const double px = (double)N * Ho * Ho;
        Ctx::ProfEvent ev{X.prof_get_event(), X.prof_get_event(), 9, 2.0 * px * ((double)Cin * f1 + 9.0 * f1 * f1 + (double)(f1 + Cin) * C),
                          4.0 * (px * Cin + px * C + (double)Cin * f1 + 9.0 * f1 * f1 + (double)(f1 + Cin) * C)};      // the sampled input pixels once, the output once, the panels
        if (!ev.a || !ev.b) return P2P_ERR_HIP;
        HIP_TRY(hipEventRecord(ev.a, st));
        HIP_TRY(launch_resproj(p, f1, st));
        HIP_TRY(hipEventRecord(ev.b, st));
        X.prof_pending.push_back(ev);
        return P2P_OK;
    }
    HIP_TRY(launch_resproj(p, f1, st));
    return P2P_OK;
}

static bool fused_proj_blocks() { static const bool on = dev_env("P2P_NO_FUSED_PROJ") == nullptr; return on; }      // development builds: res2a / res3a on three launches

static int res_block(const Model& M, Ctx& X, const std::string& n, const float* in, int N, int H,
                     int Cin, int f1, int stride, bool shortcut, float* out)
{
    int rc;
    const int Ho = H / stride;
    if (shortcut && M.prec == PREC_F16X3 && M.L.count(n + "_2c1") && fused_blocks() && fused_proj_blocks() && M.block_ss.count(n) &&
        ((f1 == 64 && Cin == 64 && stride == 1) || (f1 == 128 && Cin == 256 && stride == 2)) && resblock_supported(f1, Ho, Ho) &&
        resblock_grid(f1, N, Ho, Ho) >= fused_min_wgs())
        return run_resproj(M, X, n, in, N, H, Cin, f1, stride, out);
    if (!shortcut && stride == 1 && M.prec == PREC_F16X3 && Cin == 4 * f1 && fused_blocks() && M.block_ss.count(n) && resblock_supported(f1, H, H) &&
        resblock_grid(f1, N, H, H) >= fused_min_wgs())
        return run_resblock(M, X, n, in, N, H, f1, out);
    float* ta = X.cur->act["t_a"];
    float* tb = X.cur->act["t_b"];
    if ((rc = conv_layer(X, M.L.at(n + "_2a"), in, N, H, H, Cin, stride, ta, ACT_RELU))) return rc;
    if ((rc = conv_layer(X, M.L.at(n + "_2b"), ta, N, Ho, Ho, f1, 1, tb, ACT_RELU))) return rc;
    const float* res = in;
    if (shortcut && M.L.count(n + "_2c1")) {
        // projection shortcut folded into the block's last convolution (pack_merged_shortcut): K = [t_b || x], x read on
        // its own grid at the block's stride
        const ConvLayer& L = M.L.at(n + "_2c1");
        ConvCall c;
        c.s0 = {tb, f1, f1, 0};
        c.s1 = {in, Cin, Cin, 0};
        if (stride != 1) { c.s1_Hin = H; c.s1_Win = H; c.s1_stride = stride; }
        c.N = N; c.Hin = c.Win = c.Hg = c.Wg = Ho;
        c.out = out; c.Hout = c.Wout = Ho; c.out_cstride = L.Cout;
        c.act = ACT_RELU;
        return run_conv(X, L, c);
    }
    if (shortcut) {
        if ((rc = conv_layer(X, M.L.at(n + "_1"), in, N, H, H, Cin, stride, X.cur->act["sc"], ACT_NONE))) return rc;
        res = X.cur->act["sc"];
    }
    return conv_layer(X, M.L.at(n + "_2c"), tb, N, Ho, Ho, f1, 1, out, ACT_RELU, res);
}

#ifdef P2P_TIMING_SWITCHES      // A/B builds only (tools/ab_build.sh model.hip -DP2P_TIMING_SWITCHES): timing experiments, results are garbage
#define dev_part() (X.dev_part)      // per context, from P2P_DEV_PART at p2p_ctx_create: 1 = ResNet front only, 2 = everything after it only
#endif

// x_dev [n,128,128,3] -> xyzp_dev [n,128,128,4]; n <= ctx.max_batch
int forward_chunk(Ctx& X, const Model& M, const float* x, int n, float* xyzp)
{
    int rc;
    hipStream_t st = X.cur->stream;
    auto& A = X.cur->act;
    const int n_grp = X.grp ? (int)X.grp->models.size() : 1;
    auto grp_model = [&](int g) -> const Model& { return X.grp ? *X.grp->models[g] : M; };
    auto g0 = [&](int g) -> int { return X.grp ? X.grp->start[g] : (g == 0 ? 0 : n); };
    const float *s1, *s2, *s3;      // skip tensors and their pixel strides / channel offsets
    int s1_stride, s1_off, s1_C, s2_stride, s2_off, s3_stride, s3_off;
    if (M.backbone == P2P_BACKBONE_RESNET50) {
#ifdef P2P_TIMING_SWITCHES
        if (dev_part() == 2) goto after_front;
#endif
        if (M.L.at("conv1").prec == PREC_F16X3) {       // matrix-core first layer: one launch, per-sample panel lookup
            if (n_grp > IGEMM_MAX_GROUPS) { set_error("forward: %d object groups exceed IGEMM_MAX_GROUPS", n_grp); return P2P_ERR_CAPACITY; }
            Conv1Groups G;
            G.n_groups = n_grp;
            for (int g = 0; g < n_grp; ++g) {
                const ConvLayer& c1 = grp_model(g).L.at("conv1");
                if (c1.prec != PREC_F16X3) { set_error("forward: objects of one grouped pass must share a precision"); return P2P_ERR_INVALID_ARG; }
                G.start[g] = g0(g); G.w[g] = c1.w; G.scale[g] = c1.scale; G.shift[g] = c1.shift;
            }
            G.start[n_grp] = n;
            HIP_TRY(launch_conv1_f16x3(x, n, 7, 64, G, ACT_RELU, LEAKY, A["f1"], A["p1"], X.range_cur, st));      // + the max-pool (f1: skip channels)
        } else {
        for (int g = 0; g < n_grp; ++g) {      // VALU first layer: one launch per object (tiny)
            const ConvLayer& c1 = grp_model(g).L.at("conv1");
            HIP_TRY(launch_conv_first(x + (size_t)g0(g) * 49152, g0(g + 1) - g0(g), 128, 128, c1.w, 7, 2, 3, 64, c1.scale, c1.shift,
                                      ACT_RELU, LEAKY, A["f1"] + (size_t)g0(g) * 64 * 64 * 64, 64, 64, st));
        }
        HIP_TRY(launch_maxpool3s2(A["f1"], n, 64, 64, 64, A["p1"], st));
        }
        if ((rc = res_block(M, X, "res2a", A["p1"], n, 32, 64, 64, 1, true, A["o_a"]))) return rc;
        if ((rc = res_block(M, X, "res2b", A["o_a"], n, 32, 256, 64, 1, false, A["o_b"]))) return rc;
        if ((rc = res_block(M, X, "res2c", A["o_b"], n, 32, 256, 64, 1, false, A["f2"]))) return rc;
        if ((rc = res_block(M, X, "res3a", A["f2"], n, 32, 256, 128, 2, true, A["o_a"]))) return rc;
        if ((rc = res_block(M, X, "res3b", A["o_a"], n, 16, 512, 128, 1, false, A["o_b"]))) return rc;
        if ((rc = res_block(M, X, "res3c", A["o_b"], n, 16, 512, 128, 1, false, A["o_a"]))) return rc;
        if ((rc = res_block(M, X, "res3d", A["o_a"], n, 16, 512, 128, 1, false, A["f3"]))) return rc;
#ifdef P2P_TIMING_SWITCHES
        if (dev_part() == 1) return P2P_OK;
    after_front:
#endif
        if ((rc = try_wino3o(X, M.L.at("conv4"), 1, A["f3"], n, 512, A["f4"])) < 0) return rc;
        if (rc == 0 && (rc = conv_layer(X, M.L.at("conv4"), A["f3"], n, 16, 16, 512, 2, A["f4"], ACT_LEAKY))) return rc;
        // ae_model.py:186-188: f1[..., :32], f2[..., :128], f3[..., :128]
        s1 = A["f1"]; s1_stride = 64; s1_off = 0; s1_C = 32;
        s2 = A["f2"]; s2_stride = 256; s2_off = 0;
        s3 = A["f3"]; s3_stride = 512; s3_off = 0;
    } else {
        // ae_model.py:74-106: each level = two parallel 5x5/2 convs concatenated [_1 || _2];
        // the skip is the _2 half, i.e. the upper channels of the merged output.
        if (M.L.at("conv1").prec == PREC_F16X3) {       // matrix-core first layer (conv1.hip): one launch, per-sample panel lookup
            if (n_grp > IGEMM_MAX_GROUPS) { set_error("forward: %d object groups exceed IGEMM_MAX_GROUPS", n_grp); return P2P_ERR_CAPACITY; }
            Conv1Groups G;
            G.n_groups = n_grp;
            for (int g = 0; g < n_grp; ++g) {
                const ConvLayer& c1 = grp_model(g).L.at("conv1");
                if (c1.prec != PREC_F16X3) { set_error("forward: objects of one grouped pass must share a precision"); return P2P_ERR_INVALID_ARG; }
                G.start[g] = g0(g); G.w[g] = c1.w; G.scale[g] = c1.scale; G.shift[g] = c1.shift;
            }
            G.start[n_grp] = n;
            HIP_TRY(launch_conv1_f16x3(x, n, 5, 128, G, ACT_LEAKY, LEAKY, A["f1"], nullptr, X.range_cur, st));
        } else
        for (int g = 0; g < n_grp; ++g) {
            const ConvLayer& c1 = grp_model(g).L.at("conv1");
            HIP_TRY(launch_conv_first(x + (size_t)g0(g) * 49152, g0(g + 1) - g0(g), 128, 128, c1.w, 5, 2, 1, 128, c1.scale, c1.shift,
                                      ACT_LEAKY, LEAKY, A["f1"] + (size_t)g0(g) * 64 * 64 * 128, 64, 64, st));
        }
        if ((rc = conv_layer(X, M.L.at("conv2"), A["f1"], n, 64, 64, 128, 2, A["f2"], ACT_LEAKY))) return rc;
        if ((rc = conv_layer(X, M.L.at("conv3"), A["f2"], n, 32, 32, 256, 2, A["f3"], ACT_LEAKY))) return rc;
        if ((rc = try_wino3o(X, M.L.at("conv4"), 1, A["f3"], n, 256, A["f4"])) < 0) return rc;
        if (rc == 0 && (rc = conv_layer(X, M.L.at("conv4"), A["f3"], n, 16, 16, 256, 2, A["f4"], ACT_LEAKY))) return rc;
        s1 = A["f1"]; s1_stride = 128; s1_off = 64; s1_C = 64;
        s2 = A["f2"]; s2_stride = 256; s2_off = 128;
        s3 = A["f3"]; s3_stride = 256; s3_off = 128;
    }
    // Flatten (HWC order) + Dense(256): split-K GEMM over K = 32768, then bias in the reduce
    {
        const ConvLayer& L = M.L.at("dense_enc");
        ConvCall c;
        c.s0 = {A["f4"], 32768, 32768, 0};
        c.N = n; c.Hin = c.Win = c.Hg = c.Wg = 1;
        c.out = A["enc"]; c.Hout = c.Wout = 1; c.out_cstride = 256;
        c.ksplit = 32; c.partial = A["part"];
        if ((rc = run_conv(X, L, c))) return rc;
        if (n_grp == 1) {
            HIP_TRY(launch_splitk_reduce(A["part"], 32, n, 256, L.scale, L.shift, ACT_NONE, LEAKY, A["enc"], M.prec == PREC_F16X3 ? X.range_cur : nullptr, st));
        } else {
            // per-object bias: every row is reduced with its own object's shift, one launch for the batch
            if (n_grp > IGEMM_MAX_GROUPS) { set_error("forward: %d object groups exceed IGEMM_MAX_GROUPS", n_grp); return P2P_ERR_CAPACITY; }
            Conv1Groups G;
            G.n_groups = n_grp;
            for (int g = 0; g < n_grp; ++g) {
                const ConvLayer& Lg = grp_model(g).L.at("dense_enc");
                G.start[g] = g0(g); G.w[g] = nullptr; G.scale[g] = Lg.scale; G.shift[g] = Lg.shift;
            }
            G.start[n_grp] = n;
            HIP_TRY(launch_splitk_reduce_groups(A["part"], 32, n, 256, G, ACT_NONE, LEAKY, A["enc"], M.prec == PREC_F16X3 ? X.range_cur : nullptr, st));
        }
    }
    {
        const ConvLayer& L = M.L.at("dense_dec");     // Dense(8*8*256) + Reshape((8,8,-1))
        ConvCall c;
        c.s0 = {A["enc"], 256, 256, 0};
        c.N = n; c.Hin = c.Win = c.Hg = c.Wg = 1;
        c.out = A["dd"]; c.Hout = c.Wout = 1; c.out_cstride = 16384;
        if ((rc = run_conv(X, L, c))) return rc;
    }
    if ((rc = deconv_layer(X, M, "up1", A["dd"], n, 8, 256, A["u1"]))) return rc;
    if ((rc = concat_conv(X, M.L.at("deconv1"), A["u1"], 256, s3, 128, s3_stride, s3_off, n, 16, A["c1"]))) return rc;
    if ((rc = deconv_layer(X, M, "up2", A["c1"], n, 16, 256, A["u2"]))) return rc;
    if ((rc = concat_conv(X, M.L.at("deconv2"), A["u2"], 128, s2, 128, s2_stride, s2_off, n, 32, A["c2"]))) return rc;
    if ((rc = deconv_layer(X, M, "up3", A["c2"], n, 32, 256, A["u3"]))) return rc;
    if ((rc = concat_conv(X, M.L.at("deconv3"), A["u3"], 64, s1, s1_C, s1_stride, s1_off, n, 64, A["c3"]))) return rc;
    {
        ConvCall c;
        c.s0 = {A["c3"], 128, 128, 0};
        c.N = n; c.Hin = c.Win = c.Hg = c.Wg = 64;
        c.out = xyzp; c.Hout = c.Wout = 128; c.os = 2; c.out_cstride = 4;
        c.mode = EPI_HEAD;
        c.algo_macs = (double)n * 64 * 64 * 25 * 128 * 4;   // two Conv2DTranspose heads, 5x5 taps, 3+1 channels
        if ((rc = run_conv(X, M.L.at("heads"), c))) return rc;
    }
    return P2P_OK;
}

int forward_async(Ctx& X, const Model& M, const float* x_dev, int n, float* xyzp_dev)
{
    int rc = X.ensure_workspace();
    if (rc) return rc;
    for (int i = 0; i < n; i += X.max_batch) {
        const int c = std::min(X.max_batch, n - i);
        rc = forward_chunk(X, M, x_dev + (size_t)i * 128 * 128 * 3, c, xyzp_dev + (size_t)i * 128 * 128 * 4);
        if (rc) return rc;
    }
    return P2P_OK;
}

thread_local ProfHook g_prof_hook = {nullptr, nullptr, nullptr};
static thread_local Ctx::ProfEvent g_open_ev;
static void prof_hook_begin(void* c, int slot, hipStream_t s)
{
    Ctx& X = *static_cast<Ctx*>(c);
    g_open_ev = Ctx::ProfEvent{X.prof_get_event(), X.prof_get_event(), slot, 0.0, 0.0};
    if (g_open_ev.a) (void)hipEventRecord(g_open_ev.a, s);
}
static void prof_hook_end(void* c, hipStream_t s)
{
    Ctx& X = *static_cast<Ctx*>(c);
    if (!g_open_ev.a || !g_open_ev.b) return;
    (void)hipEventRecord(g_open_ev.b, s);
    X.prof_pending.push_back(g_open_ev);
}
ProfHookGuard::ProfHookGuard(Ctx& X) : prev(g_prof_hook)
{
    if (X.profiling) g_prof_hook = ProfHook{&X, prof_hook_begin, prof_hook_end};
}
ProfHookGuard::~ProfHookGuard() { g_prof_hook = prev; }

hipEvent_t Ctx::prof_get_event()
{
    if (!prof_pool.empty()) { hipEvent_t e = prof_pool.back(); prof_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { set_error("hipEventCreate failed"); return nullptr; }
    return e;
}

int Ctx::prof_harvest()
{
    // events are recorded on whichever lane ran the pass: wait for all of them (an event still pending
    // makes hipEventElapsedTime fail with hipErrorNotReady)
    for (Lane& ln : lane)
        if (ln.stream) HIP_TRY(hipStreamSynchronize(ln.stream));
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipDeviceSynchronize());      // the glue / PnP slots are recorded on the pipeline's tail streams
    for (const ProfEvent& ev : prof_pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
        prof_stats[ev.cfg].launches += 1;
        prof_stats[ev.cfg].total_ms += ms;
        prof_stats[ev.cfg].algo_flops += ev.flops;
        prof_stats[ev.cfg].algo_bytes += ev.bytes;
        prof_pool.push_back(ev.a);
        prof_pool.push_back(ev.b);
    }
    prof_pending.clear();
    return P2P_OK;
}

int forward_grouped(Ctx& X, const std::vector<const Model*>& models, const std::vector<int>& counts, const float* x_dev,
                    float* xyzp_dev)
{
    int rc = X.ensure_workspace();
    if (rc) return rc;
    for (const Model* m : models)
        if (m->backbone != models[0]->backbone) { set_error("forward_grouped: mixed backbones"); return P2P_ERR_INVALID_ARG; }
    // chunks of <= max_batch samples and <= IGEMM_MAX_GROUPS objects; an object may straddle two chunks
    size_t g = 0;
    int used = 0, done = 0;       // samples of group g already consumed; samples enqueued so far
    while (g < models.size()) {
        GroupCtx G;
        G.start.push_back(0);
        int n = 0;
        while (g < models.size() && n < X.max_batch && (int)G.models.size() < IGEMM_MAX_GROUPS) {
            const int take = std::min(counts[g] - used, X.max_batch - n);
            if (take > 0) {
                G.models.push_back(models[g]);
                n += take;
                G.start.push_back(n);
            }
            used += take;
            if (used >= counts[g]) { ++g; used = 0; }
        }
        if (n == 0) break;
        X.grp = &G;
        rc = forward_chunk(X, *G.models[0], x_dev + (size_t)done * 128 * 128 * 3, n, xyzp_dev + (size_t)done * 128 * 128 * 4);
        X.grp = nullptr;
        if (rc) return rc;
        done += n;
    }
    return P2P_OK;
}

Model::~Model()
{
    for (auto& kv : L) free_layer(kv.second);
    for (auto& kv : block_ss) hipFree(kv.second);
    delete twin;
}

Ctx::~Ctx()
{
    for (Lane& ln : lane) {
        for (auto& kv : ln.act) hipFree(kv.second);
        if (ln.done) hipEventDestroy(ln.done);
        if (ln.stream && ln.stream != stream) hipStreamDestroy(ln.stream);
    }
    if (fork) hipEventDestroy(fork);
    if (range_words) hipFree(range_words);
    if (x_stage) hipFree(x_stage);
    if (xyzp_stage) hipFree(xyzp_stage);
    if (xyz_stage) hipFree(xyz_stage);
    if (prob_stage) hipFree(prob_stage);
    free_pipeline();
    for (auto& ev : prof_pending) { hipEventDestroy(ev.a); hipEventDestroy(ev.b); }
    for (auto e : prof_pool) hipEventDestroy(e);
    if (stream) hipStreamDestroy(stream);
}

}  // namespace p2p

// ==========================================================================================
// C ABI
// ==========================================================================================
using namespace p2p;

extern "C" {

int p2p_abi_version(void) { return P2P_ABI_VERSION; }

int p2p_abi_sizeof(int which)
{
    switch (which) {
    case 0: return (int)sizeof(p2p_tensor);
    case 1: return (int)sizeof(p2p_image);
    case 2: return (int)sizeof(p2p_object);
    case 3: return (int)sizeof(p2p_detection);
    case 4: return (int)sizeof(p2p_pose);
    case 5: return (int)sizeof(p2p_est_pose_opts);
    case 6: return (int)sizeof(p2p_kernel_stats);
    default: return -1;
    }
}
const char* p2p_last_error(void) { return get_error(); }

int p2p_device_count(int* count)
{
    if (!count) { set_error("p2p_device_count: null argument"); return P2P_ERR_INVALID_ARG; }
    HIP_TRY(hipGetDeviceCount(count));
    return P2P_OK;
}

int p2p_ctx_create(int device, int max_batch, p2p_ctx** out)
{
    if (!out || max_batch < 1) { set_error("p2p_ctx_create: bad arguments"); return P2P_ERR_INVALID_ARG; }
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) { set_error("p2p_ctx_create: device %d out of range (%d devices)", device, n); return P2P_ERR_INVALID_ARG; }
    HIP_TRY(hipSetDevice(device));
    Ctx* c = new Ctx();
    c->device = device;
    c->max_batch = max_batch;
    hipError_t e;
#ifdef P2P_TIMING_SWITCHES
    c->dev_part = getenv("P2P_DEV_PART") ? atoi(getenv("P2P_DEV_PART")) : 0;
    if (const char* cus = getenv("P2P_DEV_CUS")) {      // "lo:hi": the context's stream may only use CUs [lo, hi) (mask bits; KFD deals them round-robin over the XCDs)
        int lo = 0, hi = 256;
        sscanf(cus, "%d:%d", &lo, &hi);
        uint32_t mask[8] = {0};
        for (int i = lo; i < hi && i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
        e = hipExtStreamCreateWithCUMask(&c->stream, 8, mask);
    } else
#endif
    e = hipStreamCreate(&c->stream);
    if (e != hipSuccess) { set_error("hipStreamCreate failed: %s", hipGetErrorString(e)); delete c; return P2P_ERR_HIP; }
    if ((e = hipMalloc((void**)&c->range_words, 8 * sizeof(unsigned))) != hipSuccess || (e = hipMemset(c->range_words, 0, 8 * sizeof(unsigned))) != hipSuccess) {
        set_error("p2p_ctx_create: %s", hipGetErrorString(e));
        delete c;
        return P2P_ERR_HIP;
    }
    c->range_cur = c->range_words;
    *out = reinterpret_cast<p2p_ctx*>(c);
    return P2P_OK;
}

void p2p_ctx_destroy(p2p_ctx* ctx)
{
    if (!ctx) return;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    delete c;
}

int p2p_ctx_synchronize(p2p_ctx* ctx)
{
    if (!ctx) { set_error("null ctx"); return P2P_ERR_INVALID_ARG; }
    HIP_TRY(hipStreamSynchronize(reinterpret_cast<Ctx*>(ctx)->stream));
    return P2P_OK;
}

void* p2p_ctx_stream(p2p_ctx* ctx) { return ctx ? (void*)reinterpret_cast<Ctx*>(ctx)->stream : nullptr; }

int p2p_profile_enable(p2p_ctx* ctx, int on)
{
    if (!ctx) { set_error("null ctx"); return P2P_ERR_INVALID_ARG; }
    reinterpret_cast<Ctx*>(ctx)->profiling = on != 0;
    return P2P_OK;
}

int p2p_profile_read(p2p_ctx* ctx, p2p_kernel_stats* stats, int reset)
{
    if (!ctx || !stats) { set_error("p2p_profile_read: bad arguments"); return P2P_ERR_INVALID_ARG; }
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIP_TRY(hipSetDevice(c->device));
    int rc = c->prof_harvest();
    if (rc) return rc;
    for (int i = 0; i < P2P_PROFILE_SLOTS; ++i) stats[i] = c->prof_stats[i];
    if (reset) for (int i = 0; i < P2P_PROFILE_SLOTS; ++i) c->prof_stats[i] = p2p_kernel_stats{};
    return P2P_OK;
}

int p2p_model_create(p2p_ctx* ctx, const p2p_tensor* tensors, int n_tensors, int backbone, p2p_model** out)
{
    return p2p_model_create_ex(ctx, tensors, n_tensors, backbone, P2P_PREC_DEFAULT, out);
}

int p2p_model_create_ex(p2p_ctx* ctx, const p2p_tensor* tensors, int n_tensors, int backbone, int precision, p2p_model** out)
{
    if (!ctx || !tensors || !out || n_tensors <= 0) { set_error("p2p_model_create: bad arguments"); return P2P_ERR_INVALID_ARG; }
    if (precision != P2P_PREC_F32 && precision != P2P_PREC_F16X3 && precision != P2P_PREC_AUTO) { set_error("p2p_model_create: unknown precision %d", precision); return P2P_ERR_INVALID_ARG; }
    *out = nullptr;
    if (backbone != P2P_BACKBONE_PAPER && backbone != P2P_BACKBONE_RESNET50) {
        // the reference silently leaves generator_train undefined here (recognition.py:21-26)
        set_error("p2p_model_create: unknown backbone %d", backbone);
        return P2P_ERR_INVALID_ARG;
    }
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIP_TRY(hipSetDevice(c->device));
    TensorMap T;
    for (int i = 0; i < n_tensors; ++i) {
        if (!tensors[i].name || !tensors[i].data) { set_error("p2p_model_create: tensor %d is null", i); return P2P_ERR_INVALID_ARG; }
        T.m[tensors[i].name] = {tensors[i].data, tensors[i].numel};
    }
    Model* m = new Model();
    m->backbone = backbone;
    m->prec = precision == P2P_PREC_F32 ? PREC_F32 : PREC_F16X3;
    m->device = c->device;
    int rc = build_model(T, *m);
    if (!rc) rc = pack_block_ss(*m);
    if (rc) { delete m; return rc; }
    for (auto& kv : m->L) kv.second.name = kv.first;
    if (precision == P2P_PREC_AUTO) {          // the strict-fp32 twin the object falls back to after an operand-range event
        Model* t = new Model();
        t->backbone = backbone; t->prec = PREC_F32; t->device = c->device;
        m->twin = t;
        if ((rc = build_model(T, *t))) { delete m; return rc; }
        for (auto& kv : t->L) kv.second.name = kv.first;
    }
    *out = reinterpret_cast<p2p_model*>(m);
    return P2P_OK;
}

void p2p_model_destroy(p2p_model* model)
{
    if (!model) return;
    Model* m = reinterpret_cast<Model*>(model);
    hipSetDevice(m->device);
    delete m;
}

int p2p_forward_async(p2p_ctx* ctx, const p2p_model* model, const float* x_dev, int n, float* xyzp_dev)
{
    if (!ctx || !model || !x_dev || !xyzp_dev || n < 0) { set_error("p2p_forward_async: bad arguments"); return P2P_ERR_INVALID_ARG; }
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIP_TRY(hipSetDevice(c->device));
    c->range_cur = c->range_words;
    return forward_async(*c, *reinterpret_cast<const Model*>(model)->effective(), x_dev, n, xyzp_dev);
}

int p2p_ctx_range_event(p2p_ctx* ctx, float* max_abs)
{
    if (!ctx || !max_abs) { set_error("p2p_ctx_range_event: bad arguments"); return P2P_ERR_INVALID_ARG; }
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return c->range_read(0, max_abs);
}

int p2p_ctx_set_winograd(p2p_ctx* ctx, int mode)
{
    if (!ctx || mode < P2P_WINOGRAD_OFF || mode > P2P_WINOGRAD_ALWAYS) { set_error("p2p_ctx_set_winograd: bad arguments"); return P2P_ERR_INVALID_ARG; }
    reinterpret_cast<Ctx*>(ctx)->wino_mode = mode;
    return P2P_OK;
}

int p2p_model_precision(const p2p_model* model)
{
    if (!model) return P2P_ERR_INVALID_ARG;
    return reinterpret_cast<const Model*>(model)->effective()->prec == PREC_F32 ? P2P_PREC_F32 : P2P_PREC_F16X3;
}

int p2p_predict(p2p_ctx* ctx, const p2p_model* model, const float* x, int n, float* xyz, float* prob, int mem)
{
    if (!ctx || !model || n < 0 || (n > 0 && (!x || !xyz || !prob))) { set_error("p2p_predict: bad arguments"); return P2P_ERR_INVALID_ARG; }
    if (mem != P2P_MEM_HOST && mem != P2P_MEM_DEVICE) { set_error("p2p_predict: bad mem flag %d", mem); return P2P_ERR_INVALID_ARG; }
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    const Model& m0 = *reinterpret_cast<const Model*>(model);
    HIP_TRY(hipSetDevice(c->device));
    int rc = c->ensure_workspace();
    if (rc) return rc;
    const size_t px = 128 * 128;
    c->range_cur = c->range_words;
    for (int i = 0; i < n; i += c->max_batch) {
        const Model& m = *m0.effective();
        const int k = std::min(c->max_batch, n - i);
        const float* xin = x + (size_t)i * px * 3;
        if (mem == P2P_MEM_HOST) {
            HIP_TRY(hipMemcpyAsync(c->x_stage, xin, (size_t)k * px * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
            xin = c->x_stage;
        }
        if ((rc = forward_chunk(*c, m, xin, k, c->xyzp_stage))) return rc;
        float* oxyz = mem == P2P_MEM_HOST ? c->xyz_stage : xyz + (size_t)i * px * 3;
        float* oprob = mem == P2P_MEM_HOST ? c->prob_stage : prob + (size_t)i * px;
        HIP_TRY(launch_split_xyzp(c->xyzp_stage, (int64_t)k * px, oxyz, oprob, c->stream));
        if (mem == P2P_MEM_HOST) {
            HIP_TRY(hipMemcpyAsync(xyz + (size_t)i * px * 3, oxyz, (size_t)k * px * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(prob + (size_t)i * px, oprob, (size_t)k * px * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (m.prec == PREC_F16X3) {          // operand-range guard of the split-f16 arithmetic
            float amax = 0.f;
            if ((rc = c->range_read(0, &amax))) return rc;
            if (amax > 0.f) {
                if (m0.twin) { m0.use_twin = true; i -= c->max_batch; continue; }      // P2P_PREC_AUTO: this chunk again, and all later ones, in fp32
                set_error("p2p_predict: an activation of magnitude %g exceeds the split-f16 operand range (%g): create the model with "
                          "P2P_PREC_F32 or P2P_PREC_AUTO", (double)amax, (double)RANGE_LIMIT);
                return P2P_ERR_RANGE;
            }
        }
    }
    return P2P_OK;
}

}  // extern "C"
