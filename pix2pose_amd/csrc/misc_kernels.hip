// Small non-GEMM kernels of the generator forward pass (gfx950).
#include "kernels.h"

namespace p2p {

// ------------------------------------------------------------------------------------------
// First-layer direct convolution (Cin = 3): conv1 7x7/2 'valid' after ZeroPadding2D(3)
// (reference resnet50_mod.py:200-203) and conv1_1/conv1_2 5x5/2 'SAME' of the paper encoder
// (reference ae_model.py:74-81, both branches merged into one 128-channel convolution).
// K = kh*kw*3 (147 / 75) is too ragged for the 32-wide K-steps of the MFMA kernel and the
// layer is <1 % of the MACs, so it runs on the VALU: one workgroup = 8x8 output pixels of one
// image, all Cout channels; weights and the input patch live in LDS; a wave = 64 pixels of
// one channel group, so weight reads are LDS broadcasts.
template <int KH, int STRIDE, int COUT>
__global__ __launch_bounds__(256) void conv_first_kernel(const float* __restrict__ x, int H, int W,
                                                         const float* __restrict__ wp, int pad,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int act, float alpha,
                                                         float* __restrict__ out, int Ho, int Wo)
{
    constexpr int TP = 8;                          // output tile side
    constexpr int PW = (TP - 1) * STRIDE + KH;     // input patch side
    constexpr int KK = KH * KH * 3;
    constexpr int CPT = COUT / 4;                  // couts per thread
    __shared__ __attribute__((aligned(16))) float ws[KK * COUT];
    __shared__ float patch[PW * PW * 3];

    const int tid = threadIdx.x;
    const int tiles_x = Wo / TP;
    const int tiles_per_img = tiles_x * (Ho / TP);
    const int n = blockIdx.x / tiles_per_img;
    const int tt = blockIdx.x - n * tiles_per_img;
    const int ty = tt / tiles_x, tx = tt - ty * tiles_x;

    for (int i = tid * 4; i < KK * COUT; i += 1024)
        *reinterpret_cast<float4*>(ws + i) = *reinterpret_cast<const float4*>(wp + i);
    const int iy0 = ty * TP * STRIDE - pad, ix0 = tx * TP * STRIDE - pad;
    for (int i = tid; i < PW * PW * 3; i += 256) {
        const int c = i % 3;
        const int px = (i / 3) % PW;
        const int py = i / (3 * PW);
        const int iy = iy0 + py, ix = ix0 + px;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = x[((size_t)(n * H + iy) * W + ix) * 3 + c];
        patch[i] = v;
    }
    __syncthreads();

    const int pix = tid & 63, cg = tid >> 6;
    const int oy = pix >> 3, ox = pix & 7;
    float acc[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) acc[c] = 0.f;
    for (int kh = 0; kh < KH; ++kh)
        for (int kw = 0; kw < KH; ++kw) {
            const float* pp = patch + ((oy * STRIDE + kh) * PW + ox * STRIDE + kw) * 3;
            const float* wq = ws + ((kh * KH + kw) * 3) * COUT + cg * CPT;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) {
                const float xv = pp[ci];
#pragma unroll
                for (int c = 0; c < CPT; c += 4) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wq + ci * COUT + c);
                    acc[c] = fmaf(xv, w4.x, acc[c]);
                    acc[c + 1] = fmaf(xv, w4.y, acc[c + 1]);
                    acc[c + 2] = fmaf(xv, w4.z, acc[c + 2]);
                    acc[c + 3] = fmaf(xv, w4.w, acc[c + 3]);
                }
            }
        }
    float* op = out + ((size_t)(n * Ho + ty * TP + oy) * Wo + tx * TP + ox) * COUT + cg * CPT;
#pragma unroll
    for (int c = 0; c < CPT; c += 4) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = cg * CPT + c + e;
            float t = fmaf(acc[c + e], scale[co], shift[co]);
            if (act == ACT_RELU) t = relu_nan(t);
            else if (act == ACT_LEAKY) t = t > 0.f ? t : t * alpha;
            v[e] = t;
        }
        *reinterpret_cast<float4*>(op + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

hipError_t launch_conv_first(const float* x, int N, int H, int W, const float* w_packed, int KH, int stride,
                             int pad, int Cout, const float* scale, const float* shift, int act, float alpha,
                             float* out, int Ho, int Wo, hipStream_t s)
{
    if (Ho % 8 || Wo % 8 || stride != 2) return hipErrorInvalidValue;
    dim3 grid(N * (Ho / 8) * (Wo / 8));
    if (KH == 7 && Cout == 64)
        hipLaunchKernelGGL((conv_first_kernel<7, 2, 64>), grid, dim3(256), 0, s, x, H, W, w_packed, pad, scale, shift,
                           act, alpha, out, Ho, Wo);
    else if (KH == 5 && Cout == 128)
        hipLaunchKernelGGL((conv_first_kernel<5, 2, 128>), grid, dim3(256), 0, s, x, H, W, w_packed, pad, scale, shift,
                           act, alpha, out, Ho, Wo);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// MaxPooling2D(3x3, stride 2, padding='same') -- reference resnet50_mod.py:204 (a modification of
// stock ResNet-50: 'same' => TF pads 0 before / 1 after on an even input; padded cells ignored).
__global__ void maxpool3s2_kernel(const float* __restrict__ x, int N, int H, int W, int C4,
                                  float* __restrict__ out, int Ho, int Wo)
{
    const size_t total = (size_t)N * Ho * Wo * C4;
    // Workgroup b runs on XCD b % 8 and each XCD has its own L2: give every XCD a contiguous run of output (whole images), so that
    // the input rows two output rows share are re-read from the SAME L2 (in launch order the two rows sat on different XCDs and
    // the kernel fetched 1.5x its input).
    const unsigned nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const unsigned lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    for (size_t i = (size_t)lb * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t r = i / C4;
        const int ow = (int)(r % Wo); r /= Wo;
        const int oh = (int)(r % Ho);
        const int n = (int)(r / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int dy = 0; dy < 3; ++dy) {
            const int ih = oh * 2 + dy;
            if (ih >= H) continue;
            for (int dx = 0; dx < 3; ++dx) {
                const int iw = ow * 2 + dx;
                if (iw >= W) continue;
                const float4 v = reinterpret_cast<const float4*>(x)[((size_t)(n * H + ih) * W + iw) * C4 + c];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        reinterpret_cast<float4*>(out)[i] = m;
    }
}

hipError_t launch_maxpool3s2(const float* x, int N, int H, int W, int C, float* out, hipStream_t s)
{
    if (C % 4 || H % 2 || W % 2) return hipErrorInvalidValue;
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo * (C / 4);
    const int blocks = (int)min((size_t)1 << 20, (total + 255) / 256);       // one pass: the XCD mapping above needs the whole range in one sweep
    hipLaunchKernelGGL(maxpool3s2_kernel, dim3(blocks), dim3(256), 0, s, x, N, H, W, C / 4, out, Ho, Wo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
__global__ void split_xyzp_kernel(const float4* __restrict__ xyzp, size_t npix, float* __restrict__ xyz,
                                  float* __restrict__ prob)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = xyzp[i];
        xyz[3 * i] = v.x; xyz[3 * i + 1] = v.y; xyz[3 * i + 2] = v.z;
        prob[i] = v.w;
    }
}

hipError_t launch_split_xyzp(const float* xyzp, int64_t npix, float* xyz, float* prob, hipStream_t s)
{
    const int blocks = (int)min((int64_t)4096, (npix + 255) / 256);
    hipLaunchKernelGGL(split_xyzp_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(xyzp),
                       (size_t)npix, xyz, prob);
    return hipGetLastError();
}

}  // namespace p2p
