/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of the layer arithmetic the reference obtains from Keras 2.2.1 /
 * TensorFlow 1.9 (un-vendored: requirements.txt:1-2) for the Pix2Pose generator graphs
 *   pix2pose_model/ae_model.py:70-150   (aemodel_unet_prob)
 *   pix2pose_model/ae_model.py:175-240  (aemodel_unet_resnet50)
 *   pix2pose_model/resnet50_mod.py:40-118,200-213
 *
 * PARITY UNPINNED: TensorFlow/Keras cannot be imported here and the reference has no
 * tests or golden vectors, so these primitives follow the published semantics of the
 * pinned versions and are cross-checked against an independent
 * torch-CPU formulation in tests/test_oracle_ae.py.
 *
 * All tensors NHWC float32 (Keras channels_last; bn_axis=3, ae_model.py:72,176).
 * Accumulation is in double so the oracle sits between any two fp32 summation orders.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define P2PO_ACT_NONE 0
#define P2PO_ACT_RELU 1
#define P2PO_ACT_LEAKY 2
#define P2PO_ACT_TANH 3
#define P2PO_ACT_SIGMOID 4

/* TF "SAME": out = ceil(in/s); pad_total = max((out-1)*s + k - in, 0); before = total/2. */
void p2po_same_pad(int in, int k, int s, int* out, int* before)
{
    int o = (in + s - 1) / s;
    int tot = (o - 1) * s + k - in;
    if (tot < 0) tot = 0;
    *out = o;
    *before = tot / 2;
}

/* Conv2D, kernel HWIO (kh,kw,Cin,Cout), cross-correlation, explicit top/left pad
 * (zero outside), bias added.  y[n,oh,ow,co] = b[co] + sum x[n,oh*s+i-pt,ow*s+j-pl,ci]*w[i,j,ci,co]. */
void p2po_conv2d(const float* x, int N, int H, int W, int Cin,
                 const float* w, const float* b, int KH, int KW, int Cout,
                 int stride, int pad_t, int pad_l, int Ho, int Wo, float* y)
{
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * (size_t)Cout);
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int oh = 0; oh < Ho; ++oh)
                for (int ow = 0; ow < Wo; ++ow) {
                    for (int co = 0; co < Cout; ++co) acc[co] = b ? (double)b[co] : 0.0;
                    for (int i = 0; i < KH; ++i) {
                        int ih = oh * stride + i - pad_t;
                        if (ih < 0 || ih >= H) continue;
                        for (int j = 0; j < KW; ++j) {
                            int iw = ow * stride + j - pad_l;
                            if (iw < 0 || iw >= W) continue;
                            const float* xp = x + (((size_t)n * H + ih) * W + iw) * Cin;
                            const float* wp = w + ((size_t)(i * KW + j) * Cin) * Cout;
                            for (int ci = 0; ci < Cin; ++ci) {
                                double xv = xp[ci];
                                const float* wr = wp + (size_t)ci * Cout;
                                for (int co = 0; co < Cout; ++co) acc[co] += xv * (double)wr[co];
                            }
                        }
                    }
                    float* yp = y + (((size_t)n * Ho + oh) * Wo + ow) * Cout;
                    for (int co = 0; co < Cout; ++co) yp[co] = (float)acc[co];
                }
        free(acc);
    }
}

/* Conv2DTranspose 5x5 (any k) stride s, TF "SAME": out = s*in; kernel (kh,kw,Cout,Cin),
 * no flip: y[o] = sum_i sum_k x[i]*w[k] over o = s*i + k - crop, crop = (k - s)/2 rounded
 * down (k=5,s=2 -> full output of size 2n+3 cropped 1 before / 2 after).  SURVEY 8a-N5. */
void p2po_conv2d_transpose(const float* x, int N, int H, int W, int Cin,
                           const float* w, const float* b, int KH, int KW, int Cout,
                           int stride, float* y)
{
    const int Ho = H * stride, Wo = W * stride;
    const int crop_t = (KH - stride) / 2, crop_l = (KW - stride) / 2;
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * (size_t)Cout);
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int oh = 0; oh < Ho; ++oh)
                for (int ow = 0; ow < Wo; ++ow) {
                    for (int co = 0; co < Cout; ++co) acc[co] = b ? (double)b[co] : 0.0;
                    for (int i = 0; i < KH; ++i) {
                        int t = oh + crop_t - i;
                        if (t < 0 || t % stride) continue;
                        int ih = t / stride;
                        if (ih >= H) continue;
                        for (int j = 0; j < KW; ++j) {
                            int u = ow + crop_l - j;
                            if (u < 0 || u % stride) continue;
                            int iw = u / stride;
                            if (iw >= W) continue;
                            const float* xp = x + (((size_t)n * H + ih) * W + iw) * Cin;
                            const float* wp = w + ((size_t)(i * KW + j) * Cout) * Cin;
                            for (int co = 0; co < Cout; ++co) {
                                const float* wr = wp + (size_t)co * Cin;
                                double s = 0.0;
                                for (int ci = 0; ci < Cin; ++ci) s += (double)xp[ci] * (double)wr[ci];
                                acc[co] += s;
                            }
                        }
                    }
                    float* yp = y + (((size_t)n * Ho + oh) * Wo + ow) * Cout;
                    for (int co = 0; co < Cout; ++co) yp[co] = (float)acc[co];
                }
        free(acc);
    }
}

/* BatchNormalization (inference): y = gamma*(x-mean)/sqrt(var+eps)+beta, then activation. */
void p2po_bn_act(float* x, size_t npix, int C, const float* gamma, const float* beta,
                 const float* mean, const float* var, double eps, int act, double alpha)
{
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < npix; ++p) {
        float* xp = x + p * C;
        for (int c = 0; c < C; ++c) {
            double v = xp[c];
            if (gamma) v = (double)gamma[c] * (v - (double)mean[c]) / sqrt((double)var[c] + eps) + (double)beta[c];
            switch (act) {
            case P2PO_ACT_RELU: v = v > 0 ? v : 0; break;
            case P2PO_ACT_LEAKY: v = v > 0 ? v : alpha * v; break;
            case P2PO_ACT_TANH: v = tanh(v); break;
            case P2PO_ACT_SIGMOID: v = 1.0 / (1.0 + exp(-v)); break;
            default: break;
            }
            xp[c] = (float)v;
        }
    }
}

/* y = relu(a + b) elementwise (resnet50_mod.py:71-72,116-117). */
void p2po_add_relu(const float* a, const float* b, size_t n, float* y)
{
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        float v = a[i] + b[i];
        y[i] = v > 0 ? v : 0;
    }
}

/* MaxPooling2D k x k stride s with explicit top/left pad; padded cells ignored (-inf). */
void p2po_maxpool(const float* x, int N, int H, int W, int C, int k, int s,
                  int pad_t, int pad_l, int Ho, int Wo, float* y)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oh = 0; oh < Ho; ++oh)
            for (int ow = 0; ow < Wo; ++ow)
                for (int c = 0; c < C; ++c) {
                    float m = -INFINITY;
                    for (int i = 0; i < k; ++i) {
                        int ih = oh * s + i - pad_t;
                        if (ih < 0 || ih >= H) continue;
                        for (int j = 0; j < k; ++j) {
                            int iw = ow * s + j - pad_l;
                            if (iw < 0 || iw >= W) continue;
                            float v = x[(((size_t)n * H + ih) * W + iw) * C + c];
                            if (v > m) m = v;
                        }
                    }
                    y[(((size_t)n * Ho + oh) * Wo + ow) * C + c] = m;
                }
}

/* Dense: y[n,o] = b[o] + sum_i x[n,i]*w[i,o]; kernel (in,out). */
void p2po_dense(const float* x, int N, int In, const float* w, const float* b, int Out, float* y)
{
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * (size_t)Out);
#pragma omp for schedule(static)
        for (int n = 0; n < N; ++n) {
            for (int o = 0; o < Out; ++o) acc[o] = b ? (double)b[o] : 0.0;
            for (int i = 0; i < In; ++i) {
                double xv = x[(size_t)n * In + i];
                const float* wr = w + (size_t)i * Out;
                for (int o = 0; o < Out; ++o) acc[o] += xv * (double)wr[o];
            }
            for (int o = 0; o < Out; ++o) y[(size_t)n * Out + o] = (float)acc[o];
        }
        free(acc);
    }
}
