// Anti-aliased resizes: the Gaussian pre-filter scikit-image 0.17 - 0.18 applies by default before down-scaling
// (skimage.transform.resize(anti_aliasing=True) -> scipy.ndimage.gaussian_filter(image, sigma, mode, cval) -> warp), for all
// six resize call sites of est_pose (reference recognition.py:82,103,121,134,144,146).  The reference does not pin its
// scikit-image version (requirements.txt does not list it): p2p_est_pose_opts.resize_anti_aliasing selects this behaviour,
// default off = scikit-image <= 0.14.
//
// What is restated here is scipy's separable filter exactly as it runs for skimage:
//   sigma  = max(0, (n_in / n_out - 1) / 2)                                (skimage _warps.resize)
//   radius = int(4 * sigma + 0.5)                                          (gaussian_filter1d, truncate = 4)
//   w[x]   = exp((-0.5 / sigma^2) * x^2), x = -radius .. radius, divided by numpy's pairwise sum of the vector
//   per axis (axis 0 first, then axis 1; the channel axis has sigma 0 and is skipped), per output element
//       t = in[0] * w[0];  for d = radius .. 1:  t += (in[-d] + in[+d]) * w[d]        (NI_Correlate1D, symmetric branch)
//   in double, no FMA; the result of each axis pass is stored in the ARRAY's dtype -- float32 for the prob / img_pred maps
//   (round32), float64 for everything else; borders 'mirror' (d c b | a b c d | c b a) or 'constant' (cval).
// The images are tiny next to the generator passes (a 300-px crop is 2 MB), so the kernels are plain: one thread per
// output element, taps from L2.
#include "pipeline.h"

#include <cmath>
#include <map>
#include <mutex>

#pragma clang fp contract(off)

namespace p2p {

namespace {

constexpr int AA_MAX_SIDE = 4096;

// numpy's pairwise summation of a float64 vector (np.add.reduce, pairwise_sum in loops_utils.h): up to 128 elements in 8
// partial sums, longer vectors split in halves (first half rounded down to a multiple of 8)
double np_sum(const double* a, int n)
{
    if (n < 8) {
        double r = 0.;
        for (int i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_sum(a, n2) + np_sum(a + n2, n - n2);
}

}  // namespace

// sigma / radius / one-sided weights for crop side `side` against the 128-px network resolution:
// side > 128 -> the side x side canvas is filtered before shrinking to 128; side < 128 -> the 128x128 map is filtered
// before shrinking to side.  Returns the radius (0 = no filtering); w receives radius + 1 values, centre first.
int aa_weights_for_side(int side, std::vector<double>& w)
{
    w.clear();
    if (side <= 0 || side == 128) return 0;
    const double n_in = side > 128 ? (double)side : 128.0, n_out = side > 128 ? 128.0 : (double)side;
    const double factor = n_in / n_out;
    double sigma = (factor - 1) / 2;
    if (!(sigma > 0)) return 0;
    const int radius = (int)(4.0 * sigma + 0.5);
    if (radius <= 0) return 0;                     // scipy still correlates with the 1-tap kernel [1.0]: the identity
    const double sigma2 = sigma * sigma;
    const double c = -0.5 / sigma2;
    std::vector<double> phi(2 * radius + 1);
    for (int x = -radius; x <= radius; ++x) phi[x + radius] = std::exp(c * (double)((long long)x * x));
    const double s = np_sum(phi.data(), (int)phi.size());
    w.resize(radius + 1);
    for (int d = 0; d <= radius; ++d) w[d] = phi[radius + d] / s;
    return radius;
}

int aa_table_get(int device, AaTable* out)
{
    static std::mutex mu;
    static std::map<int, AaTable> tables;
    std::lock_guard<std::mutex> lk(mu);
    auto it = tables.find(device);
    if (it != tables.end()) { *out = it->second; return P2P_OK; }
    std::vector<double> pool, w;
    std::vector<int> off(AA_MAX_SIDE + 1, 0), rad(AA_MAX_SIDE + 1, 0);
    for (int s = 1; s <= AA_MAX_SIDE; ++s) {
        rad[s] = aa_weights_for_side(s, w);
        off[s] = (int)pool.size();
        pool.insert(pool.end(), w.begin(), w.end());
    }
    if (pool.empty()) pool.push_back(1.0);
    double* dw = nullptr;
    int *doff = nullptr, *drad = nullptr;
    hipError_t e;
    if ((e = hipMalloc((void**)&dw, pool.size() * sizeof(double))) != hipSuccess ||
        (e = hipMalloc((void**)&doff, off.size() * sizeof(int))) != hipSuccess ||
        (e = hipMalloc((void**)&drad, rad.size() * sizeof(int))) != hipSuccess ||
        (e = hipMemcpy(dw, pool.data(), pool.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(doff, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(drad, rad.data(), rad.size() * sizeof(int), hipMemcpyHostToDevice)) != hipSuccess) {
        set_error("anti-aliasing weight table: %s", hipGetErrorString(e));
        return P2P_ERR_HIP;
    }
    AaTable t;
    t.w = dw; t.off = doff; t.rad = drad; t.max_side = AA_MAX_SIDE;
    tables[device] = t;            // lives as long as the process (a few MB per device)
    *out = t;
    return P2P_OK;
}

namespace {

__device__ inline int mirror_idx(int i, int n)   // scipy 'mirror' == numpy-pad 'reflect'
{
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}

// order-preserving map double -> uint64 (atomicMin / atomicMax on the keys = min / max of the values)
__device__ inline unsigned long long range_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : b | 0x8000000000000000ull;
}
__device__ inline double range_val(unsigned long long k)
{
    return __longlong_as_double((long long)((k >> 63) ? k & 0x7FFFFFFFFFFFFFFFull : ~k));
}

// one axis pass of every item.  A thread produces V consecutive outputs ALONG the filter axis from one register window of V + 2 r samples
// (radius <= AA_RMAX = 8, i.e. crop sides below 928 px; larger radii take one output at a time) -- V = 8 rows in pass 0, where neighbouring threads take neighbouring memory elements of a
// row; V = 4 columns in pass 1 (channel fastest over the threads: a wave's loads of one window position are strided, of all positions together
// they cover whole cache lines; 8 columns, and 4 ROWS per thread with the taps' border handling resolved once, both measured slower).
// One output per thread re-read 2 r + 1 samples and paid the item's set-up, two divisions and 2 r mirror computations per output: 2.7 ms per
// 256-detection step at 40 - 300-px boxes, 1.6 ms now; 15 -> 6 ms at 250 - 420-px boxes (tools/aa_big_sides.sh).  An output's sum is the expression it always was -- centre first, then the pairs from
// the outside in --, so the bits do not depend on the grouping.
// The second pass also takes the [min, max] of what it writes -- the range skimage's clip=True clips the warp output to (a separate pass, one
// workgroup per image, was 2 ms per step): the first pass resets the item's keys, aa_range_finish_kernel turns them into vmin / vmax.
#ifndef P2P_AA_V1      // A/B builds (tools/ab_build.sh resize_aa.hip -DP2P_AA_V1=2)
#define P2P_AA_V1 4
#endif
constexpr int AA_V0 = 8, AA_V1 = P2P_AA_V1, AA_RMAX = 8;
template <int AXIS>
__global__ __launch_bounds__(256) void aa_filter_kernel(AaItem* __restrict__ items)
{
    constexpr int V = AXIS == 0 ? AA_V0 : AA_V1;
    __shared__ double s_lo[4], s_hi[4];
    AaItem& I = items[blockIdx.y];
    const int r = I.radius;
    if (r <= 0) return;
    if (AXIS == 0 && blockIdx.x == 0 && threadIdx.x == 0) { I.kmin = ~0ull; I.kmax = 0ull; }
    double lo = 1e300, hi = -1e300;
    const double* src = AXIS == 0 ? I.a : I.tmp;
    double* dst = AXIS == 0 ? I.tmp : I.a;
    const int H = I.H, W = I.W, C = I.C;
    // Region of interest (AaItem::r0..c1): outside it the image is exactly zero, so a filter pass is exactly zero outside the region grown by the
    // radius along its axis -- 0 * w + (0 + 0) * w = +0.0 -- and neither computed nor stored; taps outside what the previous stage wrote read +0.0.
    const bool roi = I.r1 > I.r0;
    const int R0 = roi ? max(0, I.r0 - r) : 0, R1 = roi ? min(H, I.r1 + r) : H;                      // rows both passes produce
    const int C0 = roi ? (AXIS == 0 ? I.c0 : max(0, I.c0 - r)) : 0, C1 = roi ? (AXIS == 0 ? I.c1 : min(W, I.c1 + r)) : W;
    // what the source holds: the image itself inside the region (pass 0), pass 0's output (pass 1)
    const int SR0 = roi ? (AXIS == 0 ? I.r0 : R0) : 0, SR1 = roi ? (AXIS == 0 ? I.r1 : R1) : H;
    const int SC0 = roi ? I.c0 : 0, SC1 = roi ? I.c1 : W;
    const int mode = I.mode, round32 = I.round32;
    const double cval = I.cval;
    const double* w = I.w;
    const int len = AXIS == 0 ? H : W;
    const int A0 = AXIS == 0 ? R0 : C0, A1 = AXIS == 0 ? R1 : C1;          // outputs along the filter axis
    const int groups = (A1 - A0 + V - 1) / V;                              // V of them per thread
    // pass 0: a thread = (group of rows, column, channel), (column, channel) fastest; pass 1: a thread = (row, group of columns, channel), channel fastest
    const unsigned inner = AXIS == 0 ? (unsigned)((C1 - C0) * C) : (unsigned)(groups * C);
    const unsigned total = AXIS == 0 ? (unsigned)groups * inner : (unsigned)(R1 - R0) * inner;
    double wr[AA_RMAX + 1];
#pragma unroll
    for (int d = 0; d <= AA_RMAX; ++d) wr[d] = d <= r ? w[d] : 0.0;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const unsigned o = e / inner, rem = e - o * inner;
        const unsigned xx = rem / (unsigned)C;
        const int c = (int)(rem - xx * (unsigned)C);
        const int pos0 = A0 + V * (int)(AXIS == 0 ? o : xx);                // first output of this thread along the axis
        const int fixed = AXIS == 0 ? C0 + (int)xx : R0 + (int)o;           // the other coordinate
        auto at = [&](int q) -> double {              // source sample at position q along the axis (border handled by the caller)
            const int sy = AXIS == 0 ? q : fixed, sx = AXIS == 0 ? fixed : q;
            if (roi && (sy < SR0 || sy >= SR1 || sx < SC0 || sx >= SC1)) return 0.0;
            return src[((long long)sy * W + sx) * C + c];
        };
        auto sample = [&](int q) -> double {          // the image extended past its ends: 'mirror' or the constant
            if ((unsigned)q < (unsigned)len) return at(q);
            return mode == 0 ? at(mirror_idx(q, len)) : cval;
        };
        auto emit = [&](int pos, double t) {
            const double v = round32 ? (double)(float)t : t;
            const int y = AXIS == 0 ? pos : fixed, x = AXIS == 0 ? fixed : pos;
            dst[((long long)y * W + x) * C + c] = v;
#ifndef P2P_ABL_AA_MINMAX      // timing ablation (A/B builds only: the clip range is garbage with it)
            if (AXIS == 1) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
#endif
        };
        if (r <= AA_RMAX) {                          // (wave-uniform: the radius belongs to the item)
            double win[V + 2 * AA_RMAX];
#pragma unroll
            for (int k = 0; k < V + 2 * AA_RMAX; ++k)
                win[k] = (k >= AA_RMAX - r && k < V + AA_RMAX + r && pos0 + k - AA_RMAX - r < A1) ? sample(pos0 + k - AA_RMAX) : 0.0;
#pragma unroll
            for (int i = 0; i < V; ++i) {
                if (pos0 + i >= A1) break;
                double t = win[i + AA_RMAX] * wr[0];
#pragma unroll
                for (int d = AA_RMAX; d >= 1; --d)
                    if (d <= r) t += (win[i + AA_RMAX - d] + win[i + AA_RMAX + d]) * wr[d];
                emit(pos0 + i, t);
            }
        } else {
            for (int i = 0; i < V && pos0 + i < A1; ++i) {
                const int pos = pos0 + i;
                double t = at(pos) * w[0];
                for (int d = r; d >= 1; --d) t += (sample(pos - d) + sample(pos + d)) * w[d];
                emit(pos, t);
            }
        }
    }
    if (AXIS == 1) {
        if (roi && blockIdx.x == 0 && threadIdx.x == 0 && (R0 > 0 || R1 < H || C0 > 0 || C1 < W)) { lo = 0.0 < lo ? 0.0 : lo; hi = 0.0 > hi ? 0.0 : hi; }      // the zeros outside
        for (int o = 32; o > 0; o >>= 1) {
            const double l2 = __shfl_down(lo, o, 64), h2 = __shfl_down(hi, o, 64);
            lo = l2 < lo ? l2 : lo;
            hi = h2 > hi ? h2 : hi;
        }
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 4; ++k) { lo = s_lo[k] < lo ? s_lo[k] : lo; hi = s_hi[k] > hi ? s_hi[k] : hi; }
            if (lo <= hi) { atomicMin(&I.kmin, range_key(lo)); atomicMax(&I.kmax, range_key(hi)); }
        }
    }
}

// keys -> [vmin, vmax] of every filtered image
__global__ void aa_range_finish_kernel(AaItem* __restrict__ items, int n_items)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    AaItem& I = items[i];
    if (I.radius <= 0) return;
    const bool any = I.kmin <= I.kmax;
    I.vmin = any ? range_val(I.kmin) : 1e300;
    I.vmax = any ? range_val(I.kmax) : -1e300;
}

}  // namespace

hipError_t launch_aa_filter(AaItem* items, int n_items, int max_elems, hipStream_t s)
{
    if (n_items <= 0) return hipSuccess;
    const int bx = std::max(1, std::min(256, (max_elems / AA_V1 + 255) / 256));      // a thread takes AA_V1 .. AA_V0 outputs (grid-stride loop: any grid is complete)
    for (int i0 = 0; i0 < n_items; i0 += 65535) {        // gridDim.y limit
        const int ni = std::min(65535, n_items - i0);
        { ProfScope ps(16, s); hipLaunchKernelGGL((aa_filter_kernel<0>), dim3(bx, ni), dim3(256), 0, s, items + i0); }
        { ProfScope ps(17, s); hipLaunchKernelGGL((aa_filter_kernel<1>), dim3(bx, ni), dim3(256), 0, s, items + i0); }
    }
    hipLaunchKernelGGL(aa_range_finish_kernel, dim3((n_items + 255) / 256), dim3(256), 0, s, items, n_items);
    return hipGetLastError();
}

}  // namespace p2p

// Test hook (no GPU needed): the host-side weights of one crop side, so that the CPU suite can hold them against
// scipy.ndimage's own kernel bit for bit.  w must hold 256 doubles; returns the radius, -1 on a bad side.
extern "C" int p2p_aa_weights(int side, double* w)
{
    if (!w || side <= 0 || side > p2p::AA_MAX_SIDE) return -1;
    std::vector<double> v;
    const int r = p2p::aa_weights_for_side(side, v);
    if (r + 1 > 256) return -1;
    for (int d = 0; d <= r && d < (int)v.size(); ++d) w[d] = v[d];
    return r;
}
