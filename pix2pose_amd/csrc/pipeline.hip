// est_pose as a batched device pipeline (gfx950).  Replaces, for n detections at once,
// pix2pose.est_pose (reference pix2pose_model/recognition.py:70-193), get_boxes (:28-69) and the
// correspondence building of pnp_ransac (:195-213); the PnP-RANSAC solve itself is pnp.hip.
//
// Everything between the two generator passes stays on the device: no host round trip decides the
// stage-2 geometry.  The many O(H*W) temporaries and the 16 skimage.resize calls per detection of
// the reference collapse into per-pixel gathers: every resized map is evaluated on the fly from
// the 128x128 network output with skimage's bilinear rule (half-pixel centres,
// floor/ceil taps, 'reflect' or 'constant'+cval borders, no anti-aliasing), in float64 like the
// reference, with FMA contraction off so thresholds (>0.9, <th, uint8 truncation) see the same
// values as the numpy formulation.
#include "pipeline.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <numeric>

#pragma clang fp contract(off)

namespace p2p {

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return P2P_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return P2P_OK;
    release();
    const size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return P2P_ERR_HIP;
    }
    cap = want;
    return P2P_OK;
}
void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}
int PinnedBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return P2P_OK;
    release();
    const size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e != hipSuccess) {
        p = nullptr;
        set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
        return P2P_ERR_HIP;
    }
    cap = want;
    return P2P_OK;
}
void PinnedBuf::release()
{
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
}
Pipeline::~Pipeline()
{
    for (Slot& s : slot) {
        for (DevBuf* b : {&s.det, &s.s1, &s.cand, &s.probs, &s.results, &s.poses, &s.corr, &s.hyp, &s.x1, &s.y1, &s.x2, &s.y2, &s.images, &s.crec, &s.cseg, &s.sacc,
                          &s.mask, &s.pred, &s.dmask, &s.mstat, &s.crange, &s.aa_items, &s.aa_cv, &s.aa_cv_tmp, &s.aa_bk, &s.aa_bk_tmp, &s.keepf}) b->release();
        for (PinnedBuf* b : {&s.h_mask, &s.h_pred, &s.h_stat, &s.h_frames, &s.h_range}) b->release();
        if (s.host_poses) (void)hipHostFree(s.host_poses);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    if (corr_ready) (void)hipEventDestroy(corr_ready);
    if (frames_ready) (void)hipEventDestroy(frames_ready);
    if (tail_stream) (void)hipStreamDestroy(tail_stream);
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
}
void Ctx::free_pipeline()
{
    delete pipe;
    pipe = nullptr;
}

// ------------------------------------------------------------------------------------------
// geometry (host + device): recognition.py:28-69
// ------------------------------------------------------------------------------------------
__host__ __device__ inline Boxes get_boxes(const double bbox[4], int v_max, int u_max, double box_size, bool has_ct,
                                           int ct_v, int ct_u, double max_w)
{
    if (!has_ct || ct_v == -1) {      // the reference's "no centre given" sentinel is ct[0] == -1 (recognition.py:28-34): a genuine
                                      // centre row of -1 falls back to the box centre there too
        ct_v = (int)((bbox[0] + bbox[2]) / 2);
        ct_u = (int)((bbox[1] + bbox[3]) / 2);
    }
    const double width = bbox[3] - bbox[1], height = bbox[2] - bbox[0];
    const double wa = width * box_size, wb = height * box_size;
    double w = wa > wb ? wa : wb;
    if (max_w < w) w = max_w;
    const int half = (int)(w / 2);
    Boxes b;
    b.v1_ori = ct_v - half; b.v2_ori = ct_v + half;
    b.u1_ori = ct_u - half; b.u2_ori = ct_u + half;
    b.v1 = b.v1_ori; b.v2 = b.v2_ori; b.u1 = b.u1_ori; b.u2 = b.u2_ori;
    int sv0 = 0, su0 = 0, sv1 = 0, su1 = 0;
    if (b.v1_ori < 0) { sv0 = -b.v1_ori; b.v1 = 0; }
    if (b.v2_ori > v_max) { sv1 = -(b.v2_ori - v_max); b.v2 = v_max; }
    if (b.u1_ori < 0) { su0 = -b.u1_ori; b.u1 = 0; }
    if (b.u2_ori > u_max) { su1 = -(b.u2_ori - u_max); b.u2 = u_max; }
    b.vv1 = sv0; b.vv2 = sv1 + (b.v2_ori - b.v1_ori);
    b.uu1 = su0; b.uu2 = su1 + (b.u2_ori - b.u1_ori);
    return b;
}

// crop usable?  (recognition.py:78-79 / :117-119; boxes the reference would crash on -- entirely
// outside the frame, where numpy's negative slice indices wrap -- are rejected as well)
__host__ __device__ inline bool boxes_ok(const Boxes& b, int H, int W)
{
    const int side_v = b.v2_ori - b.v1_ori, side_u = b.u2_ori - b.u1_ori;
    if (side_v < 5 || side_u < 5) return false;
    if (b.v2 - b.v1 < 5 || b.u2 - b.u1 < 5) return false;
    if (b.v2 <= 0 || b.u2 <= 0 || b.v1 >= H || b.u1 >= W) return false;
    return true;
}

// ------------------------------------------------------------------------------------------
// skimage-style bilinear sampling helpers
// ------------------------------------------------------------------------------------------
struct Tap {
    int i0, i1;
    double d;
};

__device__ inline Tap axis_tap(int o, int n_in, int n_out)
{
    // identity resize (every resize of a 128-px crop): src = o * 1.0 + (0.5 - 0.5) = o exactly -- the same taps and weight without the
    // fp64 division (a quarter of stage2_input_kernel's time at BASELINE.json configs[2])
    if (n_in == n_out) { Tap t; t.i0 = t.i1 = o; t.d = 0.0; return t; }
    const double s = (double)n_in / (double)n_out;
    const double src = (double)o * s + (0.5 * s - 0.5);
    const double lo = floor(src);
    Tap t;
    t.i0 = (int)lo;
    t.i1 = (int)ceil(src);
    t.d = src - lo;
    return t;
}

__device__ inline int reflect_idx(int i, int n)   // numpy-pad 'reflect' (edge sample not repeated)
{
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}

// A tap whose weight is exactly zero (d == 0: the source coordinate is an integer, floor == ceil, as in every resize of a
// 128-px crop) is not evaluated: (1 - 0) * v + 0 * anything-finite == v, so its value can be any finite number.  For the
// identity resizes of BASELINE.json configs[1-3] that is 1 tap instead of 4 -- and 1 instead of 16 where taps nest.
#define P2P_TAP_LIVE(a, e, tr, tc) (((a) == 0 || (tr).d != 0) && ((e) == 0 || (tc).d != 0))

__device__ inline double lerp2(double tl, double tr, double bl, double br, double dr, double dc)
{
    const double top = (1 - dc) * tl + dc * tr;
    const double bot = (1 - dc) * bl + dc * br;
    return (1 - dr) * top + dr * bot;
}

// skimage's clip=True (every version the reference can run on): the warp output is clamped to [min, max] of the warp
// INPUT; in 'constant' mode with cval outside that range, outputs exactly equal to cval are left alone
// (skimage.transform._warps._clip_warp_output).  A no-op unless taps fall outside the image (up-scaling borders) or
// the anti-aliasing filter mixed cval in.
__device__ inline double clip_warp(double v, double lo, double hi, double cval)
{
    if (!(lo <= cval && cval <= hi) && v == cval) return v;
    return v < lo ? lo : (v > hi ? hi : v);
}

// (pixel - 128) / 128 of frame pixel (y, x), channel ch
__device__ inline double frame_px(const DetInfo& D, int y, int x, int ch)
{
    const size_t o = ((size_t)y * D.W + x) * 3 + ch;
    const double p = D.img_f32 ? (double)reinterpret_cast<const float*>(D.img)[o]
                               : (double)reinterpret_cast<const unsigned char*>(D.img)[o];
    return (p - 128.0) / 128.0;
}

__device__ inline bool non_gray_at(const float* y4)   // np.linalg.norm(decode, axis=2) > 0.3 in float32
{
    const float s = (y4[0] * y4[0] + y4[1] * y4[1]) + y4[2] * y4[2];
    return sqrtf(s) > 0.3f;
}

// ------------------------------------------------------------------------------------------
// K1: stage-1 network inputs  (recognition.py:75-82)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stage1_input_kernel(const DetInfo* __restrict__ dets, float* __restrict__ x1, AaPtrs aa)
{
    const int d = blockIdx.x >> 6;
    const int pix = ((blockIdx.x & 63) << 8) | threadIdx.x;
    const int oy = pix >> 7, ox = pix & 127;
    const DetInfo& D = dets[d];
    float* out = x1 + ((size_t)d * 16384 + pix) * 3;
    if (!D.ok1) { out[0] = out[1] = out[2] = 0.f; return; }
    const Boxes& b = D.b1;
    const int S = b.v2_ori - b.v1_ori, Sw = b.u2_ori - b.u1_ori;
    const Tap tr = axis_tap(oy, S, 128), tc = axis_tap(ox, Sw, 128);
    const int r[2] = {reflect_idx(tr.i0, S), reflect_idx(tr.i1, S)};
    const int c[2] = {reflect_idx(tc.i0, Sw), reflect_idx(tc.i1, Sw)};
    const double* cv = (aa.k0 && aa.k0[d].radius > 0) ? aa.k0[d].a : nullptr;      // anti-aliased canvas instead of the frame
    for (int ch = 0; ch < 3; ++ch) {
        double v[2][2];
        for (int a = 0; a < 2; ++a)
            for (int e = 0; e < 2; ++e) {
                if (!P2P_TAP_LIVE(a, e, tr, tc)) { v[a][e] = 0.0; continue; }
                if (cv) { v[a][e] = cv[((size_t)r[a] * Sw + c[e]) * 3 + ch]; continue; }
                const bool in = r[a] >= b.vv1 && r[a] < b.vv2 && c[e] >= b.uu1 && c[e] < b.uu2;
                v[a][e] = in ? frame_px(D, b.v1 + r[a] - b.vv1, b.u1 + c[e] - b.uu1, ch) : 0.0;
            }
        out[ch] = (float)lerp2(v[0][0], v[0][1], v[1][0], v[1][1], tr.d, tc.d);
    }
}

// stage-2 geometry from the reductions of one detection (recognition.py:96-110)
__device__ inline void stage1_finalize(const DetInfo& D, int s_n, int s_minv, int s_minu, int s_maxv, int s_maxu, long long s_sv, long long s_su,
                                       const int* s_keep, Stage1* out)
{
    Stage1 o;
    memset(&o, 0, sizeof(o));
    o.n_init_mask = s_n;
    o.bb[0] = s_minv; o.bb[1] = s_minu; o.bb[2] = s_maxv; o.bb[3] = s_maxu;
    o.sum_v = s_sv; o.sum_u = s_su;
    o.b2 = D.b1;
    bool any = false;
    if (D.ok1 && s_n > 0) {
        const Boxes& b = D.b1;
        const double sy = (double)(b.v2_ori - b.v1_ori) / 128, sx = (double)(b.u2_ori - b.u1_ori) / 128;
        const double bb[4] = {s_minv * sy, s_minu * sx, s_maxv * sy, s_maxu * sx};          // :101-102
        const double mean_u = (double)s_su / (double)s_n, mean_v = (double)s_sv / (double)s_n;
        const int cx_m = (int)((mean_u - (127.0 / 2)) + D.cx_o);                            // :108
        const int cy_m = (int)((mean_v - (127.0 / 2)) + D.cy_o);                            // :109
        const Boxes b2 = get_boxes(bb, D.H, D.W, D.box_size, true, cy_m, cx_m, (double)(b.v2_ori - b.v1_ori));   // :110
        const bool geo_ok = boxes_ok(b2, D.H, D.W);
        for (int k = 0; k < D.n_th; ++k) {
            o.keep_cnt[k] = s_keep[k];
            o.kmin[k] = s_keep[k] == 16384 ? 1.0 : 0.0;      // range of the bool mask as float (clip=True of the :103 resize)
            o.kmax[k] = s_keep[k] > 0 ? 1.0 : 0.0;
            o.valid2[k] = (s_keep[k] >= 10 && geo_ok) ? 1 : 0;                             // :96-97, :117-119
            if (o.valid2[k]) { any = true; ++o.n_cand; }
        }
        if (any) o.b2 = b2;
    }
    *out = o;
}

// ------------------------------------------------------------------------------------------
// K3: stage-1 reductions + stage-2 geometry  (recognition.py:89-110)
// ------------------------------------------------------------------------------------------
// (blockDim.x = 256, or 1024 for a handful of detections: one workgroup per detection walks 16 384 pixels, and with one detection at a
// time -- the reference's caller -- its 64 trips were 50 us of the call)
__global__ __launch_bounds__(1024) void stage1_stats_kernel(const DetInfo* __restrict__ dets, const float* __restrict__ y1,
                                                            Stage1* __restrict__ s1)
{
    __shared__ int s_n, s_minv, s_minu, s_maxv, s_maxu, s_sv, s_su;
    __shared__ int s_keep[MAX_TH];
    const int d = blockIdx.x, tid = threadIdx.x;
    const DetInfo& D = dets[d];
    if (tid == 0) { s_n = 0; s_minv = s_minu = 1 << 30; s_maxv = s_maxu = -1; s_sv = s_su = 0; }
    if (tid < MAX_TH) s_keep[tid] = 0;
    __syncthreads();
    int n = 0, minv = 1 << 30, minu = 1 << 30, maxv = -1, maxu = -1, sv = 0, su = 0;
    int keep[MAX_TH];
#pragma unroll
    for (int k = 0; k < MAX_TH; ++k) keep[k] = 0;
    const float* y = y1 + (size_t)d * 16384 * 4;
    for (int p = tid; p < 16384; p += blockDim.x) {
        const float4 q = reinterpret_cast<const float4*>(y)[p];
        const float v4[4] = {q.x, q.y, q.z, q.w};
        if (!non_gray_at(v4)) continue;
        const int v = p >> 7, u = p & 127;
        ++n; sv += v; su += u;
        minv = min(minv, v); maxv = max(maxv, v); minu = min(minu, u); maxu = max(maxu, u);
#pragma unroll
        for (int k = 0; k < MAX_TH; ++k)
            if (k < D.n_th && q.w < D.th_o[k]) ++keep[k];
    }
    atomicAdd(&s_n, n); atomicAdd(&s_sv, sv); atomicAdd(&s_su, su);
    atomicMin(&s_minv, minv); atomicMin(&s_minu, minu); atomicMax(&s_maxv, maxv); atomicMax(&s_maxu, maxu);
#pragma unroll
    for (int k = 0; k < MAX_TH; ++k)
        if (keep[k]) atomicAdd(&s_keep[k], keep[k]);
    __syncthreads();
    if (tid != 0) return;
    stage1_finalize(D, s_n, s_minv, s_minu, s_maxv, s_maxu, s_sv, s_su, s_keep, &s1[d]);
}

// The same reductions for a HANDFUL of detections (one at a time: the reference's caller), spread over STATS_SEG workgroups per detection: each
// adds its share to per-detection accumulators in global memory, and whichever finishes last turns them into the Stage1 record and
// leaves them cleared for the next call (integer sums and extrema: the order of the additions is immaterial).
constexpr int STATS_SEG = 8;
struct StatsAcc { int n, minv, minu, maxv, maxu, sv, su, keep[MAX_TH], done; };

__global__ void stats_acc_init_kernel(StatsAcc* acc, int n)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    StatsAcc a;
    memset(&a, 0, sizeof(a));
    a.minv = a.minu = 1 << 30; a.maxv = a.maxu = -1;
    acc[d] = a;
}

__global__ __launch_bounds__(256) void stage1_stats_seg_kernel(const DetInfo* __restrict__ dets, const float* __restrict__ y1,
                                                               Stage1* __restrict__ s1, StatsAcc* __restrict__ acc)
{
    __shared__ int s_last;
    const int d = blockIdx.y, seg = blockIdx.x, tid = threadIdx.x;
    const DetInfo& D = dets[d];
    int n = 0, minv = 1 << 30, minu = 1 << 30, maxv = -1, maxu = -1, sv = 0, su = 0;
    int keep[MAX_TH];
#pragma unroll
    for (int k = 0; k < MAX_TH; ++k) keep[k] = 0;
    const float* y = y1 + (size_t)d * 16384 * 4;
    constexpr int PER = 16384 / STATS_SEG;
    for (int p = seg * PER + tid; p < (seg + 1) * PER; p += 256) {
        const float4 q = reinterpret_cast<const float4*>(y)[p];
        const float v4[4] = {q.x, q.y, q.z, q.w};
        if (!non_gray_at(v4)) continue;
        const int v = p >> 7, u = p & 127;
        ++n; sv += v; su += u;
        minv = min(minv, v); maxv = max(maxv, v); minu = min(minu, u); maxu = max(maxu, u);
#pragma unroll
        for (int k = 0; k < MAX_TH; ++k)
            if (k < D.n_th && q.w < D.th_o[k]) ++keep[k];
    }
    for (int o = 32; o > 0; o >>= 1) {
        n += __shfl_down(n, o, 64); sv += __shfl_down(sv, o, 64); su += __shfl_down(su, o, 64);
        minv = min(minv, __shfl_down(minv, o, 64)); minu = min(minu, __shfl_down(minu, o, 64));
        maxv = max(maxv, __shfl_down(maxv, o, 64)); maxu = max(maxu, __shfl_down(maxu, o, 64));
#pragma unroll
        for (int k = 0; k < MAX_TH; ++k) keep[k] += __shfl_down(keep[k], o, 64);
    }
    StatsAcc& A = acc[d];
    if ((tid & 63) == 0 && n > 0) {
        atomicAdd(&A.n, n); atomicAdd(&A.sv, sv); atomicAdd(&A.su, su);
        atomicMin(&A.minv, minv); atomicMin(&A.minu, minu); atomicMax(&A.maxv, maxv); atomicMax(&A.maxu, maxu);
#pragma unroll
        for (int k = 0; k < MAX_TH; ++k)
            if (keep[k]) atomicAdd(&A.keep[k], keep[k]);
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&A.done, 1) == STATS_SEG - 1;
    __syncthreads();
    if (!s_last || tid != 0) return;
    __threadfence();
    int kp[MAX_TH];
    for (int k = 0; k < MAX_TH; ++k) kp[k] = atomicAdd(&A.keep[k], 0);
    stage1_finalize(D, atomicAdd(&A.n, 0), atomicAdd(&A.minv, 0), atomicAdd(&A.minu, 0), atomicAdd(&A.maxv, 0), atomicAdd(&A.maxu, 0),
                    atomicAdd(&A.sv, 0), atomicAdd(&A.su, 0), kp, &s1[d]);
    StatsAcc z;
    memset(&z, 0, sizeof(z));
    z.minv = z.minu = 1 << 30; z.maxv = z.maxu = -1;
    A = z;                                           // cleared for the next batch that uses this slot
}

// resize(keep, (S,S), 'constant', 0) > 0.9 at canvas position (r, c) of the stage-1 square.  The keep mask is a BOOL image: scikit-image
// <= 0.14 and 0.17 / 0.18 do not filter it, 0.15 / 0.16 (generation 2) do, into a bool array; [kmin, kmax]: range of the warp input (clip=True).
// kf (generation 2, stage-1 side < 128): the keep mask after scipy's Gaussian filter with BOOL output (keep_filter_kernel), [128 * 128] bytes.
__device__ inline bool keep_ori_at(const float* y1d, float th, double kmin, double kmax, int r, int c, int S, int Sw, const unsigned char* kf = nullptr)
{
    const Tap tr = axis_tap(r, 128, S), tc = axis_tap(c, 128, Sw);
    double v[2][2];
    const int ri[2] = {tr.i0, tr.i1}, cj[2] = {tc.i0, tc.i1};
    for (int a = 0; a < 2; ++a)
        for (int e = 0; e < 2; ++e) {
            double k = 0.0;
            if (P2P_TAP_LIVE(a, e, tr, tc) && ri[a] >= 0 && ri[a] < 128 && cj[e] >= 0 && cj[e] < 128) {
                const float* q = y1d + ((size_t)ri[a] * 128 + cj[e]) * 4;
                k = kf ? (kf[ri[a] * 128 + cj[e]] ? 1.0 : 0.0) : ((non_gray_at(q) && q[3] < th) ? 1.0 : 0.0);
            }
            v[a][e] = k;
        }
    return clip_warp(lerp2(v[0][0], v[0][1], v[1][0], v[1][1], tr.d, tc.d), kmin, kmax, 0.0) > 0.9;
}

// value of the stage-2 canvas (recognition.py:113-120) at canvas position (r, c), channel ch: the frame pixel where the
// kept mask says foreground, zero elsewhere
__device__ inline bool stage2_fg(const DetInfo& D, const Stage1& S, const float* y1d, int slot, int r, int c, int* fy, int* fx, const unsigned char* kf = nullptr)
{
    const Boxes& b1 = D.b1;
    const Boxes& b = S.b2;
    const bool in = r >= b.vv1 && r < b.vv2 && c >= b.uu1 && c < b.uu2;
    if (!in) return false;
    const int y = b.v1 + r - b.vv1, x = b.u1 + c - b.uu1;
    *fy = y; *fx = x;
    // bg_full: True outside the stage-1 clipped crop, ~keep_ori inside   (:105-106)
    if (y >= b1.v1 && y < b1.v2 && x >= b1.u1 && x < b1.u2)
        return keep_ori_at(y1d, D.th_o[slot], S.kmin[slot], S.kmax[slot], y - b1.v1_ori, x - b1.u1_ori, b1.v2_ori - b1.v1_ori, b1.u2_ori - b1.u1_ori, kf);
    return false;
}

// ------------------------------------------------------------------------------------------
// K4: stage-2 network inputs  (recognition.py:103-121)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stage2_input_kernel(const DetInfo* __restrict__ dets, const Stage1* __restrict__ s1,
                                                           const float* __restrict__ y1, int K, float* __restrict__ x2, AaPtrs aa)
{
    const int cand = blockIdx.x >> 6;
    const int d = cand / K, slot = cand - d * K;
    const int pix = ((blockIdx.x & 63) << 8) | threadIdx.x;
    const int oy = pix >> 7, ox = pix & 127;
    const DetInfo& D = dets[d];
    const Stage1& S = s1[d];
    float* out = x2 + ((size_t)cand * 16384 + pix) * 3;
    if (slot >= D.n_th || !S.valid2[slot]) { out[0] = out[1] = out[2] = 0.f; return; }
    const Boxes& b = S.b2;
    const int S2 = b.v2_ori - b.v1_ori, S2w = b.u2_ori - b.u1_ori;
    const Tap tr = axis_tap(oy, S2, 128), tc = axis_tap(ox, S2w, 128);
    const int r[2] = {reflect_idx(tr.i0, S2), reflect_idx(tr.i1, S2)};
    const int c[2] = {reflect_idx(tc.i0, S2w), reflect_idx(tc.i1, S2w)};
    const double* cv = (aa.k1 && aa.k1[cand].radius > 0) ? aa.k1[cand].a : nullptr;   // anti-aliased canvas
    if (cv) {
        // the filtered canvas exists inside its region of interest grown by the filter radius; it is +0.0 everywhere else
        const AaItem& I = aa.k1[cand];
        const int R0 = max(0, I.r0 - I.radius), R1 = min(I.H, I.r1 + I.radius), C0 = max(0, I.c0 - I.radius), C1 = min(I.W, I.c1 + I.radius);
        bool in[2][2];
        for (int a = 0; a < 2; ++a)
            for (int e = 0; e < 2; ++e) in[a][e] = P2P_TAP_LIVE(a, e, tr, tc) && r[a] >= R0 && r[a] < R1 && c[e] >= C0 && c[e] < C1;
        for (int ch = 0; ch < 3; ++ch) {
            double v[2][2];
            for (int a = 0; a < 2; ++a)
                for (int e = 0; e < 2; ++e) v[a][e] = in[a][e] ? cv[((size_t)r[a] * S2w + c[e]) * 3 + ch] : 0.0;
            out[ch] = (float)lerp2(v[0][0], v[0][1], v[1][0], v[1][1], tr.d, tc.d);
        }
        return;
    }
    const float* y1d = y1 + (size_t)d * 16384 * 4;
    const unsigned char* kf = (aa.keepf && S.keepf[slot]) ? aa.keepf + (size_t)cand * 16384 : nullptr;      // generation 2: the filtered bool keep mask
    bool fg[2][2];
    int fy[2][2], fx[2][2];
    for (int a = 0; a < 2; ++a)
        for (int e = 0; e < 2; ++e) {
            fy[a][e] = fx[a][e] = 0;
            fg[a][e] = P2P_TAP_LIVE(a, e, tr, tc) && stage2_fg(D, S, y1d, slot, r[a], c[e], &fy[a][e], &fx[a][e], kf);
        }
    for (int ch = 0; ch < 3; ++ch) {
        double v[2][2];
        for (int a = 0; a < 2; ++a)
            for (int e = 0; e < 2; ++e) v[a][e] = fg[a][e] ? frame_px(D, fy[a][e], fx[a][e], ch) : 0.0;
        out[ch] = (float)lerp2(v[0][0], v[0][1], v[1][0], v[1][1], tr.d, tc.d);
    }
}

// ------------------------------------------------------------------------------------------
// per-pixel evaluation of one candidate at crop resolution  (recognition.py:134-154, 196-204)
// ------------------------------------------------------------------------------------------
struct CandPixel {
    bool non_gray;
    bool valid;
    bool below;            // img_prob_ori < th_inlier by itself (p2p_debug_back_resize; dead code elsewhere)
    unsigned char q[3];
};

// raw 128x128 maps of one network output pixel (recognition.py:137-143): prob, non_gray as float, img_pred
__device__ inline void cand_raw(const float* q, double* prob, double* ng, double pred[3])
{
    const float s = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
    const bool gray = sqrtf(s) < 0.3f;                                  // :137
    *prob = (double)q[3];
    *ng = gray ? 0.0 : 1.0;
    for (int ch = 0; ch < 3; ++ch) {
        const float dq = gray ? 0.f : q[ch];                            // :139
        float ip = (dq + 1.0f) / 2.0f;                                  // :141 (float32)
        ip = ip > 1.f ? 1.f : (ip < 0.f ? 0.f : ip);                    // :142-143
        pred[ch] = (double)ip;
    }
}

// scikit-image 0.17 / 0.18 keep a FLOAT32 image float32 through warp() -- the prob map of recognition.py:134 and img_pred of :144 are
// float32 arrays (Keras output; (decode + 1) / 2 stays float32) -- and the compiled _warp_fast[float32] / bilinear_interpolation[float32]
// of the 0.18.3 wheel do, per output pixel (disassembled; the test suite's CPU restatement of exactly this is checked bit for bit against
// the real library, tests/golden/external_vectors.json, and this kernel against both):
//   matrix cast to float32: ms = (float)(n_in / n_out), mt = (float)(0.5 * n_in / n_out - 0.5)
//   c = ms * (float)col + mt                 two float32 roundings (mulss, addss)
//   taps floorf(c), ceilf(c); dc = c - floorf(c) in float32
//   top = (1.0 - (double)dc) * (double)tl + (double)(dc * tr)          dc * tr is a float32 product
//   out = (float)((1.0 - (double)dr) * top + (double)dr * bottom)
struct TapF {
    int i0, i1;
    float d;
};

__device__ inline TapF axis_tap_f32(int o, int n_in, int n_out)
{
    if (n_in == n_out) { TapF t; t.i0 = t.i1 = o; t.d = 0.f; return t; }      // ms = 1, mt = 0: src = (float)o exactly
    const double s = (double)n_in / (double)n_out;
    const float ms = (float)s, mt = (float)(s * 0.5 - 0.5);
    const float src = ms * (float)o + mt;            // contraction is off in this file
    const float lo = floorf(src);
    TapF t;
    t.i0 = (int)lo;
    t.i1 = (int)ceilf(src);
    t.d = src - lo;
    return t;
}

__device__ inline float lerp2_f32(float tl, float tr, float bl, float br, float dr, float dc)
{
    const double top = (1.0 - (double)dc) * (double)tl + (double)(dc * tr);
    const double bot = (1.0 - (double)dc) * (double)bl + (double)(dc * br);
    return (float)((1.0 - (double)dr) * top + (double)dr * bot);
}

// bk: the candidate's five anti-aliased planes [prob | pred r | g | b | non_gray][128*128] (null: raw maps from y2c)
// gen: 0 = every image warped in double (scikit-image <= 0.14, and 0.15 / 0.16 -- there on the filtered planes `bk`), 1 = 0.17 / 0.18 (prob and
// img_pred warped in float32, compared with th_inlier and multiplied by 255 in float32; the non_gray image of :146 is a float64 array in every version)
__device__ inline CandPixel cand_pixel(const float* y2c, const double* bk, const CandRange& R, int r, int c, int S2, int S2w, double th_i, int gen)
{
    const Tap tr = axis_tap(r, 128, S2), tc = axis_tap(c, 128, S2w);
    const int ri[2] = {tr.i0, tr.i1}, cj[2] = {tc.i0, tc.i1};
    CandPixel o;
    if (gen) {
        // float64 image: non_gray
        double ng[2][2];
        for (int a = 0; a < 2; ++a)
            for (int e = 0; e < 2; ++e) {
                ng[a][e] = 0.0;
                if (P2P_TAP_LIVE(a, e, tr, tc) && ri[a] >= 0 && ri[a] < 128 && cj[e] >= 0 && cj[e] < 128) {
                    const int idx = ri[a] * 128 + cj[e];
                    if (bk) ng[a][e] = bk[4 * 16384 + idx];
                    else {
                        const float* q = y2c + (size_t)idx * 4;
                        ng[a][e] = sqrtf((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) < 0.3f ? 0.0 : 1.0;
                    }
                }
            }
        o.non_gray = clip_warp(lerp2(ng[0][0], ng[0][1], ng[1][0], ng[1][1], tr.d, tc.d), R.gmin, R.gmax, 0.0) > 0.9;
        // float32 images: prob, img_pred -- their own (float32) source coordinates and taps
        const TapF fr = axis_tap_f32(r, 128, S2), fc = axis_tap_f32(c, 128, S2w);
        const int fi[2] = {fr.i0, fr.i1}, fj[2] = {fc.i0, fc.i1};
        float prob[2][2], pred[3][2][2];
        for (int a = 0; a < 2; ++a)
            for (int e = 0; e < 2; ++e) {
                if (!P2P_TAP_LIVE(a, e, fr, fc)) {
                    prob[a][e] = 0.f;
                    pred[0][a][e] = pred[1][a][e] = pred[2][a][e] = 0.f;
                } else if (fi[a] >= 0 && fi[a] < 128 && fj[e] >= 0 && fj[e] < 128) {
                    const int idx = fi[a] * 128 + fj[e];
                    if (bk) {                      // filtered planes of a float32 image hold float32 values (AaItem::round32)
                        prob[a][e] = (float)bk[idx];
                        for (int ch = 0; ch < 3; ++ch) pred[ch][a][e] = (float)bk[(1 + ch) * 16384 + idx];
                    } else {
                        double pr, g, pr3[3];
                        cand_raw(y2c + (size_t)idx * 4, &pr, &g, pr3);
                        prob[a][e] = (float)pr;
                        for (int ch = 0; ch < 3; ++ch) pred[ch][a][e] = (float)pr3[ch];
                    }
                } else {
                    prob[a][e] = 1.f;
                    pred[0][a][e] = pred[1][a][e] = pred[2][a][e] = 0.5f;
                }
            }
        const float pr = (float)clip_warp((double)lerp2_f32(prob[0][0], prob[0][1], prob[1][0], prob[1][1], fr.d, fc.d), R.pmin, R.pmax, 1.0);
        o.below = pr < (float)th_i;                                                     // float32 array < python float: a float32 comparison
        o.valid = o.non_gray && o.below;
        for (int ch = 0; ch < 3; ++ch) {
            const float w = (float)clip_warp((double)lerp2_f32(pred[ch][0][0], pred[ch][0][1], pred[ch][1][0], pred[ch][1][1], fr.d, fc.d), R.qmin, R.qmax, 0.5);
            const float v = w * 255.0f;                                                 // :144  float32 array * 255
            o.q[ch] = (unsigned char)(int)v;                                            // uint8 canvas: truncation (:152-154)
        }
        return o;
    }
    double prob[2][2], ng[2][2], pred[3][2][2];
    for (int a = 0; a < 2; ++a)
        for (int e = 0; e < 2; ++e) {
            if (!P2P_TAP_LIVE(a, e, tr, tc)) {
                prob[a][e] = 0.0; ng[a][e] = 0.0;
                pred[0][a][e] = pred[1][a][e] = pred[2][a][e] = 0.0;
            } else if (ri[a] >= 0 && ri[a] < 128 && cj[e] >= 0 && cj[e] < 128) {
                const int idx = ri[a] * 128 + cj[e];
                if (bk) {
                    prob[a][e] = bk[idx];
                    for (int ch = 0; ch < 3; ++ch) pred[ch][a][e] = bk[(1 + ch) * 16384 + idx];
                    ng[a][e] = bk[4 * 16384 + idx];
                } else {
                    double pr3[3];
                    cand_raw(y2c + (size_t)idx * 4, &prob[a][e], &ng[a][e], pr3);
                    for (int ch = 0; ch < 3; ++ch) pred[ch][a][e] = pr3[ch];
                }
            } else {          // resize(..., mode='constant', cval=1 / 0.5 / 0)       :134,144,146
                prob[a][e] = 1.0; ng[a][e] = 0.0;
                pred[0][a][e] = pred[1][a][e] = pred[2][a][e] = 0.5;
            }
        }
    o.non_gray = clip_warp(lerp2(ng[0][0], ng[0][1], ng[1][0], ng[1][1], tr.d, tc.d), R.gmin, R.gmax, 0.0) > 0.9;
    const double pr = clip_warp(lerp2(prob[0][0], prob[0][1], prob[1][0], prob[1][1], tr.d, tc.d), R.pmin, R.pmax, 1.0);
    o.below = pr < th_i;
    o.valid = o.non_gray && o.below;                                                // :203-204
    for (int ch = 0; ch < 3; ++ch) {
        const double v = clip_warp(lerp2(pred[ch][0][0], pred[ch][0][1], pred[ch][1][0], pred[ch][1][1], tr.d, tc.d), R.qmin, R.qmax, 0.5) * 255;
        o.q[ch] = (unsigned char)(int)v;                                            // uint8 canvas: truncation (:152-154)
    }
    return o;
}

// [min, max] of the raw 128x128 maps of every candidate (the warp inputs when no anti-aliasing filter ran)
__global__ __launch_bounds__(1024) void cand_range_kernel(const float* __restrict__ y2, CandRange* __restrict__ out)
{
    __shared__ double s_v[16][6];
    const float* y2c = y2 + (size_t)blockIdx.x * 16384 * 4;
    double v[6] = {1e300, -1e300, 1e300, -1e300, 1e300, -1e300};
    for (int p = threadIdx.x; p < 16384; p += blockDim.x) {
        double prob, ng, pred[3];
        cand_raw(y2c + (size_t)p * 4, &prob, &ng, pred);
        v[0] = fmin(v[0], prob); v[1] = fmax(v[1], prob);
        for (int ch = 0; ch < 3; ++ch) { v[2] = fmin(v[2], pred[ch]); v[3] = fmax(v[3], pred[ch]); }
        v[4] = fmin(v[4], ng); v[5] = fmax(v[5], ng);
    }
    for (int o = 32; o > 0; o >>= 1)
        for (int k = 0; k < 6; ++k) {
            const double w = __shfl_down(v[k], o, 64);
            v[k] = (k & 1) ? fmax(v[k], w) : fmin(v[k], w);
        }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 6; ++k) s_v[threadIdx.x >> 6][k] = v[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
            for (int k = 0; k < 6; ++k) v[k] = (k & 1) ? fmax(v[k], s_v[w][k]) : fmin(v[k], s_v[w][k]);
        CandRange R;
        R.pmin = v[0]; R.pmax = v[1]; R.qmin = v[2]; R.qmax = v[3]; R.gmin = v[4]; R.gmax = v[5];
        out[blockIdx.x] = R;
    }
}

// ------------------------------------------------------------------------------------------
// K6: correspondences per candidate, compacted in row-major order  (recognition.py:134-151,196-213)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void cand_corr_kernel(const DetInfo* __restrict__ dets, const Stage1* __restrict__ s1,
                                                         const float* __restrict__ y2, int K, float* __restrict__ corr,
                                                         CandStat* __restrict__ cstat, PnpProblem* __restrict__ probs,
                                                         const CandRange* __restrict__ crange, AaPtrs aa)
{
    __shared__ int s_wave[16];
    const int NT = blockDim.x, n_waves = NT >> 6;      // 256 threads, or 1024 for a handful of candidates (a 128-px candidate = 64 trips of 256)
    __shared__ int s_ng;
    __shared__ unsigned long long s_sv, s_su;
    const int cand = blockIdx.x;
    const int d = cand / K, slot = cand - d * K;
    const DetInfo& D = dets[d];
    const Stage1& S = s1[d];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* pts = corr + D.corr_off + (size_t)slot * 5 * D.corr_cap;
    if (tid == 0) { s_ng = 0; s_sv = 0; s_su = 0; }
    __syncthreads();
    int total = 0;
    if (slot < D.n_th && S.valid2[slot]) {
        const Boxes& b = S.b2;
        const int S2 = b.v2_ori - b.v1_ori, S2w = b.u2_ori - b.u1_ori;
        const int h = b.v2 - b.v1, w = b.u2 - b.u1;
        const int npx = h * w;
        const float* y2c = y2 + (size_t)cand * 16384 * 4;
        const double* bk = (aa.k3 && aa.k3[cand * 5].radius > 0) ? aa.k3[cand * 5].a : nullptr;
        const CandRange R = crange[cand];
        int ng_cnt = 0;
        unsigned long long sv = 0, su = 0;
        float* PX = pts; float* PY = pts + D.corr_cap; float* PZ = pts + 2 * (size_t)D.corr_cap;
        float* PU = pts + 3 * (size_t)D.corr_cap; float* PV = pts + 4 * (size_t)D.corr_cap;
        for (int base = 0; base < npx; base += NT) {
            const int p = base + tid;
            bool valid = false;
            CandPixel cp;
            int rr = 0, cc = 0;
            if (p < npx) {
                rr = p / w; cc = p - rr * w;
                cp = cand_pixel(y2c, bk, R, b.vv1 + rr, b.uu1 + cc, S2, S2w, D.th_i, D.aa == 1);
                valid = cp.valid;
                if (cp.non_gray) { ++ng_cnt; sv += (unsigned)(b.v1 + rr); su += (unsigned)(b.u1 + cc); }
            }
            const unsigned long long bal = __ballot(valid);
            const int before = __popcll(bal & ((1ULL << lane) - 1ULL));
            if (lane == 0) s_wave[wave] = __popcll(bal);
            __syncthreads();
            int off = total, chunk = 0;
            for (int k = 0; k < n_waves; ++k) {
                if (k < wave) off += s_wave[k];
                chunk += s_wave[k];
            }
            if (valid) {
                const int o = off + before;       // o < corr_cap: the clipped stage-2 region fits the stage-1 square
                for (int ch = 0; ch < 3; ++ch) {
                    double x = (double)cp.q[ch];
                    x = x / 255;                                                   // :198
                    x = x * 2 - 1;                                                 // :199
                    x = x * D.obj_scale[ch] + D.obj_ct[ch];                        // :200-202
                    (ch == 0 ? PX : ch == 1 ? PY : PZ)[o] = (float)x;             // solvePnPRansac stores float32
                }
                PU[o] = (float)(b.u1 + cc);                                        // :207-209 (u, v) + (u1, v1)
                PV[o] = (float)(b.v1 + rr);
            }
            total += chunk;
            __syncthreads();
        }
        atomicAdd(&s_ng, ng_cnt);
        atomicAdd(&s_sv, sv);
        atomicAdd(&s_su, su);
    }
    __syncthreads();
    if (tid == 0) {
        CandStat cs;
        cs.n_non_gray = s_ng; cs.n_corr = total; cs.sum_v = (long long)s_sv; cs.sum_u = (long long)s_su;
        cstat[cand] = cs;
        PnpProblem pb;
        pb.pts = pts; pb.cap = D.corr_cap;
        // n_non_gray < 10 -> the reference skips the candidate before PnP (:149-150)
        pb.n = (s_ng >= 10) ? total : 0;
        for (int k = 0; k < 9; ++k) pb.K[k] = D.K[k];
        pb.mask = nullptr;
        probs[cand] = pb;
    }
}

// The same work for a HANDFUL of candidates (one detection at a time: three), as two launches over CORR_SEG pixel segments per candidate:
// one workgroup per candidate spent 75 us walking its 16 384 pixels (the per-pixel evaluation is ~300 fp64 instructions); eight
// segments on eight CUs evaluate them at once and leave a 4-byte record per pixel, then the compaction -- which needs the counts of
// the segments before it, hence the second launch -- writes the correspondences in the same row-major order.
constexpr int CORR_SEG = 8;
struct CorrSeg { int n_valid, n_ng; unsigned long long sv, su; };

__global__ __launch_bounds__(256) void cand_eval_kernel(const DetInfo* __restrict__ dets, const Stage1* __restrict__ s1, const float* __restrict__ y2,
                                                        int K, unsigned* __restrict__ rec, CorrSeg* __restrict__ segs,
                                                        const CandRange* __restrict__ crange, AaPtrs aa)
{
    __shared__ int s_nv[4], s_ng[4];
    __shared__ unsigned long long s_sv[4], s_su[4];
    const int cand = blockIdx.y, seg = blockIdx.x;
    const int d = cand / K, slot = cand - d * K;
    const DetInfo& D = dets[d];
    const Stage1& S = s1[d];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int nv = 0, ng = 0;
    unsigned long long sv = 0, su = 0;
    if (slot < D.n_th && S.valid2[slot]) {
        const Boxes& b = S.b2;
        const int S2 = b.v2_ori - b.v1_ori, S2w = b.u2_ori - b.u1_ori;
        const int h = b.v2 - b.v1, w = b.u2 - b.u1;
        const int npx = h * w, per = (npx + CORR_SEG - 1) / CORR_SEG;
        const float* y2c = y2 + (size_t)cand * 16384 * 4;
        const double* bk = (aa.k3 && aa.k3[cand * 5].radius > 0) ? aa.k3[cand * 5].a : nullptr;
        const CandRange R = crange[cand];
        unsigned* r = rec + D.corr_off / 5 + (size_t)slot * D.corr_cap;
        for (int p = seg * per + tid; p < min(npx, (seg + 1) * per); p += 256) {
            const int rr = p / w, cc = p - rr * w;
            const CandPixel cp = cand_pixel(y2c, bk, R, b.vv1 + rr, b.uu1 + cc, S2, S2w, D.th_i, D.aa == 1);
            r[p] = (unsigned)cp.q[0] | ((unsigned)cp.q[1] << 8) | ((unsigned)cp.q[2] << 16) | (cp.valid ? 1u << 24 : 0u);
            nv += cp.valid;
            if (cp.non_gray) { ++ng; sv += (unsigned)(b.v1 + rr); su += (unsigned)(b.u1 + cc); }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nv += __shfl_down(nv, o, 64); ng += __shfl_down(ng, o, 64);
        sv += __shfl_down(sv, o, 64); su += __shfl_down(su, o, 64);
    }
    if (lane == 0) { s_nv[wave] = nv; s_ng[wave] = ng; s_sv[wave] = sv; s_su[wave] = su; }
    __syncthreads();
    if (tid == 0) {
        CorrSeg o;
        o.n_valid = s_nv[0] + s_nv[1] + s_nv[2] + s_nv[3]; o.n_ng = s_ng[0] + s_ng[1] + s_ng[2] + s_ng[3];
        o.sv = s_sv[0] + s_sv[1] + s_sv[2] + s_sv[3]; o.su = s_su[0] + s_su[1] + s_su[2] + s_su[3];
        segs[cand * CORR_SEG + seg] = o;
    }
}

__global__ __launch_bounds__(256) void cand_compact_kernel(const DetInfo* __restrict__ dets, const Stage1* __restrict__ s1, int K,
                                                           const unsigned* __restrict__ rec, const CorrSeg* __restrict__ segs,
                                                           float* __restrict__ corr, CandStat* __restrict__ cstat, PnpProblem* __restrict__ probs)
{
    __shared__ int s_wave[4];
    const int cand = blockIdx.y, seg = blockIdx.x;
    const int d = cand / K, slot = cand - d * K;
    const DetInfo& D = dets[d];
    const Stage1& S = s1[d];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* pts = corr + D.corr_off + (size_t)slot * 5 * D.corr_cap;
    int before = 0, all = 0, ng = 0;
    unsigned long long sv = 0, su = 0;
    for (int k = 0; k < CORR_SEG; ++k) {
        const CorrSeg c = segs[cand * CORR_SEG + k];
        if (k < seg) before += c.n_valid;
        all += c.n_valid; ng += c.n_ng; sv += c.sv; su += c.su;
    }
    if (slot < D.n_th && S.valid2[slot]) {
        const Boxes& b = S.b2;
        const int h = b.v2 - b.v1, w = b.u2 - b.u1;
        const int npx = h * w, per = (npx + CORR_SEG - 1) / CORR_SEG;
        const unsigned* r = rec + D.corr_off / 5 + (size_t)slot * D.corr_cap;
        float* PX = pts; float* PY = pts + D.corr_cap; float* PZ = pts + 2 * (size_t)D.corr_cap;
        float* PU = pts + 3 * (size_t)D.corr_cap; float* PV = pts + 4 * (size_t)D.corr_cap;
        int total = before;
        const int p_end = min(npx, (seg + 1) * per);
        for (int base = seg * per; base < p_end; base += 256) {
            const int p = base + tid;
            const unsigned v = p < p_end ? r[p] : 0u;
            const bool valid = (v >> 24) & 1u;
            const unsigned long long bal = __ballot(valid);
            const int rank = __popcll(bal & ((1ULL << lane) - 1ULL));
            if (lane == 0) s_wave[wave] = __popcll(bal);
            __syncthreads();
            int off = total, chunk = 0;
            for (int k = 0; k < 4; ++k) {
                if (k < wave) off += s_wave[k];
                chunk += s_wave[k];
            }
            if (valid) {
                const int o = off + rank;
                const int rr = p / w, cc = p - rr * w;
                for (int ch = 0; ch < 3; ++ch) {
                    double x = (double)((v >> (8 * ch)) & 255u);
                    x = x / 255;                                                   // :198
                    x = x * 2 - 1;                                                 // :199
                    x = x * D.obj_scale[ch] + D.obj_ct[ch];                        // :200-202
                    (ch == 0 ? PX : ch == 1 ? PY : PZ)[o] = (float)x;             // solvePnPRansac stores float32
                }
                PU[o] = (float)(b.u1 + cc);                                        // :207-209 (u, v) + (u1, v1)
                PV[o] = (float)(b.v1 + rr);
            }
            total += chunk;
            __syncthreads();
        }
    }
    if (seg == 0 && tid == 0) {
        const bool on = slot < D.n_th && S.valid2[slot];
        CandStat cs;
        cs.n_non_gray = on ? ng : 0; cs.n_corr = on ? all : 0; cs.sum_v = on ? (long long)sv : 0; cs.sum_u = on ? (long long)su : 0;
        cstat[cand] = cs;
        PnpProblem pb;
        pb.pts = pts; pb.cap = D.corr_cap;
        pb.n = (on && ng >= 10) ? all : 0;                                         // n_non_gray < 10 -> skipped before PnP (:149-150)
        for (int k = 0; k < 9; ++k) pb.K[k] = D.K[k];
        pb.mask = nullptr;
        probs[cand] = pb;
    }
}

// ------------------------------------------------------------------------------------------
// K8: candidate selection  (recognition.py:130-131,158-178,189-193)
// ------------------------------------------------------------------------------------------
__global__ void select_kernel(const DetInfo* __restrict__ dets, const Stage1* __restrict__ s1,
                              const CandStat* __restrict__ cstat, const PnpResult* __restrict__ res, int K, int n_det,
                              p2p_pose* __restrict__ poses)
{
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n_det) return;
    const DetInfo& D = dets[d];
    const Stage1& S = s1[d];
    p2p_pose o;
    memset(&o, 0, sizeof(o));
    o.R[0] = o.R[4] = o.R[8] = 1.0;
    o.n_inliers = -1; o.best_slot = -1; o.frac_inlier = -1.0;
    o.n_init_mask = S.n_init_mask;
    o.bbox_t[0] = D.b1.v1; o.bbox_t[1] = D.b1.v2; o.bbox_t[2] = D.b1.u1; o.bbox_t[3] = D.b1.u2;
    o.n_candidates = S.n_cand;
    if (!D.ok1) { o.status = P2P_POSE_CROP_TOO_SMALL; poses[d] = o; return; }
    if (S.n_cand == 0) { o.status = P2P_POSE_NO_CANDIDATE; poses[d] = o; return; }
    // the reference returns the box of the LAST candidate iterated (:133, :193)
    o.bbox_t[0] = S.b2.v1; o.bbox_t[1] = S.b2.v2; o.bbox_t[2] = S.b2.u1; o.bbox_t[3] = S.b2.u2;
    int max_inlier = -1;
    double min_dist = 9999999;
    for (int slot = 0; slot < D.n_th; ++slot) {
        if (!S.valid2[slot]) continue;
        const CandStat& cs = cstat[d * K + slot];
        if (cs.n_non_gray < 10) continue;                                           // :149-150
        const PnpResult& r = res[d * K + slot];
        const double ct_v = (double)cs.sum_v / (double)cs.n_non_gray, ct_u = (double)cs.sum_u / (double)cs.n_non_gray;
        double dist;
        if (r.t[2] == 0) dist = 99999;                                             // :163-164
        else {
            const double pu = D.K[0] * r.t[0] / r.t[2] + D.K[2];
            const double pv = D.K[4] * r.t[1] / r.t[2] + D.K[5];
            dist = ((pv - ct_v) * (pv - ct_v) + (pu - ct_u) * (pu - ct_u)) / ((double)r.n_inliers + 1E-6);   // :168
        }
        if (dist < min_dist) {                                                      // :170-174
            for (int k = 0; k < 9; ++k) o.R[k] = r.R[k];
            for (int k = 0; k < 3; ++k) o.t[k] = r.t[k];
            max_inlier = r.n_inliers;
            min_dist = dist;
            o.best_slot = slot;
            o.ransac_iters = r.iters;
        }
    }
    o.n_inliers = max_inlier;
    if (max_inlier == -1) {                                                         // :189-191
        o.status = P2P_POSE_PNP_FAILED;
        o.R[0] = o.R[4] = o.R[8] = 1.0; o.R[1] = o.R[2] = o.R[3] = o.R[5] = o.R[6] = o.R[7] = 0.0;
        o.t[0] = o.t[1] = o.t[2] = 0.0;
        o.best_slot = -1;
    } else {
        o.status = P2P_POSE_OK;
        o.frac_inlier = (double)max_inlier / (double)S.n_init_mask;                 // :193
    }
    poses[d] = o;
}

// ------------------------------------------------------------------------------------------
// K9 (optional outputs): valid_mask_full and img_pred_f of the selected candidate (:175-177).  Both leave the device
// COMPACT (row-major over the candidate's clipped box, which is all that is non-zero of the full-frame mask): 16 KB per
// 128-px detection over PCIe instead of a 300 KB frame; finish_batch() pastes the mask into the caller's full frames.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void render_best_kernel(const DetInfo* __restrict__ dets, const Stage1* __restrict__ s1,
                                                          const p2p_pose* __restrict__ poses, const float* __restrict__ y2,
                                                          int K, unsigned char* __restrict__ mask, long long mask_stride,
                                                          unsigned char* __restrict__ pred, long long pred_stride,
                                                          const CandRange* __restrict__ crange, AaPtrs aa)
{
    const int d = blockIdx.y;
    const p2p_pose& P = poses[d];
    if (P.status != P2P_POSE_OK) return;
    const DetInfo& D = dets[d];
    const Boxes& b = s1[d].b2;
    const int S2 = b.v2_ori - b.v1_ori, S2w = b.u2_ori - b.u1_ori;
    const int h = b.v2 - b.v1, w = b.u2 - b.u1;
    const int cand = d * K + P.best_slot;
    const float* y2c = y2 + (size_t)cand * 16384 * 4;
    const double* bk = (aa.k3 && aa.k3[cand * 5].radius > 0) ? aa.k3[cand * 5].a : nullptr;
    const CandRange R = crange[cand];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < h * w; p += gridDim.x * 256) {
        const int rr = p / w, cc = p - rr * w;
        const CandPixel cp = cand_pixel(y2c, bk, R, b.vv1 + rr, b.uu1 + cc, S2, S2w, D.th_i, D.aa == 1);
        if (mask && p < mask_stride) mask[(size_t)d * mask_stride + p] = cp.valid ? 1 : 0;      // compact: row-major over the clipped box
        if (pred && (long long)(p + 1) * 3 <= pred_stride) {
            unsigned char* q = pred + (size_t)d * pred_stride + (size_t)p * 3;
            q[0] = cp.q[0]; q[1] = cp.q[1]; q[2] = cp.q[2];
        }
    }
}

// ------------------------------------------------------------------------------------------
// score_type 2: |det_mask AND valid_mask|, |det_mask OR valid_mask| (reference
// tools/5_evaluation_bop_basic.py:307-316).  valid_mask_full is zero outside the selected
// candidate's clipped box, so: union = |det_mask| + |valid| - inter, with |det_mask| over the frame.
// stats[d] = {inter, det_count, valid_count}
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_iou_kernel(const DetInfo* __restrict__ dets, const Stage1* __restrict__ s1,
                                                       const p2p_pose* __restrict__ poses, const float* __restrict__ y2, int K,
                                                       const unsigned char* __restrict__ det_mask, long long stride,
                                                       unsigned long long* __restrict__ stats,
                                                       const CandRange* __restrict__ crange, AaPtrs aa)
{
    const int d = blockIdx.y;
    const DetInfo& D = dets[d];
    const unsigned char* dm = det_mask + (size_t)d * stride;
    unsigned long long inter = 0, dcount = 0, vcount = 0;
    const long long npx = (long long)D.H * D.W;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < npx; p += (long long)gridDim.x * 256) dcount += dm[p] != 0;
    const p2p_pose& P = poses[d];
    if (P.status == P2P_POSE_OK) {
        const Boxes& b = s1[d].b2;
        const int S2 = b.v2_ori - b.v1_ori, S2w = b.u2_ori - b.u1_ori;
        const int h = b.v2 - b.v1, w = b.u2 - b.u1;
        const int cand = d * K + P.best_slot;
        const float* y2c = y2 + (size_t)cand * 16384 * 4;
        const double* bk = (aa.k3 && aa.k3[cand * 5].radius > 0) ? aa.k3[cand * 5].a : nullptr;
        const CandRange R = crange[cand];
        for (int p = blockIdx.x * 256 + threadIdx.x; p < h * w; p += gridDim.x * 256) {
            const int rr = p / w, cc = p - rr * w;
            const CandPixel cp = cand_pixel(y2c, bk, R, b.vv1 + rr, b.uu1 + cc, S2, S2w, D.th_i, D.aa == 1);
            if (cp.valid) {
                ++vcount;
                inter += dm[(size_t)(b.v1 + rr) * D.W + (b.u1 + cc)] != 0;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        inter += __shfl_down(inter, off, 64); dcount += __shfl_down(dcount, off, 64); vcount += __shfl_down(vcount, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (inter) atomicAdd(&stats[3 * d], inter);
        if (dcount) atomicAdd(&stats[3 * d + 1], dcount);
        if (vcount) atomicAdd(&stats[3 * d + 2], vcount);
    }
}

// ------------------------------------------------------------------------------------------
// anti-aliased resizes (p2p_est_pose_opts.resize_anti_aliasing; filter itself: resize_aa.hip)
// ------------------------------------------------------------------------------------------
__device__ inline void aa_item_off(AaItem& I)
{
    I.a = I.tmp = nullptr; I.w = nullptr;
    I.H = I.W = I.C = 0; I.radius = 0; I.mode = 0; I.round32 = 0; I.cval = 0; I.vmin = I.vmax = 0;
    I.r0 = I.r1 = I.c0 = I.c1 = 0;
}

// phase 0 (before stage 1): every descriptor off, stage-1 canvases planned.  phase 1 (after the stage-1 reductions, which
// fix the stage-2 geometry on the device): stage-2 canvases, keep masks, back-resize planes.  One thread per (detection, slot).
__global__ void aa_plan_kernel(const DetInfo* __restrict__ dets, const Stage1* __restrict__ s1, int n, int K, int phase, AaTable tab,
                               AaBufs B, AaPtrs aa)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * K) return;
    const int d = t / K, slot = t - d * K;
    const DetInfo& D = dets[d];
    const int S1 = D.b1.v2_ori - D.b1.v1_ori;
    const size_t cap3 = (size_t)D.corr_cap * 3;
    if (phase == 0) {
        aa_item_off(aa.k1[t]);
        for (int j = 0; j < 5; ++j) aa_item_off(aa.k3[t * 5 + j]);
        if (slot == 0) {
            AaItem I;
            aa_item_off(I);
            if (D.aa && D.ok1 && S1 > 128 && S1 <= tab.max_side && tab.rad[S1] > 0) {
                I.a = B.cv + D.cv_off; I.tmp = B.cv_tmp + D.cv_off;
                I.H = S1; I.W = D.b1.u2_ori - D.b1.u1_ori; I.C = 3;
                I.radius = tab.rad[S1]; I.w = tab.w + tab.off[S1];
            }
            aa.k0[d] = I;
        }
        return;
    }
    const Stage1& S = s1[d];
    if (!D.aa || !D.ok1 || slot >= D.n_th) return;
    // :103 shrinks the 128x128 keep mask to the stage-1 square -- a BOOL image: scikit-image 0.17 / 0.18 do not filter those
    // (anti_aliasing defaults to "not bool"); 0.15 / 0.16 run the filter into a bool array: keep_filter_kernel (generation 2), not an AaItem.
    if (!S.valid2[slot]) return;
    const int S2 = S.b2.v2_ori - S.b2.v1_ori;
    if (S2 > 128 && S2 <= tab.max_side && tab.rad[S2] > 0) {              // :121 shrinks the stage-2 canvas to 128
        AaItem& I = aa.k1[t];
        I.a = B.cv + D.cv_off + (size_t)(1 + slot) * cap3; I.tmp = B.cv_tmp + D.cv_off + (size_t)(1 + slot) * cap3;
        I.H = S2; I.W = S.b2.u2_ori - S.b2.u1_ori; I.C = 3; I.radius = tab.rad[S2]; I.w = tab.w + tab.off[S2];
        // Region of interest: the canvas is zero wherever the kept mask is (recognition.py:113-120), and the kept mask lies inside the bounding box
        // of non_gray (Stage1::bb, 128-px rows / columns of the stage-1 square).  A stage-1 canvas row r is foreground only if one of its two
        // bilinear taps floor(src), ceil(src), src = (r + 0.5) 128 / S1 - 0.5, falls into [bb0, bb2]: r in ((bb0 - 0.5) S1 / 128 - 0.5,
        // (bb2 + 1.5) S1 / 128 - 0.5); one pixel of slack on either side; stage-1 row r is stage-2 canvas row r + v1_ori(1) - v1_ori(2).
        const int S1w = D.b1.u2_ori - D.b1.u1_ori;
        const int dv = D.b1.v1_ori - S.b2.v1_ori, du = D.b1.u1_ori - S.b2.u1_ori;
        int r0 = (int)floor(((double)S.bb[0] - 0.5) * S1 / 128.0 - 0.5) - 1 + dv, r1 = (int)ceil(((double)S.bb[2] + 1.5) * S1 / 128.0 - 0.5) + 2 + dv;
        int c0 = (int)floor(((double)S.bb[1] - 0.5) * S1w / 128.0 - 0.5) - 1 + du, c1 = (int)ceil(((double)S.bb[3] + 1.5) * S1w / 128.0 - 0.5) + 2 + du;
        r0 = max(r0, 0); c0 = max(c0, 0); r1 = min(r1, I.H); c1 = min(c1, I.W);
        if (r1 <= r0 || c1 <= c0) { r0 = c0 = 0; r1 = c1 = 1; }      // nothing of the object inside the canvas: one (zero) pixel
        I.r0 = r0; I.r1 = r1; I.c0 = c0; I.c1 = c1;
    }
    if (S2 < 128 && S2 > 0 && tab.rad[S2] > 0)                           // :134,144,146 shrink the network output maps
        for (int j = 0; j < 5; ++j) {
            AaItem& I = aa.k3[t * 5 + j];
            I.a = B.bk + ((size_t)t * 5 + j) * 16384; I.tmp = B.bk_tmp + ((size_t)t * 5 + j) * 16384;
            I.H = I.W = 128; I.C = 1; I.radius = tab.rad[S2]; I.w = tab.w + tab.off[S2];
            I.mode = 1; I.cval = j == 0 ? 1.0 : (j == 4 ? 0.0 : 0.5);
            I.round32 = j < 4;                                            // prob and img_pred are float32 arrays, non_gray.astype(float) is float64
        }
}

// Generation 2 (scikit-image 0.15 / 0.16), stage-1 sides < 128: resize() runs scipy.ndimage.gaussian_filter on the BOOL keep mask of
// recognition.py:103 before shrinking it, and scipy's output array takes the input's dtype -- every axis pass computes, in double,
//     t = x[0] w[0];  for d = radius .. 1:  t += (x[-d] + x[+d]) w[d]            (NI_Correlate1D, symmetric branch; border 'constant', cval 0)
// and stores (npy_bool)t: a pixel survives a pass only where its weighted sum reaches 1.0.  An erosion whose outcome hangs on how the weights'
// sum rounds: for some crop sides part of the mask survives, for others NOTHING does (tests/test_real_libraries_cpu.py holds this restatement to
// scipy itself for every side).  One workgroup per candidate, both passes through LDS; the filtered plane feeds keep_ori_at(), its range clip=True.
__global__ __launch_bounds__(256) void keep_filter_kernel(const DetInfo* __restrict__ dets, Stage1* __restrict__ s1, const float* __restrict__ y1,
                                                          int K, AaTable tab, unsigned char* __restrict__ keepf)
{
    __shared__ unsigned char p0[16384], p1[16384];
    __shared__ int s_any, s_all;
    const int cand = blockIdx.x, d = cand / K, slot = cand - d * K;
    const DetInfo& D = dets[d];
    Stage1& S = s1[d];
    const int S1 = D.b1.v2_ori - D.b1.v1_ori;
    const bool on = D.aa == 2 && D.ok1 && slot < D.n_th && S.valid2[slot] && S1 < 128 && S1 > 0 && tab.rad[S1] > 0;
    if (!on) {
        if (threadIdx.x == 0 && slot < MAX_TH) S.keepf[slot] = 0;
        return;
    }
    const int r = tab.rad[S1];
    const double* w = tab.w + tab.off[S1];
    const float* y1d = y1 + (size_t)d * 16384 * 4;
    const float th = D.th_o[slot];
    if (threadIdx.x == 0) { s_any = 0; s_all = 1; }
    for (int p = threadIdx.x; p < 16384; p += 256) {
        const float* q = y1d + (size_t)p * 4;
        p0[p] = (non_gray_at(q) && q[3] < th) ? 1 : 0;
    }
    __syncthreads();
    for (int p = threadIdx.x; p < 16384; p += 256) {              // axis 0 (rows)
        const int v = p >> 7, u = p & 127;
        double t = (double)p0[p] * w[0];
        for (int dd = r; dd >= 1; --dd) {
            const double a = v - dd >= 0 ? (double)p0[(v - dd) * 128 + u] : 0.0;
            const double b = v + dd < 128 ? (double)p0[(v + dd) * 128 + u] : 0.0;
            t += (a + b) * w[dd];
        }
        p1[p] = (unsigned char)t ? 1 : 0;                         // (npy_bool)t: truncation toward zero
    }
    __syncthreads();
    int any = 0, all = 1;
    unsigned char* out = keepf + (size_t)cand * 16384;
    for (int p = threadIdx.x; p < 16384; p += 256) {              // axis 1 (columns)
        const int v = p >> 7, u = p & 127;
        double t = (double)p1[p] * w[0];
        for (int dd = r; dd >= 1; --dd) {
            const double a = u - dd >= 0 ? (double)p1[v * 128 + u - dd] : 0.0;
            const double b = u + dd < 128 ? (double)p1[v * 128 + u + dd] : 0.0;
            t += (a + b) * w[dd];
        }
        const unsigned char k = (unsigned char)t ? 1 : 0;
        out[p] = k;
        any |= k; all &= k;
    }
    if (any) atomicOr(&s_any, 1);
    if (!all) atomicAnd(&s_all, 0);
    __syncthreads();
    if (threadIdx.x == 0) {
        S.keepf[slot] = 1;
        S.kmin[slot] = s_all ? 1.0 : 0.0;                         // range of the FILTERED mask as float (clip=True of the :103 resize)
        S.kmax[slot] = s_any ? 1.0 : 0.0;
    }
}

// stage-1 canvas (recognition.py:75-81): zeros, the normalised clipped crop pasted in.  grid (64, n)
__global__ __launch_bounds__(256) void aa_canvas1_kernel(const DetInfo* __restrict__ dets, AaPtrs aa)
{
    const int d = blockIdx.y;
    const AaItem& I = aa.k0[d];
    if (I.radius <= 0) return;
    const DetInfo& D = dets[d];
    const Boxes& b = D.b1;
    const int npx = I.H * I.W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256) {
        const int r = p / I.W, c = p - r * I.W;
        const bool in = r >= b.vv1 && r < b.vv2 && c >= b.uu1 && c < b.uu2;
        for (int ch = 0; ch < 3; ++ch) I.a[(size_t)p * 3 + ch] = in ? frame_px(D, b.v1 + r - b.vv1, b.u1 + c - b.uu1, ch) : 0.0;
    }
}

// stage-2 canvases (recognition.py:113-120).  grid (64, n*K)
__global__ __launch_bounds__(256) void aa_canvas2_kernel(const DetInfo* __restrict__ dets, const Stage1* __restrict__ s1,
                                                         const float* __restrict__ y1, int K, AaPtrs aa)
{
    const int t = blockIdx.y;
    const AaItem& I = aa.k1[t];
    if (I.radius <= 0) return;
    const int d = t / K, slot = t - d * K;
    const DetInfo& D = dets[d];
    const Stage1& S = s1[d];
    const float* y1d = y1 + (size_t)d * 16384 * 4;
    const int rw = I.c1 - I.c0, npx = (I.r1 - I.r0) * rw;      // the region of interest (aa_plan_kernel): the canvas is zero outside it and not stored there
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256) {
        const int r = I.r0 + p / rw, c = I.c0 + p % rw;
        int fy = 0, fx = 0;
        const bool fg = stage2_fg(D, S, y1d, slot, r, c, &fy, &fx);
        for (int ch = 0; ch < 3; ++ch) I.a[((size_t)r * I.W + c) * 3 + ch] = fg ? frame_px(D, fy, fx, ch) : 0.0;
    }
}

// the five maps of a candidate the back-resizes read (recognition.py:134-146).  grid (64, n*K)
__global__ __launch_bounds__(256) void aa_back_fill_kernel(const float* __restrict__ y2, AaPtrs aa)
{
    const int t = blockIdx.y;
    const AaItem& I = aa.k3[t * 5];
    if (I.radius <= 0) return;
    const float* y2c = y2 + (size_t)t * 16384 * 4;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < 16384; p += gridDim.x * 256) {
        double prob, ng, pred[3];
        cand_raw(y2c + (size_t)p * 4, &prob, &ng, pred);
        I.a[p] = prob;
        for (int ch = 0; ch < 3; ++ch) I.a[(size_t)(1 + ch) * 16384 + p] = pred[ch];
        I.a[(size_t)4 * 16384 + p] = ng;
    }
}

// filtered ranges -> CandRange.  one thread per candidate
__global__ void aa_back_range_kernel(CandRange* __restrict__ crange, int n_cand, AaPtrs aa)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_cand) return;
    const AaItem* I = aa.k3 + (size_t)t * 5;
    if (I[0].radius <= 0) return;
    CandRange R;
    R.pmin = I[0].vmin; R.pmax = I[0].vmax;
    R.qmin = fmin(I[1].vmin, fmin(I[2].vmin, I[3].vmin));
    R.qmax = fmax(I[1].vmax, fmax(I[2].vmax, I[3].vmax));
    R.gmin = I[4].vmin; R.gmax = I[4].vmax;
    crange[t] = R;
}

// ------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------
// The work of a batch falls into four pieces, all enqueued without ever returning to the host in between:
//   front   per-detection constants, frames, stage-1 network inputs                          (enqueue_front)
//   pass 1  generator over the n stage-1 inputs
//   mid     stage-1 reductions, stage-2 geometry, stage-2 network inputs                      (enqueue_mid)
//   pass 2  generator over the n*K stage-2 inputs
//   tail    correspondences, PnP-RANSAC, selection, optional outputs, D2H                     (enqueue_tail)
// A blocking call runs them back to back.  Asynchronous batches (submit / collect) run as a STREAM on one HIP stream (the chain
// pass 1 -> mid -> pass 2 of consecutive batches is serial), with every batch's latency-bound tail on the tail stream under the
// next batch's passes.  With p2p_est_pose_opts.merge_stream_passes a batch's pass 2 is NOT enqueued by its own submit: the next
// submit merges it with its own pass 1 into one generator pass over [x2(k) | x1(k+1)] (one 1024-input pass per step instead of
// a 256- and a 768-input pass), or collect(k) runs it alone.  That pays for small batches; at 256 detections per batch it
// measures equal (7570 vs 7610 crops/s) and 30-object batches lose a little to the split group table, so it is off by default.

// Hand a finished batch over to the caller: sorted order -> caller order, pinned landing buffers -> the caller's
// (pageable) output arrays named in the options of the submit / blocking call.
// -> 0, or P2P_ERR_RANGE when a generator pass of the batch stored an activation beyond the split-f16 operand range.  The flag is per batch,
// not per object: every split-f16 object of the batch that was created with P2P_PREC_AUTO switches to its fp32 twin (*switched = how many did);
// the blocking caller then runs the batch once more -- if the offender was one of them the second run is clean, otherwise it reports again
// a pass merged over two batches: its range word m goes to both batches' words (bit patterns of non-negative floats order like unsigned integers)
__global__ void range_fold_kernel(const unsigned* m, unsigned* a, unsigned* b)
{
    const unsigned v = *m;
    if (v > *a) *a = v;
    if (v > *b) *b = v;
}

static int range_verdict(const Slot& s, int* switched)
{
    *switched = 0;
    float amax = 0.f;
    if (s.h_range.p) memcpy(&amax, s.h_range.p, sizeof(amax));
    if (!(amax > 0.f)) return P2P_OK;
    int n_f16 = 0;
    for (const BatchGroup& g : s.groups) {
        const Model* m = reinterpret_cast<const Model*>(s.objs[g.obj].model);
        if (m->effective()->prec != PREC_F16X3) continue;
        ++n_f16;
        if (m->twin) { m->use_twin = true; ++*switched; }
    }
    set_error("est_pose: an activation of magnitude %g exceeds the split-f16 operand range (%g); %d of the batch's %d split-f16 objects switched to their "
              "fp32 twin (P2P_PREC_AUTO)%s", (double)amax, (double)RANGE_LIMIT, *switched, n_f16,
              *switched < n_f16 ? "; objects without one need P2P_PREC_F32 or P2P_PREC_AUTO at creation" : "");
    return P2P_ERR_RANGE;
}

static void finish_batch(const Slot& s, p2p_pose* poses)
{
    const p2p_est_pose_opts& opt = s.opt;
    const unsigned long long* hstat = s.h_stat.as<unsigned long long>();
    for (int i = 0; i < s.n; ++i) {
        const int o = s.perm[i];
        poses[o] = s.host_poses[i];
        if (opt.valid_mask) {       // valid_mask_full = zeros((H, W)); [v1:v2, u1:u2] = valid_mask   (recognition.py:175-176)
            unsigned char* dst = opt.valid_mask + (size_t)o * opt.mask_stride;
            if (!opt.mask_prezeroed) memset(dst, 0, (size_t)opt.mask_stride);
            const p2p_pose& P = s.host_poses[i];
            if (P.status == P2P_POSE_OK) {
                const int v1 = P.bbox_t[0], v2 = P.bbox_t[1], u1 = P.bbox_t[2], u2 = P.bbox_t[3], w = u2 - u1, W = s.img_w[i];
                const unsigned char* src = s.h_mask.as<unsigned char>() + (size_t)i * s.cmask_stride;
                for (int r = 0; r < v2 - v1 && (long long)(r + 1) * w <= s.cmask_stride; ++r) memcpy(dst + (size_t)(v1 + r) * W + u1, src + (size_t)r * w, (size_t)w);
            }
        }
        if (opt.img_pred) {
            unsigned char* dst = opt.img_pred + (size_t)o * opt.pred_stride;
            memcpy(dst, s.h_pred.as<unsigned char>() + (size_t)i * s.cpred_stride, (size_t)s.cpred_stride);
            if (opt.pred_stride > s.cpred_stride) memset(dst + s.cpred_stride, 0, (size_t)(opt.pred_stride - s.cpred_stride));
        }
        if (opt.det_mask && opt.mask_stats) {
            const long long inter = (long long)hstat[3 * i], dc = (long long)hstat[3 * i + 1], vc = (long long)hstat[3 * i + 2];
            opt.mask_stats[3 * o] = inter; opt.mask_stats[3 * o + 1] = dc + vc - inter; opt.mask_stats[3 * o + 2] = vc;
            poses[o].mask_stats[0] = inter; poses[o].mask_stats[1] = dc + vc - inter; poses[o].mask_stats[2] = vc;      // the record carries them too
        }
    }
}

static int forward_parts(Ctx& X, hipStream_t st, const Slot* const* parts, const int* per_det, int n_parts, const float* xin, float* yout,
                         int range_word = -1)
{
    // (model, count) runs of the concatenated, object-sorted parts
    std::vector<const Model*> ms;
    std::vector<int> cnt;
    bool same_backbone = true;
    for (int k = 0; k < n_parts; ++k)
        for (const BatchGroup& g : parts[k]->groups) {
            const Model* m = reinterpret_cast<const Model*>(parts[k]->objs[g.obj].model)->effective();
            const int c = (g.end - g.begin) * per_det[k];
            if (!ms.empty() && ms.back() == m) { cnt.back() += c; continue; }      // the same network on both sides of a part boundary: one run
            ms.push_back(m);
            cnt.push_back(c);
            same_backbone = same_backbone && m->backbone == ms[0]->backbone;
        }
    X.cur = &X.lane[0];
    // operand-range events of this pass are reported with parts[0]'s batch -- or, for a pass MERGED over two batches, to a word of its own
    // that is folded into both batches' words afterwards (an event of the merged pass cannot be attributed to one of them)
    X.range_cur = X.range_words + (range_word >= 0 ? range_word : parts[0]->range_word);
    bool same_prec = true;
    for (const Model* m : ms) same_prec = same_prec && m->prec == ms[0]->prec;
    if (ms.size() == 1) return forward_async(X, *ms[0], xin, cnt[0], yout);
    if (same_backbone && same_prec) return forward_grouped(X, ms, cnt, xin, yout);      // one grouped pass: every M-tile uses its object's weights
    // mixed backbones (or precisions: an object that fell back to its fp32 twin): per-object passes round-robin over the context's lanes (stream + private activation workspace) so that
    // the small per-object launch sequences overlap; fork/join with events around them
    const int nl = std::min<int>((int)ms.size(), Ctx::N_LANES);
    for (int l = 1; l < nl; ++l) { int r = X.ensure_lane(l); if (r) return r; }
    HIP_TRY(hipEventRecord(X.fork, st));
    for (int l = 1; l < nl; ++l) HIP_TRY(hipStreamWaitEvent(X.lane[l].stream, X.fork, 0));
    int r = P2P_OK;
    size_t off = 0;
    for (size_t gi = 0; gi < ms.size() && !r; ++gi) {
        X.cur = &X.lane[gi % nl];
        r = forward_async(X, *ms[gi], xin + off * 16384 * 3, cnt[gi], yout + off * 16384 * 4);
        off += cnt[gi];
    }
    X.cur = &X.lane[0];
    if (r) return r;
    for (int l = 1; l < nl; ++l) {
        HIP_TRY(hipEventRecord(X.lane[l].done, X.lane[l].stream));
        HIP_TRY(hipStreamWaitEvent(st, X.lane[l].done, 0));
    }
    return P2P_OK;
}

// Test / bench hook: the caller's decoder maps (in the caller's detection order) replace the generator output.  One gather launch
// (a memcpy per detection cost 5 us each on the stream: 2.5 ms per 256-detection step whenever the batch mixes objects).
__global__ __launch_bounds__(256) void inject_gather_kernel(const DetInfo* __restrict__ dets, const float4* __restrict__ src,
                                                            float4* __restrict__ dst, unsigned per_det4)
{
    const int i = blockIdx.y;
    const float4* s = src + (size_t)dets[i].src_index * per_det4;
    float4* d = dst + (size_t)i * per_det4;
    for (unsigned k = blockIdx.x * 256 + threadIdx.x; k < per_det4; k += gridDim.x * 256) d[k] = s[k];
}

static int inject_maps(const Slot& SL, hipStream_t st, const float* src, float* dst, size_t per_det)
{
    if (SL.identity) {
        HIP_TRY(hipMemcpyAsync(dst, src, per_det * SL.n * sizeof(float), hipMemcpyDeviceToDevice, st));
    } else if (SL.n > 0) {
        hipLaunchKernelGGL(inject_gather_kernel, dim3(16, SL.n), dim3(256), 0, st, SL.det.as<DetInfo>(), reinterpret_cast<const float4*>(src),
                           reinterpret_cast<float4*>(dst), (unsigned)(per_det / 4));
        HIP_TRY(hipGetLastError());
    }
    return P2P_OK;
}

// front: validation, frames, per-detection constants, buffers, stage-1 network inputs into x1_dst (null: the slot's own x1)
static int enqueue_front(Ctx& X, Pipeline& P, Slot& SL, hipStream_t st, const p2p_object* objects, int n_obj, const p2p_image* images, int n_img,
                         const p2p_detection* dets, int n, const p2p_est_pose_opts& opt, float* x1_dst)
{
    int rc;
    // -- validate, order detections by object (weights locality; one generator pass per group)
    int K = 0;
    for (int o = 0; o < n_obj; ++o) {
        if (!objects[o].model || objects[o].n_outlier_th < 1 || objects[o].n_outlier_th > MAX_TH) {
            set_error("object %d: model missing or n_outlier_th out of [1,%d]", o, MAX_TH);
            return P2P_ERR_INVALID_ARG;
        }
        K = std::max(K, objects[o].n_outlier_th);
    }
    if (opt.inject2 && opt.inject_slots != K) { set_error("inject_slots (%d) must equal the largest n_outlier_th (%d)", opt.inject_slots, K); return P2P_ERR_INVALID_ARG; }
    std::vector<int>& perm = SL.perm;
    perm.resize(n);
    std::iota(perm.begin(), perm.end(), 0);
    for (int i = 0; i < n; ++i) {
        if (dets[i].object < 0 || dets[i].object >= n_obj || dets[i].image < 0 || dets[i].image >= n_img) {
            set_error("detection %d references object %d / image %d out of range", i, dets[i].object, dets[i].image);
            return P2P_ERR_INVALID_ARG;
        }
    }
    std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return dets[a].object < dets[b].object; });
    SL.identity = true;
    for (int i = 0; i < n; ++i) SL.identity = SL.identity && perm[i] == i;
    SL.n = n;
    SL.K = K;
    SL.opt = opt;
    SL.objs.assign(objects, objects + n_obj);
    SL.use_aa = opt.resize_anti_aliasing != 0;
    const int generation = opt.resize_anti_aliasing;      // 0: <= 0.14, 1: 0.17 / 0.18, 2: 0.15 / 0.16 (validated in run_est_pose)

    for (int i = 0; i < n_img; ++i)
        if (!images[i].data || images[i].height <= 0 || images[i].width <= 0) { set_error("image %d is empty", i); return P2P_ERR_INVALID_ARG; }

    // -- per-detection constants + stage-1 geometry (recognition.py:71-79)
    std::vector<DetInfo>& hd = SL.hd;
    hd.resize(n);
    long long corr_total = 0, cv_total = 0;
    int max_side = 0;
    const bool use_aa = SL.use_aa;
    for (int i = 0; i < n; ++i) {
        const p2p_detection& dt = dets[perm[i]];
        const p2p_object& ob = objects[dt.object];
        const p2p_image& im = images[dt.image];
        DetInfo& D = hd[i];
        memset(&D, 0, sizeof(D));
        D.img = nullptr;                    // set below, once the row ranges of the frames are known
        D.H = im.height; D.W = im.width; D.img_f32 = im.dtype == P2P_IMG_F32;
        D.obj = dt.object;
        D.src_index = perm[i];
        D.n_th = ob.n_outlier_th;
        for (int k = 0; k < D.n_th; ++k) D.th_o[k] = (float)ob.outlier_th[k];
        D.th_i = ob.inlier_th;
        D.box_size = ob.box_size > 0 ? ob.box_size : 1.5;
        D.cx_o = (dt.bbox[3] + dt.bbox[1]) / 2.0;
        D.cy_o = (dt.bbox[2] + dt.bbox[0]) / 2.0;
        for (int k = 0; k < 9; ++k) D.K[k] = dt.camK[k];
        for (int k = 0; k < 3; ++k) { D.obj_scale[k] = ob.obj_scale[k]; D.obj_ct[k] = ob.obj_ct[k]; }
        const double bb[4] = {(double)dt.bbox[0], (double)dt.bbox[1], (double)dt.bbox[2], (double)dt.bbox[3]};
        D.b1 = get_boxes(bb, D.H, D.W, D.box_size, false, 0, 0, 9999);
        D.ok1 = boxes_ok(D.b1, D.H, D.W) ? 1 : 0;
        const long long side = std::max(D.b1.v2_ori - D.b1.v1_ori, 0);
        D.corr_cap = D.ok1 ? (int)(side * side) : 0;
        D.corr_off = corr_total;
        corr_total += (long long)D.corr_cap * 5 * K;
        D.aa = generation;
        D.cv_off = cv_total;
        if (use_aa && D.ok1 && side > 128) {
            if (side > 4096) { set_error("detection %d: crop side %lld exceeds the anti-aliasing table (4096)", perm[i], side); return P2P_ERR_CAPACITY; }
            cv_total += (long long)(1 + K) * D.corr_cap * 3;
        }
        if (D.ok1) max_side = std::max(max_side, (int)side);
    }
    // -- frames.  Host frames (the reference's boundary: est_pose takes a numpy frame, recognition.py:70): only the rows some detection's
    // stage-1 crop covers are ever read (the stage-2 canvas is zero outside the stage-1 crop, recognition.py:105-106), so only those rows
    // of the caller's pageable memory are copied into the slot's pinned staging buffer (a plain memcpy: the frames are free again when
    // submit returns) -- packed, one row range per image -- and go up as ONE DMA on its own stream, under the generator passes already
    // queued on `st`; `st` waits for it before the first kernel.  (hipMemcpyAsync straight from pageable memory cost 1.6 ms per 30 MB
    // step even on a separate stream; whole frames instead of row ranges made one 128-px detection move 0.9 MB instead of 0.25.)
    {
        std::vector<int> r0(n_img, 1 << 30), r1(n_img, 0);
        for (int i = 0; i < n; ++i) {
            if (!hd[i].ok1) continue;
            const int im = dets[perm[i]].image;
            r0[im] = std::min(r0[im], hd[i].b1.v1);
            r1[im] = std::max(r1[im], hd[i].b1.v2);
        }
        std::vector<const char*> img_dev(n_img, nullptr);
        std::vector<size_t> off(n_img, 0);
        size_t need = 0;
        for (int i = 0; i < n_img; ++i) {
            if (images[i].mem != P2P_MEM_HOST) { img_dev[i] = reinterpret_cast<const char*>(images[i].data); continue; }
            if (r1[i] <= r0[i]) { r0[i] = r1[i] = 0; }
            off[i] = need;
            need += ((size_t)(r1[i] - r0[i]) * images[i].width * 3 * (images[i].dtype ? 4 : 1) + 255) / 256 * 256;
        }
        if ((rc = SL.images.reserve(need)) || (rc = SL.h_frames.reserve(need))) return rc;
        for (int i = 0; i < n_img; ++i) {
            if (images[i].mem != P2P_MEM_HOST) continue;
            const size_t row = (size_t)images[i].width * 3 * (images[i].dtype ? 4 : 1);
            memcpy(SL.h_frames.as<char>() + off[i], reinterpret_cast<const char*>(images[i].data) + (size_t)r0[i] * row, (size_t)(r1[i] - r0[i]) * row);
            img_dev[i] = SL.images.as<char>() + off[i] - (size_t)r0[i] * row;      // address of the frame's row 0 (rows outside [r0, r1) are never read)
        }
        for (int i = 0; i < n; ++i) hd[i].img = img_dev[dets[perm[i]].image];
        if (need) {
            HIP_TRY(hipMemcpyAsync(SL.images.p, SL.h_frames.p, need, hipMemcpyHostToDevice, P.copy_stream));
            HIP_TRY(hipEventRecord(P.frames_ready, P.copy_stream));
            HIP_TRY(hipStreamWaitEvent(st, P.frames_ready, 0));
        }
    }
    SL.max_side = max_side;
    // operand-range word of this batch's generator passes: cleared in stream order before pass 1, landed with the poses
    SL.range_word = 1 + (int)(&SL - P.slot);
    SL.range_dev = X.range_words + SL.range_word;
    if ((rc = SL.h_range.reserve(sizeof(unsigned)))) return rc;
    memset(SL.h_range.p, 0, sizeof(unsigned));
    HIP_TRY(hipMemsetAsync(SL.range_dev, 0, sizeof(unsigned), st));
    SL.img_hw.resize(n);
    SL.img_w.resize(n);
    for (int i = 0; i < n; ++i) { SL.img_hw[i] = hd[i].H * hd[i].W; SL.img_w[i] = hd[i].W; }
    // object groups (contiguous in the sorted order)
    SL.groups.clear();
    SL.same_backbone = true;
    for (int i = 0; i < n;) {
        int j = i;
        while (j < n && hd[j].obj == hd[i].obj) ++j;
        SL.groups.push_back({hd[i].obj, i, j});
        SL.same_backbone = SL.same_backbone && reinterpret_cast<const Model*>(objects[hd[i].obj].model)->backbone ==
                                                    reinterpret_cast<const Model*>(objects[hd[0].obj].model)->backbone;
        i = j;
    }
    // the stage-2 buffers leave room for the stage-1 inputs / outputs of a following batch of up to the same size (merged pass)
    SL.tail_cap = n;
    if ((rc = SL.det.reserve(sizeof(DetInfo) * n))) return rc;
    if ((rc = SL.s1.reserve(sizeof(Stage1) * n))) return rc;
    if ((rc = SL.cand.reserve(sizeof(CandStat) * n * K))) return rc;
    if ((rc = SL.probs.reserve(sizeof(PnpProblem) * n * K))) return rc;
    if ((rc = SL.results.reserve(sizeof(PnpResult) * n * K))) return rc;
    if ((rc = SL.hyp.reserve(pnp_workspace_bytes(n * K)))) return rc;
    if ((rc = SL.poses.reserve(sizeof(p2p_pose) * n))) return rc;
    if ((rc = SL.x1.reserve(sizeof(float) * 16384 * 3 * (size_t)n))) return rc;
    if ((rc = SL.y1.reserve(sizeof(float) * 16384 * 4 * (size_t)n))) return rc;
    if ((rc = SL.x2.reserve(sizeof(float) * 16384 * 3 * ((size_t)n * K + SL.tail_cap)))) return rc;
    if ((rc = SL.y2.reserve(sizeof(float) * 16384 * 4 * ((size_t)n * K + SL.tail_cap)))) return rc;
    if ((rc = SL.corr.reserve(sizeof(float) * (size_t)std::max<long long>(corr_total, 1)))) return rc;
    if ((rc = SL.crange.reserve(sizeof(CandRange) * (size_t)n * K))) return rc;
    // two-launch correspondence build (cand_eval_kernel): a handful of candidates, or crops large enough that one workgroup per candidate
    // leaves the launch waiting for its largest member
    const bool corr_seg = n * K <= 16 || max_side > 192;
    if (corr_seg && ((rc = SL.crec.reserve(sizeof(unsigned) * (size_t)std::max<long long>(corr_total / 5, 1))) ||
                        (rc = SL.cseg.reserve(sizeof(CorrSeg) * CORR_SEG * (size_t)n * K)))) return rc;
    SL.aa = {nullptr, nullptr, nullptr, nullptr};
    AaBufs aab = {nullptr, nullptr, nullptr, nullptr};
    AaTable aat;
    if (use_aa) {
        if ((rc = aa_table_get(X.device, &aat))) return rc;
        const size_t cvb = sizeof(double) * (size_t)std::max<long long>(cv_total, 1), plane = sizeof(double) * 16384 * (size_t)n * K;
        if ((rc = SL.aa_items.reserve(sizeof(AaItem) * (size_t)(n + 6 * n * K))) || (rc = SL.aa_cv.reserve(cvb)) || (rc = SL.aa_cv_tmp.reserve(cvb)) ||
            (rc = SL.aa_bk.reserve(plane * 5)) || (rc = SL.aa_bk_tmp.reserve(plane * 5))) return rc;
        AaItem* it = SL.aa_items.as<AaItem>();
        SL.aa = {it, it + n, it + n + n * K, nullptr};
        aab = {SL.aa_cv.as<double>(), SL.aa_cv_tmp.as<double>(), SL.aa_bk.as<double>(), SL.aa_bk_tmp.as<double>()};
        if (generation == 2) {      // filtered bool keep masks, one 128 x 128 byte plane per candidate (keep_filter_kernel)
            if ((rc = SL.keepf.reserve((size_t)16384 * n * K))) return rc;
            SL.aa.keepf = SL.keepf.as<unsigned char>();
        }
    }
    // -- optional outputs: argument checks, landing buffers and the detector-mask upload (read during submit)
    const bool want_mask = opt.valid_mask != nullptr, want_pred = opt.img_pred != nullptr, want_iou = opt.det_mask != nullptr;
    for (int i = 0; i < n; ++i) {
        const long long hw = (long long)hd[i].H * hd[i].W;
        if (want_mask && hw > opt.mask_stride) { set_error("mask_stride too small for detection %d", perm[i]); return P2P_ERR_CAPACITY; }
        if (want_iou && hw > opt.det_mask_stride) { set_error("det_mask_stride too small for detection %d", perm[i]); return P2P_ERR_CAPACITY; }
    }
    // compact strides: a candidate's clipped box never exceeds the largest stage-1 square of the batch
    SL.cmask_stride = std::max(1LL, (long long)max_side * max_side);
    SL.cpred_stride = std::max(1LL, std::min<long long>(opt.pred_stride, SL.cmask_stride * 3));
    if (want_mask && ((rc = SL.mask.reserve((size_t)SL.cmask_stride * n)) || (rc = SL.h_mask.reserve((size_t)SL.cmask_stride * n)))) return rc;
    if (want_pred && ((rc = SL.pred.reserve((size_t)SL.cpred_stride * n)) || (rc = SL.h_pred.reserve((size_t)SL.cpred_stride * n)))) return rc;
    if (want_iou) {
        if ((rc = SL.dmask.reserve((size_t)opt.det_mask_stride * n)) || (rc = SL.mstat.reserve(sizeof(unsigned long long) * 3 * n)) ||
            (rc = SL.h_stat.reserve(sizeof(unsigned long long) * 3 * n))) return rc;
        for (int i = 0; i < n; ++i)      // caller order -> sorted order
            HIP_TRY(hipMemcpyAsync(SL.dmask.as<unsigned char>() + (size_t)i * opt.det_mask_stride,
                                   opt.det_mask + (size_t)perm[i] * opt.det_mask_stride, (size_t)hd[i].H * hd[i].W,
                                   hipMemcpyHostToDevice, st));
    }
    if (SL.host_cap < (size_t)n) {
        if (SL.host_poses) (void)hipHostFree(SL.host_poses);
        SL.host_poses = nullptr;
        HIP_TRY(hipHostMalloc((void**)&SL.host_poses, sizeof(p2p_pose) * (size_t)n * 2, hipHostMallocDefault));
        SL.host_cap = (size_t)n * 2;
    }
    HIP_TRY(hipMemcpyAsync(SL.det.p, hd.data(), sizeof(DetInfo) * n, hipMemcpyHostToDevice, st));

    // -- stage-1 network inputs
    const DetInfo* d_det = SL.det.as<DetInfo>();
    if (use_aa) {      // anti-aliased stage-1 canvases (sides > 128): build, filter in place
        hipLaunchKernelGGL(aa_plan_kernel, dim3((n * K + 63) / 64), dim3(64), 0, st, d_det, SL.s1.as<Stage1>(), n, K, 0, aat, aab, SL.aa);
        hipLaunchKernelGGL(aa_canvas1_kernel, dim3(64, n), dim3(256), 0, st, d_det, SL.aa);
        HIP_TRY(hipGetLastError());
        HIP_TRY(launch_aa_filter(SL.aa.k0, n, max_side * max_side * 3, st));
    }
    hipLaunchKernelGGL(stage1_input_kernel, dim3(n * 64), dim3(256), 0, st, d_det, x1_dst ? x1_dst : SL.x1.as<float>(), SL.aa);
    HIP_TRY(hipGetLastError());
    return P2P_OK;
}

// mid: stage-1 network output y1 -> reductions, stage-2 geometry, stage-2 network inputs (into the slot's x2)
static int enqueue_mid(Ctx& X, Slot& SL, hipStream_t st, float* y1)
{
    int rc;
    const int n = SL.n, K = SL.K;
    const DetInfo* d_det = SL.det.as<DetInfo>();
    Stage1* d_s1 = SL.s1.as<Stage1>();
    if (SL.opt.inject1 && (rc = inject_maps(SL, st, SL.opt.inject1, y1, 16384 * 4))) return rc;
    if (n <= 16) {            // a handful of detections: STATS_SEG workgroups each (see stage1_stats_seg_kernel)
        if (SL.sacc_n < n) {  // accumulators: initialised when (re)allocated, self-clearing afterwards
            if ((rc = SL.sacc.reserve(sizeof(StatsAcc) * 16))) return rc;
            hipLaunchKernelGGL(stats_acc_init_kernel, dim3(1), dim3(64), 0, st, SL.sacc.as<StatsAcc>(), 16);
            SL.sacc_n = 16;
        }
        hipLaunchKernelGGL(stage1_stats_seg_kernel, dim3(STATS_SEG, n), dim3(256), 0, st, d_det, y1, d_s1, SL.sacc.as<StatsAcc>());
    } else
        hipLaunchKernelGGL(stage1_stats_kernel, dim3(n), dim3(n <= 64 ? 1024 : 256), 0, st, d_det, y1, d_s1);
    HIP_TRY(hipGetLastError());
    if (SL.use_aa) {      // anti-aliased stage-2 canvases (sides > 128; the bool keep masks of sides < 128 are not filtered, see aa_plan_kernel)
        AaTable aat;
        if ((rc = aa_table_get(X.device, &aat))) return rc;
        AaBufs aab = {SL.aa_cv.as<double>(), SL.aa_cv_tmp.as<double>(), SL.aa_bk.as<double>(), SL.aa_bk_tmp.as<double>()};
        hipLaunchKernelGGL(aa_plan_kernel, dim3((n * K + 63) / 64), dim3(64), 0, st, d_det, d_s1, n, K, 1, aat, aab, SL.aa);
        hipLaunchKernelGGL(aa_canvas2_kernel, dim3(64, n * K), dim3(256), 0, st, d_det, d_s1, y1, K, SL.aa);
        HIP_TRY(hipGetLastError());
        HIP_TRY(launch_aa_filter(SL.aa.k1, n * K, SL.max_side * SL.max_side * 3, st));
    }
    if (SL.aa.keepf) {    // generation 2: the bool keep masks of stage-1 sides < 128 are filtered too (into bool planes)
        AaTable aat;
        if ((rc = aa_table_get(X.device, &aat))) return rc;
        hipLaunchKernelGGL(keep_filter_kernel, dim3(n * K), dim3(256), 0, st, d_det, d_s1, y1, K, aat, const_cast<unsigned char*>(SL.aa.keepf));
        HIP_TRY(hipGetLastError());
    }
    {
        ProfScope ps(19, st);
        hipLaunchKernelGGL(stage2_input_kernel, dim3(n * K * 64), dim3(256), 0, st, d_det, d_s1, y1, K, SL.x2.as<float>(), SL.aa);
    }
    HIP_TRY(hipGetLastError());
    return P2P_OK;
}

// tail: stage-2 network output (the slot's y2) -> correspondences, PnP-RANSAC, selection, optional outputs, D2H.
// async: PnP and everything after it go to the tail stream and the slot's `done` event is recorded there.
static int enqueue_tail(Pipeline& P, Slot& SL, hipStream_t st, bool async)
{
    int rc;
    const int n = SL.n, K = SL.K;
    const p2p_est_pose_opts& opt = SL.opt;
    const DetInfo* d_det = SL.det.as<DetInfo>();
    Stage1* d_s1 = SL.s1.as<Stage1>();
    float* y2 = SL.y2.as<float>();
    const AaPtrs aa = SL.aa;
    if (opt.inject2 && (rc = inject_maps(SL, st, opt.inject2, y2, (size_t)K * 16384 * 4))) return rc;

    // -- ranges of the back-resize inputs (clip=True), anti-aliased maps where the stage-2 side is < 128
    CandRange* d_cr = SL.crange.as<CandRange>();
    const int glue_nt = n * K <= 64 ? 1024 : 256;      // per-candidate workgroups: wide when there are few of them
    hipLaunchKernelGGL(cand_range_kernel, dim3(n * K), dim3(glue_nt), 0, st, y2, d_cr);
    HIP_TRY(hipGetLastError());
    if (SL.use_aa) {
        hipLaunchKernelGGL(aa_back_fill_kernel, dim3(64, n * K), dim3(256), 0, st, y2, aa);
        HIP_TRY(hipGetLastError());
        HIP_TRY(launch_aa_filter(aa.k3, 5 * n * K, 16384, st));
        hipLaunchKernelGGL(aa_back_range_kernel, dim3((n * K + 63) / 64), dim3(64), 0, st, d_cr, n * K, aa);
        HIP_TRY(hipGetLastError());
    }

    // -- correspondences, PnP-RANSAC, selection
    static const bool corr_split = dev_env("P2P_CORR_SPLIT") == nullptr || atoi(dev_env("P2P_CORR_SPLIT")) != 0;      // development switch (A/B)
    {
    ProfScope ps_corr(18, st);
    if ((n * K <= 16 || SL.max_side > 192) && corr_split) {        // evaluate on CORR_SEG CUs per candidate, then compact (see cand_eval_kernel)
        hipLaunchKernelGGL(cand_eval_kernel, dim3(CORR_SEG, n * K), dim3(256), 0, st, d_det, d_s1, y2, K, SL.crec.as<unsigned>(), SL.cseg.as<CorrSeg>(), d_cr, aa);
        hipLaunchKernelGGL(cand_compact_kernel, dim3(CORR_SEG, n * K), dim3(256), 0, st, d_det, d_s1, K, SL.crec.as<unsigned>(), SL.cseg.as<CorrSeg>(),
                           SL.corr.as<float>(), SL.cand.as<CandStat>(), SL.probs.as<PnpProblem>());
    } else
        hipLaunchKernelGGL(cand_corr_kernel, dim3(n * K), dim3(glue_nt), 0, st, d_det, d_s1, y2, K, SL.corr.as<float>(),
                           SL.cand.as<CandStat>(), SL.probs.as<PnpProblem>(), d_cr, aa);
    }
    HIP_TRY(hipGetLastError());
    const int iters = opt.ransac_iterations > 0 ? opt.ransac_iterations : 100;
    const double rerr = opt.reprojection_error > 0 ? opt.reprojection_error : 5.0;
    const double conf = opt.confidence > 0 ? opt.confidence : 0.99;
    hipStream_t ts = st;
    if (async) {      // the latency-bound PnP tail runs beside the next generator pass
        ts = P.tail_stream;
        HIP_TRY(hipEventRecord(P.corr_ready, st));
        HIP_TRY(hipStreamWaitEvent(ts, P.corr_ready, 0));
    }
#ifdef P2P_TIMING_SWITCHES      // tools/ab_build.sh pipeline.hip -DP2P_TIMING_SWITCHES: never in the shipped library (poses are garbage with it)
    static const bool skip_pnp = getenv("P2P_SKIP_PNP") != nullptr;      // timing experiment: what the PnP tail costs a stream of batches
#else
    constexpr bool skip_pnp = false;
#endif
    if (!skip_pnp)
    HIP_TRY(launch_pnp_ransac(SL.probs.as<PnpProblem>(), SL.results.as<PnpResult>(), n * K, iters, rerr, conf, 6, std::max(1, SL.max_side * SL.max_side),
                              SL.hyp.as<double>(), ts));
    hipLaunchKernelGGL(select_kernel, dim3((n + 63) / 64), dim3(64), 0, ts, d_det, d_s1, SL.cand.as<CandStat>(),
                       SL.results.as<PnpResult>(), K, n, SL.poses.as<p2p_pose>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(SL.host_poses, SL.poses.p, sizeof(p2p_pose) * n, hipMemcpyDeviceToHost, ts));
    HIP_TRY(hipMemcpyAsync(SL.h_range.p, SL.range_dev, sizeof(unsigned), hipMemcpyDeviceToHost, ts));      // both generator passes are behind us on this stream

    // -- optional outputs of the reference's return tuple (recognition.py:189-193: img_pred_f, valid_mask_full) and the
    //    score_type-2 mask sums: rendered on the tail stream, landed in the slot's pinned buffers, handed over by finish_batch()
    const bool want_mask = opt.valid_mask != nullptr, want_pred = opt.img_pred != nullptr, want_iou = opt.det_mask != nullptr;
    const long long cms = SL.cmask_stride, cps = SL.cpred_stride;
    if (want_mask || want_pred) {
        if (want_mask) HIP_TRY(hipMemsetAsync(SL.mask.p, 0, (size_t)cms * n, ts));
        if (want_pred) HIP_TRY(hipMemsetAsync(SL.pred.p, 0, (size_t)cps * n, ts));
        hipLaunchKernelGGL(render_best_kernel, dim3(64, n), dim3(256), 0, ts, d_det, d_s1, SL.poses.as<p2p_pose>(), y2, K,
                           want_mask ? SL.mask.as<unsigned char>() : nullptr, cms,
                           want_pred ? SL.pred.as<unsigned char>() : nullptr, cps, d_cr, aa);
        HIP_TRY(hipGetLastError());
        if (want_mask) HIP_TRY(hipMemcpyAsync(SL.h_mask.p, SL.mask.p, (size_t)cms * n, hipMemcpyDeviceToHost, ts));
        if (want_pred) HIP_TRY(hipMemcpyAsync(SL.h_pred.p, SL.pred.p, (size_t)cps * n, hipMemcpyDeviceToHost, ts));
    }
    if (want_iou) {
        HIP_TRY(hipMemsetAsync(SL.mstat.p, 0, sizeof(unsigned long long) * 3 * n, ts));
        hipLaunchKernelGGL(mask_iou_kernel, dim3(32, n), dim3(256), 0, ts, d_det, d_s1, SL.poses.as<p2p_pose>(), y2, K,
                           SL.dmask.as<unsigned char>(), (long long)opt.det_mask_stride, SL.mstat.as<unsigned long long>(), d_cr, aa);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(SL.h_stat.p, SL.mstat.p, sizeof(unsigned long long) * 3 * n, hipMemcpyDeviceToHost, ts));
    }
    if (async) HIP_TRY(hipEventRecord(SL.done, ts));
    SL.stage2_pending = false;
    return P2P_OK;
}

// pass 2 + tail of a batch whose stage-2 inputs are waiting (no later batch to merge with)
static int flush_stage2(Ctx& X, Pipeline& P, Slot& SL, hipStream_t st, bool async)
{
    int rc;
    const Slot* parts[1] = {&SL};
    const int per[1] = {SL.K};
    if ((rc = forward_parts(X, st, parts, per, 1, SL.x2.as<float>(), SL.y2.as<float>()))) return rc;
    return enqueue_tail(P, SL, st, async);
}

// async_ticket == nullptr: blocking call, results in `poses`.  Otherwise the batch is only enqueued and
// p2p_est_pose_collect() picks the results up.
static int run_est_pose(Ctx& X, const p2p_object* objects, int n_obj, const p2p_image* images, int n_img,
                        const p2p_detection* dets, int n, p2p_pose* poses, const p2p_est_pose_opts& opt, int* async_ticket)
{
    int rc;
    if (!X.pipe) X.pipe = new Pipeline();
    Pipeline& P = *X.pipe;
    if ((rc = X.ensure_workspace())) return rc;
    const bool async = async_ticket != nullptr;
    hipStream_t st = X.lane[0].stream;           // the chain pass 1 -> mid -> pass 2 of consecutive batches is serial: one stream
    if (!P.tail_stream) {
        HIP_TRY(hipStreamCreate(&P.tail_stream));
        HIP_TRY(hipStreamCreate(&P.copy_stream));
        HIP_TRY(hipEventCreateWithFlags(&P.corr_ready, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&P.frames_ready, hipEventDisableTiming));
        for (Slot& s : P.slot) HIP_TRY(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
    int slot_idx = 0;
    if (async) {          // any free slot (tickets may be collected out of order)
        slot_idx = P.next_ticket % Pipeline::N_SLOTS;
        for (int k = 0; k < Pipeline::N_SLOTS; ++k)
            if (P.slot[(slot_idx + k) % Pipeline::N_SLOTS].ticket < 0) { slot_idx = (slot_idx + k) % Pipeline::N_SLOTS; break; }
    }
    Slot& SL = P.slot[slot_idx];
    if (SL.ticket >= 0) {
        if (async) { set_error("%d batches are already in flight: collect ticket %d first", Pipeline::N_SLOTS, SL.ticket); return P2P_ERR_CAPACITY; }
        set_error("an asynchronous batch (ticket %d) is in flight: collect it before a blocking call", SL.ticket);
        return P2P_ERR_CAPACITY;
    }
    if (async && (opt.dbg_x1 || opt.dbg_x2 || opt.dbg_boxes2 || opt.dbg_cand || opt.dbg_y1 || opt.dbg_y2)) {
        set_error("the debug taps are only available from the blocking call");
        return P2P_ERR_INVALID_ARG;
    }
    if (opt.ransac_iterations > P2P_MAX_RANSAC_ITERATIONS) {
        set_error("ransac_iterations %d exceeds P2P_MAX_RANSAC_ITERATIONS (%d)", opt.ransac_iterations, P2P_MAX_RANSAC_ITERATIONS);
        return P2P_ERR_INVALID_ARG;
    }
    if (opt.resize_anti_aliasing < 0 || opt.resize_anti_aliasing > 2) {
        set_error("resize_anti_aliasing = %d: the scikit-image generation is 0 (<= 0.14), 1 (0.17 / 0.18) or 2 (0.15 / 0.16)", opt.resize_anti_aliasing);
        return P2P_ERR_INVALID_ARG;
    }
    if ((opt.det_mask != nullptr) != (opt.mask_stats != nullptr)) {
        set_error("det_mask and mask_stats go together");
        return P2P_ERR_INVALID_ARG;
    }

    // the batch whose stage-2 inputs wait for a generator pass (at most one: the previous asynchronous batch)
    Slot* PS = nullptr;
    for (Slot& s : P.slot)
        if (&s != &SL && s.ticket >= 0 && s.stage2_pending) PS = &s;
    // its pass 2 is merged with this batch's pass 1 when the new stage-1 inputs fit behind its stage-2 inputs
    const bool merge = async && PS && PS->opt.merge_stream_passes && n <= PS->tail_cap;
    if (PS && !merge && (rc = flush_stage2(X, P, *PS, st, true))) return rc;
    float* x1 = merge ? PS->x2.as<float>() + (size_t)PS->n * PS->K * 16384 * 3 : nullptr;
    if ((rc = enqueue_front(X, P, SL, st, objects, n_obj, images, n_img, dets, n, opt, x1))) {
        SL.stage2_pending = false;
        if (PS && merge) (void)flush_stage2(X, P, *PS, st, true);      // do not strand the waiting batch behind a rejected one
        return rc;
    }
    const int K = SL.K;
    float* y1 = SL.y1.as<float>();
    if (merge) {
        // mixed backbones inside either batch: no grouped pass over the concatenation -- fall back to two passes
        const Model* m0 = reinterpret_cast<const Model*>(PS->objs[PS->groups[0].obj].model);
        const Model* m1 = reinterpret_cast<const Model*>(SL.objs[SL.groups[0].obj].model);
        if (PS->same_backbone && SL.same_backbone && m0->backbone == m1->backbone) {
            const Slot* parts[2] = {PS, &SL};
            const int per[2] = {PS->K, 1};
            // the merged pass reports to its own range word (3 + slot), folded into BOTH batches' words behind it: PS's stage-1 events stay PS's,
            // the newcomer's word starts from the merged pass only
            const int mw = 3 + (int)(&SL - P.slot);
            HIP_TRY(hipMemsetAsync(X.range_words + mw, 0, sizeof(unsigned), st));
            if ((rc = forward_parts(X, st, parts, per, 2, PS->x2.as<float>(), PS->y2.as<float>(), mw))) return rc;
            y1 = PS->y2.as<float>() + (size_t)PS->n * PS->K * 16384 * 4;
            hipLaunchKernelGGL(range_fold_kernel, dim3(1), dim3(1), 0, st, X.range_words + mw, PS->range_dev, SL.range_dev);
            HIP_TRY(hipGetLastError());
            if ((rc = enqueue_tail(P, *PS, st, true))) return rc;
        } else {
            if ((rc = flush_stage2(X, P, *PS, st, true))) return rc;
            const Slot* parts[1] = {&SL};
            const int per[1] = {1};
            y1 = PS->y2.as<float>() + (size_t)PS->n * PS->K * 16384 * 4;
            if ((rc = forward_parts(X, st, parts, per, 1, x1, y1))) return rc;
        }
    } else {
        const Slot* parts[1] = {&SL};
        const int per[1] = {1};
        if ((rc = forward_parts(X, st, parts, per, 1, SL.x1.as<float>(), y1))) return rc;
    }
    if ((rc = enqueue_mid(X, SL, st, y1))) return rc;
    SL.stage2_pending = true;
    if (async) {
        // pass 2 + tail now -- unless the caller asked for merged passes, in which case they wait for the next submit (or collect)
        if (!opt.merge_stream_passes && (rc = flush_stage2(X, P, SL, st, true))) return rc;
        SL.ticket = P.next_ticket;
        *async_ticket = P.next_ticket++;
        return P2P_OK;
    }

    // -- blocking call: pass 2 and the tail right away
    if ((rc = flush_stage2(X, P, SL, st, false))) return rc;
    // debug taps
    const float *x1h = SL.x1.as<float>(), *x2h = SL.x2.as<float>();
    std::vector<float> hx1, hx2, hy1, hy2;
    std::vector<Stage1> hs1;
    std::vector<CandStat> hcs;
    std::vector<PnpResult> hres;
    if (opt.dbg_x1) { hx1.resize((size_t)n * 16384 * 3); HIP_TRY(hipMemcpyAsync(hx1.data(), x1h, hx1.size() * 4, hipMemcpyDeviceToHost, st)); }
    if (opt.dbg_x2) { hx2.resize((size_t)n * K * 16384 * 3); HIP_TRY(hipMemcpyAsync(hx2.data(), x2h, hx2.size() * 4, hipMemcpyDeviceToHost, st)); }
    if (opt.dbg_y1) { hy1.resize((size_t)n * 16384 * 4); HIP_TRY(hipMemcpyAsync(hy1.data(), SL.y1.as<float>(), hy1.size() * 4, hipMemcpyDeviceToHost, st)); }
    if (opt.dbg_y2) { hy2.resize((size_t)n * K * 16384 * 4); HIP_TRY(hipMemcpyAsync(hy2.data(), SL.y2.as<float>(), hy2.size() * 4, hipMemcpyDeviceToHost, st)); }
    if (opt.dbg_boxes2 || opt.dbg_cand) {
        hs1.resize(n); hcs.resize((size_t)n * K); hres.resize((size_t)n * K);
        HIP_TRY(hipMemcpyAsync(hs1.data(), SL.s1.p, sizeof(Stage1) * n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hcs.data(), SL.cand.p, sizeof(CandStat) * n * K, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(hres.data(), SL.results.p, sizeof(PnpResult) * n * K, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    {
        // the verdict BEFORE anything is handed over: the masks / images / sums of a pass that left the operand range never reach the caller's
        // buffers (with mask_prezeroed the repeated run only rewrites the rows of detections that succeed then)
        int switched = 0;
        if ((rc = range_verdict(SL, &switched))) {
            static thread_local int depth = 0;
            if (!switched || depth > 0) return rc;
            ++depth;                               // objects with an fp32 twin (P2P_PREC_AUTO) now use it: once more (a second event = an object without one)
            rc = run_est_pose(X, objects, n_obj, images, n_img, dets, n, poses, opt, nullptr);
            --depth;
            return rc;
        }
    }
    finish_batch(SL, poses);
    for (int i = 0; i < n; ++i) {
        const int o = SL.perm[i];
        if (opt.dbg_x1) memcpy(opt.dbg_x1 + (size_t)o * 16384 * 3, hx1.data() + (size_t)i * 16384 * 3, 16384 * 3 * 4);
        if (opt.dbg_x2) memcpy(opt.dbg_x2 + (size_t)o * K * 16384 * 3, hx2.data() + (size_t)i * K * 16384 * 3, (size_t)K * 16384 * 3 * 4);
        if (opt.dbg_y1) memcpy(opt.dbg_y1 + (size_t)o * 16384 * 4, hy1.data() + (size_t)i * 16384 * 4, 16384 * 4 * 4);
        if (opt.dbg_y2) memcpy(opt.dbg_y2 + (size_t)o * K * 16384 * 4, hy2.data() + (size_t)i * K * 16384 * 4, (size_t)K * 16384 * 4 * 4);
        if (opt.dbg_boxes2) memcpy(opt.dbg_boxes2 + (size_t)o * 12, &hs1[i].b2, sizeof(Boxes));
        if (opt.dbg_cand)
            for (int k = 0; k < K; ++k) {
                int* c = opt.dbg_cand + ((size_t)o * K + k) * 6;
                c[0] = hs1[i].valid2[k]; c[1] = hcs[(size_t)i * K + k].n_non_gray; c[2] = hcs[(size_t)i * K + k].n_corr;
                c[3] = hres[(size_t)i * K + k].n_inliers;
                c[4] = hres[(size_t)i * K + k].iters; c[5] = hres[(size_t)i * K + k].best_iter;
            }
    }
    return P2P_OK;
}

// ------------------------------------------------------------------------------------------
// Test hook (p2p_debug_back_resize): the back-resizes of recognition.py:134,144,146 on caller-supplied 128 x 128 planes, through
// exactly the code the pipeline runs (cand_pixel, the anti-aliasing filter, the clip ranges).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void back_resize_probe_kernel(const double* __restrict__ planes, CandRange R, int S2, int S2w, double th_i, int gen,
                                                                unsigned char* __restrict__ q, unsigned char* __restrict__ below,
                                                                unsigned char* __restrict__ non_gray)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= S2 * S2w) return;
    const int r = p / S2w, c = p - r * S2w;
    const CandPixel cp = cand_pixel(nullptr, planes, R, r, c, S2, S2w, th_i, gen);
    q[3 * p] = cp.q[0]; q[3 * p + 1] = cp.q[1]; q[3 * p + 2] = cp.q[2];
    below[p] = cp.below ? 1 : 0;
    non_gray[p] = cp.non_gray ? 1 : 0;
}

// comm != nullptr: the records of every rank's batch are all-gathered (device to device, on the tail stream, behind this batch's tail)
// into gathered[world][n_max].  The collective is entered on EVERY path once the arguments name a communicator -- a rank that returned
// early would leave its peers blocked in ncclAllGather for good -- with an all-padding block when this rank has nothing valid to send:
// no batch this step (ticket == P2P_TICKET_NONE: an empty shard), an unknown ticket, a batch larger than n_max, a failed stage-2 flush.
// The operand-range verdict is taken BEFORE the records are packed: a batch that left the split-f16 range travels as P2P_POSE_RANGE
// records, so the peers see that those detections exist and that their poses are not to be used; this rank gets P2P_ERR_RANGE.
static int collect_est_pose(Ctx& X, int ticket, p2p_pose* poses, Comm* comm = nullptr, int n_max = 0, p2p_pose* gathered = nullptr)
{
    Slot* S = nullptr;
    int local = P2P_OK;
    if (!(comm && ticket == P2P_TICKET_NONE)) {
        if (X.pipe)
            for (Slot& s : X.pipe->slot)
                if (s.ticket == ticket && ticket >= 0) S = &s;
        if (!S) { set_error(X.pipe ? "ticket %d is not in flight" : "no batch was submitted (ticket %d)", ticket); local = P2P_ERR_INVALID_ARG; }
    }
    if (local && !comm) return local;
    if (S && comm && S->n > n_max) {
        set_error("p2p_est_pose_collect_gathered: the batch holds %d detections, n_max is %d (the ticket stays in flight)", S->n, n_max);
        local = P2P_ERR_INVALID_ARG;
    }
    if (S && !local && S->stage2_pending) local = flush_stage2(X, *X.pipe, *S, X.lane[0].stream, true);      // no later submit picked the stage-2 pass up
    int verdict = P2P_OK;
    if (S && !local) {
        if (hipEventSynchronize(S->done) != hipSuccess) { set_error("hipEventSynchronize(batch done) failed"); local = P2P_ERR_HIP; }
        else { int switched = 0; verdict = range_verdict(*S, &switched); }
    }
    if (comm) {
        const int rc = comm_gather(X, *comm, (S && !local) ? S : nullptr, verdict != P2P_OK, X.pipe ? X.pipe->tail_stream : X.stream, n_max, gathered);
        if (rc && !local) local = rc;
    }
    if (local) return local;
    if (!S) return P2P_OK;                       // empty shard: joined the collective, nothing of its own to hand over
    // P2P_ERR_RANGE: nothing of the pass is handed over (poses, masks, images stay untouched); with P2P_PREC_AUTO objects a re-submission runs in fp32
    if (verdict == P2P_OK) finish_batch(*S, poses);
    S->ticket = -1;
    return verdict;
}

}  // namespace p2p

using namespace p2p;

extern "C" {

int p2p_est_pose_submit(p2p_ctx* ctx, const p2p_object* objects, int n_objects, const p2p_image* images, int n_images,
                        const p2p_detection* dets, int n_dets, const p2p_est_pose_opts* opts, int* ticket)
{
    if (!ctx || !ticket || n_dets <= 0 || !objects || !images || !dets || n_objects <= 0 || n_images <= 0) {
        set_error("p2p_est_pose_submit: bad arguments");
        return P2P_ERR_INVALID_ARG;
    }
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    ProfHookGuard prof_guard(*c);
    HIP_TRY(hipSetDevice(c->device));
    p2p_est_pose_opts o;
    memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    return run_est_pose(*c, objects, n_objects, images, n_images, dets, n_dets, nullptr, o, ticket);
}

int p2p_est_pose_collect(p2p_ctx* ctx, int ticket, p2p_pose* poses)
{
    if (!ctx || !poses) { set_error("p2p_est_pose_collect: bad arguments"); return P2P_ERR_INVALID_ARG; }
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    ProfHookGuard prof_guard(*c);
    HIP_TRY(hipSetDevice(c->device));
    return collect_est_pose(*c, ticket, poses);
}

int p2p_est_pose_collect_gathered(p2p_ctx* ctx, p2p_comm* comm, int ticket, p2p_pose* poses, int n_max, p2p_pose* gathered)
{
    // (argument errors are local: nothing was enqueued and the caller's peers are its own to unblock; poses may be null for an empty shard)
    if (!ctx || !comm || !gathered || n_max < 1 || (!poses && ticket != P2P_TICKET_NONE)) { set_error("p2p_est_pose_collect_gathered: bad arguments"); return P2P_ERR_INVALID_ARG; }
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    ProfHookGuard prof_guard(*c);
    HIP_TRY(hipSetDevice(c->device));
    return collect_est_pose(*c, ticket, poses, reinterpret_cast<Comm*>(comm), n_max, gathered);
}

int p2p_debug_back_resize(p2p_ctx* ctx, const float* prob, const float* pred, const float* non_gray, int out_h, int out_w, double th_inlier,
                          int generation, unsigned char* q, unsigned char* below, unsigned char* ng_out)
{
    if (!ctx || !prob || !pred || !non_gray || !q || !below || !ng_out || out_h < 1 || out_w < 1 || out_h > 1024 || out_w > 1024) {
        set_error("p2p_debug_back_resize: bad arguments");
        return P2P_ERR_INVALID_ARG;
    }
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    // five planes as the pipeline keeps them: [prob | pred r | g | b | non_gray][128 * 128] doubles (float32-valued where the reference's array is float32)
    std::vector<double> h(5 * 16384);
    double lo[5], hi[5];
    for (int k = 0; k < 5; ++k) { lo[k] = 1e300; hi[k] = -1e300; }
    for (int i = 0; i < 16384; ++i) {
        h[i] = prob[i];
        for (int ch = 0; ch < 3; ++ch) h[(1 + ch) * 16384 + i] = pred[3 * i + ch];
        h[4 * 16384 + i] = non_gray[i];
        for (int k = 0; k < 5; ++k) { lo[k] = std::min(lo[k], h[k * 16384 + i]); hi[k] = std::max(hi[k], h[k * 16384 + i]); }
    }
    DevBuf planes, tmp, items, dq, db, dg;
    int rc;
    const size_t npx = (size_t)out_h * out_w;
    auto cleanup = [&]() { planes.release(); tmp.release(); items.release(); dq.release(); db.release(); dg.release(); };
    if ((rc = planes.reserve(h.size() * 8)) || (rc = tmp.reserve(h.size() * 8)) || (rc = items.reserve(5 * sizeof(AaItem))) ||
        (rc = dq.reserve(npx * 3)) || (rc = db.reserve(npx)) || (rc = dg.reserve(npx))) { cleanup(); return rc; }
    hipError_t e = hipMemcpyAsync(planes.p, h.data(), h.size() * 8, hipMemcpyHostToDevice, st);
    CandRange R;
    R.pmin = lo[0]; R.pmax = hi[0];
    R.qmin = std::min(lo[1], std::min(lo[2], lo[3])); R.qmax = std::max(hi[1], std::max(hi[2], hi[3]));
    R.gmin = lo[4]; R.gmax = hi[4];
    if (e == hipSuccess && generation && out_h < 128) {          // scikit-image 0.17 / 0.18: anti-aliasing filter before a shrinking resize (square outputs, like the path's)
        AaTable tab;
        if ((rc = aa_table_get(c->device, &tab))) { cleanup(); return rc; }
        std::vector<int> rad(tab.max_side + 1), off(tab.max_side + 1);
        e = hipMemcpy(rad.data(), tab.rad, rad.size() * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(off.data(), tab.off, off.size() * 4, hipMemcpyDeviceToHost);
        if (e == hipSuccess && rad[out_h] > 0) {
            AaItem it[5];
            for (int k = 0; k < 5; ++k) {
                memset(&it[k], 0, sizeof(AaItem));
                it[k].a = planes.as<double>() + (size_t)k * 16384; it[k].tmp = tmp.as<double>() + (size_t)k * 16384;
                it[k].H = it[k].W = 128; it[k].C = 1; it[k].radius = rad[out_h]; it[k].w = tab.w + off[out_h];
                it[k].mode = 1; it[k].cval = k == 0 ? 1.0 : (k == 4 ? 0.0 : 0.5); it[k].round32 = k < 4;
            }
            e = hipMemcpyAsync(items.p, it, sizeof(it), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = launch_aa_filter(items.as<AaItem>(), 5, 16384, st);
            if (e == hipSuccess) e = hipMemcpyAsync(it, items.p, sizeof(it), hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e == hipSuccess) {
                R.pmin = it[0].vmin; R.pmax = it[0].vmax;
                R.qmin = std::min(it[1].vmin, std::min(it[2].vmin, it[3].vmin)); R.qmax = std::max(it[1].vmax, std::max(it[2].vmax, it[3].vmax));
                R.gmin = it[4].vmin; R.gmax = it[4].vmax;
            }
        }
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(back_resize_probe_kernel, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, st, planes.as<double>(), R, out_h, out_w, th_inlier,
                           generation == 1 ? 1 : 0, dq.as<unsigned char>(), db.as<unsigned char>(), dg.as<unsigned char>());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(q, dq.p, npx * 3, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(below, db.p, npx, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipMemcpyAsync(ng_out, dg.p, npx, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess) { set_error("p2p_debug_back_resize: %s", hipGetErrorString(e)); return P2P_ERR_HIP; }
    return P2P_OK;
}

int p2p_est_pose_batch(p2p_ctx* ctx, const p2p_object* objects, int n_objects, const p2p_image* images, int n_images,
                       const p2p_detection* dets, int n_dets, p2p_pose* poses, const p2p_est_pose_opts* opts)
{
    if (!ctx || n_dets < 0 || (n_dets > 0 && (!objects || !images || !dets || !poses || n_objects <= 0 || n_images <= 0))) {
        set_error("p2p_est_pose_batch: bad arguments");
        return P2P_ERR_INVALID_ARG;
    }
    if (n_dets == 0) return P2P_OK;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    ProfHookGuard prof_guard(*c);
    HIP_TRY(hipSetDevice(c->device));
    p2p_est_pose_opts o;
    memset(&o, 0, sizeof(o));
    if (opts) o = *opts;
    return run_est_pose(*c, objects, n_objects, images, n_images, dets, n_dets, poses, o, nullptr);
}

int p2p_pnp_ransac_batch(p2p_ctx* ctx, const double* camK, const double* obj_pts, const double* img_pts, const int* offsets,
                         int n_problems, int iterations, double reprojection_error, double confidence, double* R, double* t,
                         int* info, int* ok, unsigned char* inlier_mask)
{
    if (!ctx || n_problems < 0 || (n_problems > 0 && (!camK || !obj_pts || !img_pts || !offsets || !R || !t || !info || !ok))) {
        set_error("p2p_pnp_ransac_batch: bad arguments");
        return P2P_ERR_INVALID_ARG;
    }
    if (iterations > P2P_MAX_RANSAC_ITERATIONS) {
        set_error("p2p_pnp_ransac_batch: iterations %d exceeds P2P_MAX_RANSAC_ITERATIONS (%d)", iterations, P2P_MAX_RANSAC_ITERATIONS);
        return P2P_ERR_INVALID_ARG;
    }
    if (n_problems == 0) return P2P_OK;
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    ProfHookGuard prof_guard(*c);
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t st = c->stream;
    const int N = offsets[n_problems];
    // OpenCV converts the point sets to float32 before RANSAC; store them SoA per problem
    std::vector<float> pts((size_t)std::max(N, 1) * 5);
    std::vector<PnpProblem> pb(n_problems);
    DevBuf dpts, dprob, dres, dmask, dhyp;
    int rc;
    if ((rc = dpts.reserve(pts.size() * 4)) || (rc = dprob.reserve(sizeof(PnpProblem) * n_problems)) ||
        (rc = dres.reserve(sizeof(PnpResult) * n_problems)) || (rc = dmask.reserve((size_t)std::max(N, 1))) ||
        (rc = dhyp.reserve(pnp_workspace_bytes(n_problems)))) {
        dpts.release(); dprob.release(); dres.release(); dmask.release(); dhyp.release();
        return rc;
    }
    int max_n = 1;
    for (int p = 0; p < n_problems; ++p) {
        const int o = offsets[p], n = offsets[p + 1] - o;
        max_n = std::max(max_n, n);
        float* base = pts.data() + (size_t)o * 5;
        for (int i = 0; i < n; ++i) {
            base[i] = (float)obj_pts[3 * (size_t)(o + i)];
            base[n + i] = (float)obj_pts[3 * (size_t)(o + i) + 1];
            base[2 * n + i] = (float)obj_pts[3 * (size_t)(o + i) + 2];
            base[3 * n + i] = (float)img_pts[2 * (size_t)(o + i)];
            base[4 * n + i] = (float)img_pts[2 * (size_t)(o + i) + 1];
        }
        pb[p].pts = dpts.as<float>() + (size_t)o * 5;
        pb[p].cap = n; pb[p].n = n;
        for (int k = 0; k < 9; ++k) pb[p].K[k] = camK[9 * p + k];
        pb[p].mask = inlier_mask ? dmask.as<unsigned char>() + o : nullptr;
    }
    auto cleanup = [&]() { dpts.release(); dprob.release(); dres.release(); dmask.release(); dhyp.release(); };
    hipError_t e;
    std::vector<PnpResult> res(n_problems);
    if ((e = hipMemcpyAsync(dpts.p, pts.data(), pts.size() * 4, hipMemcpyHostToDevice, st)) != hipSuccess ||
        (e = hipMemcpyAsync(dprob.p, pb.data(), sizeof(PnpProblem) * n_problems, hipMemcpyHostToDevice, st)) != hipSuccess ||
        (e = hipMemsetAsync(dmask.p, 0, (size_t)std::max(N, 1), st)) != hipSuccess ||
        (e = launch_pnp_ransac(dprob.as<PnpProblem>(), dres.as<PnpResult>(), n_problems, iterations > 0 ? iterations : 100,
                               reprojection_error > 0 ? reprojection_error : 5.0, confidence > 0 ? confidence : 0.99, 5, max_n, dhyp.as<double>(), st)) != hipSuccess ||
        (e = hipMemcpyAsync(res.data(), dres.p, sizeof(PnpResult) * n_problems, hipMemcpyDeviceToHost, st)) != hipSuccess ||
        (inlier_mask && (e = hipMemcpyAsync(inlier_mask, dmask.p, (size_t)N, hipMemcpyDeviceToHost, st)) != hipSuccess) ||
        (e = hipStreamSynchronize(st)) != hipSuccess) {
        set_error("p2p_pnp_ransac_batch: %s", hipGetErrorString(e));
        cleanup();
        return P2P_ERR_HIP;
    }
    for (int p = 0; p < n_problems; ++p) {
        for (int k = 0; k < 9; ++k) R[9 * p + k] = res[p].R[k];
        for (int k = 0; k < 3; ++k) t[3 * p + k] = res[p].t[k];
        info[3 * p] = res[p].n_inliers; info[3 * p + 1] = res[p].iters; info[3 * p + 2] = res[p].best_iter;
        ok[p] = res[p].ok;
    }
    cleanup();
    return P2P_OK;
}

}  // extern "C"
