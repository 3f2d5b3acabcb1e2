// Batched EPnP + RANSAC on gfx950 (one problem = one stage-2 candidate).
//
// Replaces, per candidate, the call
//     cv2.solvePnPRansac(obj, img, camK, None, flags=SOLVEPNP_EPNP, reprojectionError=5,
//                        iterationsCount=100)  + cv2.Rodrigues(rvec)
// of the reference (pix2pose_model/recognition.py:216-223; OpenCV 3.4.2.17, requirements.txt:3).
//
// OpenCV's RANSAC is sequential but its sampling is not data dependent: the RNG is re-seeded with
// (uint64)-1 on every call and each iteration draws five distinct indices, so all minimal sets are
// known up front.  The work is five kernels (launch_pnp_ransac):
//   1. pnp_hypotheses_kernel: a lane replays the multiply-with-carry generator, lane quads solve the 5-point EPnP
//      hypotheses in fp64 -- lazily, in rounds [0, 16), [16, 64), [64, 100): RANSAC's adaptive bound rarely asks for more
//      (launches of a handful of problems: teams of 32 lanes per hypothesis, rounds [0, 48), [48, 100) -- jacobi12_team),
//   2. pnp_count_kernel: inlier counts of the round's hypotheses (integers: independent workgroups over slices of the points),
//   3. pnp_score_kernel: OpenCV's "best so far" rule and adaptive iteration bound replayed over the counts in OpenCV's order
//      (it stops where OpenCV stops and picks the hypothesis OpenCV picks; a problem that needs the next round parks its
//      state), then the refit's reductions over the inlier set (the 2n x 12 system is never formed: M^T M has only four
//      distinct weighted Gram sums of the barycentric coordinates),
//   4. pnp_fit_solve_kernel: the 12x12 SVD and the same beta / Gauss-Newton / absolute-orientation steps (quad / team forms likewise),
//   5. pnp_fit_select_kernel: the candidate with the smallest mean reprojection error, Rodrigues round trip.
// All arithmetic that decides an inlier uses OpenCV's types (float32 point storage, float32
// projected points and squared distance) with FMA contraction disabled.
#include "pipeline.h"
#include <algorithm>

// minimum waves per SIMD the two PnP kernels are compiled for (caps their register allocation)
#ifndef P2P_PNP_HYP_WAVES
#define P2P_PNP_HYP_WAVES 1
#endif
#ifndef P2P_PNP_SOLVE_WAVES
#define P2P_PNP_SOLVE_WAVES 1
#endif
#ifndef P2P_PNP_FIT_WAVES
#define P2P_PNP_FIT_WAVES 2
#endif

#pragma clang fp contract(off)

namespace p2p {
namespace pnp {

constexpr double kDblEps = 2.220446049250313e-16;
constexpr double kDblMin = 2.2250738585072014e-308;

// Correctly rounded double sqrt.  The gfx950 expansion of sqrt(double) (v_rsq_f64 + Goldschmidt
// steps) is within 1 ulp but not always the IEEE result, and a 5-point EPnP system has a 2-D
// null space whose basis is decided by rounding -- so one differing ulp changes a hypothesis.
// One exact-residual correction step (fma(-g, g, x) is exact for g within 1 ulp of the root)
// makes the device agree bit for bit with a host libm.
__device__ __noinline__ double sqrt_cr_slow(double x)      // zero, subnormal-range, infinite and NaN arguments
{
    double g = __builtin_sqrt(x);
    if (!(x > 0.0) || !(g < 1.7976931348623157e308)) return g;
    const double r = __builtin_fma(-g, g, x);
    if (r > 0.0) {
        const double gn = __longlong_as_double(__double_as_longlong(g) + 1);
        const double rn = __builtin_fma(-gn, gn, x);
        if (rn >= 0.0 || r > -rn) g = gn;
    } else if (r < 0.0) {
        const double gp = __longlong_as_double(__double_as_longlong(g) - 1);
        const double rp = __builtin_fma(-gp, gp, x);
        if (rp <= 0.0 || -r > rp) g = gp;
    }
    return g;
}
// The Jacobi sweeps spend most of their instructions here (three roots per rotation, each a dependent chain), and the lanes of a wave
// solve different systems: a branchy root costs the wave every path.  So: the compiler's own sequence for arguments that need no
// scaling (x >= 2^-767: the same operations it emits, minus the range handling), then the correction WITHOUT branches -- both
// neighbours' residuals are computed and the root is selected.  The correctly rounded root is unique: same bits as before.
__device__ __forceinline__ double sqrt_cr(double x)
{
    if (!(x >= 0x1p-767 && x < 1.7976931348623157e308)) return sqrt_cr_slow(x);
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r0 = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r0, g);
    h = __builtin_fma(h, r0, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    const double r = __builtin_fma(-g, g, x);
    const double gn = __longlong_as_double(__double_as_longlong(g) + 1), gp = __longlong_as_double(__double_as_longlong(g) - 1);
    const double rn = __builtin_fma(-gn, gn, x), rp = __builtin_fma(-gp, gp, x);
    const bool up = (r > 0.0) & ((rn >= 0.0) | (r > -rn));
    const bool dn = (r < 0.0) & ((rp <= 0.0) | (-r > rp));
    return up ? gn : (dn ? gp : g);
}

// OpenCV's "is this pair already orthogonal" test, fabs(p) <= eps * sqrt(a * b), without the square root where the answer cannot depend
// on it: p^2 against eps^2 a b with a relative margin a million times wider than any rounding in either form decides all but a sliver of
// cases; inside the sliver (and wherever a product could leave the normal range) the original expression is evaluated.  Same decisions.
__device__ __forceinline__ bool jacobi_skip(const double p, const double a, const double b, const double eps)
{
    const double ab = a * b, pp = p * p, lim = (eps * eps) * ab;
    if ((p == 0.0) & (ab >= 0.0)) return true;          // 0 <= eps * sqrt(ab) whatever ab is (zero-padded rows land here)
    const bool in_range = (pp > 1e-280) & (pp < 1e280) & (ab > 1e-250) & (ab < 1e280);
    if (in_range & (pp > lim * 1.000001)) return false;
    if (in_range & (pp < lim * 0.999999)) return true;
    return fabs(p) <= eps * sqrt_cr(ab);
}

// (c, s) of a one-sided Jacobi rotation from the dot product p and the squared norms a, b of two rows (OpenCV JacobiSVDImpl_).  OpenCV
// branches on the sign of beta; the two branches are the same four operations on swapped roles, so they are written once on selected
// operands: lanes with either sign walk the same instructions (a branch costs a wave both paths).  hypot() is written out: identical
// bits on every libm (see oracle/pnp_oracle.c).
__device__ __forceinline__ void jacobi_cs(double p, const double a, const double b, double& c, double& s)
{
    p *= 2;
    const double beta = a - b, gamma = sqrt_cr(p * p + beta * beta);
    const bool neg = beta < 0;
    // beta < 0:  delta = (gamma - beta) * 0.5;  s = sqrt(delta / gamma);             c = p / (gamma * s * 2)
    // else:                                     c = sqrt((gamma + beta) / (gamma * 2)); s = p / (gamma * c * 2)
    const double num = neg ? (gamma - beta) * 0.5 : gamma + beta;
    const double den = neg ? gamma : gamma * 2;
    const double r1 = sqrt_cr(num / den);
    const double r2 = p / (gamma * r1 * 2);
    s = neg ? r1 : r2;
    c = neg ? r2 : r1;
}

// ---------------------------------------------------------------- cv::RNG (multiply with carry)
struct Rng {
    unsigned long long state;
    __device__ explicit Rng(unsigned long long s) : state(s ? s : 0xffffffffULL) {}
    __device__ unsigned next()
    {
        state = (unsigned long long)(unsigned)state * 4164903690ULL + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    __device__ int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

// ---------------------------------------------------------------- one-sided Jacobi SVD
// At: N rows of length M (the columns of the M x N matrix A).  On exit the rows are the left
// singular vectors, W the singular values in descending order, Vt (optional, N x N) the right ones.
//
// Compile-time sizes and fully unrolled loops: every index is static, so the matrix lives in
// registers (a 12x12 fp64 matrix = 288 VGPRs) instead of per-lane scratch.  The solver is a long
// serial dependency chain; with scratch arrays every element access was a memory round trip and one
// 5-point EPnP took ~3 ms.  Operation order is unchanged (sequential sums), so results are bit for
// bit those of the rolled loops.
// n_live < N: rows n_live .. N - 1 are zero padding (a narrower system run in the code of the wider one, betas_approx_any).  A zero row never
// rotates (p = 0 is "already orthogonal"), sorts behind every live row and is dropped by the back-substitution; it only has to be kept out
// of OpenCV's fill-in of zero singular values.  The live rows see the operations of jacobi_svd_t<M, n_live> in the same order.
template <int M, int N, bool WITH_V>
__device__ __forceinline__ void jacobi_svd_t(double (&At)[N * M], double (&W)[N], double* Vt, const int n_live = N)
{
    const double eps = kDblEps * 10;
    constexpr int max_iter = M > 30 ? M : 30;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) { const double t = At[i * M + k]; sd += t * t; }
        W[i] = sd;
        if (WITH_V) {
#pragma unroll
            for (int k = 0; k < N; k++) Vt[i * N + k] = (k == i) ? 1. : 0.;
        }
    }
#pragma unroll 1
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; i++)
#pragma unroll
            for (int j = i + 1; j < N; j++) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < M; k++) p += At[i * M + k] * At[j * M + k];
                if (jacobi_skip(p, a, b, eps)) continue;
                double c, s;
                jacobi_cs(p, a, b, c, s);
                a = b = 0;
#pragma unroll
                for (int k = 0; k < M; k++) {
                    const double t0 = c * At[i * M + k] + s * At[j * M + k];
                    const double t1 = -s * At[i * M + k] + c * At[j * M + k];
                    At[i * M + k] = t0; At[j * M + k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
                if (WITH_V) {
#pragma unroll
                    for (int k = 0; k < N; k++) {
                        const double t0 = c * Vt[i * N + k] + s * Vt[j * N + k];
                        const double t1 = -s * Vt[i * N + k] + c * Vt[j * N + k];
                        Vt[i * N + k] = t0; Vt[j * N + k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; k++) { const double t = At[i * M + k]; sd += t * t; }
        W[i] = sqrt_cr(sd);
    }
    // selection sort, descending (row i <-> row argmax_{k>=i} W[k]); the dynamic argmax becomes a
    // chain of predicated swaps so the rows stay in registers
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < N; k++)
            if (wj < W[k]) { j = k; wj = W[k]; }
#pragma unroll
        for (int k = i + 1; k < N; k++)
            if (j == k) {
                double t = W[i]; W[i] = W[k]; W[k] = t;
#pragma unroll
                for (int e = 0; e < M; e++) { t = At[i * M + e]; At[i * M + e] = At[k * M + e]; At[k * M + e] = t; }
                if (WITH_V) {
#pragma unroll
                    for (int e = 0; e < N; e++) { t = Vt[i * N + e]; Vt[i * N + e] = Vt[k * N + e]; Vt[k * N + e] = t; }
                }
            }
    }
    // left vectors = rows / singular value; a zero singular value gets a deterministic
    // pseudo-random direction orthogonal to the previous rows (what OpenCV's JacobiSVD does).
    Rng rng(0x12345678ULL);
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = W[i];
#pragma unroll 1
        for (int ii = 0; ii < 100 && sd <= kDblMin && i < n_live; ii++) {
            const double val0 = 1. / M;
#pragma unroll
            for (int k = 0; k < M; k++) At[i * M + k] = (rng.next() & 256) != 0 ? val0 : -val0;
#pragma unroll 1
            for (int it = 0; it < 2; it++)
#pragma unroll
                for (int j = 0; j < i; j++) {
                    sd = 0;
#pragma unroll
                    for (int k = 0; k < M; k++) sd += At[i * M + k] * At[j * M + k];
                    double asum = 0;
#pragma unroll
                    for (int k = 0; k < M; k++) {
                        const double t = At[i * M + k] - sd * At[j * M + k];
                        At[i * M + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
                    for (int k = 0; k < M; k++) At[i * M + k] *= asum;
                }
            sd = 0;
#pragma unroll
            for (int k = 0; k < M; k++) { const double t = At[i * M + k]; sd += t * t; }
            sd = sqrt_cr(sd);
        }
        const double s = sd > kDblMin ? 1 / sd : 0.;
#pragma unroll
        for (int k = 0; k < M; k++) At[i * M + k] *= s;
    }
}

// The same routine for M = N = 12 without V, ROLLED: run-time indices, the matrix in the lane's scratch memory.  Only the fall-back of the
// quad / team forms runs it (a numerically zero singular value: OpenCV's random fill-in), i.e. practically never -- but unrolled in place its
// 288 matrix registers were what sized the register file of the kernels that contain it.  Same operations in the same order: same bits.
__device__ __noinline__ void jacobi_svd12_rolled(double* __restrict__ At, double* __restrict__ W)
{
    constexpr int M = 12, N = 12;
    const double eps = kDblEps * 10;
    constexpr int max_iter = 30;
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll 1
        for (int k = 0; k < M; k++) { const double t = At[i * M + k]; sd += t * t; }
        W[i] = sd;
    }
#pragma unroll 1
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll 1
        for (int i = 0; i < N - 1; i++)
#pragma unroll 1
            for (int j = i + 1; j < N; j++) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll 1
                for (int k = 0; k < M; k++) p += At[i * M + k] * At[j * M + k];
                if (jacobi_skip(p, a, b, eps)) continue;
                double c, s;
                jacobi_cs(p, a, b, c, s);
                a = b = 0;
#pragma unroll 1
                for (int k = 0; k < M; k++) {
                    const double t0 = c * At[i * M + k] + s * At[j * M + k];
                    const double t1 = -s * At[i * M + k] + c * At[j * M + k];
                    At[i * M + k] = t0; At[j * M + k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
            }
        if (!changed) break;
    }
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll 1
        for (int k = 0; k < M; k++) { const double t = At[i * M + k]; sd += t * t; }
        W[i] = sqrt_cr(sd);
    }
#pragma unroll 1
    for (int i = 0; i < N - 1; i++) {
        int j = i;
#pragma unroll 1
        for (int k = i + 1; k < N; k++)
            if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
#pragma unroll 1
            for (int e = 0; e < M; e++) { t = At[i * M + e]; At[i * M + e] = At[j * M + e]; At[j * M + e] = t; }
        }
    }
    Rng rng(0x12345678ULL);
#pragma unroll 1
    for (int i = 0; i < N; i++) {
        double sd = W[i];
#pragma unroll 1
        for (int ii = 0; ii < 100 && sd <= kDblMin; ii++) {
            const double val0 = 1. / M;
#pragma unroll 1
            for (int k = 0; k < M; k++) At[i * M + k] = (rng.next() & 256) != 0 ? val0 : -val0;
#pragma unroll 1
            for (int it = 0; it < 2; it++)
#pragma unroll 1
                for (int j = 0; j < i; j++) {
                    sd = 0;
#pragma unroll 1
                    for (int k = 0; k < M; k++) sd += At[i * M + k] * At[j * M + k];
                    double asum = 0;
#pragma unroll 1
                    for (int k = 0; k < M; k++) {
                        const double t = At[i * M + k] - sd * At[j * M + k];
                        At[i * M + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll 1
                    for (int k = 0; k < M; k++) At[i * M + k] *= asum;
                }
            sd = 0;
#pragma unroll 1
            for (int k = 0; k < M; k++) { const double t = At[i * M + k]; sd += t * t; }
            sd = sqrt_cr(sd);
        }
        const double s = sd > kDblMin ? 1 / sd : 0.;
#pragma unroll 1
        for (int k = 0; k < M; k++) At[i * M + k] *= s;
    }
}

// ---------------------------------------------------------------- 12x12 SVD on a LANE QUAD
// The 12x12 Jacobi SVD of M^T M is most of an EPnP solve, and on one lane it is bound by instruction issue and by its registers: the
// matrix alone is 288 of them -- more than the 256 architectural VGPRs, so every rotation shuffles rows through the accumulation
// registers and scratch -- and a sweep is 66 pairs x ~600 instructions.  Here lanes (4h .. 4h + 3) share one solve: lane q holds
// columns [3q, 3q + 3) of every row (36 doubles).  Products and rotations are per column; the three sums of a pair -- the dot
// product and the two squared norms -- are OpenCV's sequential sums over k = 0..11, so they stay ONE chain of additions in the same
// order: lane 0 adds its three terms and hands the partial to lane 1 (DPP), and so on; lane 3 broadcasts the total.  Same operations
// on the same values in the same order => the same bits as jacobi_svd_t<12, 12, false>.
template <int CTRL>
__device__ __forceinline__ double quad_dpp(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int QUAD_PREV = 0x90;      // every lane reads the lane before it in its quad (lane 0: itself): quad_perm [0, 0, 1, 2]
template <int K> __device__ __forceinline__ double quad_bcast(double v) { return quad_dpp<K | (K << 2) | (K << 4) | (K << 6)>(v); }   // lane K -> all four

// s = 0; for k = 0..11: s += x_k   with x_k on lane k / 3 (x[k % 3]); all four lanes get the result
__device__ __forceinline__ double chain12(const double (&x)[3])
{
    double s = 0;
    s += x[0]; s += x[1]; s += x[2];              // lane 0 holds the partial over k = 0..2
#pragma unroll
    for (int q = 1; q < 4; q++) {                  // after pass q, lane q holds the partial over k = 0 .. 3q + 2
        s = quad_dpp<QUAD_PREV>(s);
        s += x[0]; s += x[1]; s += x[2];
    }
    return quad_bcast<3>(s);
}

// two such chains at once (independent: the additions interleave)
__device__ __forceinline__ void chain12x2(const double (&x)[3], const double (&y)[3], double& sx, double& sy)
{
    double a = 0, b = 0;
    a += x[0]; b += y[0]; a += x[1]; b += y[1]; a += x[2]; b += y[2];
#pragma unroll
    for (int q = 1; q < 4; q++) {
        a = quad_dpp<QUAD_PREV>(a); b = quad_dpp<QUAD_PREV>(b);
        a += x[0]; b += y[0]; a += x[1]; b += y[1]; a += x[2]; b += y[2];
    }
    sx = quad_bcast<3>(a);
    sy = quad_bcast<3>(b);
}

// A[i * 3 + e] = element (i, 3 q + e) of the row image (a symmetric matrix: rows == columns).  On exit the rows are the left singular
// vectors (this lane's columns of them), W the singular values, descending.  Returns false when a singular value is (numerically) zero:
// OpenCV then fills in pseudo-random directions -- the caller redoes the solve on the single-lane path, which implements that.
__device__ __forceinline__ bool jacobi12_quad(double (&A)[36], double (&W)[12])
{
    const double eps = kDblEps * 10;
    constexpr int max_iter = 30;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        double x[3];
#pragma unroll
        for (int e = 0; e < 3; e++) { const double t = A[i * 3 + e]; x[e] = t * t; }
        W[i] = chain12(x);
    }
#pragma unroll 1
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < 11; i++)
#pragma unroll
            for (int j = i + 1; j < 12; j++) {
                double a = W[i], b = W[j];
                double x[3], y[3];
#pragma unroll
                for (int e = 0; e < 3; e++) x[e] = A[i * 3 + e] * A[j * 3 + e];
                double p = chain12(x);
                if (jacobi_skip(p, a, b, eps)) continue;                 // the same decision on all four lanes: p, a, b are the same values
                double c, s;
                jacobi_cs(p, a, b, c, s);
#pragma unroll
                for (int e = 0; e < 3; e++) {
                    const double t0 = c * A[i * 3 + e] + s * A[j * 3 + e];
                    const double t1 = -s * A[i * 3 + e] + c * A[j * 3 + e];
                    A[i * 3 + e] = t0; A[j * 3 + e] = t1;
                    x[e] = t0 * t0; y[e] = t1 * t1;
                }
                chain12x2(x, y, a, b);
                W[i] = a; W[j] = b;
                changed = true;
            }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < 12; i++) {
        double x[3];
#pragma unroll
        for (int e = 0; e < 3; e++) { const double t = A[i * 3 + e]; x[e] = t * t; }
        W[i] = sqrt_cr(chain12(x));
    }
    // selection sort, descending, as predicated swaps (see jacobi_svd_t); W is the same on all four lanes, so are the swaps
#pragma unroll
    for (int i = 0; i < 11; i++) {
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < 12; k++)
            if (wj < W[k]) { j = k; wj = W[k]; }
#pragma unroll
        for (int k = i + 1; k < 12; k++)
            if (j == k) {
                double t = W[i]; W[i] = W[k]; W[k] = t;
#pragma unroll
                for (int e = 0; e < 3; e++) { t = A[i * 3 + e]; A[i * 3 + e] = A[k * 3 + e]; A[k * 3 + e] = t; }
            }
    }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        const double sd = W[i];
        if (sd <= kDblMin) ok = false;
        const double s = sd > kDblMin ? 1 / sd : 0.;
#pragma unroll
        for (int e = 0; e < 3; e++) A[i * 3 + e] *= s;
    }
    return ok;
}

// rows 8..11 of the left singular vectors with all twelve columns on all four lanes: u8[r * 12 + k] = row 8 + r
__device__ __forceinline__ void gather_rows8(const double (&A)[36], double (&u8)[48])
{
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const double v = A[(8 + r) * 3 + e];
            u8[r * 12 + e] = quad_bcast<0>(v);
            u8[r * 12 + 3 + e] = quad_bcast<1>(v);
            u8[r * 12 + 6 + e] = quad_bcast<2>(v);
            u8[r * 12 + 9 + e] = quad_bcast<3>(v);
        }
}

// ---------------------------------------------------------------- 12x12 SVD on a TEAM of lane quads (the latency form)
// A handful of problems (one detection at a time: the reference's own call shape) leaves the chip idle while a lone wave walks the 66
// pair visits of a sweep one after the other.  The pairs (i, j) of one anti-diagonal i + j = t share no row and have their inputs once
// the diagonals before are done, so a sweep is 21 steps of up to six INDEPENDENT rotations: a team of 32 lanes (eight quads, six of
// them with a pair) walks the diagonals, quad g of the team taking pair (max(0, t - 11) + g, t - i).  The matrix lives in LDS (row
// image, 12 x 12 doubles + the 12 squared norms per team); a quad reads its two rows, rotates them exactly as jacobi12_quad does (lane
// q = columns 3q .. 3q + 2, the three sums as one chain of additions in OpenCV's order) and writes them back.  Every rotation sees the
// values the sequential order would hand it => the same bits as jacobi12_quad / jacobi_svd_t<12, 12, false>.  The code of a step is
// one loop body with run-time row indices (a few KB) where the register form is 66 unrolled bodies per sweep.
constexpr int TEAM_LANES = 32;
constexpr int TEAM_DOUBLES = 160;          // 144 matrix + 12 norms + padding
typedef __attribute__((address_space(3))) double lds_double;      // the team's matrix is handed down as an LDS pointer: ds_read / ds_write, not flat accesses
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sm: this team's TEAM_DOUBLES of LDS, the matrix already stored (sm[i * 12 + k], symmetric) and visible.  tl = lane within the team.
// live = false: the team only keeps the wave company.  On exit u8 = rows 8..11 of the left singular vectors (all twelve columns) on
// every lane of the team.  Returns false when a singular value is (numerically) zero (see jacobi12_quad).
__device__ __forceinline__ bool jacobi12_team(lds_double* sm, const int tl, const bool live, double (&u8)[48])
{
    const double eps = kDblEps * 10;
    constexpr int max_iter = 30;
    const int g = tl >> 2, q = tl & 3;
    lds_double* sW = sm + 144;
    // squared norms of rows g and g + 6
    if (g < 6) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int i = g + 6 * h;
            double x[3];
#pragma unroll
            for (int e = 0; e < 3; e++) { const double t = sm[i * 12 + 3 * q + e]; x[e] = t * t; }
            const double w = chain12(x);
            if (q == 0) sW[i] = w;
        }
    }
    wave_lds_sync();
    // Sweeps overlap: pair (i, j) of sweep s has its inputs at step 12 s + i + j (the last visits of its rows: (i, j - 1) or (i - 1, i) one or two
    // steps before, (i - 1, j) one step before, for i = 0 the rows' last visits of sweep s - 1 at 12 s + j - 1), so while sweep s - 1 walks its
    // diagonals 13 .. 21, sweep s walks 1 .. 9 -- never more than six pairs in one step.  OpenCV leaves after a sweep without a rotation; the
    // diagonals of the next sweep taken by then found the matrix that sweep had found: they rotated nothing either.
    bool team_live = live;
    bool chg_lo = false, chg_hi = false;            // "changed" of the older / the newer sweep in flight
    const int team_shift = (int)(threadIdx.x & 63 & ~(TEAM_LANES - 1));
#pragma unroll 1
    for (int s_hi = 0; s_hi <= max_iter; s_hi++) {
        if (__builtin_amdgcn_ballot_w64(team_live) == 0) break;
#pragma unroll 1
        for (int t_hi = 1; t_hi <= 12; t_hi++) {
            const int t_lo = t_hi + 12;
            const int n_lo = (s_hi >= 1 && t_lo <= 21) ? ((t_lo - 1) >> 1) - (t_lo - 11) + 1 : 0;
            const bool lo = g < n_lo;
            const int t = lo ? t_lo : t_hi;
            const int i = (t > 11 ? t - 11 : 0) + (lo ? g : g - n_lo), j = t - i;
            const bool valid = team_live & (i < j) & (lo | (s_hi < max_iter));
            if (valid) {
                double a = sW[i], b = sW[j];
                double ri[3], rj[3], x[3], y[3];
#pragma unroll
                for (int e = 0; e < 3; e++) { ri[e] = sm[i * 12 + 3 * q + e]; rj[e] = sm[j * 12 + 3 * q + e]; x[e] = ri[e] * rj[e]; }
                const double p = chain12(x);
                if (!jacobi_skip(p, a, b, eps)) {
                    double c, s;
                    jacobi_cs(p, a, b, c, s);
#pragma unroll
                    for (int e = 0; e < 3; e++) {
                        const double t0 = c * ri[e] + s * rj[e];
                        const double t1 = -s * ri[e] + c * rj[e];
                        sm[i * 12 + 3 * q + e] = t0; sm[j * 12 + 3 * q + e] = t1;
                        x[e] = t0 * t0; y[e] = t1 * t1;
                    }
                    chain12x2(x, y, a, b);
                    if (q == 0) { sW[i] = a; sW[j] = b; }
                    if (lo) chg_lo = true; else chg_hi = true;
                }
            }
            wave_lds_sync();
            if (t_hi == 9 && s_hi >= 1) {           // sweep s_hi - 1 has walked its last diagonal: the team's quads decide together
                const unsigned long long any = __builtin_amdgcn_ballot_w64(chg_lo);
                if (((any >> team_shift) & 0xffffffffULL) == 0) team_live = false;
            }
        }
        chg_lo = chg_hi; chg_hi = false;
    }
    if (g < 6) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int i = g + 6 * h;
            double x[3];
#pragma unroll
            for (int e = 0; e < 3; e++) { const double t = sm[i * 12 + 3 * q + e]; x[e] = t * t; }
            const double w = sqrt_cr(chain12(x));
            if (q == 0) sW[i] = w;
        }
    }
    wave_lds_sync();
    // selection sort, descending, on the singular values with the row index riding along (jacobi_svd_t swaps the rows themselves)
    double W[12];
    int idx[12];
#pragma unroll
    for (int i = 0; i < 12; i++) { W[i] = sW[i]; idx[i] = i; }
#pragma unroll
    for (int i = 0; i < 11; i++) {
        int j = i;
        double wj = W[i];
#pragma unroll
        for (int k = i + 1; k < 12; k++)
            if (wj < W[k]) { j = k; wj = W[k]; }
#pragma unroll
        for (int k = i + 1; k < 12; k++)
            if (j == k) {
                const double t = W[i]; W[i] = W[k]; W[k] = t;
                const int ti = idx[i]; idx[i] = idx[k]; idx[k] = ti;
            }
    }
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 12; i++)
        if (W[i] <= kDblMin) ok = false;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double sd = W[8 + r];
        const double s = sd > kDblMin ? 1 / sd : 0.;
        const lds_double* row = sm + idx[8 + r] * 12;
#pragma unroll
        for (int k = 0; k < 12; k++) u8[r * 12 + k] = row[k] * s;
    }
    return ok;
}

// x = V diag(1/w) U^T b, singular values below sum(w)*2*eps dropped (OpenCV SVD back-substitution)
template <int M, int N>
__device__ __forceinline__ void svd_backsubst_t(const double (&w)[N], const double (&ut)[N * M], const double (&vt)[N * N],
                                                const double* b, double* x)
{
    double thr = 0;
#pragma unroll
    for (int i = 0; i < N; i++) thr += w[i];
    thr *= kDblEps * 2;
#pragma unroll
    for (int j = 0; j < N; j++) x[j] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double wi = w[i];
        if (fabs(wi) <= thr) continue;
        wi = 1 / wi;
        double s = 0;
#pragma unroll
        for (int k = 0; k < M; k++) s += ut[i * M + k] * b[k];
        s *= wi;
#pragma unroll
        for (int j = 0; j < N; j++) x[j] += s * vt[i * N + j];
    }
}

// least squares A x = b (A row-major 6 x N, N <= 5) through the SVD
template <int N>
__device__ __forceinline__ void solve_svd6(const double* A, const double* b, double* x, const int n_live = N)
{
    double w[N], ut[N * 6], vt[N * N];
#pragma unroll
    for (int i = 0; i < N; i++)
#pragma unroll
        for (int k = 0; k < 6; k++) ut[i * 6 + k] = A[k * N + i];
    jacobi_svd_t<6, N, true>(ut, w, vt, n_live);
    svd_backsubst_t<6, N>(w, ut, vt, b, x);
}

// SVD of a row-major 3x3: w, ut (rows = left vectors), vt
__device__ __forceinline__ void svd3(const double* A, double (&w)[3], double (&ut)[9], double (&vt)[9])
{
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) ut[i * 3 + k] = A[k * 3 + i];
    jacobi_svd_t<3, 3, true>(ut, w, vt);
}

// 3x3 back-substitution used for the control-point inverse
__device__ __forceinline__ void svd3_backsubst(const double (&w)[3], const double (&ut)[9], const double (&vt)[9], const double* b, double* x)
{
    svd_backsubst_t<3, 3>(w, ut, vt, b, x);
}

__device__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ double dist2(const double* a, const double* b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

// ---------------------------------------------------------------- Rodrigues
__device__ void rodrigues_v2r(const double* r, double* R)
{
    const double theta = sqrt_cr(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < kDblEps) {
        #pragma unroll
        for (int k = 0; k < 9; k++) R[k] = (k % 4 == 0) ? 1. : 0.;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, it = 1. / theta;
    const double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    const double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    const double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    #pragma unroll
    for (int k = 0; k < 9; k++) R[k] = c * (k % 4 == 0 ? 1. : 0.) + c1 * rrt[k] + s * rx[k];
}

__device__ void rodrigues_r2v(const double* R, double* r)
{
    double w[3], ut[9], vt[9], Rn[9];
    svd3(R, w, ut, vt);
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
            #pragma unroll
            for (int k = 0; k < 3; k++) s += ut[k * 3 + i] * vt[k * 3 + j];
            Rn[i * 3 + j] = s;
        }
    double rx = Rn[7] - Rn[5], ry = Rn[2] - Rn[6], rz = Rn[3] - Rn[1];
    const double s = sqrt_cr((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (Rn[0] + Rn[4] + Rn[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (Rn[0] + 1) * 0.5; rx = sqrt_cr(t > 0. ? t : 0.);
        t = (Rn[4] + 1) * 0.5; ry = sqrt_cr(t > 0. ? t : 0.) * (Rn[1] < 0 ? -1. : 1.);
        t = (Rn[8] + 1) * 0.5; rz = sqrt_cr(t > 0. ? t : 0.) * (Rn[2] < 0 ? -1. : 1.);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (Rn[5] > 0) != (ry * rz > 0)) rz = -rz;
        theta /= sqrt_cr(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
        return;
    }
    double vth = 1 / (2 * s);
    vth *= theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

// ---------------------------------------------------------------- EPnP pieces shared by both paths
struct Cam { double fu, fv, uc, vc; };

// control points from centroid + covariance (3x3 SVD): cws[4][3]
__device__ void control_points(const double* c0, const double* ptp, int n, double cws[4][3])
{
    double dc[3], uct[9], vt[9];
    svd3(ptp, dc, uct, vt);
    #pragma unroll
    for (int j = 0; j < 3; j++) cws[0][j] = c0[j];
    #pragma unroll
    for (int i = 1; i < 4; i++) {
        const double k = sqrt_cr(dc[i - 1] / n);
        #pragma unroll
        for (int j = 0; j < 3; j++) cws[i][j] = c0[j] + k * uct[3 * (i - 1) + j];
    }
}

// inverse (SVD pseudo-inverse) of CC[i][j-1] = cws[j][i] - cws[0][i]
__device__ void cc_inverse(const double cws[4][3], double* ci)
{
    double cc[9], w[3], ut[9], vt[9];
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
    svd3(cc, w, ut, vt);
    #pragma unroll
    for (int col = 0; col < 3; col++) {
        double b[3] = {0, 0, 0}, x[3];
        b[col] = 1;
        svd3_backsubst(w, ut, vt, b, x);
        #pragma unroll
        for (int r = 0; r < 3; r++) ci[3 * r + col] = x[r];
    }
}

__device__ void barycentric(const double* ci, const double cws[4][3], const double* p, double* a)
{
    #pragma unroll
    for (int j = 0; j < 3; j++)
        a[1 + j] = ci[3 * j] * (p[0] - cws[0][0]) + ci[3 * j + 1] * (p[1] - cws[0][1]) + ci[3 * j + 2] * (p[2] - cws[0][2]);
    a[0] = 1.0 - a[1] - a[2] - a[3];
}

// u8: rows 8..11 of the 12x12 left singular vectors (the four smallest singular values), 12 values each
__device__ void compute_L_6x10(const double* u8, double* l)
{
    const double* v[4] = {u8 + 12 * 3, u8 + 12 * 2, u8 + 12 * 1, u8};
    double dv[4][6][3];
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        int a = 0, b = 1;
        #pragma unroll
        for (int j = 0; j < 6; j++) {
            dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
            dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
            dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
            b++;
            if (b > 3) { a++; b = a + 1; }
        }
    }
    #pragma unroll
    for (int i = 0; i < 6; i++) {
        double* row = l + 10 * i;
        row[0] = dot3(dv[0][i], dv[0][i]);
        row[1] = 2.0 * dot3(dv[0][i], dv[1][i]);
        row[2] = dot3(dv[1][i], dv[1][i]);
        row[3] = 2.0 * dot3(dv[0][i], dv[2][i]);
        row[4] = 2.0 * dot3(dv[1][i], dv[2][i]);
        row[5] = dot3(dv[2][i], dv[2][i]);
        row[6] = 2.0 * dot3(dv[0][i], dv[3][i]);
        row[7] = 2.0 * dot3(dv[1][i], dv[3][i]);
        row[8] = 2.0 * dot3(dv[2][i], dv[3][i]);
        row[9] = dot3(dv[3][i], dv[3][i]);
    }
}

__device__ void compute_rho(const double cws[4][3], double* rho)
{
    rho[0] = dist2(cws[0], cws[1]); rho[1] = dist2(cws[0], cws[2]); rho[2] = dist2(cws[0], cws[3]);
    rho[3] = dist2(cws[1], cws[2]); rho[4] = dist2(cws[1], cws[3]); rho[5] = dist2(cws[2], cws[3]);
}

// betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44]
__device__ void betas_approx_1(const double* l, const double* rho, double* betas)      // [B11 B12 B13 B14]
{
    double l4[24], b4[4];
    #pragma unroll
    for (int i = 0; i < 6; i++) {
        l4[4 * i] = l[10 * i]; l4[4 * i + 1] = l[10 * i + 1]; l4[4 * i + 2] = l[10 * i + 3]; l4[4 * i + 3] = l[10 * i + 6];
    }
    solve_svd6<4>(l4, rho, b4);
    if (b4[0] < 0) {
        betas[0] = sqrt_cr(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0];
    } else {
        betas[0] = sqrt_cr(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0];
    }
}

__device__ void betas_approx_2(const double* l, const double* rho, double* betas)      // [B11 B12 B22]
{
    double l3[18], b3[3];
    #pragma unroll
    for (int i = 0; i < 6; i++) { l3[3 * i] = l[10 * i]; l3[3 * i + 1] = l[10 * i + 1]; l3[3 * i + 2] = l[10 * i + 2]; }
    solve_svd6<3>(l3, rho, b3);
    if (b3[0] < 0) {
        betas[0] = sqrt_cr(-b3[0]);
        betas[1] = (b3[2] < 0) ? sqrt_cr(-b3[2]) : 0.0;
    } else {
        betas[0] = sqrt_cr(b3[0]);
        betas[1] = (b3[2] > 0) ? sqrt_cr(b3[2]) : 0.0;
    }
    if (b3[1] < 0) betas[0] = -betas[0];
    betas[2] = 0.0; betas[3] = 0.0;
}

__device__ void betas_approx_3(const double* l, const double* rho, double* betas)      // [B11 B12 B22 B13 B23]
{
    double l5[30], b5[5];
    #pragma unroll
    for (int i = 0; i < 6; i++)
        #pragma unroll
        for (int j = 0; j < 5; j++) l5[5 * i + j] = l[10 * i + j];
    solve_svd6<5>(l5, rho, b5);
    if (b5[0] < 0) {
        betas[0] = sqrt_cr(-b5[0]);
        betas[1] = (b5[2] < 0) ? sqrt_cr(-b5[2]) : 0.0;
    } else {
        betas[0] = sqrt_cr(b5[0]);
        betas[1] = (b5[2] > 0) ? sqrt_cr(b5[2]) : 0.0;
    }
    if (b5[1] < 0) betas[0] = -betas[0];
    betas[2] = b5[3] / betas[0];
    betas[3] = 0.0;
}

// The three starting points in ONE instruction stream, for lanes that hold different cases side by side (the quad / team forms: walked one
// after the other the three routines above were a third of a solve).  Case c's system is written into the 6 x 5 frame of case 3 -- its own
// columns first, zero columns behind them -- and solved by the same code; the live columns go through exactly the operations of
// solve_svd6<4> / <3> / <5> (see jacobi_svd_t), so the betas are those of betas_approx_c bit for bit.
__device__ void betas_approx_any(const double* l, const double* rho, const int c, double* betas)
{
    double l5[30], b5[5];
    #pragma unroll
    for (int i = 0; i < 6; i++) {
        l5[5 * i] = l[10 * i];
        l5[5 * i + 1] = l[10 * i + 1];
        l5[5 * i + 2] = c == 1 ? l[10 * i + 3] : l[10 * i + 2];
        l5[5 * i + 3] = c == 1 ? l[10 * i + 6] : c == 3 ? l[10 * i + 3] : 0.0;
        l5[5 * i + 4] = c == 3 ? l[10 * i + 4] : 0.0;
    }
    solve_svd6<5>(l5, rho, b5, c == 1 ? 4 : c == 2 ? 3 : 5);
    const bool neg = b5[0] < 0;
    const double b0 = sqrt_cr(neg ? -b5[0] : b5[0]);
    if (c == 1) {
        betas[0] = b0;
        betas[1] = (neg ? -b5[1] : b5[1]) / b0; betas[2] = (neg ? -b5[2] : b5[2]) / b0; betas[3] = (neg ? -b5[3] : b5[3]) / b0;
    } else {
        betas[0] = b5[1] < 0 ? -b0 : b0;
        betas[1] = (neg ? b5[2] < 0 : b5[2] > 0) ? sqrt_cr(neg ? -b5[2] : b5[2]) : 0.0;
        betas[2] = c == 3 ? b5[3] / betas[0] : 0.0;
        betas[3] = 0.0;
    }
}

// Householder least squares for the 6x4 Gauss-Newton system (A destroyed)
__device__ void qr_solve(double* A, double* b, double* X)
{
    const int nr = 6, nc = 4;
    double A1[4], A2[4];
    #pragma unroll
    for (int k = 0; k < nc; k++) {
        double eta = fabs(A[k * nc + k]);
        // (the scan below mirrors epnp.cpp: rows k .. nr-2; eta only rescales the column)
        #pragma unroll
        for (int i = k + 1; i < nr; i++) {
            const double elt = fabs(A[(i - 1) * nc + k]);
            if (eta < elt) eta = elt;
        }
        if (eta == 0) return;
        double sum2 = 0.0;
        const double inv_eta = 1. / eta;
        #pragma unroll
        for (int i = k; i < nr; i++) {
            A[i * nc + k] *= inv_eta;
            sum2 += A[i * nc + k] * A[i * nc + k];
        }
        double sigma = sqrt_cr(sum2);
        if (A[k * nc + k] < 0) sigma = -sigma;
        A[k * nc + k] += sigma;
        A1[k] = sigma * A[k * nc + k];
        A2[k] = -eta * sigma;
        #pragma unroll
        for (int j = k + 1; j < nc; j++) {
            double sum = 0;
            #pragma unroll
            for (int i = k; i < nr; i++) sum += A[i * nc + k] * A[i * nc + j];
            const double tau = sum / A1[k];
            #pragma unroll
            for (int i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
        }
    }
    #pragma unroll
    for (int j = 0; j < nc; j++) {
        double tau = 0;
        #pragma unroll
        for (int i = j; i < nr; i++) tau += A[i * nc + j] * b[i];
        tau /= A1[j];
        #pragma unroll
        for (int i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
    }
    X[nc - 1] = b[nc - 1] / A2[nc - 1];
    #pragma unroll
    for (int i = nc - 2; i >= 0; i--) {
        double sum = 0;
        #pragma unroll
        for (int j = i + 1; j < nc; j++) sum += A[i * nc + j] * X[j];
        X[i] = (b[i] - sum) / A2[i];
    }
}

__device__ void gauss_newton(const double* l, const double* rho, double* betas)
{
    #pragma unroll 1
    for (int k = 0; k < 5; k++) {
        double A[24], B[6], X[4] = {0, 0, 0, 0};
        #pragma unroll
        for (int i = 0; i < 6; i++) {
            const double* r = l + i * 10;
            double* a = A + i * 4;
            a[0] = 2 * r[0] * betas[0] + r[1] * betas[1] + r[3] * betas[2] + r[6] * betas[3];
            a[1] = r[1] * betas[0] + 2 * r[2] * betas[1] + r[4] * betas[2] + r[7] * betas[3];
            a[2] = r[3] * betas[0] + r[4] * betas[1] + 2 * r[5] * betas[2] + r[8] * betas[3];
            a[3] = r[6] * betas[0] + r[7] * betas[1] + r[8] * betas[2] + 2 * r[9] * betas[3];
            B[i] = rho[i] - (r[0] * betas[0] * betas[0] + r[1] * betas[0] * betas[1] + r[2] * betas[1] * betas[1] +
                             r[3] * betas[0] * betas[2] + r[4] * betas[1] * betas[2] + r[5] * betas[2] * betas[2] +
                             r[6] * betas[0] * betas[3] + r[7] * betas[1] * betas[3] + r[8] * betas[2] * betas[3] +
                             r[9] * betas[3] * betas[3]);
        }
        qr_solve(A, B, X);
        #pragma unroll
        for (int i = 0; i < 4; i++) betas[i] += X[i];
    }
}

__device__ void compute_ccs(const double* betas, const double* u8, double ccs[4][3])
{
    #pragma unroll
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
    #pragma unroll
    for (int i = 0; i < 4; i++) {
        const double* v = u8 + 12 * (3 - i);
        #pragma unroll
        for (int j = 0; j < 4; j++)
            #pragma unroll
            for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
    }
}

// absolute orientation from ABt (3x3) and the two centroids
__device__ void orientation(const double* abt, const double* pc0, const double* pw0, double R[3][3], double t[3])
{
    double d[3], ut[9], vt[9];
    svd3(abt, d, ut, vt);
    #pragma unroll
    for (int i = 0; i < 3; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            double s = 0;
            #pragma unroll
            for (int k = 0; k < 3; k++) s += ut[k * 3 + i] * vt[k * 3 + j];
            R[i][j] = s;
        }
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                       R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) { R[2][0] = -R[2][0]; R[2][1] = -R[2][1]; R[2][2] = -R[2][2]; }
    t[0] = pc0[0] - dot3(R[0], pw0);
    t[1] = pc0[1] - dot3(R[1], pw0);
    t[2] = pc0[2] - dot3(R[2], pw0);
}

// ---------------------------------------------------------------- 5-point EPnP, one lane, sequential
// pws[15] (object, mm), us[10] (pixels).  Sequential summation order = OpenCV's.
// MODE 1 (quad): lanes (4h .. 4h + 3) run this together on the same inputs; the 12x12 SVD is shared between them (jacobi12_quad),
// everything else is computed by all four (same inputs, same bits).  q = lane & 3.  MODE 2 (team): the 32 lanes of a team run it together,
// the SVD walks the anti-diagonals of a sweep on six of their quads (jacobi12_team; sm = the team's LDS, tl = lane in the team).
template <int MODE>
__device__ void epnp5(const Cam& cam, const double* pws, const double* us, double* Rout, double* tout, const int q,
                      lds_double* sm = nullptr, const int tl = 0, const bool live = true)
{
    constexpr bool QUAD = MODE != 0;
    const int n = 5;
    double cws[4][3], c0[3] = {0, 0, 0}, ptp[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    #pragma unroll
    for (int i = 0; i < n; i++)
        #pragma unroll
        for (int j = 0; j < 3; j++) c0[j] += pws[3 * i + j];
    #pragma unroll
    for (int j = 0; j < 3; j++) c0[j] /= n;
    #pragma unroll
    for (int i = 0; i < n; i++) {
        double d[3];
        #pragma unroll
        for (int j = 0; j < 3; j++) d[j] = pws[3 * i + j] - c0[j];
        #pragma unroll
        for (int a = 0; a < 3; a++)
            #pragma unroll
            for (int b = 0; b < 3; b++) ptp[a * 3 + b] += d[a] * d[b];
    }
#ifdef P2P_PNP_TIMING
    long long tk[12]; int nk = 0;
    tk[nk++] = clock64();
#endif
    control_points(c0, ptp, n, cws);
    double ci[9], alphas[20];
    cc_inverse(cws, ci);
#ifdef P2P_PNP_TIMING
    tk[nk++] = clock64();
#endif
    #pragma unroll
    for (int i = 0; i < n; i++) barycentric(ci, cws, pws + 3 * i, alphas + 4 * i);

    double u8[48];          // rows 8..11 of the left singular vectors of M^T M
    bool solved = false;
    if (QUAD) {
        double A[36];       // this lane's three columns of M^T M (symmetric: rows == columns, the row image of A^T is the matrix itself)
        #pragma unroll
        for (int k = 0; k < 36; k++) A[k] = 0;
        #pragma unroll
        for (int i = 0; i < n; i++) {
            const double* a = alphas + 4 * i;
            double m1[12], m2[12], m1q[3], m2q[3];
            const double u = us[2 * i], v = us[2 * i + 1];
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                m1[3 * j] = a[j] * cam.fu; m1[3 * j + 1] = 0.0; m1[3 * j + 2] = a[j] * (cam.uc - u);
                m2[3 * j] = 0.0; m2[3 * j + 1] = a[j] * cam.fv; m2[3 * j + 2] = a[j] * (cam.vc - v);
            }
            #pragma unroll
            for (int e = 0; e < 3; e++) {
                m1q[e] = q == 0 ? m1[e] : q == 1 ? m1[3 + e] : q == 2 ? m1[6 + e] : m1[9 + e];
                m2q[e] = q == 0 ? m2[e] : q == 1 ? m2[3 + e] : q == 2 ? m2[6 + e] : m2[9 + e];
            }
            #pragma unroll
            for (int p = 0; p < 12; p++)
                #pragma unroll
                for (int e = 0; e < 3; e++) A[p * 3 + e] += m1[p] * m1q[e];
            #pragma unroll
            for (int p = 0; p < 12; p++)
                #pragma unroll
                for (int e = 0; e < 3; e++) A[p * 3 + e] += m2[p] * m2q[e];
        }
        if (MODE == 2) {
            if (tl < 4) {
                #pragma unroll
                for (int p = 0; p < 12; p++)
                    #pragma unroll
                    for (int e = 0; e < 3; e++) sm[p * 12 + 3 * q + e] = A[p * 3 + e];
            }
            wave_lds_sync();
            solved = jacobi12_team(sm, tl, live, u8);
        } else {
            double w12[12];
            solved = jacobi12_quad(A, w12);
            if (solved) gather_rows8(A, u8);
        }
    }
    if (!solved) {          // single-lane solve (not QUAD, or a zero singular value: OpenCV's random fill-in lives in jacobi_svd_t)
        double mtm[144];
        #pragma unroll
        for (int k = 0; k < 144; k++) mtm[k] = 0;
        #pragma unroll
        for (int i = 0; i < n; i++) {
            const double* a = alphas + 4 * i;
            double m1[12], m2[12];
            const double u = us[2 * i], v = us[2 * i + 1];
            #pragma unroll
            for (int j = 0; j < 4; j++) {
                m1[3 * j] = a[j] * cam.fu; m1[3 * j + 1] = 0.0; m1[3 * j + 2] = a[j] * (cam.uc - u);
                m2[3 * j] = 0.0; m2[3 * j + 1] = a[j] * cam.fv; m2[3 * j + 2] = a[j] * (cam.vc - v);
            }
            #pragma unroll
            for (int p = 0; p < 12; p++)
                #pragma unroll
                for (int q = 0; q < 12; q++) mtm[p * 12 + q] += m1[p] * m1[q];
            #pragma unroll
            for (int p = 0; p < 12; p++)
                #pragma unroll
                for (int q = 0; q < 12; q++) mtm[p * 12 + q] += m2[p] * m2[q];
        }
        double w12[12];
        if (QUAD) jacobi_svd12_rolled(mtm, w12);
        else jacobi_svd_t<12, 12, false>(mtm, w12, nullptr);
        #pragma unroll
        for (int k = 0; k < 48; k++) u8[k] = mtm[96 + k];
    }
    const double* ut = u8;
#ifdef P2P_PNP_TIMING
    tk[nk++] = clock64();
#endif

    double l[60], rho[6];
    compute_L_6x10(ut, l);
    compute_rho(cws, rho);

    // one beta case: Gauss-Newton from the case's starting point, camera-frame control points, absolute orientation, mean reprojection error
    auto finish_case = [&](double* betas, double (&R)[3][3], double (&t)[3]) -> double {
        gauss_newton(l, rho, betas);
#ifdef P2P_PNP_TIMING
        tk[nk++] = clock64();
#endif
        double ccs[4][3], pcs[15];
        compute_ccs(betas, ut, ccs);
        #pragma unroll
        for (int i = 0; i < n; i++) {
            const double* a = alphas + 4 * i;
            #pragma unroll
            for (int j = 0; j < 3; j++) pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
        }
        if (pcs[2] < 0.0)
            #pragma unroll
            for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];
        double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
        #pragma unroll
        for (int i = 0; i < n; i++)
            #pragma unroll
            for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
        #pragma unroll
        for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
        double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        #pragma unroll
        for (int i = 0; i < n; i++)
            #pragma unroll
            for (int j = 0; j < 3; j++) {
                abt[3 * j] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i] - pw0[0]);
                abt[3 * j + 1] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + 1] - pw0[1]);
                abt[3 * j + 2] += (pcs[3 * i + j] - pc0[j]) * (pws[3 * i + 2] - pw0[2]);
            }
        orientation(abt, pc0, pw0, R, t);
#ifdef P2P_PNP_TIMING
        tk[nk++] = clock64();
#endif
        double sum2 = 0.0;
        #pragma unroll
        for (int i = 0; i < n; i++) {
            const double* pw = pws + 3 * i;
            const double Xc = dot3(R[0], pw) + t[0], Yc = dot3(R[1], pw) + t[1];
            const double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
            const double ue = cam.uc + cam.fu * Xc * inv_Zc, ve = cam.vc + cam.fv * Yc * inv_Zc;
            const double u = us[2 * i], v = us[2 * i + 1];
            sum2 += sqrt_cr((u - ue) * (u - ue) + (v - ve) * (v - ve));
        }
        return sum2 / n;
    };
    if (QUAD) {
        // Lanes 0, 1, 2 of the quad take the beta cases 1, 2, 3 (lane 3 repeats case 3).  The starting points are three different routines
        // -- the wave walks them one after the other, as a single lane did -- but everything after them is the same code for every case
        // and now runs ONCE for the three: two of the three tails (a third of the solve) are gone.
        const int c = q < 2 ? q + 1 : 3;
        double betas[4], R[3][3], t[3];
        betas_approx_any(l, rho, c, betas);
#ifdef P2P_PNP_TIMING
        tk[nk++] = clock64();
#endif
        const double err = finish_case(betas, R, t);
#ifdef P2P_PNP_TIMING
        tk[nk++] = clock64();
#endif
        // OpenCV's rule, in its order: keep case 1; take case 2 if its error is smaller; then case 3 if smaller than the best so far
        const double e1 = quad_bcast<0>(err), e2 = quad_bcast<1>(err), e3 = quad_bcast<2>(err);
        int best = 0;
        double be = e1;
        if (e2 < be) { best = 1; be = e2; }
        if (e3 < be) best = 2;
        #pragma unroll
        for (int i = 0; i < 3; i++) {
            #pragma unroll
            for (int j = 0; j < 3; j++) {
                const double v0 = quad_bcast<0>(R[i][j]), v1 = quad_bcast<1>(R[i][j]), v2 = quad_bcast<2>(R[i][j]);
                Rout[3 * i + j] = best == 0 ? v0 : best == 1 ? v1 : v2;
            }
            const double v0 = quad_bcast<0>(t[i]), v1 = quad_bcast<1>(t[i]), v2 = quad_bcast<2>(t[i]);
            tout[i] = best == 0 ? v0 : best == 1 ? v1 : v2;
        }
    } else {
        double best_err = 0;
        #pragma unroll 1
        for (int c = 1; c <= 3; c++) {
            double betas[4], R[3][3], t[3];
            if (c == 1) betas_approx_1(l, rho, betas);
            else if (c == 2) betas_approx_2(l, rho, betas);
            else betas_approx_3(l, rho, betas);
            const double err = finish_case(betas, R, t);
#ifdef P2P_PNP_TIMING
            tk[nk++] = clock64();
#endif
            if (c == 1 || err < best_err) {
                best_err = err;
                #pragma unroll
                for (int i = 0; i < 3; i++) {
                    #pragma unroll
                    for (int j = 0; j < 3; j++) Rout[3 * i + j] = R[i][j];
                    tout[i] = t[i];
                }
            }
        }
    }
#ifdef P2P_PNP_TIMING
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (QUAD)
            printf("epnp5 cycles: setup %lld  mtm+svd12 %lld  L+rho+betas_approx x3 %lld  gauss_newton %lld  ccs+orientation %lld  error %lld\n", tk[1] - tk[0], tk[2] - tk[1],
                   tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4], tk[6] - tk[5]);
        else
            printf("epnp5 cycles: setup %lld  mtm+svd12 %lld  rest %lld\n", tk[1] - tk[0], tk[2] - tk[1], tk[nk - 1] - tk[2]);
    }
#endif
}

// ---------------------------------------------------------------- workgroup reductions
template <int NV>
__device__ void block_reduce(double (&v)[NV], double* red /* LDS: >= 4*NV doubles */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;          // callers run four waves; any further ones only keep the barriers
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0 && wave < 4) red[wave * NV + k] = x;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; k++) v[k] = red[k] + red[NV + k] + red[2 * NV + k] + red[3 * NV + k];
    __syncthreads();
}

// The same reduction for sums only ONE thread consumes (the 56 Gram sums of the refit go straight to the problem's PnpFit record): the second
// stage runs on thread 0 alone, sum by sum.  block_reduce<56> made every thread form all 56 totals, and the scheduler issued their 224 LDS reads
// together: 448 registers -- pnp_score_kernel held 254 VGPRs + 246 AGPRs, i.e. one 4-wave workgroup per CU.  Same additions in the same order.
template <int NV>
__device__ void block_reduce_store(double (&v)[NV], double* red /* LDS: >= 4*NV doubles */, double* __restrict__ dst)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; k++) {
        double x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if (lane == 0 && wave < 4) red[wave * NV + k] = x;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll 1
        for (int k = 0; k < NV; k++) dst[k] = red[k] + red[NV + k] + red[2 * NV + k] + red[3 * NV + k];
    }
    __syncthreads();
}

__device__ int block_reduce_int(int x, int* red)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if (lane == 0) red[wave] = x;
    __syncthreads();
    const int s = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return s;
}

__device__ int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > kDblMin ? 1. - p : kDblMin;
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < kDblMin) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

// inlier test of OpenCV's PnPRansacCallback::computeError + findInliers
__device__ __forceinline__ bool is_inlier(const double* R, const double* t, const Cam& cam, float X, float Y, float Z,
                                          float iu, float iv, float thr2)
{
    const double Xd = X, Yd = Y, Zd = Z;
    double x = R[0] * Xd + R[1] * Yd + R[2] * Zd + t[0];
    double y = R[3] * Xd + R[4] * Yd + R[5] * Zd + t[1];
    double z = R[6] * Xd + R[7] * Yd + R[8] * Zd + t[2];
    z = z != 0. ? 1. / z : 1.;
    x *= z; y *= z;
    const float pu = (float)(x * cam.fu + cam.uc), pv = (float)(y * cam.fv + cam.vc);
    const float du = iu - pu, dv = iv - pv;
    const float err = du * du + dv * dv;
    return err <= thr2;
}

constexpr int MAX_ITERS = P2P_MAX_RANSAC_ITERATIONS;   // the entry points reject larger requests

// State handed from the scoring kernel to the refit kernels (one record per problem).
struct PnpFit {
    double cws[12];      // control points of the inlier set
    double ci[9];        // inverse of the control-point basis
    double g[56];        // Gram sums (see pass B)
    double a_first[4];   // barycentric coordinates of the first inlier (sign disambiguation)
    double cand[36];     // three (R, t) candidates from the refit solve
    int best, max_good, iters, state;   // state: 0 = refit pending, 1 = result already final, 2 = needs the next hypothesis round
    int niters, pad_[3];                // state 2: RANSAC's iteration bound so far (best / max_good / iters hold the rest of the rule's state)
};

// Kernel 1 of 4 -- hypotheses.  A 16-lane group (round 0: four problems per wave) or a whole wave per problem: its first lane replays the sampler,
// then one lane per minimal set solves the 5-point EPnP and stores the model (R from rvec, t) to
// `hyp`.  A separate kernel because the register footprint of the inlined fp64 solver (it takes the
// whole 512-entry file) must not be imposed on the scoring / refit phases, and so that the models of
// ALL problems are solved in one resident round.
constexpr int SCORE_CHUNK = 8;   // hypotheses whose inlier counts are taken in one pass over the points (pnp_score_kernel)
// Hypotheses are solved lazily in three rounds: RANSAC's adaptive bound stops long before 100 on good data (mean ~12 here), so
// [0, 16) are solved first -- ONE problem per wave, 16 lane quads: the inlined fp64 solver takes the whole 512-entry register
// file, i.e. a resident wave blocks its SIMD for everything else, and a quarter of the waves blocks a quarter of the SIMD time --,
// then [16, 64) and [64, iterations) only for the problems whose scoring ran past what it had (one problem per wave).
constexpr int HYP_ROUND0 = 16, HYP_ROUND1 = 64;
// A workgroup is FOUR such waves (one per SIMD of a CU), each with its own problems: a 256-thread workgroup cannot be placed on a CU
// one of whose SIMDs is held by a 512-register wave, so single-wave workgroups -- which the dispatcher spreads over the chip --
// took a whole CU each away from the generator kernels of the next batch (192 of 256 CUs for the first round of a 256-detection
// batch: the ResNet front running under it was 2.7x slower); packed four to a CU they take a quarter as many.
// MODE 2 (launches of a handful of problems): workgroups of ONE wave, two hypotheses per wave, each on a team of 32 lanes (jacobi12_team).
template <int MODE>
__global__ __launch_bounds__(256, P2P_PNP_HYP_WAVES) void pnp_hypotheses_kernel(const PnpProblem* __restrict__ probs, double* __restrict__ hyp,
                                                            const PnpFit* __restrict__ fits, int n_problems, int iterations, int min_points,
                                                            int h_begin, int h_stop, int ppb, int* __restrict__ act, int round, int wpp)
{
    __shared__ int s_idx_wg[4][MAX_ITERS][5];    // per wave; ppb == 4: rows 16 sub + it (it < 16); ppb == 1: row it
    __shared__ double s_team[MODE == 2 ? 4 * (64 / TEAM_LANES) * TEAM_DOUBLES : 1];
    int (*s_idx)[5] = s_idx_wg[threadIdx.x >> 6];
    const int tid = threadIdx.x & 63;
    const int lanes = 64 / ppb;                  // lanes per problem: FOUR per hypothesis (lane quads share the 12x12 SVD, jacobi12_quad)
    const int hpp = MODE == 2 ? lanes / TEAM_LANES : lanes >> 2;      // hypotheses per pass of a problem's lanes
    const int sub = tid / lanes, lane_in = tid - sub * lanes;
    // Round 0 takes the problems in order (and clears the lists of the later rounds); rounds 1 and 2 take theirs from the list the
    // scoring pass of the round before appended to (act: [count1, count2, list1[n], list2[n]]), so that the few problems still
    // running are packed four to a workgroup: the waves of the other workgroups leave at once.
    // wpp waves share a problem's hypothesis range, 16 hypotheses each (the later rounds: 48 and 36 hypotheses -- walked by one wave
    // they were three passes of 0.5 ms on the critical path of exactly the detections that are slow already)
    const int wave_g = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int prob = (wave_g / wpp) * ppb + sub;
    h_begin += (wave_g % wpp) * hpp;
    h_stop = min(h_stop, h_begin + (wpp > 1 ? hpp : MAX_ITERS));
    if (round == 0) {
        if (blockIdx.x == 0 && threadIdx.x < 2) act[threadIdx.x] = 0;
        if (prob < n_problems) {                 // round 0 is one wave per problem: it clears the problem's inlier counters (pnp_count_kernel adds to them)
            int* cnt = act + 2 + 2 * n_problems + (size_t)prob * MAX_ITERS;
            for (int i = tid; i < MAX_ITERS; i += 64) cnt[i] = 0;
        }
    } else
        prob = prob < act[round - 1] ? act[2 + (round - 1) * n_problems + prob] : n_problems;
    bool active = prob < n_problems && (round == 0 || fits[prob].state == 2);
    PnpProblem pb;
    pb.n = 0; pb.cap = 0; pb.pts = nullptr; pb.mask = nullptr;
    if (active) pb = probs[prob];
    const int n = pb.n;
    active = active && n >= min_points && n >= 5;
    if (iterations > MAX_ITERS) iterations = MAX_ITERS;
    const int n_hyp = n == 5 ? 1 : (iterations > 1 ? iterations : 1);
    // later rounds: nothing past the bound the rule has reached so far is ever read (pnp_count_kernel and pnp_score_kernel stop there)
    const int h_end = min(min(n_hyp, h_stop), round > 0 && active ? fits[prob].niters : MAX_ITERS);
    active = active && h_begin < h_end;
    const int row0 = ppb > 1 ? sub * HYP_ROUND0 : 0;

    // ---- 1. replay the sampler (one lane per problem)
    if (active && lane_in == 0) {
        if (n == 5) {
            for (int i = 0; i < 5; i++) s_idx[row0][i] = i;
        } else {
            Rng rng(~0ULL);
            int cur[5];
            for (int it = 0; it < h_end; it++) {     // the generator state of sample `it` depends on all earlier samples
                for (int i = 0; i < 5;) {
                    int idx_i, j;
                    for (;;) {
                        idx_i = cur[i] = rng.uniform(0, n);
                        for (j = 0; j < i; j++) if (idx_i == cur[j]) break;
                        if (j == i) break;
                    }
                    i++;
                }
                if (ppb == 1 || it < HYP_ROUND0)
                    for (int i = 0; i < 5; i++) s_idx[row0 + it][i] = cur[i];
            }
        }
    }
    __syncthreads();

    // ---- 2. hypotheses: one lane QUAD each; a wave's lane group walks its range in passes of lanes / 4
    const int q4 = lane_in & 3;
    for (int h0 = h_begin; h0 < h_end; h0 += hpp) {
    const int hi = h0 + (MODE == 2 ? lane_in / TEAM_LANES : lane_in >> 2);
    if (active && hi < h_end) {
        const float* PX = pb.pts;
        const float* PY = pb.pts + (size_t)pb.cap;
        const float* PZ = pb.pts + 2 * (size_t)pb.cap;
        const float* PU = pb.pts + 3 * (size_t)pb.cap;
        const float* PV = pb.pts + 4 * (size_t)pb.cap;
        Cam cam{pb.K[0], pb.K[4], pb.K[2], pb.K[5]};
        const double ifx = 1. / pb.K[0], ify = 1. / pb.K[4];
        double pws[15], us[10];
        for (int i = 0; i < 5; i++) {
            const int id = s_idx[row0 + (ppb > 1 ? hi - h_begin : hi)][i];
            pws[3 * i] = PX[id]; pws[3 * i + 1] = PY[id]; pws[3 * i + 2] = PZ[id];
            // undistortPoints (identity distortion) stores float32 normalised coordinates; epnp re-applies fu, uc
            const double xn = (double)(float)(((double)PU[id] - pb.K[2]) * ifx);
            const double yn = (double)(float)(((double)PV[id] - pb.K[5]) * ify);
            us[2 * i] = xn * pb.K[0] + pb.K[2];
            us[2 * i + 1] = yn * pb.K[4] + pb.K[5];
        }
        double R[9], t[3], rvec[3];
        if (MODE == 2)
            epnp5<2>(cam, pws, us, R, t, q4, (lds_double*)(s_team + ((threadIdx.x >> 6) * (64 / TEAM_LANES) + (tid / TEAM_LANES)) * TEAM_DOUBLES), tid & (TEAM_LANES - 1));
        else
            epnp5<1>(cam, pws, us, R, t, q4);
        rodrigues_r2v(R, rvec);          // the model handed to RANSAC is (rvec, tvec)
        rodrigues_v2r(rvec, R);
        if (MODE == 2 ? (tid & (TEAM_LANES - 1)) == 0 : q4 == 0) {
            double* h = hyp + ((size_t)prob * MAX_ITERS + hi) * 12;
            for (int k = 0; k < 9; k++) h[k] = R[k];
            for (int k = 0; k < 3; k++) h[9 + k] = t[k];
        }
    }
    }
}

// Kernel 2a -- inlier counts.  Only the COUNT of a hypothesis enters OpenCV's sequential rule (keep the best, shrink the iteration
// bound), and a count is an integer: any partition of the points and any order of the hypotheses gives the same numbers.  So the
// counts of a round's hypotheses are taken by independent workgroups -- work item = (problem, chunk of SCORE_CHUNK hypotheses, slice of the
// points): a point is loaded once and tested against the chunk's models (scalar registers), four points in flight per thread -- and added to
// the problem's counters; pnp_score_kernel then replays the rule over them.  With detections of 200 - 450 px a candidate carries up
// to 200 000 correspondences: one workgroup per problem walking chunk after chunk (and starting over after every hypothesis round)
// took 9 ms of kernel time per 256-detection step at bbox sides of 40 - 300 px, most of it in a few workgroups on an idle chip.
// Later rounds: the problems on the round's work list, and only chunks below the bound the rule has reached so far.
constexpr int COUNT_NT = 512;
__global__ __launch_bounds__(COUNT_NT) void pnp_count_kernel(const PnpProblem* __restrict__ probs, const double* __restrict__ hyp,
                                                             const PnpFit* __restrict__ fits, int* __restrict__ act, int iterations,
                                                             double reproj_err, int min_points, int h_begin, int n_solved, int nchunks,
                                                             int slices, int round, int n_problems)
{
    __shared__ int s_cnt[SCORE_CHUNK];
    int* counts = act + 2 + 2 * n_problems;
    const int tid = threadIdx.x;
    const int per_prob = nchunks * slices;
    const int n_items = (round == 0 ? n_problems : act[round - 1]) * per_prob;
    if (iterations > MAX_ITERS) iterations = MAX_ITERS;
    const float thr2 = (float)(reproj_err * reproj_err);
    for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int slot = item / per_prob, rem = item - slot * per_prob;
        const int c = rem / slices, sl = rem - c * slices;
        const int prob = __builtin_amdgcn_readfirstlane(round == 0 ? slot : act[2 + (round - 1) * n_problems + slot]);      // (wave-uniform: scalar loads below)
        const PnpProblem& pb = probs[prob];
        const int n = pb.n;
        if (n < min_points || n <= 5) continue;                          // n == 5: the direct solution, nothing to count
        if (sl * 4 * COUNT_NT >= n) continue;
        const int n_hyp = iterations > 1 ? iterations : 1;
        const int n_avail = min(n_hyp, n_solved);
        const int it0 = h_begin + c * SCORE_CHUNK;
        const int hc = __builtin_amdgcn_readfirstlane(min(SCORE_CHUNK, min(n_avail, round == 0 ? n_avail : fits[prob].niters) - it0));
        if (hc <= 0) continue;
        __syncthreads();                                                 // the previous item's counters are no longer read
        // The chunk's models are the same for every lane: read straight from the hypothesis array at wave-uniform addresses they arrive through
        // the scalar cache in SGPRs, and an fp64 VALU instruction takes one of them as its scalar operand.  (Staged in LDS they were hoisted into
        // 192 VGPRs -- 251 in all, two waves per SIMD, or re-read at 4 LDS cycles per broadcast double: 532 -> 358 us per launch at 40 - 300-px
        // boxes, 69 VGPRs.)  Model h: R = M[12 h .. 12 h + 8], t = M[12 h + 9 .. 12 h + 11]; models past hc are never counted.
        const double* __restrict__ M = hyp + ((size_t)prob * MAX_ITERS + it0) * 12;
        if (tid < SCORE_CHUNK) s_cnt[tid] = 0;
        __syncthreads();
        const float* PX = pb.pts;
        const float* PY = pb.pts + (size_t)pb.cap;
        const float* PZ = pb.pts + 2 * (size_t)pb.cap;
        const float* PU = pb.pts + 3 * (size_t)pb.cap;
        const float* PV = pb.pts + 4 * (size_t)pb.cap;
        const Cam cam{pb.K[0], pb.K[4], pb.K[2], pb.K[5]};
        int cnt[SCORE_CHUNK];
#pragma unroll
        for (int h = 0; h < SCORE_CHUNK; ++h) cnt[h] = 0;
        for (int i0 = sl * 4 * COUNT_NT + tid; i0 < n; i0 += slices * 4 * COUNT_NT) {     // four points per trip: their 20 loads are in flight together
            float px[4], py[4], pz[4], pu[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + COUNT_NT * u, n - 1);
                px[u] = PX[i]; py[u] = PY[i]; pz[u] = PZ[i]; pu[u] = PU[i]; pv[u] = PV[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + COUNT_NT * u >= n) break;
#pragma unroll
                for (int h = 0; h < SCORE_CHUNK; ++h)
                    if (h < hc) cnt[h] += is_inlier(M + 12 * h, M + 12 * h + 9, cam, px[u], py[u], pz[u], pu[u], pv[u], thr2) ? 1 : 0;
            }
        }
#pragma unroll
        for (int h = 0; h < SCORE_CHUNK; ++h) {
            int x = cnt[h];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((tid & 63) == 0 && x) atomicAdd(&s_cnt[h], x);
        }
        __syncthreads();
        if (tid < hc && s_cnt[tid]) atomicAdd(&counts[(size_t)prob * MAX_ITERS + it0 + tid], s_cnt[tid]);
    }
}

// Kernel 2b -- OpenCV's sequential rule over the counts (keep the best, shrink the bound: same decisions, same iteration count as a
// hypothesis-by-hypothesis walk), then the reduction passes of the EPnP refit over the inliers of the winning hypothesis
// (centroid / covariance -> control points; Gram sums).  One workgroup (256 threads) per problem; the floating-point sums keep the
// partition they always had (point i belongs to thread i mod 256), so a pose has the bits it always had.
// A problem whose bound asks for hypotheses that are not solved yet parks the rule's state in its PnpFit (state 2), puts itself on the
// next round's work list and resumes there.
constexpr int SCORE_NT = 256;
__global__ __launch_bounds__(SCORE_NT) void pnp_score_kernel(const PnpProblem* __restrict__ probs, const double* __restrict__ hyp,
                                                             PnpResult* __restrict__ results, PnpFit* __restrict__ fits, int iterations,
                                                             double reproj_err, double confidence, int min_points, int h_begin, int n_solved,
                                                             int first, int* __restrict__ act, int round, int n_problems)
{
    PnpFit& fit = fits[blockIdx.x];
    if (!first && fit.state != 2) return;        // later passes: only problems whose scoring ran out of hypotheses
    __shared__ double s_red[4 * 56];
    __shared__ int s_ctl[5];         // niters, best, max_good, iter, 1 = needs the next hypothesis round
    __shared__ double s_fit[24];     // control points + inverse, thread 0 -> all

    const PnpProblem pb = probs[blockIdx.x];
    PnpResult& out = results[blockIdx.x];
    const int tid = threadIdx.x;
    const int n = pb.n;
    const float* PX = pb.pts;
    const float* PY = pb.pts + (size_t)pb.cap;
    const float* PZ = pb.pts + 2 * (size_t)pb.cap;
    const float* PU = pb.pts + 3 * (size_t)pb.cap;
    const float* PV = pb.pts + 4 * (size_t)pb.cap;
    Cam cam{pb.K[0], pb.K[4], pb.K[2], pb.K[5]};
    if (iterations > MAX_ITERS) iterations = MAX_ITERS;

    if (n < min_points || n < 5) {
        if (tid == 0) {
            for (int k = 0; k < 9; k++) out.R[k] = (k % 4 == 0) ? 1. : 0.;
            out.t[0] = out.t[1] = out.t[2] = 0;
            out.n_inliers = -1; out.iters = 0; out.best_iter = -1; out.ok = 0;
            fit.state = 1;
        }
        return;
    }
    const int n_hyp = n == 5 ? 1 : (iterations > 1 ? iterations : 1);
    const int n_avail = min(n_hyp, n_solved);                           // hypotheses solved so far
    const float thr2 = (float)(reproj_err * reproj_err);

    // ---- 3. OpenCV's rule over the counts of hypotheses [h_begin, n_avail)
    if (tid == 0) {
        if (n == 5) { s_ctl[0] = 1; s_ctl[1] = 0; s_ctl[2] = 5; s_ctl[3] = 0; s_ctl[4] = 0; }
        else {
            int niters = n_hyp, best = -1, max_good = 0, it = 0, more = 0;
            if (!first) { niters = fit.niters; best = fit.best; max_good = fit.max_good; it = fit.iters; }
            const int* cnt = act + 2 + 2 * n_problems + (size_t)blockIdx.x * MAX_ITERS;
            for (it = h_begin; it < niters; ++it) {
                if (it >= n_avail) { more = 1; break; }      // (n_avail < n_hyp only) the bound still asks for more: solve the next round
                const int good = cnt[it];
                if (good > (max_good > 4 ? max_good : 4)) {
                    max_good = good;
                    best = it;
                    niters = ransac_update_num_iters(confidence, (double)(n - good) / n, 5, niters);
                }
            }
            // `it` = hypotheses OpenCV would have walked: the loop leaves at it == niters (bound reached) or at n_avail
            s_ctl[0] = niters; s_ctl[1] = best; s_ctl[2] = max_good; s_ctl[3] = it; s_ctl[4] = more;
            if (more) {
                fit.niters = niters; fit.best = best; fit.max_good = max_good; fit.iters = it; fit.state = 2;
                if (round < 2) act[2 + round * n_problems + atomicAdd(&act[round], 1)] = blockIdx.x;      // work list of the next hypothesis round
            }
        }
    }
    __syncthreads();
    if (s_ctl[4]) return;
    const int best = s_ctl[1], max_good = s_ctl[2], iters_run = s_ctl[3];
    if (best < 0 || max_good <= 0) {
        if (tid == 0) {
            for (int k = 0; k < 9; k++) out.R[k] = (k % 4 == 0) ? 1. : 0.;
            out.t[0] = out.t[1] = out.t[2] = 0;
            out.n_inliers = -1; out.iters = iters_run; out.best_iter = -1; out.ok = 0;
            fit.state = 1;
        }
        return;
    }
    const double* hb = hyp + ((size_t)blockIdx.x * MAX_ITERS + best) * 12;
    if (n == 5) {   // solvePnPRansac returns the direct solution when npoints == model_points
        if (tid == 0) {
            for (int k = 0; k < 9; k++) out.R[k] = hb[k];
            for (int k = 0; k < 3; k++) out.t[k] = hb[9 + k];
            out.n_inliers = 5; out.iters = 0; out.best_iter = 0; out.ok = 1;
            fit.state = 1;
        }
        if (pb.mask) for (int i = tid; i < n; i += SCORE_NT) pb.mask[i] = 1;
        return;
    }

    // ---- 4. re-fit EPnP on the inliers of the best hypothesis
    double Rb[9], tb[3];
    for (int k = 0; k < 9; k++) Rb[k] = hb[k];
    for (int k = 0; k < 3; k++) tb[k] = hb[9 + k];
    const int m = max_good;

    // The three refit passes below walk the points in the same per-thread order as before (i = tid, tid + 256, ...: the partial sums
    // and therefore the results are unchanged) but four points per trip, so that their loads are in flight together, and the
    // inlier flags of the winning hypothesis are computed ONCE (pass A) and kept as a per-thread bit mask for the other two passes
    // (up to 64 points per thread = 16 384 correspondences, the 128-px crop; larger problems re-evaluate the test).
    const bool use_mask = n <= 64 * 256;
    unsigned long long inmask = 0;
    // pass A: centroid and covariance of the inlier object points
    {
        double v[3] = {0, 0, 0};
        for (int i0 = tid < 256 ? tid : n, k0 = 0; i0 < n; i0 += 4 * 256, k0 += 4) {
            float px[4], py[4], pz[4], pu[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + 256 * u, n - 1);
                px[u] = PX[i]; py[u] = PY[i]; pz[u] = PZ[i]; pu[u] = PU[i]; pv[u] = PV[i];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + 256 * u;
                if (i >= n) break;
                const bool in = is_inlier(Rb, tb, cam, px[u], py[u], pz[u], pu[u], pv[u], thr2);
                if (pb.mask) pb.mask[i] = in ? 1 : 0;
                if (in) { v[0] += px[u]; v[1] += py[u]; v[2] += pz[u]; if (use_mask) inmask |= 1ull << (k0 + u); }
            }
        }
        block_reduce<3>(v, s_red);
        double c0[3] = {v[0] / m, v[1] / m, v[2] / m};
        double q[6] = {0, 0, 0, 0, 0, 0};
        for (int i0 = tid < 256 ? tid : n, k0 = 0; i0 < n; i0 += 4 * 256, k0 += 4) {
            float px[4], py[4], pz[4], pu[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = min(i0 + 256 * u, n - 1);
                px[u] = PX[i]; py[u] = PY[i]; pz[u] = PZ[i];
                if (!use_mask) { pu[u] = PU[i]; pv[u] = PV[i]; } else { pu[u] = pv[u] = 0.f; }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + 256 * u >= n) break;
                const bool in = use_mask ? ((inmask >> (k0 + u)) & 1ull) != 0 : is_inlier(Rb, tb, cam, px[u], py[u], pz[u], pu[u], pv[u], thr2);
                if (in) {
                    const double dx = px[u] - c0[0], dy = py[u] - c0[1], dz = pz[u] - c0[2];
                    q[0] += dx * dx; q[1] += dx * dy; q[2] += dx * dz; q[3] += dy * dy; q[4] += dy * dz; q[5] += dz * dz;
                }
            }
        }
        block_reduce<6>(q, s_red);
        if (tid == 0) {
            const double ptp[9] = {q[0], q[1], q[2], q[1], q[3], q[4], q[2], q[4], q[5]};
            double cws[4][3], ci[9];
            control_points(c0, ptp, m, cws);
            cc_inverse(cws, ci);
            for (int i = 0; i < 4; i++)
                for (int j = 0; j < 3; j++) s_fit[3 * i + j] = cws[i][j];
            for (int k = 0; k < 9; k++) s_fit[12 + k] = ci[k];
        }
        __syncthreads();
    }
    double cws[4][3], ci[9];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) cws[i][j] = s_fit[3 * i + j];
    for (int k = 0; k < 9; k++) ci[k] = s_fit[12 + k];

    // pass B: Gram sums.  M^T M[(j,p),(k,q)] only needs, over pairs (j<=k) of barycentric coords,
    //   S0 = sum a_j a_k, S1 = sum a_j a_k du, S2 = sum a_j a_k dv, S3 = sum a_j a_k (du^2+dv^2)
    // with du = uc-u, dv = vc-v; plus Sa[j] = sum a_j and A[j][c] = sum a_j (pw_c - pw0_c) for the
    // absolute-orientation step (pc_i is linear in a_i, so no further pass over the points).
    double g[56];
#pragma unroll
    for (int k = 0; k < 56; k++) g[k] = 0;
    for (int i0 = tid < 256 ? tid : n, k0 = 0; i0 < n; i0 += 4 * 256, k0 += 4) {
        float px[4], py[4], pz[4], pu[4], pv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + 256 * u, n - 1);
            px[u] = PX[i]; py[u] = PY[i]; pz[u] = PZ[i]; pu[u] = PU[i]; pv[u] = PV[i];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + 256 * u >= n) break;
            const bool in = use_mask ? ((inmask >> (k0 + u)) & 1ull) != 0 : is_inlier(Rb, tb, cam, px[u], py[u], pz[u], pu[u], pv[u], thr2);
            if (!in) continue;
            const double p[3] = {px[u], py[u], pz[u]};
            double a[4];
            barycentric(ci, cws, p, a);
            const double du = cam.uc - (double)pu[u], dv = cam.vc - (double)pv[u];
            const double dd = du * du + dv * dv;
            int e = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = j; k < 4; k++) {
                    const double ajk = a[j] * a[k];
                    g[e] += ajk; g[10 + e] += ajk * du; g[20 + e] += ajk * dv; g[30 + e] += ajk * dd;
                    ++e;
                }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                g[40 + j] += a[j];
                g[44 + 3 * j] += a[j] * (p[0] - cws[0][0]);
                g[45 + 3 * j] += a[j] * (p[1] - cws[0][1]);
                g[46 + 3 * j] += a[j] * (p[2] - cws[0][2]);
            }
        }
    }
    block_reduce_store<56>(g, s_red, fit.g);

    if (tid == 0) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 3; j++) fit.cws[3 * i + j] = cws[i][j];
        for (int k = 0; k < 9; k++) fit.ci[k] = ci[k];
        // epnp's solve_for_sign looks at the depth of the FIRST inlier's camera-frame point
        int first = 0;
        while (first < n && !is_inlier(Rb, tb, cam, PX[first], PY[first], PZ[first], PU[first], PV[first], thr2)) ++first;
        const double p[3] = {PX[first], PY[first], PZ[first]};
        double a[4];
        barycentric(ci, cws, p, a);
        for (int k = 0; k < 4; k++) fit.a_first[k] = a[k];
        fit.best = best; fit.max_good = max_good; fit.iters = iters_run; fit.state = 0;
    }
}

// element (p, q) of M^T M from the Gram sums of pass B (p, q compile-time after unrolling)
__device__ __forceinline__ double mtm_entry(const double (&g)[56], const Cam& cam, const int p, const int q)
{
    // pair index of (j,k), j <= k, in the order pass B enumerates them
    constexpr int PAIR[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
    const int j = p / 3, rp = p - 3 * j, k = q / 3, rq = q - 3 * k, e = PAIR[j][k];
    const double s0 = g[e], s1 = g[10 + e], s2 = g[20 + e], s3 = g[30 + e];
    if (rp == 0) return rq == 0 ? cam.fu * cam.fu * s0 : rq == 1 ? 0.0 : cam.fu * s1;
    if (rp == 1) return rq == 0 ? 0.0 : rq == 1 ? cam.fv * cam.fv * s0 : cam.fv * s2;
    return rq == 0 ? cam.fu * s1 : rq == 1 ? cam.fv * s2 : s3;
}

// Kernel 3 of 4 -- the refit solve: one LANE per (problem, beta case) -- register-resident 12x12 SVD (computed by all three lanes
// of a problem: same inputs, same bits), then this lane's betas, Gauss-Newton and absolute orientation.  The three cases used to
// run one after the other on one lane; the kernel is a pure latency chain (the tail of a blocking call and of a single detection
// waits for it): 0.62 -> 0.54 ms for the 768 problems of a 256-detection batch, 0.45 -> 0.39 ms for the three of one detection
// (the 12x12 SVD dominates).
// TEAM (launches of a handful of problems): a team of 32 lanes per problem -- the SVD on six of its quads (jacobi12_team), the beta
// cases on quads 0..2 -- in workgroups of one wave.
template <bool TEAM>
__global__ __launch_bounds__(256, P2P_PNP_SOLVE_WAVES) void pnp_fit_solve_kernel(const PnpProblem* __restrict__ probs, PnpFit* __restrict__ fits, int n_problems)
{
    __shared__ double s_team[TEAM ? 4 * (64 / TEAM_LANES) * TEAM_DOUBLES : 1];
    const int li = blockIdx.x * blockDim.x + threadIdx.x;      // four waves per workgroup = one CU, see pnp_hypotheses_kernel
    const int q4 = li & 3, qi = li >> 2;                // a lane QUAD per (problem, beta case): the quad shares the 12x12 SVD
    const int tl = li & (TEAM_LANES - 1);
    const int pi = TEAM ? li / TEAM_LANES : qi / 3, c = TEAM ? min(tl >> 2, 2) : qi - pi * 3;
    if (pi >= n_problems) return;
    PnpFit& fit = fits[pi];
    if (fit.state != 0) return;
    const PnpProblem& pb = probs[pi];
    Cam cam{pb.K[0], pb.K[4], pb.K[2], pb.K[5]};
    double g[56], cws[4][3];
#pragma unroll
    for (int k = 0; k < 56; k++) g[k] = fit.g[k];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) cws[i][j] = fit.cws[3 * i + j];
    const int m = fit.max_good;
    double u8[48];          // rows 8..11 of the left singular vectors of M^T M
    bool solved = false;
    {
        double A[36], w12[12];
#pragma unroll
        for (int p = 0; p < 12; p++)
#pragma unroll
            for (int e = 0; e < 3; e++) {
                const double v0 = mtm_entry(g, cam, p, e), v1 = mtm_entry(g, cam, p, 3 + e), v2 = mtm_entry(g, cam, p, 6 + e), v3 = mtm_entry(g, cam, p, 9 + e);
                A[p * 3 + e] = q4 == 0 ? v0 : q4 == 1 ? v1 : q4 == 2 ? v2 : v3;
            }
        if (TEAM) {
            lds_double* sm = (lds_double*)(s_team + (threadIdx.x / TEAM_LANES) * TEAM_DOUBLES);
            if (tl < 4) {
#pragma unroll
                for (int p = 0; p < 12; p++)
#pragma unroll
                    for (int e = 0; e < 3; e++) sm[p * 12 + 3 * q4 + e] = A[p * 3 + e];
            }
            wave_lds_sync();
            solved = jacobi12_team(sm, tl, true, u8);
        } else {
            solved = jacobi12_quad(A, w12);
            if (solved) gather_rows8(A, u8);
        }
    }
    if (!solved) {          // a zero singular value: the single-lane routine has OpenCV's random fill-in
        double mtm[144], w12[12];
#pragma unroll
        for (int p = 0; p < 12; p++)
#pragma unroll
            for (int k = 0; k < 12; k++) mtm[p * 12 + k] = mtm_entry(g, cam, p, k);
        jacobi_svd12_rolled(mtm, w12);
#pragma unroll
        for (int k = 0; k < 48; k++) u8[k] = mtm[96 + k];
    }
    const double* ut = u8;
    double l[60], rho[6];
    compute_L_6x10(ut, l);
    compute_rho(cws, rho);
    {
        double betas[4];
        betas_approx_any(l, rho, c + 1, betas);
        gauss_newton(l, rho, betas);
        double ccs[4][3];
        compute_ccs(betas, ut, ccs);
        const double z = fit.a_first[0] * ccs[0][2] + fit.a_first[1] * ccs[1][2] + fit.a_first[2] * ccs[2][2] + fit.a_first[3] * ccs[3][2];
        if (z < 0.0) {
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
        }
        double pc0[3], abt[9];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            pc0[j] = (g[40] * ccs[0][j] + g[41] * ccs[1][j] + g[42] * ccs[2][j] + g[43] * ccs[3][j]) / m;
#pragma unroll
            for (int k = 0; k < 3; k++)
                abt[3 * j + k] = ccs[0][j] * g[44 + k] + ccs[1][j] * g[47 + k] + ccs[2][j] * g[50 + k] + ccs[3][j] * g[53 + k];
        }
        double R[3][3], t[3];
        orientation(abt, pc0, cws[0], R, t);
        if (TEAM ? (tl < 12 && q4 == 0) : q4 == 0) {
#pragma unroll
            for (int i = 0; i < 3; i++) {
#pragma unroll
                for (int j = 0; j < 3; j++) fit.cand[12 * c + 3 * i + j] = R[i][j];
                fit.cand[12 * c + 9 + i] = t[i];
            }
        }
    }
}

// Kernel 4 of 4 -- pick among the three candidates by mean reprojection error over the inliers and
// write the result (through the same rvec round trip solvePnP + cv2.Rodrigues perform).
__global__ __launch_bounds__(256) void pnp_fit_select_kernel(const PnpProblem* __restrict__ probs, const double* __restrict__ hyp,
                                                             const PnpFit* __restrict__ fits, PnpResult* __restrict__ results,
                                                             double reproj_err)
{
    __shared__ double s_red[4 * 3];
    const PnpFit& fit = fits[blockIdx.x];
    if (fit.state != 0) return;
    const PnpProblem pb = probs[blockIdx.x];
    PnpResult& out = results[blockIdx.x];
    const int tid = threadIdx.x;
    const int n = pb.n;
    const float* PX = pb.pts;
    const float* PY = pb.pts + (size_t)pb.cap;
    const float* PZ = pb.pts + 2 * (size_t)pb.cap;
    const float* PU = pb.pts + 3 * (size_t)pb.cap;
    const float* PV = pb.pts + 4 * (size_t)pb.cap;
    Cam cam{pb.K[0], pb.K[4], pb.K[2], pb.K[5]};
    const float thr2 = (float)(reproj_err * reproj_err);
    double Rb[9], tb[3];
    const double* hb = hyp + ((size_t)blockIdx.x * MAX_ITERS + fit.best) * 12;
    for (int k = 0; k < 9; k++) Rb[k] = hb[k];
    for (int k = 0; k < 3; k++) tb[k] = hb[9 + k];
    double Rc[3][9], tc[3][3];
    for (int c = 0; c < 3; c++) {
        for (int k = 0; k < 9; k++) Rc[c][k] = fit.cand[12 * c + k];
        for (int k = 0; k < 3; k++) tc[c][k] = fit.cand[12 * c + 9 + k];
    }
    double e3[3] = {0, 0, 0};
    // a thread walks its points i = tid, tid + 256, ... in that order (the partial sums keep their bits), four of them per trip so that their
    // loads are in flight together (one at a time the loop was a chain of load latencies: 0.93 ms per launch at 250 - 420-px boxes)
    for (int i0 = tid; i0 < n; i0 += 4 * 256) {
        float px[4], py[4], pz[4], pu[4], pv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = min(i0 + 256 * q, n - 1);
            px[q] = PX[i]; py[q] = PY[i]; pz[q] = PZ[i]; pu[q] = PU[i]; pv[q] = PV[i];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (i0 + 256 * q >= n) break;
            if (!is_inlier(Rb, tb, cam, px[q], py[q], pz[q], pu[q], pv[q], thr2)) continue;
            const double p[3] = {px[q], py[q], pz[q]};
            const double u = pu[q], v = pv[q];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double Xc = dot3(Rc[c], p) + tc[c][0], Yc = dot3(Rc[c] + 3, p) + tc[c][1];
                const double inv_Zc = 1.0 / (dot3(Rc[c] + 6, p) + tc[c][2]);
                const double ue = cam.uc + cam.fu * Xc * inv_Zc, ve = cam.vc + cam.fv * Yc * inv_Zc;
                e3[c] += sqrt_cr((u - ue) * (u - ue) + (v - ve) * (v - ve));
            }
        }
    }
    block_reduce<3>(e3, s_red);
    if (tid == 0) {
        int N = 0;
        if (e3[1] < e3[0]) N = 1;
        if (e3[2] < e3[N]) N = 2;
        double rvec[3], R[9], RN[9], tN[3];      // (selected by value: a run-time index would put Rc / tc into scratch memory for the loop above too)
        for (int k = 0; k < 9; k++) RN[k] = N == 0 ? Rc[0][k] : N == 1 ? Rc[1][k] : Rc[2][k];
        for (int k = 0; k < 3; k++) tN[k] = N == 0 ? tc[0][k] : N == 1 ? tc[1][k] : tc[2][k];
        rodrigues_r2v(RN, rvec);     // solvePnP returns rvec; the caller converts back (recognition.py:223)
        rodrigues_v2r(rvec, R);
        for (int k = 0; k < 9; k++) out.R[k] = R[k];
        for (int k = 0; k < 3; k++) out.t[k] = tN[k];
        out.n_inliers = fit.max_good; out.iters = fit.iters; out.best_iter = fit.best; out.ok = 1;
    }
}

}  // namespace pnp

size_t pnp_workspace_bytes(int n_problems)
{
    // hypothesis models, refit records, [count1, count2, list1[n], list2[n]] work lists, inlier counters [n][MAX_ITERS]
    return (size_t)n_problems * (pnp::MAX_ITERS * 12 * sizeof(double) + sizeof(pnp::PnpFit)) +
           (2 + 2 * (size_t)n_problems + (size_t)n_problems * pnp::MAX_ITERS) * sizeof(int) + 16;
}

hipError_t launch_pnp_ransac(const PnpProblem* probs, PnpResult* results, int n_problems, int iterations,
                             double reproj_err, double confidence, int min_points, int max_points, double* workspace, hipStream_t s)
{
    if (n_problems <= 0) return hipSuccess;
    pnp::PnpFit* fits = reinterpret_cast<pnp::PnpFit*>(workspace + (size_t)n_problems * pnp::MAX_ITERS * 12);
    int* act = reinterpret_cast<int*>(((uintptr_t)(fits + n_problems) + 15) & ~(uintptr_t)15);      // work lists of hypothesis rounds 1 and 2
    hipError_t e;
    // round 0: hypotheses [0, 16), four problems per wave; later rounds only for problems whose scoring ran past what it had --
    // everything else leaves those launches at once
    // A handful of problems (one detection at a time: the reference's own call shape) solves [0, 48) up front, three waves per problem side by
    // side: a round is one EPnP solve deep (0.42 ms of dependent fp64) wherever it runs, the chip is idle, and the problem that needs more than
    // 16 hypotheses no longer waits for a second round (0.46 ms of a 2.4-ms call: p90 of the single call 2.88 -> 2.47 ms, mean 2.56 -> 2.43,
    // median unchanged; solving [0, 64) up front costs the median 0.03 ms).  Which hypotheses exist when does not change what the rule reads:
    // same counts in the same order, same poses.
    static const int eager_max = dev_env("P2P_PNP_EAGER_MAX") ? atoi(dev_env("P2P_PNP_EAGER_MAX")) : 48;      // development builds: 0 = rounds of 16 / 48 / 36 always
    static const int eager_stop = dev_env("P2P_PNP_EAGER_STOP") ? atoi(dev_env("P2P_PNP_EAGER_STOP")) : 48;   // development builds: 32 / 48 / 64
    const bool eager = n_problems <= eager_max;
    // the same handful of problems: every 12x12 SVD on a team of 32 lanes (jacobi12_team) -- same bits, a third of the dependent steps
    static const bool team_ok = !(dev_env("P2P_PNP_TEAM") && atoi(dev_env("P2P_PNP_TEAM")) == 0);      // development builds: 0 = the quad form always
    const bool team = eager && team_ok;
    // (eager: everything past the first 48 in ONE further round -- the rare problem that needs it finds the chip idle anyway, and the common
    // one is spared three launches that only find out there is nothing to do: 14 us of a 1.6-ms call)
    const int stops[3] = {eager ? eager_stop : pnp::HYP_ROUND0, eager ? pnp::MAX_ITERS : pnp::HYP_ROUND1, pnp::MAX_ITERS};
    int h_begin = 0;
    for (int r = 0; r < 3; ++r) {
        const int ppb = 1;          // a wave per problem: 16 hypotheses x 4 lanes
        const int per_wave = team ? 64 / pnp::TEAM_LANES : pnp::HYP_ROUND0;
        const int wpp = (std::min(stops[r], iterations) - h_begin + per_wave - 1) / per_wave;      // waves per problem: 16 hypotheses each (team form: 2)
        {
            ProfScope ps(12, s);
            if (team)
                hipLaunchKernelGGL(pnp::pnp_hypotheses_kernel<2>, dim3(n_problems * wpp), dim3(64), 0, s, probs, workspace, fits, n_problems, iterations,
                                   min_points, h_begin, stops[r], ppb, act, r, wpp);
            else
                hipLaunchKernelGGL(pnp::pnp_hypotheses_kernel<1>, dim3((n_problems * wpp + 3) / 4), dim3(256), 0, s, probs, workspace, fits, n_problems, iterations,
                                   min_points, h_begin, stops[r], ppb, act, r, wpp);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // counts of the round's hypotheses: work items (problem, chunk of 8 hypotheses, slice of the points) walked by a bounded grid.
        // Slices: 16 384 points each (the 128-px crop), more of them when the launch would not fill the chip otherwise
        const int h_end = std::min(stops[r], std::min(iterations, pnp::MAX_ITERS));
        const int nchunks = std::max(1, (h_end - h_begin + pnp::SCORE_CHUNK - 1) / pnp::SCORE_CHUNK);
        const int per_trip = 4 * pnp::COUNT_NT;
        int slices = std::max((max_points + 8 * per_trip - 1) / (8 * per_trip), 512 / std::max(1, n_problems * nchunks));
        slices = std::max(1, std::min(slices, (max_points + per_trip - 1) / per_trip));
        const long long items = (long long)n_problems * nchunks * slices;
        {
            ProfScope ps(13, s);
            hipLaunchKernelGGL(pnp::pnp_count_kernel, dim3((unsigned)std::min<long long>(items, 8192)), dim3(pnp::COUNT_NT), 0, s, probs, workspace, fits, act,
                               iterations, reproj_err, min_points, h_begin, stops[r], nchunks, slices, r, n_problems);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        {
            ProfScope ps(14, s);
            hipLaunchKernelGGL(pnp::pnp_score_kernel, dim3(n_problems), dim3(pnp::SCORE_NT), 0, s, probs, workspace, results, fits, iterations,
                               reproj_err, confidence, min_points, h_begin, stops[r], r == 0 ? 1 : 0, act, r, n_problems);
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        h_begin = stops[r];
        if (iterations <= h_begin) break;
    }
    ProfScope ps(15, s);
    if (team)
        hipLaunchKernelGGL(pnp::pnp_fit_solve_kernel<true>, dim3((n_problems + 1) / 2), dim3(64), 0, s, probs, fits, n_problems);
    else
        hipLaunchKernelGGL(pnp::pnp_fit_solve_kernel<false>, dim3((12 * n_problems + 255) / 256), dim3(256), 0, s, probs, fits, n_problems);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(pnp::pnp_fit_select_kernel, dim3(n_problems), dim3(256), 0, s, probs, workspace, fits, results, reproj_err);
    return hipGetLastError();
}

}  // namespace p2p
