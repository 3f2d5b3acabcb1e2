/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of what the reference obtains from
 *     cv2.solvePnPRansac(obj, img, camK, None, flags=cv2.SOLVEPNP_EPNP,
 *                        reprojectionError=5, iterationsCount=100)   + cv2.Rodrigues
 * at pix2pose_model/recognition.py:216-223.  The algorithm lives in an un-vendored
 * dependency, opencv-python==3.4.2.17 (requirements.txt:3): calib3d solvepnp.cpp
 * (solvePnPRansac, PnPRansacCallback), ptsetreg.cpp (RANSACPointSetRegistrator::run,
 * getSubset, RANSACUpdateNumIters), epnp.cpp (Lepetit/Moreno-Noguer/Fua EPnP), core lapack.cpp
 * (one-sided Jacobi SVD, SVD back-substitution), core rand.cpp (cv::RNG multiply-with-carry).
 * This file restates those published algorithms, in double precision, with the same
 * sequencing (fixed RNG seed per call, 5-point minimal sets, adaptive iteration count,
 * final EPnP re-fit on the inlier set, float32 storage of the RANSAC point sets and of the
 * projected points).
 *
 * PARITY UNPINNED: OpenCV is not installed here and the reference holds no golden vectors, so
 * this restatement is pinned only by known-answer tests (synthetic pose -> project ->
 * recover) and an independent numpy EPnP in tests/test_oracle_pnp.py.
 */
/* The 5-point systems are rank deficient (2-D null space), so which basis a solver lands on is
 * decided by rounding.  Keep this file at strict sequential IEEE semantics: gcc's SLP vectoriser
 * (-O3) re-associates some of the straight-line sums below and changes hypotheses. */
#pragma GCC optimize("O2", "no-tree-vectorize", "no-tree-slp-vectorize")
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ cv::RNG (rand.cpp) */
typedef struct { uint64_t state; } cv_rng;
static void rng_init(cv_rng* r, uint64_t s) { r->state = s ? s : 0xffffffffULL; }
static unsigned rng_next(cv_rng* r)
{
    r->state = (uint64_t)(unsigned)r->state * 4164903690U + (unsigned)(r->state >> 32);
    return (unsigned)r->state;
}
static int rng_uniform(cv_rng* r, int a, int b) { return a == b ? a : (int)(rng_next(r) % (unsigned)(b - a) + a); }

/* exported for the replay-vector test */
void p2po_rng_sequence(uint64_t seed, int n, unsigned* out)
{
    cv_rng r;
    rng_init(&r, seed);
    for (int i = 0; i < n; ++i) out[i] = rng_next(&r);
}

/* --------------------------------------------------- one-sided Jacobi SVD (lapack.cpp) */
/* At: n rows of length m (= columns of A, A is m x n, m >= n).  On exit At rows are the left
 * singular vectors (U^T), W the singular values (descending), Vt (n x n) the right ones. */
static void jacobi_svd(double* At, int astep, double* W, double* Vt, int vstep, int m, int n, int n1)
{
    const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
    int i, j, k, iter, max_iter = m > 30 ? m : 30;
    double c, s, sd;
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) { double t = At[i * astep + k]; sd += t * t; }
        W[i] = sd;
        if (Vt) { for (k = 0; k < n; k++) Vt[i * vstep + k] = 0; Vt[i * vstep + i] = 1; }
    }
    for (iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At + i * astep, *Aj = At + j * astep;
                double a = W[i], p = 0, b = W[j];
                for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                /* OpenCV calls hypot(); written out so every libm gives the same bits (a minimal-set
                 * system has a 2-D null space whose basis is decided by rounding) */
                double beta = a - b, gamma = sqrt(p * p + beta * beta);
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = 1;
                if (Vt) {
                    double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
                    for (k = 0; k < n; k++) {
                        double t0 = c * Vi[k] + s * Vj[k];
                        double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0; Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }
    for (i = 0; i < n; i++) {
        for (k = 0, sd = 0; k < m; k++) { double t = At[i * astep + k]; sd += t * t; }
        W[i] = sqrt(sd);
    }
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            if (Vt) {
                for (k = 0; k < m; k++) { t = At[i * astep + k]; At[i * astep + k] = At[j * astep + k]; At[j * astep + k] = t; }
                for (k = 0; k < n; k++) { t = Vt[i * vstep + k]; Vt[i * vstep + k] = Vt[j * vstep + k]; Vt[j * vstep + k] = t; }
            }
        }
    }
    if (!Vt) return;
    cv_rng rng;
    rng_init(&rng, 0x12345678);
    for (i = 0; i < n1; i++) {
        sd = i < n ? W[i] : 0;
        for (int ii = 0; ii < 100 && sd <= minval; ii++) {
            /* zero singular value: random vector, orthogonalised against the previous rows */
            const double val0 = 1. / m;
            for (k = 0; k < m; k++) At[i * astep + k] = (rng_next(&rng) & 256) != 0 ? val0 : -val0;
            for (iter = 0; iter < 2; iter++)
                for (j = 0; j < i; j++) {
                    sd = 0;
                    for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
                    double asum = 0;
                    for (k = 0; k < m; k++) {
                        double t = At[i * astep + k] - sd * At[j * astep + k];
                        At[i * astep + k] = t;
                        asum += fabs(t);
                    }
                    asum = asum > eps * 100 ? 1 / asum : 0;
                    for (k = 0; k < m; k++) At[i * astep + k] *= asum;
                }
            sd = 0;
            for (k = 0; k < m; k++) { double t = At[i * astep + k]; sd += t * t; }
            sd = sqrt(sd);
        }
        s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++) At[i * astep + k] *= s;
    }
}

/* SVD of row-major A (m x n, m >= n, n <= 12): w[n], ut (n x m: rows = left vectors), vt (n x n). */
static void svd_mn(const double* A, int m, int n, double* w, double* ut, double* vt)
{
    for (int i = 0; i < n; i++)
        for (int k = 0; k < m; k++) ut[i * m + k] = A[k * n + i];   /* transpose: rows of At = columns of A */
    jacobi_svd(ut, m, w, vt, n, m, n, n);
}

/* x = V diag(1/w) U^T b with OpenCV's SVBkSb threshold (sum(w) * 2*DBL_EPSILON) */
static void svd_backsubst(const double* w, const double* ut, const double* vt, int m, int n, const double* b, double* x)
{
    double thr = 0;
    for (int i = 0; i < n; i++) thr += w[i];
    thr *= DBL_EPSILON * 2;
    for (int j = 0; j < n; j++) x[j] = 0;
    for (int i = 0; i < n; i++) {
        double wi = w[i];
        if (fabs(wi) <= thr) continue;
        wi = 1 / wi;
        double s = 0;
        for (int k = 0; k < m; k++) s += ut[i * m + k] * b[k];
        s *= wi;
        for (int j = 0; j < n; j++) x[j] += s * vt[i * n + j];
    }
}

/* cvSolve(A, b, x, CV_SVD) for m x n, m >= n <= 6 */
static void solve_svd(const double* A, int m, int n, const double* b, double* x)
{
    double w[12], ut[12 * 12], vt[12 * 12];
    svd_mn(A, m, n, w, ut, vt);
    svd_backsubst(w, ut, vt, m, n, b, x);
}

/* --------------------------------------------------------------------- Rodrigues */
static void rodrigues_r2v(const double* R, double* r)
{
    /* cvRodrigues2 3x3 -> 3x1: orthogonalise with an SVD first, then axis-angle */
    double w[3], ut[9], vt[9], Rn[9];
    svd_mn(R, 3, 3, w, ut, vt);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;   /* R = U * Vt,  U[i][k] = ut[k][i] */
            for (int k = 0; k < 3; k++) s += ut[k * 3 + i] * vt[k * 3 + j];
            Rn[i * 3 + j] = s;
        }
    double rx = Rn[7] - Rn[5], ry = Rn[2] - Rn[6], rz = Rn[3] - Rn[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (Rn[0] + Rn[4] + Rn[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { r[0] = r[1] = r[2] = 0; return; }
        double t;
        t = (Rn[0] + 1) * 0.5; rx = sqrt(t > 0. ? t : 0.);
        t = (Rn[4] + 1) * 0.5; ry = sqrt(t > 0. ? t : 0.) * (Rn[1] < 0 ? -1. : 1.);
        t = (Rn[8] + 1) * 0.5; rz = sqrt(t > 0. ? t : 0.) * (Rn[2] < 0 ? -1. : 1.);
        if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (Rn[5] > 0) != (ry * rz > 0)) rz = -rz;
        theta /= sqrt(rx * rx + ry * ry + rz * rz);
        r[0] = rx * theta; r[1] = ry * theta; r[2] = rz * theta;
        return;
    }
    double vth = 1 / (2 * s);
    vth *= theta;
    r[0] = rx * vth; r[1] = ry * vth; r[2] = rz * vth;
}

void p2po_rodrigues_v2r(const double* r, double* R)
{
    /* R = cos(theta) I + (1-cos(theta)) r r^T + sin(theta) [r]x */
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < DBL_EPSILON) {
        memset(R, 0, 9 * sizeof(double));
        R[0] = R[4] = R[8] = 1;
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, it = theta ? 1. / theta : 0.;
    double x = r[0] * it, y = r[1] * it, z = r[2] * it;
    double rrt[9] = {x * x, x * y, x * z, x * y, y * y, y * z, x * z, y * z, z * z};
    double rx[9] = {0, -z, y, z, 0, -x, -y, x, 0};
    for (int k = 0; k < 9; k++) R[k] = c * (k % 4 == 0 ? 1. : 0.) + c1 * rrt[k] + s * rx[k];
}

/* --------------------------------------------------------------------------- EPnP */
typedef struct {
    double uc, vc, fu, fv;
    int n;
    double *pws, *us, *alphas, *pcs;   /* n*3, n*2, n*4, n*3 */
    double cws[4][3], ccs[4][3];
} epnp_t;

static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dist2(const double* a, const double* b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

static void choose_control_points(epnp_t* e)
{
    const int n = e->n;
    e->cws[0][0] = e->cws[0][1] = e->cws[0][2] = 0;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) e->cws[0][j] += e->pws[3 * i + j];
    for (int j = 0; j < 3; j++) e->cws[0][j] /= n;
    double ptp[9] = {0};
    for (int i = 0; i < n; i++) {
        double d[3];
        for (int j = 0; j < 3; j++) d[j] = e->pws[3 * i + j] - e->cws[0][j];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) ptp[a * 3 + b] += d[a] * d[b];
    }
    double dc[3], uct[9], vt[9];
    svd_mn(ptp, 3, 3, dc, uct, vt);
    for (int i = 1; i < 4; i++) {
        double k = sqrt(dc[i - 1] / n);
        for (int j = 0; j < 3; j++) e->cws[i][j] = e->cws[0][j] + k * uct[3 * (i - 1) + j];
    }
}

static void compute_barycentric_coordinates(epnp_t* e)
{
    double cc[9], w[3], ut[9], vt[9], ci[9];
    for (int i = 0; i < 3; i++)
        for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e->cws[j][i] - e->cws[0][i];
    /* cvInvert(CV_SVD): inv = V diag(1/w) U^T, column by column through back-substitution */
    svd_mn(cc, 3, 3, w, ut, vt);
    for (int col = 0; col < 3; col++) {
        double b[3] = {0, 0, 0}, x[3];
        b[col] = 1;
        svd_backsubst(w, ut, vt, 3, 3, b, x);
        for (int r = 0; r < 3; r++) ci[3 * r + col] = x[r];
    }
    for (int i = 0; i < e->n; i++) {
        const double* pi = e->pws + 3 * i;
        double* a = e->alphas + 4 * i;
        for (int j = 0; j < 3; j++)
            a[1 + j] = ci[3 * j] * (pi[0] - e->cws[0][0]) + ci[3 * j + 1] * (pi[1] - e->cws[0][1]) +
                       ci[3 * j + 2] * (pi[2] - e->cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
}

static void compute_ccs(epnp_t* e, const double* betas, const double* ut)
{
    for (int i = 0; i < 4; i++) e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0.0;
    for (int i = 0; i < 4; i++) {
        const double* v = ut + 12 * (11 - i);
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 3; k++) e->ccs[j][k] += betas[i] * v[3 * j + k];
    }
}

static void compute_pcs(epnp_t* e)
{
    for (int i = 0; i < e->n; i++) {
        const double* a = e->alphas + 4 * i;
        double* pc = e->pcs + 3 * i;
        for (int j = 0; j < 3; j++)
            pc[j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] + a[3] * e->ccs[3][j];
    }
}

static void solve_for_sign(epnp_t* e)
{
    if (e->pcs[2] < 0.0) {
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 3; j++) e->ccs[i][j] = -e->ccs[i][j];
        for (int i = 0; i < e->n; i++) {
            e->pcs[3 * i] = -e->pcs[3 * i];
            e->pcs[3 * i + 1] = -e->pcs[3 * i + 1];
            e->pcs[3 * i + 2] = -e->pcs[3 * i + 2];
        }
    }
}

static void estimate_R_and_t(epnp_t* e, double R[3][3], double t[3])
{
    const int n = e->n;
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) { pc0[j] += e->pcs[3 * i + j]; pw0[j] += e->pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0};
    for (int i = 0; i < n; i++) {
        const double* pc = e->pcs + 3 * i;
        const double* pw = e->pws + 3 * i;
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    double d[3], ut[9], vt[9];
    svd_mn(abt, 3, 3, d, ut, vt);
    /* R = U V^T; U[i][k] = ut[k][i], V[j][k] = vt[k][j] */
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
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

static double reprojection_error(const epnp_t* e, double R[3][3], const double t[3])
{
    double sum2 = 0.0;
    for (int i = 0; i < e->n; i++) {
        const double* pw = e->pws + 3 * i;
        double Xc = dot3(R[0], pw) + t[0];
        double Yc = dot3(R[1], pw) + t[1];
        double inv_Zc = 1.0 / (dot3(R[2], pw) + t[2]);
        double ue = e->uc + e->fu * Xc * inv_Zc;
        double ve = e->vc + e->fv * Yc * inv_Zc;
        double u = e->us[2 * i], v = e->us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / e->n;
}

static double compute_R_and_t(epnp_t* e, const double* ut, const double* betas, double R[3][3], double t[3])
{
    compute_ccs(e, betas, ut);
    compute_pcs(e);
    solve_for_sign(e);
    estimate_R_and_t(e, R, t);
    return reprojection_error(e, R, t);
}

static void compute_L_6x10(const double* ut, double* l)
{
    const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
        int a = 0, b = 1;
        for (int j = 0; j < 6; j++) {
            dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
            dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
            dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
            b++;
            if (b > 3) { a++; b = a + 1; }
        }
    }
    for (int i = 0; i < 6; i++) {
        double* row = l + 10 * i;
        row[0] = dot3(dv[0][i], dv[0][i]);
        row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
        row[2] = dot3(dv[1][i], dv[1][i]);
        row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
        row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
        row[5] = dot3(dv[2][i], dv[2][i]);
        row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
        row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
        row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
        row[9] = dot3(dv[3][i], dv[3][i]);
    }
}

static void compute_rho(const epnp_t* e, double* rho)
{
    rho[0] = dist2(e->cws[0], e->cws[1]);
    rho[1] = dist2(e->cws[0], e->cws[2]);
    rho[2] = dist2(e->cws[0], e->cws[3]);
    rho[3] = dist2(e->cws[1], e->cws[2]);
    rho[4] = dist2(e->cws[1], e->cws[3]);
    rho[5] = dist2(e->cws[2], e->cws[3]);
}

/* betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44]; approx_1 = [B11 B12 B13 B14] */
static void find_betas_approx_1(const double* l, const double* rho, double* betas)
{
    double l4[24], b4[4];
    for (int i = 0; i < 6; i++) {
        l4[4 * i] = l[10 * i]; l4[4 * i + 1] = l[10 * i + 1]; l4[4 * i + 2] = l[10 * i + 3]; l4[4 * i + 3] = l[10 * i + 6];
    }
    solve_svd(l4, 6, 4, rho, b4);
    if (b4[0] < 0) {
        betas[0] = sqrt(-b4[0]); betas[1] = -b4[1] / betas[0]; betas[2] = -b4[2] / betas[0]; betas[3] = -b4[3] / betas[0];
    } else {
        betas[0] = sqrt(b4[0]); betas[1] = b4[1] / betas[0]; betas[2] = b4[2] / betas[0]; betas[3] = b4[3] / betas[0];
    }
}

/* approx_2 = [B11 B12 B22] */
static void find_betas_approx_2(const double* l, const double* rho, double* betas)
{
    double l3[18], b3[3];
    for (int i = 0; i < 6; i++) { l3[3 * i] = l[10 * i]; l3[3 * i + 1] = l[10 * i + 1]; l3[3 * i + 2] = l[10 * i + 2]; }
    solve_svd(l3, 6, 3, rho, b3);
    if (b3[0] < 0) {
        betas[0] = sqrt(-b3[0]);
        betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0;
    } else {
        betas[0] = sqrt(b3[0]);
        betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0;
    }
    if (b3[1] < 0) betas[0] = -betas[0];
    betas[2] = 0.0;
    betas[3] = 0.0;
}

/* approx_3 = [B11 B12 B22 B13 B23] */
static void find_betas_approx_3(const double* l, const double* rho, double* betas)
{
    double l5[30], b5[5];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 5; j++) l5[5 * i + j] = l[10 * i + j];
    solve_svd(l5, 6, 5, rho, b5);
    if (b5[0] < 0) {
        betas[0] = sqrt(-b5[0]);
        betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0;
    } else {
        betas[0] = sqrt(b5[0]);
        betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0;
    }
    if (b5[1] < 0) betas[0] = -betas[0];
    betas[2] = b5[3] / betas[0];
    betas[3] = 0.0;
}

/* Householder least squares as in epnp.cpp qr_solve (A is 6x4, destroyed) */
static void qr_solve(double* A, double* b, double* X)
{
    const int nr = 6, nc = 4;
    double A1[4], A2[4];
    double* ppAkk = A;
    for (int k = 0; k < nc; k++) {
        double *ppAik1 = ppAkk, eta = fabs(*ppAik1);
        for (int i = k + 1; i < nr; i++) {
            double elt = fabs(*ppAik1);
            if (eta < elt) eta = elt;
            ppAik1 += nc;
        }
        if (eta == 0) {
            A1[k] = A2[k] = 0.0;
            return;   /* singular: X keeps its previous content (zeros here) */
        } else {
            double *ppAik2 = ppAkk, sum2 = 0.0, inv_eta = 1. / eta;
            for (int i = k; i < nr; i++) {
                *ppAik2 *= inv_eta;
                sum2 += *ppAik2 * *ppAik2;
                ppAik2 += nc;
            }
            double sigma = sqrt(sum2);
            if (*ppAkk < 0) sigma = -sigma;
            *ppAkk += sigma;
            A1[k] = sigma * *ppAkk;
            A2[k] = -eta * sigma;
            for (int j = k + 1; j < nc; j++) {
                double *ppAik = ppAkk, sum = 0;
                for (int i = k; i < nr; i++) { sum += *ppAik * ppAik[j - k]; ppAik += nc; }
                double tau = sum / A1[k];
                ppAik = ppAkk;
                for (int i = k; i < nr; i++) { ppAik[j - k] -= tau * *ppAik; ppAik += nc; }
            }
        }
        ppAkk += nc + 1;
    }
    double* ppAjj = A;
    for (int j = 0; j < nc; j++) {
        double *ppAij = ppAjj, tau = 0;
        for (int i = j; i < nr; i++) { tau += *ppAij * b[i]; ppAij += nc; }
        tau /= A1[j];
        ppAij = ppAjj;
        for (int i = j; i < nr; i++) { b[i] -= tau * *ppAij; ppAij += nc; }
        ppAjj += nc + 1;
    }
    X[nc - 1] = b[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; i--) {
        double *ppAij = A + i * nc + (i + 1), sum = 0;
        for (int j = i + 1; j < nc; j++) { sum += *ppAij * X[j]; ppAij++; }
        X[i] = (b[i] - sum) / A2[i];
    }
}

static void gauss_newton(const double* l, const double* rho, double* betas)
{
    for (int k = 0; k < 5; k++) {
        double A[24], B[6], X[4] = {0, 0, 0, 0};
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
        for (int i = 0; i < 4; i++) betas[i] += X[i];
    }
}

/* EPnP on n >= 4 correspondences.  pws n*3 (object, mm), us n*2 (pixels). */
static void epnp_compute_pose(const double* K, const double* pws, const double* us, int n, double Rout[9], double tout[3])
{
    epnp_t e;
    e.fu = K[0]; e.fv = K[4]; e.uc = K[2]; e.vc = K[5];
    e.n = n;
    e.pws = (double*)pws; e.us = (double*)us;
    e.alphas = (double*)malloc(sizeof(double) * 4 * n);
    e.pcs = (double*)malloc(sizeof(double) * 3 * n);
    choose_control_points(&e);
    compute_barycentric_coordinates(&e);

    double mtm[144] = {0};
    for (int i = 0; i < n; i++) {
        const double* a = e.alphas + 4 * i;
        double m1[12], m2[12];
        const double u = us[2 * i], v = us[2 * i + 1];
        for (int j = 0; j < 4; j++) {
            m1[3 * j] = a[j] * e.fu; m1[3 * j + 1] = 0.0; m1[3 * j + 2] = a[j] * (e.uc - u);
            m2[3 * j] = 0.0; m2[3 * j + 1] = a[j] * e.fv; m2[3 * j + 2] = a[j] * (e.vc - v);
        }
        for (int p = 0; p < 12; p++)
            for (int q = 0; q < 12; q++) mtm[p * 12 + q] += m1[p] * m1[q];
        for (int p = 0; p < 12; p++)
            for (int q = 0; q < 12; q++) mtm[p * 12 + q] += m2[p] * m2[q];
    }
    double d[12], ut[144], vt[144];
    svd_mn(mtm, 12, 12, d, ut, vt);

    double l[60], rho[6];
    compute_L_6x10(ut, l);
    compute_rho(&e, rho);

    double betas[4][4], rep[4];
    double Rs[4][3][3], ts[4][3];
    find_betas_approx_1(l, rho, betas[1]);
    gauss_newton(l, rho, betas[1]);
    rep[1] = compute_R_and_t(&e, ut, betas[1], Rs[1], ts[1]);
    find_betas_approx_2(l, rho, betas[2]);
    gauss_newton(l, rho, betas[2]);
    rep[2] = compute_R_and_t(&e, ut, betas[2], Rs[2], ts[2]);
    find_betas_approx_3(l, rho, betas[3]);
    gauss_newton(l, rho, betas[3]);
    rep[3] = compute_R_and_t(&e, ut, betas[3], Rs[3], ts[3]);
    int N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) Rout[3 * i + j] = Rs[N][i][j];
        tout[i] = ts[N][i];
    }
    free(e.alphas);
    free(e.pcs);
}

/* solvePnP(flags=EPNP): undistortPoints (identity distortion; float32 storage when the inputs
 * are float32) -> epnp (re-applies fu,uc to the normalised points) -> Rodrigues(R) = rvec. */
static void solve_pnp_epnp(const double* K, const double* obj, const double* img, int n, int float_pts,
                           double rvec[3], double tvec[3])
{
    double* us = (double*)malloc(sizeof(double) * 2 * n);
    const double ifx = 1. / K[0], ify = 1. / K[4];
    for (int i = 0; i < n; i++) {
        double x = (img[2 * i] - K[2]) * ifx, y = (img[2 * i + 1] - K[5]) * ify;
        if (float_pts) { x = (double)(float)x; y = (double)(float)y; }
        us[2 * i] = x * K[0] + K[2];
        us[2 * i + 1] = y * K[4] + K[5];
    }
    double R[9];
    epnp_compute_pose(K, obj, us, n, R, tvec);
    rodrigues_r2v(R, rvec);
    free(us);
}

void p2po_solve_pnp_epnp(const double* K, const double* obj, const double* img, int n, double R[9], double t[3])
{
    double rvec[3];
    solve_pnp_epnp(K, obj, img, n, 0, rvec, t);
    p2po_rodrigues_v2r(rvec, R);
}

static int ransac_update_num_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* findInliers with PnPRansacCallback::computeError: project in double, store float32, squared
 * distance in float32, inlier if err <= thr^2. */
static int find_inliers(const double* K, const float* objf, const float* imgf, int n, const double rvec[3],
                        const double tvec[3], double thr, unsigned char* mask)
{
    double R[9];
    p2po_rodrigues_v2r(rvec, R);
    const float t = (float)(thr * thr);
    int nz = 0;
    for (int i = 0; i < n; i++) {
        const double X = objf[3 * i], Y = objf[3 * i + 1], Z = objf[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        const float pu = (float)(x * K[0] + K[2]), pv = (float)(y * K[4] + K[5]);
        const float du = imgf[2 * i] - pu, dv = imgf[2 * i + 1] - pv;
        const float err = du * du + dv * dv;
        const int f = err <= t;
        mask[i] = (unsigned char)f;
        nz += f;
    }
    return nz;
}

/*
 * cv2.solvePnPRansac(obj, img, K, None, flags=EPNP, reprojectionError, iterationsCount, confidence).
 * Returns 1 and fills R (= Rodrigues(rvec)), t, inlier mask (n bytes) and counts on success;
 * returns 0 when OpenCV would return inliers=None.  info[0]=#inliers, info[1]=RANSAC iterations
 * actually run, info[2]=index of the winning iteration.
 */
int p2po_solve_pnp_ransac(const double* K, const double* obj, const double* img, int n, int iterations,
                          double reproj_err, double confidence, double R[9], double t[3],
                          unsigned char* inlier_mask, int* info)
{
    const int model_points = 5;
    info[0] = info[1] = 0; info[2] = -1;
    if (n < model_points) return 0;
    float* objf = (float*)malloc(sizeof(float) * 3 * n);
    float* imgf = (float*)malloc(sizeof(float) * 2 * n);
    for (int i = 0; i < 3 * n; i++) objf[i] = (float)obj[i];     /* convertTo(CV_32F) */
    for (int i = 0; i < 2 * n; i++) imgf[i] = (float)img[i];
    unsigned char* mask = (unsigned char*)malloc(n);
    unsigned char* best_mask = (unsigned char*)malloc(n);
    double best_rvec[3] = {0, 0, 0}, best_tvec[3] = {0, 0, 0};
    int max_good = 0, ok = 0;
    double rvec[3], tvec[3];

    if (n == model_points) {
        double o5[15], i5[10];
        for (int i = 0; i < 15; i++) o5[i] = objf[i];
        for (int i = 0; i < 10; i++) i5[i] = imgf[i];
        solve_pnp_epnp(K, o5, i5, 5, 1, best_rvec, best_tvec);
        memset(best_mask, 1, n);
        max_good = n;
        ok = 1;
        info[1] = 0; info[2] = 0;
    } else {
        cv_rng rng;
        rng_init(&rng, (uint64_t)-1);
        int niters = iterations > 1 ? iterations : 1;
        int iter;
        for (iter = 0; iter < niters; iter++) {
            /* getSubset: model_points distinct indices, rng.uniform(0, n) with rejection */
            int idx[5];
            double o5[15], i5[10];
            for (int i = 0; i < model_points;) {
                int idx_i, j;
                for (;;) {
                    idx_i = idx[i] = rng_uniform(&rng, 0, n);
                    for (j = 0; j < i; j++) if (idx_i == idx[j]) break;
                    if (j == i) break;
                }
                for (int k = 0; k < 3; k++) o5[3 * i + k] = objf[3 * idx_i + k];
                for (int k = 0; k < 2; k++) i5[2 * i + k] = imgf[2 * idx_i + k];
                i++;
            }
            solve_pnp_epnp(K, o5, i5, 5, 1, rvec, tvec);
            const int good = find_inliers(K, objf, imgf, n, rvec, tvec, reproj_err, mask);
            if (good > (max_good > model_points - 1 ? max_good : model_points - 1)) {
                unsigned char* tmp = mask; mask = best_mask; best_mask = tmp;
                memcpy(best_rvec, rvec, sizeof(rvec));
                memcpy(best_tvec, tvec, sizeof(tvec));
                max_good = good;
                info[2] = iter;
                niters = ransac_update_num_iters(confidence, (double)(n - good) / n, model_points, niters);
            }
        }
        info[1] = iter;
        ok = max_good > 0;
    }
    if (ok && n == model_points) {
        /* npoints == model_points: solvePnPRansac returns the direct solvePnP result */
        p2po_rodrigues_v2r(best_rvec, R);
        t[0] = best_tvec[0]; t[1] = best_tvec[1]; t[2] = best_tvec[2];
        memcpy(inlier_mask, best_mask, n);
        info[0] = n;
    } else if (ok) {
        /* re-solve EPnP on all inliers (points keep their float32 rounding, promoted to double) */
        double* oi = (double*)malloc(sizeof(double) * 3 * max_good);
        double* ii = (double*)malloc(sizeof(double) * 2 * max_good);
        int m = 0;
        for (int i = 0; i < n; i++)
            if (best_mask[i]) {
                for (int k = 0; k < 3; k++) oi[3 * m + k] = objf[3 * i + k];
                for (int k = 0; k < 2; k++) ii[2 * m + k] = imgf[2 * i + k];
                m++;
            }
        solve_pnp_epnp(K, oi, ii, m, 0, rvec, tvec);
        p2po_rodrigues_v2r(rvec, R);
        t[0] = tvec[0]; t[1] = tvec[1]; t[2] = tvec[2];
        memcpy(inlier_mask, best_mask, n);
        info[0] = max_good;
        free(oi); free(ii);
    }
    free(objf); free(imgf); free(mask); free(best_mask);
    return ok;
}

/* many independent problems, one per OpenMP thread (bench.py cpu_baseline leg) */
void p2po_solve_pnp_ransac_batch(const double* K, const double* obj, const double* img, const int* offsets,
                                 int n_prob, int iterations, double reproj_err, double confidence, double* R,
                                 double* t, int* info, int* ok)
{
#pragma omp parallel for schedule(dynamic)
    for (int p = 0; p < n_prob; ++p) {
        const int o = offsets[p], n = offsets[p + 1] - o;
        unsigned char* mask = (unsigned char*)malloc(n > 0 ? n : 1);
        ok[p] = p2po_solve_pnp_ransac(K + 9 * p, obj + 3 * (size_t)o, img + 2 * (size_t)o, n, iterations, reproj_err,
                                      confidence, R + 9 * p, t + 3 * p, mask, info + 3 * p);
        free(mask);
    }
}

/* debug/test helper: the (rvec, tvec) model RANSAC iteration `which` produces */
int p2po_debug_hypothesis(const double* K, const double* obj, const double* img, int n, int which, double* rvec,
                          double* tvec, int* idx_out)
{
    if (n < 6) return 0;
    float* objf = (float*)malloc(sizeof(float) * 3 * n);
    float* imgf = (float*)malloc(sizeof(float) * 2 * n);
    for (int i = 0; i < 3 * n; i++) objf[i] = (float)obj[i];
    for (int i = 0; i < 2 * n; i++) imgf[i] = (float)img[i];
    cv_rng rng;
    rng_init(&rng, (uint64_t)-1);
    for (int iter = 0; iter <= which; iter++) {
        int idx[5];
        double o5[15], i5[10];
        for (int i = 0; i < 5;) {
            int idx_i, j;
            for (;;) {
                idx_i = idx[i] = rng_uniform(&rng, 0, n);
                for (j = 0; j < i; j++) if (idx_i == idx[j]) break;
                if (j == i) break;
            }
            for (int k = 0; k < 3; k++) o5[3 * i + k] = objf[3 * idx_i + k];
            for (int k = 0; k < 2; k++) i5[2 * i + k] = imgf[2 * idx_i + k];
            i++;
        }
        if (iter == which) {
            solve_pnp_epnp(K, o5, i5, 5, 1, rvec, tvec);
            for (int i = 0; i < 5; i++) idx_out[i] = idx[i];
        }
    }
    free(objf); free(imgf);
    return 1;
}
