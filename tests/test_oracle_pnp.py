"""Pins oracle/pnp_oracle.c (restatement of OpenCV 3.4.2 solvePnPRansac/EPnP; OpenCV is not
installable here and the reference has no golden vectors => known-answer tests only)."""
import numpy as np
import pytest

from oracle import pnp_oracle as P
from pix2pose_amd import synthetic as synth


def _mwc(seed, n):
    """cv::RNG multiply-with-carry, independent Python big-int formulation."""
    state = seed if seed else 0xFFFFFFFF
    out = []
    for _ in range(n):
        state = ((state & 0xFFFFFFFF) * 4164903690 + (state >> 32)) & 0xFFFFFFFFFFFFFFFF
        out.append(state & 0xFFFFFFFF)
    return np.array(out, np.uint32)


def test_rng_replay_vector():
    # RANSAC seeds RNG((uint64)-1) on every call => the sample sequence is deterministic
    got = P.rng_sequence(0xFFFFFFFFFFFFFFFF, 20)
    np.testing.assert_array_equal(got, _mwc(0xFFFFFFFFFFFFFFFF, 20))
    # golden values (computed from the recurrence by hand-checkable big-int arithmetic)
    assert int(got[0]) == ((0xFFFFFFFF * 4164903690 + 0xFFFFFFFF) & 0xFFFFFFFF)


def test_rodrigues_matches_scipy():
    from scipy.spatial.transform import Rotation
    rs = np.random.RandomState(0)
    for _ in range(20):
        r = rs.randn(3) * rs.uniform(0.01, 3.0)
        np.testing.assert_allclose(P.rodrigues(r), Rotation.from_rotvec(r).as_matrix(), atol=1e-12)
    np.testing.assert_allclose(P.rodrigues(np.zeros(3)), np.eye(3))


def _scene(seed, n=200, noise=0.0, outlier_frac=0.0):
    rs = np.random.RandomState(seed)
    R = synth.random_rotation(rs)
    t = np.array([rs.uniform(-60, 60), rs.uniform(-60, 60), rs.uniform(400, 1200)])
    Pm = rs.uniform(-1, 1, (n, 3)) * synth.OBJ_PARAM[:3]
    uv = synth.project(synth.LM_K, R, t, Pm) + noise * rs.randn(n, 2)
    n_out = int(outlier_frac * n)
    if n_out:
        uv[:n_out] += rs.uniform(20, 60, (n_out, 2)) * rs.choice([-1, 1], (n_out, 2))
    return R, t, Pm, uv, n_out


@pytest.mark.parametrize("n", [5, 6, 12, 200, 5000])
def test_epnp_exact_data_recovers_pose(n):
    R, t, Pm, uv, _ = _scene(n, n=n)
    Re, te = P.solve_pnp_epnp(Pm, uv, synth.LM_K)
    dt, dr = synth.pose_error(R, t, Re, te)
    assert dt < 1e-3 and dr < 1e-4, (dt, dr)
    assert abs(np.linalg.det(Re) - 1) < 1e-9


def test_epnp_noisy_is_near_least_squares_optimum():
    """Independent check: refine the EPnP pose with scipy least_squares on the reprojection
    error; EPnP must already be close to that optimum (it is an algebraic, near-ML solver)."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    R, t, Pm, uv, _ = _scene(7, n=400, noise=0.5)
    Re, te = P.solve_pnp_epnp(Pm, uv, synth.LM_K)

    def res(x):
        Rm = Rotation.from_rotvec(x[:3]).as_matrix()
        return (synth.project(synth.LM_K, Rm, x[3:], Pm) - uv).ravel()

    x0 = np.concatenate([Rotation.from_matrix(Re).as_rotvec(), te])
    sol = least_squares(res, x0)
    Ro = Rotation.from_rotvec(sol.x[:3]).as_matrix()
    dt, dr = synth.pose_error(Ro, sol.x[3:], Re, te)
    assert dt < 3.0 and dr < 0.3, (dt, dr)
    dt, dr = synth.pose_error(R, t, Re, te)
    assert dt < 5.0 and dr < 0.5, (dt, dr)


@pytest.mark.parametrize("frac", [0.0, 0.2, 0.4])
def test_ransac_with_outliers(frac):
    R, t, Pm, uv, n_out = _scene(11, n=600, noise=0.3, outlier_frac=frac)
    ok, Re, te, inl, meta = P.solve_pnp_ransac(Pm, uv, synth.LM_K)
    assert ok
    dt, dr = synth.pose_error(R, t, Re, te)
    assert dt < 3.0 and dr < 0.3, (dt, dr)
    assert not np.any(inl < n_out)                      # no gross outlier among the inliers
    assert len(inl) >= 0.95 * (600 - n_out)
    assert meta["n_inliers"] == len(inl)
    if frac == 0.0:
        # adaptive stop: with all inliers the iteration count collapses (log(0.01)/log(1-1) -> 0)
        assert meta["iterations"] <= 2


def test_ransac_is_deterministic_and_degenerate_inputs():
    R, t, Pm, uv, _ = _scene(5, n=300, noise=0.3, outlier_frac=0.3)
    a = P.solve_pnp_ransac(Pm, uv, synth.LM_K)
    b = P.solve_pnp_ransac(Pm, uv, synth.LM_K)
    np.testing.assert_array_equal(a[1], b[1])
    np.testing.assert_array_equal(a[3], b[3])
    assert P.solve_pnp_ransac(Pm[:4], uv[:4], synth.LM_K)[0] is False      # n < 5
    ok, Re, te, inl, _ = P.solve_pnp_ransac(Pm[:5], uv[:5], synth.LM_K)    # n == model points: direct solve
    assert ok and len(inl) == 5
    # pure garbage: no consistent model with > 4 inliers => cv2 returns inliers=None
    rs = np.random.RandomState(1)
    ok, *_ = P.solve_pnp_ransac(rs.uniform(-40, 40, (60, 3)), rs.uniform(0, 640, (60, 2)), synth.LM_K)
    assert ok is False


def test_ransac_on_quantised_nocs_render():
    """End-to-end shape of the reference's use (recognition.py:196-217): 8-bit XYZ, pixel grid."""
    rs = np.random.RandomState(3)
    R = synth.random_rotation(rs)
    t = np.array([10.0, -20.0, 700.0])
    ctr = synth.project(synth.LM_K, R, t, np.zeros((1, 3)))[0]
    u0, v0 = int(ctr[0]) - 64, int(ctr[1]) - 64
    nocs, hit = synth.render_ellipsoid_nocs(R, t, synth.LM_K, synth.OBJ_PARAM[:3], u0, v0, 128, 128)
    q = ((nocs + 1) / 2 * 255).astype(np.uint8)                    # truncation, as the uint8 canvas does
    xyz = (q / 255.0 * 2 - 1) * synth.OBJ_PARAM[:3] + synth.OBJ_PARAM[3:]
    vs, us = np.nonzero(hit)
    obj = xyz[vs, us]
    img = np.stack([us + u0, vs + v0], 1).astype(np.float64)
    ok, Re, te, inl, meta = P.solve_pnp_ransac(obj, img, synth.LM_K)
    assert ok
    dt, dr = synth.pose_error(R, t, Re, te)
    assert dt < 8.0 and dr < 1.5, (dt, dr)          # bounded by the 8-bit quantisation (scale*2/255 mm)
    assert len(inl) > 0.9 * len(obj)


# ---------------------------------------------------------------------------------------------------------------------------------
# The schedule of the team form of the 12x12 Jacobi SVD (pix2pose_amd/csrc/pnp.hip: jacobi12_team).  OpenCV's JacobiSVDImpl_ visits the pairs
# (i, j), i < j, row by row, sweep after sweep; the kernel visits pair (i, j) of sweep s at step 12 s + i + j, up to six pairs per step on
# six lane quads.  The claims the kernel rests on are combinatorial and hold on any machine: no two pairs of a step share a row, never more
# than six pairs per step, and every pair finds its rows as the sequential order would hand them over -- so a plain float64 restatement
# walked in both orders must agree to the last bit, including the sweep after which it stops.
def _jacobi_rot(A, W, i, j, eps):
    """One pair visit of OpenCV's one-sided Jacobi on the rows of A (sequential sums over k); returns whether it rotated."""
    a, b = W[i], W[j]
    p = 0.0
    for k in range(A.shape[1]):
        p += A[i, k] * A[j, k]
    if abs(p) <= eps * np.sqrt(a * b):
        return False
    p *= 2.0
    beta = a - b
    gamma = np.sqrt(p * p + beta * beta)
    if beta < 0:
        s = np.sqrt((gamma - beta) * 0.5 / gamma)
        c = p / (gamma * s * 2)
    else:
        c = np.sqrt((gamma + beta) / (gamma * 2))
        s = p / (gamma * c * 2)
    a = b = 0.0
    for k in range(A.shape[1]):
        t0 = c * A[i, k] + s * A[j, k]
        t1 = -s * A[i, k] + c * A[j, k]
        A[i, k], A[j, k] = t0, t1
        a += t0 * t0
        b += t1 * t1
    W[i], W[j] = a, b
    return True


def _jacobi_sequential(A, max_iter=30):
    A = A.copy()
    n = A.shape[0]
    W = np.array([sum(A[i, k] * A[i, k] for k in range(n)) for i in range(n)])
    eps = np.finfo(np.float64).eps * 10
    sweeps = 0
    for _ in range(max_iter):
        sweeps += 1
        changed = False
        for i in range(n - 1):
            for j in range(i + 1, n):
                changed |= _jacobi_rot(A, W, i, j, eps)
        if not changed:
            break
    return A, W, sweeps


def _team_steps(n=12):
    """step -> pairs of that step: (sweep offset 0 = newer / -1 = older sweep, i, j); the loop structure of jacobi12_team."""
    out = {}
    for t_hi in range(1, n + 1):
        pairs = []
        t_lo = t_hi + n
        if t_lo <= 2 * n - 3:
            pairs += [(-1, i, t_lo - i) for i in range(t_lo - (n - 1), (t_lo - 1) // 2 + 1)]
        pairs += [(0, i, t_hi - i) for i in range(max(0, t_hi - (n - 1)), (t_hi - 1) // 2 + 1)]
        out[t_hi] = pairs
    return out


def _jacobi_team(A, max_iter=30):
    A = A.copy()
    n = A.shape[0]
    W = np.array([sum(A[i, k] * A[i, k] for k in range(n)) for i in range(n)])
    eps = np.finfo(np.float64).eps * 10
    steps = _team_steps(n)
    chg_lo = chg_hi = False
    s_hi = 0
    while s_hi <= max_iter:
        for t_hi in range(1, n + 1):
            for which, i, j in steps[t_hi]:
                if which == -1 and s_hi < 1:
                    continue
                if which == 0 and s_hi >= max_iter:
                    continue
                r = _jacobi_rot(A, W, i, j, eps)
                if which == -1:
                    chg_lo |= r
                else:
                    chg_hi |= r
            if t_hi == 9 and s_hi >= 1 and not chg_lo:          # sweep s_hi - 1 walked its last diagonal (i + j = 21) without a rotation
                return A, W, s_hi
        chg_lo, chg_hi = chg_hi, False
        s_hi += 1
    return A, W, max_iter


def test_team_jacobi_schedule_is_conflict_free_and_complete():
    n = 12
    steps = _team_steps(n)
    seen = {}
    for t_hi, pairs in steps.items():
        assert len(pairs) <= 6, (t_hi, pairs)
        rows = [r for _w, i, j in pairs for r in (i, j)]
        assert len(rows) == len(set(rows)), (t_hi, pairs)                       # the pairs of a step share no row
        for w, i, j in pairs:
            assert 0 <= i < j < n
            seen.setdefault((i, j), []).append(12 * (1 if w == -1 else 0) + t_hi)
    assert sorted(seen) == [(i, j) for i in range(n) for j in range(i + 1, n)]  # every pair exactly once per sweep ...
    for (i, j), at in seen.items():
        assert at == [i + j], ((i, j), at)                                      # (diagonals 13 .. 21 ride along with the next sweep's 1 .. 9)
    # ... at step i + j of its sweep, and after the last visits of its rows (the dependency chain of OpenCV's order)
    order = [(i, j) for i in range(n) for j in range(i + 1, n)]
    last = {}
    for s in range(3):
        for (i, j) in order:
            t = 12 * s + i + j
            for r in (i, j):
                assert last.get(r, -1) < t, ((s, i, j), r, last.get(r))
            last[i] = last[j] = t


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_team_jacobi_order_gives_the_bits_of_the_sequential_order(seed):
    rs = np.random.RandomState(seed)
    M = rs.randn(10 if seed % 2 else 14, 12) * (10.0 ** rs.uniform(-2, 2, (1, 12)))
    A = M.T @ M                                                # symmetric, like M^T M of EPnP (rank 10 for the odd seeds: two null directions)
    As, Ws, ss = _jacobi_sequential(A)
    At, Wt, st = _jacobi_team(A)
    assert ss == st and 2 <= ss <= 30
    assert np.array_equal(As.view(np.uint64), At.view(np.uint64))
    assert np.array_equal(Ws.view(np.uint64), Wt.view(np.uint64))


# ---------------------------------------------------------------------------------------------------------------------------------
# betas_approx_any (pnp.hip): the 6 x 4 and 6 x 3 least-squares systems of EPnP's first two beta cases are solved in the code of the 6 x 5 one,
# their columns followed by zero columns.  The claim: a zero column never rotates (its dot products are 0: "already orthogonal"), sorts
# behind every live singular value and is dropped by the back-substitution, so the live columns see exactly the operations of the narrow
# solve.  A float64 restatement of cv::SVD (JacobiSVDImpl_ on A^T with V) + SVD::backSubst run both ways must agree to the last bit.
def _svd_solve(A, b, n_live=None):
    M, N = A.shape
    n_live = N if n_live is None else n_live
    At = A.T.copy()
    Vt = np.eye(N)
    eps = np.finfo(np.float64).eps * 10
    W = np.array([sum(At[i, k] * At[i, k] for k in range(M)) for i in range(N)])
    for _ in range(max(M, 30)):
        changed = False
        for i in range(N - 1):
            for j in range(i + 1, N):
                a, bb = W[i], W[j]
                p = 0.0
                for k in range(M):
                    p += At[i, k] * At[j, k]
                if abs(p) <= eps * np.sqrt(a * bb):
                    continue
                p *= 2.0
                beta = a - bb
                gamma = np.sqrt(p * p + beta * beta)
                if beta < 0:
                    s = np.sqrt((gamma - beta) * 0.5 / gamma)
                    c = p / (gamma * s * 2)
                else:
                    c = np.sqrt((gamma + beta) / (gamma * 2))
                    s = p / (gamma * c * 2)
                a = bb = 0.0
                for k in range(M):
                    t0 = c * At[i, k] + s * At[j, k]
                    t1 = -s * At[i, k] + c * At[j, k]
                    At[i, k], At[j, k] = t0, t1
                    a += t0 * t0
                    bb += t1 * t1
                W[i], W[j] = a, bb
                changed = True
                for k in range(N):
                    t0 = c * Vt[i, k] + s * Vt[j, k]
                    t1 = -s * Vt[i, k] + c * Vt[j, k]
                    Vt[i, k], Vt[j, k] = t0, t1
        if not changed:
            break
    for i in range(N):
        W[i] = np.sqrt(sum(At[i, k] * At[i, k] for k in range(M)))
    for i in range(N - 1):
        j = i
        for k in range(i + 1, N):
            if W[j] < W[k]:
                j = k
        if i != j:
            W[[i, j]] = W[[j, i]]
            At[[i, j]] = At[[j, i]]
            Vt[[i, j]] = Vt[[j, i]]
    for i in range(N):
        assert i >= n_live or W[i] > np.finfo(np.float64).tiny      # (no zero singular value among the live ones: OpenCV's random fill-in not restated here)
        s = 1.0 / W[i] if W[i] > np.finfo(np.float64).tiny else 0.0
        At[i] *= s
    thr = 0.0
    for i in range(N):
        thr += W[i]
    thr *= np.finfo(np.float64).eps * 2
    x = np.zeros(N)
    for i in range(N):
        if abs(W[i]) <= thr:
            continue
        s = 0.0
        for k in range(M):
            s += At[i, k] * b[k]
        s *= 1.0 / W[i]
        for j in range(N):
            x[j] += s * Vt[i, j]
    return x


@pytest.mark.parametrize("n", [3, 4])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_zero_padded_svd_solve_gives_the_bits_of_the_narrow_one(seed, n):
    rs = np.random.RandomState(10 * n + seed)
    A = rs.randn(6, n) * (10.0 ** rs.uniform(-1, 3, (1, n)))
    b = rs.randn(6) * 100
    narrow = _svd_solve(A, b)
    wide = _svd_solve(np.concatenate([A, np.zeros((6, 5 - n))], axis=1), b, n_live=n)
    assert np.array_equal(narrow.view(np.uint64), wide[:n].view(np.uint64))
    assert not wide[n:].any()
