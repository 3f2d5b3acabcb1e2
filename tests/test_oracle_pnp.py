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
