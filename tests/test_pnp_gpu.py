"""GPU parity of the batched HIP EPnP+RANSAC kernel (through p2p_pnp_ransac_batch) against the
oracle restatement of cv2.solvePnPRansac.  Integer results (inlier sets, iteration counts, winning
iteration) must match EXACTLY on every problem -- the GPU solver replays OpenCV's sampling order and
best-so-far rule -- and poses within 1e-6 mm / 1e-4 deg of the oracle (north_star bar: 1 mm / 1 deg).
The exact-match rate of each test is appended to gpurun_out/pnp_exact_match.json."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _log_rate(test, n_exact, n_total, mismatches):
    """Record the measured exact-match rate (DESIGN.md quotes it); best effort, never fails the test."""
    try:
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        fn = os.path.join(d, "pnp_exact_match.json")
        log = json.load(open(fn)) if os.path.exists(fn) else {}
        log[test] = {"exact": int(n_exact), "problems": int(n_total), "mismatches": mismatches}
        json.dump(log, open(fn, "w"), indent=1)
    except OSError:
        pass

from pix2pose_amd import synthetic as synth

pytestmark = pytest.mark.gpu


def _scenes(n_prob, seed0=100, n_pts=(6, 3000), outliers=(0.0, 0.4)):
    rs = np.random.RandomState(seed0)
    Ks, objs, imgs, gts = [], [], [], []
    for p in range(n_prob):
        n = int(rs.randint(n_pts[0], n_pts[1]))
        R = synth.random_rotation(rs)
        t = np.array([rs.uniform(-60, 60), rs.uniform(-60, 60), rs.uniform(400, 1200)])
        P = rs.uniform(-1, 1, (n, 3)) * synth.OBJ_PARAM[:3]
        uv = synth.project(synth.LM_K, R, t, P) + 0.3 * rs.randn(n, 2)
        n_out = int(rs.uniform(*outliers) * n)
        uv[:n_out] += rs.uniform(20, 60, (n_out, 2)) * rs.choice([-1, 1], (n_out, 2))
        Ks.append(synth.LM_K); objs.append(P); imgs.append(uv); gts.append((R, t, n_out))
    return Ks, objs, imgs, gts


def test_pnp_batch_matches_oracle():
    from oracle import pnp_oracle as O
    from pix2pose_amd.runtime import default_context, pnp_ransac_batch
    Ks, objs, imgs, gts = _scenes(48)
    ok, R, t, info, masks = pnp_ransac_batch(default_context(), Ks, objs, imgs, want_mask=True)
    n_exact, n_ok, bad = 0, 0, []
    for p in range(len(objs)):
        ok0, R0, t0, inl0, meta = O.solve_pnp_ransac(objs[p], imgs[p], Ks[p])
        assert bool(ok[p]) == ok0, p
        if not ok0:
            continue
        n_ok += 1
        dt, dr = synth.pose_error(R0, t0, R[p], t[p])
        if info[p, 2] == meta["best_iter"] and info[p, 0] == meta["n_inliers"] and info[p, 1] == meta["iterations"] \
                and np.array_equal(np.nonzero(masks[p])[0], inl0):
            n_exact += 1
            assert dt < 1e-6 and dr < 1e-4, (p, dt, dr)   # arccos resolves ~1e-6 deg near identity
        else:
            bad.append({"problem": p, "n": len(objs[p]), "gpu": [int(v) for v in info[p]],
                        "oracle": [meta["n_inliers"], meta["iterations"], meta["best_iter"]], "dt_mm": dt, "dr_deg": dr})
        gdt, gdr = synth.pose_error(gts[p][0], gts[p][1], R[p], t[p])
        if len(objs[p]) > 100:
            assert gdt < 5.0 and gdr < 1.0, (p, gdt, gdr)
    _log_rate("test_pnp_batch_matches_oracle", n_exact, n_ok, bad)
    assert not bad, bad               # every problem: same winning hypothesis, same inlier set, same iteration count


def test_pnp_lazy_second_hypothesis_batch_matches_oracle():
    """Heavy outlier ratios keep RANSAC's adaptive bound above 64 iterations, which exercises the second
    (lazy) hypothesis batch of the GPU solver: iteration counts, winners and inlier sets still match the
    oracle's sequential loop, including problems that use all 100 iterations."""
    from oracle import pnp_oracle as O
    from pix2pose_amd.runtime import default_context, pnp_ransac_batch
    Ks, objs, imgs, gts = _scenes(24, seed0=321, n_pts=(200, 1500), outliers=(0.45, 0.7))
    ok, R, t, info, masks = pnp_ransac_batch(default_context(), Ks, objs, imgs, want_mask=True)
    n_tail = n_exact = n_ok = 0
    bad = []
    for p in range(len(objs)):
        ok0, R0, t0, inl0, meta = O.solve_pnp_ransac(objs[p], imgs[p], Ks[p])
        assert bool(ok[p]) == ok0, p
        if not ok0:
            continue
        n_ok += 1
        n_tail += meta["iterations"] > 64
        dt, dr = synth.pose_error(R0, t0, R[p], t[p])
        if info[p, 2] == meta["best_iter"] and info[p, 0] == meta["n_inliers"] and info[p, 1] == meta["iterations"] \
                and np.array_equal(np.nonzero(masks[p])[0], inl0):
            n_exact += 1
            assert dt < 1e-6 and dr < 1e-4, (p, dt, dr)
        else:
            bad.append({"problem": p, "n": len(objs[p]), "gpu": [int(v) for v in info[p]],
                        "oracle": [meta["n_inliers"], meta["iterations"], meta["best_iter"]], "dt_mm": dt, "dr_deg": dr})
    _log_rate("test_pnp_lazy_second_hypothesis_batch_matches_oracle", n_exact, n_ok, bad)
    assert n_tail >= 5, n_tail            # the scenes really reach the second batch
    assert not bad, bad


def test_pnp_edge_cases():
    from oracle import pnp_oracle as O
    from pix2pose_amd.runtime import default_context, pnp_ransac_batch
    rs = np.random.RandomState(5)
    Ks, objs, imgs, gts = _scenes(3, seed0=7, n_pts=(200, 201))
    # exact (noise-free) five-point set: a minimal problem is only well posed without noise (the
    # 12x12 system then has a 2-D null space whose basis is rounding dependent)
    R5g, t5g = gts[0][0], gts[0][1]
    P5 = rs.uniform(-1, 1, (5, 3)) * synth.OBJ_PARAM[:3]
    uv5 = synth.project(synth.LM_K, R5g, t5g, P5)
    # too few points; exactly five; garbage (no model); empty problem
    objs += [objs[0][:4], P5, rs.uniform(-40, 40, (60, 3)), np.zeros((0, 3))]
    imgs += [imgs[0][:4], uv5, rs.uniform(0, 640, (60, 2)), np.zeros((0, 2))]
    Ks += [synth.LM_K] * 4
    ok, R, t, info, _ = pnp_ransac_batch(default_context(), Ks, objs, imgs)
    assert list(ok[:3]) == [True] * 3
    assert not ok[3] and ok[4] and not ok[5] and not ok[6]
    assert info[4, 0] == 5
    ok5, R5, t5, _, _ = O.solve_pnp_ransac(P5, uv5, synth.LM_K)
    assert ok5
    for Rx, tx in ((R5, t5), (R[4], t[4])):          # float32 point storage bounds the accuracy
        dt, dr = synth.pose_error(R5g, t5g, Rx, tx)
        assert dt < 0.05 and dr < 0.01, (dt, dr)
    with pytest.raises(Exception):       # more iterations than the solver's hypothesis storage: an error, not a silent clamp
        pnp_ransac_batch(default_context(), Ks[:1], objs[:1], imgs[:1], iterations=129)
    ok128, _, _, info128, _ = pnp_ransac_batch(default_context(), Ks[:1], objs[:1], imgs[:1], iterations=128)
    assert ok128[0] and info128[0, 1] <= 128
    for p in (3, 5, 6):      # failure convention of recognition.py:215,219: identity, zero, -1
        np.testing.assert_array_equal(R[p], np.eye(3))
        np.testing.assert_array_equal(t[p], np.zeros(3))
        assert info[p, 0] == -1


def test_pnp_full_size_problems_match_oracle():
    """Full-size candidates (n = 128*128 correspondences, the BASELINE.json configs[2-3] size; 20 % outliers): the C oracle
    handles them in milliseconds, so they are compared like the small ones -- same inlier set, winner and iteration count,
    pose to rounding -- plus ground truth and invariance to the batch position."""
    from oracle import pnp_oracle as O
    from pix2pose_amd.runtime import default_context, pnp_ransac_batch
    Ks, objs, imgs, gts = _scenes(6, seed0=11, n_pts=(16384, 16385), outliers=(0.2, 0.2))
    ok, R, t, info, masks = pnp_ransac_batch(default_context(), Ks, objs, imgs, want_mask=True)
    for p in range(6):
        ok0, R0, t0, inl0, meta = O.solve_pnp_ransac(objs[p], imgs[p], Ks[p])
        assert ok0 and ok[p]
        assert [int(v) for v in info[p]] == [meta["n_inliers"], meta["iterations"], meta["best_iter"]], p
        np.testing.assert_array_equal(np.nonzero(masks[p])[0], inl0)
        dt, dr = synth.pose_error(R0, t0, R[p], t[p])
        assert dt < 1e-6 and dr < 1e-4, (p, dt, dr)
    ok2, R2, t2, info2, _ = pnp_ransac_batch(default_context(), Ks[::-1], objs[::-1], imgs[::-1])
    np.testing.assert_array_equal(R, R2[::-1])
    np.testing.assert_array_equal(info, info2[::-1])
    for p in range(6):
        assert ok[p]
        dt, dr = synth.pose_error(gts[p][0], gts[p][1], R[p], t[p])
        assert dt < 1.0 and dr < 0.1, (dt, dr)
        assert info[p, 0] >= 0.79 * 16384
        assert info[p, 1] < 100          # adaptive stop


def test_pnp_large_crops_with_heavy_outliers_match_oracle():
    """Candidates of 200 - 450-px crops (40 000 - 130 000 correspondences) at outlier ratios that push RANSAC's bound past the first
    and the second hypothesis round: the inlier counts are taken by independent workgroups over slices of the points
    (pnp_count_kernel) and OpenCV's rule resumes from its parked state after every round (pnp_score_kernel) -- iteration counts,
    winners and inlier sets must still be the oracle's, also when small problems share the batch."""
    from oracle import pnp_oracle as O
    from pix2pose_amd.runtime import default_context, pnp_ransac_batch
    big = _scenes(5, seed0=77, n_pts=(40000, 130000), outliers=(0.35, 0.65))
    small = _scenes(3, seed0=78, n_pts=(50, 400), outliers=(0.3, 0.6))
    Ks, objs, imgs = big[0] + small[0], big[1] + small[1], big[2] + small[2]
    order = [5, 0, 1, 6, 2, 3, 7, 4]
    Ks, objs, imgs = [Ks[i] for i in order], [objs[i] for i in order], [imgs[i] for i in order]
    ok, R, t, info, masks = pnp_ransac_batch(default_context(), Ks, objs, imgs, want_mask=True)
    rounds = set()
    for p in range(len(objs)):
        ok0, R0, t0, inl0, meta = O.solve_pnp_ransac(objs[p], imgs[p], Ks[p])
        assert bool(ok[p]) == ok0, p
        if not ok0:
            continue
        assert [int(v) for v in info[p]] == [meta["n_inliers"], meta["iterations"], meta["best_iter"]], p
        np.testing.assert_array_equal(np.nonzero(masks[p])[0], inl0)
        dt, dr = synth.pose_error(R0, t0, R[p], t[p])
        assert dt < 1e-6 and dr < 1e-4, (p, dt, dr)
        if len(objs[p]) > 1000:
            rounds.add(0 if meta["iterations"] <= 16 else 1 if meta["iterations"] <= 64 else 2)
    assert len(rounds) >= 2, rounds          # the large problems really end in different hypothesis rounds


def test_small_and_large_launches_agree_bit_for_bit():
    """Launches of up to 48 problems solve hypotheses [0, 48) up front (pnp.hip: launch_pnp_ransac), larger ones in the lazy rounds
    [0, 16) [16, 64) [64, 100): WHEN a hypothesis is solved must not change any result -- the same 40 problems alone and as the head of a
    72-problem launch, compared bit for bit (noisy problems: many of them run past 16 iterations)."""
    from pix2pose_amd.runtime import default_context, pnp_ransac_batch
    Ks, objs, imgs, _ = _scenes(72, seed0=321, n_pts=(40, 2500), outliers=(0.3, 0.6))
    ok_a, R_a, t_a, info_a, m_a = pnp_ransac_batch(default_context(), Ks[:40], objs[:40], imgs[:40], want_mask=True)
    ok_b, R_b, t_b, info_b, m_b = pnp_ransac_batch(default_context(), Ks, objs, imgs, want_mask=True)
    assert (np.asarray(info_a)[:, 1] > 16).sum() >= 5          # iterations: the second round is exercised
    np.testing.assert_array_equal(np.asarray(ok_a), np.asarray(ok_b)[:40])
    np.testing.assert_array_equal(np.asarray(info_a), np.asarray(info_b)[:40])
    np.testing.assert_array_equal(np.asarray(R_a), np.asarray(R_b)[:40])
    np.testing.assert_array_equal(np.asarray(t_a), np.asarray(t_b)[:40])
    for p in range(40):
        np.testing.assert_array_equal(np.asarray(m_a[p]), np.asarray(m_b[p]))
