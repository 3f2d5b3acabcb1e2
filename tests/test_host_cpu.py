"""CPU tests of the host side: C-ABI exports, weight artefact, shim geometry vs the oracle,
resize restatement vs scipy, sharding, and the world_size-2 gloo pose gather."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    from pix2pose_amd import _lib, build
    build.build()
    hdr = open(os.path.join(ROOT, "include", "p2p_mi355.h")).read()
    names = sorted(set(re.findall(r"\b(p2p_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 14, names
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libp2p_mi355.so does not export %s" % n
    assert _lib.lib().p2p_abi_version() == _lib.ABI_VERSION
    # ... and nothing else: the library is built with -fvisibility=hidden, its C entry points are its whole dynamic symbol table
    # (no mangled C++ helpers, no kernel host stubs).  The few linker-provided names every shared object carries are filtered out.
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = sorted({ln.split()[-1] for ln in out.splitlines() if ln.strip()})
    extra = [d for d in defined if d not in names and d not in ("_init", "_fini", "__bss_start", "_edata", "_end")]
    assert extra == [], "symbols exported beside the C ABI: %s" % extra[:10]


def test_no_gpu_fails_loudly_not_silently():
    """Without a device the product path must raise -- there is no CPU fallback."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pix2pose_amd import _lib
    from pix2pose_amd.runtime import Context
    with pytest.raises(_lib.P2PError):
        Context(0)


def test_product_does_not_import_the_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "pix2pose_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "oracle/" not in src or f in ("pnp.hip",), f      # pnp.hip only mentions it in a comment


def test_weight_artefact_roundtrip(tmp_path):
    from pix2pose_amd import weights as W
    w = W.synthetic_weights("paper", 3)
    fn = str(tmp_path / "w.npz")
    W.save_weights(fn, "paper", w)
    w2 = W.load_weights(fn, "paper")
    assert all(np.array_equal(w[k], w2[k]) for k in w)
    with pytest.raises(ValueError):
        W.load_weights(fn, "resnet50")
    bad = dict(w)
    bad.pop("deconv2.kernel")
    with pytest.raises(ValueError):
        W.check_weights("paper", bad)
    w3 = W.load_weights("synthetic:paper:3", "paper")
    assert np.array_equal(w3["conv1_1.kernel"], w["conv1_1.kernel"])
    # bit-reproducible generator: a fixed probe value (integer hash, no libm)
    assert float(W._hash_normal(1, 0, 4)[2]) == float(W._hash_normal(1, 0, 4)[2])
    assert abs(float(np.std(W._hash_normal(5, 1, 200000))) - 1.0) < 0.01


@settings(max_examples=300, deadline=None)
@given(st.integers(-40, 500), st.integers(-40, 660), st.integers(1, 400), st.integers(1, 400), st.integers(100, 480),
       st.integers(100, 640))
def test_shim_get_boxes_equals_oracle(v0, u0, h, w, H, Wd):
    from oracle import est_pose_oracle as E
    from pix2pose_amd.recognition import get_boxes
    bbox = [v0, u0, v0 + h, u0 + w]
    assert list(get_boxes(bbox, H, Wd)) == E.get_boxes(bbox, H, Wd).as_list()
    ct = np.array([v0 + h // 3, u0 + w // 2])
    fb = np.array(bbox) * 0.73
    assert list(get_boxes(fb, H, Wd, 1.5, ct, 77)) == E.get_boxes(fb, H, Wd, 1.5, ct, 77).as_list()


def test_get_boxes_reference_examples():
    from pix2pose_amd.recognition import get_boxes
    # bbox 86x86 -> side 2*int(1.5*86/2) = 128; clipped at the frame border
    assert get_boxes([100, 200, 186, 286], 480, 640) == (79, 207, 179, 307, 79, 207, 179, 307, 0, 128, 0, 128)
    assert get_boxes([-10, -20, 60, 70], 240, 320) == (-42, 92, -42, 92, 0, 92, 0, 92, 42, 134, 42, 134)
    assert get_boxes([400, 600, 470, 650], 480, 640)[4:8] == (383, 480, 573, 640)


@settings(max_examples=40, deadline=None)
@given(st.integers(5, 200), st.integers(5, 200), st.sampled_from(["reflect", "constant"]), st.sampled_from([0.0, 0.5, 1.0]))
def test_resize_restatement_matches_scipy_zoom(n_in, n_out, mode, cval):
    """skimage.transform.resize(order=1, anti_aliasing=False) == scipy.ndimage.zoom(grid_mode=True)
    with 'mirror' / 'grid-constant' (that is literally what current scikit-image calls)."""
    from scipy import ndimage as ndi
    from oracle.est_pose_oracle import resize_bilinear
    a = np.random.RandomState(n_in * 1000 + n_out).rand(n_in, n_in)
    z = ndi.zoom(a, n_out / n_in, order=1, mode="mirror" if mode == "reflect" else "grid-constant", cval=cval, grid_mode=True)
    if z.shape != (n_out, n_out):
        return      # zoom rounds the output size itself for some ratios
    r = resize_bilinear(a, (n_out, n_out), mode, cval, clip=False)     # the warp itself; skimage then clips (own test below)
    assert np.abs(z - r).max() < 1e-12
    rc = resize_bilinear(a, (n_out, n_out), mode, cval)
    keep = (r == cval) if (mode == "constant" and not (a.min() <= cval <= a.max())) else np.zeros_like(r, bool)
    assert np.array_equal(rc, np.where(keep, cval, np.clip(r, a.min(), a.max())))


def test_shard_detections_balanced_and_grouped():
    from pix2pose_amd.parallel import shard_detections
    rs = np.random.RandomState(0)
    dets = [(0, int(rs.randint(0, 30)), [0, 0, 1, 1], None) for _ in range(2048)]
    order, bounds = shard_detections(dets, 8)
    assert sorted(order) == list(range(2048)) and bounds[0] == 0 and bounds[-1] == 2048
    sizes = np.diff(bounds)
    assert sizes.max() - sizes.min() <= 1
    objs = [dets[i][1] for i in order]
    assert objs == sorted(objs)
    order, bounds = shard_detections(dets[:5], 8)          # fewer detections than ranks
    assert np.diff(bounds).tolist() == [1, 1, 1, 1, 1, 0, 0, 0]


_GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from pix2pose_amd import _lib
from pix2pose_amd.parallel import gather_poses, poses_to_records, shard_detections
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N = int(os.environ.get("P2P_TEST_DETS", "11"))
dets = [(0, i %% 3, [0, 0, 1, 1], None) for i in range(N)]
order, bounds = shard_detections(dets, world)
mine = order[bounds[rank]:bounds[rank + 1]]
poses = []
for i in mine:                       # stand-in for the per-rank pipeline output: recognisable values
    p = _lib.Pose()
    p.status = 0; p.n_inliers = 100 + i; p.n_init_mask = 1000; p.frac_inlier = i / 7.0; p.best_slot = i %% 3
    for k in range(9): p.R[k] = i + k / 16.0
    for k in range(3): p.t[k] = -i - k / 3.0
    poses.append(p)
from pix2pose_amd.parallel import gather_poses_async
h1 = gather_poses_async(poses_to_records(poses, ids=mine))       # two gathers in flight, collected in order
h2 = gather_poses_async(poses_to_records(poses, ids=mine))
rec = h1.result()
assert (h2.result() == rec).all()
assert (gather_poses(poses_to_records(poses, ids=mine)) == rec).all()
assert rec.shape == (N, 20), rec.shape
assert rec[:, 0].tolist() == list(range(N))
for i in range(N):
    assert rec[i, 3] == 100 + i and rec[i, 2] == i / 7.0
    assert rec[i, 6:15].tolist() == [i + k / 16.0 for k in range(9)]      # float64 records: bit exact
    assert rec[i, 15:18].tolist() == [-i - k / 3.0 for k in range(3)]
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_pose_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29731", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == 2


def test_pose_gather_world_size_8_with_empty_shards_gloo(tmp_path):
    """Eight ranks, five detections: three ranks own an EMPTY shard and still take part in every gather (padding to the largest shard,
    padded rows dropped) -- the shape of BASELINE.json configs[3] / [4] whenever an object group or the tail of an image stream is smaller
    than the node (tools/5_evaluation_bop_basic.py:289-304 makes the batches ragged)."""
    script = tmp_path / "worker8.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", P2P_TEST_DETS="5", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                        "--master-addr", "127.0.0.1", "--master-port", "29733", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count("ok") == 8


def test_binding_refuses_a_stale_library(monkeypatch):
    """A .so built from other sources, or with other struct layouts, must not be loaded silently (it is git-ignored and
    travels out of band): the binding compares p2p_build_id() with the tree's source hash and p2p_abi_sizeof() with
    its own ctypes declarations."""
    from pix2pose_amd import _lib, build
    build.build()
    assert _lib._stale_reason(_lib.LIB_PATH) is None
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 32)
    assert "other sources" in _lib._stale_reason(_lib.LIB_PATH)
    monkeypatch.undo()

    class Fat(_lib.C.Structure):
        _fields_ = _lib.Pose._fields_ + [("extra", _lib.C.c_int * 4)]
    monkeypatch.setattr(_lib, "Pose", Fat)
    assert "sizeof(Fat)" in _lib._stale_reason(_lib.LIB_PATH)


def test_stale_check_sees_a_rebuilt_library(tmp_path):
    """lib() checks a library, rebuilds it when it is stale and checks again UNDER THE SAME PATH: the check must close its
    handle, else the loader hands back the image it already mapped and the fresh build is reported stale ("... after a
    rebuild": seen on a GPU box that received a library built from an edited tree)."""
    import subprocess
    from pix2pose_amd import _lib
    path = str(tmp_path / "libprobe.so")
    for version in (111, 222):
        src = tmp_path / ("v%d.c" % version)
        src.write_text("int p2p_abi_version(void) { return %d; }\n" % version)
        tmp = str(tmp_path / ("v%d.so" % version))
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", tmp, str(src)])
        os.replace(tmp, path)
        assert "ABI version %d" % version in _lib._stale_reason(path)


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` outside a launcher re-executes itself under torch.distributed.run on 127.0.0.1."""
    import importlib
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(os, "execve", lambda exe, cmd, env: seen.update(exe=exe, cmd=cmd, env=env))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "3"])
    bench._self_launch(8)
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_anti_aliasing_weights_equal_scipy_gaussian_kernel():
    """The library's host-side Gaussian weights (p2p_aa_weights: sigma = (in/out - 1)/2, radius int(4 sigma + .5),
    numpy's pairwise normalisation) against scipy.ndimage's own kernel builder: identical up to the last bit of exp().
    numpy >= 1.19 evaluates float64 exp with its own SIMD routine, which differs from libm's by 1 ulp on ~30 % of the
    vectors; the reference's era (numpy <= 1.18) called libm like the library does.  So: bit-exact against the scipy
    formula evaluated with libm's exp, and within a few ulp of today's scipy."""
    import ctypes as C
    import math
    from scipy.ndimage import _filters
    from pix2pose_amd import _lib
    L = _lib.lib()
    n = 0
    for side in list(range(5, 128)) + list(range(129, 700, 3)):
        w = (C.c_double * 256)()
        r = L.p2p_aa_weights(side, w)
        n_in, n_out = (side, 128) if side > 128 else (128, side)
        sigma = max(0.0, (n_in / n_out - 1) / 2)
        lw = int(4.0 * float(sigma) + 0.5)
        assert r == lw, side
        if lw == 0:
            continue
        n += 1
        x = np.arange(-lw, lw + 1)
        phi = np.array([math.exp(-0.5 / (sigma * sigma) * float(v * v)) for v in x])
        phi = phi / phi.sum()
        assert np.array_equal(np.array(w[:lw + 1]), phi[lw:]), side
        k = _filters._gaussian_kernel1d(sigma, 0, lw)[::-1]
        assert np.abs(np.array(w[:lw + 1]) - k[lw:]).max() <= 1e-15 * k.max(), side     # a few ulp: exp() and the normalising sum
    assert n > 250
    assert L.p2p_aa_weights(128, (C.c_double * 256)()) == 0 and L.p2p_aa_weights(0, (C.c_double * 256)()) == -1


@settings(max_examples=30, deadline=None)
@given(st.integers(130, 400), st.sampled_from(["reflect", "constant"]), st.sampled_from([0.0, 0.5, 1.0]), st.booleans())
def test_resize_restatement_anti_aliasing_and_clip(n_in, mode, cval, f32):
    """Down-scaling with anti_aliasing: Gaussian filter (scipy itself) then the bilinear warp == gaussian_filter + zoom of
    current scipy (what scikit-image >= 0.19 literally does for float images); clip keeps the result inside the filtered
    image's range, and a float32 map is filtered in float32 like scipy does for skimage."""
    from scipy import ndimage as ndi
    from oracle.est_pose_oracle import resize_bilinear
    a = np.random.RandomState(n_in).rand(n_in, n_in)
    if f32:
        a = a.astype(np.float32)
    ndi_mode = "mirror" if mode == "reflect" else "constant"
    sig = (n_in / 128 - 1) / 2
    filt = ndi.gaussian_filter(a, (sig, sig), cval=cval, mode=ndi_mode)
    assert filt.dtype == a.dtype
    z = ndi.zoom(filt.astype(np.float64), 128 / n_in, order=1, mode="mirror" if mode == "reflect" else "grid-constant", cval=cval, grid_mode=True)
    r = resize_bilinear(a, (128, 128), mode, cval, anti_aliasing=True)
    if z.shape == (128, 128):
        zc = np.clip(z, filt.min(), filt.max())
        # a float32 map is warped in float32 by the 0.17 / 0.18 generation (coordinates, taps, result: 1e-7 from the double warp);
        # with keep_float32=False the restatement is the all-double warp zoom() performs
        assert np.abs(zc - r).max() < (5e-7 if f32 else 1e-12)
        assert np.abs(zc - resize_bilinear(a, (128, 128), mode, cval, anti_aliasing=True, keep_float32=False)).max() < 1e-12
    assert r.min() >= float(filt.min()) and r.max() <= float(filt.max())


def test_resize_clip_preserves_cval_like_skimage():
    """clip=True: up-scaling a map whose values all sit below cval -- border pixels mix cval in, are clipped back to the
    map's maximum, and pixels exactly equal to cval are left alone (skimage _clip_warp_output)."""
    from oracle.est_pose_oracle import resize_bilinear
    a = np.full((8, 8), 0.25)
    a[3, 3] = 0.3
    r = resize_bilinear(a, (20, 20), "constant", 1.0)
    assert r.max() == 0.3 and r[0, 0] == 0.3                 # the corner mixed cval = 1 in; clipped back to [0.25, 0.3]
    r2 = resize_bilinear(a, (20, 20), "constant", 1.0, clip=False)
    assert r2[0, 0] > 0.5
    b = np.ones((8, 8))
    r3 = resize_bilinear(b, (20, 20), "constant", 0.0)       # all-ones mask, cval 0 outside its range: everything clips up to 1
    assert r3.min() == 1.0
