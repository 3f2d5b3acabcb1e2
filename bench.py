#!/usr/bin/env python
"""bench.py -- the Pix2Pose hot path on MI355X: crops/sec (AE fwd + PnP-RANSAC) at 128x128.

One "step" = one pass of the hot path over one batch of synthetic detections per GPU:
BASELINE.json configs[2] -- 256 detections (128x128 crops, LM-O-like object, outlier_th
[0.2,0.3,0.35], inlier_th 0.2): stage-1 generator pass (256 inputs), stage-2 generator pass
(768 inputs), 768 EPnP-RANSAC solves, candidate selection.  Frames, weights and the injected
decoder maps are resident in HBM before the timed region (SURVEY.md section 8d explains why the
decoder outputs are injected: no trained weights exist offline; the generator passes still run
and are timed).  `value` = detections ("crops") per second over all ranks; each costs 4 generator
forwards (10.70 GFLOP each) + 3 PnP-RANSAC solves.

Multi-GPU: one process per GPU, detections sharded with no data-path collective; the final
(R, t, score) records are all-gathered with RCCL inside the timed step.  Weak scaling.
`python bench.py --gpus N` launches its own ranks (torch.distributed.run on 127.0.0.1) when it is
not already running under a launcher; rank 0 prints the one JSON line.

    python bench.py --gpus 1 --steps 5 --warmup 2

Extra legs (N=1, after the timed region; none of them changes `value`):
  f32_mode          the same steps with the strict-fp32 generator (fp32 MFMA), its own roofline
  host_frames_value frames handed over as HOST uint8 arrays, different frames every step, H2D inside the timed region
                    (the reference's boundary: est_pose(rgb, bbox) takes a numpy frame, recognition.py:70)
  single_det_ms     latency of ONE shim est_pose() call (batch of 1, masks returned), median of 100; drop_in_loop_value = detections/s of the
                    reference's own loop shape -- est_pose once per roi (tools/5_evaluation_bop_basic.py:289-304)
  contract          host frames in AND the full return tuple out (valid_mask, img_pred) at the headline's K steps
  general_crops     the same 256-detection step with bbox sides ~U(40, 300) px (non-identity resizes, n = side^2 correspondences), with and
                    without the anti-aliasing filter of scikit-image 0.17 - 0.18
  pose_delta_vs_oracle  the CPU-baseline sample's detections also run on the GPU: max |dt| (mm), max rotation delta (deg), exact integer matches
  cpu_baseline      the CPU restatement on the host cores (tuned torch/oneDNN fp32 network + numpy/C glue and PnP)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

AE_GFLOP = {"resnet50": 10.70, "paper": 12.58}        # SURVEY.md section 8a-L / BASELINE.md section 2
PEAK_F32_MFMA_TFLOPS = 157.3                           # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_F16_MFMA_TFLOPS = 2500.0                          # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak
PEAK_HBM_TBPS = 8.0                                    # MI355X_MICROARCH.md: HBM3E peak
POWER_WALL_F16X3_TFLOPS = 1497.0                       # v_mfma_f32_32x32x16_f16 out of registers, 12 waves per CU, THIS arithmetic's operand mix: profiles/r04_mfma_power_wall.txt (tools/mfma_f16_wall.hip)
STREAM_HBM_TBPS = 5.9                                  # what a plain 2-reads-1-write elementwise stream sustains on this chip (tools/bw_probe.py: copy 5.45, add 5.91)
TH_O, TH_I = [0.2, 0.3, 0.35], 0.2                     # cfg/cfg_bop2020.json:8-9


# ------------------------------------------------------------------------------------------ CPU baseline
def _usable_cpus():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota (a container that sees 256
    CPUs but owns a quota of 32 runs 256 threads 8x oversubscribed -- round 1's 2.5 s per input)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for fn in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, p = open(fn).read().split()[:2]
            if q != "max":
                n = max(1, min(n, int(float(q) / float(p) + 0.5)))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = max(1, min(n, int(q / p + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(backbone, weights, n_sample):
    """The CPU restatement of the reference path timed on this host, on a bounded sample of the same workload:
    n_sample detections = 4*n_sample generator forwards + 3*n_sample PnP-RANSAC solves + the numpy glue.
    Network leg: oracle/ae_torch.py (torch-CPU = oneDNN, fp32, batch 64 -- the closest thing here to Keras-on-CPU), with
    the thread count tuned on this host (all SMT threads is rarely the best, and a cgroup quota makes it pathological).
    Glue + PnP leg: oracle/est_pose_oracle.py + pnp_oracle.c, one detection after the other like the reference
    (tools/5_evaluation_bop_basic.py:289-304).  value = n_sample / (network time + glue time)."""
    import torch
    from oracle import ae_torch, est_pose_oracle
    from pix2pose_amd import synthetic
    sc = synthetic.make_scene(n_sample, seed=12345)

    # -- glue + PnP leg (also records the network inputs the reference would have fed to predict())
    recorded = []
    refs = []
    t0 = time.time()
    for i in range(n_sample):
        def predict(x, stage, slots=None, i=i):
            recorded.append(np.asarray(x, np.float32))
            m = sc["inject1"][i][None] if stage == 1 else sc["inject2"][i][slots]
            return [m[..., :3].copy(), m[..., 3:].copy()]
        img_i, _, bbox, K = sc["dets"][i]
        refs.append(est_pose_oracle.est_pose(sc["images"][img_i], bbox, predict, K, sc["obj_param"], TH_O, TH_I))
    t_glue = time.time() - t0
    x_all = np.concatenate(recorded, 0)

    # -- network leg: pick the thread count on a small batch, then time the recorded inputs in batches of 64
    ncpu = _usable_cpus()
    cands = sorted({c for c in (8, 16, 32, 64, ncpu // 2, ncpu) if 1 <= c <= ncpu} | {min(ncpu, 8)})
    probe = x_all[:8]
    best, best_t, tuning = cands[0], float("inf"), {}
    with torch.no_grad():
        for c in cands:                      # ascending; stop once more threads make it clearly slower
            torch.set_num_threads(c)
            ae_torch.forward(weights, probe[:2], backbone, dtype=torch.float32)           # warm up (oneDNN primitive cache)
            dt = float("inf")
            for _ in range(2):
                t1 = time.time()
                ae_torch.forward(weights, probe, backbone, dtype=torch.float32)
                dt = min(dt, time.time() - t1)
            tuning[c] = round(dt, 3)
            if dt < best_t:
                best, best_t = c, dt
            elif dt > 1.5 * best_t:
                break
        torch.set_num_threads(best)
        ae_torch.forward(weights, x_all[:64], backbone, dtype=torch.float32)              # warm up at the timed batch size
        t1 = time.time()
        for b0 in range(0, len(x_all), 64):
            ae_torch.forward(weights, x_all[b0:b0 + 64], backbone, dtype=torch.float32)
        t_net = time.time() - t1
    cpu_baseline.sample = (sc, refs)          # the same detections go through the GPU path for pose_delta_vs_oracle
    return {"value": n_sample / (t_net + t_glue), "unit": "crops/s", "cores": best, "kind": "port",
            "host_cpus_visible": os.cpu_count(), "host_cpus_usable": ncpu, "thread_tuning_s_per_8_inputs": tuning,
            "network_inputs_per_s": len(x_all) / t_net, "glue_pnp_detections_per_s": n_sample / t_glue,
            "sample": "%d detections of the same synthetic workload = %d generator forwards (torch-CPU/oneDNN fp32, batch 64, %d threads "
                      "chosen by a probe over %s: %.1f s) + the sequential numpy/C glue with %d OpenCV-style EPnP-RANSAC solves (1 thread: %.1f s)"
                      % (n_sample, len(x_all), best, sorted(tuning), t_net, 3 * n_sample, t_glue)}


# ------------------------------------------------------------------------------------------ launcher
def _self_launch(n):
    """`python bench.py --gpus N` from a plain shell: start N ranks of this script under torch.distributed.run
    (one process per GPU, rendezvous on 127.0.0.1) and become that launcher."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execve(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="detections per GPU per step")
    ap.add_argument("--backbone", default="resnet50")
    ap.add_argument("--chunk", type=int, default=1024, help="generator inputs per pass (activation workspace: 13 MB per input; 1024 holds the 768 stage-2 inputs of a 256-detection batch in one pass)")
    ap.add_argument("--cpu-sample", type=int, default=32, help="detections in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-inject", action="store_true", help="let PnP consume the random-weight generator output")
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f32"],
                    help="generator arithmetic: fp32 emulated with 3 split-f16 MFMAs (default) or fp32 MFMA")
    ap.add_argument("--objects", type=int, default=1, help="number of object models the detections are spread over "
                    "(BASELINE.json configs[3] uses 30; the headline config uses 1)")
    ap.add_argument("--blocking", action="store_true", help="one blocking p2p_est_pose_batch per step instead of the default detection-stream "
                    "mode (p2p_est_pose_submit / collect, two batches in flight: the PnP-RANSAC tail, the D2H copy and the pose gather "
                    "of step i run on a second HIP stream under the generator passes of step i+1; all K steps complete inside the timed region)")
    ap.add_argument("--overlap", action="store_true", help="(default; kept for old command lines)")
    ap.add_argument("--inflight", type=int, default=2, help="stream mode: batches in flight before the oldest is collected (the library holds 2)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL over xGMI; gloo for dry runs)")
    ap.add_argument("--same-device", action="store_true", help="debug: all ranks share cuda:0 (needs --backend gloo)")
    ap.add_argument("--torch-gather", action="store_true", help="gather the pose records with torch.distributed (all_gather_into_tensor) instead of "
                    "the C ABI's own RCCL all-gather (p2p_est_pose_collect_gathered; the default with --backend nccl in stream mode)")
    ap.add_argument("--collective", action="store_true", help="create the process group and run the pose all-gather even with one rank "
                    "(exercises the RCCL path -- communicator with device_id, device-side all_gather_into_tensor -- on a single GPU)")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the ranks of a node to disjoint CPU slices")
    ap.add_argument("--masks", action="store_true", help="also return valid_mask / img_pred of every detection (the full reference tuple) inside the timed region")
    ap.add_argument("--f32-steps", type=int, default=3, help="steps of the strict-fp32 leg (N=1, single object; 0 = skip)")
    ap.add_argument("--host-frames", type=int, default=-1, help="steps of the host-frame and contract legs (N=1; -1 = as many as --steps, 0 = skip)")
    ap.add_argument("--general", type=int, default=-1, help="steps of the general-crop-size legs (N=1; -1 = as many as --steps, 0 = skip)")
    ap.add_argument("--latency", type=int, default=100, help="calls of the single-detection latency leg (N=1; 0 = skip)")
    ap.add_argument("--winograd", default="auto", choices=["auto", "off", "always"],
                    help="form of the 5x5 layers deconv1-3 / up1-3 / conv4 (p2p_ctx_set_winograd): auto = the fastest form per pass size, i.e. Winograd F(4,5) / F(4,3) "
                         "for every pass of this workload (default), off = the direct kernels (the arithmetic of rounds 1-5)")
    ap.add_argument("--merge", action="store_true", help="stream mode: merge step i's stage-2 generator pass with step i+1's stage-1 pass (p2p_est_pose_opts.merge_stream_passes)")
    ap.add_argument("--anti-aliasing", action="store_true", help="scikit-image 0.17 - 0.18 resize semantics (Gaussian pre-filter whenever a resize shrinks)")
    ap.add_argument("--bbox-side", default="86,86", help="range of detection box sides in px (default 86 = 128-px crops, the headline workload; "
                    "e.g. 40,300 for general crop sizes -- profiling runs)")
    ap.add_argument("--batch64", type=int, default=20, help="passes of the configs[1] leg (64-input generator forward only, N=1; 0 = skip)")
    ap.add_argument("--no-legs", action="store_true", help="skip the f32 / host-frame / latency / CPU legs (profiling runs)")
    args = ap.parse_args()
    if args.no_legs:
        args.f32_steps = args.host_frames = args.latency = args.cpu_sample = args.general = args.batch64 = 0
    if args.host_frames < 0:
        args.host_frames = args.steps
    if args.general < 0:
        args.general = args.steps
    args.overlap = not args.blocking
    args.inflight = max(1, min(args.inflight, 2))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args.gpus)                                   # does not return

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.collective
    pinned = None
    if world > 1 and not args.no_pin:
        from pix2pose_amd.parallel import pin_rank_to_cpus
        # ranks of THIS node share its CPUs: LOCAL_WORLD_SIZE (torchrun), not the global world size; the affinity is inherited by every
        # thread the process starts later (torch, the HIP runtime)
        pinned = pin_rank_to_cpus(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)), _usable_cpus())
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                       # --collective from a plain shell: a one-rank group on the loopback
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(so.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    coll_dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")

    from pix2pose_amd import _lib, synthetic, weights as W
    from pix2pose_amd.parallel import gather_poses_async, poses_to_records
    from pix2pose_amd.runtime import Context, Generator, ObjectSpec, est_pose_batch, est_pose_submit

    ctx = Context(local_rank, max_batch=args.chunk, winograd=args.winograd)
    wts = W.synthetic_weights(args.backbone, 1)

    def make_specs(precision):
        gen = Generator(wts, args.backbone, ctx, precision=precision)
        return [ObjectSpec(gen, synthetic.OBJ_PARAM, TH_O, TH_I)] + \
               [ObjectSpec(Generator(W.synthetic_weights(args.backbone, 1 + k), args.backbone, ctx, precision=precision), synthetic.OBJ_PARAM, TH_O, TH_I)
                for k in range(1, args.objects)]
    specs = make_specs(args.precision)
    bb_lo, bb_hi = [int(v) for v in args.bbox_side.split(",")]
    sc = synthetic.make_scene(args.batch, seed=1000 + rank, bbox_side=(bb_lo, bb_hi))
    if args.objects > 1:
        sc["dets"] = [(d[0], i % args.objects, d[2], d[3]) for i, d in enumerate(sc["dets"])]
    frames = torch.from_numpy(sc["images"]).cuda()
    images = [(frames[i].data_ptr(), frames.shape[1], frames.shape[2], "u8") for i in range(frames.shape[0])]
    inj1 = torch.from_numpy(sc["inject1"]).cuda()
    inj2 = torch.from_numpy(sc["inject2"]).cuda()
    torch.cuda.synchronize()
    kw = {} if args.no_inject else dict(inject1=inj1.data_ptr(), inject2=inj2.data_ptr(), inject_slots=3)

    # Pose gather: with the RCCL backend the records go through the library's own communicator (C ABI: p2p_comm_create /
    # p2p_est_pose_collect_gathered -- ncclAllGather on device-resident records, behind the batch's tail on its stream);
    # torch.distributed only hands the communicator id round.  Blocking steps and the gloo dry runs keep the torch gather.
    comm, comm_note = None, None
    if use_dist and args.backend == "nccl" and not args.torch_gather:
        try:
            from pix2pose_amd.parallel import create_comm, gathered_to_records
            comm = create_comm(ctx)
            from pix2pose_amd.runtime import Comm as _Comm
            comm_note = "C ABI ncclAllGather (%s)" % _Comm.library()
        except Exception as e:      # noqa: BLE001 -- an RCCL that cannot be bound: the torch path still works
            comm, comm_note = None, "torch.distributed (C ABI communicator unavailable: %s)" % (e,)
        # the choice must be the same on every rank (a rank gathering through torch while the others sit in ncclAllGather would hang)
        agree = torch.tensor([1 if comm is not None else 0], device=coll_dev, dtype=torch.int32)
        dist.all_reduce(agree, op=dist.ReduceOp.MIN)
        if int(agree.item()) == 0 and comm is not None:
            comm.close()
            comm, comm_note = None, "torch.distributed (another rank could not create the C ABI communicator)"
    submit_s = []           # host seconds inside every est_pose_submit call (marshalling + enqueueing one batch)
    gather_q = []           # all-gathers in flight: started when a step is collected, waited for one step later

    def finish(poses, last=False):
        """Pose records of a collected step -> the node-wide all-gather (RCCL over xGMI).  The gather of step i is only WAITED for when
        step i + 1 is collected (or at the end of the run), so a rank that is late does not stall the others' next submit."""
        rec = poses_to_records(poses, base_id=rank * args.batch)
        if not use_dist:
            return poses, rec
        gather_q.append(gather_poses_async(rec, device=coll_dev, pad_to=args.batch))
        while len(gather_q) > (0 if last else 1):
            rec = gather_q.pop(0).result()
        return poses, rec

    def run_steps(k, specs=specs, images_of_step=None, blocking=None, masks=None, dets=None, kw=kw, aa=None):
        """k steps.  Blocking: one p2p_est_pose_batch per step.  Default: detection-stream mode -- step i+1 is enqueued
        before step i is collected, so the PnP-RANSAC tail (second HIP stream), the D2H and the pose gather overlap the
        next step's generator passes; every step's work still completes inside the call."""
        blocking = (not args.overlap) if blocking is None else blocking
        masks = args.masks if masks is None else masks
        imgs = (lambda i: images) if images_of_step is None else images_of_step
        dets = sc["dets"] if dets is None else dets
        aa = args.anti_aliasing if aa is None else aa
        out = None
        if blocking:
            for i in range(k):
                out = finish(est_pose_batch(ctx, specs, imgs(i), dets, want_masks=masks, anti_aliasing=aa, **kw)[0], last=i == k - 1)
            return out
        pending = []
        for i in range(k):
            t_s = time.perf_counter()
            pending.append(est_pose_submit(ctx, specs, imgs(i), dets, want_masks=masks, merge_passes=args.merge, anti_aliasing=aa, **kw))
            submit_s.append(time.perf_counter() - t_s)
            if len(pending) >= args.inflight:
                out = collect(pending.pop(0))
        while pending:
            pb = pending.pop(0)
            out = collect(pb, last=not pending)
        return out

    def collect(pb, last=False):
        if comm is None:
            return finish(pb.collect(), last=last)
        poses, allp = pb.collect_gathered(comm, args.batch)
        return poses, gathered_to_records(allp, world, args.batch, args.batch)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def profiled_blocking_steps(k, specs=specs):
        """Per-kernel figures: HIP events around every launch of k blocking steps (one batch on the GPU at a time)."""
        ctx.profile(True)
        ctx.profile_read(reset=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(k, specs=specs, blocking=True, masks=False)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t1
        st = ctx.profile_read(reset=True)
        ctx.profile(False)
        return st, dtp

    if args.warmup:
        run_steps(args.warmup)
    if args.blocking:
        ctx.profile(True)
        ctx.profile_read(reset=True)
    barrier()
    t0 = time.perf_counter()
    poses, rec = run_steps(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if args.blocking:
        stats = ctx.profile_read(reset=True)
        ctx.profile(False)
        prof_dt, prof_note = dt, "HIP events around every launch of the timed region"
    else:
        # Stream mode keeps kernels of two batches on the GPU at once, so a launch's event-to-event time there is not the
        # kernel's own duration.  The per-kernel figures come from PROF_STEPS extra blocking steps right after the timed region
        # (same data, same kernels, one batch at a time); `python bench.py --blocking` measures them inside the timed region.
        PROF_STEPS = 2
        stats, _ = profiled_blocking_steps(PROF_STEPS)
        # shares: kernel time of the instrumented steps over the wall time of the same number of UN-instrumented blocking steps (the events
        # around every launch stretch a step from ~34 to ~50 ms; rocprofv3 of `--blocking` agrees with the un-instrumented figure)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(PROF_STEPS, blocking=True, masks=False)
        torch.cuda.synchronize()
        prof_dt = time.perf_counter() - t1
        prof_note = ("HIP events around every launch of %d extra blocking steps after the timed region (in stream mode kernels of two "
                     "batches overlap, which inflates per-launch times); shares = those launches' time / wall time of %d un-instrumented "
                     "blocking steps" % (PROF_STEPS, PROF_STEPS))
    host_submit_ms = float(np.median(submit_s[-args.steps:]) * 1e3) if submit_s else None
    if use_dist:
        tt = torch.tensor([dt], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    n_ok = sum(1 for p in poses if p.status == 0)
    errs = [synthetic.pose_error(sc["gt"][i][0], sc["gt"][i][1], np.array(p.R).reshape(3, 3), np.array(p.t))
            for i, p in enumerate(poses) if p.status == 0]
    total = world * args.batch * args.steps
    value = total / dt

    def roofline_of(stats, prof_dt, precision, note):
        dom = max(range(len(stats)), key=lambda i: stats[i]["total_ms"])     # dominant kernel family
        s0 = stats[dom]
        prec_id = 1 if precision == "f16x3" else 0
        dom_label, dom_name = _lib.PROFILE_KERNELS[dom]
        if "%d" in dom_name:
            dom_name = dom_name % prec_id
        ach = s0["algo_flops"] / (s0["total_ms"] * 1e-3) / 1e12 if s0["total_ms"] > 0 else 0.0
        all_ms = sum(s["total_ms"] for s in stats)
        # peak in the same unit as `achieved` (ALGORITHMIC FLOPs): the f16x3 arithmetic spends three dense-f16 MFMA products per
        # algorithmic MAC, so its ceiling is the dense f16 peak / 3; the fp32 mode is priced against the fp32-MFMA peak
        # The Winograd form of the 5x5 layers (profile slot 10) forms 10 products where the direct form has 25: 3 x 10 / 25 = 1.2 MFMA FLOPs
        # per algorithmic FLOP (algo_flops stays the DIRECT form's 2 x MACs of the layer)
        # (the F(4,3) form of the transposed convolutions, slot 20: 15 position-products per input pixel where the four phases have 25: 1.8)
        mfma_per_algo = {10: 1.2, 20: 1.8}.get(dom, 3.0) if precision == "f16x3" else 1.0
        peak = PEAK_F16_MFMA_TFLOPS / mfma_per_algo if precision == "f16x3" else PEAK_F32_MFMA_TFLOPS
        r = {"bound": "mfma", "kernel": "%s: %s" % (dom_name, dom_label),
             "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
             "frac_algorithmic": ach / (PEAK_F16_MFMA_TFLOPS if precision == "f16x3" else PEAK_F32_MFMA_TFLOPS),
             "peak_dense_f16_mfma": PEAK_F16_MFMA_TFLOPS if precision == "f16x3" else None,
             # what the chip sustains when it does NOTHING but these MFMAs on operands distributed like the split's (data-dependent power: all-zero
             # operands run at 2231): the ceiling any kernel shape is under, measured with tools/mfma_f16_wall.hip
             "measured_mfma_power_wall_tflops": POWER_WALL_F16X3_TFLOPS if precision == "f16x3" else None,
             "frac_of_measured_power_wall": (mfma_per_algo * ach / POWER_WALL_F16X3_TFLOPS) if precision == "f16x3" else None,
             "mfma_flops_per_algorithmic_flop": mfma_per_algo,
             "note": ("achieved = algorithmic FLOPs (2 x MACs of the layers in their DIRECT form) / launch time.  frac = utilisation of the f16 matrix "
                      "pipe: the split-f16 arithmetic (fp32-equivalent results) issues 3 MFMA products per MAC it forms%s, so peak = 2500 / %.1f and the "
                      "pipe sustains %.0f of its 2500 TFLOP/s.  frac_algorithmic = achieved / 2500 (no credit for the emulation overhead).  The "
                      "power-limited ceiling of dense f16 MFMA on random operands is ~1330 TFLOP/s (cdna_hip_programming.md 5.4 rule 25); the kernel "
                      "delivers %.2fx the fp32-MFMA peak (157.3)" % (", and the Winograd form F(4,5) forms 10 MACs where the direct 5x5 row has 25" if dom == 10 else "",
                                                                    mfma_per_algo, mfma_per_algo * ach, ach / PEAK_F32_MFMA_TFLOPS))
                     if precision == "f16x3" else "fp32 MFMA (v_mfma_f32_32x32x2_f32), peak 157.3 TFLOP/s",
             "avg_launch_ms": s0["total_ms"] / max(s0["launches"], 1), "launches": s0["launches"],
             "algo_gflop_per_launch": s0["algo_flops"] / max(s0["launches"], 1) / 1e9,
             "measured_over": note, "share_of_step_time": s0["total_ms"] * 1e-3 / prof_dt, "all_conv_kernels_share_of_step_time": all_ms * 1e-3 / prof_dt,
             "families": {_lib.PROFILE_KERNELS[i][0]: {"launches": st["launches"], "total_ms": st["total_ms"],
                                                       "algo_tflops": (st["algo_flops"] / (st["total_ms"] * 1e-3) / 1e12 if st["total_ms"] > 0 else 0.0)}
                          for i, st in enumerate(stats) if st["launches"]},
             "traffic": None}
        return r, dom_name

    def hbm_rooflines(stats):
        """north_star: "achieved HBM GB/s for the decoder".  The decoder's 5x5 layers sit at ~780 FLOP/B and are held to the MFMA roofline
        above; the bandwidth-leaning kernel families are priced here: compulsory bytes of their launches (inputs + residual + weights read
        once, outputs written once, fp32; p2p_kernel_stats.algo_bytes) / HIP-event time of the same launches, against 8 TB/s."""
        out = []
        for i, label in ((5, "decoder output heads"), (9, "ResNet identity bottleneck blocks, fused: 1x1 -> 3x3 -> 1x1 + residual in one launch (block input read once + "
                                                          "its 3x3 halo, output written once; the three-launch route moved 2x these bytes)"),
                         (11, "input transform of the Winograd layers (x read once, the split-f16 V -- two positions per input column, 8 bytes per element -- written once)"),
                         (21, "input transform of the Winograd transposed convolutions up2 / up3 (x read once, the split-f16 V -- 6 bytes per element -- written once)"),
                         (0, "1x1 layers of the projection blocks res2a / res3a (+ dense_dec)"), (1, "Cout = 64 1x1 layers (res2a 2a)")):
            st = stats[i]
            if not st["launches"] or st["total_ms"] <= 0:
                continue
            name = _lib.PROFILE_KERNELS[i][1]
            tbps = st["algo_bytes"] / (st["total_ms"] * 1e-3) / 1e12
            out.append({"kernel": (name % 1 if "%d" in name else name), "layers": label, "bound": "hbm", "achieved_TBps": tbps, "peak_TBps": PEAK_HBM_TBPS,
                        "frac": tbps / PEAK_HBM_TBPS, "frac_of_plain_stream": tbps / STREAM_HBM_TBPS, "algo_MB_per_launch": st["algo_bytes"] / st["launches"] / 1e6,
                        "avg_launch_ms": st["total_ms"] / st["launches"], "launches": st["launches"],
                        "algo_tflops": st["algo_flops"] / (st["total_ms"] * 1e-3) / 1e12})
        return out

    roof, dom_name = roofline_of(stats, prof_dt, args.precision, prof_note)
    out = {
        "metric": "crops/sec (AE fwd + PnP-RANSAC) at 128x128", "value": value, "unit": "crops/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16x3 (fp32 operands split into two f16 halves, 3 MFMAs per product block, fp32 accumulate)" if args.precision == "f16x3" else "f32",
        "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[%d]: %d detections/GPU/step, 128x128 crops, %s generator "
                               "(1 stage-1 + 3 stage-2 forwards per detection) + 3 EPnP-RANSAC solves per detection, "
                               "outlier_th=[0.2,0.3,0.35], injected ellipsoid-NOCS decoder maps%s"
                               % (3 if args.objects > 1 else 2, args.batch, args.backbone,
                                  ", %d object models (grouped generator passes)" % args.objects if args.objects > 1 else ""),
                   "detections_per_gpu": args.batch, "backbone": args.backbone, "precision": args.precision, "parallelism": "dp%d" % world,
                   "generator_chunk": args.chunk, "objects": args.objects, "returns_masks": bool(args.masks),
                   "mode": ("stream (submit/collect, %d in flight)" % args.inflight) if args.overlap else "blocking"},
        "ae_inputs_per_s": 4 * value,
        "ae_tflops_per_gpu": 4 * value * AE_GFLOP[args.backbone] / 1e3 / world,
        "gathered_records": int(len(rec)),
        "collective": ({"backend": args.backend, "device": str(coll_dev), "async": comm is None, "world": world} if use_dist else None),
        "gather_impl": comm_note if use_dist else None,
        "host_submit_ms_per_step": host_submit_ms, "rank_cpus": (len(pinned) if pinned else None),
        "poses_ok": n_ok, "ransac_iters_mean_of_selected": float(np.mean([p.ransac_iters for p in poses])),
        "pose_err_vs_gt_median_mm_deg": [float(np.median([e[0] for e in errs])), float(np.median([e[1] for e in errs]))] if errs else None,
        "roofline": roof,
        "roofline_hbm": hbm_rooflines(stats) if args.precision == "f16x3" else None,
    }
    # HBM bytes per launch of the dominant kernel from separate rocprofv3 --pmc passes of this same
    # command (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; tools/pmc_traffic.py), committed under profiles/
    import glob
    tfns = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if tfns:
        tr = json.load(open(tfns[-1]))
        for k, v in tr.items():
            if dom_name in k:
                out["roofline"]["traffic"] = v["hbm_bytes_per_launch"]
                out["roofline"]["traffic_note"] = ("bytes/launch, rocprofv3 PMC (FETCH_SIZE*2 + WRITE_SIZE), profiles/%s" % os.path.basename(tfns[-1]))

    solo = rank == 0 and world == 1
    # -- strict-fp32 leg: the same steps on generators built with P2P_PREC_F32 (fp32 matrix instructions, bitwise an fmaf chain)
    if solo and args.f32_steps > 0 and args.objects == 1 and args.precision == "f16x3":
        specs32 = make_specs("f32")
        run_steps(1, specs=specs32)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(args.f32_steps, specs=specs32)
        torch.cuda.synchronize()
        d32 = time.perf_counter() - t1
        st32, pdt32 = profiled_blocking_steps(1, specs=specs32)
        r32, _ = roofline_of(st32, pdt32, "f32", "HIP events around every launch of 1 blocking step")
        out["f32_mode"] = {"value": args.batch * args.f32_steps / d32, "unit": "crops/s", "steps": args.f32_steps,
                           "ms_per_step": d32 / args.f32_steps * 1e3, "dtype": "f32 (v_mfma_f32_32x32x2_f32)",
                           "roofline": {k: r32[k] for k in ("kernel", "achieved", "peak", "frac", "unit", "avg_launch_ms", "launches", "share_of_step_time")}}
        del specs32
    # -- host-frame leg: the reference's boundary hands over numpy frames (recognition.py:70; caller
    #    tools/5_evaluation_bop_basic.py:272,303-304): 32 frames of 640x480x3 uint8 per step (8 detections each), new frames
    #    every step, pageable host memory, H2D inside the timed region
    if solo and args.host_frames > 0:
        n_fr = 32
        n_sets = min(args.host_frames, 4) + 1            # frame sets cycled through (every step still uploads its frames)
        pool = [np.random.RandomState(77 + s).randint(0, 256, (n_fr,) + sc["images"].shape[1:], dtype=np.uint8) for s in range(n_sets)]
        hdets = [(i % n_fr, d[1], d[2], d[3]) for i, d in enumerate(sc["dets"])]
        for key, masks in (("host_frames", False), ("contract", True)):
            run_steps(2, images_of_step=lambda i: list(pool[-1]), dets=hdets, masks=masks)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run_steps(args.host_frames, images_of_step=lambda i: list(pool[i % (n_sets - 1)]), dets=hdets, masks=masks)
            torch.cuda.synchronize()
            dh = time.perf_counter() - t1
            out[key] = {"value": args.batch * args.host_frames / dh, "unit": "crops/s", "steps": args.host_frames, "ms_per_step": dh / args.host_frames * 1e3,
                        "frames_per_step": n_fr, "frame_bytes_per_step": int(pool[0].nbytes), "returns_masks": masks,
                        "note": ("frames are host uint8 numpy arrays (pageable), handed over anew every step; the library uploads the row ranges the crops "
                                 "cover inside the timed region" + ("; every detection's valid_mask and img_pred come back (the reference's full return "
                                 "tuple, recognition.py:189-193)" if masks else ""))}
        out["host_frames_value"] = out["host_frames"]["value"]
        del pool
    # -- general crop sizes: the same step with bbox sides ~U(40, 300) px: every resize is a real resampling (recognition.py:82,103,121,134-146)
    #    and a candidate carries side^2 correspondences instead of 16 384; with and without the anti-aliasing filter (scikit-image 0.17 - 0.18)
    if solo and args.general > 0:
        scg = synthetic.make_scene(args.batch, seed=2000, bbox_side=(40, 300))
        gj1 = torch.from_numpy(scg["inject1"]).cuda()
        gj2 = torch.from_numpy(scg["inject2"]).cuda()
        gfr = torch.from_numpy(scg["images"]).cuda()
        gimg = [(gfr[i].data_ptr(), gfr.shape[1], gfr.shape[2], "u8") for i in range(gfr.shape[0])]
        gkw = {} if args.no_inject else dict(inject1=gj1.data_ptr(), inject2=gj2.data_ptr(), inject_slots=3)
        torch.cuda.synchronize()
        sides = [2 * int(1.5 * (d[2][2] - d[2][0]) / 2) for d in scg["dets"]]
        out["general_crops"] = {"bbox_side_px": [40, 300], "crop_side_px_mean": float(np.mean(sides)), "steps": args.general, "unit": "crops/s"}
        for key, aa in (("value", False), ("value_anti_aliasing", True)):
            run_steps(2, images_of_step=lambda i: gimg, dets=scg["dets"], kw=gkw, aa=aa)      # both batch slots sized before the timed steps
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            gp, _ = run_steps(args.general, images_of_step=lambda i: gimg, dets=scg["dets"], kw=gkw, aa=aa)
            torch.cuda.synchronize()
            dg = time.perf_counter() - t1
            out["general_crops"][key] = args.batch * args.general / dg
            out["general_crops"]["poses_ok" + ("_anti_aliasing" if aa else "")] = sum(1 for q in gp if q.status == 0)
            if not aa:      # how far RANSAC ran on the selected candidates: hypotheses are solved in rounds [0,16), [16,64), [64,100)
                it = np.array([q.ransac_iters for q in gp if q.status == 0])
                out["general_crops"]["ransac_iters_selected_le16_le32_le64_le100"] = [int((it <= 16).sum()), int(((it > 16) & (it <= 32)).sum()),
                                                                                       int(((it > 32) & (it <= 64)).sum()), int((it > 64).sum())]
            # Roofline entries of the bandwidth- / latency-leaning kernels of this leg: HIP-event times of two extra blocking
            # steps (p2p_profile_read slots 12-19) over bytes computed from the batch's own counts -- the correspondences of every candidate and
            # how far its RANSAC ran come from one extra debug-tap call.
            try:
                _, exg = est_pose_batch(ctx, specs, gimg, scg["dets"], debug=True, anti_aliasing=aa, **gkw)
                cand = exg["cand"].astype(np.int64)                       # [n, K, 6]: valid, n_non_gray, n_corr, n_inliers, ransac iters, best iter
                vmask = cand[..., 0] > 0
                ncorr, iters = cand[..., 2] * vmask, cand[..., 4] * vmask
                hyp = np.where(iters <= 16, 16, np.where(iters <= 64, 64, 100)) * vmask      # hypotheses are counted in rounds [0,16), [16,64), [64,100)
                passes = (hyp + 7) // 8                                    # pnp_count_kernel: one pass over a candidate's points per 8 models
                S2 = np.array([sd * sd for sd in sides], np.float64)
                Kc = cand.shape[1]
                big = np.array([sd > 128 for sd in sides])
                ctx.profile(True); ctx.profile_read(reset=True)
                for _ in range(2):
                    est_pose_batch(ctx, specs, gimg, scg["dets"], anti_aliasing=aa, **gkw)
                gst = ctx.profile_read(reset=True); ctx.profile(False)
                n_corr_total = float(ncorr.sum())
                byt = {13: (float((ncorr * passes).sum()) * 20.0, float((ncorr * passes).sum()) * 4.0, "a candidate's correspondences (X Y Z U V float32 = 20 B; floor: u8 XYZ + mask = 4 B) once per 8 hypotheses of the rounds its RANSAC reached"),
                       14: (n_corr_total * 20.0 * float(np.mean(np.where(iters <= 16, 1, np.where(iters <= 64, 2, 3))[vmask])) if vmask.any() else 0.0, n_corr_total * 4.0, "the correspondences once per round (Gram sums of the best model's inliers)"),
                       15: (n_corr_total * 20.0, n_corr_total * 4.0, "the correspondences once (final inlier mask); the refit itself is a fp64 dependency chain"),
                       18: (float(vmask.sum()) * 16384 * 16.0 + n_corr_total * 20.0, float(vmask.sum()) * 16384 * 16.0 + n_corr_total * 4.0, "the candidate's 128x128x4 float map read once, its correspondences written once"),
                       19: (float((S2[:, None] * vmask).sum()) * 3.0 + float(vmask.sum()) * 16384 * 12.0, None, "the crop's u8 pixels read once, the 128x128x3 float32 network input written once")}
                if aa:
                    fb = float(S2[big].sum()) * (1 + Kc) * 3 * 8 * 2 + float((~big).sum()) * Kc * 5 * 16384 * 8 * 2
                    byt[16] = (fb, None, "every float64 canvas / plane read and written once per axis pass (full canvases: the region-of-interest skip lowers the real traffic)")
                    byt[17] = (fb, None, byt[16][2])
                rl = []
                for slot in sorted(byt):
                    stq = gst[slot]
                    if not stq["launches"] or stq["total_ms"] <= 0:
                        continue
                    b_now, b_floor, what = byt[slot]
                    ms = stq["total_ms"] / 2.0                             # per step
                    rl.append({"kernel": _lib.PROFILE_KERNELS[slot][1], "bound": "hbm", "ms_per_step": ms, "launches_per_step": stq["launches"] / 2.0,
                               "algo_MB_per_step": b_now / 1e6, "achieved_TBps": b_now / (ms * 1e-3) / 1e12, "peak_TBps": PEAK_HBM_TBPS,
                               "frac": b_now / (ms * 1e-3) / 1e12 / PEAK_HBM_TBPS,
                               "floor_MB_per_step": (b_floor / 1e6 if b_floor is not None else None), "bytes": what})
                hy = gst[12]
                out["general_crops"]["roofline_hbm" + ("_anti_aliasing" if aa else "")] = rl
                out["general_crops"]["pnp_ms_per_step" + ("_anti_aliasing" if aa else "")] = {
                    _lib.PROFILE_KERNELS[k][1]: gst[k]["total_ms"] / 2.0 for k in (12, 13, 14, 15) if gst[k]["launches"]}
                out["general_crops"]["correspondences_per_step"] = n_corr_total
                out["general_crops"]["note_rooflines"] = ("pnp_hypotheses_kernel / pnp_fit_solve_kernel are fp64 dependency chains of a few waves (latency-bound: "
                                                          "%.2f ms per step for %d hypothesis launches), not priced against a bandwidth" % (hy["total_ms"] / 2.0, int(hy["launches"] / 2)))
            except Exception as e:                                         # noqa: BLE001 -- a reporting leg must not take the bench line down
                out["general_crops"]["roofline_error"] = repr(e)
        del gj1, gj2, gfr
    # -- BASELINE.json configs[1]: batch = 64 synthetic crops, one object model, generator forward ONLY (p2p_forward_async on device buffers,
    #    HIP events on the context's stream).  Passes of 256 and 768 inputs beside it: a 64-input pass under-fills the short-K layers' launches.
    if solo and args.batch64 > 0:
        gen0 = specs[0].generator
        cst = torch.cuda.ExternalStream(ctx.stream)
        b64 = {"workload": "BASELINE.json configs[1]: 64 synthetic 128x128 crops, 1 object model, generator forward only", "unit": "inputs/s", "passes": args.batch64}
        for n_in in (64, 256, 768):
            if n_in > args.chunk:
                continue
            xin = (torch.randint(0, 256, (n_in, 128, 128, 3), device="cuda", generator=torch.Generator("cuda").manual_seed(5)).float() - 128) / 128
            yout = torch.empty(n_in, 128, 128, 4, device="cuda")
            torch.cuda.synchronize()
            for _ in range(2):
                gen0.forward_device(xin.data_ptr(), n_in, yout.data_ptr())
            ctx.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(cst):
                e0.record(cst)
                for _ in range(args.batch64):
                    gen0.forward_device(xin.data_ptr(), n_in, yout.data_ptr())
                e1.record(cst)
            ctx.synchronize()
            ms = e0.elapsed_time(e1) / args.batch64
            b64["n%d" % n_in] = {"ms_per_pass": ms, "inputs_per_s": n_in / ms * 1e3, "algo_tflops": n_in * AE_GFLOP[args.backbone] / ms}
            del xin, yout
        if "n64" in b64:
            b64["value"] = b64["n64"]["inputs_per_s"]
            if "n256" in b64:
                b64["per_input_rate_vs_256"] = b64["n64"]["inputs_per_s"] / b64["n256"]["inputs_per_s"]
        out["batch64"] = b64
    # -- latency leg: ONE detection through the drop-in shim, masks and image returned like the reference's est_pose
    if solo and args.latency > 0:
        from pix2pose_amd.recognition import pix2pose
        shim = pix2pose({k: v for k, v in wts.items()}, synthetic.LM_K, 640, 480, synthetic.OBJ_PARAM, th_outlier=TH_O, th_inlier=TH_I,
                        backbone=args.backbone, ctx=ctx, skimage="0.14")      # 128-px crops: every resize is the identity in every generation
        lat, n_ret = [], 0
        for i in range(args.latency + 5):
            j = i % args.batch
            if not args.no_inject:
                shim._inject = (inj1[j:j + 1].data_ptr(), inj2[j:j + 1].data_ptr(), 3)
            img_i, _, bbox, K = sc["dets"][j]
            shim.camK = K
            t1 = time.perf_counter()
            r = shim.est_pose(sc["images"][img_i], bbox)
            lat.append(time.perf_counter() - t1)
            n_ret += not (isinstance(r[4], int) and r[4] == -1)
        lat = np.array(lat[5:]) * 1e3
        out["single_det_ms"] = float(np.median(lat))
        out["drop_in_loop_value"] = float(len(lat) / (lat.sum() * 1e-3))       # detections/s of est_pose called once per roi, back to back
        out["single_det"] = {"median_ms": float(np.median(lat)), "p90_ms": float(np.percentile(lat, 90)), "calls": args.latency, "poses_returned": int(n_ret),
                             "note": "recognition.pix2pose.est_pose(rgb, bbox) on one detection: host frame in, (img_pred, valid_mask, R, t, frac_inlier, box) out"}
    if solo and args.cpu_sample > 0:
        out["cpu_baseline"] = cpu_baseline(args.backbone, wts, args.cpu_sample)
        # the sample's detections through the GPU path (same injected maps), against what the oracle returned for them
        scc, refs = cpu_baseline.sample
        cj1 = torch.from_numpy(scc["inject1"]).cuda()
        cj2 = torch.from_numpy(scc["inject2"]).cuda()
        torch.cuda.synchronize()
        gp, _ = est_pose_batch(ctx, specs[:1], list(scc["images"]), scc["dets"], inject1=cj1.data_ptr(), inject2=cj2.data_ptr(), inject_slots=3)
        dts, drs, exact, both, status_mismatches = [], [], 0, 0, 0
        for q, r in zip(gp, refs):
            ok_ref = not (isinstance(r[4], int) and r[4] == -1)
            if (q.status == 0) != ok_ref:                   # one side produced a pose the other rejected: counted, never skipped silently
                status_mismatches += 1
                continue
            if not ok_ref:
                exact += list(q.bbox_t) == [int(v) for v in r[5]]
                continue
            both += 1
            dt_, dr_ = synthetic.pose_error(r[2], r[3], np.array(q.R).reshape(3, 3), np.array(q.t))
            dts.append(dt_); drs.append(dr_)
            exact += (list(q.bbox_t) == [int(v) for v in r[5]]) and (q.frac_inlier == r[4])
        out["pose_delta_vs_oracle"] = {"detections": len(refs), "poses_compared": both, "max_dt_mm": float(max(dts)) if dts else None,
                                       "max_drot_deg": float(max(drs)) if drs else None, "exact_integer_matches": int(exact),
                                       "status_mismatches": int(status_mismatches),       # expected 0; > 0 voids the delta figures
                                       "note": "GPU path vs the CPU restatement on the cpu_baseline sample: pose delta, and detections whose returned box and "
                                               "inlier fraction (n_inliers / n_init_mask: the RANSAC outcome) are identical; north_star bar 1 mm / 1 deg"}
        if status_mismatches:                               # a pose on one side only voids the delta figures (null: `Infinity` is not JSON)
            out["pose_delta_vs_oracle"]["max_dt_mm"] = out["pose_delta_vs_oracle"]["max_drot_deg"] = None
        out["pose_delta_vs_oracle_max_mm_deg"] = [out["pose_delta_vs_oracle"]["max_dt_mm"], out["pose_delta_vs_oracle"]["max_drot_deg"]]
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if comm is not None:
        comm.close()                   # ncclCommDestroy before torch tears its own communicator (and the shared RCCL) down
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
