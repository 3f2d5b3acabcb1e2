"""Counterpart of the reference's RGB-only BOP evaluation driver (tools/5_evaluation_bop_basic.py)
for the MI355X hot path -- SURVEY.md section 8f-1.

Same contract where it touches the hot path: argv ``[gpu_id] [cfg] [dataset]``, the cfg keys
(``backbone``, ``outlier_th`` 1-D => per-detection multi-threshold / 2-D => one fixed threshold per
object, ``inlier_th``, ``score_type``, ``task_type``, ``cand_factor``, ``path_to_output``), per-image
``camK``, candidate limiting, score_type-2 score ``det_score * frac_inlier * mask_iou * union``,
per-image score normalisation + sort + ViVo truncation, and the bop19 CSV
``pix2pose-iccv19_<dataset>-test.csv``.

What differs: the 2D detector is an external repo and out of scope, so detections arrive
*pre-dumped* (BASELINE.json configs[4]: "Mask-RCNN boxes pre-dumped"); and detections of many
images are pooled into one device batch instead of one est_pose call each.

Pre-dumped detections file (JSON)::

    {"im_size": [W, H], "model_ids": [1, 5, ...],
     "norm_factor": {"1": {"x_scale":..,"y_scale":..,"z_scale":..,"x_ct":..,"y_ct":..,"z_ct":..}, ...},
     "weights": {"1": "path/to/obj01.npz" | "synthetic:resnet50:1", ...},
     "targets": [{"scene_id":..,"im_id":..,"obj_id":..,"inst_count":..}, ...],      # test_targets_bop19.json
     "images": [{"scene_id":..,"im_id":..,"rgb": "frame.npy|.png", "cam_K": [9 floats],
                 "rois": [[v1,u1,v2,u2], ...], "obj_ids": [...], "scores": [...], "masks": "masks.npy" (optional, [H,W,n])
                 | "segmentations": [COCO run-length mask per detection]}]}

``bop_dataset.build_dump`` writes this dict from a BOP-format dataset directory (cfg ``dataset_dir``, ``test_target``,
``norm_factor_fn``, ``target_obj`` as in the reference) and a COCO-style detection list; the CLI takes either file.

Several GPUs: under ``torch.distributed.run`` (or with RANK / WORLD_SIZE / LOCAL_RANK set) every rank takes every
WORLD_SIZE-th image of the target list, runs the whole pipeline on its own GPU and rank 0 gathers the result rows (fixed-size
records, one all-gather) and writes the CSV in target-list order -- images are independent in the reference
(tools/5_evaluation_bop_basic.py:229-352), so there is no other exchange.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np


# ---------------------------------------------------------------------------------- config
def model_params_to_obj_param(mp: dict) -> np.ndarray:
    """tools/bop_io.py:33-42: [x_scale, y_scale, z_scale, x_ct, y_ct, z_ct]."""
    return np.array([mp["x_scale"], mp["y_scale"], mp["z_scale"], mp["x_ct"], mp["y_ct"], mp["z_ct"]], np.float64)


def outlier_thresholds(cfg: dict, n_models: int):
    """tools/5_evaluation_bop_basic.py:164-169,217-219: a 1-D list is shared by every object
    (multi-threshold stage 2); a 2-D list gives one fixed threshold per object."""
    th = cfg["outlier_th"]
    if isinstance(th[0], list):
        per = np.squeeze(np.array(th))
        return [[float(per[m])] for m in range(n_models)]
    return [[float(t) for t in th] for _ in range(n_models)]


def group_targets(targets):
    """tools/bop_io.py:9-31 get_target_list: consecutive targets of one image are grouped."""
    out = []
    prev = None
    for tg in targets:
        key = (tg["scene_id"], tg["im_id"])
        if key != prev:
            out.append([tg["scene_id"], tg["im_id"], [], []])
            prev = key
        out[-1][2].append(tg["obj_id"])
        out[-1][3].append(tg["inst_count"])
    return out


def _resize_generation(cfg: dict) -> int:
    """cfg key ``skimage`` ("0.14" | "0.15" | "0.16" | "0.17" | "0.18": the scikit-image the reference environment resolves to -- the
    reference's requirements.txt does not pin it) or the older ``resize_anti_aliasing`` (bool / 0-2).  Neither: generation 0, with a
    warning that it is the one generation no real library in the build image can check."""
    from . import runtime
    if "skimage" in cfg:
        return runtime.resize_generation(str(cfg["skimage"]))
    if "resize_anti_aliasing" in cfg:
        return runtime.resize_generation(cfg["resize_anti_aliasing"])
    runtime.warn_unpinned_generation("eval_bop cfg (key 'skimage')")
    return 0


# ---------------------------------------------------------------------------------- per-image logic
def select_detections(rois, obj_ids, obj_id_targets, inst_counts, cand_factor):
    """tools/5_evaluation_bop_basic.py:289-300: skip (-1,-1) rois and non-target objects; stop
    taking candidates of an object once more than inst_count*cand_factor were taken."""
    pred = np.zeros(len(inst_counts))
    keep = []
    for r_id, roi in enumerate(rois):
        if roi[0] == -1 and roi[1] == -1:
            continue
        obj_id = obj_ids[r_id]
        if obj_id not in obj_id_targets:
            continue
        g = obj_id_targets.index(obj_id)
        if pred[g] > inst_counts[g] * cand_factor:
            continue
        pred[g] += 1
        keep.append(r_id)
    return keep


def detection_score(det_score, frac_inlier, mask_stats, score_type, detect_type="rcnn"):
    """:307-318.  mask_stats = (intersection, union) of the detector mask with valid_mask_full."""
    if score_type == 2 and detect_type == "rcnn":
        inter, union = mask_stats
        mask_iou = 0 if union <= 0 else inter / union
        return det_score * frac_inlier * mask_iou * union
    return det_score


def rank_image_results(results, obj_id_targets, inst_counts, task_type, scene_id, im_id, time_spend):
    """:325-349.  results: list of dict(obj_id, score, R, t).  Normalise by the max score, sort
    descending, apply the ViVo instance limits (only when task_type is the *string* '2', exactly as
    the reference compares it)."""
    if not results:
        return []
    score = np.array([r["score"] for r in results], np.float64)
    score = score / np.max(score)
    order = np.argsort(1 - score)
    est = np.zeros(len(inst_counts))
    total, n_inst = 0, np.sum(inst_counts)
    rows = []
    for rid in order:
        r = results[rid]
        g = obj_id_targets.index(r["obj_id"])
        est[g] += 1
        if task_type == '2' and est[g] > inst_counts[g]:
            continue
        rows.append({"scene_id": scene_id, "im_id": im_id, "obj_id": r["obj_id"], "score": score[rid],
                     "R": np.asarray(r["R"]).flatten(), "t": np.asarray(r["t"]).flatten(), "time": time_spend})
        total += 1
        if task_type == '2' and total > n_inst:
            break
    return rows


def save_bop_results(path, rows):
    """bop_toolkit inout.save_bop_results, version 'bop19' (un-vendored; format per SURVEY 8f-1):
    scene_id,im_id,obj_id,score,R,t,time with R (9) and t (3, mm) space separated."""
    lines = ["scene_id,im_id,obj_id,score,R,t,time"]
    for r in rows:
        lines.append("{},{},{},{},{},{},{}".format(r["scene_id"], r["im_id"], r["obj_id"], r["score"],
                                                   " ".join(map(str, np.asarray(r["R"]).flatten().tolist())),
                                                   " ".join(map(str, np.asarray(r["t"]).flatten().tolist())),
                                                   r.get("time", -1)))
    with open(path, "w") as f:
        f.write("\n".join(lines))


def output_name(dataset):
    """:353-356."""
    return "pix2pose-iccv19_%s-test-primesense.csv" % dataset if dataset == "tless" else "pix2pose-iccv19_%s-test.csv" % dataset


def _load_frame(path):
    if path.endswith(".npy"):
        return np.load(path)
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("reading %s needs Pillow; dump frames as .npy instead" % path) from e
    a = np.array(Image.open(path))
    if a.ndim == 2:                      # gray datasets: copy to three channels (:260-266)
        a = np.repeat(a[:, :, None], 3, axis=2)
    return a[:, :, :3]


class FramePrefetcher:
    """Decodes the frames of upcoming chunks on a few threads (PNG decoding releases the GIL; one 640x480 frame costs ~5 ms,
    a 32-image chunk 160 ms on one thread -- more than the GPU needs for its 256 detections).  request() schedules,
    get() returns the decoded frame (decoding it now if it was never requested) and forgets it."""

    def __init__(self, threads: int = 8):
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=max(1, int(threads))) if threads and threads > 0 else None
        self._pending = {}

    def request(self, paths):
        if self._pool is None:
            return
        for p in paths:
            if p not in self._pending:
                self._pending[p] = self._pool.submit(_load_frame, p)

    def get(self, path):
        f = self._pending.pop(path, None)
        return f.result() if f is not None else _load_frame(path)

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=False, cancel_futures=True)
            self._pool = None


# ---------------------------------------------------------------------------------- driver
def run(cfg: dict, dataset: str, dump: dict, device: int = 0, base_dir: str = ".", batch_images: int = 32,
        detect_type: str = "rcnn", est_pose_kwargs=None, shard=None, write_csv: bool = True, inject=None, make_gather=None):
    """Evaluate a pre-dumped detection stream.  Returns the result rows (also written as CSV when
    cfg['path_to_output'] is set).  shard = (rank, world): only every world-th image of the target list, starting at rank;
    every row carries "_order" = (position of its image in the full target list, rank inside the image) so that the shards'
    rows merge back into the single-process order.
    make_gather (with shard; a callable taking the runtime Context -> parallel.CabiPoseGather / TorchPoseGather): LOCKSTEP mode.
    Every rank walks the same number of steps (the longest shard's), and every pooled batch is collected through ONE collective of
    pose records per step -- p2p_est_pose_collect_gathered over RCCL in the C ABI -- which a rank whose images have run out (or whose
    chunk holds no detection: the ragged loop of tools/5_evaluation_bop_basic.py:289-323) joins as an empty shard.  The gathered
    records carry the score_type-2 mask sums, and every rank holds the whole dump, so every rank builds the rows of ALL ranks: the
    returned rows are then the complete, merged result on every rank.  inject (tests): {"key": [N, 2] (image position in the target list, detection
    index in the image), "inject1": [N,128,128,4], "inject2": [N,K,128,128,4]} -- decoder maps that replace the generator output
    of the listed detections, the way bench.py and the parity tests drive the pipeline with random-weight networks."""
    from . import runtime, weights as W
    backbone = cfg.get("backbone", "paper")                                   # :202-205
    model_ids = list(dump["model_ids"])
    th_o = outlier_thresholds(cfg, len(model_ids))
    th_i = cfg["inlier_th"]
    score_type, task_type, cand_factor = cfg["score_type"], cfg["task_type"], float(cfg["cand_factor"])
    ctx = runtime.Context(device, max_batch=int(cfg.get("generator_chunk", 256)))
    specs = []
    for m, mid in enumerate(model_ids):                                       # one network per object (:206-225)
        wfn = dump["weights"][str(mid)]
        if not wfn.startswith(("synthetic:", "trained-like:")):
            wfn = os.path.join(base_dir, wfn)
        gen = runtime.Generator(W.load_weights(wfn, backbone), backbone, ctx)
        specs.append(runtime.ObjectSpec(gen, model_params_to_obj_param(dump["norm_factor"][str(mid)]), th_o[m], th_i))
    by_image = {(im["scene_id"], im["im_id"]): im for im in dump["images"]}
    rows = []
    full_tlist = [t + [gi] for gi, t in enumerate(group_targets(dump["targets"]))]
    tlist = full_tlist[shard[0]::shard[1]] if shard is not None else full_tlist

    def prepare(chunk):
        frames, dets, det_masks, owners = [], [], [], []
        for ti, (scene_id, im_id, obj_id_targets, inst_counts, _gi) in enumerate(chunk):
            im = by_image.get((scene_id, im_id))
            if im is None:
                continue
            frame = loader.get(os.path.join(base_dir, im["rgb"]))
            masks = np.load(os.path.join(base_dir, im["masks"])) if im.get("masks") else None
            segs = im.get("segmentations") if masks is None else None
            fi = len(frames)
            frames.append(frame)
            for r_id in select_detections(im["rois"], im["obj_ids"], obj_id_targets, inst_counts, cand_factor):
                dets.append((fi, model_ids.index(im["obj_ids"][r_id]), [int(v) for v in im["rois"][r_id]], np.array(im["cam_K"], float).reshape(3, 3)))
                owners.append((ti, r_id))
                if score_type == 2 and detect_type == "rcnn":
                    if masks is not None:
                        det_masks.append(masks[:, :, r_id])
                    elif segs is not None and segs[r_id] is not None:
                        from .bop_dataset import rle_decode
                        det_masks.append(rle_decode(segs[r_id]))
                    else:
                        raise ValueError("score_type 2 needs detector masks for scene %s image %s" % (scene_id, im_id))
        return frames, dets, det_masks, owners

    def plan(chunk):
        """(image index in the chunk, detection index in the image) of the detections prepare() will submit for it, in its order --
        a function of the dump alone, so every rank can plan every other rank's chunks."""
        owners = []
        for ti, (scene_id, im_id, obj_id_targets, inst_counts, _gi) in enumerate(chunk):
            im = by_image.get((scene_id, im_id))
            if im is None:
                continue
            owners += [(ti, r_id) for r_id in select_detections(im["rois"], im["obj_ids"], obj_id_targets, inst_counts, cand_factor)]
        return owners

    def rows_of(chunk, owners, recs, dt):
        """recs[k]: the pose record of owners[k] (ctypes p2p_pose or a row of a _lib.POSE_DTYPE array)."""
        per_image = {}
        for k, (ti, r_id) in enumerate(owners):
            p = recs[k]
            if int(p["status"]) == -2:                                        # _lib.POSE_RANGE: the producing rank's batch left the split-f16 operand range
                range_events.append(tuple(chunk[ti][:2]))                     # (every rank sees the same gathered records: all of them raise at the end)
            if int(p["status"]) != 0:                                         # frac_inlier == -1 (:305-306)
                continue
            scene_id, im_id = chunk[ti][:2]
            im = by_image[(scene_id, im_id)]
            ms = (int(p["mask_stats"][0]), int(p["mask_stats"][1])) if score_type == 2 and detect_type == "rcnn" else None
            sc = detection_score(im["scores"][r_id], float(p["frac_inlier"]), ms, score_type, detect_type)
            per_image.setdefault(ti, []).append({"obj_id": im["obj_ids"][r_id], "score": sc,
                                                 "R": np.array(p["R"], np.float64).reshape(3, 3), "t": np.array(p["t"], np.float64)})
        for ti, (scene_id, im_id, obj_id_targets, inst_counts, gi) in enumerate(chunk):
            n_here = sum(1 for o in owners if o[0] == ti)
            new = rank_image_results(per_image.get(ti, []), obj_id_targets, inst_counts, task_type, scene_id, im_id,
                                     dt * n_here / max(len(owners), 1))          # batch time amortised over its detections
            for k, r in enumerate(new):
                r["_order"] = (gi, k)
            rows.extend(new)

    def finish(job):
        chunk, owners, pending, t1, _held, step = job    # _held: injected maps stay alive until the batch is collected
        from . import _lib
        if pending is not None:
            assert len(plan(chunk)) == pending.n, "plan() and prepare() disagree on the detections of this step (%d vs %d)" % (len(plan(chunk)), pending.n)
        if gather is None:
            pending.collect()
            rows_of(chunk, owners, np.frombuffer(pending.pose_array, dtype=_lib.POSE_DTYPE, count=max(pending.n, 1)), time.time() - t1)
            return
        try:
            _own, allp = gather(pending, n_max)           # collective: every rank, every step; pending is None for an empty shard
        except Exception as e:                            # noqa: BLE001 -- reported after the collective ran (the library joins with padding records):
            step_errors.append(e)                         # keep walking the steps so that the peers' remaining collectives are joined, re-raise at the end
            return
        dt = time.time() - t1
        for r in range(shard[1]):                          # every rank builds every rank's rows of this step from the gathered records
            ch = chunks_of(r)[step] if step < len(chunks_of(r)) else []
            rows_of(ch, plan(ch), allp[r * n_max:(r + 1) * n_max], dt)

    # detection stream: chunk i+1 is read from disk and enqueued (p2p_est_pose_submit) while chunk i is on the GPU;
    # the score_type-2 mask sums come back through the same asynchronous call
    loader = FramePrefetcher(int(cfg.get("loader_threads", 8)))

    def frame_paths(chunk):
        return [os.path.join(base_dir, by_image[(c[0], c[1])]["rgb"]) for c in chunk if (c[0], c[1]) in by_image]

    step_errors, range_events = [], []
    gather, n_max, n_steps = None, 0, (len(tlist) + batch_images - 1) // batch_images
    _chunks = {}

    def chunks_of(r):
        if r not in _chunks:
            tl = full_tlist[r::shard[1]]
            _chunks[r] = [tl[b:b + batch_images] for b in range(0, len(tl), batch_images)]
        return _chunks[r]

    if make_gather is not None and shard is not None:
        gather = make_gather(ctx)
        n_steps = max(len(chunks_of(r)) for r in range(shard[1]))
        n_max = max([1] + [len(plan(ch)) for r in range(shard[1]) for ch in chunks_of(r)])       # equal on all ranks by construction
    loader.request(frame_paths(tlist[:batch_images]))
    in_flight = []
    n_submitted = 0
    inject_row = {(int(a), int(b)): i for i, (a, b) in enumerate(inject["key"])} if inject is not None else None
    for step in range(n_steps):
        b0 = step * batch_images
        chunk = tlist[b0:b0 + batch_images]
        t1 = time.time()
        loader.request(frame_paths(tlist[b0 + batch_images:b0 + 2 * batch_images]))      # decoded while this chunk is prepared and runs
        try:
            frames, dets, det_masks, owners = prepare(chunk)
        except Exception as e:                            # noqa: BLE001 -- a missing mask / unreadable frame on ONE rank
            if gather is None:
                raise
            step_errors.append(e)                         # lockstep: this rank keeps joining every remaining collective (as an empty shard) and re-raises at the end
            frames, dets, det_masks, owners = [], [], [], []
        if not dets:
            if gather is not None:                        # lockstep: an empty shard still joins this step's collective
                in_flight.append((chunk, owners, None, t1, None, step))
                if len(in_flight) == 2:
                    finish(in_flight.pop(0))
            continue
        # est_pose_kwargs: extra arguments of the batch call (tests inject decoder maps); a callable gets (index of the chunk's
        # first detection in stream order, number of detections) and returns them per chunk
        extra = est_pose_kwargs(n_submitted, len(dets)) if callable(est_pose_kwargs) else (est_pose_kwargs or {})
        n_submitted += len(dets)
        held = None
        if inject is not None:
            import torch
            idx = [inject_row[(chunk[ti][4], r_id)] for ti, r_id in owners]
            held = (torch.from_numpy(np.ascontiguousarray(inject["inject1"][idx])).cuda(device),
                    torch.from_numpy(np.ascontiguousarray(inject["inject2"][idx])).cuda(device))
            torch.cuda.synchronize(device)
            extra = dict(extra, inject1=held[0].data_ptr(), inject2=held[1].data_ptr(), inject_slots=int(held[1].shape[1]))
        try:
            pending = runtime.est_pose_submit(ctx, specs, frames, dets, det_masks=det_masks if det_masks else None,
                                              anti_aliasing=_resize_generation(cfg),
                                              **extra)
        except Exception as e:                            # noqa: BLE001
            if gather is None:
                raise
            step_errors.append(e)
            pending = None
        in_flight.append((chunk, owners, pending, t1, held, step))
        if len(in_flight) == 2:
            finish(in_flight.pop(0))
    while in_flight:
        finish(in_flight.pop(0))
    if gather is not None:
        rows.sort(key=lambda r: r["_order"])
        gather.close()
    loader.close()
    if step_errors:
        raise step_errors[0]
    if range_events:
        from . import _lib as _l
        raise _l.P2PRangeError("a rank's generator pass left the split-f16 operand range on image(s) %s: its detections are missing from the "
                                    "rows (use precision='auto' or 'f32')" % sorted(set(range_events))[:4])
    out_dir = cfg.get("path_to_output")
    if out_dir and write_csv:
        os.makedirs(out_dir, exist_ok=True)
        save_bop_results(os.path.join(out_dir, output_name(dataset)), rows)
    return rows


# ---------------------------------------------------------------------------------- several GPUs
ROW_REC = 20     # floats per gathered row: order key, scene_id, im_id, obj_id, score, R[9], t[3], time, pad (= parallel.REC)


def rows_to_records(rows) -> np.ndarray:
    out = np.zeros((len(rows), ROW_REC), np.float64)
    for i, r in enumerate(rows):
        gi, k = r["_order"]
        out[i, 0] = gi * 4096 + k                     # < 2^53: exact in float64
        out[i, 1:5] = r["scene_id"], r["im_id"], r["obj_id"], r["score"]
        out[i, 5:14] = np.asarray(r["R"], np.float64).reshape(-1)
        out[i, 14:17] = np.asarray(r["t"], np.float64).reshape(-1)
        out[i, 17] = r["time"]
    return out


def records_to_rows(rec: np.ndarray):
    return [{"scene_id": int(r[1]), "im_id": int(r[2]), "obj_id": int(r[3]), "score": float(r[4]), "R": r[5:14].reshape(3, 3).copy(),
             "t": r[14:17].copy(), "time": float(r[17]), "_order": (int(r[0]) // 4096, int(r[0]) % 4096)} for r in rec]


def run_distributed(cfg: dict, dataset: str, dump: dict, base_dir: str = ".", backend: str = None, same_device: bool = False, **kw):
    """One process per GPU (torch.distributed.run): shard the images, gather the rows on every rank (one all-gather over RCCL
    when the backend is "nccl"), rank 0 writes the CSV.  Returns the merged rows."""
    import torch
    import torch.distributed as dist
    from . import parallel
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if same_device else int(os.environ.get("LOCAL_RANK", rank))
    backend = backend or os.environ.get("P2P_EVAL_BACKEND", "nccl")
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    assert parallel.REC == ROW_REC
    # the pose records of every pooled batch travel through ONE collective per step -- inside the C ABI (RCCL all-gather of the device-resident
    # p2p_pose records, p2p_est_pose_collect_gathered; a rank whose shard is empty at a step joins with P2P_TICKET_NONE) on the "nccl" backend,
    # through torch.distributed otherwise (gloo dry runs, ranks sharing a device) or when P2P_EVAL_TORCH_GATHER is set; every rank then holds
    # the merged rows.  P2P_EVAL_FINAL_GATHER=1 keeps the old scheme: independent shards + one all-gather of the result rows at the end.
    if os.environ.get("P2P_EVAL_FINAL_GATHER"):
        rows = run(cfg, dataset, dump, device=local, base_dir=base_dir, shard=(rank, world), write_csv=False, **kw)
        rec = parallel.gather_poses(rows_to_records(rows), device=torch.device("cuda", local) if backend == "nccl" else None)
        merged = records_to_rows(rec)
    else:
        if backend == "nccl" and not os.environ.get("P2P_EVAL_TORCH_GATHER"):
            make_gather = parallel.CabiPoseGather
        else:
            make_gather = lambda ctx: parallel.TorchPoseGather(torch.device("cuda", local) if backend == "nccl" else None)      # noqa: E731
        merged = run(cfg, dataset, dump, device=local, base_dir=base_dir, shard=(rank, world), write_csv=False, make_gather=make_gather, **kw)
    out_dir = cfg.get("path_to_output")
    if out_dir and rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        save_bop_results(os.path.join(out_dir, output_name(dataset)), merged)
    dist.barrier()
    return merged


def main(argv):
    if len(argv) < 4:
        print("usage: python -m pix2pose_amd.eval_bop <gpu_id> <cfg.json> <dataset> [detections.json]\n"
              "  detections.json: the harness's own dump (a dict, see the module docstring) or a COCO-style detection list for the\n"
              "  BOP directory cfg['dataset_dir'] (bop_dataset.build_dump).  Under torch.distributed.run the images are sharded\n"
              "  over the ranks (gpu_id is ignored: rank i uses GPU LOCAL_RANK) and rank 0 writes the CSV.")
        return 2
    device, cfg_fn, dataset = int(argv[1]), argv[2], argv[3]
    cfg = json.load(open(cfg_fn))
    det_fn = argv[4] if len(argv) > 4 else os.path.join(cfg["dataset_dir"], dataset, "detections_mi355.json")
    dump = json.load(open(det_fn))
    base_dir = os.path.dirname(os.path.abspath(det_fn))
    if isinstance(dump, list):                        # COCO-style detections for a BOP directory
        from . import bop_dataset
        dump = bop_dataset.build_dump(cfg, dataset, dump)
    inject = None
    if os.environ.get("P2P_EVAL_INJECT"):             # test hook, see run()
        with np.load(os.environ["P2P_EVAL_INJECT"]) as z:
            inject = {k: z[k] for k in ("key", "inject1", "inject2")}
    detect_type = cfg.get("detection_pipeline", "rcnn")          # tools/5_evaluation_bop_basic.py:36 ('rcnn' | 'retinanet': no masks)
    bi = int(cfg.get("batch_images", 32))             # images pooled into one device batch (this build's own key)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        rows = run_distributed(cfg, dataset, dump, base_dir=base_dir, same_device=bool(os.environ.get("P2P_EVAL_SAME_DEVICE")), inject=inject,
                               detect_type=detect_type, batch_images=bi)
        if int(os.environ["RANK"]) != 0:
            return 0
    else:
        rows = run(cfg, dataset, dump, device=device, base_dir=base_dir, inject=inject, detect_type=detect_type, batch_images=bi)
    print("Saving %d results to %s" % (len(rows), os.path.join(cfg.get("path_to_output", "."), output_name(dataset))))
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main(sys.argv))
