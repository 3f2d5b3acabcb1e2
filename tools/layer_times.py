"""Per-layer kernel durations of the stage-1 generator pass from a rocprofv3 rocpd database, annotated with
algorithmic GFLOP, TFLOP/s (2 x MACs; against 833 = dense f16 MFMA peak / 3 for the split-f16 arithmetic) and the
algorithmic activation traffic (input + residual + output tensors once, fp32) as TB/s against ~6.3 achievable / 8 peak."""
import sqlite3
import sys

LAYERS = [  # (name, MMAC per sample) in launch order for the resnet50 backbone
    ("conv1", 38.5), ("maxpool", 0), ("res2a_2a", 4.2), ("res2a_2b", 37.7), ("res2a_2c+1", 33.6),
    ("res2b_2a", 16.8), ("res2b_2b", 37.7), ("res2b_2c", 16.8), ("res2c_2a", 16.8), ("res2c_2b", 37.7), ("res2c_2c", 16.8),
    ("res3a_2a", 8.4), ("res3a_2b", 37.7), ("res3a_2c+1", 50.4),
    ("res3b_2a", 16.8), ("res3b_2b", 37.7), ("res3b_2c", 16.8), ("res3c_2a", 16.8), ("res3c_2b", 37.7), ("res3c_2c", 16.8),
    ("res3d_2a", 16.8), ("res3d_2b", 37.7), ("res3d_2c", 16.8), ("conv4", 419.4), ("dense_enc", 8.4), ("splitk_reduce", 0),
    ("dense_dec", 4.2), ("up1_p0", 16.8), ("up1_p1", 25.2), ("up1_p2", 25.2), ("up1_p3", 37.7), ("deconv1", 629.1),
    ("up2_p0", 33.6), ("up2_p1", 50.3), ("up2_p2", 50.3), ("up2_p3", 75.5), ("deconv2", 1677.7),
    ("up3_p0", 67.1), ("up3_p1", 100.7), ("up3_p2", 100.7), ("up3_p3", 151.0), ("deconv3", 1258.3), ("heads", 52.4)]


# algorithmic activation traffic per sample in kilo-floats: tensors a layer must read (input, skip / residual) + write
KF = {"conv1": 49 + 262, "maxpool": 262 + 66, "res2a_2a": 66 + 66, "res2a_2b": 66 + 66, "res2a_2c+1": 66 + 66 + 262,
      "res3a_2a": 66 + 33, "res3a_2b": 33 + 33, "res3a_2c+1": 33 + 66 + 131,
      "conv4": 131 + 33, "dense_enc": 33, "dense_dec": 16, "deconv1": 66 + 33 + 66, "deconv2": 131 + 131 + 262,
      "deconv3": 262 + 131 + 524, "heads": 524 + 66}
for _b in ("res2b", "res2c"):
    KF.update({_b + "_2a": 262 + 66, _b + "_2b": 66 + 66, _b + "_2c": 66 + 262 + 262})
for _b in ("res3b", "res3c", "res3d"):
    KF.update({_b + "_2a": 131 + 33, _b + "_2b": 33 + 33, _b + "_2c": 33 + 131 + 131})
for _p in range(4):
    KF.update({"up1_p%d" % _p: 16 + 16, "up2_p%d" % _p: 66 + 33, "up3_p%d" % _p: 262 + 66})


def main(path, batch=256):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "stage1_input" in r[0]][-1]
    ks = [r for r in rows[idx + 1:] if "copyBuffer" not in r[0]]
    layers = list(LAYERS)
    if "maxpool" not in ks[1][0]:              # first layer and max-pool in one kernel (conv1.hip, large batches)
        layers = [("conv1+pool", LAYERS[0][1])] + LAYERS[2:]
        KF["conv1+pool"] = 49 + 131 + 66       # input, the skip's 32 channels, the pooled tensor
    if any("resblock_kernel" in r[0] for r in ks[:len(layers)]):      # identity bottleneck blocks as one launch each (resblock.hip)
        seq = [r[0] for r in ks[:12]]
        n_fused_front = sum("resblock_kernel" in nm for nm in seq)
        fused = ("res2a", "res2b", "res2c", "res3a", "res3b", "res3c", "res3d") if "resblock_kernel" in seq[1] else ("res2b", "res2c", "res3b", "res3c", "res3d")
        out = []
        for nm, mmac in layers:
            b = nm.split("_")[0]
            if b in fused:
                if nm.endswith("_2a"):
                    out.append((b + " (fused)", sum(m for n2, m in layers if n2.split("_")[0] == b)))
                    KF[b + " (fused)"] = {"res2a": 66 + 262, "res3a": 66 + 131}.get(b, 2 * (262 if b.startswith("res2") else 131))      # block input read once, output written once
                continue
            out.append((nm, mmac))
        layers = out
    # small launches (one detection at a time, and up1 at 64 inputs): the four phases of a transposed convolution are ONE streaming launch
    out, i = [], 0
    for nm, mmac in layers:
        if nm.endswith("_p0") and i < len(ks) and "igemm_stream" in ks[i][0]:
            b = nm.split("_")[0]
            out.append((b + " (4 phases)", sum(m for n2, m in layers if n2.startswith(b + "_p"))))
            KF[b + " (4 phases)"] = 4 * KF.get(nm, 0)
            i += 1
            continue
        if out and out[-1][0] == nm.split("_")[0] + " (4 phases)":
            continue
        out.append((nm, mmac))
        i += 1
    layers = out
    # 5x5 stride-1 layers in Winograd form (wino.hip) and transposed convolutions in Winograd F(4,3) form (wino3.hip): two launches per layer --
    # the input transform, then the position GEMMs (all four phases of a transposed convolution in one launch)
    out, i, skip = [], 0, ""
    for nm, mmac in layers:
        if skip and nm.startswith(skip):
            continue
        if nm.endswith("_p0") and i < len(ks) and ("wino3_input" in ks[i][0] or "wino3o_input" in ks[i][0]):
            b = nm.split("_")[0]
            out.append((b + " V", 0))
            KF[b + " V"] = {"up1": 16 * 2.5, "up2": 66 * 2.5, "up3": 262 * 2.5}.get(b, 0)             # x read once, V (1.5x the elements) written once
            out.append((b + " gemm", sum(m for n2, m in layers if n2.startswith(b + "_p"))))
            KF[b + " gemm"] = {"up1": 16 * 1.5 + 66, "up2": 66 * 1.5 + 131, "up3": 262 * 1.5 + 262}.get(b, 0)
            skip = b + "_p"
            i += 2
            continue
        if nm == "conv4" and i < len(ks) and "wino3o_input" in ks[i][0]:                                # conv4 in Winograd F(4,3) form on its parity planes (wino3o.hip)
            out.append(("conv4 V", 0))
            KF["conv4 V"] = 131 * 2.5
            out.append(("conv4 gemm", mmac))
            KF["conv4 gemm"] = 131 * 1.5 + 33
            i += 2
            if i < len(ks) and "splitk_reduce" in ks[i][0]:                                          # small launches: K split over the four planes
                out.append(("conv4 reduce", 0))
                KF["conv4 reduce"] = 5 * 33
                i += 1
            continue
        if nm.startswith("deconv") and i < len(ks) and "wino_input" in ks[i][0]:
            out.append((nm + " V", 0))
            KF[nm + " V"] = {"deconv1": 3 * 98, "deconv2": 3 * 262, "deconv3": 3 * 393}.get(nm, 0)      # x read once, V (2x the elements) written once
            out.append((nm + " gemm", mmac))
            KF[nm + " gemm"] = KF.get(nm, 0)
            i += 2
            if i < len(ks) and "splitk_reduce" in ks[i][0]:                                          # launches under half a workgroup per CU: K split over ranges of channel slices
                out.append((nm + " reduce", 0))
                KF[nm + " reduce"] = {"deconv1": 3 * 66, "deconv2": 3 * 262, "deconv3": 3 * 524}.get(nm, 0)      # (two or more) partial tensors read, the output written
                i += 1
            continue
        if nm == "conv4" and i + 1 < len(ks) and "igemm_stream" in ks[i][0] and "splitk_reduce" in ks[i + 1][0]:      # a few inputs: conv4 on the streaming kernel, K split
            out.append((nm, mmac))
            out.append(("conv4 reduce", 0))
            KF["conv4 reduce"] = 5 * 33
            i += 2
            continue
        out.append((nm, mmac))
        i += 1
    layers = out
    ks = ks[:len(layers)]
    tot = 0
    for (nm, mmac), r in zip(layers, ks):
        us = (r[2] - r[1]) / 1e3
        tot += us
        gf = 2 * mmac * batch / 1e3
        tb = KF.get(nm, 0) * 1e3 * 4 * batch / (us * 1e-6) / 1e12 if us else 0
        print("%-16s %-28s %9.1f us %9.1f GFLOP %7.1f TFLOP/s %6.2f TB/s" % (nm, r[0].replace("void p2p::", "").replace("p2p::", "").replace("(anonymous namespace)::", "")[:28],
                                                                         us, gf, gf / us * 1e3 if us else 0, tb))
    print("total %.1f us -> %.0f inputs/s" % (tot, batch / tot * 1e6))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 256)
