#!/bin/bash
# Same-lease regression gate under the headline: the previous round's head against HEAD (and HEAD without the operand-range guard),
# INTERLEAVED on one GPU lease, so that box-to-box and power variance cancel out of every ratio that DESIGN.md quotes.
#
#   tools/ab_round.sh build [prev-commit]   build container: `git archive` of the previous round's head (default 4f9273c = round 5's) into tools/ab/prev/ + its library.
#                                           tools/ab/ is git-ignored and listed in .gpurunignore: take that line out for the `run` call.
#   tools/ab_round.sh run [rounds]          GPU box: `rounds` (default 3) interleaved passes of bench.py --steps 20 over prev / head / nowino (= head with --winograd off),
#                                           shader clock + power sampled with rocm-smi during each, then a rocprofv3 kernel trace of one blocking
#                                           step of prev and head -> per-layer table with ratios.  Output: gpurun_out/r06_vs_prev.txt
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
AB=$ROOT/tools/ab
MODE=${1:-run}

if [ "$MODE" == "build" ]; then
    set -e
    PREV=${2:-4f9273c}
    rm -rf $AB/prev && mkdir -p $AB/prev
    (cd $ROOT && git archive $PREV) | tar -x -C $AB/prev
    echo $PREV > $AB/prev/.commit
    (cd $AB/prev && python pix2pose_amd/build.py > /dev/null)
    echo "prev ($PREV): $AB/prev/pix2pose_amd/libp2p_mi355.so"
    python $ROOT/pix2pose_amd/build.py > /dev/null          # HEAD's own library
    exit 0
fi

# ---------------------------------------------------------------------------------------------------------------- run (GPU box)
ROUNDS=${2:-3}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
REP=$OUT/r06_vs_prev.txt
cd /tmp && export TMPDIR=/tmp
PREV_COMMIT=$(cat $AB/prev/.commit 2>/dev/null || echo "?")
{
    echo "# same-lease A/B: prev = $PREV_COMMIT (tools/ab/prev), head = this tree, nowino = this tree with bench.py --winograd off (the 5x5 decoder layers on the direct kernels)"
    echo "# bench.py --steps 20 --warmup 3 --no-legs, $ROUNDS interleaved rounds; sclk / power = mean of rocm-smi samples (0.25 s) while the bench ran"
    rocm-smi --showproductname 2>/dev/null | grep -i "card series\|gfx" | head -2
} > $REP

sample_smi() {      # $1 = file; samples until the file $1.stop appears
    while [ ! -f $1.stop ]; do
        rocm-smi --showclocks --showpower --json 2>/dev/null >> $1
        echo >> $1
        sleep 0.25
    done
}
one_bench() {       # $1 = label, $2 = tree, $3 = P2P_LIB or "", $4 = extra environment assignment (route switch of the development twin) or ""
    local f=$OUT/ab_${1}.smi
    rm -f $f $f.stop
    sample_smi $f &
    local sp=$!
    local line
    if [ -n "$3" ]; then line=$(cd $2 && env P2P_LIB=$3 ${4:-P2P_AB=1} python bench.py --steps 20 --warmup 3 --no-legs 2>/dev/null | tail -1)
    else line=$(cd $2 && python bench.py --steps 20 --warmup 3 --no-legs ${5:-} 2>/dev/null | tail -1); fi
    touch $f.stop; wait $sp 2>/dev/null
    python - "$1" "$line" $f <<'EOF'
import json, sys, re
label, line, f = sys.argv[1:4]
try:
    d = json.loads(line)
    v, ms = d["value"], d["ms_per_step"]
except Exception:
    v, ms = float("nan"), float("nan")
sclk, pw = [], []
for ln in open(f):
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    try:
        j = json.loads(ln)
    except Exception:
        continue
    for card in j.values():
        if not isinstance(card, dict):
            continue
        for k, val in card.items():
            m = re.search(r"([0-9.]+)", str(val))
            if not m:
                continue
            if "sclk" in k.lower():
                sclk.append(float(m.group(1)))
            elif "power" in k.lower() and "socket" in k.lower():
                pw.append(float(m.group(1)))
busy = sorted(sclk)[len(sclk) // 2:] if sclk else []           # the upper half of the samples = while kernels ran
print("%-8s %8.1f crops/s  %6.2f ms/step   sclk(busy half) %s MHz   power(max) %s W   (%d samples)" % (
    label, v, ms, "%.0f" % (sum(busy) / len(busy)) if busy else "n/a", "%.0f" % max(pw) if pw else "n/a", len(sclk)))
EOF
    rm -f $f $f.stop
}
for r in $(seq 1 $ROUNDS); do
    echo "## round $r" >> $REP
    one_bench prev $AB/prev "" >> $REP
    one_bench head $ROOT "" >> $REP
    one_bench nowino $ROOT "" "" "--winograd off" >> $REP      # head with the 5x5 decoder layers on the direct kernels (round 5's arithmetic)
done

# per-layer times of one blocking step: prev, head, and head with the three-launch blocks -- two interleaved traces each, the smaller time of
# a layer counts (one trace of a 10-us launch, or of a whole run that met a power excursion, is noise)
for rep in 1 2; do
for v in prev head nowino; do
    tree=$ROOT; [ $v == prev ] && tree=$AB/prev
    rm -rf $OUT/ab_prof_$v
    if [ $v == nowino ]; then
        (cd $tree && rocprofv3 --kernel-trace -d $OUT/ab_prof_$v -o t -- python bench.py --steps 2 --warmup 1 --blocking --no-legs --winograd off > /dev/null 2>&1)
    else
    (cd $tree && rocprofv3 --kernel-trace -d $OUT/ab_prof_$v -o t -- python bench.py --steps 2 --warmup 1 --blocking --no-legs > /dev/null 2>&1)
    fi
    db=$(find $OUT/ab_prof_$v -name "t_results.db" | head -1)
    python $tree/tools/layer_times.py $db > $OUT/ab_layers_${v}_$rep.txt 2>&1
    rm -rf $OUT/ab_prof_$v
done
done
python - $OUT/ab_layers_prev $OUT/ab_layers_head $OUT/ab_layers_nowino >> $REP <<'EOF'
import sys
def load1(f):
    d, order = {}, []
    for ln in open(f):
        p = ln.split()
        if len(p) > 3 and "us" in p:
            k = p[0].split("_")[0] if p[0].startswith("res") else p[0]        # ResNet blocks compare as blocks (head runs the identity blocks as ONE launch)
            if p[0].startswith("deconv") and p[1] in ("V", "gemm"): k = p[0]       # Winograd layers: input transform + GEMM launch summed per layer
            if k not in d:
                d[k] = 0.0; order.append(k)
            d[k] += float(p[p.index("us") - 1])
        elif ln.startswith("total"):
            d["total"] = float(p[1]); order.append("total")
    return d, order
def load(stem):           # the smaller time of the two traces, layer by layer (the total: sum of those)
    d1, o1 = load1(stem + "_1.txt"); d2, _ = load1(stem + "_2.txt")
    d = {k: min(v, d2.get(k, v)) for k, v in d1.items()}
    d["total"] = sum(v for k, v in d.items() if k != "total")
    return d, o1
a, oa = load(sys.argv[1]); b, ob = load(sys.argv[2]); c, oc = load(sys.argv[3])
print("## per-layer times of one blocking 256-input pass (rocprofv3 kernel trace, us; two interleaved traces per variant, the smaller time per layer; ResNet blocks summed per block):")
print("## %-12s %9s %9s %9s   %s" % ("layer", "prev", "head", "nowino", "head/prev  nowino/prev"))
for k in ob:
    if k in a:
        flag = "   <-- nowino (= this round's code on last round's route) > 2 % slower than prev" if k in c and c[k] / a[k] > 1.02 and k != "total" else ""
        print("%-14s %9.1f %9.1f %9.1f   %5.3f  %5.3f%s" % (k, a[k], b[k], c.get(k, float("nan")), b[k] / a[k], c.get(k, float("nan")) / a[k], flag))
    else:
        print("%-14s %9s %9.1f" % (k, "-", b[k]))
for k in oa:
    if k not in b:
        print("%-14s %9.1f %9s" % (k, a[k], "-"))
EOF
cat $REP
