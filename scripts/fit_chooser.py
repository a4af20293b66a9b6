"""Developer aid: fits the per-shape constants of the 32x32x16 launch-shape chooser (choose_mfma32 in
csrc/w4a16.hip) to the output of scripts/exp_shared_sweep.py.

Model of a shape's time on a layer:  ceil(workgroups / 256) x (K / 128 x a + b)  microseconds.  a and b per shape
by least squares on relative error over every (layer, rows) the shape was measured on; then the loss of the
model's choice against the best measured shape, overall and with each layer held out of the fit in turn.
Usage: python scripts/fit_chooser.py gpurun_out/sweep.log [min_rows]"""
import ast
import json
import sys
from collections import defaultdict

import numpy as np

rows_min = int(sys.argv[2]) if len(sys.argv) > 2 else 33
recs = [json.loads(l[5:]) for l in open(sys.argv[1]) if l.startswith("JSON ")]
recs = [r for r in recs if r["M"] >= rows_min]


def shape_of(key):
    c = ast.literal_eval(key)
    if c is None:
        return None
    _, mt, t, kw = c[:4]
    return (mt, t, -kw, 1) if kw < 0 else (mt, t, 1, kw)  # (mt, t, nw, kw)


def wgs(shape, M, N):
    mt, t, nw, kw = shape
    return -(-(-(-N // 32)) // (t * nw)) * -(-M // (32 * mt))


def rounds_of(shape, M, N):
    """Whole rounds of 256 workgroups.  (Tried and not better on the data: 512 slots for the shape two of which
    share a CU; a fractional last round once there are several.)"""
    return -(-wgs(shape, M, N) // 256)


def fit(records):
    data = defaultdict(list)
    for r in records:
        for key, us in r["us"].items():
            sh = shape_of(key)
            if sh is None:
                continue
            if sh[3] == 1 and r["K"] < 256:
                continue
            rounds = rounds_of(sh, r["M"], r["N"])
            data[sh].append((rounds * r["K"] / 128.0, rounds, us))
    out = {}
    for sh, pts in data.items():
        A = np.array([[p[0] / p[2], p[1] / p[2]] for p in pts])  # relative error: (a x + b y) / us ~ 1
        sol, *_ = np.linalg.lstsq(A, np.ones(len(pts)), rcond=None)
        out[sh] = (max(sol[0], 1e-4), max(sol[1], 0.0))
    return out


def choose(consts, M, K, N):
    best, best_cost = None, 1e300
    for sh, (a, b) in consts.items():
        if sh[3] == 1 and K < 256:
            continue
        cost = rounds_of(sh, M, N) * (K / 128.0 * a + b)
        if cost < best_cost:
            best, best_cost = sh, cost
    return best


def loss(consts, records):
    ls = []
    for r in records:
        times = {shape_of(k): v for k, v in r["us"].items() if shape_of(k) is not None}
        pick = choose(consts, r["M"], r["K"], r["N"])
        ls.append(times[pick] / min(times.values()) - 1.0)
    return np.array(ls)


consts = fit(recs)
print("static const Shape shapes[] = {")
for sh in sorted(consts):
    a, b = consts[sh]
    print(f"      {{{sh[0]}, {sh[1]}, {sh[2]}, {sh[3]}, {a:.4f}, {b:.3f}}},")
print("};")
l = loss(consts, recs)
print(f"all data: mean loss {l.mean() * 100:.2f} %, worst {l.max() * 100:.1f} %  ({len(recs)} cases)")
layers = sorted({(r["K"], r["N"]) for r in recs})
held = []
for lay in layers:
    c = fit([r for r in recs if (r["K"], r["N"]) != lay])
    held.append(loss(c, [r for r in recs if (r["K"], r["N"]) == lay]))
held = np.concatenate(held)
print(f"each layer held out in turn: mean loss {held.mean() * 100:.2f} %, worst {held.max() * 100:.1f} %")
cur = [r["us"]["None"] / min(v for k, v in r["us"].items() if k != "None") - 1.0 for r in recs]
print(f"the library's current choice on the same cases: mean loss {np.mean(cur) * 100:.2f} %, worst {np.max(cur) * 100:.1f} %")
if len(sys.argv) > 3 and sys.argv[3] != "--tune":  # list the cases the fitted model loses most on
    for r in recs:
        times = {shape_of(k): v for k, v in r["us"].items() if shape_of(k) is not None}
        pick = choose(consts, r["M"], r["K"], r["N"])
        ls_ = times[pick] / min(times.values()) - 1.0
        if ls_ > float(sys.argv[3]):
            best = min(times, key=times.get)
            print(f"K={r['K']} N={r['N']} M={r['M']}: picks {pick} {times[pick]:.1f} us, best {best} {times[best]:.1f} us (+{ls_ * 100:.1f} %)")

if "--tune" in sys.argv:
    # Second step: least squares fits each shape's TIMES; what matters is the CHOICE.  Scale each shape's pair of
    # constants by a common factor (0.90 .. 1.10), shape by shape, keeping whatever lowers the mean loss of the choice.
    def tune(c0, records):
        c = dict(c0)
        for _ in range(3):
            for sh in sorted(c):
                base = c[sh]
                best_f, best_l = 1.0, loss(c, records).mean()
                for f in (0.80, 0.85, 0.90, 0.93, 0.96, 0.98, 1.02, 1.04, 1.07, 1.10, 1.15, 1.20, 1.25):
                    c[sh] = (base[0] * f, base[1] * f)
                    l_ = loss(c, records).mean()
                    if l_ < best_l - 1e-5:
                        best_f, best_l = f, l_
                c[sh] = (base[0] * best_f, base[1] * best_f)
        return c
    tuned = tune(consts, recs)
    print("tuned:")
    for sh in sorted(tuned):
        print(f"      {{{sh[0]}, {sh[1]}, {sh[2]}, {sh[3]}, {tuned[sh][0]:.4f}, {tuned[sh][1]:.3f}}},")
    l = loss(tuned, recs)
    print(f"tuned, all data: mean loss {l.mean() * 100:.2f} %, worst {l.max() * 100:.1f} %")
    held = []
    for lay in layers:
        rest = [r for r in recs if (r["K"], r["N"]) != lay]
        held.append(loss(tune(fit(rest), rest), [r for r in recs if (r["K"], r["N"]) == lay]))
    held = np.concatenate(held)
    print(f"tuned, each layer held out in turn: mean loss {held.mean() * 100:.2f} %, worst {held.max() * 100:.1f} %")
