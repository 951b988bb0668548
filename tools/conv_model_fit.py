"""Offline fit of the schedule cost model in conv_mfma.hip:plan_conv() against a tools/conv_sweep.py JSON dump."""
import json, math, sys, itertools
data = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "tools/data/conv_sweep_iam_b4a2.json"))
SPL = [1, 2, 3, 4, 6, 8, 12, 16]

def cdiv(a, b): return -(-a // b)

def model(rec, prm):
    N, H, W, C, K, R, S, stride, pad, dil, tr = rec["shape"]
    P, Q = rec["P"], rec["Q"]
    classes = stride[0] * stride[1] if tr else 1
    Mc = N * cdiv(P, stride[0]) * cdiv(Q, stride[1]) if tr else N * P * Q
    bk = 32 if C % 32 == 0 else 16
    T_total = R * S / classes * (C // bk)
    min_taps = (R // stride[0]) * (S // stride[1]) if tr else R * S
    min_steps = max(min_taps, 1) * (C // bk)
    out_bytes = 4.0 * N * P * Q * K
    best = (1e30, None)
    for (bm, bn, tf, ov) in prm["tiles"]:
        if K <= 32:
            if bn != 32: continue
        elif bn == 32 or (bn == 128 and K < 96):
            continue
        step = 2.0 * bm * bn * bk / (tf * 1e12 / 256)
        per_xcd = cdiv(cdiv(Mc, bm), 8) * cdiv(K, bn) * classes
        for n in SPL:
            if n > 1 and (min_steps // n < 3 or out_bytes * n > 1.5e9): break
            q = per_xcd * n / 32.0
            if q <= prm["ceil_upto"]: quanta = math.ceil(q)
            else: quanta = q + prm["tail"]
            t = quanta * (T_total / n + ov) * step
            if n > 1: t += (n + 1) * out_bytes / prm["bw"] + prm["lat"]
            if t < best[0]: best = (t, (bm, bn, n))
    return best

def regret(prm, verbose=False):
    tot = tb = 0
    for rec in data:
        t, cfg = model(rec, prm)
        key = "%d,%d,%d" % cfg
        if key not in rec["results_us"]:
            # forced config clipped by the library (split guard): take nearest lower split
            bm, bn, n = cfg
            while n > 1 and "%d,%d,%d" % (bm, bn, n) not in rec["results_us"]: n = SPL[SPL.index(n) - 1]
            key = "%d,%d,%d" % (bm, bn, n)
        got = rec["results_us"][key]; bestt = min(rec["results_us"].values())
        tot += got * rec["launches_per_step"]; tb += bestt * rec["launches_per_step"]
        if verbose and (got - bestt) * rec["launches_per_step"] > 5:
            bk_ = min(rec["results_us"], key=rec["results_us"].get)
            print("  %6.1f us/step lost: model %s=%.0f (pred %.0f)  best %s=%.0f  %s" % ((got - bestt) * rec["launches_per_step"], key, got, t * 1e6, bk_, bestt, rec["shape"]))
    return tot, tb

base = {"tiles": [(128, 128, 108.0, 1.5), (128, 64, 98.0, 2.0), (64, 64, 90.0, 2.5), (128, 32, 62.0, 2.5)], "ceil_upto": 2, "tail": 0.5, "bw": 4.0e12, "lat": 3e-6}
print("current", regret(base))
best = (1e30, None)
for cu, tail, bw, lat, r128, r12864, r64, ov128, ov64 in itertools.product([2, 4, 6, 8], [0.5, 0.8, 1.0], [4e12, 6e12, 8e12], [3e-6, 6e-6], [104, 108, 112], [94, 98, 102], [84, 88, 92, 96],
                                                                          [1.0, 2.0, 3.0], [1.5, 2.5, 4.0]):
    prm = {"tiles": [(128, 128, r128, ov128), (128, 64, r12864, (ov128 + ov64) / 2), (64, 64, r64, ov64), (128, 32, 62.0, ov64)], "ceil_upto": cu, "tail": tail, "bw": bw, "lat": lat}
    r = regret(prm)[0]
    if r < best[0]: best = (r, prm)
print("best", best)
regret(best[1], True)
