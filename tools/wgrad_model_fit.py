"""Offline fit of the wgrad schedule cost model (conv_mfma.hip:plan_wgrad) against a tools/wgrad_sweep.py JSON dump."""
import json, math, sys, itertools
data = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "tools/data/wgrad_sweep_iam_b4a2.json"))
CFG = {0: (128, 128, 32), 1: (64, 64, 32)}   # 0: 16 waves, 1: 8 waves (2 K groups)
def cdiv(a, b): return -(-a // b)

def plan(rec, cfg, target):
    N, H, W, C, K, R, S = rec["shape"][:7]
    bmu, bnv, bkp = CFG[cfg]
    M = N * rec["P"] * rec["Q"]
    base = R * S * cdiv(K, bmu) * cdiv(C, bnv)
    ns = max(1, min(cdiv(target, base), cdiv(M, 4 * bkp)))
    chunk = cdiv(cdiv(M, ns), bkp) * bkp
    ns = max(1, cdiv(M, chunk))
    return base, ns, chunk

def predict(rec, cfg, ns, chunk, base, prm):
    N, H, W, C, K, R, S = rec["shape"][:7]
    bmu, bnv, bkp = CFG[cfg]
    tf, ov = prm["cfg"][cfg]
    blocks = base * ns
    q = blocks / 256.0
    wpb = 16 if cfg == 0 else 8
    resident = min(max(q, 1.0), 32.0 / wpb) * wpb          # waves resident per CU
    tf = tf * (prm["occ_floor"] + (1 - prm["occ_floor"]) * min(1.0, resident / prm["occ_w"]))
    step = 2.0 * bmu * bnv * bkp / (tf * 1e12 / 256)
    quanta = math.ceil(q) if q <= prm["ceil_upto"] else q + 0.5
    t = quanta * (chunk / bkp + ov) * step
    t += (ns + 1) * 4.0 * R * S * K * C / prm["bw"] + prm["lat"]
    return t

def evaluate(prm, verbose=False):
    tot = tb = 0
    for rec in data:
        res = rec["results_us"]
        if "default" in res: continue
        cands = {}
        for key, us in res.items():
            cfg, tg = map(int, key.split(","))
            if cfg not in CFG: continue
            base, ns, chunk = plan(rec, cfg, tg)
            cands[(cfg, ns)] = (us, predict(rec, cfg, ns, chunk, base, prm))
        if not cands: continue
        pick = min(cands, key=lambda k: cands[k][1])
        bestk = min(cands, key=lambda k: cands[k][0])
        tot += cands[pick][0] * rec["launches_per_step"]; tb += cands[bestk][0] * rec["launches_per_step"]
        if verbose and (cands[pick][0] - cands[bestk][0]) * rec["launches_per_step"] > 4:
            print("  lost %.1f: pick %s=%.0f (pred %.0f) best %s=%.0f %s" % ((cands[pick][0] - cands[bestk][0]) * rec["launches_per_step"], pick, cands[pick][0], cands[pick][1] * 1e6, bestk, cands[bestk][0], rec["shape"][:7]))
    return tot, tb

best = (1e30, None)
for cu, r0, r3, ov0, ov3, bw, lat, ofl, ow in itertools.product([2, 8], [94, 100, 106], [90, 100, 110], [1, 2], [1, 2], [5e12, 8e12], [3e-6], [0.5, 0.65, 0.8, 1.0], [16, 24, 32]):
    prm = {"cfg": {0: (r0, ov0), 1: (r3, ov3)}, "ceil_upto": cu, "bw": bw, "lat": lat, "occ_floor": ofl, "occ_w": ow}
    r = evaluate(prm)[0]
    if r < best[0]: best = (r, prm)
print(best, evaluate(best[1]))
evaluate(best[1], True)
