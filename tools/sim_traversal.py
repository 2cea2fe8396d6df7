#!/usr/bin/env python3
"""Lock-step simulation of a 64-lane wave traversing the Cornell BVH (host dumps of libptmi's builders; no GPU, no oracle): how many
wave-steps, and how many lanes per section, for (a) the round-2 step (instance + up to 4 branch levels + leaf in one step, t_max pruning
at every box test) and (b) the round-3 two-phase step (branch steps with leaves DEFERRED to a per-lane candidate list, then a leaf phase
that drains the lists in order).  Rays: cosine-distributed directions from uniformly chosen surface points (what bounces 1.. look like).
Binary64 arithmetic: this estimates utilisation, it does not check parity.   sim_traversal.py [n_waves] [K1] [levels]"""
import os, sys, math, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from path_tracer_amd import api, scenes

EPS = 5e-4
random.seed(1)
sc = scenes.cornell_box(64, 64)
r = api.Renderer(sc, 64, 64)
tl = r.tlas_dump(0)
blas = [r.blas_dump(b) for b in range(r.blas_count())]
tris = [m.positions.astype(np.float64) for m in sc.models]          # [n,3,3] load order
nrm = [m.normals.astype(np.float64) for m in sc.models]


def slab(box, o, inv):
    tn, tf = EPS, math.inf
    for k in range(3):
        a, b = (box[k] - o[k]) * inv[k], (box[3 + k] - o[k]) * inv[k]
        if a != a or b != b: continue
        if a > b: a, b = b, a
        tn, tf = max(tn, a), min(tf, b)
    return tn, tf


def tri_t(p, o, d):
    e1, e2 = p[1] - p[0], p[2] - p[0]
    h = np.cross(d, e2); a = float(np.dot(e1, h))
    if abs(a) < 1e-12: return None
    f = 1.0 / a; s = o - p[0]; u = f * float(np.dot(s, h))
    if u < 0 or u > 1: return None
    q = np.cross(s, e1); v = f * float(np.dot(d, q))
    if v < 0 or u + v > 1: return None
    return f * float(np.dot(e2, q))


def make_ray():
    # area-weighted surface point, cosine direction about the stored normal
    areas = []
    for m, t in enumerate(tris):
        for k in range(len(t)):
            areas.append((0.5 * np.linalg.norm(np.cross(t[k, 1] - t[k, 0], t[k, 2] - t[k, 0])), m, k))
    tot = sum(a for a, _, _ in areas)
    while True:
        x = random.random() * tot
        for a, m, k in areas:
            x -= a
            if x <= 0: break
        u, v = random.random(), random.random()
        if u + v > 1: u, v = 1 - u, 1 - v
        p = tris[m][k, 0] * (1 - u - v) + tris[m][k, 1] * u + tris[m][k, 2] * v
        n = nrm[m][k, 0]
        # cosine hemisphere
        r1, r2 = random.random(), random.random()
        phi = 2 * math.pi * r1; s = math.sqrt(r2)
        a = np.array([1.0, 0, 0]) if abs(n[0]) < 0.9 else np.array([0, 1.0, 0])
        t = np.cross(n, a); t /= np.linalg.norm(t); b = np.cross(n, t)
        d = t * (math.cos(phi) * s) + b * (math.sin(phi) * s) + n * math.sqrt(1 - r2)
        d /= np.linalg.norm(d)
        return p + 0.0, d


class Ray:
    __slots__ = ("o", "d", "inv", "t_max", "stack", "in_blas", "inst", "cands", "best", "r_inst", "r_pruned", "ts_inst")

    def __init__(self):
        self.o, self.d = make_ray()
        self.inv = np.array([1.0 / x if x != 0 else math.inf for x in self.d])
        self.t_max = math.inf
        self.stack = [("T", tl["root"], 0.0)]
        self.in_blas = -1
        self.inst = -1
        self.cands = []
        self.r_inst = -1
        self.r_pruned = False
        self.ts_inst = 0.0


def node_of(kind, idx, inst):
    """-> ('branch', (childL), (childR)) / ('inst', i) / ('leaf', inst, first, count); children = (space, idx, box)"""
    if kind == "T":
        if tl["kind"][idx] == 0: return ("branch", ("T", int(tl["a"][idx])), ("T", int(tl["b"][idx])))
        return ("inst", int(tl["a"][idx]))
    b = blas[inst]
    if b["kind"][idx] == 0: return ("branch", ("B", int(b["a"][idx])), ("B", int(b["b"][idx])))
    return ("leaf", inst, int(b["a"][idx]), int(b["b"][idx]))


def box_of(space, idx, inst):
    return tl["boxes"][idx] if space == "T" else blas[inst]["boxes"][idx]


def test_leaf(ray, inst, first, count):
    ids = blas[inst]["prim_ids"][first:first + count]
    for pid in ids:
        t = tri_t(tris[inst][pid], ray.o, ray.d)
        if t is not None and EPS <= t <= ray.t_max:
            ray.t_max = t
            ray.best = (inst, int(pid))


# ---------------------------------------------------------------- (a) round-2 step
def step_r2(ray, levels, sec):
    """one traversal step of the round-2 kernel; sec collects which sections this lane ran: 'i', number of branch levels, leaf pairs"""
    if not ray.stack: return False
    space, idx, ts = ray.stack.pop()
    if ts > ray.t_max: return True
    inst = ray.inst if space == "B" else -1
    nd = node_of(space, idx, inst)
    if nd[0] == "inst":
        sec["i"] = 1
        ray.inst = inst = nd[1]
        space, idx, ts = "B", int(blas[inst]["root"]), 0.0
        nd = node_of("B", idx, inst)
    lv = 0
    while nd[0] == "branch" and lv < levels:
        lv += 1
        (sl, il), (sr, ir) = nd[1], nd[2]
        tnl, tfl = slab(box_of(sl, il, inst), ray.o, ray.inv)
        tnr, tfr = slab(box_of(sr, ir, inst), ray.o, ray.inv)
        hl, hr = tnl <= min(tfl, ray.t_max), tnr <= min(tfr, ray.t_max)
        left_near = tnl < tnr
        L, R = (sl, il, tnl), (sr, ir, tnr)
        if hl and hr: ray.stack.append(R if left_near else L)
        nd = ("none",)
        if hl or hr:
            near = L if (hl and (left_near or not hr)) else R
            nn = node_of(near[0], near[1], inst)
            if nn[0] == "leaf" or (nn[0] == "branch" and lv < levels): nd, ts = nn, near[2]
            else: ray.stack.append(near)
    sec["b"] = lv
    if nd[0] == "leaf":
        sec["l"] = (nd[3] + 1) // 2
        test_leaf(ray, nd[1], nd[2], nd[3])
    elif nd[0] == "branch":
        ray.stack.append((space, idx, ts))  # cannot happen (loop ends on levels only with a pushed near)
    return True


def run_r2(n_waves, levels, steps_per_round=8, refill_below=40):
    cost = dict(steps=0, lanes=0, wi=0, li=0, wb=0, lb=0, wl=0, ll=0, rays=0, valu=0.0)
    for _ in range(n_waves):
        lanes = [None] * 64
        budget = 64 * 20
        while True:
            n_act = sum(1 for x in lanes if x is not None)
            if n_act <= refill_below and budget > 0:
                for i in range(64):
                    if lanes[i] is None and budget > 0:
                        lanes[i] = Ray(); budget -= 1; cost["rays"] += 1
                cost["valu"] += 150
            if all(x is None for x in lanes): break
            for _ in range(steps_per_round):
                secs = []
                for i in range(64):
                    if lanes[i] is None: continue
                    s = {}
                    if not step_r2(lanes[i], levels, s): lanes[i] = None; continue
                    secs.append(s)
                if not secs: continue
                cost["steps"] += 1; cost["lanes"] += len(secs)
                ni = sum(1 for s in secs if "i" in s); nb = sum(1 for s in secs if s.get("b", 0) > 0); nl = sum(1 for s in secs if "l" in s)
                mb = max((s.get("b", 0) for s in secs), default=0); ml = max((s.get("l", 0) for s in secs), default=0)
                cost["wi"] += ni > 0; cost["li"] += ni; cost["wb"] += nb > 0; cost["lb"] += nb; cost["wl"] += nl > 0; cost["ll"] += nl
                cost["valu"] += 25 + (45 if ni else 0) + mb * 62 + ml * 120
    return cost


# ---------------------------------------------------------------- (b) two-phase step
def step_p1(ray, levels):
    """branch phase: pop, expand up to `levels`, defer leaves.  returns (did work, levels run)"""
    if not ray.stack: return False, 0
    space, idx, ts = ray.stack.pop()
    if ts > ray.t_max: return True, 0
    inst = ray.inst if space == "B" else -1
    nd = node_of(space, idx, inst)
    if nd[0] == "inst":
        ray.inst = inst = nd[1]
        ray.ts_inst = ts
        space, idx, ts = "B", int(blas[inst]["root"]), 0.0
        nd = node_of("B", idx, inst)
    lv = 0
    while nd[0] == "branch" and lv < levels:
        lv += 1
        (sl, il), (sr, ir) = nd[1], nd[2]
        tnl, tfl = slab(box_of(sl, il, inst), ray.o, ray.inv)
        tnr, tfr = slab(box_of(sr, ir, inst), ray.o, ray.inv)
        hl, hr = tnl <= min(tfl, ray.t_max), tnr <= min(tfr, ray.t_max)   # t_max may be stale: conservative
        left_near = tnl < tnr
        L, R = (sl, il, tnl), (sr, ir, tnr)
        nd = ("none",)
        near = far = None
        if hl and hr: near, far = (L, R) if left_near else (R, L)
        elif hl: near = L
        elif hr: near = R
        if near is not None:
            nn = node_of(near[0], near[1], inst)
            nf = node_of(far[0], far[1], inst) if far is not None else None
            if nn[0] == "leaf":
                ray.cands.append((nn, near[2], inst, ray.ts_inst))
                if nf is not None and nf[0] == "leaf": ray.cands.append((nf, far[2], inst, ray.ts_inst)); far = None
            if far is not None: ray.stack.append(far)
            if nn[0] == "branch":
                if lv < levels: nd, ts = nn, near[2]
                else: ray.stack.append(near)
            elif nn[0] == "inst": ray.stack.append(near)
    if nd[0] == "leaf": ray.cands.append((nd, ts, inst, ray.ts_inst))
    return True, lv


def step_p2(ray):
    nd, ts, inst, ts_inst = ray.cands.pop(0)
    if inst != ray.r_inst:
        ray.r_inst = inst; ray.r_pruned = ts_inst > ray.t_max
    if ray.r_pruned or ts > ray.t_max: return 0
    test_leaf(ray, nd[1], nd[2], nd[3])
    return (nd[3] + 1) // 2


def run_p(n_waves, levels, K1, refill_below=40, rounds_between_service=2):
    cost = dict(p1=0, p1_lanes=0, p2=0, p2_lanes=0, p2_tested=0, rays=0, valu=0.0, max_c=0)
    for _ in range(n_waves):
        lanes = [None] * 64
        budget = 64 * 20
        while True:
            n_act = sum(1 for x in lanes if x is not None)
            if n_act <= refill_below and budget > 0:
                for i in range(64):
                    if lanes[i] is None and budget > 0:
                        lanes[i] = Ray(); budget -= 1; cost["rays"] += 1
                cost["valu"] += 150
            if all(x is None for x in lanes): break
            for _ in range(rounds_between_service):
                for _ in range(K1):
                    n = 0; ml = 0
                    for x in lanes:
                        if x is None: continue
                        did, lv = step_p1(x, levels)
                        n += did; ml = max(ml, lv)
                    if n: cost["p1"] += 1; cost["p1_lanes"] += n; cost["valu"] += 30 + ml * 62
                cost["max_c"] = max(cost["max_c"], max((len(x.cands) for x in lanes if x is not None), default=0))
                while True:
                    n = 0; mt = 0; tested = 0
                    for x in lanes:
                        if x is None or not x.cands: continue
                        k = step_p2(x); n += 1; mt = max(mt, k); tested += k > 0
                    if not n: break
                    cost["p2"] += 1; cost["p2_lanes"] += n; cost["p2_tested"] += tested; cost["valu"] += 25 + mt * 120
                for i in range(64):
                    if lanes[i] is not None and not lanes[i].stack: lanes[i] = None
    return cost


if __name__ == "__main__":
    nw = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    K1 = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    levels = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    c = run_r2(nw, 4)
    print(f"round-2 step : {c['rays']} rays, {c['steps']} wave-steps ({c['steps'] / c['rays']:.3f}/ray), {c['lanes'] / c['steps']:.1f} lanes/step; "
          f"inst {c['li'] / max(c['wi'], 1):.1f} lanes in {c['wi'] / c['steps']:.2f}, branch {c['lb'] / max(c['wb'], 1):.1f} in {c['wb'] / c['steps']:.2f}, "
          f"leaf {c['ll'] / max(c['wl'], 1):.1f} in {c['wl'] / c['steps']:.2f}; model VALU/ray {c['valu'] / c['rays']:.1f}")
    for K, lv, rb in ((K1, levels, 2), (2, 2, 4), (4, 1, 2), (4, 3, 2), (8, 2, 1), (3, 2, 3)):
        random.seed(1)
        c = run_p(nw, lv, K, rounds_between_service=rb)
        print(f"two-phase K1={K} levels={lv} rounds={rb}: {c['rays']} rays, branch steps {c['p1']} ({c['p1_lanes'] / max(c['p1'], 1):.1f} lanes), leaf steps {c['p2']} "
              f"({c['p2_lanes'] / max(c['p2'], 1):.1f} lanes, {c['p2_tested'] / max(c['p2'], 1):.1f} testing), max list {c['max_c']}; model VALU/ray {c['valu'] / c['rays']:.1f}")


def run_q(n_waves, levels, T, C, refill_below=40, service_every=8, budget0=64 * 20):
    """every wave-step is ONE kind, chosen by vote: a leaf round when >= T lanes hold a candidate (or a list is full, or nobody can
    branch), else a branch step; lists hold up to C candidates"""
    cost = dict(p1=0, p1_lanes=0, p2=0, p2_lanes=0, p2_tested=0, rays=0, valu=0.0)
    for _ in range(n_waves):
        lanes = [None] * 64
        budget = budget0
        it = 0
        while True:
            n_act = sum(1 for x in lanes if x is not None)
            if it % service_every == 0:
                for i in range(64):
                    if lanes[i] is not None and not lanes[i].stack and not lanes[i].cands: lanes[i] = None
                n_act = sum(1 for x in lanes if x is not None)
                if n_act <= refill_below and budget > 0:
                    for i in range(64):
                        if lanes[i] is None and budget > 0:
                            lanes[i] = Ray(); budget -= 1; cost["rays"] += 1
                    cost["valu"] += 150
                if all(x is None for x in lanes): break
            it += 1
            can_branch = [x for x in lanes if x is not None and x.stack and len(x.cands) + 2 <= C]
            has_cand = [x for x in lanes if x is not None and x.cands]
            if has_cand and (len(has_cand) >= T or not can_branch):
                mt = 0; tested = 0
                for x in has_cand:
                    k = step_p2(x); mt = max(mt, k); tested += k > 0
                cost["p2"] += 1; cost["p2_lanes"] += len(has_cand); cost["p2_tested"] += tested; cost["valu"] += 25 + mt * 120
            elif can_branch:
                ml = 0
                for x in can_branch:
                    did, lv = step_p1(x, levels); ml = max(ml, lv)
                cost["p1"] += 1; cost["p1_lanes"] += len(can_branch); cost["valu"] += 30 + ml * 62
            else:
                it = (it + service_every - 1) // service_every * service_every  # nothing to do: service now
    return cost


if __name__ == "__main__":
    for T, C, lv in ((32, 4, 2), (40, 4, 2), (48, 6, 2), (32, 4, 1), (24, 4, 2), (32, 8, 2)):
        random.seed(1)
        c = run_q(2, lv, T, C)
        print(f"voted T={T} C={C} levels={lv}: {c['rays']} rays, branch steps {c['p1']} ({c['p1_lanes'] / max(c['p1'], 1):.1f} lanes), leaf steps {c['p2']} "
              f"({c['p2_lanes'] / max(c['p2'], 1):.1f} lanes, {c['p2_tested'] / max(c['p2'], 1):.1f} testing); model VALU/ray {c['valu'] / c['rays']:.1f}")
    random.seed(1)
    c = run_r2(2, 4) if False else None
    print("streaming (lanes refilled at every step / every 2 steps):")
    for T, C, lv, se in ((24, 4, 2, 1), (32, 4, 2, 1), (40, 4, 2, 1), (32, 4, 2, 2), (40, 4, 2, 2), (48, 4, 2, 2), (40, 4, 1, 2)):
        random.seed(1)
        c = run_q(2, lv, T, C, refill_below=63, service_every=se)
        print(f"voted T={T} C={C} levels={lv} refill every {se}: {c['rays']} rays, branch steps {c['p1']} ({c['p1_lanes'] / max(c['p1'], 1):.1f} lanes), leaf steps {c['p2']} "
              f"({c['p2_lanes'] / max(c['p2'], 1):.1f} lanes, {c['p2_tested'] / max(c['p2'], 1):.1f} testing); steps/ray {(c['p1'] + c['p2']) / c['rays']:.3f}")
