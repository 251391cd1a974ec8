#!/usr/bin/env python3
"""Scheduling study for the traversal kernel, on the CPU: how full would the wave's rounds be if every lane held
K rays instead of one?  The oracle lists each ray's steps (B box round, T triangle, E enter, L leave / finish,
C culled pop); a wave of 64 lanes x K slots is then stepped through them under the kernel's voting rules and
the instruction cost of every round is added up with a simple cost table.  No GPU involved.

usage: sched_sim.py [scene] [width] [height]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cuda_path_tracer_amd as pt  # noqa: E402
import oracle_lib as ora  # noqa: E402

COST = {"B": 80, "T": 70, "E": 90, "L": 70, "R": 70}  # vector instructions of one round of each kind
VOTE = 12                                              # ... plus the vote / bookkeeping around a round


def ray_ops(scene, w, h, iters=1):
    spec = pt.SceneSpec.load(os.path.join(ROOT, "scenes", scene + ".npz"))
    hs = spec.build(width=w, height=h)
    desc = hs.desc()
    rays = np.concatenate([ora.capture_bounce_rays(desc, w, h, it, b) for it in range(iters) for b in (1, 2, 3)])
    lib = ora.lib()
    lib.ora_trace_ops.restype = C.c_size_t
    offs = np.zeros(len(rays) + 1, np.uint64)
    r = np.ascontiguousarray(rays, pt.abi.RAY)
    need = lib.ora_trace_ops(C.byref(desc), pt.abi.vptr(r), C.c_size_t(len(r)), None, C.c_size_t(0), pt.abi.vptr(offs))
    buf = np.zeros(need, np.uint8)
    lib.ora_trace_ops(C.byref(desc), pt.abi.vptr(r), C.c_size_t(len(r)), buf.ctypes.data_as(C.c_char_p), C.c_size_t(need), pt.abi.vptr(offs))
    s = buf.tobytes().decode()
    return [s[int(offs[i]):int(offs[i + 1])] for i in range(len(rays))]


def simulate(ops, K, vote_box=24, vote_tri=12, vote_ent=8, vote_lvl=12, refill_min=12, lanes=64, waves=16, extra=0):
    """Returns (instructions per ray, {kind: (rounds, lane-rounds)}).  `extra`: instructions added to every round
    for fetching / storing the chosen slot's state when K > 1."""
    n = len(ops)
    nxt = [0]
    stats = {k: [0, 0] for k in "BTELRC"}
    instr = 0
    for wv in range(waves):
        share = ops[wv::waves]
        feed = iter(share)
        slot_ray = [[None] * K for _ in range(lanes)]  # the ray's step string
        slot_pos = [[0] * K for _ in range(lanes)]
        exhausted = False

        def want(l, s):
            r = slot_ray[l][s]
            return None if r is None else r[slot_pos[l][s]]

        def step(l, s):
            slot_pos[l][s] += 1
            if slot_pos[l][s] >= len(slot_ray[l][s]):
                slot_ray[l][s] = None

        def lanes_wanting(kind):
            out = []
            for l in range(lanes):
                for s in range(K):
                    if want(l, s) == kind:
                        out.append((l, s))
                        break
            return out

        def run_round(kind, who):
            nonlocal instr
            stats[kind][0] += 1
            stats[kind][1] += len(who)
            instr += COST[kind] + VOTE + (extra if K > 1 else 0)
            for l, s in who:
                step(l, s)

        while True:
            idle = sum(1 for l in range(lanes) for s in range(K) if slot_ray[l][s] is None)
            if not exhausted and idle >= refill_min * K:
                got = 0
                for l in range(lanes):
                    for s in range(K):
                        if slot_ray[l][s] is None:
                            r = next(feed, None)
                            if r is None:
                                exhausted = True
                                break
                            slot_ray[l][s] = r
                            slot_pos[l][s] = 0
                            got += 1
                    if exhausted:
                        break
                stats["R"][0] += 1
                stats["R"][1] += got
                instr += COST["R"] + VOTE
            if idle == lanes * K and exhausted:
                break
            drain = exhausted
            # box loop: lanes whose chosen slot holds a culled pop spend the round popping
            while True:
                pops = lanes_wanting("C")
                box = lanes_wanting("B")
                for l, s in pops:
                    if all(l != bl for bl, _ in box):
                        stats["C"][1] += 1
                        step(l, s)
                thr = max(1, (sum(1 for l in range(lanes) if any(slot_ray[l][s] is not None for s in range(K))) + 1) // 2) if drain else vote_box
                if len(box) < thr:
                    if pops:
                        stats["C"][0] += 1
                        instr += 20  # the pop + vote part of a box round that did not run
                    break
                run_round("B", box)
            box, tri, ent, lvl = (lanes_wanting(k) for k in "BTEL")
            n_max = max(len(box), len(tri), len(ent), len(lvl))
            if n_max == 0:
                continue
            if len(box) and (len(box) == n_max or drain):
                run_round("B", box)
            if len(tri) and (len(tri) >= vote_tri or len(tri) == n_max or drain):
                while True:
                    run_round("T", tri)
                    tri = lanes_wanting("T")
                    if len(tri) < vote_tri:
                        break
            if len(ent) and (len(ent) >= vote_ent or len(ent) == n_max or drain):
                run_round("E", ent)
            idle = sum(1 for l in range(lanes) for s in range(K) if slot_ray[l][s] is None)
            if len(lvl) and (len(lvl) >= vote_lvl or len(lvl) == n_max or drain or (not exhausted and len(lvl) + idle >= refill_min)):
                run_round("L", lvl)
    return instr / n, stats


def simulate_pool(ops, P, extra=20, lanes=64, waves=8, refill_min=16, min_fill=0.5):
    """A pool of P rays per wave that is not tied to lanes: every round serves up to 64 rays that want the same
    kind of step (the kind most rays want, box rounds preferred on ties); a round of a kind runs only once
    min_fill * 64 rays want it, unless nothing else can run."""
    n = len(ops)
    stats = {k: [0, 0] for k in "BTELRC"}
    instr = 0
    for wv in range(waves):
        feed = iter(ops[wv::waves])
        ray = [None] * P
        pos = [0] * P
        exhausted = False
        while True:
            idle = [i for i in range(P) if ray[i] is None]
            if not exhausted and len(idle) >= refill_min:
                got = 0
                for i in idle:
                    r = next(feed, None)
                    if r is None:
                        exhausted = True
                        break
                    ray[i], pos[i] = r, 0
                    got += 1
                stats["R"][0] += 1
                stats["R"][1] += got
                instr += COST["R"] + VOTE
            if exhausted and all(r is None for r in ray):
                break
            wants = {k: [] for k in "BTELC"}
            for i in range(P):
                if ray[i] is not None:
                    wants[ray[i][pos[i]]].append(i)
            # culled pops ride along with whatever round runs (they only touch the stack)
            for i in wants["C"]:
                stats["C"][1] += 1
                pos[i] += 1
            kind = max("BTEL", key=lambda k: (min(len(wants[k]), lanes), k == "B"))
            who = wants[kind][:lanes]
            if not who:
                continue
            stats[kind][0] += 1
            stats[kind][1] += len(who)
            instr += COST[kind] + VOTE + extra
            for i in who:
                pos[i] += 1
                if pos[i] >= len(ray[i]):
                    ray[i] = None
    return instr / n, stats


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "helmet"
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 180
    ops = ray_ops(scene, w, h)
    n = len(ops)
    tot = {k: sum(o.count(k) for o in ops) / n for k in "BTELC"}
    print(f"{scene} {w}x{h}: {n} rays; per ray " + ", ".join(f"{k} {v:.2f}" for k, v in tot.items()))
    for K, extra in ((1, 0), (2, 0), (2, 12), (3, 12), (4, 12)):
        for vb, vt in ((24, 12), (32, 16), (40, 24), (48, 32)):
            per_ray, st = simulate(ops, K, vote_box=vb, vote_tri=vt, waves=8, extra=extra)
            util = {k: (v[1] / (64 * v[0]) if v[0] else 0.0) for k, v in st.items()}
            print(f"K={K} extra={extra:2d} votes {vb}/{vt}: {per_ray:6.1f} instr/ray   utilisation " +
                  " ".join(f"{k} {util[k]:.2f}" for k in "BTELR") + f"   rounds/ray " + " ".join(f"{k} {st[k][0] / n:.3f}" for k in "BTELRC"))


def pool_table(ops):
    n = len(ops)
    for P in (64, 96, 128, 192, 256):
        for extra in (0, 20, 30):
            per_ray, st = simulate_pool(ops, P, extra=extra)
            util = {k: (v[1] / (64 * v[0]) if v[0] else 0.0) for k, v in st.items()}
            print(f"pool {P:3d} extra={extra:2d}: {per_ray:6.1f} instr/ray   utilisation " + " ".join(f"{k} {util[k]:.2f}" for k in "BTELR") +
                  "   rounds/ray " + " ".join(f"{k} {st[k][0] / n:.3f}" for k in "BTELR"))


if __name__ == "__main__":
    if os.environ.get("POOL"):
        pool_table(ray_ops(sys.argv[1] if len(sys.argv) > 1 else "helmet", 320, 180))
        sys.exit(0)
    main()
