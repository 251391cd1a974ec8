#!/usr/bin/env python3
"""Trace modes side by side on one GPU: closest-hit records of mode 2 (4-wide BVH; mode 3 = 2 + any-hit culling with t_max) against mode 0
(reference-exact), any-hit flags of modes 1 and 2 against mode 0, kernel times of each, per scene,
on synthetic incoherent rays and on the scene's own bounce / shadow rays.

    python tools/mode_compare.py [scene ...] [--width W --height H] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenes", nargs="*", default=["helmet"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--synthetic", type=int, default=1 << 22)
    ap.add_argument("--json", default=None)
    ap.add_argument("--dump", default=None, help="directory for the rays whose ids differ between modes 0 and 2")
    ap.add_argument("--sets", default="bounce,shadow,synthetic")
    args = ap.parse_args()
    import torch

    import cuda_path_tracer_amd as pt

    dev = torch.device("cuda", 0)
    out = {}
    for name in args.scenes:
        W, H = args.width, args.height
        spec = pt.SceneSpec.load(os.path.join(ROOT, "scenes", name + ".npz"))
        hs = spec.build(width=W, height=H, max_depth=8)
        ctx = pt.Context(0)
        hs.update_device(ctx)
        ctx.integrator_init(W, H)
        sets = {}
        tmp = torch.empty(W * H * 8, dtype=torch.float32, device=dev)
        chunks, sh_chunks = [], []
        for it in range(args.iters):
            for b in (1, 2, 3):
                n = ctx.capture_bounce_rays(W, H, it, b, tmp.data_ptr(), W * H)
                chunks.append(tmp[: n * 8].clone())
            for b in (0, 1):
                n = ctx.capture_shadow_rays(W, H, it, b, tmp.data_ptr(), W * H)
                sh_chunks.append(tmp[: n * 8].clone())
        sets["bounce"] = torch.cat(chunks)
        sets["shadow"] = torch.cat(sh_chunks)
        syn = torch.empty(args.synthetic * 8, dtype=torch.float32, device=dev)
        ctx.generate_incoherent_rays(syn.data_ptr(), args.synthetic, 0)
        sets["synthetic"] = syn
        del tmp, chunks, sh_chunks
        rec = {}
        for sname, rays in sets.items():
            if sname not in args.sets.split(","):
                continue
            n = rays.numel() // 8
            hits = {m: torch.empty(n * 8, dtype=torch.float32, device=dev) for m in (0, 2)}
            occ = {m: torch.empty(n, dtype=torch.uint8, device=dev) for m in (0, 1, 2, 3)}
            times = {}
            for m in (0, 1, 2, 3):
                ctx.set_trace_mode(m)
                if m in (0, 2):
                    for _ in range(2):
                        ctx.trace_closest(rays.data_ptr(), n, hits[m].data_ptr())
                    ctx.synchronize()
                    ctx.reset_stats()
                    ctx.set_profiling(True)
                    for _ in range(5):
                        ctx.trace_closest(rays.data_ptr(), n, hits[m].data_ptr())
                    ctx.synchronize()
                    ctx.set_profiling(False)
                    st = ctx.stats()
                    times[f"closest_ms_mode{m}"] = st.ms_closest / max(1, st.n_closest)
                    times[f"spills_per_launch_mode{m}"] = st.stack_spills / max(1, st.n_closest)
                    times[f"drops_mode{m}"] = st.stack_drops
                ctx.trace_any(rays.data_ptr(), n, occ[m].data_ptr())
                ctx.synchronize()
                ctx.reset_stats()
                ctx.set_profiling(True)
                for _ in range(3):
                    ctx.trace_any(rays.data_ptr(), n, occ[m].data_ptr())
                ctx.synchronize()
                ctx.set_profiling(False)
                st = ctx.stats()
                times[f"any_ms_mode{m}"] = st.ms_any / max(1, st.n_any)
            ctx.set_trace_mode(0)
            h0 = hits[0].cpu().numpy().view(pt.abi.HIT).reshape(-1)
            h2 = hits[2].cpu().numpy().view(pt.abi.HIT).reshape(-1)
            ids = (h0["prim"] != h2["prim"]) | (h0["inst"] != h2["inst"])
            both = (h0["inst"] >= 0) & (h2["inst"] >= 0)
            rel = np.abs(h0["t"][both] - h2["t"][both]) / np.maximum(np.abs(h0["t"][both]), 1e-30)
            hit_miss = int(((h0["inst"] >= 0) != (h2["inst"] >= 0)).sum())
            o0, o1, o2, o3 = (occ[m].cpu().numpy() for m in (0, 1, 2, 3))
            if args.dump and ids.any():
                sel = np.nonzero(ids)[0][:2000]
                rh = rays.cpu().numpy().view(pt.abi.RAY).reshape(-1)
                np.savez(os.path.join(args.dump, f"mismatch_{name}_{sname}.npz"), rays=rh[sel], h0=h0[sel], h2=h2[sel])
            rec[sname] = {"rays": n, "id_mismatch": int(ids.sum()), "id_mismatch_frac": float(ids.mean()), "hit_vs_miss": hit_miss,
                          "t_rel_max": float(rel.max()) if rel.size else 0.0, "t_rel_gt_1e-5": int((rel > 1e-5).sum()),
                          "t_bits_differ_same_id": int((h0["t"].view(np.uint32) != h2["t"].view(np.uint32))[~ids].sum()),
                          "any_mismatch_mode1": int((o0 != o1).sum()), "any_mismatch_mode2": int((o0 != o2).sum()), "any_mismatch_mode3": int((o0 != o3).sum()),
                          "any_mode3_missed_occluder": int(((o0 == 1) & (o3 == 0)).sum()),
                          **times, "grays_mode0": n / times["closest_ms_mode0"] / 1e6, "grays_mode2": n / times["closest_ms_mode2"] / 1e6}
            print(name, sname, json.dumps(rec[sname]), flush=True)
        out[name] = rec
        ctx.close()
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
