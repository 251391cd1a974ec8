#!/usr/bin/env python3
"""Trace mode 4 (one world-space tree over the whole scene) against mode 0 (reference-exact) on one GPU:
closest-hit records, any-hit flags, kernel times of modes 0 / 2 / 4 (closest) and 0 / 3 / 4 (any), visit
counts per ray, per scene, on the scene's own bounce / shadow rays and on synthetic incoherent rays.

    python tools/flat_compare.py [scene ...] [--width W --height H] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

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
    ap.add_argument("--sets", default="bounce,shadow,synthetic")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dump", default=None, help="directory for the rays whose records differ between modes 0 and 4")
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
        t0 = time.time()
        hs.update_device(ctx)
        upload_s = time.time() - t0
        ctx.set_trace_mode(4)
        layout = ctx.trace_layout()
        ctx.set_trace_mode(0)
        print(f"{name}: upload {upload_s:.3f} s, mode 4 walks layout {layout}", flush=True)
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
        rec = {"upload_s": upload_s, "layout_mode4": layout}
        for sname, rays in sets.items():
            if sname not in args.sets.split(","):
                continue
            n = rays.numel() // 8
            hits = {m: torch.empty(n * 8, dtype=torch.float32, device=dev) for m in (0, 2, 4)}
            occ = {m: torch.empty(n, dtype=torch.uint8, device=dev) for m in (0, 3, 4)}
            r = {"rays": n}
            for m in (0, 2, 4):
                ctx.set_trace_mode(m)
                for _ in range(2):
                    ctx.trace_closest(rays.data_ptr(), n, hits[m].data_ptr())
                ctx.synchronize()
                ctx.reset_stats()
                ctx.set_profiling(True)
                for _ in range(args.reps):
                    ctx.trace_closest(rays.data_ptr(), n, hits[m].data_ptr())
                ctx.synchronize()
                ctx.set_profiling(False)
                st = ctx.stats()
                ms = st.ms_closest / max(1, st.n_closest)
                r[f"closest_ms_mode{m}"] = ms
                r[f"grays_mode{m}"] = n / ms / 1e6
                r[f"spills_mode{m}"] = st.stack_spills / max(1, st.n_closest)
                r[f"drops_mode{m}"] = st.stack_drops
            for m in (0, 3, 4):
                ctx.set_trace_mode(m)
                ctx.trace_any(rays.data_ptr(), n, occ[m].data_ptr())
                ctx.synchronize()
                ctx.reset_stats()
                ctx.set_profiling(True)
                for _ in range(3):
                    ctx.trace_any(rays.data_ptr(), n, occ[m].data_ptr())
                ctx.synchronize()
                ctx.set_profiling(False)
                st = ctx.stats()
                r[f"any_ms_mode{m}"] = st.ms_any / max(1, st.n_any)
            # visit counts per ray
            nc = min(n, 1 << 20)
            for m in (0, 2, 4):
                ctx.set_trace_mode(m)
                ctr = torch.zeros(nc * 4, dtype=torch.int32, device=dev)
                hh = torch.empty(nc * 8, dtype=torch.float32, device=dev)
                ctx.trace_closest_counted(rays.data_ptr(), nc, hh.data_ptr(), ctr.data_ptr())
                ctx.synchronize()
                c = ctr.view(nc, 4).double().mean(0).cpu().numpy()
                r[f"visits_mode{m}"] = {"interior": float(c[0]), "instances": float(c[1]), "triangles": float(c[2]), "leaves": float(c[3])}
            ctx.set_trace_mode(0)
            h0 = hits[0].cpu().numpy().view(pt.abi.HIT).reshape(-1)
            for m in (2, 4):
                hm = hits[m].cpu().numpy().view(pt.abi.HIT).reshape(-1)
                ids = (h0["prim"] != hm["prim"]) | (h0["inst"] != hm["inst"])
                both = (h0["inst"] >= 0) & (hm["inst"] >= 0)
                rel = np.abs(h0["t"][both] - hm["t"][both]) / np.maximum(np.abs(h0["t"][both]), 1e-30)
                same = ~ids
                r[f"mode{m}_vs_0"] = {
                    "id_mismatch": int(ids.sum()), "id_mismatch_frac": float(ids.mean()),
                    "hit_vs_miss": int(((h0["inst"] >= 0) != (hm["inst"] >= 0)).sum()),
                    "t_rel_max": float(rel.max()) if rel.size else 0.0, "t_rel_gt_1e-5": int((rel > 1e-5).sum()),
                    "t_bits_differ_same_id": int((h0["t"].view(np.uint32) != hm["t"].view(np.uint32))[same].sum()),
                    "uv_bits_differ_same_id": int(((h0["u"].view(np.uint32) != hm["u"].view(np.uint32)) | (h0["v"].view(np.uint32) != hm["v"].view(np.uint32)))[same].sum()),
                }
                if m == 4 and args.dump and ids.any():
                    sel = np.nonzero(ids)[0][:4000]
                    rh = rays.cpu().numpy().view(pt.abi.RAY).reshape(-1)
                    np.savez(os.path.join(args.dump, f"flat_mismatch_{name}_{sname}.npz"), rays=rh[sel], h0=h0[sel], h4=hm[sel])
            o0, o3, o4 = (occ[m].cpu().numpy() for m in (0, 3, 4))
            r["any_mismatch_mode3"] = int((o0 != o3).sum())
            r["any_mismatch_mode4"] = int((o0 != o4).sum())
            r["any_mode4_missed_occluder"] = int(((o0 == 1) & (o4 == 0)).sum())
            rec[sname] = r
            print(name, sname, json.dumps(r), flush=True)
        out[name] = rec
        ctx.close()
    if args.json:
        with open(args.json, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
