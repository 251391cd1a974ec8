#!/bin/bash
# One bench line per configuration of the reference's benchmark list (scene, resolution), each in
# its own bounded process; the script stops at the first failure.  Results: gpurun_out/configs/.
set -e -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/configs
mkdir -p $OUT
for scene in cornell pbr_metallic pbr_ior teapot helmet instancing_stress party_armadillo; do
  timeout -k 10 300 python3 -u bench.py --scene $scene --cpu-seconds 0 --spp-steps 1024 > $OUT/$scene.json 2> $OUT/$scene.err
  tail -1 $OUT/$scene.json | cut -c1-200
done
for scene in helmet party_armadillo; do
  timeout -k 10 300 python3 -u bench.py --scene $scene --width 3840 --height 2160 --cpu-seconds 0 --spp-steps 256 --spp-batch 32 > $OUT/${scene}_4k.json 2> $OUT/${scene}_4k.err
  tail -1 $OUT/${scene}_4k.json | cut -c1-200
done
python3 - <<'EOF'
import glob, json, os
rows = []
for f in sorted(glob.glob("gpurun_out/configs/*.json")):
    lines = [l for l in open(f) if l.startswith("{")]
    if not lines:
        continue
    d = json.loads(lines[-1])
    m, s = d["trace_modes"], d.get("spp") or {}
    v = d["roofline"]["visits_per_ray"]
    by = s.get("spp_per_s_by_trace_mode", {})
    rows.append("| %s | %.1f M | %.1f / %.1f | %.2f / %.2f | %.0f / %.0f | %d | %s | %.0f |" % (
        os.path.basename(f)[:-5], d["config"]["rays_per_rank"] / 1e6, v["interior"], v["triangles"], m["0"]["kernel_ms"], m["2"]["kernel_ms"],
        m["0"]["mrays_per_s"], m["2"]["mrays_per_s"], m["2"]["ids_differing_from_mode_0"],
        " / ".join("%.0f" % by[k] for k in ("0", "1", "2", "3") if k in by), s.get("mpaths_per_s", 0.0)))
print("| configuration | rays per launch | interior / triangles per ray | closest-hit ms (mode 0 / 2) | Mrays/s (mode 0 / 2) | ids differing in mode 2 | spp/s (modes 0 / 1 / 2 / 3) | Mpaths/s (mode 0) |")
print("|---|---|---|---|---|---|---|---|")
print("\n".join(rows))
EOF
