"""Host logic of the 4-wide layout's clipped instance boxes (csrc/wide_bvh.h: MeshSupport, TightInstanceBox), checked
against brute force over the vertices by tests/host_harness/wide_support_check.cpp, which this test compiles with the
address and undefined-behaviour sanitizers on its host code.  Needs hipcc (the header pulls in the HIP vector types) but
no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="needs hipcc to compile the harness")
def test_mesh_support_and_clipped_boxes_against_brute_force(tmp_path):
    exe = tmp_path / "wide_support_check"
    hipcc = HIPCC if os.path.exists(HIPCC) else shutil.which("hipcc")
    build = subprocess.run(
        [hipcc, "--offload-arch=gfx950", "-O1", "-g", "-std=c++17",
         "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host", "-fno-sanitize-recover=undefined",
         os.path.join(ROOT, "tests", "host_harness", "wide_support_check.cpp"), "-o", str(exe)],
        capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-2000:]
    # 300 random meshes (1 to 3000 triangles), 8 directions each with an ample and four starved budgets, one random
    # rotation / scale / translation each: the support never falls short of brute force and equals it when the search is
    # not cut, the clipped box holds every transformed vertex, stays inside the reference's box and sits on the mesh
    run = subprocess.run([str(exe), "300"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-3000:])
    assert "0 failures" in run.stdout
