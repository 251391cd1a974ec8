"""Pins the CPU oracle itself.  The reference ships no tests or golden vectors and cannot be
compiled in this image (parity unpinned), so the oracle is checked against what can be known
independently: closed-form intersections, integer-exact PCG values computed in Python, the
brute-force all-triangles search (the reference's own cross-check), structural BVH invariants
and alias-table reconstruction of the input distribution."""
import os

import numpy as np
import pytest

from conftest import make_vine_scene, rays_from_arrays

FLT_MAX = np.float32(3.4028235e38)


# ---------------------------------------------------------------- slab test (math/intersection.h:9-35)
def test_aabb_known_answers(ora):
    lo, hi = [-1, -1, -1], [1, 1, 1]
    hit, t = ora.intersect_aabb([0, 0, -5], [0, 0, 1], lo, hi)
    assert hit and t == 4.0
    hit, t = ora.intersect_aabb([0, 0, 0], [0, 0, 1], lo, hi)  # origin inside: tnear negative
    assert hit and t == -1.0
    assert not ora.intersect_aabb([0, 0, 5], [0, 0, 1], lo, hi)[0]  # box behind the ray (tfar <= 0)
    assert not ora.intersect_aabb([0, 0, -5], [0, 0, 1], lo, hi, t_max=3.5)[0]  # tnear >= t_max
    assert ora.intersect_aabb([0, 0, -5], [0, 0, 1], lo, hi, t_max=4.5)[0]
    assert not ora.intersect_aabb([2, 0, -5], [0, 0, 1], lo, hi)[0]  # parallel miss, inv_d = inf
    # axis-parallel ray lying exactly in a face plane: 0 * inf = NaN, fminf/fmaxf keep the other
    # operand (-inf on the max face, +inf on the min face), so the slab closes and the ray misses
    assert not ora.intersect_aabb([1, 0, -5], [0, 0, 1], lo, hi)[0]
    assert not ora.intersect_aabb([-1, 0, -5], [0, 0, 1], lo, hi)[0]
    # ... while a ray just inside the face hits
    hit, t = ora.intersect_aabb([0.999, 0, -5], [0, 0, 1], lo, hi)
    assert hit and t == 4.0
    # diagonal through a corner
    d = np.array([1, 1, 1], np.float32) / np.sqrt(np.float32(3))
    hit, t = ora.intersect_aabb([-3, -3, -3], d, lo, hi)
    assert hit and abs(t - 2 * np.sqrt(3)) < 1e-5


# -------------------------------------------------- Moeller-Trumbore (math/intersection.h:37-71)
def test_triangle_known_answers(ora):
    v0, v1, v2 = [0, 0, 0], [1, 0, 0], [0, 1, 0]
    hit, t, u, v = ora.intersect_triangle([0.25, 0.5, 2], [0, 0, -1], v0, v1, v2)
    assert hit and t == 2.0 and u == 0.25 and v == 0.5
    hit, t, u, v = ora.intersect_triangle([0.25, 0.5, -2], [0, 0, 1], v0, v1, v2)  # two-sided
    assert hit and t == 2.0
    assert not ora.intersect_triangle([0.25, 0.5, 2], [1, 0, 0], v0, v1, v2)[0]  # parallel: |det| < 1e-9
    assert not ora.intersect_triangle([0.25, 0.5, -2], [0, 0, -1], v0, v1, v2)[0]  # behind: t <= 1e-9
    assert not ora.intersect_triangle([0.25, 0.5, 2], [0, 0, -1], v0, v1, v2, t_max=1.5)[0]  # t > t_max
    assert ora.intersect_triangle([0.25, 0.5, 2], [0, 0, -1], v0, v1, v2, t_max=2.0)[0]  # t == t_max accepted
    # edges and vertices are inclusive (u, v in [0,1], u+v <= 1)
    assert ora.intersect_triangle([0.5, 0.5, 1], [0, 0, -1], v0, v1, v2)[0]  # on the hypotenuse
    assert ora.intersect_triangle([0, 0, 1], [0, 0, -1], v0, v1, v2)[0]  # vertex 0
    assert ora.intersect_triangle([1, 0, 1], [0, 0, -1], v0, v1, v2)[0]  # vertex 1
    assert not ora.intersect_triangle([0.51, 0.5, 1], [0, 0, -1], v0, v1, v2)[0]
    assert not ora.intersect_triangle([-1e-3, 0.5, 1], [0, 0, -1], v0, v1, v2)[0]
    # a ray through the shared edge of two triangles hits both (no cracks)
    a = ora.intersect_triangle([0.5, 0.5, 1], [0, 0, -1], [0, 0, 0], [1, 0, 0], [0, 1, 0])[0]
    b = ora.intersect_triangle([0.5, 0.5, 1], [0, 0, -1], [1, 1, 0], [0, 1, 0], [1, 0, 0])[0]
    assert a and b


# -------------------------------------------------------------- PCG stream (independent.h:14-32)
def _pcg_reference(px, py, iteration, n):
    M32, M64 = (1 << 32) - 1, (1 << 64) - 1
    key = ((px & 0xFFFF) | ((py & 0xFFFF) << 16)) ^ ((iteration * 0x9E3779B9) & M32)
    state = (key * 747796405 + 2891336453) & M32
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & M32
    hashed = ((word >> 22) ^ word) & M32
    s = ((hashed << 32) | (hashed ^ 0xDEADBEEF)) & M64
    out = []
    for _ in range(n):
        old = s
        s = (old * 6364136223846793005 + 1) & M64
        x = (((old >> 18) ^ old) >> 27) & M32
        r = old >> 59
        v = ((x >> r) | (x << ((-r) & 31))) & M32
        out.append(np.float32(v) * np.float32(1.0 / 4294967296.0))
    return np.array(out, np.float32)


@pytest.mark.parametrize("px,py,it", [(0, 0, 0), (17, 5, 3), (1919, 1079, 8123), (65535, 65535, 1)])
def test_pcg_stream_integer_exact(ora, px, py, it):
    assert np.array_equal(ora.pcg_stream(px, py, it, 64), _pcg_reference(px, py, it, 64))


# ------------------------------------------------------------ alias tables (discrete_sampler.cpp)
@pytest.mark.parametrize("weights", [[1.0], [1, 1, 1, 1], [0.1, 0.2, 0.3, 0.4], [5, 0, 0, 1], list(np.linspace(0.01, 3, 97))])
def test_alias_table_reconstructs_distribution(ora, weights):
    w = np.asarray(weights, np.float32)
    st, prob, alias = ora.alias_build(w)
    assert st == 0
    n = len(w)
    mass = np.zeros(n, np.float64)
    for i in range(n):
        mass[i] += prob[i] / n
        if prob[i] < 1.0:
            assert alias[i] < n
            mass[alias[i]] += (1.0 - prob[i]) / n
    assert np.allclose(mass, w / w.sum(), atol=3e-5)


def test_alias_table_rejects_zero_sum(ora):
    st, _, _ = ora.alias_build([0.0, 0.0])
    assert st != 0


# ------------------------------------------------------------------ BVH structure (bvh.cpp:5-332)
def _check_bvh(arr):
    geos, blases, nodes, tri = arr["geometries"], arr["blases"], arr["blas_nodes"], arr["tri_indices"]
    pos, idx = arr["positions"], arr["indices"]
    done = set()
    for g in geos:
        if int(g["blas_index"]) in done:
            continue
        done.add(int(g["blas_index"]))
        b = blases[g["blas_index"]]
        assert b["node_count"] > 0 and b["index_count"] == g["index_count"] // 3
        perm = tri[b["first_index"]: b["first_index"] + b["index_count"]]
        assert np.array_equal(np.sort(perm), np.arange(b["index_count"], dtype=np.uint32))  # a permutation
        seen = np.zeros(b["index_count"], bool)
        stack = [int(b["first_node"])]
        visited = 0
        while stack:
            k = stack.pop()
            visited += 1
            n = nodes[k]
            if n["primitive_count"] > 0:
                rng = slice(int(n["first_primitive"]), int(n["first_primitive"] + n["primitive_count"]))
                assert not seen[rng].any()
                seen[rng] = True
                tris = perm[rng]
                vid = idx[g["first_index"] + (tris[:, None] * 3 + np.arange(3)[None, :])].reshape(-1)
                p = pos[vid]
                assert np.array_equal(p.min(axis=0), n["min"]) and np.array_equal(p.max(axis=0), n["max"])  # tight
            else:
                l, r = int(n["first_primitive"]), int(n["first_primitive"]) + 1
                assert l > k  # children are allocated after their parent, adjacently
                assert np.array_equal(np.minimum(nodes[l]["min"], nodes[r]["min"]), n["min"])
                assert np.array_equal(np.maximum(nodes[l]["max"], nodes[r]["max"]), n["max"])
                stack += [l, r]
        assert seen.all() and visited == b["node_count"]
    # TLAS: slot 0 is the root copy, leaves 1..n in instance order, every instance reachable once
    tl = arr["tlas_nodes"]
    n = len(geos)
    assert len(tl) == 2 * n
    for i in range(n):
        assert tl[1 + i]["left_right"] == 0 and tl[1 + i]["instance"] == i
    reach = []
    stack = [0]
    while stack:
        node = tl[stack.pop()]
        if node["left_right"] == 0:
            reach.append(int(node["instance"]))
        else:
            stack += [int(node["left_right"] & 0xFFFF), int(node["left_right"] >> 16)]
    assert sorted(reach) == list(range(n))


@pytest.mark.parametrize("name", ["teapot", "cornell", "instancing_stress"])
def test_oracle_bvh_invariants(ora, scene_factory, name):
    hs = scene_factory(name, 64, 64, 4)
    arr = hs.arrays()
    built = ora.tlas_build(arr["positions"], arr["indices"], arr["geometries"])
    arr.update(built)
    _check_bvh(arr)


# ------------------------------------------------------------- traversal vs brute force (naive)
@pytest.mark.parametrize("name", ["teapot", "cornell", "tiny"])
def test_oracle_bvh_traversal_equals_brute_force(ora, pt, scene_factory, tiny_scene, name):
    hs = tiny_scene if name == "tiny" else scene_factory(name, 64, 64, 4)
    d = hs.desc()
    n = 1500 if name == "teapot" else 4000
    rays = ora.generate_incoherent_rays(d, n, 11)
    hits = ora.trace_closest(d, rays, nthreads=4)
    naive = ora.trace_naive(d, rays, nthreads=4)
    assert np.array_equal(hits["inst"] >= 0, naive["inst"] >= 0)
    both = hits["inst"] >= 0
    assert both.sum() > n // 10
    # same winner except where two candidates are within rounding of each other
    same = (hits["prim"] == naive["prim"]) & (hits["inst"] == naive["inst"])
    rel = np.abs(hits["t"] - naive["t"]) / np.maximum(np.abs(naive["t"]), 1e-20)
    assert (rel[both] < 2e-5).all()
    assert (~same[both]).sum() <= max(1, both.sum() // 200)
    # any-hit agrees with "closest hit exists within tmax" away from the boundary
    rays_any = rays.copy()
    rays_any["tmax"] = np.where(both, hits["t"] * np.float32(1.5), np.float32(10.0))
    occ = ora.trace_any(d, rays_any, nthreads=4)
    assert np.array_equal(occ.astype(bool), both)
    rays_any["tmax"] = np.where(both, hits["t"] * np.float32(0.5), np.float32(10.0))
    occ = ora.trace_any(d, rays_any, nthreads=4)
    assert not occ.any()


def test_oracle_sphere_analytic(ora, pt, scene_factory):
    """Closest hits on the tessellated radius-0.5 sphere (scaled 3x, centred at (-1,4,-1) in the
    legacy cornell scene) stay within the tessellation error of the analytic sphere."""
    hs = scene_factory("cornell", 64, 64, 4)
    d = hs.desc()
    arr = hs.arrays()
    # instance 6 is the first sphere of Scenes/cornell.json
    g = arr["geometries"][6]
    T = g["transform"].reshape(4, 4).T
    centre = T[:3, 3]
    radius = 0.5 * T[0, 0]
    rs = np.random.RandomState(3)
    dirs = rs.normal(size=(2000, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    origins = (centre[None, :] + 0.05 * rs.normal(size=(2000, 3))).astype(np.float32)  # inside the sphere
    hits = ora.trace_closest(d, rays_from_arrays(pt, origins, dirs), nthreads=4)
    inside = hits["inst"] == 6
    assert inside.mean() > 0.99
    p = origins + hits["t"][:, None] * dirs
    dist = np.linalg.norm(p - centre[None, :], axis=1)
    assert (np.abs(dist[inside] - radius) < 0.004 * radius + 1e-4).all()


# --------------------------------------------------------------------- camera and tone map
def test_camera_rays(ora):
    cam = ora.camera_update([0, 0, 5], [0, 0, 0], [0, 1, 0], 90.0, 4, 2)
    assert np.allclose(list(cam.front), [0, 0, -1]) and np.allclose(list(cam.right), [1, 0, 0]) and np.allclose(list(cam.up), [0, 1, 0])
    # fovy 90 -> half height 1, aspect 2 -> pixel length 1 in both axes
    assert np.allclose(list(cam.pixel_length), [1.0, 1.0])
    import ctypes as C
    o, dr = np.zeros(3, np.float32), np.zeros(3, np.float32)
    fp = C.POINTER(C.c_float)
    ora.lib().ora_camera_ray(C.byref(cam), 2, 1, 0.0, 0.0, o.ctypes.data_as(fp), dr.ctypes.data_as(fp))
    assert np.allclose(o, [0, 0, 5]) and np.allclose(dr, [0, 0, -1])  # pixel (W/2, H/2) corner = optical axis
    ora.lib().ora_camera_ray(C.byref(cam), 0, 0, 0.0, 0.0, o.ctypes.data_as(fp), dr.ctypes.data_as(fp))
    assert np.allclose(dr, np.array([-2, 1, -1]) / np.sqrt(6), atol=1e-6)  # image y points down


def test_tonemap_known_values(ora):
    film = np.zeros((1, 4, 3), np.float32)
    film[0, 0] = 0.0
    film[0, 1] = 2.0   # two iterations -> radiance 1.0
    film[0, 2] = 2000.0
    film[0, 3] = [0.36, 0.36, 0.36]
    out = ora.tonemap(film, iteration=1)
    aces = lambda x: min(max((x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14), 0.0), 1.0)
    assert list(out[0, 0]) == [0, 0, 0, 255]
    assert out[0, 1, 0] == int(aces(1.0) * 255.0)
    assert out[0, 2, 0] == 255
    assert out[0, 3, 0] == int(np.float32(aces(0.18)) * 255.0) or abs(int(out[0, 3, 0]) - int(aces(0.18) * 255.0)) <= 1


def test_render_is_deterministic_and_thread_independent(ora, tiny_scene):
    d = tiny_scene.desc()
    a, sa = ora.render(d, 64, 48, 2, nthreads=1)
    b, sb = ora.render(d, 64, 48, 2, nthreads=4)
    assert np.array_equal(a, b) and sa == sb
    assert np.isfinite(a).all() and a.mean() > 0.01


# ------------------------------------------------------------------ debug integrator
def _debug_state(pt, visualize_bvh=0, bvh_type=0, depth=500, mode=0):
    return pt.abi.IntegratorState(visualize_bvh, bvh_type, depth, mode)


def test_debug_heat_map_follows_the_visit_counters(pt, ora, tiny_scene):
    """BoundsHit / TLAS / TrianglesHit colours are functions of the traversal counters of the
    pixel-centre camera ray (debug.cu:119-178), recomputed here in float64."""
    w, h = 64, 48
    desc = tiny_scene.desc()
    rays = np.zeros(w * h, pt.abi.RAY)
    o, d = np.zeros(3, np.float32), np.zeros(3, np.float32)
    for y in range(h):
        for x in range(w):
            ora.lib().ora_camera_ray(desc.camera, x, y, 0.5, 0.5, pt.abi.fptr(o), pt.abi.fptr(d))
            rays[y * w + x]["origin"], rays[y * w + x]["direction"], rays[y * w + x]["tmax"] = o, d, np.float32(3.4028235e38)
    tris = np.zeros(w * h, np.int64)
    visits = np.zeros(w * h, np.int64)
    for i in range(w * h):
        _, c = ora.trace_closest(desc, rays[i:i + 1], counters=True)
        tris[i] = c[2]
        visits[i] = c[0] + c[1] + c[3]
    image, rgba = ora.debug_render(desc, w, h, _debug_state(pt, 1, pt.abi.PT_BVH_TRIANGLES_HIT))
    t = np.clip(np.log(tris + 1.0) / np.log(51.0), 0, 1).reshape(h, w)
    assert np.abs(image[..., 0] - t).max() < 1e-5
    assert np.abs(image[..., 1] - (0.3 * (1 - t) + t)).max() < 1e-5
    assert not image[..., 2].any() and (rgba[..., 3] == 255).all()

    def heat(depth, scale):  # HeatMapColor, debug.cu:42-49 + math/color.h:19-87
        lam = 380.0 + 400.0 * depth / scale

        def g(mu, s1, s2):
            return np.exp(-0.5 * ((lam - mu) * np.where(lam < mu, s1, s2)) ** 2)
        X = 0.362 * g(442.0, 0.0624, 0.0374) + 1.056 * g(599.8, 0.0264, 0.0323) - 0.065 * g(501.1, 0.0490, 0.0382)
        Y = 0.821 * g(568.8, 0.0213, 0.0247) + 0.286 * g(530.9, 0.0613, 0.0322)
        Z = 1.217 * g(437.0, 0.0845, 0.0278) + 0.681 * g(459.0, 0.0385, 0.0725)
        rgb = np.stack([3.240479 * X - 1.537150 * Y - 0.498535 * Z, -0.969256 * X + 1.875991 * Y + 0.041556 * Z,
                        0.055648 * X - 0.204043 * Y + 1.057311 * Z], -1).clip(0, 1)
        return np.where(rgb <= 0.0031308, 12.92 * rgb, 1.055 * rgb ** (1 / 2.4) - 0.055)

    image, rgba = ora.debug_render(desc, w, h, _debug_state(pt, 1, pt.abi.PT_BVH_BOUNDS_HIT, 60))
    ref = heat(visits.reshape(h, w).astype(np.float64), 60.0)
    assert np.abs(image - ref).max() < 2e-4
    assert visits.max() > visits.min() and image.std() > 1e-2
    # 8-bit conversion: truncation of p * 255 (debug.cu:190-192)
    assert np.abs(rgba[..., :3].astype(int) - np.floor(ref * 255.0).clip(0, 255).astype(int)).max() <= 1
    # TLAS nodes visited never exceed all nodes visited
    tl, _ = ora.debug_render(desc, w, h, _debug_state(pt, 1, pt.abi.PT_BVH_TLAS, 60))
    assert tl.shape == image.shape and np.isfinite(tl).all()


def test_debug_attribute_views(pt, ora, tiny_scene):
    """DebugView (debug.cu:51-117): background 0.4 off the scene; normals are unit vectors; depth is
    the clamped normalised hit distance; albedo is zero on materials without an albedo view."""
    w, h = 64, 48
    desc = tiny_scene.desc()
    normals, _ = ora.debug_render(desc, w, h, _debug_state(pt, mode=pt.abi.PT_DEBUG_GEOMETRIC_NORMALS))
    depth, _ = ora.debug_render(desc, w, h, _debug_state(pt, mode=pt.abi.PT_DEBUG_DEPTH))
    uv, _ = ora.debug_render(desc, w, h, _debug_state(pt, mode=pt.abi.PT_DEBUG_UV))
    albedo, rgba = ora.debug_render(desc, w, h, _debug_state(pt, mode=pt.abi.PT_DEBUG_ALBEDO))
    bg = np.all(normals == np.float32(0.4), axis=2)
    assert 0 < bg.sum() < w * h
    assert np.allclose(np.linalg.norm(normals[~bg], axis=1), 1.0, atol=1e-5)
    assert np.all(depth[bg] == np.float32(0.4)) and np.all(uv[bg] == np.float32(0.4))
    assert (depth[~bg] >= 0).all() and (depth[~bg] <= 1).all() and not uv[~bg][:, 2].any()
    assert np.all(depth[..., 0] == depth[..., 1])
    # the checkerboard floor shows exactly its two colours somewhere; mirror / dielectric / emissive give 0
    flat = albedo[~bg].reshape(-1, 3)
    colours = {tuple(round(float(x), 4) for x in c) for c in np.unique(flat, axis=0)}
    assert (0.8, 0.8, 0.8) in colours and (0.2, 0.3, 0.8) in colours and (0.0, 0.0, 0.0) in colours
    assert (rgba[..., :3][bg] == 102).all()  # (int)(0.4f * 255.0) = 102
    none_img, none_rgba = ora.debug_render(desc, w, h, _debug_state(pt))
    assert not none_img.any() and not none_rgba.any()


@pytest.mark.parametrize("depth,nearest_is_deepest", [(40, True), (63, True), (64, True), (65, True), (100, True), (100, False)])
def test_oracle_stack_overflow_follows_the_reference_stack(pt, ora, depth, nearest_is_deepest):
    """utils/stack.h:8-11: 64 entries, pushes beyond that are ignored.  The vine BLAS pushes one leaf
    per level; an independent replay of IntersectBLAS (kernel/intersection.h:52-137) with a capped
    Python list must name the same triangle as the oracle."""
    desc, keep = make_vine_scene(pt, depth, nearest_is_deepest)
    nodes = keep["nodes"]
    o = np.array([[-10.0, 0.25, 0.25]], np.float32)
    d = np.array([[1.0, 0.0, 0.0]], np.float32)
    ref = ora.trace_closest(desc, rays_from_arrays(pt, o, d))

    def box_entry(i):  # the ray runs along +x through every box of the vine
        return float(np.float32(nodes[i]["min"][0]) - np.float32(-10.0))

    stack, best, prim = [(0, 0.0)], np.inf, -1

    def push(item):
        if len(stack) < 64:
            stack.append(item)
    while stack:
        i, entry = stack.pop()
        if entry >= best:
            continue
        nd = nodes[i]
        if nd["primitive_count"] > 0:
            k = int(nd["first_primitive"])
            t = float(np.float32(nd["min"][0] + np.float32(0.01)) + 10.0)  # the triangle's plane
            if t < best:
                best, prim = t, k
            continue
        left, right = int(nd["first_primitive"]), int(nd["first_primitive"]) + 1
        tl, tr = box_entry(left), box_entry(right)
        hl, hr = tl < best, tr < best
        if hl and hr:
            if tl < tr:
                push((right, tr))
                push((left, tl))
            else:
                push((left, tl))
                push((right, tr))
        elif hl:
            push((left, tl))
        elif hr:
            push((right, tr))
    assert int(ref["prim"][0]) == prim
    true_nearest = depth if nearest_is_deepest else 0
    assert (prim == true_nearest) == (depth < 64 or not nearest_is_deepest)


def test_texture_filter_options(pt, ora):
    """The software filter's two options (8-bit fixed-point weights, sRGB decode of tagged images) as the
    oracle defines them: the default is plain fp32 bilinear filtering of the stored texels."""
    hs = pt.HostScene()
    hs.set_camera([0, 0, 3], [0, 0, 0], [0, 1, 0], 45.0, 8, 8)
    rs = np.random.RandomState(11)
    texels = rs.rand(4, 4, 3).astype(np.float32)
    tagged = hs.add_image_texture(hs.add_image(texels, srgb=True), 4, 4, 3)
    plain = hs.add_image_texture(hs.add_image(texels, srgb=False), 4, 4, 3)
    mat = hs.add_material(pt.abi.PT_MAT_LAMBERTIAN, tagged)
    p = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    first = hs.add_mesh(p, np.tile([[0, 0, 1]], (3, 1)).astype(np.float32), p[:, :2].copy(), np.arange(3, dtype=np.uint32))
    hs.add_instance(first, 3, 3, mat, np.eye(4, dtype=np.float32).reshape(-1), hs.new_blas_index())
    hs.finalize()
    desc = hs.desc()
    try:
        u, v = 0.3371, 0.6113  # x = u*4 - 0.5 = 0.8484: weight 0.8484 -> 217/256 = 0.84766 in fixed point
        base = ora.sample_texture3(desc, tagged, u, v)
        assert np.array_equal(base, ora.sample_texture3(desc, plain, u, v))
        x, y = u * 4 - 0.5, v * 4 - 0.5
        fx, fy = x - np.floor(x), y - np.floor(y)
        i0, j0 = int(np.floor(x)), int(np.floor(y))
        taps = texels[[j0, j0, j0 + 1, j0 + 1], [i0, i0 + 1, i0, i0 + 1]].astype(np.float64)

        def bilinear(t, fx, fy):
            return (t[0] * (1 - fx) + t[1] * fx) * (1 - fy) + (t[2] * (1 - fx) + t[3] * fx) * fy
        assert np.allclose(base, bilinear(taps, fx, fy), atol=1e-6)
        ora.set_texture_filter(1)
        q = ora.sample_texture3(desc, tagged, u, v)
        qx, qy = np.floor(fx * 256 + 0.5) / 256, np.floor(fy * 256 + 0.5) / 256
        assert np.allclose(q, bilinear(taps, qx, qy), atol=1e-6) and not np.array_equal(q, base)
        assert np.abs(q - base).max() <= 1.0 / 256
        ora.set_texture_filter(2)
        lin = np.where(taps <= 0.04045, taps / 12.92, ((taps + 0.055) / 1.055) ** 2.4)
        assert np.allclose(ora.sample_texture3(desc, tagged, u, v), bilinear(lin, fx, fy), atol=1e-6)
        assert np.array_equal(ora.sample_texture3(desc, plain, u, v), base)  # untagged images are never decoded
    finally:
        ora.set_texture_filter(0)


def test_any_hit_with_negative_recomputed_distance_counts_as_free(pt, ora, scene_factory):
    """A shadow ray of the helmet scene (bounce 1, captured on the GPU) whose origin touches a triangle: the
    triangle test accepts it at t = 6.5e-6, the distance recomputed in world space comes out negative, and the
    reference's searches then report no hit at all (kernel/intersection.h:139-149, :229-232), for the any-hit
    query and for the closest-hit query alike, although brute force over the triangles finds the hit."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "any_hit_negative_distance.npz"))
    hs = scene_factory(str(g["scene"]), int(g["width"]), int(g["height"]), 8)
    desc = hs.desc()
    rays = g["rays"].astype(pt.abi.RAY)
    assert list(ora.trace_any(desc, rays)) == [0]
    assert ora.trace_closest(desc, rays)["inst"][0] == -1
    brute = ora.trace_naive(desc, rays)
    assert brute["inst"][0] >= 0 and 0 < brute["t"][0] < 1e-5


def test_azimuth_sincos_is_accurate(ora):
    """The azimuth sin / cos is a fixed sequence of fp32 operations shared (by definition, not by code) with the
    device; against double precision over [0, 2 pi] and at the quadrant seams its error stays below 1.5e-7,
    which is what the single-precision library functions it stands in for deliver too."""
    phi = np.linspace(0.0, 2 * np.pi, 2000001).astype(np.float32)
    seams = np.array([k * np.pi / 4 for k in range(9)], np.float32)
    phi = np.concatenate([phi, seams, np.nextafter(seams, np.float32(100)), np.nextafter(seams, np.float32(-100)).clip(0)]).astype(np.float32)
    s, c = ora.sincos_azimuth(phi)
    assert np.abs(s - np.sin(phi.astype(np.float64))).max() < 1.5e-7
    assert np.abs(c - np.cos(phi.astype(np.float64))).max() < 1.5e-7
    assert np.abs(s.astype(np.float64) ** 2 + c.astype(np.float64) ** 2 - 1.0).max() < 4e-7
    assert s[0] == 0.0 and c[0] == 1.0
