"""Scene ingestion for the host facade: scene JSON (current schema and the legacy CIS-565 style
schema), glTF 2.0, Wavefront OBJ, image decode, and a compact "cooked" container (.npz) so
that GPU hosts need neither the asset files nor an image decoder.

A SceneSpec is the list of calls a loader makes against the facade's Scene (meshes,
textures, materials, instances, camera, integrator) in the order the reference's loader makes
them, so material / texture / instance ids come out as scene.cpp:156-578 assigns them:
  * texture 0 / material 0 are the defaults (scene.cpp:235-260);
  * named materials follow in lexicographic key order (nlohmann::json's std::map,
    scene.cpp:337-356), each ParseTexture call appending a new texture (scene.h:76-179);
  * glTF objects append their own materials and images at load time (scene.cpp:439-462) and
    emit one instance per primitive per node in the loader's stack order
    (mesh/gltf_model.cpp:175-193, scene.cpp:472-512); every object entry reloads its file.

Legacy schema (capitalised "Materials"/"Camera"/"Objects", TYPE cube/sphere, flat colour
arrays): the reference's parser at this commit cannot load these files, so the adapter below
is this project's own definition: cube = unit cube [-0.5,0.5]^3 with face normals, sphere =
radius 0.5 UV sphere (32 stacks x 64 slices, smooth normals), RoughDielectric = Dielectric
ignoring ROUGHNESS.
"""
from __future__ import annotations

import base64
import hashlib
import io
import json
import os
import struct
from dataclasses import dataclass, field

import numpy as np

from . import abi
from .runtime import HostScene, gltf_local_matrix, matrix_mul, transform_matrix

F32 = np.float32


# ------------------------------------------------------------------------------- spec
@dataclass
class SceneSpec:
    max_depth: int = 4
    max_iterations: int = 1000
    camera: dict = field(default_factory=dict)  # eye, look_at, up, fovy, width, height
    ops: list = field(default_factory=list)      # ordered facade calls
    arrays: dict = field(default_factory=dict)   # name -> ndarray (mesh data, image bytes)
    _digests: dict = field(default_factory=dict, repr=False)

    # ---- builders used by the loaders
    def _arr(self, a: np.ndarray) -> str:
        a = np.ascontiguousarray(a)
        digest = hashlib.sha1(a.tobytes() + str((a.shape, a.dtype.str)).encode()).hexdigest()
        key = self._digests.get(digest)
        if key is None:
            key = f"a{len(self.arrays)}"
            self.arrays[key] = a
            self._digests[digest] = key
        return key

    def add_image_file(self, file_bytes: bytes, mode: str, channel, rule: str, srgb: bool = False):
        """Records an encoded image (JPEG/PNG bytes); returns (width, height).  `mode` is the PIL
        conversion ("RGB" / "RGBA" / "" = as stored), `channel` None = first three channels."""
        px = _decode(np.frombuffer(file_bytes, np.uint8), mode, channel)
        self.ops.append({"op": "image", "data": self._arr(np.frombuffer(file_bytes, np.uint8)), "encoded": True, "mode": mode,
                         "channel": channel, "rule": rule, "srgb": bool(srgb)})
        return px.shape[1], px.shape[0]

    def add_mesh(self, positions, normals, uvs, indices) -> int:
        op = {"op": "mesh", "p": self._arr(np.asarray(positions, F32)), "n": self._arr(np.asarray(normals, F32)),
              "t": self._arr(np.asarray(uvs, F32)), "i": self._arr(np.asarray(indices, np.uint32))}
        self.ops.append(op)
        return sum(1 for o in self.ops if o["op"] == "mesh") - 1

    def build(self, width: int | None = None, height: int | None = None, max_depth: int | None = None, ctx=None) -> HostScene:
        """Replays the calls against a fresh HostScene and finalises it (BVH + light tables); with a
        Context the acceleration structures are built on its GPU (identical arrays)."""
        hs = HostScene()
        hs.set_integrator(max_depth or self.max_depth, self.max_iterations)
        c = self.camera
        hs.set_camera(c["eye"], c["look_at"], c["up"], c["fovy"], width or c["width"], height or c["height"])
        meshes = []
        for o in self.ops:
            k = o["op"]
            if k == "mesh":
                p, n, t, i = (self.arrays[o[x]] for x in ("p", "n", "t", "i"))
                first = hs.add_mesh(p, n, t, i)
                meshes.append((first, len(i), len(p)))
            elif k == "tex3_const":
                hs.add_constant_texture3(o["rgb"])
            elif k == "tex1_const":
                hs.add_constant_texture1(o["value"])
            elif k == "image":
                raw = self.arrays[o["data"]]
                if o.get("encoded"):
                    raw = _decode(raw, o["mode"], o["channel"])
                rule = o["rule"]
                if rule == "mul_inv255":      # image.cpp:44-49, :88-97: v * (1.0f / 255.0f)
                    tex = raw.astype(F32) * F32(1.0 / 255.0)
                elif rule == "div255":        # mesh/gltf_model.h:96-152: v / 255.0f
                    tex = raw.astype(F32) / F32(255.0)
                else:
                    tex = raw.astype(F32)
                hs.add_image(tex, srgb=o.get("srgb", False))
            elif k == "tex_image":
                hs.add_image_texture(o["image_id"], o["w"], o["h"], o["channels"])
            elif k == "material":
                w1 = F32(o["w1f"]) if "w1f" in o else int(o.get("w1i", 0))
                hs.add_material(o["type"], o.get("tex0", 0), w1, o.get("w2", 0))
            elif k == "instance":
                first, icount, vcount = meshes[o["mesh"]]
                first += o.get("first_index", 0)
                icount = o.get("index_count", icount)
                vcount = o.get("vertex_count", vcount)
                hs.add_instance(first, icount, vcount, o["material"], np.asarray(o["transform"], F32), o["blas"])
            else:
                raise ValueError(f"unknown op {k}")
        hs.finalize(ctx)
        return hs

    # ---- cooked container
    def save(self, path: str):
        meta = {"max_depth": self.max_depth, "max_iterations": self.max_iterations, "camera": self.camera, "ops": self.ops}
        np.savez_compressed(path, __meta__=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8), **self.arrays)

    @staticmethod
    def load(path: str) -> "SceneSpec":
        z = np.load(path, allow_pickle=False)
        meta = json.loads(bytes(z["__meta__"]).decode())
        s = SceneSpec(meta["max_depth"], meta["max_iterations"], meta["camera"], meta["ops"])
        s.arrays = {k: z[k] for k in z.files if k != "__meta__"}
        return s


def _decode(file_bytes: np.ndarray, mode: str, channel) -> np.ndarray:
    """Decodes image file bytes with PIL to uint8 [H,W] or [H,W,3]."""
    from PIL import Image
    img = Image.open(io.BytesIO(file_bytes.tobytes()))
    if mode:
        img = img.convert(mode)
    px = np.asarray(img, dtype=np.uint8)
    if px.ndim == 2:
        px = px[:, :, None]
    if channel is None:
        return np.ascontiguousarray(px[:, :, :3])
    ch = channel if 0 <= channel < px.shape[2] else 0
    return np.ascontiguousarray(px[:, :, ch])


# ------------------------------------------------------------------------ small fp32 helpers
def _normalize_rows(v: np.ndarray) -> np.ndarray:
    """glm::normalize per row in fp32: v * (1 / sqrt((x*x + y*y) + z*z))."""
    v = v.astype(F32)
    d = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = (F32(1.0) / np.sqrt(d, dtype=F32)).astype(F32)
    return (v * inv[:, None]).astype(F32)


def _cross_rows(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = a.astype(F32)
    b = b.astype(F32)
    return np.stack([a[:, 1] * b[:, 2] - b[:, 1] * a[:, 2], a[:, 2] * b[:, 0] - b[:, 2] * a[:, 0],
                     a[:, 0] * b[:, 1] - b[:, 0] * a[:, 1]], axis=1).astype(F32)


# ---------------------------------------------------------------------------------- OBJ
def load_obj(path: str, face_normals: bool, invert_normals: bool):
    """LoadWavefrontObjMesh (mesh/wavefront_obj.cpp:7-107): unindexed triangles; normals are
    recomputed per face when the file has none or FACE_NORMALS is set."""
    v, vn, vt, faces = [], [], [], []
    with open(path) as f:
        for line in f:
            s = line.split()
            if not s:
                continue
            if s[0] == "v":
                v.append([float(x) for x in s[1:4]])
            elif s[0] == "vn":
                vn.append([float(x) for x in s[1:4]])
            elif s[0] == "vt":
                vt.append([float(x) for x in s[1:3]])
            elif s[0] == "f":
                corners = []
                for tok in s[1:]:
                    parts = tok.split("/")
                    vi = int(parts[0])
                    ti = int(parts[1]) if len(parts) > 1 and parts[1] else 0
                    ni = int(parts[2]) if len(parts) > 2 and parts[2] else 0
                    corners.append((vi, ti, ni))
                for k in range(1, len(corners) - 1):  # fan triangulation of polygons
                    faces.append((corners[0], corners[k], corners[k + 1]))
    v = np.asarray(v, F32).reshape(-1, 3)
    vn = np.asarray(vn, F32).reshape(-1, 3)
    vt = np.asarray(vt, F32).reshape(-1, 2)

    def fix(i, n):
        return i - 1 if i > 0 else n + i

    compute = len(vn) == 0 or face_normals
    pos = np.zeros((len(faces) * 3, 3), F32)
    nrm = np.zeros((len(faces) * 3, 3), F32)
    uv = np.zeros((len(faces) * 3, 2), F32)
    for f_i, tri in enumerate(faces):
        for c, (vi, ti, ni) in enumerate(tri):
            pos[f_i * 3 + c] = v[fix(vi, len(v))]
            if ti:
                uv[f_i * 3 + c] = vt[fix(ti, len(vt))]
            if not compute and ni:
                nrm[f_i * 3 + c] = vn[fix(ni, len(vn))]
    if compute:
        p0, p1, p2 = pos[0::3], pos[1::3], pos[2::3]
        e1 = _normalize_rows(p1 - p0)
        e2 = _normalize_rows(p2 - p0)
        fn = _normalize_rows(_cross_rows(e1, e2))
        nrm = np.repeat(fn, 3, axis=0)
    else:
        nrm = _normalize_rows(nrm)
    if invert_normals:
        nrm = -nrm
    idx = np.arange(len(pos), dtype=np.uint32)
    return pos, nrm.astype(F32), uv, idx


# --------------------------------------------------------------------------------- glTF
_COMP = {5120: ("i1", 1), 5121: ("u1", 1), 5122: ("<i2", 2), 5123: ("<u2", 2), 5125: ("<u4", 4), 5126: ("<f4", 4)}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


class _Gltf:
    def __init__(self, path: str):
        self.dir = os.path.dirname(path)
        with open(path) as f:
            self.j = json.load(f)
        self.buffers = []
        for b in self.j.get("buffers", []):
            uri = b["uri"]
            if uri.startswith("data:"):
                self.buffers.append(base64.b64decode(uri.split(",", 1)[1]))
            else:
                with open(os.path.join(self.dir, uri), "rb") as f:
                    self.buffers.append(f.read())

    def accessor(self, idx: int) -> np.ndarray:
        a = self.j["accessors"][idx]
        bv = self.j["bufferViews"][a["bufferView"]]
        dt, size = _COMP[a["componentType"]]
        nc = _NCOMP[a["type"]]
        stride = bv.get("byteStride", 0) or size * nc
        start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        buf = self.buffers[bv["buffer"]]
        count = a["count"]
        out = np.zeros((count, nc), dtype=np.dtype(dt))
        raw = np.frombuffer(buf, dtype=np.uint8)
        for c in range(nc):
            offs = start + c * size + stride * np.arange(count)
            gather = raw[(offs[:, None] + np.arange(size)[None, :])].reshape(count, size)
            out[:, c] = gather.copy().view(np.dtype(dt)).reshape(count)
        return out

    def image_bytes(self, idx: int) -> bytes:
        """Encoded file bytes of image `idx` (tinygltf decodes them through stb as RGBA)."""
        im = self.j["images"][idx]
        if "uri" in im:
            uri = im["uri"]
            data = base64.b64decode(uri.split(",", 1)[1]) if uri.startswith("data:") else open(os.path.join(self.dir, uri), "rb").read()
        else:
            bv = self.j["bufferViews"][im["bufferView"]]
            data = self.buffers[bv["buffer"]][bv.get("byteOffset", 0):bv.get("byteOffset", 0) + bv["byteLength"]]
        return bytes(data)


def _load_gltf_object(spec: SceneSpec, state: dict, path: str, face_normals: bool, invert_normals: bool,
                      instance_transform: np.ndarray, material_override: int | None):
    """GLTFModel::LoadGLTF + the glTF branch of Scene::ParseObject
    (mesh/gltf_model.cpp:21-339, scene.cpp:425-515)."""
    g = _Gltf(path)
    j = g.j
    # ---- meshes -> one aggregate mesh, primitives remember their index ranges
    agg_p, agg_n, agg_t, agg_i = [], [], [], []
    nverts = 0
    nidx = 0
    mesh_prims = []
    for mesh in j.get("meshes", []):
        prims = []
        for prim in mesh["primitives"]:
            if prim.get("mode", 4) != 4:
                continue
            pos = g.accessor(prim["attributes"]["POSITION"]).astype(F32)
            vc = len(pos)
            use_face = face_normals or "NORMAL" not in prim["attributes"]
            nrm = np.zeros((vc, 3), F32)
            if not use_face:
                nrm = _normalize_rows(g.accessor(prim["attributes"]["NORMAL"]).astype(F32))
            uv = np.zeros((vc, 2), F32)
            if "TEXCOORD_0" in prim["attributes"]:
                uv = g.accessor(prim["attributes"]["TEXCOORD_0"]).astype(F32)
            idx = g.accessor(prim["indices"]).astype(np.uint32).reshape(-1) if "indices" in prim else np.arange(vc, dtype=np.uint32)
            first_index = nidx
            if not use_face:
                agg_i.append(idx + np.uint32(nverts))
                agg_p.append(pos)
                agg_n.append(nrm)
                agg_t.append(uv)
                out_vc = vc
            else:
                ntri = len(idx) // 3
                tri = idx[:ntri * 3]
                p0, p1, p2 = pos[tri[0::3]], pos[tri[1::3]], pos[tri[2::3]]
                fn = _normalize_rows(_cross_rows(p1 - p0, p2 - p0))
                agg_p.append(pos[tri])
                agg_n.append(np.repeat(fn, 3, axis=0))
                agg_t.append(uv[tri])
                agg_i.append(np.arange(nverts, nverts + ntri * 3, dtype=np.uint32))
                out_vc = len(idx)
            index_count = len(idx)
            nverts += len(agg_p[-1])
            nidx += len(agg_i[-1])
            prims.append({"first_index": first_index, "index_count": index_count, "vertex_count": out_vc,
                          "material_id": max(prim.get("material", 0), 0)})
        mesh_prims.append(prims)
    P = np.concatenate(agg_p).astype(F32)
    N = np.concatenate(agg_n).astype(F32)
    T = np.concatenate(agg_t).astype(F32)
    I = np.concatenate(agg_i).astype(np.uint32)
    if invert_normals:
        N = -N
    mesh_id = spec.add_mesh(P, N, T, I)

    # ---- materials (appended to the scene's list; images decoded per object load)
    base_material = state["n_materials"]
    albedo_map, metallic_map, rough_map = {}, {}, {}
    for mat in j.get("materials", []):
        pbr = mat.get("pbrMetallicRoughness", {})
        tex0, metallic_tex, rough_tex = 0, 0, 0  # absent textures fall back to texture 0
        if "baseColorTexture" in pbr:
            src = j["textures"][pbr["baseColorTexture"]["index"]]["source"]
            if src not in albedo_map:
                w, h = spec.add_image_file(g.image_bytes(src), "RGBA", None, "div255", True)
                spec.ops.append({"op": "tex_image", "image_id": state["n_images3"], "w": w, "h": h, "channels": 3})
                albedo_map[src] = state["n_tex3"]
                state["n_images3"] += 1
                state["n_tex3"] += 1
            tex0 = albedo_map[src]
        if "metallicRoughnessTexture" in pbr:
            src = j["textures"][pbr["metallicRoughnessTexture"]["index"]]["source"]
            if src not in metallic_map:
                for channel, table in ((2, metallic_map), (1, rough_map)):  # B = metallic, G = roughness
                    w, h = spec.add_image_file(g.image_bytes(src), "RGBA", channel, "div255")
                    spec.ops.append({"op": "tex_image", "image_id": state["n_images1"], "w": w, "h": h, "channels": 1})
                    table[src] = state["n_tex1"]
                    state["n_images1"] += 1
                    state["n_tex1"] += 1
            metallic_tex, rough_tex = metallic_map[src], rough_map[src]
        spec.ops.append({"op": "material", "type": abi.PT_MAT_METALLIC_ROUGHNESS, "tex0": tex0, "w1i": metallic_tex, "w2": rough_tex})
        state["n_materials"] += 1

    # ---- node hierarchy: world = parent * (T * R * S * matrix)
    ident = np.eye(4, dtype=F32).reshape(16)
    nodes_out = []

    def load_node(ni, parent_world):
        n = j["nodes"][ni]
        local = gltf_local_matrix(n.get("translation", [0, 0, 0]), n.get("rotation", [0, 0, 0, 1]),
                                  n.get("scale", [1, 1, 1]), n.get("matrix", ident))
        # a node without "rotation" keeps the value-initialised (all-zero) quaternion of
        # mesh/gltf_model.h:52, whose matrix is the identity, the same as (0,0,0,1) gives
        world = matrix_mul(parent_world, local)
        rec = {"world": world, "mesh": n.get("mesh", -1), "children": []}
        for ci in n.get("children", []):
            rec["children"].append(load_node(ci, world))
        return rec

    scene_idx = j.get("scene", 0)
    roots = [load_node(ni, ident) for ni in j["scenes"][scene_idx]["nodes"]]
    stack = list(roots)  # ForEachNode: push roots in order, pop from the back
    while stack:
        cur = stack.pop()
        nodes_out.append(cur)
        for ch in cur["children"]:
            stack.append(ch)
    blas_indices = {}
    for node in nodes_out:
        if node["mesh"] < 0:
            continue
        world = matrix_mul(instance_transform, node["world"])
        for pi, prim in enumerate(mesh_prims[node["mesh"]]):
            key = (node["mesh"], pi)
            if key not in blas_indices:
                blas_indices[key] = state["n_blas"]
                state["n_blas"] += 1
            material = material_override if material_override is not None else base_material + prim["material_id"]
            spec.ops.append({"op": "instance", "mesh": mesh_id, "first_index": prim["first_index"], "index_count": prim["index_count"],
                             "vertex_count": prim["vertex_count"], "material": material, "transform": [float(x) for x in world],
                             "blas": blas_indices[key]})


# ------------------------------------------------------------------ procedural primitives
def unit_cube():
    """Unit cube [-0.5,0.5]^3, 12 triangles, per-face normals, per-face [0,1]^2 uvs."""
    faces = [((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (-1, 0, 0), (0, 1, 0)), ((1, 0, 0), (0, 0, -1), (0, 1, 0)),
             ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (1, 0, 0), (0, 0, -1)), ((0, -1, 0), (1, 0, 0), (0, 0, 1))]
    pos, nrm, uv, idx = [], [], [], []
    for n, u, v in faces:
        n, u, v = np.asarray(n, F32), np.asarray(u, F32), np.asarray(v, F32)
        base = len(pos)
        for su, sv in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            pos.append(F32(0.5) * (n + F32(su) * u + F32(sv) * v))
            nrm.append(n)
            uv.append([(su + 1) / 2, (sv + 1) / 2])
        idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return np.asarray(pos, F32), np.asarray(nrm, F32), np.asarray(uv, F32), np.asarray(idx, np.uint32)


def uv_sphere(stacks: int = 32, slices: int = 64, radius: float = 0.5):
    """Radius-0.5 UV sphere with smooth normals."""
    pos, nrm, uv = [], [], []
    for i in range(stacks + 1):
        theta = np.pi * i / stacks
        for k in range(slices + 1):
            phi = 2.0 * np.pi * k / slices
            n = np.array([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], np.float64)
            nn = n.astype(F32)
            pos.append((radius * n).astype(F32))
            nrm.append(nn)
            uv.append([k / slices, i / stacks])
    idx = []
    row = slices + 1
    for i in range(stacks):
        for k in range(slices):
            a, b, c, d = i * row + k, i * row + k + 1, (i + 1) * row + k, (i + 1) * row + k + 1
            if i != 0:
                idx += [a, b, c]
            if i != stacks - 1:
                idx += [b, d, c]
    return np.asarray(pos, F32), _normalize_rows(np.asarray(nrm, F32)), np.asarray(uv, F32), np.asarray(idx, np.uint32)


# --------------------------------------------------------------------------- scene JSON
def _vec(d, key, default):
    v = d.get(key, default)
    return [float(x) for x in v] if isinstance(v, (list, tuple)) and len(v) == len(default) else list(default)


def _parse_texture3(spec, state, tex, scene_dir, is_albedo=False) -> int:
    """ParseTexture<glm::vec3> (scene.h:76-179)."""
    tid = state["n_tex3"]
    if isinstance(tex, dict) and "PATH" in tex and isinstance(tex["PATH"], str):
        p = tex["PATH"] if os.path.exists(tex["PATH"]) else os.path.join(scene_dir, tex["PATH"])
        if os.path.exists(p):
            w, h = spec.add_image_file(open(p, "rb").read(), "RGB", None, "mul_inv255", bool(is_albedo))
            spec.ops.append({"op": "tex_image", "image_id": state["n_images3"], "w": w, "h": h, "channels": 3})
            state["n_images3"] += 1
            state["n_tex3"] += 1
            return tid
    rgb = None
    if isinstance(tex, dict) and "RGB" in tex and isinstance(tex["RGB"], list) and len(tex["RGB"]) == 3:
        rgb = tex["RGB"]
    elif isinstance(tex, list) and len(tex) == 3:  # legacy flat colour
        rgb = tex
    if rgb is None:
        return 0
    spec.ops.append({"op": "tex3_const", "rgb": [float(x) for x in rgb]})
    state["n_tex3"] += 1
    return tid


def _parse_texture1(spec, state, tex, scene_dir, channel=0) -> int:
    """ParseTexture<float> (scene.h:76-179)."""
    tid = state["n_tex1"]
    if isinstance(tex, dict) and "PATH" in tex and isinstance(tex["PATH"], str):
        p = tex["PATH"] if os.path.exists(tex["PATH"]) else os.path.join(scene_dir, tex["PATH"])
        if os.path.exists(p):
            w, h = spec.add_image_file(open(p, "rb").read(), "", channel, "mul_inv255")
            spec.ops.append({"op": "tex_image", "image_id": state["n_images1"], "w": w, "h": h, "channels": 1})
            state["n_images1"] += 1
            state["n_tex1"] += 1
            return tid
    value = None
    if isinstance(tex, dict) and "VALUE" in tex and isinstance(tex["VALUE"], (int, float)):
        value = tex["VALUE"]
    elif isinstance(tex, (int, float)):  # legacy bare number
        value = tex
    if value is None:
        return 0
    spec.ops.append({"op": "tex1_const", "value": float(value)})
    state["n_tex1"] += 1
    return tid


def _parse_material(spec, state, m, scene_dir, legacy: bool):
    """Scene::ParseMaterial (scene.cpp:262-335) plus the legacy spellings."""
    t = str(m.get("TYPE", "")).lower()
    op = None
    if t == "emitting":
        if "RADIANCE" in m:
            tex = _parse_texture3(spec, state, m["RADIANCE"], scene_dir)
            emit = float(m["RADIANCE"].get("EMITTANCE", 1.0))
        elif legacy and ("COLOR" in m or "RGB" in m):
            tex = _parse_texture3(spec, state, m.get("COLOR", m.get("RGB")), scene_dir)
            emit = float(m.get("EMITTANCE", 1.0))
        else:
            tex, emit = 0, 1.0
        op = {"op": "material", "type": abi.PT_MAT_EMISSIVE, "tex0": tex, "w1f": emit}
    elif t in ("diffuse", "lambertian"):
        src = m.get("ALBEDO", m.get("RGB") if legacy else None)
        tex = _parse_texture3(spec, state, src, scene_dir, True) if src is not None else 0
        op = {"op": "material", "type": abi.PT_MAT_LAMBERTIAN, "tex0": tex}
    elif t == "mirror" or (legacy and t == "specular"):
        src = m.get("ALBEDO", m.get("RGB") if legacy else None)
        tex = _parse_texture3(spec, state, src, scene_dir, True) if src is not None else 0
        op = {"op": "material", "type": abi.PT_MAT_MIRROR, "tex0": tex}
    elif t == "dielectric" or (legacy and t == "roughdielectric"):
        ior = m.get("INDEX_OF_REFRACTION", m.get("IOR"))
        src = m.get("ALBEDO", m.get("RGB") if legacy else None)
        tex = _parse_texture3(spec, state, src, scene_dir, True) if (src is not None and ior is not None) else 0
        op = {"op": "material", "type": abi.PT_MAT_DIELECTRIC, "tex0": tex, "w1f": float(ior if ior is not None else 1.0)}
    elif t == "metallic_roughness" or (legacy and t == "metallicroughness"):
        src = m.get("ALBEDO")
        tex = _parse_texture3(spec, state, src, scene_dir, True) if src is not None else 0
        rough = _parse_texture1(spec, state, m["ROUGHNESS"], scene_dir) if "ROUGHNESS" in m else 0
        metal = _parse_texture1(spec, state, m["METALLIC"], scene_dir) if "METALLIC" in m else 0
        if metal == 0 and rough == 0 and "ORM" in m:
            rough = _parse_texture1(spec, state, m["ORM"], scene_dir, 1)
            metal = _parse_texture1(spec, state, m["ORM"], scene_dir, 2)
        op = {"op": "material", "type": abi.PT_MAT_METALLIC_ROUGHNESS, "tex0": tex, "w1i": metal, "w2": rough}
    else:
        # the reference pushes an uninitialised Material here; define it as the default Lambertian
        op = {"op": "material", "type": abi.PT_MAT_LAMBERTIAN, "tex0": 0}
    spec.ops.append(op)
    state["n_materials"] += 1


def load_scene_json(path: str) -> SceneSpec:
    """Scene::LoadScene (scene.cpp:649-687) for the current schema, or the legacy adapter."""
    with open(path) as f:
        data = json.load(f)
    scene_dir = os.path.dirname(os.path.abspath(path))
    legacy = "camera" not in data and "Camera" in data
    spec = SceneSpec()
    # defaults: texture 0 of each kind and material 0 exist before anything is parsed
    state = {"n_tex1": 1, "n_tex3": 1, "n_images1": 0, "n_images3": 0, "n_materials": 1, "n_blas": 0}
    if legacy:
        cam = data["Camera"]
        spec.max_depth = int(cam.get("DEPTH", 4))
        spec.max_iterations = int(cam.get("ITERATIONS", 1000))
        res = cam.get("RES", [800, 800])
        spec.camera = {"eye": _vec(cam, "EYE", [0, 0, 0]), "look_at": _vec(cam, "LOOKAT", [0, 0, -1]), "up": _vec(cam, "UP", [0, 1, 0]),
                       "fovy": float(cam.get("FOVY", 45.0)), "width": int(res[0]), "height": int(res[1])}
        materials = data.get("Materials", {})
        objects = data.get("Objects", [])
    else:
        integ = data.get("integrator", {})
        spec.max_depth = int(integ.get("DEPTH", 4))
        spec.max_iterations = int(integ.get("ITERATIONS", 1000))
        cam = data["camera"]
        film = cam["film"]
        spec.camera = {"eye": _vec(cam, "EYE", [0, 0, 0]), "look_at": _vec(cam, "LOOKAT", [0, 0, -1]), "up": _vec(cam, "UP", [0, 1, 0]),
                       "fovy": float(cam.get("FOVY", 45.0)), "width": int(film["WIDTH"]), "height": int(film["HEIGHT"])}
        materials = data.get("materials", {})
        objects = data.get("objects", [])
    material_ids = {}
    for key in sorted(materials.keys()):
        material_ids[key] = state["n_materials"]
        _parse_material(spec, state, materials[key], scene_dir, legacy)

    named_instances = {}

    def parse_object(obj) -> list:
        t = str(obj.get("TYPE", "")).lower()
        mat_name = obj.get("MATERIAL")
        override = material_ids.get(mat_name) if mat_name in material_ids else None
        xf = transform_matrix(_vec(obj, "TRANS", [0, 0, 0]), _vec(obj, "ROTAT", [0, 0, 0]), _vec(obj, "SCALE", [1, 1, 1]))
        before = len(spec.ops)
        if t == "obj":
            p = obj["PATH"] if os.path.isabs(obj["PATH"]) else os.path.join(scene_dir, obj["PATH"])
            pos, nrm, uv, idx = load_obj(p, bool(obj.get("FACE_NORMALS", False)), bool(obj.get("INVERT_NORMALS", False)))
            mesh = spec.add_mesh(pos, nrm, uv, idx)
            spec.ops.append({"op": "instance", "mesh": mesh, "material": override if override is not None else 0,
                             "transform": [float(x) for x in xf], "blas": state["n_blas"]})
            state["n_blas"] += 1
        elif t == "gltf":
            p = obj["PATH"] if os.path.isabs(obj["PATH"]) else os.path.join(scene_dir, obj["PATH"])
            _load_gltf_object(spec, state, p, bool(obj.get("FACE_NORMALS", False)), bool(obj.get("INVERT_NORMALS", False)), xf, override)
        elif legacy and t in ("cube", "sphere"):
            pos, nrm, uv, idx = unit_cube() if t == "cube" else uv_sphere()
            mesh = spec.add_mesh(pos, nrm, uv, idx)
            spec.ops.append({"op": "instance", "mesh": mesh, "material": override if override is not None else 0,
                             "transform": [float(x) for x in xf], "blas": state["n_blas"]})
            state["n_blas"] += 1
        return [o for o in spec.ops[before:] if o["op"] == "instance"]

    # pass 1 (scene.cpp:526-547)
    for obj in objects:
        t = str(obj.get("TYPE", "")).lower()
        parse_object(obj)
        if t == "instance" and "NAME" in obj and "INSTANCE" in obj:
            before = len(spec.ops)
            insts = parse_object(obj["INSTANCE"])
            # the base of a named instance is not itself placed in the scene
            spec.ops[before:] = [o for o in spec.ops[before:] if o["op"] != "instance"]
            named_instances[obj["NAME"]] = insts
    # pass 2 (scene.cpp:549-577)
    for obj in objects:
        if str(obj.get("TYPE", "")).lower() == "instance" and "SOURCE" in obj:
            xf = transform_matrix(_vec(obj, "TRANS", [0, 0, 0]), _vec(obj, "ROTAT", [0, 0, 0]), _vec(obj, "SCALE", [1, 1, 1]))
            for base in named_instances.get(obj["SOURCE"], []):
                o = dict(base)
                o["transform"] = [float(x) for x in matrix_mul(xf, np.asarray(base["transform"], F32))]
                spec.ops.append(o)
    return spec
