"""Minimal Wavefront OBJ ingest for running the path without Godot.

Produces a scenes.Mesh the way Godot's importer hands the reference an ArrayMesh: one surface per `usemtl`
group, de-indexed (position, normal, uv) triples, polygons fan-triangulated, and the winding flipped to Godot's
clockwise front faces (main.glsl:254-255 derives `front` from it).  Missing normals are replaced by the face
normal, missing uvs by (0, 0).  Plumbing only: nothing here is on the timed path.
"""
from __future__ import annotations

import numpy as np

from .scenes import Mesh, Surface


def load_obj(text: str) -> Mesh:
    pos, nrm, uvs = [], [], []
    groups = {}          # material name -> dict(key -> index), vertex lists, indices
    order = []
    cur = None

    def group(name):
        nonlocal cur
        if name not in groups:
            groups[name] = dict(lut={}, v=[], n=[], t=[], idx=[])
            order.append(name)
        cur = groups[name]

    group("")
    for line in text.splitlines():
        p = line.split()
        if not p or p[0].startswith("#"):
            continue
        if p[0] == "v":
            pos.append([float(x) for x in p[1:4]])
        elif p[0] == "vn":
            nrm.append([float(x) for x in p[1:4]])
        elif p[0] == "vt":
            uvs.append([float(p[1]), float(p[2]) if len(p) > 2 else 0.0])
        elif p[0] == "usemtl":
            group(p[1] if len(p) > 1 else "")
        elif p[0] == "f":
            corners = []
            for tok in p[1:]:
                f = (tok.split("/") + ["", ""])[:3]
                vi = int(f[0]); ti = int(f[1]) if f[1] else 0; ni = int(f[2]) if f[2] else 0
                vi = vi - 1 if vi > 0 else len(pos) + vi
                ti = (ti - 1 if ti > 0 else len(uvs) + ti) if f[1] else -1
                ni = (ni - 1 if ni > 0 else len(nrm) + ni) if f[2] else -1
                corners.append((vi, ti, ni))
            fn = None
            if any(c[2] < 0 for c in corners):
                a, b, c = (np.asarray(pos[corners[k][0]]) for k in range(3))
                fn = np.cross(b - a, c - a)
                fn = fn / max(np.linalg.norm(fn), 1e-30)
            ids = []
            for k, (vi, ti, ni) in enumerate(corners):
                key = (vi, ti, ni if ni >= 0 else ("f", len(cur["idx"]), k))
                if key not in cur["lut"]:
                    cur["lut"][key] = len(cur["v"])
                    cur["v"].append(pos[vi])
                    cur["n"].append(nrm[ni] if ni >= 0 else fn)
                    cur["t"].append(uvs[ti] if ti >= 0 else [0.0, 0.0])
                ids.append(cur["lut"][key])
            for k in range(1, len(ids) - 1):          # fan; (0, k+1, k) = clockwise for Godot
                cur["idx"] += [ids[0], ids[k + 1], ids[k]]
    surfaces = [Surface(np.asarray(g["v"], np.float32), np.asarray(g["n"], np.float32), np.asarray(g["t"], np.float32),
                        np.asarray(g["idx"], np.int32)) for g in (groups[n] for n in order) if g["idx"]]
    return Mesh(surfaces)


def pack_texture_array(images, resolution: int) -> np.ndarray:
    """The texture-array packing of GeometryGroup3D::build (geometry_group3d.cpp:294-300): every albedo image becomes
    one RGBA8 layer of `resolution` x `resolution` (the reference: clear_mipmaps, decompress, Image::resize).  Returns
    (layers, resolution, resolution, 4) uint8 for Scene.textures / jpt_scene_set_textures; an empty list gives the one
    blank layer the reference creates (:301-303).

    The resampling is a plain bilinear filter over pixel centres.  Godot's Image::resize is engine code (absent here),
    so a layer made from an image of another size is NOT claimed to match the reference's texel for texel; images
    that already have the array's resolution pass through unchanged."""
    if not images:
        return np.zeros((1, resolution, resolution, 4), dtype=np.uint8)
    layers = []
    for img in images:
        a = np.asarray(img)
        if a.ndim == 2:
            a = a[..., None]
        if a.shape[-1] == 1:
            a = np.repeat(a, 3, axis=-1)
        if a.shape[-1] == 3:
            a = np.concatenate([a, np.full(a.shape[:2] + (1,), 255, dtype=a.dtype)], axis=-1)
        a = a[..., :4].astype(np.uint8)
        h, w = a.shape[:2]
        if (h, w) != (resolution, resolution):
            ys = (np.arange(resolution) + 0.5) * h / resolution - 0.5
            xs = (np.arange(resolution) + 0.5) * w / resolution - 0.5
            y0 = np.clip(np.floor(ys).astype(np.int64), 0, h - 1); y1 = np.clip(y0 + 1, 0, h - 1)
            x0 = np.clip(np.floor(xs).astype(np.int64), 0, w - 1); x1 = np.clip(x0 + 1, 0, w - 1)
            fy = np.clip(ys - np.floor(ys), 0.0, 1.0)[:, None, None]
            fx = np.clip(xs - np.floor(xs), 0.0, 1.0)[None, :, None]
            f = a.astype(np.float64)
            top = f[y0][:, x0] * (1.0 - fx) + f[y0][:, x1] * fx
            bot = f[y1][:, x0] * (1.0 - fx) + f[y1][:, x1] * fx
            a = np.clip(np.floor(top * (1.0 - fy) + bot * fy + 0.5), 0, 255).astype(np.uint8)
        layers.append(a)
    return np.stack(layers)


def load_mtl(text: str):
    """Wavefront MTL -> (dict name -> GpuMaterial record (wire.MATERIAL), list of albedo map file names), with the
    conventions of include/jpt_host.hpp::load_mtl (the C++ product code; this is its test mirror): Kd -> albedo; Ke ->
    emission colour, energy multiplier 1 or the largest component when that exceeds 1; Pr -> roughness, else
    sqrt(2 / (Ns + 2)) from the Phong exponent; Pm -> metallic; map_Kd -> albedo texture index in order of first use."""
    from .scenes import material
    out, maps = {}, []
    cur = None
    for line in text.splitlines():
        p = line.split()
        if not p or p[0].startswith("#"):
            continue
        if p[0] == "newmtl":
            cur = dict(albedo=(1.0, 1.0, 1.0), emission=(0.0, 0.0, 0.0), energy=1.0, metallic=0.0, roughness=1.0, texture=-1, has_pr=False)
            out[p[1] if len(p) > 1 else ""] = cur
        elif cur is None:
            continue
        elif p[0] == "Kd":
            cur["albedo"] = tuple(np.float32(x) for x in p[1:4])
        elif p[0] == "Ke":
            r, g, b = (np.float32(x) for x in p[1:4])
            m = max(r, g, b)
            if m > 1.0:
                cur["emission"], cur["energy"] = (r / m, g / m, b / m), m
            else:
                cur["emission"], cur["energy"] = (r, g, b), np.float32(1.0)
        elif p[0] == "Pr":
            cur["roughness"], cur["has_pr"] = np.float32(p[1]), True
        elif p[0] == "Ns" and not cur["has_pr"]:
            cur["roughness"] = np.sqrt(np.float32(2.0) / (max(np.float32(p[1]), np.float32(0.0)) + np.float32(2.0)))
        elif p[0] == "Pm":
            cur["metallic"] = np.float32(p[1])
        elif p[0] == "map_Kd":
            if p[1] not in maps:
                maps.append(p[1])
            cur["texture"] = maps.index(p[1])
    recs = {k: material(albedo=v["albedo"], emission=v["emission"], energy=float(v["energy"]), metallic=float(v["metallic"]),
                        roughness=float(v["roughness"]), texture=v["texture"]) for k, v in out.items()}
    return recs, maps
