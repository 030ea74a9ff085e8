"""Seeded procedural scenes.

The reference's example asset (``example/scene.glb``) is not available, so the
benchmark and parity configurations run on deterministic stand-ins with the
same feature mix as ``example/render.png``: a textured checker floor, a metal
sphere, a blue transmissive mesh with volume absorption, emissive geometry, an
alpha-blended textured panel, alpha-masked cards and a normal-mapped object.
Everything here is plain input data for the renderer (the ``Model`` of
``src/model_loader.h:59-65``); nothing in this module is on the hot path.
"""
from __future__ import annotations

import numpy as np

from .model import Camera, ModelBuilder, SceneModel, rotate, scale, translate


# ---------------------------------------------------------------------------------------------
# geometry helpers
# ---------------------------------------------------------------------------------------------
def _grid_indices(nu: int, nv: int) -> np.ndarray:
    """Two triangles per cell of an (nv+1) x (nu+1) vertex grid, counter-clockwise for +normal = du x dv... flipped below."""
    j, i = np.meshgrid(np.arange(nv, dtype=np.uint32), np.arange(nu, dtype=np.uint32), indexing="ij")
    a = j * (nu + 1) + i
    b = a + 1
    c = a + (nu + 1)
    d = c + 1
    tris = np.stack([a, c, b, b, c, d], axis=-1)
    return tris.reshape(-1, 3)


def _vertex_normals(pos: np.ndarray, tris: np.ndarray) -> np.ndarray:
    p = pos.astype(np.float64)
    fn = np.cross(p[tris[:, 1]] - p[tris[:, 0]], p[tris[:, 2]] - p[tris[:, 0]])
    n = np.zeros_like(p)
    for k in range(3):
        for c in range(3):
            n[:, c] += np.bincount(tris[:, k], weights=fn[:, c], minlength=len(p))
    return n


def _normalize(v: np.ndarray) -> np.ndarray:
    l = np.linalg.norm(v, axis=-1, keepdims=True)
    return v / np.maximum(l, 1e-30)


def plane(nx: int, nz: int, size_x: float, size_z: float, uv_scale: float = 1.0):
    """XZ plane at y = 0 facing +Y, tangent +X."""
    u, v = np.meshgrid(np.linspace(0, 1, nx + 1), np.linspace(0, 1, nz + 1), indexing="xy")
    pos = np.stack([(u - 0.5) * size_x, np.zeros_like(u), (v - 0.5) * size_z], -1).reshape(-1, 3)
    nrm = np.tile(np.array([0.0, 1.0, 0.0]), (len(pos), 1))
    uv = np.stack([u * uv_scale, v * uv_scale], -1).reshape(-1, 2)
    tan = np.tile(np.array([1.0, 0.0, 0.0, 1.0]), (len(pos), 1))
    # +Y normal: (a, c, b) with c = next row (+z) -> cross(+z, +x) = +y
    return pos, nrm, _grid_indices(nx, nz), uv, tan


def quad(width: float, height: float):
    """XY quad facing +Z, uv (0,0) at the top-left like an image."""
    pos = np.array([[-width / 2, height / 2, 0], [width / 2, height / 2, 0], [-width / 2, -height / 2, 0],
                    [width / 2, -height / 2, 0]], np.float64)
    nrm = np.tile(np.array([0.0, 0.0, 1.0]), (4, 1))
    uv = np.array([[0, 0], [1, 0], [0, 1], [1, 1]], np.float64)
    tan = np.tile(np.array([1.0, 0.0, 0.0, 1.0]), (4, 1))
    idx = np.array([[0, 2, 1], [1, 2, 3]], np.uint32)
    return pos, nrm, idx, uv, tan


def _sphere_grid(n_lat: int, n_lon: int):
    v, u = np.meshgrid(np.linspace(0, 1, n_lat + 1), np.linspace(0, 1, n_lon + 1), indexing="ij")
    theta = v * np.pi          # 0 at +Y pole
    phi = u * 2 * np.pi
    d = np.stack([np.sin(theta) * np.cos(phi), np.cos(theta), np.sin(theta) * np.sin(phi)], -1)
    return d, u, v, theta, phi


def _sphere_indices(n_lat: int, n_lon: int) -> np.ndarray:
    tris = _grid_indices(n_lon, n_lat).reshape(n_lat, n_lon, 2, 3)
    # drop the degenerate triangle of each pole cell; orientation: outward normals
    keep = np.ones((n_lat, n_lon, 2), bool)
    keep[0, :, 0] = False
    keep[-1, :, 1] = False
    t = tris[keep]
    return t[:, [0, 2, 1]]


def uv_sphere(n_lat: int, n_lon: int, radius: float = 1.0):
    d, u, v, theta, phi = _sphere_grid(n_lat, n_lon)
    pos = (d * radius).reshape(-1, 3)
    nrm = d.reshape(-1, 3)
    uv = np.stack([u, v], -1).reshape(-1, 2)
    t = np.stack([-np.sin(phi), np.zeros_like(phi), np.cos(phi)], -1).reshape(-1, 3)
    tan = np.concatenate([t, -np.ones((len(t), 1))], 1)  # cross(n, t) points to -v (up): w = -1 maps it to +v
    return pos, nrm, _sphere_indices(n_lat, n_lon), uv, tan


def blob(n_lat: int, n_lon: int, radius: float = 1.0, amplitude: float = 0.18, seed: int = 7, octaves: int = 5):
    """Sphere displaced by a seeded sum of 3-D sinusoids: a closed, bumpy mesh standing in for the dragon."""
    rng = np.random.default_rng(seed)
    d, u, v, theta, phi = _sphere_grid(n_lat, n_lon)
    f = np.zeros(d.shape[:2])
    for o in range(octaves):
        w = rng.normal(size=3) * (1.5 * 1.9 ** o)
        f += np.sin(d @ w + rng.uniform(0, 2 * np.pi)) * (0.6 ** o)
    f /= np.abs(f).max()
    pos = (d * (radius * (1.0 + amplitude * f))[..., None])
    pos[:, -1] = pos[:, 0]  # close the seam exactly
    flat = pos.reshape(-1, 3)
    tris = _sphere_indices(n_lat, n_lon)
    n = _vertex_normals(flat, tris).reshape(n_lat + 1, n_lon + 1, 3)
    seam = n[:, 0] + n[:, -1]
    n[:, 0] = seam
    n[:, -1] = seam
    n[0] = n[0].sum(0)
    n[-1] = n[-1].sum(0)
    nrm = _normalize(n).reshape(-1, 3)
    uv = np.stack([u, v], -1).reshape(-1, 2)
    t = np.stack([-np.sin(phi), np.zeros_like(phi), np.cos(phi)], -1).reshape(-1, 3)
    tan = np.concatenate([t, -np.ones((len(t), 1))], 1)
    return flat, nrm, tris, uv, tan


def torus(n_major: int, n_minor: int, R: float = 1.0, r: float = 0.35):
    v, u = np.meshgrid(np.linspace(0, 1, n_minor + 1), np.linspace(0, 1, n_major + 1), indexing="ij")
    a, b = u * 2 * np.pi, v * 2 * np.pi
    ring = np.stack([np.cos(a), np.zeros_like(a), np.sin(a)], -1)
    nrm = ring * np.cos(b)[..., None] + np.array([0, 1.0, 0]) * np.sin(b)[..., None]
    pos = ring * R + nrm * r
    t = np.stack([-np.sin(a), np.zeros_like(a), np.cos(a)], -1)
    tan = np.concatenate([t.reshape(-1, 3), np.ones((t.size // 3, 1))], 1)
    tris = _grid_indices(n_major, n_minor)
    return pos.reshape(-1, 3), nrm.reshape(-1, 3), tris, np.stack([u * 4, v], -1).reshape(-1, 2), tan


def box(sx: float, sy: float, sz: float):
    pos, nrm, uv, tan, idx = [], [], [], [], []
    faces = [((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (-1, 0, 0), (0, 1, 0)), ((1, 0, 0), (0, 0, -1), (0, 1, 0)),
             ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (1, 0, 0), (0, 0, -1)), ((0, -1, 0), (1, 0, 0), (0, 0, 1))]
    h = np.array([sx, sy, sz]) / 2
    for k, (n, t, b) in enumerate(faces):
        n, t, b = map(np.array, (n, t, b))
        for (cu, cv) in ((-1, 1), (1, 1), (-1, -1), (1, -1)):
            pos.append((n + cu * t + cv * b) * h)
            nrm.append(n)
            uv.append(((cu + 1) / 2, (1 - cv) / 2))
            tan.append((*t, 1.0))
        idx += [[4 * k, 4 * k + 2, 4 * k + 1], [4 * k + 1, 4 * k + 2, 4 * k + 3]]
    return (np.array(pos, float), np.array(nrm, float), np.array(idx, np.uint32), np.array(uv, float),
            np.array(tan, float))


# ---------------------------------------------------------------------------------------------
# textures
# ---------------------------------------------------------------------------------------------
def checker_texture(n: int = 256, cells: int = 8, a=(200, 200, 200), b=(60, 70, 90), seed: int = 1) -> np.ndarray:
    y, x = np.mgrid[0:n, 0:n]
    mask = ((x * cells // n) + (y * cells // n)) % 2
    rng = np.random.default_rng(seed)
    img = np.where(mask[..., None] == 0, np.array(a), np.array(b)).astype(np.float64)
    img += rng.normal(0, 6, img.shape)
    out = np.empty((n, n, 4), np.uint8)
    out[..., :3] = np.clip(img, 0, 255)
    out[..., 3] = 255
    return out


def orm_texture(n: int = 128, seed: int = 2) -> np.ndarray:
    """Occlusion/roughness/metallic: stripes of varying roughness, metallic 0."""
    y, x = np.mgrid[0:n, 0:n]
    out = np.zeros((n, n, 4), np.uint8)
    out[..., 0] = 255
    out[..., 1] = (96 + 120 * (0.5 + 0.5 * np.sin(x * 2 * np.pi * 3 / n))).astype(np.uint8)
    out[..., 2] = 0
    out[..., 3] = 255
    return out


def bump_normal_map(n: int = 128, freq: int = 6, strength: float = 0.6) -> np.ndarray:
    y, x = np.mgrid[0:n, 0:n] / n
    hx = strength * np.cos(x * 2 * np.pi * freq) * np.sin(y * 2 * np.pi * freq)
    hy = strength * np.sin(x * 2 * np.pi * freq) * np.cos(y * 2 * np.pi * freq)
    nv = _normalize(np.stack([-hx, -hy, np.ones_like(hx)], -1))
    out = np.empty((n, n, 4), np.uint8)
    out[..., :3] = np.clip((nv * 0.5 + 0.5) * 255 + 0.5, 0, 255)
    out[..., 3] = 255
    return out


def alpha_gradient_texture(n: int = 128) -> np.ndarray:
    y, x = np.mgrid[0:n, 0:n] / (n - 1)
    out = np.empty((n, n, 4), np.uint8)
    out[..., 0] = (255 * x).astype(np.uint8)
    out[..., 1] = (255 * (1 - x) * y).astype(np.uint8)
    out[..., 2] = (255 * (1 - y)).astype(np.uint8)
    out[..., 3] = (255 * (0.15 + 0.7 * (0.5 + 0.5 * np.sin(6 * np.pi * x) * np.cos(4 * np.pi * y)))).astype(np.uint8)
    return out


def leaf_mask_texture(n: int = 128, seed: int = 3) -> np.ndarray:
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:n, 0:n] / (n - 1) - 0.5
    out = np.zeros((n, n, 4), np.uint8)
    out[..., 0] = 40
    out[..., 1] = 150
    out[..., 2] = 50
    a = np.zeros((n, n))
    for _ in range(9):
        cx, cy, r = rng.uniform(-0.35, 0.35), rng.uniform(-0.35, 0.35), rng.uniform(0.08, 0.18)
        a = np.maximum(a, ((x - cx) ** 2 + (y - cy) ** 2 < r * r) * 1.0)
    out[..., 3] = (a * 255).astype(np.uint8)
    return out


def eyes_emissive_texture(n: int = 128) -> np.ndarray:
    y, x = np.mgrid[0:n, 0:n] / (n - 1)
    out = np.zeros((n, n, 4), np.uint8)
    for cx in (0.20, 0.30):
        m = ((x - cx) ** 2 * 4 + (y - 0.42) ** 2) < 0.0009
        out[m, 0] = 255
        out[m, 1] = 120
        out[m, 2] = 30
    out[..., 3] = 255
    return out


def synthetic_env(width: int = 64, height: int = 32, seed: int = 5) -> np.ndarray:
    """Small smooth HDR sky used when the farmland HDR file is not wanted: RGBA float32 rows."""
    v, u = np.mgrid[0:height, 0:width]
    el = (0.5 - (v + 0.5) / height) * np.pi
    az = (u + 0.5) / width * 2 * np.pi
    sky = np.stack([0.35 + 0.3 * np.sin(el), 0.45 + 0.35 * np.sin(el), 0.6 + 0.5 * np.sin(el)], -1)
    sky = np.clip(sky, 0.02, None)
    sun = np.exp(-((az - 1.1) ** 2 + (el - 0.7) ** 2) * 30.0)[..., None] * np.array([30.0, 26.0, 20.0])
    ground = (el < 0)[..., None] * np.array([0.12, 0.10, 0.08])
    img = np.where((el < 0)[..., None], ground, sky + sun)
    out = np.ones((height, width, 4), np.float32)
    out[..., :3] = img
    return out


# ---------------------------------------------------------------------------------------------
# scenes
# ---------------------------------------------------------------------------------------------
def readme_standin(detail: float = 1.0, seed: int = 7) -> SceneModel:
    """Stand-in for ``example/scene.glb`` (see module docstring). ``detail`` scales tessellation:
    1.0 gives about one million triangles, dominated by the transmissive blob."""
    b = ModelBuilder("readme_standin")
    floor_tex = b.add_image(checker_texture(256, 8))
    floor_orm = b.add_image(orm_texture(128))
    nmap = b.add_image(bump_normal_map(128))
    panel_tex = b.add_image(alpha_gradient_texture(128))
    leaf_tex = b.add_image(leaf_mask_texture(128))
    eyes_tex = b.add_image(eyes_emissive_texture(128))

    m_floor = b.add_material(base_color=(1, 1, 1, 1), metallic=0.0, roughness=1.0, base_color_texture=floor_tex,
                             metallic_roughness_texture=floor_orm)
    m_metal = b.add_material(base_color=(0.95, 0.8, 0.45, 1), metallic=1.0, roughness=0.18)
    m_glass = b.add_material(base_color=(0.85, 0.92, 1.0, 1), metallic=0.0, roughness=0.06, ior=1.45, transmission=1.0,
                             attenuation_color=(0.15, 0.35, 0.9), attenuation_distance=0.6)
    m_head = b.add_material(base_color=(0.75, 0.72, 0.68, 1), metallic=0.0, roughness=0.55, emissive=(1, 1, 1),
                            emissive_texture=eyes_tex, emissive_strength=25.0)
    m_panel = b.add_material(base_color=(1, 1, 1, 1), metallic=0.0, roughness=0.4, base_color_texture=panel_tex,
                             alpha_mode="BLEND", double_sided=True)
    m_leaf = b.add_material(base_color=(1, 1, 1, 1), metallic=0.0, roughness=0.8, base_color_texture=leaf_tex,
                            alpha_mode="MASK", alpha_cutoff=0.5, double_sided=True)
    m_bumpy = b.add_material(base_color=(0.8, 0.25, 0.2, 1), metallic=0.0, roughness=0.35, normal_texture=nmap)
    m_lamp = b.add_material(base_color=(1, 1, 1, 1), metallic=0.0, roughness=1.0, emissive=(1.0, 0.85, 0.6),
                            emissive_strength=12.0)

    d = max(detail, 0.02)
    nf = max(2, int(64 * d))
    floor = b.add_mesh(*_args(plane(nf, nf, 14.0, 14.0, uv_scale=4.0)), material=m_floor)
    ns = max(8, int(160 * d))
    sphere = b.add_mesh(*_args(uv_sphere(ns, 2 * ns, 1.0)), material=m_metal)
    nb = max(12, int(700 * d))
    dragon = b.add_mesh(*_args(blob(nb, nb, 1.0, 0.22, seed=seed)), material=m_glass)
    nh = max(8, int(120 * d))
    head = b.add_mesh(*_args(blob(nh, 2 * nh, 1.0, 0.08, seed=seed + 1, octaves=3)), material=m_head)
    panel = b.add_mesh(*_args(quad(2.4, 1.6)), material=m_panel)
    leaf = b.add_mesh(*_args(quad(1.0, 1.0)), material=m_leaf)
    nt = max(8, int(96 * d))
    tor = b.add_mesh(*_args(torus(2 * nt, nt, 0.8, 0.28)), material=m_bumpy)
    lamp = b.add_mesh(*_args(uv_sphere(max(6, int(24 * d)), max(12, int(48 * d)), 1.0)), material=m_lamp)

    b.add_object(floor, translate(0, 0, 0))
    b.add_object(sphere, translate(-2.3, 0.9, -0.6) @ scale(0.9))
    b.add_object(dragon, translate(0.2, 1.25, 0.2) @ rotate((0, 1, 0), 25) @ scale(1.2, 1.0, 1.0))
    b.add_object(head, translate(2.6, 0.8, -0.9) @ rotate((0, 1, 0), -35) @ scale(0.8, 0.95, 0.8))
    b.add_object(head, translate(-0.9, 0.45, 2.0) @ rotate((0, 1, 0), 60) @ scale(0.45))
    b.add_object(panel, translate(1.9, 1.0, 1.7) @ rotate((0, 1, 0), -30))
    for k in range(3):
        b.add_object(leaf, translate(-2.6 + 0.5 * k, 0.55, 1.4 + 0.35 * k) @ rotate((0, 1, 0), 20 + 35 * k))
    b.add_object(tor, translate(-0.4, 0.3, -2.6) @ rotate((1, 0, 0), 0))
    b.add_object(lamp, translate(3.4, 3.2, 2.4) @ scale(0.35))

    pos = np.array([0.4, 2.6, 7.2])
    target = np.array([0.1, 0.9, 0.0])
    b.add_camera(Camera(position=pos, direction=_normalize(target - pos), up=(0, 1, 0), yfov=np.radians(38.0),
                        znear=0.1, zfar=100.0))
    return b.build()


def stress_mesh(n_triangles: int = 10_000_000, seed: int = 11, opening_deg: float = 98.0) -> SceneModel:
    """One opaque dielectric GGX object (roughness 0.5, metallic 0, no textures), identity transform: the BVH-traversal /
    HBM stress configuration.  The mesh is a cavern: a bumpy core (outward faces) inside a bumpy shell (inward faces) whose
    top is open to the white environment within `opening_deg` of the pole; the camera stands between the two, so most
    paths keep bouncing between core and shell (tens of rays per sample at 8 bounces, in every direction through every part
    of the tree) instead of leaving for the environment after the first hit."""
    n_outer = max(8, int(round(np.sqrt(n_triangles * 0.62 / 2.0 / (1.0 - opening_deg / 180.0)))))
    n_inner = max(8, int(round(np.sqrt(n_triangles * 0.38 / 2.0))))
    b = ModelBuilder(f"stress_{n_triangles}")
    mat = b.add_material(base_color=(0.8, 0.8, 0.8, 1), metallic=0.0, roughness=0.5)
    cp, cn, ci, cuv, ct = blob(n_inner, n_inner, 1.0, 0.25, seed=seed, octaves=9)
    sp, sn, si, suv, st = blob(n_outer, n_outer, 2.6, 0.09, seed=seed + 1, octaves=9)
    # the shell faces inwards (winding and normals reversed) and loses the rows of triangles around its +Y pole
    first_row = int(np.ceil(opening_deg / 180.0 * n_outer))
    rows = si.min(axis=1) // (n_outer + 1)   # latitude row of a triangle = row of its lowest-numbered vertex
    si = si[rows >= first_row][:, [0, 2, 1]]
    sn = -sn
    pos = np.concatenate([cp, sp])
    nrm = np.concatenate([cn, sn])
    idx = np.concatenate([ci, si + len(cp)])
    uv = np.concatenate([cuv, suv])
    tan = np.concatenate([ct, st])
    mesh = b.add_mesh(pos, nrm, idx, uv, tan, material=mat)
    b.add_object(mesh)
    eye = np.array([0.35, -0.25, 1.85])
    target = np.array([-1.6, -0.6, 0.2])
    b.add_camera(Camera(position=eye, direction=_normalize(target - eye), up=(0, 1, 0), yfov=np.radians(55.0), znear=0.05,
                        zfar=100.0))
    return b.build()


def material_stress(detail: float = 1.0, seed: int = 21) -> SceneModel:
    """Transmissive + volume + ior object, alpha-masked cards, alpha-blended panel, emissive object, textured floor."""
    b = ModelBuilder("material_stress")
    floor_tex = b.add_image(checker_texture(256, 16, a=(220, 210, 190), b=(40, 40, 60)))
    panel_tex = b.add_image(alpha_gradient_texture(128))
    leaf_tex = b.add_image(leaf_mask_texture(128, seed=seed))
    m_floor = b.add_material(metallic=0.0, roughness=0.9, base_color_texture=floor_tex)
    m_glass = b.add_material(base_color=(1, 1, 1, 1), metallic=0.0, roughness=0.02, ior=1.33, transmission=1.0,
                             attenuation_color=(0.9, 0.3, 0.2), attenuation_distance=0.8)
    m_glass2 = b.add_material(base_color=(0.9, 1.0, 0.9, 1), metallic=0.0, roughness=0.25, ior=1.6, transmission=0.8,
                              attenuation_color=(0.3, 0.9, 0.4), attenuation_distance=0.4)
    m_panel = b.add_material(metallic=0.0, roughness=0.3, base_color_texture=panel_tex, alpha_mode="BLEND",
                             double_sided=True)
    m_leaf = b.add_material(metallic=0.0, roughness=0.7, base_color_texture=leaf_tex, alpha_mode="MASK",
                            alpha_cutoff=0.5, double_sided=True)
    m_emit = b.add_material(base_color=(0.1, 0.1, 0.1, 1), metallic=0.0, roughness=1.0, emissive=(0.3, 0.7, 1.0),
                            emissive_strength=18.0)
    d = max(detail, 0.02)
    nf = max(2, int(32 * d))
    floor = b.add_mesh(*_args(plane(nf, nf, 16, 16, uv_scale=6.0)), material=m_floor)
    ng = max(12, int(300 * d))
    g1 = b.add_mesh(*_args(blob(ng, ng, 1.0, 0.2, seed=seed)), material=m_glass)
    g2 = b.add_mesh(*_args(torus(max(16, int(256 * d)), max(8, int(128 * d)), 0.9, 0.33)), material=m_glass2)
    panel = b.add_mesh(*_args(quad(2.0, 2.0)), material=m_panel)
    leaf = b.add_mesh(*_args(quad(0.9, 0.9)), material=m_leaf)
    emit = b.add_mesh(*_args(box(0.5, 0.5, 0.5)), material=m_emit)
    b.add_object(floor)
    b.add_object(g1, translate(-1.2, 1.2, 0))
    b.add_object(g2, translate(1.5, 0.8, -0.4) @ rotate((1, 0, 0), 70))
    for k in range(4):
        b.add_object(panel, translate(-2.5 + 1.6 * k, 1.1, 2.2 - 0.3 * k) @ rotate((0, 1, 0), -20 + 12 * k))
    rng = np.random.default_rng(seed)
    for k in range(12):
        b.add_object(leaf, translate(rng.uniform(-3, 3), rng.uniform(0.5, 2.2), rng.uniform(-2.5, 1.0)) @
                     rotate(rng.normal(size=3), rng.uniform(0, 360)))
    b.add_object(emit, translate(0.2, 2.9, -1.5))
    b.add_object(emit, translate(-2.8, 0.3, 1.2) @ scale(0.6))
    pos = np.array([0.0, 2.2, 6.5])
    target = np.array([0.0, 1.0, 0.0])
    b.add_camera(Camera(position=pos, direction=_normalize(target - pos), up=(0, 1, 0), yfov=np.radians(42.0)))
    return b.build()


def sparse_clouds(n_objects: int = 14, tris_per_object: int = 60, seed: int = 5) -> SceneModel:
    """``n_objects`` instances whose bounding volumes all cover the same cube while each holds only a sparse cloud of small
    triangles: most rays pierce every bounding volume and hit few triangles, so a ray has many candidate instances."""
    rng = np.random.default_rng(seed)
    b = ModelBuilder("sparse_clouds")
    for k in range(n_objects):
        mat = b.add_material(base_color=(0.3 + 0.05 * (k % 10), 0.6, 0.9 - 0.05 * (k % 10), 1), metallic=0.0, roughness=0.6,
                             double_sided=(k % 3 == 0))
        c = rng.uniform(-1.0, 1.0, (tris_per_object, 1, 3))
        pos = (c + rng.normal(scale=0.12, size=(tris_per_object, 3, 3))).reshape(-1, 3).astype(np.float32)
        # two far corner triangles pin every object's bounding volume to the same cube
        pos = np.concatenate([pos, np.array([[-1.2, -1.2, -1.2], [-1.19, -1.2, -1.2], [-1.2, -1.19, -1.2],
                                             [1.2, 1.2, 1.2], [1.19, 1.2, 1.2], [1.2, 1.19, 1.2]], np.float32)])
        idx = np.arange(len(pos), dtype=np.uint32)
        tri = pos.reshape(-1, 3, 3)
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        n = np.repeat(_normalize(n + 1e-12), 3, axis=0).astype(np.float32)
        mesh = b.add_mesh(pos, n, idx, material=mat)
        m = np.eye(4)
        m[:3, 3] = rng.uniform(-0.02, 0.02, 3)
        b.add_object(mesh, m)
    b.add_camera(Camera((0.0, 0.3, 4.5), (0.0, -0.05, -1.0), (0.0, 1.0, 0.0), np.radians(40.0), 0.1, 100.0))
    return b.build()


def tied_bounds(n_singles: int = 7, tris_per_object: int = 60, seed: int = 8) -> SceneModel:
    """Two instances whose bounding volumes are IDENTICAL (equal entry distance for every ray) inside ``n_singles`` instances
    with larger, nested bounds (nearer entry distances), each holding a sparse cloud of small triangles in the inner cube.
    The tied pair comes first in the instance table (most triangles) and every other instance is sorted in front of it: the
    case in which a sorted candidate list must keep table order among equal keys while entries move past them."""
    rng = np.random.default_rng(seed)
    b = ModelBuilder("tied_bounds")
    halves = [1.0, 1.0] + [1.05 + 0.05 * k for k in range(n_singles)]
    for k, half in enumerate(halves):
        mat = b.add_material(base_color=(0.3 + 0.05 * (k % 10), 0.6, 0.9 - 0.05 * (k % 10), 1), metallic=0.0, roughness=0.6,
                             double_sided=(k % 3 == 0))
        n_tris = tris_per_object + 2 * (len(halves) - k)  # falling counts: the table keeps this order
        c = rng.uniform(-0.8, 0.8, (n_tris, 1, 3))
        pos = (c + rng.normal(scale=0.1, size=(n_tris, 3, 3))).clip(-0.95, 0.95).reshape(-1, 3).astype(np.float32)
        e = np.float32(half)
        pins = np.array([[-e, -e, -e], [-e + 0.01, -e, -e], [-e, -e + 0.01, -e], [e, e, e], [e - 0.01, e, e], [e, e - 0.01, e],
                         [-e, e, e], [-e + 0.01, e, e], [-e, e - 0.01, e], [e, -e, -e], [e - 0.01, -e, -e], [e, -e + 0.01, -e],
                         [e, -e, e], [e - 0.01, -e, e], [e, -e + 0.01, e], [-e, e, -e], [-e + 0.01, e, -e], [-e, e - 0.01, -e],
                         [e, e, -e], [e - 0.01, e, -e], [e, e - 0.01, -e], [-e, -e, e], [-e + 0.01, -e, e], [-e, -e + 0.01, e]], np.float32)
        pos = np.concatenate([pos, pins])
        tri = pos.reshape(-1, 3, 3)
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        n = np.repeat(_normalize(n + 1e-12), 3, axis=0).astype(np.float32)
        mesh = b.add_mesh(pos, n, np.arange(len(pos), dtype=np.uint32), material=mat)
        b.add_object(mesh, np.eye(4))
    b.add_camera(Camera((0.0, 0.3, 4.5), (0.0, -0.05, -1.0), (0.0, 1.0, 0.0), np.radians(40.0), 0.1, 100.0))
    return b.build()


def endless_pass_through() -> SceneModel:
    """An alpha-blended quad whose vertex normals point AWAY from the side its winding calls the front (an exporter's flipped
    normals), over a floor.  A ray that hits the front spawns its pass-through ray at position - normal * 1e-5
    (src/ray_program.cc:265-271) — which, with this normal, is in front of the quad again: the chain never ends and the
    reference's frame loop never returns."""
    b = ModelBuilder("endless_pass_through")
    mat = b.add_material(base_color=(0.8, 0.7, 0.6, 0.5), metallic=0.0, roughness=0.5, alpha_mode="BLEND")
    pos, nrm, idx, uv, tan = _args(quad(2.0, 2.0))
    b.add_object(b.add_mesh(pos, -np.asarray(nrm), idx, uv, tan, material=mat))
    floor = b.add_mesh(*_args(plane(2, 2, 8.0, 8.0)), material=b.add_material(metallic=0.0, roughness=0.8))
    b.add_object(floor, translate(0.0, -1.5, 0.0))
    b.add_camera(Camera((0.0, 0.0, 5.0), (0.0, 0.0, -1.0), (0.0, 1.0, 0.0), np.radians(40.0), 0.1, 100.0))
    return b.build()


def random_soup(seed: int, n_objects: int = 5, tris_per_object: int = 4000, offset: float = 0.0, flat: bool = False) -> SceneModel:
    """Triangle soups under random affine transforms (rotation, non-uniform scale, shear): no structure a builder or a
    node layout could rely on.  ``offset`` moves every mesh far from its object-space origin (large coordinates, small
    extents: box planes and quantisation steps lose bits); ``flat`` adds axis-aligned quads (boxes of zero thickness)."""
    rng = np.random.default_rng(seed)
    b = ModelBuilder(f"random_soup_{seed}")
    for k in range(n_objects):
        mat = b.add_material(base_color=tuple(rng.uniform(0.2, 0.9, 3)) + (1.0,), metallic=float(k % 2), roughness=0.5,
                             double_sided=(k % 2 == 0))
        centres = rng.uniform(-1.0, 1.0, (tris_per_object, 1, 3)) * rng.uniform(0.3, 1.0, 3)
        size = rng.choice([0.01, 0.05, 0.3], (tris_per_object, 1, 1))
        pos = (centres + rng.normal(size=(tris_per_object, 3, 3)) * size)
        if flat:
            q = rng.uniform(-1, 1, (tris_per_object // 4, 1, 3))
            axis = rng.integers(0, 3, len(q))
            e = rng.uniform(0.05, 0.4, (len(q), 3, 3)) * np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]])[None]
            quad = q + e
            for a in range(3):  # rotate the in-plane axes so that the flat axis is `a`
                sel = axis == a
                quad[sel] = np.roll(quad[sel], a + 1, axis=2)
            pos = np.concatenate([pos, quad])
        pos = (pos + offset).reshape(-1, 3).astype(np.float32)
        tri = pos.reshape(-1, 3, 3)
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        n = np.repeat(_normalize(n + 1e-20), 3, axis=0).astype(np.float32)
        mesh = b.add_mesh(pos, n, np.arange(len(pos), dtype=np.uint32), material=mat)
        m = np.eye(4)
        a = rng.normal(size=(3, 3))
        qm, _ = np.linalg.qr(a)
        m[:3, :3] = qm @ np.diag(rng.uniform(0.4, 1.6, 3)) @ (np.eye(3) + np.triu(rng.uniform(-0.2, 0.2, (3, 3)), 1))
        m[:3, 3] = rng.uniform(-0.5, 0.5, 3) - m[:3, :3] @ np.full(3, offset)
        b.add_object(mesh, m)
    b.add_camera(Camera((0.0, 0.2, 5.0), (0.0, -0.03, -1.0), (0.0, 1.0, 0.0), np.radians(45.0), 0.1, 100.0))
    return b.build()


def many_objects(n_objects: int = 400, seed: int = 9) -> SceneModel:
    """``n_objects`` small objects (instances of four meshes: blob, torus, box, sphere) scattered over a floor, each under
    its own rotation and scale: the case in which testing every object's bound for every ray is what costs."""
    rng = np.random.default_rng(seed)
    b = ModelBuilder("many_objects")
    floor_mat = b.add_material(base_color=(0.7, 0.7, 0.72, 1), metallic=0.0, roughness=0.8)
    side = 0.6 * np.sqrt(n_objects)
    b.add_object(b.add_mesh(*_args(plane(8, 8, 2.2 * side, 2.2 * side)), material=floor_mat), np.eye(4))
    meshes = []
    for k, shape in enumerate((blob(10, 20, 0.25), torus(16, 8, 0.22, 0.08), box(0.3, 0.3, 0.3), uv_sphere(8, 16, 0.2))):
        mat = b.add_material(base_color=(0.3 + 0.2 * k, 0.8 - 0.15 * k, 0.4 + 0.1 * k, 1), metallic=float(k % 2), roughness=0.3 + 0.15 * k)
        meshes.append(b.add_mesh(*_args(shape), material=mat))
    for i in range(n_objects):
        m = np.eye(4)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        m[:3, :3] = q * rng.uniform(0.6, 1.5)
        m[:3, 3] = (rng.uniform(-side, side), rng.uniform(0.3, 1.6), rng.uniform(-side, side))
        b.add_object(meshes[i % len(meshes)], m)
    b.add_camera(Camera((0.0, 0.9 * side, 2.0 * side), (0.0, -0.45, -1.0), (0.0, 1.0, 0.0), np.radians(50.0), 0.1, 1000.0))
    return b.build()


def tiny_scene(seed: int = 3) -> SceneModel:
    """A few hundred triangles exercising every material feature; small enough for the CPU oracle in tests."""
    return readme_standin(detail=0.03, seed=seed)


def _args(shape):
    pos, nrm, idx, uv, tan = shape
    return pos, nrm, idx, uv, tan
