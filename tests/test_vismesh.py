"""The visual mesh library of the colour renderer (av_aloha_amd/compiler/vismesh.py, models/visual_meshes.avv) and the oracle's
ray caster over it (oracle/orc_vis.c): CPU checks -- the decimation keeps every mesh inside its bounding box and every scene under
the triangle budget, the instances of a model expand to what the manifest says, known answers of the images."""
import json
import os

import numpy as np

from av_aloha_amd.compiler import vismesh
from av_aloha_amd.compiler.compile import read_blob
from orc_env import OrcEnv
from test_oracle_physics import OBJ, ROOT

LIB = os.path.join(ROOT, "models", "visual_meshes.avv")


def test_cluster_decimation_of_a_sphere():
    V, F = vismesh.unit_sphere(sub=3)                 # 1280 triangles
    V2, F2 = vismesh.cluster_decimate(V, F, 0.5)
    assert 8 <= len(F2) < len(F) / 4
    assert np.abs(V2).max() <= 1.0 + 1e-12 and np.linalg.norm(V2, axis=1).min() > 0.5      # cluster means stay inside the hull, near the surface
    assert (F2[:, 0] != F2[:, 1]).all() and (F2[:, 1] != F2[:, 2]).all() and F2.max() < len(V2)
    V3, F3 = vismesh.cluster_decimate(V, F, 1e-6)      # a cell smaller than every edge changes nothing
    assert len(F3) == len(F)


def test_primitives_are_closed_surfaces():
    for V, F in (vismesh.unit_box(), vismesh.unit_cylinder(), vismesh.unit_sphere()):
        e = np.sort(np.concatenate([F[:, [0, 1]], F[:, [1, 2]], F[:, [2, 0]]]), axis=1)
        _, cnt = np.unique(e, axis=0, return_counts=True)
        assert (cnt == 2).all()                         # every edge belongs to two triangles
        n = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])
        assert (np.einsum("ij,ij->i", n, V[F].mean(1)) > 0).all()       # outward winding


def test_library_and_instances_of_every_model():
    lib = read_blob(LIB)
    info = json.load(open(os.path.join(ROOT, "models", "visual_meshes.json")))
    assert 0.003 < info["cell_m"] < 0.02
    assert max(info["scene_triangles"].values()) <= 20000
    tex = lib["lib_tex"]
    assert tex.shape == (vismesh.TEX_N * vismesh.TEX_N,)
    rgb = np.stack([tex & 255, (tex >> 8) & 255, (tex >> 16) & 255], 1).mean(0)
    assert rgb[0] > rgb[1] > rgb[2] > 20                # the table's wood: brown
    for task in ("insert_peg", "slot_insertion", "sew_needle", "tube_transfer", "hook_package"):
        for arms in (2, 3):
            mdl = read_blob(os.path.join(ROOT, "models", f"{task}_{arms}arms.avm"))
            V, B, F, C, U, T, TN = vismesh.expand_instances(lib, mdl)
            assert len(F) <= 20000 and F.max() < len(V) and B.max() < int(mdl["nbody"][0] if "nbody" in mdl else 64)
            assert T.sum() == 220                        # tabletop.obj + tablelegs.obj carry the texture
            assert np.isfinite(V).all() and np.abs(V).max() < 3.0
    # the data-collection variant shares the library
    mdl = read_blob(os.path.join(ROOT, "models", "dc_sew_needle_3arms.avm"))
    assert len(vismesh.expand_instances(lib, mdl)[2]) <= 20000


def test_oracle_image_known_answers():
    lib = read_blob(LIB)
    mdl = read_blob(os.path.join(ROOT, "models", "slot_insertion_3arms.avm"))
    scene = vismesh.expand_instances(lib, mdl) + (lib["lib_tex"],)
    e = OrcEnv()
    e.reset(OBJ)
    H, W = 30, 40
    img, tid, dep = e.render_visual("overhead_cam", H, W, scene)
    # the overhead camera (scene.xml:56, 1.03 m above the table plane, tilted) looks at the table: a good part of the image is the
    # textured table top, about a metre away
    table = (tid >= 0) & (scene[5][np.maximum(tid, 0)] == 1)
    assert table.mean() > 0.2 and 0.9 < dep[table].min() and np.median(dep[table]) < 1.3
    # the stick (rgba .4 .8 .4) and the slot walls (.8 .4 .4) are in view, in their colours
    o = img.astype(int)
    assert ((o[..., 1] > 1.7 * o[..., 0]) & (o[..., 1] > 80)).sum() >= 1
    assert ((o[..., 0] > 1.7 * o[..., 1]) & (o[..., 0] > 80)).sum() >= 1
    # the worm's-eye camera looks up from the table edge: sky gradient (blue) above the horizon
    sky, tid2, _ = e.render_visual("worms_eye_cam", H, W, scene)
    top = sky[0].astype(int)
    assert (tid2[0] < 0).mean() > 0.5 and (top[tid2[0] < 0][:, 2] >= top[tid2[0] < 0][:, 0]).all()
    e.close()


def test_corner_normals_known_answers():
    """compiler/vismesh.py corner_normals (round 6: smooth shading): a sphere's corner normals are its radii, a box stays faceted (its 90-degree
    edges are beyond the crease angle), a cylinder is smooth round its side and flat on its caps; the library carries unit normals for every triangle
    and an instance's rotation and scale carry them into the body frame."""
    V, F = vismesh.unit_sphere()
    N = vismesh.corner_normals(V, F)
    assert np.abs(N - V[F]).max() < 0.05                       # (area-weighted facet mean against the exact radius)
    V, F = vismesh.unit_box()
    N = vismesh.corner_normals(V, F)
    fn = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])
    fn /= np.linalg.norm(fn, axis=1, keepdims=True)
    assert np.abs(N - fn[:, None, :]).max() < 1e-12
    V, F = vismesh.unit_cylinder()
    N = vismesh.corner_normals(V, F)
    fn = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])
    fn /= np.linalg.norm(fn, axis=1, keepdims=True)
    side = np.abs(fn[:, 2]) < 0.5
    radial = V[F[side]] * np.array([1.0, 1.0, 0.0])
    # (a ring vertex sits in two triangles of one facet and one of the other: the area-weighted mean leans 2.5 degrees towards the first)
    assert np.abs(N[side] - radial).max() < 0.05 and np.abs(N[side][..., 2]).max() < 1e-12 and np.abs(N[~side] - fn[~side][:, None, :]).max() < 1e-12
    lib = read_blob(LIB)
    assert lib["lib_tnorm"].shape == (len(lib["lib_tri"]), 9)
    assert np.abs(np.linalg.norm(lib["lib_tnorm"].reshape(-1, 3), axis=1) - 1).max() < 1e-9
    mdl = read_blob(os.path.join(ROOT, "models", "slot_insertion_3arms.avm"))
    V, B, F, C, U, T, TN = vismesh.expand_instances(lib, mdl)
    assert TN.shape == (len(F), 9) and np.abs(np.linalg.norm(TN.reshape(-1, 3), axis=1) - 1).max() < 1e-9
    # the corner normals lie on their own face's side (within the crease angle of it)
    fn = np.cross(V[F[:, 1]] - V[F[:, 0]], V[F[:, 2]] - V[F[:, 0]])
    ok = np.linalg.norm(fn, axis=1) > 1e-12
    fn = fn[ok] / np.linalg.norm(fn[ok], axis=1, keepdims=True)
    assert (np.einsum("tcj,tj->tc", TN.reshape(-1, 3, 3)[ok], fn) > 0.75).mean() > 0.999
