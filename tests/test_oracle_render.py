"""Known answers that pin the oracle's depth ray-caster (oracle/orc_render.c).  The reference renders through MuJoCo's
OpenGL pipeline, which is not available here, and the oracle draws collision proxies: parity with the reference's pixels is
unpinned; what is pinned is the camera model and the ray/geom intersections, from the scene's own geometry
(scene.xml:55-57: table top at z = -0.0009, overhead_cam at z = 1.02524 tilted 25 deg about x, fovy 58)."""
import numpy as np
import pytest

from orc_env import OrcEnv
from test_oracle_physics import OBJ, model_dict


@pytest.fixture(scope="module")
def env():
    e = OrcEnv()
    e.reset(OBJ)
    yield e
    e.close()


def cam_frame(md, e, name):
    from av_aloha_amd.compiler.mjcf import quat_to_mat as quat2mat
    ci = e.man["camera_names"].index(name)
    b = int(md["cam_body"][ci])
    Rb = e.arr("xmat", 9 * md["nbody"]).reshape(-1, 3, 3)[b]
    pb = e.arr("xpos", 3 * md["nbody"]).reshape(-1, 3)[b]
    return pb + Rb @ md["cam_pos"][ci], Rb @ quat2mat(md["cam_quat"][ci]), float(md["cam_fovy"][ci])


def test_overhead_camera_sees_the_table_plane_at_the_analytic_depth(env):
    md = model_dict()
    H, W = 48, 64
    img = env.render_depth("overhead_cam", H, W)
    pc, Rc, fovy = cam_frame(md, env, "overhead_cam")
    assert abs(pc[2] - 1.02524) < 1e-9 and abs(fovy - 58) < 1e-12
    scale = 2 * np.tan(np.radians(fovy) / 2) / H
    ztop = -0.0009
    checked = 0
    for i in range(0, H, 5):
        for j in range(0, W, 7):
            dc = np.array([(j + 0.5 - W / 2) * scale, -(i + 0.5 - H / 2) * scale, -1.0])
            dw = Rc @ dc
            t = (ztop - pc[2]) / dw[2]                      # plane z = ztop
            hit = pc + t * dw
            if abs(hit[0]) < 0.55 and abs(hit[1]) < 0.35 and img[i, j] > t - 1e-6:   # on the table top, nothing above it
                assert abs(img[i, j] - t) < 2e-6, (i, j, img[i, j], t)
                checked += 1
    assert checked > 10
    # the arms and objects stand above the table: some pixels are strictly nearer than the table plane
    assert (img < 0.9).sum() > 20 and img.min() > md["cam_clip"][0]


def test_object_box_top_face_and_background(env):
    md = model_dict()
    H, W = 120, 160
    img = env.render_depth("overhead_cam", H, W)
    pc, Rc, fovy = cam_frame(md, env, "overhead_cam")
    # the stick (task_slot_insertion.xml:14, box half-height 0.01, free joint at OBJ[1]) after the reset forward pass: its
    # top face is the plane z = qpos_z + 0.02 - 0.01 ... read the geom pose from the oracle instead of re-deriving it
    names = env.man["geom_names"]
    g = names.index("stick")
    gp = env.arr("geom_xpos", 3 * len(names)).reshape(-1, 3)[g]
    sz = md["geom_size"].reshape(-1, 3)[g]
    # pixel that looks at the centre of the stick's top face
    target = gp + np.array([0, 0, sz[2]])
    dcam = Rc.T @ (target - pc)
    depth = -dcam[2]
    scale = 2 * np.tan(np.radians(fovy) / 2) / H
    j = int(np.floor(dcam[0] / depth / scale + W / 2))
    i = int(np.floor(-dcam[1] / depth / scale + H / 2))
    dc = np.array([(j + 0.5 - W / 2) * scale, -(i + 0.5 - H / 2) * scale, -1.0])
    t = (target[2] - pc[2]) / (Rc @ dc)[2]
    assert abs(img[i, j] - t) < 2e-6
    # worms_eye_cam looks up from table height: its top rows see no geometry -> far plane (50 x extent 0.6 = 30 m)
    up = env.render_depth("worms_eye_cam", 30, 40)
    assert up.max() == np.float32(md["cam_clip"][1]) and (up < 30).any()


def test_wrist_and_zed_cameras_move_with_the_arms(env):
    md = model_dict()
    a = env.render_depth("wrist_cam_left", 30, 40)
    z = env.render_depth("zed_cam_left", 30, 40)
    assert (a < 30).mean() > 0.3 and (z < 30).mean() > 0.3          # both look at the table / objects at the home pose
    # pixel rays are built from the pixel centre: left-right mirrored cameras give mirrored images of a mirrored scene;
    # here only a sanity relation: the two zed cameras are 6 cm apart and see nearly the same depth at the image centre
    zr = env.render_depth("zed_cam_right", 30, 40)
    assert abs(float(z[15, 20]) - float(zr[15, 20])) < 0.05


def test_colour_image_known_answers(env):
    """orc_render_rgb: the depth of the colour pass is the depth pass; a pixel of the table top seen from the overhead camera has
    the table's flat colour under the analytic light (headlight along the ray + the vertical directional light, scene.xml:9,48);
    the worm's-eye camera's empty rows show the sky gradient (scene.xml:34); slot and stick keep their MJCF hue."""
    md = model_dict()
    H, W = 60, 80
    img, dep = env.render_rgb("overhead_cam", H, W)
    assert np.array_equal(dep, env.render_depth("overhead_cam", H, W))
    pc, Rc, fovy = cam_frame(md, env, "overhead_cam")
    scale = 2 * np.tan(np.radians(fovy) / 2) / H
    # a table pixel well away from robots and objects: front edge of the table, image centre column
    names = env.man["geom_names"]
    rgba = md["geom_rgba"][names.index("table")]
    L = md["render_light"]
    found = False
    for i in range(H - 1, H // 2, -1):
        j = W // 2
        dc = np.array([(j + 0.5 - W / 2) * scale, -(i + 0.5 - H / 2) * scale, -1.0])
        dw = Rc @ dc
        t = (-0.0009 - pc[2]) / dw[2]
        if abs(dep[i, j] - t) < 1e-5:                       # this ray ends on the table top (normal +z)
            n = np.array([0.0, 0.0, 1.0])
            lum = min(1.0, L[0] + L[1] * max(0.0, -(n @ dw) / np.linalg.norm(dw)) + L[2] * max(0.0, -(n @ (L[4:7] / np.linalg.norm(L[4:7])))))
            want = np.floor(np.clip(rgba[:3] * lum, 0, 1) * 255.0 + 0.5)
            assert np.array_equal(img[i, j].astype(np.float64), want), (i, j, img[i, j], want)
            found = True
            break
    assert found
    sky, sdep = env.render_rgb("worms_eye_cam", 30, 40)
    assert sdep[0, 20] == np.float32(md["cam_clip"][1])          # nothing along this ray
    pcw, Rcw, fw = cam_frame(md, env, "worms_eye_cam")
    sc = 2 * np.tan(np.radians(fw) / 2) / 30
    dc = np.array([(20 + 0.5 - 20) * sc, -(0 + 0.5 - 15) * sc, -1.0])
    dw = Rcw @ dc
    w = 0.5 + 0.5 * dw[2] / np.linalg.norm(dw)
    want = np.floor(np.clip(L[12:15] + (L[8:11] - L[12:15]) * w, 0, 1) * 255.0 + 0.5)
    assert np.array_equal(sky[0, 20].astype(np.float64), want)
    r, g, b = (img[..., k].astype(np.int32) for k in range(3))
    assert ((r > 1.8 * g) & (r > 60)).sum() > 5 and ((g > 1.8 * r) & (g > 60)).sum() > 5       # slot (.8 .4 .4), stick (.4 .8 .4)
