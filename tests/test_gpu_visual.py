"""Colour images of the VISUAL meshes (env.py:180-188 "pixels", :195-200 render) -- the device's triangle
rasteriser (avsim_load_visual + avsim_render_rgb through the C-ABI) against the oracle's brute-force f64 ray caster
(oracle/orc_vis.c) over the same expanded scene (compiler/vismesh.py) and the same body poses.

The two differ in method (projection + near-plane clipping + tile binning + a depth test on 1 / depth in f32, against one ray per
pixel through every triangle in f64), so they can disagree only where a pixel centre lies within rounding of a triangle edge or
where two surfaces are closer than f32 resolves: bounded at 2 % of the pixels; the others agree to +-2 levels.  Parity with the
reference's OpenGL pixels is unpinned."""
import os

import numpy as np
import pytest

from orc_env import OrcEnv
from test_gpu_physics import actions_wiggle
from test_oracle_physics import OBJ, ROOT, model_dict

pytestmark = pytest.mark.gpu
CAMS = ["zed_cam_left", "zed_cam_right", "wrist_cam_left", "wrist_cam_right", "overhead_cam", "worms_eye_cam"]


def scene_of(task, arms):
    from av_aloha_amd.compiler import vismesh
    from av_aloha_amd.compiler.compile import read_blob
    lib = read_blob(os.path.join(ROOT, "models", "visual_meshes.avv"))
    mdl = read_blob(os.path.join(ROOT, "models", f"{task}_{arms}arms.avm"))
    return vismesh.expand_instances(lib, mdl) + (lib["lib_tex"],)


def agree(img, ref, frac):
    bad = (np.abs(img.astype(np.int32) - ref.astype(np.int32)) > 2).any(-1)
    assert bad.mean() <= frac, f"{bad.sum()} of {bad.size} pixels differ"
    return bad.mean()


def test_visual_images_match_the_oracle_after_motion():
    from av_aloha_amd.sim import BatchedSim
    H, W = 60, 80
    md = model_dict()
    acts = actions_wiggle(md, 6)
    sim = BatchedSim("slot_insertion", 3, 3, f64=True, options={"solver": 1})
    e = OrcEnv()
    e.d.solver = 1
    sim.reset(np.repeat(OBJ[None], 3, 0))
    e.reset(OBJ)
    for a in acts:
        sim.step(np.repeat(a[None], 3, 0))
        e.env_step(a)
    sim.set_option("render_smooth", 1)          # (on in the gym / Cartesian facades, off in a bare handle, like shadows and multisampling)
    img = sim.render_rgb(CAMS, H, W)
    info = sim.visual_info()
    assert info["triangles"] > 15000 and info["overflow"] == 0, info
    assert img.shape == (3, len(CAMS), H, W, 3) and img.dtype == np.uint8
    assert np.array_equal(img[0], img[1]) and np.array_equal(img[0], img[2])       # identical envs, identical images
    scene = scene_of("slot_insertion", 3)
    for ci, cam in enumerate(CAMS):
        ref, tid, dep = e.render_visual(cam, H, W, scene, smooth=True)
        agree(img[0, ci], ref, 0.02)
        assert (tid >= 0).mean() > 0.3                                             # the scene fills the view
    # known answers in the overhead view: the stick is green (task_slot_insertion.xml rgba), the table shows its wood texture
    ref, tid, dep = e.render_visual("overhead_cam", H, W, scene)
    o = img[0, CAMS.index("overhead_cam")].astype(int)
    green = (o[..., 1] > 1.7 * o[..., 0]) & (o[..., 1] > 1.7 * o[..., 2]) & (o[..., 1] > 80)      # rgba .4 .8 .4
    assert green.sum() >= 5
    tex = scene[5]
    table_px = tex[np.maximum(tid, 0)].astype(bool) & (tid >= 0)
    assert table_px.mean() > 0.2
    wood = o[table_px].mean(0)
    assert wood[0] > wood[1] > wood[2] and wood[0] > 60, wood                      # small_meta_table_diffuse.png is brown
    # option render_smooth = 0: one shade per triangle (rounds 3-5), against the oracle's flat mode; and smooth shading changes the picture
    sim.set_option("render_smooth", 0)
    flat = sim.render_rgb(CAMS[:2], H, W)
    for ci, cam in enumerate(CAMS[:2]):
        agree(flat[0, ci], e.render_visual(cam, H, W, scene, smooth=False)[0], 0.02)
    assert (np.abs(flat[0, 0].astype(int) - img[0, 0].astype(int)).max(-1) > 2).mean() > 0.005       # (6 % of the pixels by more than 4 levels at 480 x 640)
    sim.set_option("render_smooth", 1)
    # the proxy image is still there, and it is a different picture (hull proxies in flat colours)
    prox = sim.render_rgb(CAMS[:1], H, W, visual=False)
    assert prox.shape == (3, 1, H, W, 3) and not np.array_equal(prox[0, 0], img[0, 0])
    sim.close()
    e.close()


def test_visual_full_size_ragged_size_and_two_arm_model():
    """480 x 640 (env.py:39-40) and the Cartesian env's 720 x 720 ZED images (sim_env.py) run without exhausting the triangle or
    tile-list capacity; a size that is not a multiple of the 8-pixel tile and the 2-arm model are checked against the oracle; the
    low-resolution image samples the same scene as the full-size one."""
    from av_aloha_amd.sim import BatchedSim
    sim = BatchedSim("slot_insertion", 3, 2)
    sim.reset(np.repeat(OBJ[None], 2, 0))
    e = OrcEnv()
    e.reset(OBJ)
    big = sim.render_rgb(CAMS, 480, 640)
    assert big.shape == (2, 6, 480, 640, 3) and sim.visual_info()["overflow"] == 0
    assert np.array_equal(big[0], big[1])
    z = sim.render_rgb(["zed_cam_left", "zed_cam_right"], 720, 720)
    assert z.shape == (2, 2, 720, 720, 3) and sim.visual_info()["overflow"] == 0
    scene = scene_of("slot_insertion", 3)
    small = sim.render_rgb(["wrist_cam_right"], 45, 75)[0, 0]
    agree(small, e.render_visual("wrist_cam_right", 45, 75, scene)[0], 0.025)
    # 120 x 160 samples the scene at the corners shared by 4 x 4 full-size pixels: most of its pixels lie within the range of their block
    low = sim.render_rgb(["overhead_cam"], 120, 160)[0, 0].astype(int)
    blk = big[0, CAMS.index("overhead_cam")].astype(int).reshape(120, 4, 160, 4, 3)
    inside = ((low >= blk.min(axis=(1, 3)) - 3) & (low <= blk.max(axis=(1, 3)) + 3)).all(-1)
    assert inside.mean() > 0.93
    sim.close()
    e.close()
    sim2 = BatchedSim("hook_package", 2, 1)
    md2 = model_dict("hook_package", 2)
    obj = md2["qpos_home"][md2["objects_qposadr"][0]:].reshape(-1, 7)
    sim2.reset(obj[None])
    e2 = OrcEnv("hook_package", 2)
    e2.reset(obj)
    agree(sim2.render_rgb(["overhead_cam"], 60, 80)[0, 0], e2.render_visual("overhead_cam", 60, 80, scene_of("hook_package", 2))[0], 0.02)
    sim2.close()
    e2.close()


def test_shadows_and_supersampling_match_the_oracle():
    """Options "render_shadows" / "render_samples" (round 5: what MuJoCo's renderer does by default [EXT] -- the directional light of scene.xml:48
    casts shadows, the offscreen buffer has 4 samples): the device's depth map from the light + 2 x 2 supersampling against the oracle's exact
    shadow ray and four rays per pixel (oracle/orc_vis.c orc_vis_render_ex).  Shadowed pixels exist and are darker; away from shadow edges the
    images agree as the plain ones do; the edges (a 512 x 512 depth map over 1.2 m against exact rays) are bounded."""
    from av_aloha_amd.sim import BatchedSim
    H, W = 60, 80
    md = model_dict()
    acts = actions_wiggle(md, 6)
    sim = BatchedSim("slot_insertion", 3, 2, f64=True, options={"solver": 1})
    e = OrcEnv()
    e.d.solver = 1
    sim.reset(np.repeat(OBJ[None], 2, 0))
    e.reset(OBJ)
    for a in acts:
        sim.step(np.repeat(a[None], 2, 0))
        e.env_step(a)
    scene = scene_of("slot_insertion", 3)
    cams = ["overhead_cam", "zed_cam_left", "wrist_cam_right"]
    plain = sim.render_rgb(cams, H, W)
    sim.set_option("render_samples", 4)
    ss = sim.render_rgb(cams, H, W)
    sim.set_option("render_shadows", 1)
    both = sim.render_rgb(cams, H, W)
    assert np.array_equal(both[0], both[1]) and sim.visual_info()["overflow"] == 0
    for ci, cam in enumerate(cams):
        ref_ss, _, _ = e.render_visual(cam, H, W, scene, ss=2)
        agree(ss[0, ci], ref_ss, 0.02)                                      # supersampling alone: as exact as the plain image
        ref, _, _ = e.render_visual(cam, H, W, scene, ss=2, shadows=True)
        frac = agree(both[0, ci], ref, 0.06)
        dark_ref = (ref.astype(int).sum(-1) < ref_ss.astype(int).sum(-1) - 12)
        dark_dev = (both[0, ci].astype(int).sum(-1) < ss[0, ci].astype(int).sum(-1) - 12)
        print(f"{cam}: shadowed pixels oracle {dark_ref.mean():.3f} device {dark_dev.mean():.3f}, differing pixels {frac:.3f}")
        if cam == "overhead_cam":
            assert dark_ref.mean() > 0.02 and abs(dark_dev.mean() - dark_ref.mean()) < 0.02      # the arms and the frame shade the table
        assert (both[0, ci].astype(int) <= ss[0, ci].astype(int) + 1).all()                          # a shadow only darkens
    assert not np.array_equal(plain[0], ss[0])
    # a finer depth map ("render_shadow_size" 2048: 0.6 mm texels) brings the shadow edges closer to the oracle's exact rays
    sim.set_option("render_shadow_size", 2048)
    fine = sim.render_rgb(cams, H, W)
    ref, _, _ = e.render_visual("overhead_cam", H, W, scene, ss=2, shadows=True)
    bad512 = (np.abs(both[0, 0].astype(int) - ref.astype(int)) > 2).any(-1).mean()
    bad2048 = (np.abs(fine[0, 0].astype(int) - ref.astype(int)) > 2).any(-1).mean()
    print(f"overhead_cam differing pixels: 512-texel map {bad512:.4f}, 2048-texel map {bad2048:.4f}")
    assert bad2048 <= bad512 + 0.002 and np.array_equal(fine[0], fine[1])
    sim.close()
    e.close()


def test_large_host_copies_take_the_pipelined_path_and_arrive_intact():
    """Results of 16 MB and more go to host memory through two pinned staging buffers and four copy threads (avsim_api.hip out_end_pipelined)
    instead of one pageable hipMemcpy: a 29 MB call (8 envs x 4 cameras x 480 x 640 x 3, not a multiple of the 32 MB chunk) and a 59 MB call
    (depth, f32: two chunks) against the same images fetched one camera at a time (7 / 9.8 MB each: the plain path)."""
    from av_aloha_amd.sim import BatchedSim
    N = 8
    sim = BatchedSim("slot_insertion", 3, N)
    rng = np.random.default_rng(3)
    obj = np.repeat(OBJ[None], N, 0).copy()
    obj[:, :, :2] += 0.02 * rng.standard_normal((N, obj.shape[1], 2))
    sim.reset(obj)
    cams = ["zed_cam_left", "wrist_cam_left", "wrist_cam_right", "overhead_cam"]
    big = sim.render_rgb(cams, 480, 640)
    assert big.nbytes > (16 << 20)
    for ci, c in enumerate(cams):
        one = sim.render_rgb([c], 480, 640)
        assert one.nbytes < (16 << 20) and np.array_equal(big[:, ci], one[:, 0]), c
    cams8 = cams + ["zed_cam_right", "worms_eye_cam"]
    dbig = sim.render_depth(cams8, 480, 640)
    assert dbig.nbytes > (32 << 20)
    for ci, c in enumerate(cams8):
        assert np.array_equal(dbig[:, ci], sim.render_depth([c], 480, 640)[:, 0]), c
    sim.close()


def test_pose_pass_and_shadow_maps_are_reused_per_state_and_renewed_after_a_step():
    """The image calls keep the body poses and the shadow maps of the state they were made for (avsim_api.hip state_ver): cameras fetched one
    call at a time give the images of one call for all of them, and a step, a set_qpos and a reset each invalidate what is kept."""
    from av_aloha_amd.sim import BatchedSim
    md = model_dict()
    sim = BatchedSim("slot_insertion", 3, 2, options={"render_shadows": 1})
    sim.reset(np.repeat(OBJ[None], 2, 0))
    cams = ["overhead_cam", "zed_cam_left"]
    a = np.repeat(actions_wiggle(md, 1)[0][None], 2, 0)
    both = sim.render_rgb(cams, 120, 160)
    for ci, c in enumerate(cams):
        assert np.array_equal(sim.render_rgb([c], 120, 160)[:, 0], both[:, ci])
    for _ in range(5):
        sim.step(a)
    moved = sim.render_rgb(cams, 120, 160)
    assert (moved != both).any(axis=-1).mean() > 1e-3                       # the arms moved: new poses, new shadows
    fresh = BatchedSim("slot_insertion", 3, 2, options={"render_shadows": 1})
    q = sim.get_state()[0]
    fresh.reset(np.repeat(OBJ[None], 2, 0))
    fresh.render_rgb(cams, 120, 160)                                         # (caches the home pose's passes)
    fresh.set_qpos(q)
    assert np.array_equal(fresh.render_rgb(cams, 120, 160), moved)
    fresh.reset(np.repeat(OBJ[None], 2, 0))
    assert np.array_equal(fresh.render_rgb(cams, 120, 160), both)
    sim.close(); fresh.close()
