"""Host-side logic that runs without a GPU: object-pose sampling vs the reference's own reset() (golden),
registry / factory surface, model blob known answers, and the 2-rank gloo path of the multi-GPU glue."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_reset_sampling_matches_reference_draw_order():
    from av_aloha_amd.env import sample_object_poses
    r = np.load(os.path.join(G, "reset_samples.npz"))
    # golden rows are in the reference's assignment order; ours are in qpos order
    perm = {"insert_peg": [0, 1], "slot_insertion": [0, 1], "sew_needle": [1, 0], "tube_transfer": [0, 1, 2], "hook_package": [0, 1]}
    for task, p in perm.items():
        for seed in range(16):
            np.random.seed(seed)
            got = sample_object_poses(task)
            np.testing.assert_array_equal(got, r[task][seed][p], err_msg=f"{task} seed {seed}")


def test_registry_and_factory_surface():
    from av_aloha_amd import env as E
    assert len(E.ENVS) == 10
    for k, v in E.ENVS.items():
        assert k.startswith("gym_guided_vision/") and k.endswith("Arms-v0")
        assert v["observation_height"] == 480 and v["observation_width"] == 640 and v["num_arms"] in (2, 3)
        assert len(v["cameras"]) == (6 if v["num_arms"] == 3 else 4)
    with pytest.raises(NotImplementedError):
        E.make_sim_env("sim_unknown_task")
    with pytest.raises(AssertionError):
        E.SlotInsertionEnv(num_arms=4, cameras=[])
    with pytest.raises(AssertionError):
        E.SlotInsertionEnv(num_arms=3, cameras=["not_a_camera"])
    import gym_guided_vision.env as ge          # drop-in import names
    import gym_guided_vision.constants as gc
    assert ge.SlotInsertionEnv is E.SlotInsertionEnv and gc.SIM_PHYSICS_ENV_STEP_RATIO == 20 and abs(gc.SIM_DT - 0.04) < 1e-15


def test_model_known_answers():
    """dimensions, masses, zero-pose screw data."""
    from av_aloha_amd.compiler.compile import read_blob
    dims = {"insert_peg": (37, 35, 31), "slot_insertion": (37, 35, 31), "sew_needle": (37, 35, 31),
            "hook_package": (37, 35, 32), "tube_transfer": (44, 41, 32)}
    for task, (nq, nv, nb) in dims.items():
        for na in (2, 3):
            md = read_blob(os.path.join(ROOT, "models", f"{task}_{na}arms.avm"))
            assert (int(md["nq"][0]), int(md["nv"][0]), int(md["nu"][0]), int(md["nbody"][0]), int(md["neq"][0])) == (nq, nv, 21, nb, 2)
    md = read_blob(os.path.join(ROOT, "models", "slot_insertion_3arms.avm"))
    man = json.load(open(os.path.join(ROOT, "models", "slot_insertion_3arms.json")))
    mass = dict(zip(man["body_names"], md["body_mass"]))
    assert abs(mass["slot"] - 100.0) < 1e-12 and abs(mass["stick"] - 0.3536) < 1e-12
    np.testing.assert_allclose(md["ik_p0"][0, :6], [[-0.469, 0.032, 0.099], [-0.469, 0.032, 0.14705], [-0.40945, 0.032, 0.44705],
                                                    [-0.20945, 0.032, 0.44705], [-0.10945, 0.032, 0.44705], [-0.039706, 0.032, 0.44705]], atol=1e-12)
    np.testing.assert_allclose(md["ik_w0"][0, :6], [[0, 0, 1], [0, 1, 0], [0, 1, 0], [1, 0, 0], [0, 1, 0], [1, 0, 0]], atol=1e-12)
    assert md["tree_dofnum"].tolist() == [8, 8, 7, 6, 6]
    # 2-arm models carry the hidden middle arm (env.py:394-395)
    md2 = read_blob(os.path.join(ROOT, "models", "slot_insertion_2arms.avm"))
    i = man["body_names"].index("middle_base_link")
    np.testing.assert_allclose(md2["body_pos"][i], [0, -2.4, -0.4])


GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from av_aloha_amd.dist import shard_ids, gather_episode_stats
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
N = 6
ids = shard_ids(rank, world, N)
ret = torch.tensor(np.sin(ids * 0.37) * 10, dtype=torch.float32)       # per-env values keyed by GLOBAL id
succ = torch.tensor((ids % 3 == 0).astype(np.int32))
all_ret, all_succ = gather_episode_stats(ret, succ, dist)
gid = np.arange(world * N)
ok = np.allclose(all_ret.numpy(), (np.sin(gid * 0.37) * 10).astype(np.float32)) and np.array_equal(all_succ.numpy(), (gid % 3 == 0).astype(np.int32))
flag = torch.tensor([1 if ok else 0])
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.barrier()
if rank == 0:
    print("GLOO_OK" if int(flag.item()) == 1 else "GLOO_BAD")
dist.destroy_process_group()
"""


def test_two_rank_gloo_gather(tmp_path):
    """world_size 2 on CPU: contiguous sharding by global env id + the end-of-rollout all-gather give, on every
    rank, exactly the vector a single process would hold."""
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert "GLOO_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_single_process_gather_is_identity():
    import torch
    from av_aloha_amd.dist import gather_episode_stats, shard_ids
    assert shard_ids(1, 4, 8).tolist() == list(range(8, 16))
    r, s = gather_episode_stats(torch.arange(5, dtype=torch.float32), torch.tensor([0, 1, 0, 1, 1]))
    assert r.tolist() == [0, 1, 2, 3, 4] and s.tolist() == [0, 1, 0, 1, 1]


def test_mat2quat_host_port_matches_the_reference_vectors():
    """av_aloha_amd.sim_env.mat2quat_xyzw (used for the obs 'poses' of the Cartesian env) vs the vectors produced by the
    reference's transform_utils.mat2quat (tests/golden/so3_helpers.npz, gen_golden.py:96)."""
    import os
    from av_aloha_amd.sim_env import mat2quat_xyzw
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "so3_helpers.npz"))
    q = mat2quat_xyzw(g["quat2mat"])
    assert np.abs(q - g["mat2quat"]).max() < 1e-12


def test_data_collection_model_variant():
    """sim_env.py loads data_collection_scripts/assets (data_collection_scripts/constants.py:5), which differ from the gym package's
    assets in exactly three places: the needle and the peg have MuJoCo's default solref (task_sew_needle.xml:17 and
    task_insert_peg.xml:7 carry solref="0.01 1" only in the gym assets) and the ZED cameras have fovy 90 instead of 66.21
    (aloha_sim.xml:357-358).  The compiled dc_* blobs differ from the gym blobs in those arrays and nowhere else."""
    import os
    from av_aloha_amd.compiler.compile import read_blob
    from av_aloha_amd.constants import MODEL_DIR
    import json
    for task, obj in (("sew_needle", "needle"), ("insert_peg", "peg"), ("slot_insertion", None), ("hook_package", None), ("tube_transfer", None)):
        a = read_blob(os.path.join(MODEL_DIR, f"{task}_3arms.avm"))
        b = read_blob(os.path.join(MODEL_DIR, f"dc_{task}_3arms.avm"))
        diff = sorted(k for k in a if not np.array_equal(a[k], b[k]))
        assert diff == (["cam_fovy", "geom_solref", "pair_solref"] if obj else ["cam_fovy"]), (task, diff)
        names = json.load(open(os.path.join(MODEL_DIR, f"dc_{task}_3arms.json")))
        cams = names["camera_names"]
        for c in ("zed_cam_left", "zed_cam_right"):
            assert a["cam_fovy"][cams.index(c)] == 66.21 and b["cam_fovy"][cams.index(c)] == 90.0
        assert names["variant"] == "data_collection"
        if obj:
            g = names["geom_names"].index(obj)
            assert a["geom_solref"].reshape(-1, 2)[g, 0] == 0.01 and b["geom_solref"].reshape(-1, 2)[g, 0] == 0.02       # MuJoCo default 0.02 1
            changed = np.nonzero(np.any(a["pair_solref"].reshape(-1, 2) != b["pair_solref"].reshape(-1, 2), axis=1))[0]
            assert len(changed) > 0 and all(g in a["pair_geom"].reshape(-1, 2)[p] for p in changed)                      # only pairs of that geom


def test_bench_gpus_flag_refuses_a_node_with_fewer_gpus():
    """bench.py --gpus N starts N ranks itself (one process per GPU); on a node with fewer GPUs it must fail loudly and
    print no line (it used to ignore the flag and report n_gpus: 1)."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "needs 2 GPUs" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
