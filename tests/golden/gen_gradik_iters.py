"""Per-iteration golden vectors of the reference's GradIK (data_collection_scripts/grad_ik.py:8-99), produced by importing it
under the shims of gen_golden.py: for 64 inputs per manipulator the controller is run with max_iterations = k for a ladder of
k, which exposes how the reference's answer evolves with the iteration count (q_out(k) = q + joint_p (best_k - q)).  Also the
limit_pose outputs (clamped target) of the same inputs.  Runs ONLY in the build container (needs /root/reference); writes
tests/golden/gradik_iters.npz.

    python tests/golden/gen_gradik_iters.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as GG  # noqa: E402

ITERS = (1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 50)


def main():
    GG.install_shims()
    import transform_utils as T
    import kinematics as K
    from grad_ik import GradIK
    from av_aloha_amd.compiler.compile import read_blob
    md = read_blob(os.path.join(GG.ROOT, "models", "slot_insertion_3arms.avm"))
    out = {"iters": np.array(ITERS)}
    for a, name in ((0, "left"), (1, "right")):
        d = np.load(os.path.join(HERE, f"gradik_{name}.npz"))
        n = 64
        nj = int(md["ik_n"][a])
        ph = GG.FakePhysics(md["ik_w0"][a, :nj], md["ik_p0"][a, :nj], md["ik_range"][a, :nj], md["ik_site0"][a])
        joints = list(range(nj))
        fk = K.create_fk_fn(ph, joints, "site")
        q, tpos, tquat = d["q"][:n], d["target_pos"][:n], d["target_quat_wxyz"][:n]
        res = np.zeros((len(ITERS), n, nj))
        for ki, k in enumerate(ITERS):
            g = GradIK(physics=ph, joints=joints, actuators=None, eef_site="site", step_size=0.0001, min_cost_delta=1.0e-12,
                       max_iterations=k, position_weight=500.0, rotation_weight=100.0,
                       joint_center_weight=np.array([10.0, 10.0, 1.0, 50.0, 1.0, 1.0]), joint_displacement_weight=np.array(6 * [50.0]),
                       position_threshold=0.001, rotation_threshold=0.001, max_pos_diff=0.1, max_rot_diff=0.3, joint_p=0.9)
            for i in range(n):
                res[ki, i] = g.run(q[i].copy(), tpos[i].copy(), tquat[i].copy())
        # the clamped target the descent works on (transform_utils.py:263-287)
        lp_pos, lp_mat = np.zeros((n, 3)), np.zeros((n, 9))
        for i in range(n):
            cur = fk(q[i])
            tm = T.quat2mat(T.wxyz_to_xyzw(tquat[i]))
            p, m_ = T.limit_pose(cur[:3, 3], cur[:3, :3], tpos[i], tm, 0.1, 0.3)
            lp_pos[i], lp_mat[i] = p, np.asarray(m_, dtype=np.float64).reshape(-1)
        out[f"{name}_q"], out[f"{name}_pos"], out[f"{name}_quat_wxyz"] = q, tpos, tquat
        out[f"{name}_q_out"], out[f"{name}_limit_pos"], out[f"{name}_limit_mat"] = res, lp_pos, lp_mat
    np.savez_compressed(os.path.join(HERE, "gradik_iters.npz"), **out)


if __name__ == "__main__":
    main()
