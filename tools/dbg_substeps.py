"""Debug aid: where inside ONE env-step does the f64 device leave the oracle?  The scripted episode of <task> runs on the device (f64); the
oracle is teacher-forced along it (tests/episode_util.py lockstep_worker); for the env-step with the largest one-step difference both sides
are then stepped SUBSTEP BY SUBSTEP from that step's state: position / velocity difference, contact count, and at the first substep whose
contact lists differ both lists.     usage: python tools/dbg_substeps.py <task> [n_envs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import episode_util as U
from av_aloha_amd.sim import BatchedSim

task = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if os.environ.get("NEWTON_TOL"):
    U.NEWTON_TOL = float(os.environ["NEWTON_TOL"])          # both sides' Newton tolerance (tests/episode_util.py)
np.set_printoptions(precision=6, suppress=False, linewidth=220)
dev = U.device_episode(task, n, f64=True, record_state=True)
ls = U.pool_map(U.lockstep_worker, [(task, dev["poses"][k], *(np.ascontiguousarray(dev[x][:, k]) for x in ("q0", "v0", "w0", "l0", "ctrl", "qpos"))) for k in range(n)])
err = np.array([r[2] for r in ls])          # [n, T]
order = np.dstack(np.unravel_index(np.argsort(-err, axis=None), err.shape))[0][:int(os.environ.get("NSHOW", "3"))]
model = U.MODEL_OF.get(task, task)
for k, t in order:
    print(f"\n=== env {k} step {t}: one-step |dq| {err[k, t]:.3e}; device ncon {dev['ncon'][t, k]} oracle {ls[k][3][t]}; reward dev {dev['reward'][t, k]} orc {ls[k][0][t]}")
    sim = BatchedSim(model, 3, 1, f64=True, variant=U.VARIANT, options={"newton_tol": U.NEWTON_TOL} if U.NEWTON_TOL is not None else None)
    sim.reset(dev["poses"][k][None])
    sim.set_state(dev["q0"][t, k][None], dev["v0"][t, k][None], dev["ctrl"][t, k][None], dev["w0"][t, k][None])
    sim.set_latch(np.array([dev["l0"][t, k]], dtype=np.int32))
    e = U._new_env(task, dev["poses"][k])
    e.qpos[:] = dev["q0"][t, k]; e.qvel[:] = dev["v0"][t, k]; e.arr("qacc_warmstart", e.nv)[:] = dev["w0"][t, k]; e.ctrl[:] = dev["ctrl"][t, k]
    names = e.man["geom_names"]
    shown = False
    for s in range(20):
        sim.step_ctrl(1)
        e.step(1)
        q, v, _, w = sim.get_state()
        nc, pairs, dist = sim.contacts()
        dq, dv = np.abs(q[0] - np.array(e.qpos)), np.abs(v[0] - np.array(e.qvel))
        d = sim.diag()[0]
        print(f"  substep {s:2d}: |dq| {dq.max():.3e} (qpos[{dq.argmax()}])  |dv| {dv.max():.3e}  ncon dev {nc[0]} orc {e.d.ncon}  nefc dev {d[1]} orc {e.d.nefc}  newton its dev {(d[3] >> 16) & 0xfff} orc {e.d.stat_sweeps}  noslip sweeps orc {e.d.stat_noslip}")
        oc = [(c.geom1, c.geom2, c.dist) for c in list(e.d.contact)[:e.d.ncon]]
        dc = [(int(pairs[0, i, 0]), int(pairs[0, i, 1]), float(dist[0, i])) for i in range(nc[0])]
        same = len(oc) == len(dc) and all(a[:2] == b[:2] and abs(a[2] - b[2]) < 1e-9 for a, b in zip(oc, dc))
        if not same and not shown:
            shown = True
            print("    contact lists differ (state AFTER this substep; geom names, dist):")
            for i in range(max(len(oc), len(dc))):
                a = f"{names[dc[i][0]]:>22s} {names[dc[i][1]]:<22s} {dc[i][2]: .6e}" if i < len(dc) else " " * 60
                b = f"{names[oc[i][0]]:>22s} {names[oc[i][1]]:<22s} {oc[i][2]: .6e}" if i < len(oc) else ""
                flag = "" if i < len(dc) and i < len(oc) and dc[i][:2] == oc[i][:2] and abs(dc[i][2] - oc[i][2]) < 1e-9 else "   <--"
                print(f"      dev {a} | orc {b}{flag}")
    sim.close(); e.close()
