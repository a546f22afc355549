import sys,os,json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import episode_util as U
for tol in (None, 1e-11, 1e-13):
    U.NEWTON_TOL = tol
    for task in ("slot_insertion","hook_package","sew_needle_thread","tube_transfer","insert_peg"):
        dev = U.device_episode(task, 16, f64=True, record_state=True)
        rows = U.compare_with_replay(task, dev)
        ls = U.compare_lockstep(task, dev)
        e=np.array([r["max_qpos_err"] for r in rows]); h=np.array([r["held_max_qpos_err"] for r in rows]); a=np.array([r["arm_max_qpos_err"] for r in rows])
        print(f"tol {tol} {task}: reward-seq identical {sum(r['first_reward_diff']==-1 for r in rows)}/16, success mism {sum(r['dev_success']!=r['orc_success'] for r in rows)}, dev success {np.mean([r['dev_success'] for r in rows]):.2f}; max_qpos p50/p90/max {np.percentile(e,50):.1e} {np.percentile(e,90):.1e} {e.max():.1e}; held {np.percentile(h,50):.1e} {np.percentile(h,90):.1e} {h.max():.1e}; arm max {a.max():.1e}; ncon diff steps max {max(r['ncon_diff_steps'] for r in rows)}; "
              f"lockstep: one-step max {max(r['max_step_err'] for r in ls):.1e}, steps>1e-6 {sum(r['steps_err_above_1e6'] for r in ls)}, ncon diffs {sum(r['ncon_diff_steps'] for r in ls)}, newton its mean {((np.array(0)))}", flush=True)
