import os, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import episode_util as U
from av_aloha_amd.sim import BatchedSim
z=np.load('/root/repo/tools/dbg_state_hook7_183.npz')
sim = BatchedSim('hook_package', 3, 1, f64=True, variant=U.VARIANT, options={"newton_tol": 1e-13, "newton_iters": 99})
sim.reset(z["pose"][None]); sim.set_state(z["q"][None], z["v"][None], z["c"][None], z["w"][None])
sim.step_ctrl(1)
