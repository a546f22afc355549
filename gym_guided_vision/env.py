"""`from gym_guided_vision.env import SlotInsertionEnv, make_sim_env, ...` as in the reference (env.py)."""
from av_aloha_amd.env import (GuidedVisionEnv, HookPackageEnv, InsertPegEnv, SewNeedleEnv, SlotInsertionEnv,  # noqa: F401
                              TubeTransferEnv, make, make_sim_env, sample_object_poses)
