"""The scripted policies live in the package (av_aloha_amd/scripted.py: they double as the data-collection teleoperator); the tests' old import name."""
from av_aloha_amd.scripted import *  # noqa: F401,F403
from av_aloha_amd.scripted import GRASP_HEIGHT, grasp_lift_targets, qmul  # noqa: F401
