"""Drop-in import name for the reference package `gym_guided_vision` (reference:
gym_guided_vision/gym_guided_vision/__init__.py:88-101): importing it registers the ten
`gym_guided_vision/<Task>-<N>Arms-v0` ids with gymnasium when gymnasium is installed.  Everything is
served by av_aloha_amd (HIP kernels behind include/avsim.h)."""
from av_aloha_amd.env import ENVS, register_with_gymnasium

register_with_gymnasium("gym_guided_vision.env")
