"""`from gym_guided_vision.constants import ...` as in the reference (constants.py)."""
from av_aloha_amd.constants import *  # noqa: F401,F403
