"""Constants of the guided-vision simulation, mirroring the values of the
reference's gym_guided_vision/gym_guided_vision/constants.py:8-88 (camera list,
time steps, home poses, element names).  Names are part of the public surface
(`from gym_guided_vision.constants import ...`), so they are kept verbatim.
"""
import os
import pathlib

_HERE = pathlib.Path(__file__).parent.resolve()
MODEL_DIR = str(_HERE.parent / "models")
XML_DIR = MODEL_DIR          # reference name; here it points at the compiled model blobs
DATA_DIR = os.path.join(str(_HERE), "data", "recordings")

CAMERAS = ["zed_cam_left", "zed_cam_right", "wrist_cam_left", "wrist_cam_right", "overhead_cam", "worms_eye_cam"]
RENDER_CAMERA = "overhead_cam"

# time stepping (constants.py:20-23): 500 Hz physics, 25 Hz control
SIM_PHYSICS_DT = 0.002
SIM_PHYSICS_ENV_STEP_RATIO = int(0.04 / SIM_PHYSICS_DT)
SIM_DT = SIM_PHYSICS_DT * SIM_PHYSICS_ENV_STEP_RATIO

# home poses (constants.py:26-28); 7th entry of the manipulators is the finger opening in metres
LEFT_ARM_POSE = [0, -0.082, 1.06, 0, -0.953, 0, 0.02239]
RIGHT_ARM_POSE = [0, -0.082, 1.06, 0, -0.953, 0, 0.02239]
MIDDLE_ARM_POSE = [0, -0.8, 0.8, 0, 0.5, 0, 0]

_ARM6 = ["waist", "shoulder", "elbow", "forearm_roll", "wrist_angle", "wrist_rotate"]
_MID7 = ["waist", "shoulder", "elbow", "forearm_roll", "wrist_1_joint", "wrist_2_joint", "wrist_3_joint"]

# NB (constants.py:45): the right arm's observed finger joint is right_right_finger while its
# actuator drives right_left_finger (joint_position_actuators.xml:17) -- preserved on purpose.
LEFT_JOINT_NAMES = [f"left_{n}" for n in _ARM6] + ["left_left_finger"]
RIGHT_JOINT_NAMES = [f"right_{n}" for n in _ARM6] + ["right_right_finger"]
MIDDLE_JOINT_NAMES = [f"middle_{n}" for n in _MID7]
LEFT_ACTUATOR_NAMES = [f"left_{n}" for n in _ARM6] + ["left_gripper"]
RIGHT_ACTUATOR_NAMES = [f"right_{n}" for n in _ARM6] + ["right_gripper"]
MIDDLE_ACTUATOR_NAMES = [f"middle_{n}" for n in _MID7]

LEFT_EEF_SITE = "left_gripper_control"
RIGHT_EEF_SITE = "right_gripper_control"
MIDDLE_EEF_SITE = "middle_zed_camera_center"
MIDDLE_BASE_LINK = "middle_base_link"
LEFT_GRIPPER_JOINT_NAMES = ["left_left_finger", "left_right_finger"]
RIGHT_GRIPPER_JOINT_NAMES = ["right_left_finger", "right_right_finger"]
