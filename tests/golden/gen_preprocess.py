"""Golden vectors of eval_scripts/eval.py:23-66 preprocess_observation, produced by importing the reference's file under stubs
for what the build image lacks (torchvision, lerobot, imageio, gymnasium, the real-robot modules).  The images are 480 x 640, the
size the gym environments produce (gym_guided_vision/__init__.py:6-19): torchvision's Resize((480, 640)) returns such an image
unchanged, so the stub's Resize (identity on that size, an error otherwise) decides nothing about the result.  Runs ONLY in the
build container (needs /root/reference); writes tests/golden/preprocess_observation.npz.

    python tests/golden/gen_preprocess.py
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def install_stubs():
    d = tempfile.mkdtemp(prefix="evalshim_")

    def mod(path, body=""):
        full = os.path.join(d, path)
        os.makedirs(os.path.dirname(full), exist_ok=True)
        open(full, "w").write(body)
    mod("torchvision/__init__.py")
    mod("torchvision/transforms.py",
        "class Resize:\n"
        "    def __init__(self, size): self.size = tuple(size)\n"
        "    def __call__(self, img):\n"
        "        assert tuple(img.shape[-2:]) == self.size, 'stub Resize: only the identity case is pinned'\n"
        "        return img\n")
    mod("imageio.py")
    mod("lerobot/__init__.py"); mod("lerobot/common/__init__.py"); mod("lerobot/common/policies/__init__.py")
    mod("lerobot/common/policies/act/__init__.py"); mod("lerobot/common/policies/act/modeling_act.py", "class ACTPolicy: pass\n")
    mod("lerobot/common/envs/__init__.py"); mod("lerobot/common/envs/utils.py", "def preprocess_observation(o): raise NotImplementedError\n")
    mod("gymnasium/__init__.py", "def make(*a, **k): raise NotImplementedError\n")
    mod("gym_guided_vision/__init__.py"); mod("gym_guided_vision/constants.py")
    mod("real_env.py", "class RealEnv: pass\n")
    mod("constants.py", "REAL_DT = 0.04\n")
    sys.path.insert(0, d)


def main():
    install_stubs()
    spec = importlib.util.spec_from_file_location("ref_eval", f"{REF}/eval_scripts/eval.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    rng = np.random.default_rng(20241022)
    cams = ["zed_cam_left", "zed_cam_right", "wrist_cam_left"]
    obs = {"pixels": {c: rng.integers(0, 256, size=(480, 640, 3), dtype=np.uint8) for c in cams},
           "agent_pos": rng.normal(size=21)}
    out = ref.preprocess_observation(obs)
    single = ref.preprocess_observation({"pixels": obs["pixels"]["zed_cam_left"], "agent_pos": obs["agent_pos"][:14]})
    # the inputs are regenerated from the seed by the test (default_rng(20241022): three 480 x 640 x 3 uint8 images, then 21
    # normals); of the outputs a 48 x 64 corner and the exact float64 sum of every image are kept (the function is elementwise)
    save = {"agent_pos": obs["agent_pos"], "keys": np.array(sorted(out)), "single_keys": np.array(sorted(single))}
    for k, v in list(out.items()) + [("single." + k, v) for k, v in single.items()]:
        a = v.numpy()
        save["shape_" + k] = np.array(a.shape)
        if a.ndim == 4:
            save["corner_" + k] = a[:, :, :48, :64].copy()
            save["sum_" + k] = np.array(a.astype(np.float64).sum())
        else:
            save["out_" + k] = a
    np.savez_compressed(os.path.join(HERE, "preprocess_observation.npz"), **save)
    print({k: (v.shape, v.dtype) for k, v in save.items()})


if __name__ == "__main__":
    main()
