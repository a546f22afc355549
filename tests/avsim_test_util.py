import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def blob(task="slot_insertion", num_arms=3):
    return open(os.path.join(ROOT, "models", f"{task}_{num_arms}arms.avm"), "rb").read()


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
