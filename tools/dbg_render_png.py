"""Debug helper: writes colour images of one env's cameras as PNG files (tools/dbg_render_png.py out_dir [task])."""
import struct
import sys
import zlib

import numpy as np


def write_png(path, img):
    h, w, _ = img.shape
    raw = b"".join(b"\x00" + img[i].tobytes() for i in range(h))
    def chunk(t, d):
        c = struct.pack(">I", len(d)) + t + d
        return c + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


if __name__ == "__main__":
    sys.path.insert(0, ".")
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from av_aloha_amd.env import make
    out = sys.argv[1]
    task = sys.argv[2] if len(sys.argv) > 2 else "SlotInsertion"
    env = make(f"gym_guided_vision/{task}-3Arms-v0", observation_height=240, observation_width=320)
    np.random.seed(0)
    obs, _ = env.reset()
    cams = list(obs["pixels"])
    rows = [np.concatenate([obs["pixels"][c] for c in cams[:3]], axis=1), np.concatenate([obs["pixels"][c] for c in cams[3:]], axis=1)]
    write_png(f"{out}/cams_{task}.png", np.concatenate(rows, axis=0))
    write_png(f"{out}/render_{task}.png", env.render())
    env.close()
