"""ctypes binding of libavsim.so (include/avsim.h).  The product path has no fallback: if the HIP
library is missing or no gfx950 device is usable, this raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libavsim.so")

AVSIM_IO_DEVICE = 1
AVSIM_F64_PHYSICS = 2
IK_REFERENCE, IK_DLS = 0, 1
NDIMS = 12

_lib = None


class AvsimError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise AvsimError(f"{LIB_PATH} is missing: build it with `python -m av_aloha_amd.build` "
                             "(there is no CPU fallback for the simulation path)")
        L = C.CDLL(LIB_PATH)
        vp, i32, u32, dbl = C.c_void_p, C.c_int, C.c_uint32, C.c_double
        L.avsim_create.argtypes = [vp, C.c_size_t, i32, i32, u32, C.POINTER(vp)]
        L.avsim_destroy.argtypes = [vp]
        L.avsim_destroy.restype = None
        L.avsim_last_error.argtypes = [vp]
        L.avsim_last_error.restype = C.c_char_p
        L.avsim_dims.argtypes = [vp, vp]
        L.avsim_set_option.argtypes = [vp, C.c_char_p, dbl]
        L.avsim_reset.argtypes = [vp, vp, vp]
        L.avsim_step.argtypes = [vp, vp, i32, vp, vp, vp]
        L.avsim_step_cartesian.argtypes = [vp, vp, i32, i32, vp, vp, vp]
        L.avsim_step_ctrl.argtypes = [vp, i32, vp, vp, vp]
        L.avsim_ik.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp]
        L.avsim_fk_jac.argtypes = [vp, i32, i32, vp, vp, vp]
        L.avsim_observe.argtypes = [vp, vp, vp, vp]
        L.avsim_set_qpos.argtypes = [vp, vp]
        L.avsim_get_state.argtypes = [vp, vp, vp, vp, vp]
        L.avsim_set_state.argtypes = [vp, vp, vp, vp, vp]
        L.avsim_get_reset_poses.argtypes = [vp, vp]
        L.avsim_set_reset_poses.argtypes = [vp, vp]
        L.avsim_get_latch.argtypes = [vp, vp]
        L.avsim_set_latch.argtypes = [vp, vp]
        L.avsim_get_contacts.argtypes = [vp, vp, vp, vp]
        L.avsim_get_diag.argtypes = [vp, vp]
        L.avsim_get_phase_cycles.argtypes = [vp, vp]
        L.avsim_render_depth.argtypes = [vp, vp, i32, i32, i32, vp]
        L.avsim_render_rgb.argtypes = [vp, vp, i32, i32, i32, vp]
        L.avsim_camera_count.argtypes = [vp]
        L.avsim_load_visual.argtypes = [vp, vp, C.c_size_t]
        L.avsim_visual_info.argtypes = [vp, vp]
        L.avsim_visual_profile.argtypes = [vp, vp, i32]
        L.avsim_reward_from_pairs.argtypes = [vp, vp, i32, i32, vp, vp]
        L.avsim_sync.argtypes = [vp]
        L.avsim_set_stream.argtypes = [vp, vp]
        L.avsim_event_record.argtypes = [vp, i32]
        L.avsim_event_elapsed_ms.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
        L.avsim_kernel_time.argtypes = [vp, i32, C.POINTER(dbl), C.POINTER(C.c_int64)]
        L.avsim_render_kernel_time.argtypes = [vp, i32, C.POINTER(dbl), C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def ptr(a):
    """Raw address of a numpy array / torch tensor / int / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous()
        return a.data_ptr()
    raise TypeError(type(a))


class Handle:
    """Thin RAII wrapper; raises AvsimError with the library's message on any non-zero status."""

    def __init__(self, blob: bytes, num_envs: int, device: int = 0, flags: int = 0):
        L = lib()
        h = C.c_void_p()
        rc = L.avsim_create(blob, len(blob), num_envs, device, flags, C.byref(h))
        if rc != 0:
            raise AvsimError(f"avsim_create failed ({rc}): {L.avsim_last_error(None).decode()}")
        self.h = h
        self.L = L
        d = np.zeros(NDIMS, dtype=np.int32)
        self.check(L.avsim_dims(h, d.ctypes.data))
        (self.nq, self.nv, self.nu, self.nj, self.nobj, self.max_reward, self.num_envs, self.task_id,
         self.maxcon, self.maxefc, self.lds_bytes, self.blocks_per_cu) = [int(x) for x in d]
        self.flags = flags

    def check(self, rc):
        if rc != 0:
            raise AvsimError(f"libavsim error {rc}: {self.L.avsim_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.avsim_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
