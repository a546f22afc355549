"""Edge cases of the C-ABI on the device: ragged batch sizes (partial workgroups, a single env, more envs than one launch round
holds), argument errors (status codes + avsim_last_error, no crash), capacities at their limits."""
import ctypes as C

import numpy as np
import pytest

from test_oracle_physics import OBJ, home_action, model_dict

pytestmark = pytest.mark.gpu


def make(task="slot_insertion", na=3, N=1, f64=False, **opt):
    from av_aloha_amd.sim import BatchedSim
    return BatchedSim(task, na, N, f64=f64, options=opt)


def wiggle(md, N, t):
    a = np.repeat(home_action(md)[None], N, 0)
    ph = 0.37 * np.arange(N)
    a[:, 0] += 0.2 * np.sin(0.3 * t + ph)
    a[:, 7] -= 0.2 * np.sin(0.25 * t + ph)
    a[:, 14] += 0.3 * np.sin(0.15 * t + ph)
    return a.astype(np.float32)


def run(N, T=3, **opt):
    md = model_dict("slot_insertion", 3)
    sim = make(N=N, **opt)
    sim.reset(np.repeat(OBJ[None], N, 0))
    for t in range(T):
        ap, rw, su = sim.step(wiggle(md, N, t))
    q, v, _, w = sim.get_state()
    d = sim.diag()
    sim.close()
    return q, v, w, ap, rw, d


def test_ragged_batch_sizes_give_the_same_envs():
    """Env i's result does not depend on how many envs the handle holds: 1, 3, 9 (one wave more than a workgroup of 8), 65 and 2500
    (more than the 2048 one launch round holds on the 256 CUs: the persistent workgroups come back for the rest) agree bit for bit on
    their common envs -- the waves take envs from a global counter, most expensive first, and nothing else is shared."""
    ref = run(2500)
    assert (ref[5][:, 2] == 0).all()
    for n in (1, 3, 9, 65):
        out = run(n)
        for a, b in zip(out, ref):
            assert np.array_equal(a, b[:n]), n


def test_argument_errors_are_status_codes():
    """Bad arguments come back as AVSIM_EINVAL / AVSIM_EMODEL with a message in avsim_last_error; the handle stays usable."""
    from av_aloha_amd import _ffi
    from av_aloha_amd.sim import load_blob
    L = _ffi.lib()
    blob, _ = load_blob("slot_insertion", 3)
    h = C.c_void_p()
    assert L.avsim_create(blob[:1000], 1000, 4, 0, 0, C.byref(h)) != 0 and b"" != L.avsim_last_error(None)     # truncated blob
    assert L.avsim_create(blob, len(blob), 0, 0, 0, C.byref(h)) != 0                                             # no envs
    assert L.avsim_create(blob, len(blob), 4, 99, 0, C.byref(h)) != 0                                            # no such device
    sim = make(N=4)
    hh, ok = sim.h.h, sim.h.L
    a = np.zeros((4, 21), dtype=np.float32)
    assert ok.avsim_step(hh, a.ctypes.data, -1, None, None, None) != 0 and b"avsim_step" in ok.avsim_last_error(hh)
    assert ok.avsim_set_option(hh, b"no_such_option", C.c_double(1.0)) != 0
    assert ok.avsim_set_option(hh, b"maxefc", C.c_double(3.0)) != 0                  # below the smallest record
    assert ok.avsim_set_option(hh, b"solver", C.c_double(7.0)) != 0
    ids = np.array([99], dtype=np.int32)
    out = np.empty((4, 1, 8, 8), dtype=np.float32)
    assert ok.avsim_render_depth(hh, ids.ctypes.data, 1, 8, 8, out.ctypes.data) != 0    # no such camera
    # still alive
    sim.reset(np.repeat(OBJ[None], 4, 0))
    ap, rw, su = sim.step(np.repeat(home_action(model_dict("slot_insertion", 3))[None], 4, 0).astype(np.float32))
    assert np.isfinite(ap).all()
    sim.close()


def test_capacity_limits_are_flagged_not_fatal():
    """With room for 8 contacts / 40 rows only (the resting scene needs 12 / 80) the step still runs: the excess contacts are dropped
    whole, the rows clipped, and the env's overflow flags say so (avsim_get_diag); with the default capacities no flag is set."""
    q, v, w, ap, rw, d = run(16, maxefc=40, maxcon=16)
    assert (d[:, 2] & 3 != 0).all() and (d[:, 1] <= 40).all() and np.isfinite(q).all()
    q, v, w, ap, rw, d = run(16)
    assert (d[:, 2] == 0).all() and (d[:, 0] == 12).all() and (d[:, 1] == 80).all()
