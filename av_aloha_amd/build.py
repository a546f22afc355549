"""Build the native pieces in-tree: libavsim.so (HIP, gfx950) and, for tests/bench only, the CPU oracle."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libavsim.so")
SRC = os.path.join(HERE, "csrc")


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950: avsim_api.hip (C-ABI, f32 product kernels, IK, render) and avsim_phys_f64.hip (the f64 parity
    kernel, -ffp-contract=off so that it rounds like the oracle) compiled side by side, linked into libavsim.so."""
    srcs = [os.path.join(SRC, f) for f in sorted(os.listdir(SRC)) if not f.endswith(".o") and not f.startswith(".")] + [os.path.join(ROOT, "include", "avsim.h"), os.path.abspath(__file__)]       # (this file holds the flags: a library built with other flags is stale too)
    if not force and not _stale(LIB, srcs):
        return LIB
    # one builder at a time (pytest next to bench.py, several ranks): the objects go to fixed paths
    import fcntl
    with open(os.path.join(SRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not _stale(LIB, srcs):       # somebody else built it while we waited
            return LIB
        return _build_hip_locked(verbose)


def _build_hip_locked(verbose):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-c"]
    units = [
        # f32 divide / sqrt through v_rcp / v_rsq (~1 ulp) instead of the correctly rounded 10-instruction sequences; the f64
        # parity mode is unaffected and the f32 tolerances of tests/test_gpu_physics.py are stated against the f64 oracle
        # f32 denormals are flushed: with them on, every division and square root carries a range-scaling sequence (ten instructions
        # instead of v_rcp + multiply); nothing in the physics lives below 1e-38
        # no SLP vectoriser: packing pairs of scalar operations into v_pk_* costs register shuffles and keeps the DPP moves of the
        # cross-lane sums from folding into their adds (config 2: 972 k -> 1009 k env-steps/s, private segment 592 -> 464 B)
        # -O2, no loop vectoriser, no atomic optimiser (the LDS atomics of a wave go to distinct addresses by construction): measured
        # together 1006 k -> 1023 k (tools/exp_flags_multi.sh, profiles/r03_experiments.txt); -O3 was the setting until then
        # -amdgpu-sched-strategy=max-ilp: the machine scheduler orders for instruction-level parallelism instead of for occupancy, which the kernels fix
        # themselves (waves_per_eu): config 2 1 028 -> 1 039 k, f64 428 -> 436 k on one box (tools/exp_flags_ab.sh, profiles/r05_experiments.txt 7b)
        ("avsim_api", ["-O2", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero", "-fno-slp-vectorize", "-fno-vectorize",
                       "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-mllvm", "-amdgpu-sched-strategy=max-ilp"] + (["-DAVSIM_RENDER_STATS"] if os.environ.get("AVSIM_RENDER_STATS") else []) + os.environ.get("AVSIM_EXTRA_FLAGS", "").split()),
        # the f64 unit is built -ffp-contract=off: the oracle's roundings.  Round 6 measured the fused build (AVSIM_EXTRA_FLAGS_F64=-ffp-contract=fast): + 4 %
        # (434 -> 452 k env-steps/s) and every f64 parity test of tests/test_gpu_physics.py / test_gpu_boxbox.py / test_gpu_configs.py / test_gpu_episode_parity.py
        # unchanged -- but the scripted closed-loop episodes become other trajectories (GradIK amplifies a rounding), and on the new SlotInsertion episodes the
        # device's contact counts differ from the FULL-hull oracle's in 1.8 % of the env-steps where tests/test_gpu_fidelity.py asserts <= 1 % (0.29 % on the unfused
        # build's episodes): not adopted (profiles/r06_experiments.txt 5)
        ("avsim_phys_f64", ["-O3", "-ffp-contract=off", "-mllvm", "-amdgpu-sched-strategy=max-ilp"] + os.environ.get("AVSIM_EXTRA_FLAGS_F64", "").split()),
    ]
    procs = []
    for name, extra in units:
        cmd = common + extra + ["-o", os.path.join(SRC, name + ".o"), os.path.join(SRC, name + ".hip")]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    # linked next to the target and renamed over it: a process that has the old library mapped keeps its (old) file
    tmp = LIB + ".tmp%d" % os.getpid()
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", tmp] + [os.path.join(SRC, n + ".o") for n, _ in units]
    if verbose:
        print(" ".join(link).replace(tmp, LIB), file=sys.stderr)
    subprocess.check_call(link)
    os.replace(tmp, LIB)
    return LIB


def build_oracle(force=False):
    d = os.path.join(ROOT, "oracle")
    so = os.path.join(d, "liborc.so")
    srcs = [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".c", ".h"))]
    if force or _stale(so, srcs):
        subprocess.check_call(["make", "-C", d, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return so


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
    print(build_oracle(force="--force" in sys.argv))
