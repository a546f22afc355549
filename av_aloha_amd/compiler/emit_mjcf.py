"""Model blob -> self-contained MJCF: the build's OWN model restated in MuJoCo's input language.

The MuJoCo pin needs the reference's environments under real MuJoCo
(gym_guided_vision/gym_guided_vision/env.py:53-56 compiles the XML, :218 steps it).  The reference's XML and meshes do not travel to the
GPU box; the compiled blobs (models/*.avm + *.json) do.  `emit(blob, manifest)` writes everything the PHYSICS of that model consists of --
the body tree with explicit inertials, joints with their limits / armature / damping / dry friction / solver parameters, position
actuators, the two finger equalities, the contact filter's inputs (contype / conaffinity, <exclude>), geoms with their contact parameters,
mesh hulls as inline `<mesh vertex="...">` (MuJoCo collides the convex hull of a mesh's vertices [EXT], so a hull's vertex set IS the mesh),
`<option noslip_iterations cone impratio><flag multiccd>` as aloha_sim.xml:2-6, timestep 0.002 as env.py:54 -- as one XML string with no
file references.  `hulls="device"` writes the 128-vertex collision hulls the device collides, `hulls=<array dict of
models/oracle_full_hulls.avh>` the full qhull vertex sets MuJoCo itself would build from the STL files.

Nothing here reads /root/reference; tests/test_mujoco_pin.py re-reads the emitted text with compiler/mjcf.py and compares it number for
number with the blob, and -- where `import mujoco` works -- steps it next to the oracle and the device.
"""
from __future__ import annotations

import numpy as np

GEOM_NAME = {2: "sphere", 5: "cylinder", 6: "box", 7: "mesh"}
JNT_NAME = {0: "free", 2: "slide", 3: "hinge"}


def _f(v):
    """Shortest decimal text that reads back as the same double."""
    return " ".join(repr(float(x)) for x in np.atleast_1d(np.asarray(v, dtype=np.float64)).ravel())


def _mesh_names(blob, manifest):
    """One mesh asset per distinct collision hull (geom_chull address), named after the manifest's hull report where it has one."""
    adrs = sorted({int(a) for (a, n), t in zip(blob["geom_chull"], blob["geom_type"]) if t == 7})
    names = list(manifest.get("hulls", {}).keys())
    return {a: (names[i] if i < len(names) else f"hull{i}") for i, a in enumerate(adrs)}


def emit(blob, manifest, hulls="device", full_hulls=None, model_name=None):
    nb = int(blob["nbody"][0])
    ng = int(blob["ngeom"][0])
    opt = blob["opt"]
    bn, jn, an, gn = manifest["body_names"], manifest["joint_names"], manifest["actuator_names"], manifest["geom_names"]
    cn = manifest.get("camera_names", [])
    out = []
    w = out.append
    w(f'<mujoco model="{model_name or "avsim_" + manifest["task"] + "_" + str(manifest["num_arms"]) + "arms"}">')
    # inertiafromgeom="false": every inertia below is explicit (bodies the reference leaves to its geoms -- the task objects -- were
    # integrated by the compiler, compile.py "body inertias"); static bodies stay massless
    w('  <compiler angle="radian" autolimits="true" inertiafromgeom="false"/>')
    cone = "elliptic" if opt[6] != 0 else "pyramidal"
    w(f'  <option timestep="{_f(opt[0])}" gravity="{_f(opt[1:4])}" impratio="{_f(opt[4])}" noslip_iterations="{int(opt[5])}" cone="{cone}">')
    w(f'    <flag multiccd="{"enable" if int(blob["opt_multiccd"][0]) else "disable"}"/>')
    w('  </option>')

    # ---- mesh assets: the hulls' vertex sets --------------------------------------------------------------------------------------
    mesh_of = _mesh_names(blob, manifest)
    w('  <asset>')
    for adr, name in mesh_of.items():
        n = next(int(c[1]) for c, t in zip(blob["geom_chull"], blob["geom_type"]) if t == 7 and int(c[0]) == adr)
        v = blob["chull_vert"][adr:adr + n]
        if hulls != "device":
            assert full_hulls is not None and "mesh_names" in full_hulls, "hulls='full' needs models/oracle_full_hulls.{avh,json}"
            i = full_hulls["mesh_names"].index(name)
            a0, cnt = int(full_hulls["full_adr"][i]), int(full_hulls["full_num"][i])
            v = full_hulls["full_vert"][a0:a0 + cnt]
        w(f'    <mesh name="{name}" vertex="{_f(v)}"/>')
    w('  </asset>')

    # ---- bodies ---------------------------------------------------------------------------------------------------------------------
    children = {i: [] for i in range(nb)}
    for i in range(1, nb):
        children[int(blob["body_parent"][i])].append(i)
    geoms_of = {i: [] for i in range(nb)}
    for k in range(ng):
        geoms_of[int(blob["geom_body"][k])].append(k)
    sites_of = {i: [] for i in range(nb)}
    for k, b in enumerate(blob["site_body"]):
        sites_of[int(b)].append(k)
    sn = manifest.get("site_names", [])
    cams_of = {i: [] for i in range(nb)}
    for k, b in enumerate(blob.get("cam_body", [])):
        cams_of[int(b)].append(k)

    def geom_xml(k, ind):
        t = int(blob["geom_type"][k])
        a = ([f'name="{gn[k]}"'] if gn[k] else []) + [f'type="{GEOM_NAME[t]}"', f'pos="{_f(blob["geom_pos"][k])}"', f'quat="{_f(blob["geom_quat"][k])}"']      # (most collision geoms of the assets are unnamed)
        if t == 7:
            a.append(f'mesh="{mesh_of[int(blob["geom_chull"][k][0])]}"')
        else:
            a.append(f'size="{_f(blob["geom_size"][k][:{2: 1, 5: 2, 6: 3}[t]])}"')
        a += [f'contype="{int(blob["geom_contype"][k])}"', f'conaffinity="{int(blob["geom_conaffinity"][k])}"',
              f'condim="{int(blob["geom_condim"][k])}"', f'priority="{int(blob["geom_priority"][k])}"',
              f'friction="{_f(blob["geom_friction"][k])}"', f'solref="{_f(blob["geom_solref"][k])}"', f'solimp="{_f(blob["geom_solimp"][k])}"',
              f'solmix="{_f(blob["geom_solmix"][k])}"', f'margin="{_f(blob["geom_margin"][k])}"', f'gap="{_f(blob["geom_gap"][k])}"',
              f'rgba="{_f(blob["geom_rgba"][k])}"']
        w(" " * ind + "<geom " + " ".join(a) + "/>")

    def body_xml(i, ind):
        sp = " " * ind
        if i > 0:
            w(f'{sp}<body name="{bn[i]}" pos="{_f(blob["body_pos"][i])}" quat="{_f(blob["body_quat"][i])}">')
            ind += 2
            sp = " " * ind
            if blob["body_mass"][i] > 0:
                w(f'{sp}<inertial pos="{_f(blob["body_ipos"][i])}" mass="{_f(blob["body_mass"][i])}" fullinertia="{_f(blob["body_inertia"][i])}"/>')
            for j in range(int(blob["body_jntadr"][i]), int(blob["body_jntadr"][i]) + int(blob["body_jntnum"][i])) if blob["body_jntnum"][i] else ():
                t = int(blob["jnt_type"][j])
                d = int(blob["jnt_dofadr"][j])
                a = [f'name="{jn[j]}"', f'type="{JNT_NAME[t]}"', f'armature="{_f(blob["dof_armature"][d])}"', f'damping="{_f(blob["dof_damping"][d])}"',
                     f'frictionloss="{_f(blob["dof_frictionloss"][d])}"']
                if t != 0:
                    a += [f'pos="{_f(blob["jnt_pos"][j])}"', f'axis="{_f(blob["jnt_axis"][j])}"', f'limited="{"true" if blob["jnt_limited"][j] else "false"}"',
                          f'margin="{_f(blob["jnt_margin"][j])}"', f'solreflimit="{_f(blob["jnt_solref"][j])}"', f'solimplimit="{_f(blob["jnt_solimp"][j])}"',
                          f'solreffriction="{_f(blob["dof_solref"][d])}"', f'solimpfriction="{_f(blob["dof_solimp"][d])}"']
                    if blob["jnt_limited"][j]:
                        a.append(f'range="{_f(blob["jnt_range"][j])}"')
                    if blob["jnt_actfrclimited"][j]:
                        a += ['actuatorfrclimited="true"', f'actuatorfrcrange="{_f(blob["jnt_actfrcrange"][j])}"']
                w(f'{sp}<joint ' + " ".join(a) + "/>")
        for k in sites_of[i]:
            w(f'{sp}<site name="{sn[k]}" pos="{_f(blob["site_pos"][k])}" quat="{_f(blob["site_quat"][k])}"/>')
        for k in cams_of[i]:
            w(f'{sp}<camera name="{cn[k]}" pos="{_f(blob["cam_pos"][k])}" quat="{_f(blob["cam_quat"][k])}" fovy="{_f(blob["cam_fovy"][k])}"/>')
        # geoms and child bodies in the order that gives the blob's geom numbering back when the text is read in document order
        # (a body's geoms and child bodies may interleave in the source; MuJoCo itself numbers geoms body by body [EXT])
        items = order_items(i)
        if i > 0:
            for kind, k in items:
                geom_xml(k, ind) if kind == "g" else body_xml(k, ind)
            w(" " * (ind - 2) + "</body>")
        return items

    def first_geom(c):
        g = [k for k in geoms_of[c]] + [x for ch in children[c] for x in [first_geom(ch)] if x is not None]
        return min(g) if g else None

    def order_items(i):
        keyed, nxt = [], float("inf")
        for c in reversed(children[i]):
            g = first_geom(c)
            key = (g - 0.5) if g is not None else (nxt - 1e-3 if nxt != float("inf") else 1e9 + c)
            nxt = key
            keyed.append((key, "b", c))
        keyed += [(float(k), "g", k) for k in geoms_of[i]]
        return [(kind, k) for _, kind, k in sorted(keyed, key=lambda t: t[0])]

    # the world's own geoms follow the bodies of their <worldbody> section (compiler/mjcf.py reads a section's bodies first): one section
    # per run of bodies followed by world geoms
    items = order_items(0)
    w('  <worldbody>')
    body_xml(0, 4)
    prev = "b"
    for kind, k in items:
        if kind == "b" and prev == "g":
            w('  </worldbody>')
            w('  <worldbody>')
        geom_xml(k, 4) if kind == "g" else body_xml(k, 4)
        prev = kind
    w('  </worldbody>')

    # ---- actuators, equalities, excludes -----------------------------------------------------------------------------------------------
    dof_jnt = blob["dof_jnt"]
    w('  <actuator>')
    for k in range(int(blob["nu"][0])):
        j = int(dof_jnt[int(blob["act_dof"][k])])
        a = [f'name="{an[k]}"', f'joint="{jn[j]}"', f'kp="{_f(blob["act_kp"][k])}"', f'kv="{_f(blob["act_kv"][k])}"', f'gear="{_f(blob["act_gear"][k])}"',
             f'ctrllimited="{"true" if blob["act_ctrllimited"][k] else "false"}"']
        if blob["act_ctrllimited"][k]:
            a.append(f'ctrlrange="{_f(blob["act_ctrlrange"][k])}"')
        w('    <position ' + " ".join(a) + "/>")
    w('  </actuator>')
    if int(blob["neq"][0]):
        w('  <equality>')
        for k in range(int(blob["neq"][0])):
            w(f'    <joint joint1="{jn[int(dof_jnt[int(blob["eq_dof1"][k])])]}" joint2="{jn[int(dof_jnt[int(blob["eq_dof2"][k])])]}" '
              f'polycoef="{_f(blob["eq_polycoef"][k])}" solref="{_f(blob["eq_solref"][k])}" solimp="{_f(blob["eq_solimp"][k])}"/>')
        w('  </equality>')
    if len(blob["exclude_body"]):
        w('  <contact>')
        for a, b in blob["exclude_body"]:
            w(f'    <exclude body1="{bn[int(a)]}" body2="{bn[int(b)]}"/>')
        w('  </contact>')
    w('</mujoco>')
    return "\n".join(out) + "\n"


def emit_files(model_dir, task, num_arms, prefix="", hulls="device"):
    """MJCF text of models/<prefix><task>_<n>arms.{avm,json}."""
    import json
    import os
    from .compile import read_blob
    base = os.path.join(model_dir, f"{prefix}{task}_{num_arms}arms")
    blob = read_blob(base + ".avm")
    man = json.load(open(base + ".json"))
    full = None
    if hulls != "device":
        full = read_blob(os.path.join(model_dir, "oracle_full_hulls.avh"))
        full["mesh_names"] = json.load(open(os.path.join(model_dir, "oracle_full_hulls.json")))["mesh_names"]
    return emit(blob, man, hulls=hulls, full_hulls=full)


if __name__ == "__main__":
    import argparse
    import os
    ap = argparse.ArgumentParser(description="write the MJCF restatement of a compiled model (for the MuJoCo pin; not committed)")
    ap.add_argument("--models", default=os.path.join(os.path.dirname(__file__), "..", "..", "models"))
    ap.add_argument("--task", default="slot_insertion")
    ap.add_argument("--arms", type=int, default=3)
    ap.add_argument("--hulls", choices=["device", "full"], default="device")
    ap.add_argument("--out", default="-")
    args = ap.parse_args()
    txt = emit_files(args.models, args.task, args.arms, hulls=args.hulls)
    if args.out == "-":
        print(txt, end="")
    else:
        open(args.out, "w").write(txt)
