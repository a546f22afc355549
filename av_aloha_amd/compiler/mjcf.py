"""Minimal MJCF reader for the subset of MuJoCo XML the AV-ALOHA scenes use.

This is an *offline* tool: it runs where the reference assets are present and
emits a compact model blob (see ``compile.py``).  Nothing in here is needed at
run time on the GPU box.

Subset handled (reference: gym_guided_vision/gym_guided_vision/assets/
aloha_sim.xml:1-382, scene.xml:1-93, joint_position_actuators.xml:1-40,
task_*.xml): ``<include>``, nested ``<default class>`` with ``childclass``,
``compiler angle=radian autolimits=true meshdir``, ``option``, bodies with
``pos/quat/euler``, explicit ``<inertial>``, ``joint`` (hinge/slide/free),
``geom`` (box/sphere/cylinder/mesh), ``site``, ``camera``, ``position``
actuators, ``equality/joint``, ``contact/exclude``, mesh assets with ``scale``.
"""
from __future__ import annotations

import os
import xml.etree.ElementTree as ET

import numpy as np


def _floats(s, n=None):
    v = np.array([float(x) for x in s.split()], dtype=np.float64)
    if n is not None and len(v) != n:
        raise ValueError(f"expected {n} floats, got {s!r}")
    return v


def quat_mul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([
        aw * bw - ax * bx - ay * by - az * bz,
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by - ax * bz + ay * bw + az * bx,
        aw * bz + ax * by - ay * bx + az * bw,
    ])


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ])


def mat_to_quat(R):
    # Shepperd's method, returns wxyz with w >= 0
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    if q[0] < 0:
        q = -q
    return q / np.linalg.norm(q)


def axis_angle_quat(axis, ang):
    axis = np.asarray(axis, dtype=np.float64)
    return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])


def euler_to_quat(e, seq="xyz"):
    """MuJoCo semantics: lower-case letters rotate about the moving frame's
    axes (post-multiply), upper-case about the fixed frame (pre-multiply)."""
    q = np.array([1.0, 0, 0, 0])
    for ang, c in zip(e, seq):
        ax = np.zeros(3)
        ax["xyz".index(c.lower())] = 1.0
        r = axis_angle_quat(ax, ang)
        q = quat_mul(q, r) if c.islower() else quat_mul(r, q)
    return q


def orientation(attrib, eulerseq="xyz"):
    """Return wxyz quaternion from quat / euler / xyaxes attributes."""
    if "quat" in attrib:
        q = _floats(attrib["quat"], 4)
        return q / np.linalg.norm(q)
    if "euler" in attrib:
        return euler_to_quat(_floats(attrib["euler"], 3), eulerseq)
    if "xyaxes" in attrib:
        v = _floats(attrib["xyaxes"], 6)
        x = v[:3] / np.linalg.norm(v[:3])
        y = v[3:] - x * np.dot(x, v[3:])
        y /= np.linalg.norm(y)
        z = np.cross(x, y)
        return mat_to_quat(np.stack([x, y, z], axis=1))
    return np.array([1.0, 0, 0, 0])


class Defaults:
    """Tree of <default class=...> nodes; attribute lookup walks to the root."""

    def __init__(self):
        self.classes = {"main": {"parent": None, "elems": {}}}

    def add(self, node, parent="main"):
        name = node.attrib.get("class", "main")
        if name not in self.classes:
            self.classes[name] = {"parent": parent if name != "main" else None, "elems": {}}
        cls = self.classes[name]
        for child in node:
            if child.tag == "default":
                self.add(child, parent=name)
            else:
                tag = child.tag
                cls["elems"].setdefault(tag, {}).update(child.attrib)

    def resolve(self, tag, cls):
        """Merged attribute dict for element type `tag` under default class `cls`.
        Actuator shortcuts (position) inherit from both their own tag and 'general'."""
        chain = []
        c = cls or "main"
        while c is not None:
            chain.append(c)
            c = self.classes[c]["parent"]
        out = {}
        for c in reversed(chain):
            out.update(self.classes[c]["elems"].get(tag, {}))
        return out


def load_xml_with_includes(path):
    """Parse `path`, recursively splicing <include file=...> children in place
    (paths relative to the including file, as MuJoCo does)."""
    tree = ET.parse(path)
    root = tree.getroot()
    base = os.path.dirname(os.path.abspath(path))

    def splice(node):
        i = 0
        while i < len(node):
            ch = node[i]
            if ch.tag == "include":
                sub = load_xml_with_includes(os.path.join(base, ch.attrib["file"]))
                node.remove(ch)
                for k, g in enumerate(list(sub)):
                    node.insert(i + k, g)
                i += len(list(sub))
            else:
                splice(ch)
                i += 1

    splice(root)
    return root


class Model:
    """Plain-python description produced by parse(); consumed by compile.py."""

    def __init__(self):
        self.option = {}
        self.compiler = {}
        self.meshes = {}      # name -> dict(file, scale)
        self.materials = {}   # name -> rgba (4,) or None for a textured material
        self.bodies = []      # list of dict, index 0 = world
        self.joints = []
        self.geoms = []
        self.sites = []
        self.cameras = []
        self.actuators = []
        self.equalities = []
        self.excludes = []


def parse(path):
    root = load_xml_with_includes(path)
    m = Model()
    defaults = Defaults()
    meshdir = ""
    for node in root:
        if node.tag == "compiler":
            m.compiler.update(node.attrib)
        elif node.tag == "option":
            m.option.update(node.attrib)
            for f in node.findall("flag"):
                m.option.update({"flag_" + k: v for k, v in f.attrib.items()})
        elif node.tag == "default":
            defaults.add(node)
    if m.compiler.get("angle", "degree") != "radian":
        raise NotImplementedError("only angle=radian supported")
    meshdir = os.path.join(os.path.dirname(os.path.abspath(path)), m.compiler.get("meshdir", ""))
    eulerseq = m.compiler.get("eulerseq", "xyz")
    autolimits = m.compiler.get("autolimits", "true") == "true"

    for node in root.findall("asset"):
        for me in node.findall("mesh"):
            scale = _floats(me.attrib.get("scale", "1 1 1"), 3)
            if "vertex" in me.attrib:        # inline vertex set (compiler/emit_mjcf.py writes the collision hulls that way)
                m.meshes[me.attrib["name"]] = {"file": None, "scale": scale, "vertex": _floats(me.attrib["vertex"]).reshape(-1, 3)}
                continue
            f = me.attrib["file"]
            name = me.attrib.get("name", os.path.splitext(os.path.basename(f))[0])
            m.meshes[name] = {"file": os.path.join(meshdir, f), "scale": scale}
        for ma in node.findall("material"):
            m.materials[ma.attrib["name"]] = _floats(ma.attrib["rgba"], 4) if "rgba" in ma.attrib else None

    def elem_attrs(tag, node, childclass):
        cls = node.attrib.get("class", childclass)
        a = dict(defaults.resolve(tag, cls))
        a.update(node.attrib)
        return a

    def add_body(node, parent, childclass):
        cc = node.attrib.get("childclass", childclass)
        bid = len(m.bodies)
        body = {
            "name": node.attrib.get("name", f"body{bid}"),
            "parent": parent,
            "pos": _floats(node.attrib.get("pos", "0 0 0"), 3),
            "quat": orientation(node.attrib, eulerseq),
            "inertial": None,
            "joints": [], "geoms": [],
        }
        m.bodies.append(body)
        for ch in node:
            if ch.tag == "inertial":
                body["inertial"] = {
                    "pos": _floats(ch.attrib["pos"], 3),
                    "quat": orientation(ch.attrib, eulerseq),
                    "mass": float(ch.attrib["mass"]),
                }
                if "fullinertia" in ch.attrib:   # Ixx Iyy Izz Ixy Ixz Iyz in the frame at `pos` aligned with the body (emit_mjcf.py)
                    body["inertial"]["fullinertia"] = _floats(ch.attrib["fullinertia"], 6)
                else:
                    body["inertial"]["diaginertia"] = _floats(ch.attrib["diaginertia"], 3)
            elif ch.tag in ("joint", "freejoint"):
                a = elem_attrs("joint", ch, cc)
                jtype = "free" if ch.tag == "freejoint" else a.get("type", "hinge")
                j = {
                    "name": a.get("name", f"joint{len(m.joints)}"),
                    "type": jtype, "body": bid,
                    "pos": _floats(a.get("pos", "0 0 0"), 3),
                    "axis": _floats(a.get("axis", "0 0 1"), 3),
                    "armature": float(a.get("armature", 0)),
                    "damping": float(a.get("damping", 0)),
                    "frictionloss": float(a.get("frictionloss", 0)),
                    "stiffness": float(a.get("stiffness", 0)),
                    "solreflimit": _floats(a.get("solreflimit", "0.02 1"), 2),
                    "solimplimit": _floats(a.get("solimplimit", "0.9 0.95 0.001 0.5 2"), 5),
                    "solreffriction": _floats(a.get("solreffriction", "0.02 1"), 2),
                    "solimpfriction": _floats(a.get("solimpfriction", "0.9 0.95 0.001 0.5 2"), 5),
                    "margin": float(a.get("margin", 0)),
                }
                if j["stiffness"] != 0:
                    raise NotImplementedError("joint stiffness")
                if "range" in a and jtype in ("hinge", "slide"):
                    lim = a.get("limited", "auto")
                    j["limited"] = (lim == "true") or (lim == "auto" and autolimits)
                    j["range"] = _floats(a["range"], 2)
                else:
                    j["limited"] = False
                    j["range"] = np.zeros(2)
                if "actuatorfrcrange" in a:
                    lim = a.get("actuatorfrclimited", "auto")
                    j["actfrclimited"] = (lim == "true") or (lim == "auto" and autolimits)
                    j["actfrcrange"] = _floats(a["actuatorfrcrange"], 2)
                else:
                    j["actfrclimited"] = False
                    j["actfrcrange"] = np.zeros(2)
                n = np.linalg.norm(j["axis"])
                if n > 0:
                    j["axis"] = j["axis"] / n
                body["joints"].append(len(m.joints))
                m.joints.append(j)
            elif ch.tag == "geom":
                a = elem_attrs("geom", ch, cc)
                gtype = a.get("type", "sphere")
                g = {
                    "name": a.get("name", ""),
                    "type": gtype, "body": bid,
                    "pos": _floats(a.get("pos", "0 0 0"), 3),
                    "quat": orientation(a, eulerseq),
                    "size": _floats(a.get("size", "0 0 0")),
                    "mesh": a.get("mesh"),
                    "contype": int(a.get("contype", 1)),
                    "conaffinity": int(a.get("conaffinity", 1)),
                    "condim": int(a.get("condim", 3)),
                    "group": int(a.get("group", 0)),
                    "priority": int(a.get("priority", 0)),
                    "friction": _floats(a.get("friction", "1 0.005 0.0001")),
                    "solref": _floats(a.get("solref", "0.02 1"), 2),
                    "solimp": _floats(a.get("solimp", "0.9 0.95 0.001 0.5 2"), 5),
                    "solmix": float(a.get("solmix", 1)),
                    "margin": float(a.get("margin", 0)),
                    "gap": float(a.get("gap", 0)),
                    "mass": float(a["mass"]) if "mass" in a else None,
                    "density": float(a.get("density", 1000)),
                    "rgba": _floats(a.get("rgba", "0.5 0.5 0.5 1"), 4),
                    "rgba_given": "rgba" in a, "material": a.get("material"),
                }
                # friction may be given with fewer than 3 numbers: pad with defaults
                fr = np.array([1.0, 0.005, 0.0001])
                fr[:len(g["friction"])] = g["friction"]
                g["friction"] = fr
                body["geoms"].append(len(m.geoms))
                m.geoms.append(g)
            elif ch.tag == "site":
                a = elem_attrs("site", ch, cc)
                m.sites.append({
                    "name": a.get("name", ""), "body": bid,
                    "pos": _floats(a.get("pos", "0 0 0"), 3),
                    "quat": orientation(a, eulerseq),
                })
            elif ch.tag == "camera":
                a = elem_attrs("camera", ch, cc)
                m.cameras.append({
                    "name": a.get("name", ""), "body": bid,
                    "pos": _floats(a.get("pos", "0 0 0"), 3),
                    "quat": orientation(a, eulerseq),
                    "fovy": float(a.get("fovy", 45)),
                })
            elif ch.tag == "body":
                add_body(ch, bid, cc)
            elif ch.tag == "light":
                pass
            else:
                raise NotImplementedError(f"unsupported body child <{ch.tag}>")
        return bid

    # world body = index 0; all <worldbody> sections are merged into it
    world = {"name": "world", "parent": -1, "pos": np.zeros(3), "quat": np.array([1.0, 0, 0, 0]),
             "inertial": None, "joints": [], "geoms": []}
    m.bodies.append(world)
    for wb in root.findall("worldbody"):
        fake = ET.Element("body")
        for ch in wb:
            if ch.tag == "body":
                add_body(ch, 0, None)
            elif ch.tag in ("geom", "site", "camera", "light"):
                fake.append(ch)
        # world-level geoms/sites/cameras: run them through the same code path
        for ch in fake:
            if ch.tag == "geom":
                a = elem_attrs("geom", ch, None)
                g = {
                    "name": a.get("name", ""), "type": a.get("type", "sphere"), "body": 0,
                    "pos": _floats(a.get("pos", "0 0 0"), 3), "quat": orientation(a, eulerseq),
                    "size": _floats(a.get("size", "0 0 0")), "mesh": a.get("mesh"),
                    "contype": int(a.get("contype", 1)), "conaffinity": int(a.get("conaffinity", 1)),
                    "condim": int(a.get("condim", 3)), "group": int(a.get("group", 0)),
                    "priority": int(a.get("priority", 0)),
                    "friction": _floats(a.get("friction", "1 0.005 0.0001")),
                    "solref": _floats(a.get("solref", "0.02 1"), 2),
                    "solimp": _floats(a.get("solimp", "0.9 0.95 0.001 0.5 2"), 5),
                    "solmix": float(a.get("solmix", 1)), "margin": float(a.get("margin", 0)),
                    "gap": float(a.get("gap", 0)),
                    "mass": float(a["mass"]) if "mass" in a else None,
                    "density": float(a.get("density", 1000)),
                    "rgba": _floats(a.get("rgba", "0.5 0.5 0.5 1"), 4),
                    "rgba_given": "rgba" in a, "material": a.get("material"),
                }
                fr = np.array([1.0, 0.005, 0.0001])
                fr[:len(g["friction"])] = g["friction"]
                g["friction"] = fr
                world["geoms"].append(len(m.geoms))
                m.geoms.append(g)
            elif ch.tag == "site":
                a = elem_attrs("site", ch, None)
                m.sites.append({"name": a.get("name", ""), "body": 0,
                                "pos": _floats(a.get("pos", "0 0 0"), 3), "quat": orientation(a, eulerseq)})
            elif ch.tag == "camera":
                a = elem_attrs("camera", ch, None)
                m.cameras.append({"name": a.get("name", ""), "body": 0,
                                  "pos": _floats(a.get("pos", "0 0 0"), 3), "quat": orientation(a, eulerseq),
                                  "fovy": float(a.get("fovy", 45))})

    for node in root.findall("actuator"):
        for ch in node:
            if ch.tag != "position":
                raise NotImplementedError(f"actuator <{ch.tag}>")
            cls = ch.attrib.get("class")
            a = dict(defaults.resolve("general", cls))
            a.update(defaults.resolve("position", cls))
            a.update(ch.attrib)
            act = {
                "name": a.get("name", ""), "joint": a["joint"],
                "kp": float(a.get("kp", 1)), "kv": float(a.get("kv", 0)),
                "gear": _floats(a.get("gear", "1"))[0],
            }
            if "ctrlrange" in a:
                lim = a.get("ctrllimited", "auto")
                act["ctrllimited"] = (lim == "true") or (lim == "auto" and autolimits)
                act["ctrlrange"] = _floats(a["ctrlrange"], 2)
            else:
                act["ctrllimited"] = False
                act["ctrlrange"] = np.zeros(2)
            if "forcerange" in a:
                raise NotImplementedError("actuator forcerange")
            m.actuators.append(act)

    for node in root.findall("equality"):
        for ch in node:
            if ch.tag != "joint":
                raise NotImplementedError(f"equality <{ch.tag}>")
            a = dict(defaults.resolve("equality", ch.attrib.get("class")))
            a.update(ch.attrib)
            m.equalities.append({
                "joint1": a["joint1"], "joint2": a.get("joint2"),
                "polycoef": _floats(a.get("polycoef", "0 1 0 0 0"), 5),
                "solref": _floats(a.get("solref", "0.02 1"), 2),
                "solimp": _floats(a.get("solimp", "0.9 0.95 0.001 0.5 2"), 5),
            })

    for node in root.findall("contact"):
        for ch in node:
            if ch.tag == "exclude":
                m.excludes.append((ch.attrib["body1"], ch.attrib["body2"]))
            else:
                raise NotImplementedError(f"contact <{ch.tag}>")
    return m
