#!/usr/bin/env python3
"""Order-aware traversal, CPU experiment (results in profiles/r03_order_experiment.txt, verdict in DESIGN.md).

Question: the kernels walk the leaves in the reference's fixed depth-first order (bvh.cuh:95-106 visits left then right, never
near-first).  For a spheres-only scene the reference's record is the minimum of (t, leaf ordinal) (sphere.cuh:66), so ANY
visiting order gives the same record.  How many box tests per ray would a near-first order save on the headline scene, over
the walk array that ships (regrouped + collapsed, 22.8 tests per ray on the device's calibration frame)?

Method: rays = every k-th ray of an oracle render of the scene (orc_ray_sample: origin, direction, time, closest t);
trees = the reference's tree and the regrouped tree (rt_regroup_leaves, bottom-up) from the host library; box tests are
counted by tools/order_experiment.c for (1) the fixed order, full tree and after the collapse DP (rt_plan_walk_array on the
counted passes), (2) a classic near-first STACK traversal of the binary tree -- the most an order can give --, (3) eight
depth-first arrays, children ordered once per direction octant, walked stacklessly like today's array, each collapsed with
its own octant's pass counts.  Every variant must reproduce the oracle's closest t for every ray (checked).
CPU only; nothing here is used by the product.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import accelerated_ray_tracer_amd as art   # noqa: E402  (host library only: scene, regrouping, collapse planner)
import oracle   # noqa: E402  (experiment tooling: the ray sample)

TMP = os.environ.get("TMPDIR", "/tmp")
EXE = os.path.join(TMP, "order_experiment")
subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", EXE, os.path.join(ROOT, "tools", "order_experiment.c"), "-lm"], check=True)

scene = sys.argv[1] if len(sys.argv) > 1 else "bouncing"
nx, ny, ns, stride = 300, 200, 8, 3
hs = art.HostScene(scene, nx, ny)
nodes = hs.nodes()
spheres = hs.spheres()
o = oracle.OracleScene(scene, nx, ny)
L = oracle.lib()
L.orc_ray_sample.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_ulonglong, C.c_void_p, C.c_int]
cap = 2_000_000
rays = np.zeros((cap, 8), np.float32)
n_rays = L.orc_ray_sample(o.h, nx, ny, ns, 0, ny, stride, rays.ctypes.data, cap)
rays = rays[:n_rays]
print(f"# {scene} {nx}x{ny} @ {ns} spp: {n_rays} sampled rays (every {stride}rd), {len(nodes)} reference nodes, {len(spheres)} spheres")


def path(name):
    return os.path.join(TMP, f"order_{name}.bin")


spheres.tofile(path("spheres"))
rays.tofile(path("rays"))
nodes.tofile(path("ref"))
os.environ["ORDER_REFERENCE_NODES"] = path("ref")


def run(mode, tree, tag, pass_out="-", octant=None):
    tree.tofile(path(tag))
    cmd = [EXE, mode, path(tag), path("spheres"), path("rays"), pass_out] + ([str(octant)] if octant is not None else [])
    out = subprocess.run(cmd, capture_output=True, text=True, check=True).stdout.strip()
    w = out.split()
    return float(w[w.index("rays,") + 1]), int(w[w.index("nodes,") + 1]), int(w[-9]), out   # tests per ray, rays used, mismatches


def collapsed(tree, tag):
    t, used, bad, _ = run("fixed", tree, tag, path(tag + "_pass"))
    passes = np.fromfile(path(tag + "_pass"), np.float64)
    walk, before, after = art.plan_walk_array(tree, passes, float(used))
    t2, _, bad2, _ = run("fixed", walk, tag + "_walk")
    return t, t2, len(walk), bad + bad2


trees = {"reference tree (bvh.cuh:29-84)": nodes, "regrouped, top-down (rt_regroup_leaves 0)": art.regroup_leaves(nodes, 0),
         "regrouped, bottom-up (rt_regroup_leaves 1)": art.regroup_leaves(nodes, 1)}
print(f"{'tree':46s} {'fixed order':>12s} {'+ collapse':>11s} {'(nodes)':>8s} {'near-first stack':>17s} {'octant arrays':>14s} {'+ collapse':>11s}")
for name, tree in trees.items():
    tag = "t" + str(abs(hash(name)) % 10000)
    full, coll, n_walk, bad = collapsed(tree, tag)
    near, _, bad_n, _ = run("near", tree, tag)
    octant, _, bad_o, _ = run("octant", tree, tag)
    # per octant: the octant's own array, collapsed on the octant's own pass counts
    tests, used_total = 0.0, 0
    for oc in range(8):
        t, used, b, _ = run("octant", tree, tag, path(tag + "_opass"), oc)
        if used == 0:
            continue
        arr = np.fromfile(path(tag + "_opass") + ".nodes", art.NODE_DTYPE)
        passes = np.fromfile(path(tag + "_opass"), np.float64)
        walk, _, _ = art.plan_walk_array(arr, passes, float(used))
        t2, used2, b2, _ = run("octant", walk, tag + "_owalk", "-", oc)
        bad_o += b + b2
        tests += t2 * used2; used_total += used2
    print(f"{name:46s} {full:12.2f} {coll:11.2f} {n_walk:8d} {near:17.2f} {octant:14.2f} {tests / max(used_total, 1):11.2f}   mismatches {bad + bad_n + bad_o}")
print("# box tests per ray; 'near-first stack' counts two child tests per interior node entered and none on a pop")
