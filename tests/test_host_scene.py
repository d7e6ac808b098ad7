"""Host logic of the product (no GPU): the reference-API mirror, BVH builder and flattener against the oracle's
independent restatement; the C ABI surface; PPM output format."""
import ctypes as C
import os
import re

import numpy as np
import pytest

SCENES = [("two_spheres", 200, 100), ("bouncing", 240, 160), ("book1", 120, 80), ("cornell", 120, 120),
          ("cornell_smoke", 120, 120), ("final", 100, 100), ("checker", 120, 60), ("earth", 120, 60), ("perlin", 120, 60),
          ("quads", 120, 60), ("simple_light", 120, 60), ("original", 80, 80), ("degenerate", 32, 16)]


@pytest.mark.parametrize("name,nx,ny", SCENES)
def test_flattened_bvh_matches_oracle(art, orc, name, nx, ny):
    hs = art.HostScene(name, nx, ny)
    ref = orc.OracleScene(name, nx, ny).nodes()
    n = hs.nodes()
    assert len(n) == len(ref)
    # boxes bit-identical, node for node, in depth-first order (bvh.cuh:29-84)
    assert np.array_equal(n["bmin"].view(np.uint32), ref[:, 0:3].view(np.uint32))
    assert np.array_equal(n["bmax"].view(np.uint32), ref[:, 3:6].view(np.uint32))
    # the same object sits in every leaf
    assert np.array_equal(hs.leaf_order(), ref[:, 6].astype(np.int32))
    # skip links: forward, inside the array, and consistent with a pre-order walk
    skip = n["skip"]
    idx = np.arange(len(n))
    assert (skip > idx).all() and (skip <= len(n)).all()
    leaf = n["prim"] >= 0
    assert (skip[leaf] == idx[leaf] + 1).all()


@pytest.mark.parametrize("name,nx,ny", SCENES)
def test_camera_matches_oracle(art, orc, name, nx, ny):
    d = art.HostScene(name, nx, ny).desc.camera
    mine = np.array(list(d.origin) + list(d.lower_left_corner) + list(d.horizontal) + list(d.vertical) + list(d.u) + list(d.v)
                    + [d.lens_radius, d.time0, d.time1], np.float32)
    assert np.array_equal(mine.view(np.uint32), orc.OracleScene(name, nx, ny).camera().view(np.uint32))


def test_bouncing_scene_contents(art):
    hs = art.HostScene("bouncing", 1200, 800)
    assert hs.desc.n_spheres == 488 and hs.desc.n_nodes == 975 and hs.desc.n_quads == 0
    m = hs.materials()
    s = hs.spheres()
    kinds = m["kind"][s["mat"]]
    # 366 lambertian, 74 metal, 17 dielectric, 31 diffuse_light
    assert [(kinds == k).sum() for k in range(4)] == [366, 74, 17, 31]
    assert (m["fuzz"] <= 1.0).all()
    moving = (s["vel"] != 0).any(axis=1)
    assert moving.sum() == 395          # every small diffuse / emissive sphere moves (main.cu:191-205)
    assert hs.use_gradient_bg == 0 and hs.gamma == pytest.approx(2.2)


def test_reference_scene_defaults(art):
    # what each reference host function passes to render<<<>>> (main.cu:656-661, 1074, 1130, 1179)
    expect = {"bouncing": (1200, 600, 10000, 0), "cornell": (600, 600, 10000, 0), "cornell_smoke": (600, 600, 1000, 0),
              "final": (800, 800, 10000, 0), "checker": (1200, 600, 500, 1), "quads": (1200, 600, 500, 1),
              "two_spheres": (200, 100, 1, 1), "random_scene": (1200, 800, 500, 0), "simple_light": (1200, 600, 10000, 0),
              "original": (800, 800, 10000, 0), "earth": (1200, 600, 500, 1), "perlin": (1200, 600, 500, 1)}
    for name, (nx, ny, ns, grad) in expect.items():
        hs = art.HostScene(name)
        assert (hs.nx, hs.ny, hs.ns, hs.use_gradient_bg) == (nx, ny, ns, grad), name


def test_unknown_scene_is_an_error(art):
    with pytest.raises(art.RtError):
        art.HostScene("no_such_scene")


def test_abi_exports_every_declared_symbol(art):
    """include/rt_abi.h <-> librt_mi355x.so: every declared function is exported (no compute call made)."""
    hdr = open(os.path.join(art.REPO_ROOT, "include", "rt_abi.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z_]+)\s*\(", hdr)) - {"rt_scene", "rt_status"}
    assert declared == set(art.RT_ABI_SYMBOLS), declared ^ set(art.RT_ABI_SYMBOLS)
    L = art.rt_lib()
    for sym in declared:
        assert hasattr(L, sym), sym
    assert L.rt_strerror(0).decode() == "ok"


def test_struct_sizes_match_header(art):
    assert art.NODE_DTYPE.itemsize == 32 and art.SPHERE_DTYPE.itemsize == 32 and art.MATERIAL_DTYPE.itemsize == 32
    assert C.sizeof(art.RtCamera) == 96 and C.sizeof(art.RtFrameDesc) == 56


def test_row_partition_covers_every_row_once(art):
    L = art.rt_lib()
    hs = art.HostScene("two_spheres")
    for ny, tile, world in [(800, 4, 8), (800, 4, 3), (101, 8, 4), (7, 4, 2), (600, 600, 1)]:
        seen = []
        for r in range(world):
            f = hs.frame(nx=16, ny=ny, ns=1, tile_rows=tile, tile_first=r, tile_stride=world)
            rows = art.local_rows_to_global(f)
            assert len(rows) == L.rt_frame_local_rows(C.byref(f))
            seen.extend(rows.tolist())
        assert sorted(seen) == list(range(ny)), (ny, tile, world)


def test_render_without_device_fails_loudly(art):
    """No CPU fallback: without a GPU the product path must raise, not quietly compute something else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(art.RtError):
        art.init(0)
    hs = art.HostScene("two_spheres", 16, 8)
    with pytest.raises(art.RtError):
        art.DeviceScene(hs)


def test_ppm_output_format(art, tmp_path):
    """main.cu:715-727: P3 header, rows top to bottom, int(255.99f*c) without clamping."""
    fb = np.zeros((2, 3, 3), np.float32)
    fb[0, 0] = [0.0, 0.5, 1.0]      # bottom-left
    fb[1, 2] = [2.0, 0.25, 0.999]   # top-right; emitters exceed 1 and are not clamped
    p = tmp_path / "o.ppm"
    art.write_ppm(str(p), fb)
    toks = p.read_text().split()
    assert toks[:4] == ["P3", "3", "2", "255"]
    px = np.array(toks[4:], int).reshape(2, 3, 3)
    assert px[1, 0].tolist() == [0, 127, 255]           # last printed row is j = 0
    assert px[0, 2].tolist() == [511, 63, 255]
    data, w, h = art.load_ppm(str(p)) if False else (None, 0, 0)   # loader takes maxval-255 files of bytes only


def test_binary_ppm_output(art, tmp_path):
    """The optional binary form: P6, the same quantisation and row order as the P3 form, one byte per
    channel -- what P3 prints above 255 (the reference does not clamp) is clamped; the texture loader reads it back."""
    fb = np.zeros((2, 3, 3), np.float32)
    fb[0, 0] = [0.0, 0.5, 1.0]
    fb[1, 2] = [2.0, 0.25, -0.5]
    p = tmp_path / "o6.ppm"
    art.write_ppm(str(p), fb, binary=True)
    raw = p.read_bytes()
    assert raw.startswith(b"P6\n3 2\n255\n") and len(raw) == len(b"P6\n3 2\n255\n") + 18
    data, w, h = art.load_ppm(str(p))
    px = data.reshape(2, 3, 3)
    assert (w, h) == (3, 2)
    assert px[1, 0].tolist() == [0, 127, 255] and px[0, 2].tolist() == [255, 63, 0]


def test_ppm_texture_loader(art, tmp_path):
    rgb = (np.arange(4 * 3 * 3) % 251).astype(np.uint8)
    p = tmp_path / "t.ppm"
    with open(p, "wb") as f:
        f.write(b"P6\n# comment\n4 3\n255\n" + rgb.tobytes())
    data, w, h = art.load_ppm(str(p))
    assert (w, h) == (4, 3) and np.array_equal(data, rgb)


@pytest.mark.parametrize("name,nx,ny,ns", [("bouncing", 96, 64, 2), ("cornell", 48, 48, 4), ("final", 40, 40, 2)])
def test_walk_array_planner(art, orc, name, nx, ny, ns):
    """rt_scene_create's planner (host only): which interior nodes of the reference's tree to drop.  With
    the oracle's own per-node pass counts as input, (1) its "before" figure is the oracle's box-test counter exactly -- the
    cost model (a node is visited as often as its parent passes) is the reference walk's; (2) every leaf survives, in
    order, with its box and object; (3) skip links still move forward; (4) the predicted tests after are what a replay of
    the model over the kept nodes gives, and fewer than before."""
    img, iw, ih = art.default_texture(name)
    hs = art.HostScene(name, nx, ny, img, iw, ih)
    nodes = hs.nodes()
    o = orc.OracleScene(name, nx, ny, img, iw, ih)
    passes, rays, box_tests = o.node_passes(ns, threads=4)
    assert len(passes) == len(nodes)
    walk, before, after = art.plan_walk_array(nodes, passes, rays)
    assert abs(before * rays - box_tests) < 0.5, (before * rays, box_tests)
    leaves = nodes[nodes["prim"] >= 0]
    wleaves = walk[walk["prim"] >= 0]
    assert len(wleaves) == len(leaves)
    for f in ("prim", "bmin", "bmax"):
        assert np.array_equal(wleaves[f], leaves[f]), f
    assert (walk["skip"] > np.arange(len(walk))).all() and walk["skip"].max() == len(walk)
    # which reference nodes were kept: they appear in the walk array in the old order
    kept = np.zeros(len(nodes), bool)
    j = 0
    for w in walk:
        while not (np.array_equal(nodes[j]["bmin"], w["bmin"]) and np.array_equal(nodes[j]["bmax"], w["bmax"]) and nodes[j]["prim"] == w["prim"]):
            j += 1
        kept[j] = True
        j += 1
    # replay the cost model: a kept node is visited as often as its nearest kept ancestor passes
    total = 0.0
    stack = []          # (end of subtree, passes of the nearest kept ancestor for nodes inside it)
    for i, n in enumerate(nodes):
        while stack and stack[-1][0] <= i:
            stack.pop()
        visits = stack[-1][1] if stack else float(rays)
        if kept[i]:
            total += visits
        if n["prim"] < 0:
            stack.append((int(n["skip"]), passes[i] if kept[i] else visits))
    assert abs(total / rays - after) < 1e-9 * max(1.0, after), (total / rays, after)
    assert after < before
    # by surface area (no counts): still every leaf, in order
    walk_sa, _, _ = art.plan_walk_array(nodes)
    assert np.array_equal(walk_sa[walk_sa["prim"] >= 0]["prim"], leaves["prim"])


@pytest.mark.parametrize("method", [0, 1])
@pytest.mark.parametrize("name,nx,ny", [("bouncing", 240, 160), ("cornell", 120, 120), ("final", 100, 100), ("two_spheres", 200, 100)])
def test_regrouped_hierarchy_keeps_what_exactness_needs(art, name, nx, ny, method):
    """rt_regroup_leaves (host only): a different hierarchy over the reference's leaves.  The walk gives the
    reference's results as long as (1) the leaves -- boxes, objects, order -- are the reference's and (2) every interior box
    contains every box below it; checked here for every node, together with the array's shape (a binary tree in depth-first
    order, skip links to the end of each subtree, interior box = exactly the union of its leaves' boxes)."""
    img, iw, ih = art.default_texture(name)
    hs = art.HostScene(name, nx, ny, img, iw, ih)
    nodes = hs.nodes()
    tree = art.regroup_leaves(nodes, method)
    leaves = nodes[nodes["prim"] >= 0]
    tleaf = tree["prim"] >= 0
    assert len(tree) == 2 * len(leaves) - 1
    for f in ("prim", "bmin", "bmax"):
        assert np.array_equal(tree[tleaf][f], leaves[f]), f
    n = len(tree)
    assert (tree["skip"] > np.arange(n)).all() and tree["skip"][0] == n
    leaf_pos = np.flatnonzero(tleaf)
    for i in range(n):
        end = int(tree["skip"][i])
        if tleaf[i]:
            assert end == i + 1
            continue
        # two children tiling the subtree
        c1 = i + 1
        c2 = int(tree["skip"][c1])
        assert c2 < end and int(tree["skip"][c2]) == end
        inside = leaf_pos[(leaf_pos > i) & (leaf_pos < end)]
        assert len(inside) >= 2
        assert np.array_equal(tree["bmin"][i], tree["bmin"][inside].min(axis=0)) and np.array_equal(tree["bmax"][i], tree["bmax"][inside].max(axis=0))
        sub = slice(i + 1, end)
        assert (tree["bmin"][sub] >= tree["bmin"][i]).all() and (tree["bmax"][sub] <= tree["bmax"][i]).all()
    # it goes through the planner like the reference's tree does (by surface area here: no device)
    walk, before, after = art.plan_walk_array(tree)
    assert np.array_equal(walk[walk["prim"] >= 0]["prim"], leaves["prim"]) and after <= before
