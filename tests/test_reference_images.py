"""The oracle and the HIP path against the only outputs of the real CUDA reference that exist: its README images.

tests/golden/reference_image_pins.npz holds pixel rows of /root/reference/images/*.png (made by
tests/golden/make_reference_image_pins.py).  Each image is the lossless 8-bit copy of the PPM the reference binary
printed (int(255.99f*c), src/main.cu:715-727) for one scene function of the current source at that function's own
nx, ny, ns (500 or 10000 spp) with the per-pixel seeds 1984 + pixel_index (main.cu:104).

What matches.  A pixel's samples are one sequential XORWOW stream, so two implementations agree to the last bit
until the first sample in which an ulp-level difference flips a branch that draws random numbers; from there on that
pixel's samples are statistically independent.  The oracle and the HIP path therefore follow the reference's real
build, not just its source: mul+add pairs are contracted into FMA exactly where nvcc's default -fmad=true contracts
them, and the camera basis (compile-time constants in the reference's scene kernels) is constant-folded without
contraction ("numerical contract"; every rule was checked against these images).  With that, seven of the
nine images are reproduced pixel for pixel: quads, checker, earth, perlin, simple_light (10000 spp), Cornell
(10000 spp: 359,998 of 360,000 pixels identical) and the headline random scene (10000 spp: 719,999 of 720,000
pixels identical, the other off by one level).  The two scenes whose every ray passes a constant_medium (final,
original) still diverge in most pixels -- the medium seeds a private RNG from the *bits* of the ray
(constant_medium.cuh:70-74), so a single ulp anywhere (CUDA's logf/powf are not correctly rounded) re-rolls the path --
and agree with the reference to Monte-Carlo noise (box-mean test below).  A wrong RNG stream, draw order, scene
constant, BVH rule, material or contraction site drops the exact-match rates to the noise level (see the control).
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
PINS = np.load(os.path.join(HERE, "golden", "reference_image_pins.npz"))
META = json.load(open(os.path.join(HERE, "golden", "reference_image_pins.json")))
TILE_ROWS, TILE_FIRST, TILE_STRIDE = META["tile_rows"], META["tile_first"], META["tile_stride"]


def to_8bit(fb):
    """The reference's output stage: int(255.99f * c), main.cu:722 (values above 255 clip in the PNG)."""
    return np.clip((fb * np.float32(255.99)).astype(np.int32), 0, 255)


def match(rows8, ref8):
    d = np.abs(rows8.astype(np.int32) - ref8.astype(np.int32))
    m = d.max(-1)
    return {"exact": float((m == 0).mean()), "within1": float((m <= 1).mean()), "within2": float((m <= 2).mean()),
            "mean": float(d.mean()), "max": int(d.max())}


# ---- CPU: the oracle on a few pinned rows (whole tiles for the 500-spp scenes, single rows at 10000 spp) ----

# scene: (local tile indices to render, rows per tile to render, min exact, min within1, max mean |d|)
ORACLE_CASES = {
    "quads":        ([3, 9, 14], 4, 0.9999, 1.0, 0.0005),     # measured 1.0
    "checker":      ([3, 9, 14], 4, 0.9995, 1.0, 0.001),      # 0.99993
    "earth":        ([6, 9, 12], 4, 0.9995, 1.0, 0.001),      # 1.0
    "perlin":       ([3, 9, 14], 4, 0.999, 1.0, 0.002),       # 0.99979
    "simple_light": ([6], 1, 0.999, 1.0, 0.002),              # 1.0
    "bouncing":     ([7], 1, 0.999, 1.0, 0.002),              # 1.0
    "cornell":      ([9], 1, 0.999, 1.0, 0.002),              # 1.0
}


@pytest.mark.parametrize("scene", list(ORACLE_CASES))
def test_oracle_reproduces_reference_image(art, orc, scene):
    tiles, nrows, min_exact, min_w1, max_mean = ORACLE_CASES[scene]
    info = META["images"][scene]
    nx, ny = info["nx"], info["ny"]
    img, iw, ih = art.default_texture(scene)
    o = orc.OracleScene(scene, nx, ny, img, iw, ih)
    assert (o.def_nx, o.def_ny) == (nx, ny), "the image has the size the reference host function renders"
    got, ref = [], []
    for t in tiles:
        row0 = (TILE_FIRST + t * TILE_STRIDE) * TILE_ROWS
        fb, _ = o.render(o.def_ns, row0=row0, row1=row0 + nrows)
        got.append(to_8bit(fb[row0:row0 + nrows]))
        ref.append(PINS["rows_" + scene][t * TILE_ROWS:t * TILE_ROWS + nrows])
    s = match(np.concatenate(got), np.concatenate(ref))
    print(scene, o.def_ns, "spp", s)
    assert s["exact"] >= min_exact and s["within1"] >= min_w1 and s["mean"] <= max_mean, (scene, s)


def test_wrong_seed_does_not_reproduce_the_image(art, orc):
    """Control: the same scene with another per-pixel seed base is a valid render but not the reference's pixels."""
    info = META["images"]["quads"]
    o = orc.OracleScene("quads", info["nx"], info["ny"])
    row0 = (TILE_FIRST + 9 * TILE_STRIDE) * TILE_ROWS
    fb, _ = o.render(o.def_ns, seed_base=1985, row0=row0, row1=row0 + TILE_ROWS)
    s = match(to_8bit(fb[row0:row0 + TILE_ROWS]), PINS["rows_quads"][9 * TILE_ROWS:10 * TILE_ROWS])
    assert s["exact"] < 0.9 and s["mean"] > 0.2, s


# ---- GPU: the HIP path on every pinned row of every image, at the reference's own spp ----

# scene: (min exact, min within1, max mean |d|); full-frame measurements are in profiles/r01_reference_image_match.txt
GPU_CASES = {
    "quads":        (0.9999, 1.0, 0.0005),    # measured 1.0 (90,000+ pixels)
    "checker":      (0.9999, 1.0, 0.0005),    # 0.99999
    "earth":        (0.9995, 1.0, 0.001),
    "perlin":       (0.9995, 1.0, 0.001),     # 0.99984
    "simple_light": (0.999, 1.0, 0.002),
    "bouncing":     (0.9999, 1.0, 0.0005),    # 1.0 at 10000 spp, 1.9 G rays
    "cornell":      (0.9995, 1.0, 0.001),     # 1.0 at 10000 spp, 3.1 G rays
    # constant_medium scenes: PARITY UNPINNED against the reference.  Their rows agree with the
    # reference image as an independent 10000-spp render does (profiles/r02_medium_log_ulp_experiment.txt); these bounds
    # only catch a gross regression -- the distribution itself is checked by the box-mean test below.
    "original":     (0.07, 0.28, 2.6),
    "final":        (0.06, 0.20, 3.5),
}


@pytest.mark.gpu
@pytest.mark.parametrize("scene", list(GPU_CASES))
def test_hip_path_reproduces_reference_image(gpu, scene):
    min_exact, min_w1, max_mean = GPU_CASES[scene]
    info = META["images"][scene]
    img, iw, ih = gpu.default_texture(scene)
    hs = gpu.HostScene(scene, 0, 0, img, iw, ih)
    assert (hs.nx, hs.ny) == (info["nx"], info["ny"])
    ds = gpu.DeviceScene(hs)
    try:
        fb, st = ds.render(hs.frame(tile_rows=TILE_ROWS, tile_first=TILE_FIRST, tile_stride=TILE_STRIDE))
    finally:
        ds.close()
    ref = PINS["rows_" + scene]
    assert fb.shape == ref.shape
    s = match(to_8bit(fb), ref)
    print(scene, hs.ns, "spp", st.rays, "rays", s)
    assert s["exact"] >= min_exact and s["within1"] >= min_w1 and s["mean"] <= max_mean, (scene, s)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,ns,max_mae,min_corr", [("final", 2500, 1.0, 0.9997), ("original", 2500, 1.0, 0.9997)])
def test_long_chain_scenes_agree_in_the_mean(gpu, scene, ns, max_mae, min_corr):
    """The two 10000-spp scenes with media / procedural textures diverge from the reference's sample streams in most
    pixels (see the module docstring), so their rows only agree to noise.  What must still hold is that the image is
    a sample of the same distribution: 8x8 box means of the whole frame (64 x ns samples each) against the
    reference image's box means.  Measured at 10000 spp: MAE 0.47 / 0.39 of 255, correlation 0.9999; at the 2500 spp
    rendered here 0.67 / 0.43 (profiles/r01g_gpu_tests.log), so the gate sits at 1.0 -- about 1.5 x the measured noise."""
    img, iw, ih = gpu.default_texture(scene)
    hs = gpu.HostScene(scene, 0, 0, img, iw, ih)
    ds = gpu.DeviceScene(hs)
    try:
        fb, _ = ds.render(hs.frame(ns=ns))
    finally:
        ds.close()
    a = to_8bit(fb).astype(np.float32)
    ny, nx, _ = a.shape
    box = a.reshape(ny // 8, 8, nx // 8, 8, 3).mean((1, 3))
    ref = PINS["box_" + scene]
    mae = float(np.abs(box - ref).mean())
    corr = float(np.corrcoef(box.ravel(), ref.ravel())[0, 1])
    print(scene, ns, "spp: box MAE", mae, "corr", corr)
    assert mae <= max_mae and corr >= min_corr, (mae, corr)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,texture", [("quads", None), ("earth", "earthmap.ppm")])
def test_drop_in_executable_prints_the_reference_image(gpu, scene, texture):
    """End to end through the drop-in for src/main.cu's main(): `rayTracer --scene S` with the reference host function's
    own nx, ny, ns prints the PPM whose pixels are the reference's published image (the pinned rows of it)."""
    import subprocess
    exe = os.path.join(gpu.LIB_DIR, "rayTracer")
    cmd = [exe, "--scene", scene]
    if texture:
        cmd += ["--texture", os.path.join(gpu.REPO_ROOT, "assets", texture)]
    r = subprocess.run(cmd, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()
    info = META["images"][scene]
    toks = r.stdout.decode().split()
    assert toks[:4] == ["P3", str(info["nx"]), str(info["ny"]), "255"]
    img = np.clip(np.array(toks[4:], np.int64).reshape(info["ny"], info["nx"], 3), 0, 255)   # top row first; PNG clips at 255
    fb = img[::-1]
    rows = [k for k in range(info["ny"]) if (k // TILE_ROWS) % TILE_STRIDE == TILE_FIRST]
    s = match(fb[rows], PINS["rows_" + scene])
    print(scene, s)
    assert s["exact"] >= 0.9995 and s["within1"] == 1.0, s
