#!/usr/bin/env python3
"""constant_medium parity: why the two reference images whose every ray passes a medium (finalScene.png,
alfredo2.png) agree with this implementation only to Monte-Carlo noise.

The medium seeds a private XORWOW from the BITS of the ray (constant_medium.cuh:70-74) and scatters at
-1/density * logf(U) (constant_medium.cuh:53-55).  CUDA's logf is not correctly rounded; here log is evaluated in double
and rounded once.  Claim: a single ulp in a fraction of the log results is enough to re-roll almost every pixel of a
10000-spp frame, so the match rate against the reference image says nothing about the rest of the path.

Experiment (CPU oracle only, one pinned 4-row tile of the final scene at the reference's own 10000 spp): render the tile
(A) as is and (B) with ORC_LOG_ULP=N, i.e. the log result moved by one ulp for one input in N.  Compare A with B and both
with the reference image's pixels.  Run each variant in its own process (the switch is read at load time):
    python tools/medium_ulp_experiment.py render A 0;  python tools/medium_ulp_experiment.py render B 16;  ... compare
"""
import json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", os.environ.get("OUT_DIR", "medium_ulp"))
TILE = 11            # local tile index into the pinned rows of tests/golden/reference_image_pins.npz
NS = int(os.environ.get("NS", "10000"))


def to8(fb):
    return np.clip((fb * np.float32(255.99)).astype(np.int32), 0, 255)


def stats(a, b):
    d = np.abs(a.astype(int) - b.astype(int)); m = d.max(-1)
    return f"exact {100 * (m == 0).mean():6.2f} %  within1 {100 * (m <= 1).mean():6.2f} %  mean|d| {d.mean():.3f}"


if sys.argv[1] == "render":
    import oracle, accelerated_ray_tracer_amd as art
    # the perturbable logf only exists in the diagnostic twin of the oracle (-DORC_DIAG); load that one, explicitly
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "librt_oracle_diag.so"], check=True, capture_output=True)
    oracle.LIB_PATH = os.path.join(ROOT, "oracle", "librt_oracle_diag.so")
    meta = json.load(open(os.path.join(ROOT, "tests/golden/reference_image_pins.json")))
    tr, tf, ts = meta["tile_rows"], meta["tile_first"], meta["tile_stride"]
    row0 = (tf + TILE * ts) * tr
    img, iw, ih = art.default_texture("final")
    o = oracle.OracleScene("final", 800, 800, img, iw, ih)
    fb, cnt = o.render(NS, row0=row0, row1=row0 + tr, threads=4)
    os.makedirs(OUT, exist_ok=True)
    np.save(os.path.join(OUT, f"{sys.argv[2]}.npy"), to8(fb[row0:row0 + tr]))
    print(sys.argv[2], "rows", row0, row0 + tr, "rays", cnt["rays"])
else:
    pins = np.load(os.path.join(ROOT, "tests/golden/reference_image_pins.npz"))
    meta = json.load(open(os.path.join(ROOT, "tests/golden/reference_image_pins.json")))
    tr = meta["tile_rows"]
    ref = pins["rows_final"][TILE * tr:(TILE + 1) * tr]
    names = sorted(f[:-4] for f in os.listdir(OUT) if f.endswith(".npy"))
    tiles = {n: np.load(os.path.join(OUT, n + ".npy")) for n in names}
    print(f"final scene, pinned tile {TILE} ({tr} rows x 800 pixels), {NS} spp, 8-bit pixels")
    for n in names:
        print(f"  {n:28s} vs finalScene.png : {stats(tiles[n], ref)}")
    for n in names[1:]:
        print(f"  {n:28s} vs {names[0]:14s} : {stats(tiles[n], tiles[names[0]])}")
