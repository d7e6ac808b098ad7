#!/usr/bin/env python3
"""Tier-sizing sweep on ONE rank's share of a frame (rank FIRST of a STRIDE-GPU run, default rank 0): every configuration given
on the command line (comma-separated rt_set_option pairs) is rendered ROUNDS times, interleaved, and the minimum kernel time is
printed.  SCENE / NX / NY / NS select the frame.  Frames must be identical across configurations (checked)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import accelerated_ray_tracer_amd as art
art.init(0)
scene, nx, ny, ns = os.environ.get("SCENE", "random_scene"), int(os.environ.get("NX", "1200")), int(os.environ.get("NY", "800")), int(os.environ.get("NS", "500"))
stride, first, rounds = int(os.environ.get("STRIDE", "8")), int(os.environ.get("FIRST", "0")), int(os.environ.get("ROUNDS", "3"))
img, iw, ih = art.default_texture(scene)
hs = art.HostScene(scene, nx, ny, img, iw, ih)
ds = art.DeviceScene(hs)
f = hs.frame(nx=nx, ny=ny, ns=ns, tile_rows=4 if stride > 1 else ny, tile_first=first, tile_stride=stride)
rows = art.rt_lib().rt_frame_local_rows(f)
buf = torch.zeros((rows, nx, 3), dtype=torch.float32, device="cuda")
cfgs = sys.argv[1:] or [""]
best, digest = {c: 1e30 for c in cfgs}, {}
for rnd in range(rounds):
    for c in cfgs:
        art.reset_options()
        for kv in filter(None, c.split(",")):
            k, v = kv.split("="); art.set_option(k, int(v))
        _, st = ds.render(f, out=buf.data_ptr(), blocking=True)
        best[c] = min(best[c], st.ms_render)
        if rnd == 0: digest[c] = hashlib.sha1(buf.cpu().numpy().tobytes()).hexdigest()[:10]
print(f"# {scene} {nx}x{ny}@{ns}, rank {first} of {stride}: min of {rounds} renders, ms")
for c in cfgs:
    print(f"{best[c]:9.3f}  {'same' if digest[c] == digest[cfgs[0]] else 'DIFFERENT'}  {c or '(defaults)'}", flush=True)
