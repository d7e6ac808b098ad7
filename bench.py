#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X render path.

Metric (BASELINE.json): Mrays/s (+ frame ms) on the 1200x800 random scene at 500 spp.
A "step" is one full frame: render_init + render over every pixel (src/main.cu:96-133),
inputs (the flattened scene) already resident in HBM.  A ray is one world->hit call from
color() (main.cu:57), counted on the device.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU; the frame is cut into 4-row tiles dealt round-robin to the
ranks (no cross-GPU rays, the per-pixel seed is global so pixels are identical to the
1-GPU frame); each rank renders its rows into a compact buffer and one RCCL gather
(torch.distributed, backend nccl) brings them to rank 0, which un-interleaves them into
the reference's frame layout.  Total work is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ALGORITHMIC bytes per ray: node, sphere and material records the
# reference's traversal touches per ray, counted on the reference's own tree for this scene
# (40.15 box tests x 32 B + 3.99 sphere tests x 32 B + ~1.1 material reads x 16 B).
ALGO_BYTES_PER_RAY = {"bouncing": 1430.0, "random_scene": 1430.0, "book1": 1430.0, "cornell": 980.0, "cornell_smoke": 1340.0, "final": 2500.0}
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# VALU issue: 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles (MI355X_MICROARCH.md "Wave scheduling",
# cycle-constants row v_fma_f32: 2 cyc throughput), max clock 2.4 GHz  ->  1228.8 G wave-instructions / s.
N_SIMD, CLOCK_HZ, VALU_ISSUE_CYCLES = 1024, 2.4e9, 2.0
VALU_PEAK_GINST = N_SIMD * CLOCK_HZ / VALU_ISSUE_CYCLES / 1e9
VALU_MEASURED_CYCLES = 3.0   # what tools/ubench/valu_rate.hip sustains on this part at >= 2 waves per SIMD (profiles/r01_ubench_valu_rate.txt)
LDS_PEAK_B_PER_CLK_CU = 256.0   # MI355X_MICROARCH.md, LDS table (ds_read_b64 / b128)
TILE_ROWS = 4


def roofline(scene, nx, ny, ns, rays_step, kernel_ms, frame_rays=None):
    """The roofline object of the JSON line.  The binding roof of this path is VALU issue (the scene is LDS-resident and
    compulsory HBM traffic is ~0.01 B/ray): achieved = VALU wave-instructions per second = the per-step instruction
    count of the committed rocprofv3 PMC record (profiles/pmc_<workload>.json, made by tools/profile_bench.sh +
    tools/pmc_to_json.py from the same bench command) / the kernel time measured live here with HIP events.  The HBM
    figure SURVEY.md 8(d) defines (algorithmic bytes / time against 8 TB/s) is kept beside it, with the measured HBM
    bytes as `traffic`."""
    bpr = ALGO_BYTES_PER_RAY.get(scene, 1430.0)
    ksec = kernel_ms * 1e-3
    hbm_alg = rays_step * bpr / ksec / 1e9
    key = f"{'random_scene' if scene == 'bouncing' else scene}_{nx}x{ny}_{ns}"
    rec = None
    path = os.path.join(ROOT, "profiles", f"pmc_{key}.json")
    if os.path.exists(path):
        try:
            rec = json.load(open(path))
        except Exception:
            rec = None
    hbm = {"achieved": round(hbm_alg, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_alg / HBM_PEAK_GBS, 4),
           "algorithmic_bytes_per_ray": bpr,
           "note": "algorithmic bytes (SURVEY.md 8(d)) are served from LDS, not HBM: a fraction of the HBM roof above 1 only says the scene never leaves the CU"}
    if rec is None:
        r = dict(hbm)
        r.update({"bound": "hbm", "traffic": None, "kernel_ms": round(kernel_ms, 3),
                  "note": hbm["note"] + "; no PMC record for this workload under profiles/, so the binding VALU-issue roof is not evaluated"})
        return r
    c = dict(rec["per_step"])
    # The record is one WHOLE frame on one GPU.  In an N-GPU run this rank traced rays_step of the frame's frame_rays rays:
    # its share of the counted instructions is taken in proportion (rows differ in cost per ray, so this is an estimate;
    # the N = 1 line is exact).
    share = 1.0
    if frame_rays and frame_rays > 0 and rays_step < frame_rays:
        share = rays_step / float(frame_rays)
        for k in ("SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
            if k in c: c[k] = c[k] * share
    valu, lanes = c["SQ_INSTS_VALU"], c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"]
    achieved = valu / ksec / 1e9
    hbm_bytes = rec.get("hbm_bytes_per_step")
    if hbm_bytes is not None and share < 1.0:
        hbm_bytes = int(hbm_bytes * share)
    hbm.update({"traffic_bytes_per_step": hbm_bytes, "traffic_gbs": None if hbm_bytes is None else round(hbm_bytes / ksec / 1e9, 2)})
    import hashlib, glob
    h = hashlib.sha1()   # = `cat csrc/* | sha1sum`, what tools/profile_bench.sh recorded beside the counters
    for fn in sorted(glob.glob(os.path.join(ROOT, "accelerated-ray-tracer_amd", "csrc", "*"))):
        h.update(open(fn, "rb").read())
    issue = achieved / VALU_PEAK_GINST
    r = {"bound": "valu_issue", "achieved": round(achieved, 1), "peak": round(VALU_PEAK_GINST, 1), "unit": "Gwave-inst/s",
         "frac": round(issue, 4), "traffic": hbm_bytes, "kernel_ms": round(kernel_ms, 3),
         "valu": {"wave_insts_per_step": valu, "wave_insts_per_ray": round(valu / rays_step, 2),
                  "lanes_per_inst": round(lanes, 2), "lane_utilisation": round(lanes / 64.0, 4),
                  "issue_frac_vs_2_cycles": round(issue, 4),
                  "issue_frac_vs_measured_3_cycles": round(issue * VALU_MEASURED_CYCLES / VALU_ISSUE_CYCLES, 4),
                  "useful_lane_frac_of_valu_peak": round(issue * lanes / 64.0, 4),
                  "wave_wait_frac": round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3) if "SQ_WAIT_ANY" in c else None,
                  "wave_issue_stall_frac": round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 3) if "SQ_WAIT_INST_ANY" in c else None},
         "lds": {"algorithmic_bytes_per_clk_per_cu": round(rays_step * bpr / (ksec * CLOCK_HZ * 256), 2), "peak": LDS_PEAK_B_PER_CLK_CU,
                 "insts_per_step": c.get("SQ_INSTS_LDS"),
                 "bank_conflict_share_of_lds_cycles": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3) if "SQ_LDS_IDX_ACTIVE" in c else None},
         "hbm": hbm,
         "pmc_record": f"profiles/pmc_{key}.json", "pmc_record_is_of_this_build": rec.get("csrc_sha1") == h.hexdigest()[:16],
         "share_of_the_record_frame": round(share, 4),
         "note": "instruction and byte counts per step come from the committed PMC record (rocprofv3 passes of this bench command); the time is this run's: the launches of the render kernels for one frame (with the cost-aware schedule two of the main kernel, each with a tier-kernel launch beside and behind it) plus the ranking kernels between them, HIP events on the launch stream"}
    return r


class RowPlan:
    """Interleaved row tiles: tile t (rows t*tile_rows ...) belongs to rank t % world.
    Mirrors rt_frame_local_rows / rt_local_to_global_row of the C ABI."""

    def __init__(self, ny: int, tile_rows: int, world: int):
        self.ny, self.tile_rows, self.world = ny, tile_rows, world
        self.n_tiles = (ny + tile_rows - 1) // tile_rows
        self._rows = [[t * tile_rows + k for t in range(r, self.n_tiles, world) for k in range(min(tile_rows, ny - t * tile_rows))]
                      for r in range(world)]
        self.max_rows = max(len(g) for g in self._rows)

    def rows_of(self, rank: int):
        return self._rows[rank]


def gather_rows(local, plan: "RowPlan", rank: int, world: int, dev):
    """One gather of the compact per-rank row buffers to rank 0 (RCCL on GPUs, gloo in the CPU test), then the
    un-interleave into the reference's frame layout (row 0 = bottom).  Returns the full frame on rank 0, else None."""
    import torch
    import torch.distributed as dist
    if local.is_cuda and dist.get_backend() == "gloo":
        # rehearsal mode only (RT_BENCH_BACKEND=gloo, several ranks sharing one GPU): gloo gathers host tensors
        host = local.cpu()
        got = [torch.zeros_like(host) for _ in range(world)] if rank == 0 else None
        dist.gather(host, got, dst=0)
        gathered = [g.to(dev) for g in got] if rank == 0 else None
    else:
        gathered = [torch.zeros_like(local) for _ in range(world)] if rank == 0 else None
        dist.gather(local, gathered, dst=0)
    if rank != 0:
        return None
    full = torch.zeros((plan.ny,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    for r in range(world):
        idx = torch.tensor(plan.rows_of(r), dtype=torch.long, device=dev)
        if idx.numel():
            full.index_copy_(0, idx, gathered[r][: idx.numel()])
    return full


def multi_gpu_report(per_rank, ms_per_step: float, args, world: int) -> dict:
    """Per-rank render-kernel time, gather time and rays of an N > 1 run, and the strong-scaling efficiency T1 / (N * T_N)
    against the committed one-GPU line of the same workload (profiles/r*_bench_n1.json, the newest by name)."""
    import glob
    k = [p[0] for p in per_rank]; g = [p[1] for p in per_rank]; r = [p[2] for p in per_rank]
    rep = {"kernel_ms_min": round(min(k), 3), "kernel_ms_max": round(max(k), 3), "kernel_ms_per_rank": [round(x, 3) for x in k],
           "gather_ms_rank0": round(g[0], 3), "gather_ms_max": round(max(g), 3),
           "rays_per_rank_min": int(min(r)), "rays_per_rank_max": int(max(r)),
           "note": "kernel_ms = HIP-event time of a rank's own render launches; gather_ms = the RCCL gather + un-interleave on the same stream, which on rank 0 includes waiting for the slowest rank"}
    t1, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")), reverse=True):
        try:
            rec = json.loads(open(path).read().strip().splitlines()[-1])
        except Exception:
            continue
        wl = rec.get("config", {}).get("workload", "")
        if rec.get("n_gpus") == 1 and f"{args.nx}x{args.ny} @ {args.ns} spp" in wl and args.scene.split("_")[0] in wl:
            t1, src = float(rec["ms_per_step"]), os.path.relpath(path, ROOT)
            break
    if t1 is not None:
        rep.update({"t1_ms": t1, "t1_source": src, "efficiency_t1_over_n_tn": round(t1 / (world * ms_per_step), 4)})
    return rep


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="random_scene")
    ap.add_argument("--nx", type=int, default=1200)
    ap.add_argument("--ny", type=int, default=800)
    ap.add_argument("--ns", type=int, default=500)
    ap.add_argument("--kernel", type=int, default=None, help="0 = pixel, 1 = persistent, 2 = parked, 3 = staged (default)")
    ap.add_argument("--opt", action="append", default=[], help="rt_set_option key=value (A/B knobs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-ns", type=int, default=48, help="spp of the bounded CPU-baseline sample (48 spp of the headline frame = ~1e8 rays, tens of core-seconds)")
    ap.add_argument("--save-ppm", default=None, help="write the last frame as ASCII PPM (rank 0)")
    return ap.parse_args()


def host_cores() -> int:
    """The host cores this process can actually use: its affinity mask, cut down to the container's CPU quota when there
    is one (cgroup cpu.max: a GPU box gives each GPU's job a share of the host, and running one thread per LOGICAL cpu of
    the whole machine inside a 16-cpu share measured 24 Mrays/s against 55 with 32 threads)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(round(q / per))))
            break
        except Exception:
            continue
    return n


def cpu_baseline(scene_name: str, nx: int, ny: int, ns: int) -> dict:
    """The oracle (a port of the reference's algorithm, see oracle/) timed on this box's host cores.  One fixed, stated
    configuration: one thread per usable core (affinity mask cut down to the cgroup CPU quota), no probing; the sample is
    the bench frame at reduced spp (48 spp = ~1e8 rays: 2 - 7 s per run on a 16-cpu share), best of three runs."""
    import oracle
    name = "bouncing" if scene_name == "random_scene" else scene_name
    sc = oracle.OracleScene(name, nx, ny)
    threads = host_cores()
    try:
        logical = len(os.sched_getaffinity(0))
    except AttributeError:
        logical = os.cpu_count() or threads
    sc.render(2, threads=threads, counters=True)   # page the library and the scene in; not timed
    # The GPU boxes' hosts are shared (256 logical cpus, this job's cgroup quota a fraction of them): the same sample ran 3.8x
    # apart minutes apart on one box (profiles/r03z_bench_n1*.json of the first pass).  So the sample is timed three times and
    # the best run is the figure; all three are listed.
    runs = []
    for _ in range(3):
        t0 = time.time()
        _, cnt = sc.render(ns, threads=threads, counters=True)
        runs.append((cnt["rays"] / (time.time() - t0) / 1e6, time.time() - t0))
    best = max(r[0] for r in runs)
    return {"value": round(best, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{scene_name} {nx}x{ny} @ {ns} spp (same scene/seeds, reduced spp), {cnt['rays']} rays, best of three runs "
                      f"({', '.join(f'{r[0]:.1f} Mrays/s in {r[1]:.1f} s' for r in runs)}) on {threads} threads = usable cores (cgroup quota / affinity; "
                      f"{logical} logical cpus in the affinity mask, os.cpu_count() = {os.cpu_count()})"}


def main():
    args = parse()
    # before anything initialises the GPU runtime: the host driver of these boxes only supports dmabuf IPC (RCCL,
    # cross-process device memory); the image exports this already, a bare environment would not
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import accelerated_ray_tracer_amd as art

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the render path has no CPU fallback")
    # RT_BENCH_DEVICE / RT_BENCH_BACKEND exist to rehearse the N > 1 path on a one-GPU box (all ranks on one device,
    # gloo instead of RCCL); the driver's multi-GPU runs use neither.
    device_index = int(os.environ.get("RT_BENCH_DEVICE", local_rank))
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    art.init(device_index)
    if args.kernel is not None:
        art.set_option("kernel", args.kernel)
    for kv in args.opt:
        k, v = kv.split("=")
        art.set_option(k, int(v))

    img, iw, ih = art.default_texture(args.scene)
    hs = art.HostScene(args.scene, args.nx, args.ny, img, iw, ih)
    ds = art.DeviceScene(hs)
    frame = hs.frame(nx=args.nx, ny=args.ny, ns=args.ns, tile_rows=TILE_ROWS if world > 1 else args.ny,
                     tile_first=rank if world > 1 else 0, tile_stride=world)
    rows = art.rt_lib().rt_frame_local_rows(frame)
    plan = RowPlan(args.ny, TILE_ROWS if world > 1 else args.ny, world)
    assert rows == len(plan.rows_of(rank))
    local = torch.zeros((plan.max_rows, args.nx, 3), dtype=torch.float32, device=dev)
    full = None

    stream = torch.cuda.current_stream().cuda_stream
    kernel_ms, gather_ms, rays_step = [], [], 0
    ev_g0, ev_g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # same stream as the render and the gather

    def step(record: bool):
        nonlocal rays_step, full
        _, st = ds.render(frame, out=local.data_ptr(), stream=stream, blocking=False)
        if world > 1:
            ev_g0.record()
            full = gather_rows(local, plan, rank, world, dev)
            ev_g1.record()
        st = ds.finish()   # waits for this rank's kernel; HIP-event duration on the launch stream
        if record:
            kernel_ms.append(st.ms_render)
            rays_step = st.rays
            if world > 1:
                ev_g1.synchronize()
                gather_ms.append(ev_g0.elapsed_time(ev_g1))   # includes waiting for the slowest rank's rows
        return st

    for _ in range(args.warmup):
        step(False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    red_dev = dev if (world == 1 or backend == "nccl") else torch.device("cpu")
    t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    r = torch.tensor([float(rays_step)], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total_rays = float(r.item())
    per_rank = None
    if world > 1:   # every rank's own kernel time and gather time, so that a scaling run explains itself
        mine = torch.tensor([float(np.mean(kernel_ms)), float(np.mean(gather_ms)), float(rays_step)], dtype=torch.float64, device=red_dev)
        got = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        per_rank = [[float(x) for x in g.tolist()] for g in got]

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = total_rays * args.steps / elapsed / 1e6
        kms = float(np.mean(kernel_ms))
        line = {
            "metric": f"Mrays/s, {args.nx}x{args.ny} {'random' if args.scene in ('random_scene', 'bouncing') else args.scene} scene @ {args.ns} spp (frame ms in ms_per_step)",
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.scene} (reference create_world_bouncing, src/main.cu:160)" if args.scene in ("random_scene", "bouncing")
                                    else f"{args.scene} (reference scene function of that name, src/main.cu)") + f" {args.nx}x{args.ny} @ {args.ns} spp, seed 1984+pixel",
                       "rays_per_frame": int(total_rays), "parallelism": f"rows{world}" if world > 1 else "single",
                       "tile_rows": TILE_ROWS if world > 1 else args.ny},
            "roofline": roofline(args.scene, args.nx, args.ny, args.ns, rays_step, kms, total_rays),
        }
        if per_rank is not None:
            line["multi_gpu"] = multi_gpu_report(per_rank, ms_per_step, args, world)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.scene, args.nx, args.ny, args.cpu_ns)
        print(json.dumps(line), flush=True)
        if args.save_ppm:
            out = (full if world > 1 else local[: args.ny]).cpu().numpy()
            art.write_ppm(args.save_ppm, out, hs.ppm_double_scale)
    ds.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
