#!/usr/bin/env python3
"""bench.py -- headline benchmark: Direct N^2 + Velocity Verlet at N = 1,048,576 (Plummer sphere,
eps = 1e-3, fp32) on 1/2/4/8 MI355X.  One "step" = one full Velocity-Verlet step (drift, all-pairs
forces, kick).  Prints ONE JSON line on rank 0.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

value   = pair interactions per second of the whole job = N^2 * K / t  (self pairs counted,
          they contribute exactly 0), inputs resident in HBM
scaling = "strong": the total body count stays 1,048,576 as the GPU count grows (BASELINE.json
          config 3); each rank owns N/P targets and all-gathers 16 B/body once per step.
roofline: the dominant kernel is nbh::direct_kernel.  It is FP32-VALU issue bound (no MFMA: the
          pair interaction is not a contraction; HBM traffic is 32 B/body per launch), so `bound`
          is "valu" with the FP32 vector peak of MICROARCH.md (157.3 TFLOP/s) and 20 flop per
          pair interaction; the HBM view is given beside it.
cpu_baseline: the oracle's restatement of the reference loop (fp32, sequential accumulation,
          OpenMP over targets) on a bounded sample of the same workload, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_PAIR = 20.0          # Nyland/Harris/Prins convention, SURVEY.md section 8d
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: Peak FP32 (vector)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--bodies", dest="n", type=int, default=1 << 20,
                    help="total bodies (default 1,048,576; not --n: torch.distributed.run treats that as an "
                         "ambiguous abbreviation of its own options)")
    ap.add_argument("--eps", type=float, default=1e-3)
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend; gloo + --one-device rehearses the N-rank path on ONE GPU")
    ap.add_argument("--one-device", action="store_true", help="every rank uses cuda:0 (rehearsal only)")
    ap.add_argument("--variant", type=int, default=-1, help="kernel variant override (experiments)")
    ap.add_argument("--tpl", type=int, default=0, help="targets per lane override")
    ap.add_argument("--splits", type=int, default=0, help="source splits override")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--atomics", action="store_true",
                    help="Direct: fp64 atomics instead of the default slot planes + fixed-order sums (not bitwise "
                         "reproducible, and slower: 190 against 175 ms per step for general masses at N = 2^20, "
                         "profiles/r03_direct_det_probe.txt; kept for A/B runs and for sizes whose slot planes do "
                         "not fit the budget)")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the Barnes-Hut / spatial-hash / general-mass objects appended to the default run")
    ap.add_argument("--workload", choices=["direct", "hash", "bh"], default="direct",
                    help="direct = the headline metric (default); hash = BASELINE config 5 (N=4,194,304 "
                         "uniform box, spatial hash, z-slab shards + halo exchange); bh = config 4 "
                         "(N=1,048,576 two-galaxy, Barnes-Hut theta 0.5, one GPU).  hash/bh report steps/s")
    ap.add_argument("--no-clock", action="store_true", help="do not sample the engine clock / package power (sysfs) beside the timed steps")
    ap.add_argument("--force-sharded", action="store_true",
                    help="use the multi-GPU (RCCL) code path even with one rank (rehearsal)")
    ap.add_argument("--sharded-host", choices=["cabi", "torch"], default="cabi",
                    help="Direct on N ranks: cabi = the whole step behind the C ABI (include/nbody_hip_comm.h: RCCL "
                         "from C++, one call per step); torch = the torch.distributed host (n-body_amd/distributed.py)")
    ap.add_argument("--pmc", action="store_true",
                    help="Direct, one GPU: before the timed run, collect FETCH_SIZE and WRITE_SIZE of the dominant kernel with "
                         "two rocprofv3 --pmc passes over this same command (child processes started BEFORE this process "
                         "touches the GPU; counters never combined with other trace domains) and report the HBM bytes per "
                         "launch they give as roofline.traffic instead of the committed profiles/pmc_summary.json value")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline wall time")
    ap.add_argument("--kernel-iters", type=int, default=3, help="launches for the roofline timing")
    return ap.parse_args()


def cpu_share():
    """CPUs this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, n)


_NATIVE_ORACLE = None  # path of the -march=native timing build (prepare_cpu_oracle), or None: the prebuilt library


def prepare_cpu_oracle():
    """Rebuilds the oracle's timing build for THIS host's cores (gcc -O3 -march=native -fopenmp).  Called at the very
    start of main(), BEFORE anything initialises the GPU: a process that holds the GPU starts no child program."""
    global _NATIVE_ORACLE
    try:
        tmp = tempfile.mkdtemp(prefix="nbody_oracle_")
        so = os.path.join(tmp, "libnbody_oracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-ffp-contract=off",
                               "-fno-fast-math", "-fvisibility=hidden", "-shared", "-o", so,
                               os.path.join(ROOT, "oracle", "nbody_oracle.c"), "-lm"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _NATIVE_ORACLE = so
    except Exception:
        _NATIVE_ORACLE = None


def cpu_baseline(ic, eps, seconds):
    """Times the oracle (kind "port": the reference has no CPU force path) on a
    bounded sample: T targets spread over the index range x all N sources, fp32 sequential sum."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bind
    lib = None
    try:  # the timing build made for this host's cores before the GPU was touched, else the prebuilt library
        lib = oracle_bind.Oracle(_NATIVE_ORACLE) if _NATIVE_ORACLE else None
    except Exception:
        lib = None
    if lib is None:
        lib = oracle_bind.load(fast=True)
    n = ic["pos_x"].size
    eps2 = float(np.float32(eps) * np.float32(eps))
    cores = min(lib.num_threads(), cpu_share())
    lib.L.oracle_set_num_threads(cores)
    x, y, z, m = ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]

    def run(t):
        idx = np.linspace(0, n - 1, t).astype(np.int64)
        t0 = time.perf_counter()
        lib.direct_forces_indexed(x, y, z, m, idx, 1.0, eps2, 0)
        return time.perf_counter() - t0

    t_probe = max(cores * 4, 32)
    dt = run(t_probe)
    while dt < 0.5 and t_probe < n // 8:  # probe long enough to see the steady rate
        t_probe *= 4
        dt = run(t_probe)
    rate = t_probe * n / dt
    targets = int(min(n, max(t_probe, rate * seconds / n)))
    targets = max(cores, targets // cores * cores)
    dt = run(targets)
    return {"value": targets * n / dt, "unit": "pair-interactions/s", "cores": cores, "kind": "port",
            "sample": f"{targets} of {n} targets x all {n} sources, fp32 sequential sum "
                      f"(oracle mode 0), OpenMP over targets, {dt:.1f} s"}


def read_pmc_issue(name):
    """Instruction-issue view of a kernel (VALU busy cycles / SIMD cycles, scalar-unit busy, mean resident waves) from
    the committed counter summary of the SAME command (tools/profile_step.sh -> tools/pmc_step_summarise.py); PMC
    passes serialise the kernels, so they are separate runs -- never collected inside the timed bench."""
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            d = json.load(f)
        out = dict(d["issue"])
        out.update({"mean_waves_per_simd": d.get("mean_waves_per_simd"), "valu_per_wave": d["per_wave"].get("VALU"),
                    "salu_per_wave": d["per_wave"].get("SALU"), "launch_ms_profiled": d.get("launch_ms_profiled"),
                    "source": "profiles/" + name})
        return out
    except Exception:
        return None


_PMC_MEASURED = None  # {"hbm_bytes_per_launch", ...} of collect_pmc_traffic (--pmc), else None


def collect_pmc_traffic(a):
    """--pmc: HBM bytes per launch of the dominant Direct kernel, measured by two rocprofv3 counter passes over THIS
    command (FETCH_SIZE and WRITE_SIZE cannot share a pass: MICROARCH.md, rocprofv3 PMC slots), gfx950 corrections as
    that guide prescribes (both counters in KB; FETCH_SIZE x 2 for wide coalesced reads).  Runs child processes, so it
    is called before anything initialises the GPU here; the children get --no-clock / --no-cpu-baseline / --no-extra
    (nothing but the kernels under the counters) and the program under rocprofv3 is python3 itself."""
    global _PMC_MEASURED
    import csv
    import shutil
    if shutil.which("rocprofv3") is None:
        return
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) over `bench.py --steps 2 --warmup 1 "
                     f"--bodies {a.n}` in this run: FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes, per launch of the longest "
                     "Direct kernel"}
    tmp = tempfile.mkdtemp(prefix="nbody_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    kb = {}
    for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        d = os.path.join(tmp, name)
        cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "run", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--bodies", str(a.n),
               "--eps", str(a.eps), "--no-cpu-baseline", "--no-extra", "--no-clock", "--kernel-iters", "1"]
        if a.atomics:
            cmd.append("--atomics")
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
            per_kernel = {}
            for root, _, files in os.walk(d):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        for row in csv.DictReader(open(os.path.join(root, f))):
                            if row["Counter_Name"] != ctr or "direct" not in row["Kernel_Name"]:
                                continue
                            k = per_kernel.setdefault(row["Kernel_Name"].split("(")[0], {"v": {}, "ns": 0})
                            k["v"][row["Dispatch_Id"]] = k["v"].get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
                            k["ns"] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            if not per_kernel:
                return
            kname, best = max(per_kernel.items(), key=lambda kv: kv[1]["ns"])  # the dominant kernel
            kb[name] = sum(best["v"].values()) / len(best["v"])
            out["kernel"] = kname.replace("void ", "")
        except Exception:
            return
        finally:
            shutil.rmtree(os.path.join(tmp, name), ignore_errors=True)
    out["FETCH_SIZE_KB"], out["WRITE_SIZE_KB"] = kb["fetch"], kb["write"]
    out["hbm_bytes_per_launch"] = 2.0 * 1024.0 * kb["fetch"] + 1024.0 * kb["write"]
    _PMC_MEASURED = out


def read_pmc_traffic():
    """HBM bytes per direct_kernel launch from the committed rocprofv3 PMC summary, if any."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        with open(p) as f:
            return json.load(f).get("direct_kernel", {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


class ClockSampler:
    """Engine clock and package power of the GPU WHILE the timed region runs, read from sysfs by a host thread every
    ~0.1 s (pp_dpm_sclk / hwmon freq1_input and power1_average of the PCI device torch reports for cuda:current; nothing
    touches the GPU's queues and NO child process is started: a process that has initialised the GPU must not spawn
    `rocm-smi`, a `#!/usr/bin/env python3` script, least of all under rocprofv3's preloaded tool library).  The Direct
    kernel keeps every VALU busy and runs into the 1,400 W package limit: the sustained clock is 2.30-2.40 GHz depending
    on the box, which is the box-to-box spread of the step time.  `roofline.clock` reports it and the
    roofline fraction re-priced at the clock that was actually there.  No readable sysfs node: no clock record."""

    def __init__(self, device_index=0):
        import threading
        self._stop = threading.Event()
        self.samples = []
        self.src = self.sysfs_sources(device_index)
        self._thread = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def under_profiler():
        env = os.environ
        return any("rocprof" in env.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))

    @staticmethod
    def sysfs_sources(device_index=0):
        """sysfs files of the GPU torch calls cuda:<device_index> (matched by PCI address; card0 need not be it)."""
        import glob
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            addr = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        except Exception:
            return {}
        base = os.path.join("/sys/bus/pci/devices", addr)
        out = {}
        if os.path.exists(os.path.join(base, "pp_dpm_sclk")):
            out["sclk"] = os.path.join(base, "pp_dpm_sclk")
        for h in sorted(glob.glob(os.path.join(base, "hwmon", "hwmon*"))):
            for name in ("power1_average", "power1_input"):
                if "power" not in out and os.path.exists(os.path.join(h, name)):
                    out["power"] = os.path.join(h, name)
            if os.path.exists(os.path.join(h, "freq1_input")):
                out["freq"] = os.path.join(h, "freq1_input")
        return out if ("sclk" in out or "freq" in out) else {}

    def read(self):
        rec = {}
        try:
            if "freq" in self.src:
                rec["sclk_mhz"] = int(open(self.src["freq"]).read()) / 1e6
            else:
                for line in open(self.src["sclk"]):
                    if "*" in line:
                        rec["sclk_mhz"] = float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
            if "power" in self.src:
                rec["power_w"] = int(open(self.src["power"]).read()) / 1e6
        except Exception:  # unreadable node, another format: the record simply has no clock
            return None
        return rec if "sclk_mhz" in rec else None

    def _run(self):
        while not self._stop.is_set():
            rec = self.read()
            if rec:
                self.samples.append(rec)
            self._stop.wait(0.1)

    def start(self):
        if self.src:
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread.is_alive():
            self._thread.join(timeout=10)
        busy = [r for r in self.samples if r.get("power_w", 0.0) > 600.0] or self.samples
        if not busy:
            return None
        clk = sorted(r["sclk_mhz"] for r in busy)
        pw = [r.get("power_w", 0.0) for r in busy]
        return {"sclk_mhz_median": clk[len(clk) // 2], "sclk_mhz_min": clk[0], "sclk_mhz_max": clk[-1],
                "power_w_mean": sum(pw) / len(pw), "samples": len(busy),
                "source": "sysfs (" + ", ".join(sorted(os.path.basename(v) for v in self.src.values()))
                          + ") beside the timed steps (samples above 600 W)"}


def run_other_workload(a, nb, ctx, world, rank, sharded, dist, torch, workload=None, n_bodies=None,
                       steps=None, warmup=None):
    """BASELINE configs 4 and 5: steps/s of a full Velocity-Verlet step (not the headline metric)."""
    from nbody_amd.distributed import HipBackend, ShardedHashSystem
    dt = a.dt
    workload = workload or a.workload
    steps = steps or a.steps
    warmup = a.warmup if warmup is None else warmup
    sync_system = None
    if workload == "hash":
        n = n_bodies or (a.n if a.n != (1 << 20) else 4194304)
        half = 0.5 * (n / 16.0) ** (1.0 / 3.0)  # 16 bodies per unit volume
        ic = nb.ic.uniform_box(n, seed=42, lo=-half, hi=half)
        cell, cutoff, eps = 1.0, 1.0, 0.01
        if sharded:
            csys, why = None, ""
            if a.sharded_host == "cabi" and a.dist_backend == "nccl" and not a.one_device:
                try:  # the whole step behind the C ABI (csrc/sharded_hash.hip), RCCL communicator made from C++
                    from nbody_amd.sharded import Comm, ShardedHash
                    comm = Comm.from_torch_distributed(torch.cuda.current_device())
                    csys = ShardedHash(comm, n, 1.0, eps, cell, cutoff)
                    csys.set_state(ic)
                    csys.forces()
                    csys.synchronize()
                except Exception as e:
                    why = f" (C-ABI host unavailable: {type(e).__name__}: {e})"
                    csys = None
                ok = torch.tensor([1 if csys is not None else 0], device="cuda")
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                if int(ok.item()) == 0:
                    csys = None
            if csys is not None:
                step = lambda: csys.step(dt, 1)  # noqa: E731
                sync_system = csys.synchronize
                path = ("C ABI (nbody_hip_sharded_hash_step): z-slab shards; per step all-reduce of the box, one partition "
                        "pass + all-reduce of the counts, ONE host sync, migrating bodies point to point, halo layers "
                        "exchanged while the own x own force kernel runs")
            else:
                sysm = ShardedHashSystem(ic, 1.0, eps, cell, cutoff, backend=HipBackend(ctx))
                sysm.initial_forces()
                step = lambda: sysm.step(dt)  # noqa: E731
                path = ("torch.distributed host: z-slab shards: all-reduce bbox, one partition pass + all-reduce of counts, "
                        "ONE host sync, all-to-all of the migrating bodies, halo all-to-all overlapped with the own x own "
                        "force kernel" + why)
        else:
            ps = nb.ParticleSystem()
            ps.initialize(nb.SimulationConfig(particle_count=n, force_method=nb.ForceMethod.SPATIAL_HASH, dt=dt,
                                              softening=eps, spatial_hash_cell_size=cell,
                                              spatial_hash_cutoff=cutoff), initial_conditions=ic)
            step = lambda: ps.update(dt)  # noqa: E731
            path = "ParticleSystem(SPATIAL_HASH)"
        name = f"uniform_box_N{n}_spatial_hash_cell1_cutoff1_velocity_verlet"
    else:
        n = n_bodies or a.n
        ic = nb.ic.two_galaxies(n, seed=42)
        ic["mass"] = (ic["mass"] / np.float32(n)).astype(np.float32)
        if sharded:
            from nbody_amd.distributed import ShardedTreeSystem
            sysm = ShardedTreeSystem(ic, 1.0, 0.1, 0.5, backend=HipBackend(ctx))
            sysm.initial_forces()
            step = lambda: sysm.step(dt)  # noqa: E731
            path = "replicated tree, walk partitioned over the ranks, all-gather + reduce-scatter per step"
        else:
            ps = nb.ParticleSystem()
            ps.initialize(nb.SimulationConfig(particle_count=n, force_method=nb.ForceMethod.BARNES_HUT, dt=dt,
                                              softening=0.1, barnes_hut_theta=0.5), initial_conditions=ic)
            step = lambda: ps.update(dt)  # noqa: E731
            path = "ParticleSystem(BARNES_HUT, theta 0.5)"
        name = f"two_galaxies_N{n}_barnes_hut_theta0.5_velocity_verlet"

    def barrier():
        if sync_system is not None:  # the C-ABI system's own streams first
            sync_system()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    out = {"metric": "steps_per_s", "value": steps / elapsed, "unit": "steps/s", "n_gpus": world,
           "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic", "config": {"workload": name, "bodies": n, "dt": dt, "path": path}}
    if not sharded:
        out["roofline"] = other_roofline(a, nb, ps, torch, workload)
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = other_cpu_baseline(workload, ic, a.cpu_seconds / 2.0)
    return out


def other_cpu_baseline(workload, ic, seconds):
    """The oracle's Barnes-Hut / spatial hash (kind "port") on this box's cores: the
    serial build once + the walk of a bounded target sample, extrapolated to one force evaluation
    over all bodies (the O(N) Velocity-Verlet passes are not included: they favour the CPU figure)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bind
    lib = oracle_bind.load(fast=True)
    cores = min(lib.num_threads(), cpu_share())
    lib.L.oracle_set_num_threads(cores)
    x, y, z, m = (np.ascontiguousarray(ic[k], dtype=np.float32) for k in ("pos_x", "pos_y", "pos_z", "mass"))
    n = x.size
    if workload == "bh":
        eps2, theta = float(np.float32(0.1) * np.float32(0.1)), 0.5

        def run(t):
            idx = np.linspace(0, n - 1, t).astype(np.int64)
            t0 = time.perf_counter()
            lib.barnes_hut_forces(x, y, z, m, idx, 1.0, eps2, theta)
            return time.perf_counter() - t0
        what = "oracle octree (serial sort + build) + theta-0.5 walk"
    else:
        eps2 = float(np.float32(0.01) * np.float32(0.01))
        lo, _, dims = lib.hash_grid(x, y, z, 1.0)

        def run(t):  # targets = the first t bodies (the uniform box has no index/position correlation)
            t0 = time.perf_counter()
            lib.spatial_hash_forces_grid(x, y, z, m, t, 1.0, eps2, 1.0, 1.0, lo, dims)
            return time.perf_counter() - t0
        what = "oracle grid (serial counting sort) + 27-cell cutoff sums"
    t_small = cores
    dt_small = run(t_small)                      # ~ the build alone
    t_probe = cores * 256
    dt_probe = run(t_probe)
    per_target = max(dt_probe - dt_small, 1e-9) / (t_probe - t_small)
    targets = int(min(n, max(t_probe, (seconds - dt_small) / per_target)))
    dt_big = run(targets)
    per_target = max(dt_big - dt_small, 1e-9) / (targets - t_small)
    t_eval = dt_small + per_target * n
    return {"value": 1.0 / t_eval, "unit": "force-evaluations/s", "cores": cores, "kind": "port",
            "sample": f"{what}: build {dt_small:.2f} s + {targets} of {n} targets in {dt_big - dt_small:.2f} s, "
                      f"extrapolated to all {n} targets = {t_eval:.1f} s per force evaluation; OpenMP over targets"}


def n_cells_dense(cnt):
    """the grid picks the wave-per-cell kernel with two bodies per lane from 8 bodies per cell"""
    return cnt.sum() / max(cnt.size, 1) >= 8.0


def other_roofline(a, nb, ps, torch, workload):
    """SURVEY 8d figures for the force kernel of configs 4 / 5, timed alone with events on the
    stream it runs on (the null stream = torch's current stream)."""
    fc, d = ps.force_calculator_, ps.d_particles_
    # (these launches are ~1 ms: 30 at least, so that the walk's cost-ordered schedule -- fed by the previous walk -- has
    # settled as it has inside a run of steps; the first few walks after a pause are up to 10 % slower)
    iters = max(30, a.kernel_iters)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if workload == "bh":
        tree = fc.getTree()
        run = lambda: tree.computeForces(d, fc.theta_, fc.G_, fc.softening_eps_)  # noqa: E731
    else:
        grid = fc.getGrid()
        run = lambda: grid.computeForces(d, fc.cutoff_radius_, fc.G_, fc.softening_eps_)  # noqa: E731
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    if workload == "bh":
        tree.countVisits(True)  # one more walk with the diagnostics instantiation (slow: atomics per node), untimed
        run()
        st = tree.stats()  # node records fetched by the last walk, summed over waves
        tree.countVisits(False)
        visits = int(st["nodes_visited"])
        nbytes = 32.0 * visits + 16.0 * d.count  # 32-byte record per visit + the body itself
        kernel = "bh_traverse_pair_kernel" if d.count >= 98304 else "bh_traverse_kernel"  # few bodies: split walk
        return {"kernel": kernel, "bound": "hbm", "achieved": nbytes / t / 1e9, "peak": 8000.0,
                "unit": "GB/s", "frac": nbytes / t / 8e12, "traffic": None, "avg_kernel_ms": t * 1e3,
                "node_visits_per_s": visits / t, "nodes": st["node_count"],
                "issue": read_pmc_issue("r04_pmc_bh_pair_kernel.json") or read_pmc_issue("r03_pmc_bh_pair_kernel.json"),
                "note": "32 B per node record a wave fetches (scalar cache / L2, mostly not HBM) + 16 B per body; "
                        "the walk is latency- and issue-bound, see DESIGN.md 4.5"}
    cs, ce, _, _ = grid.copyCellDataToHost()
    gx, gy, gz = grid.getGridDims()
    cnt = (ce - cs).astype(np.int64).reshape(gz, gy, gx)
    pad = np.pad(cnt, 1)
    nb27 = np.zeros_like(cnt)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                nb27 += pad[dz:dz + gz, dy:dy + gy, dx:dx + gx]
    pairs = float((cnt * nb27).sum())  # candidate pairs: every body against the bodies of its 27 cells
    flops = 20.0 * pairs
    return {"kernel": ("hash_cell_force_kernel<false,2,...,FILTER=true> (wave per cell, two targets per lane, window filtered by the "
                       "box of the targets)") if n_cells_dense(cnt) else "hash_force_kernel",
            "bound": "valu", "achieved": flops / t / 1e12, "peak": 157.3,
            "unit": "TFLOP/s", "frac": flops / t / 157.3e12, "traffic": None, "avg_kernel_ms": t * 1e3,
            "candidate_pairs_per_s": pairs / t, "gather_bytes_per_s": 16.0 * pairs / t,
            "issue": read_pmc_issue("r04_pmc_hash_cell_kernel.json") or read_pmc_issue("r03_pmc_hash_cell_kernel.json"),
            "note": "20 flop per candidate pair (distance + cutoff test + force); 16 B per candidate pair is the "
                    "SURVEY 8d gather figure, served from LDS tiles (HBM traffic is 28 B per body)"}


def main():
    a = parse()
    # stdout carries exactly ONE JSON line: RCCL prints a version banner to stdout when its first
    # communicator comes up, so everything else written to fd 1 is sent to stderr instead
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if (int(os.environ.get("RANK", "0")) == 0 and int(os.environ.get("WORLD_SIZE", "1")) == 1
            and not a.no_cpu_baseline and not ClockSampler.under_profiler()):
        prepare_cpu_oracle()  # (child processes of a run are started here, before the GPU is initialised)
    if (a.pmc and int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.workload == "direct" and not a.force_sharded
            and not ClockSampler.under_profiler()):
        collect_pmc_traffic(a)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    if a.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or a.force_sharded
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    import nbody_amd as nb
    from nbody_amd.distributed import HipBackend, ShardedDirectSystem

    ctx = nb.default_context(local_rank)
    if a.variant >= 0 or a.tpl or a.splits:
        ctx.tuning(a.variant, a.tpl, a.splits)
    if a.atomics:
        ctx.deterministic(False)

    n = a.n
    G, eps, dt = 1.0, a.eps, a.dt
    if a.workload != "direct":
        out = run_other_workload(a, nb, ctx, world, rank, sharded, dist, torch)
        if sharded:
            dist.barrier()
            dist.destroy_process_group()
        if out is not None:
            sys.stdout.flush()
            os.write(real_stdout, (json.dumps(out) + "\n").encode())
        return
    ic = nb.ic.plummer(n, seed=42)  # every rank builds the same bodies, keeps its shard

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    if not sharded:
        # the drop-in path: 13-array ParticleData behind ForceCalculator / Integrator
        d = nb.ParticleData()
        nb.ParticleDataManager.allocateDevice(d, n)
        h = nb.ParticleData()
        nb.ParticleDataManager.allocateHost(h, n)
        for k, v in ic.items():
            getattr(h, k)[:] = v
        nb.ParticleDataManager.copyToDevice(d, h)
        fc = nb.DirectForceCalculator()
        fc.setGravitationalConstant(G)
        fc.setSofteningParameter(eps)
        integ = nb.Integrator()
        fc.computeForces(d)  # a(0), ParticleSystem::initialize (particle_system.cpp:88-91)
        step = lambda: integ.integrate(d, fc, dt)  # noqa: E731
        path = "ParticleSystem-style: Integrator.integrate(ParticleData, DirectForceCalculator)"
    else:
        csys = None
        why = ""
        selfcheck = 0.0
        if a.sharded_host == "cabi" and a.dist_backend == "nccl" and not a.one_device:
            # the whole step behind the C ABI: RCCL communicator made from C++ (id broadcast over the process
            # group), drift -> all-gather || own x own -> shard pairs -> point-to-point reaction exchange -> kick
            dog = None
            try:
                from nbody_amd.sharded import Comm, ShardedDirect
                comm = Comm.from_torch_distributed(local_rank)
                csys = ShardedDirect(comm, n, G, eps)
                csys.set_state(ic)
                # a first evaluation that never returns (a collective one rank did not post) must not cost the caller its
                # whole time limit: after 180 s the process says so and leaves
                import threading
                dog = threading.Timer(180.0, lambda: (sys.stderr.write("bench.py: the C-ABI sharded host did not finish its "
                                                                       "first force evaluation within 180 s; run with "
                                                                       "--sharded-host torch\n"), sys.stderr.flush(), os._exit(3)))
                dog.daemon = True
                dog.start()
                csys.forces()
                csys.synchronize()
                dog.cancel()
                # the exchange over RCCL with W > 1 cannot be rehearsed on a one-GPU box: before the timed steps, this
                # rank's rows of a(0) against the one-sided kernel over ALL bodies (N^2 / W pair evaluations, ~30 ms).
                # Two fp32 evaluations (a deterministic sum against a one-sided running sum), each held to 1e-5 of the
                # fp64-accumulated oracle by the parity tests: 2e-5 between them -- a reaction block summed twice or dropped
                # is orders above it.
                from nbody_amd.sharded import shard_bounds
                _, lo, hi = shard_bounds(n, world, rank)
                mine = csys.get_state(gather=False, what=("acc",))
                got = np.stack([mine[k][lo:hi] for k in ("acc_x", "acc_y", "acc_z")], 1)
                pall = torch.from_numpy(np.ascontiguousarray(
                    np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
                ref = nb.direct_forces_packed(ctx, pall[lo:hi].contiguous(), pall, G,
                                              float(np.float32(eps) * np.float32(eps))).cpu().numpy()[:, :3]
                del pall
                err = np.linalg.norm(got - ref, axis=1) / np.maximum(np.linalg.norm(ref, axis=1), 1e-30)
                if os.environ.get("NBODY_BENCH_INJECT_SELFCHECK_FAILURE") == "1":  # tests/test_facade_gpu.py: the fallback
                    err = err + 1.0
                if not np.all(np.isfinite(got)) or float(err.max()) > 2e-5:
                    raise RuntimeError(f"self-check of the sharded a(0) failed on rank {rank}: max relative error {err.max():.3e}")
                selfcheck = float(err.max())
            except Exception as e:  # every rank runs the same build: all fall back alike
                why = f" (C-ABI host unavailable: {type(e).__name__}: {e})"
                if csys is not None:
                    try:
                        csys.close()
                    except Exception:
                        pass
                csys = None
            finally:
                if dog is not None:
                    dog.cancel()  # (also when forces() raised: the fallback below must not be shot by the watchdog)
            ok = torch.tensor([1 if csys is not None else 0], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                csys = None
        if csys is not None:
            step = lambda: csys.step(dt, 1)  # noqa: E731
            path = ("C ABI (nbody_hip_sharded_direct_step): index-range shards; per step one RCCL all-gather of float4 "
                    "positions overlapped with own x own, each shard pair evaluated once by one rank, reaction blocks "
                    "sent point-to-point to their owners, fixed-order sum fused with the kick; a(0) of rank 0 checked "
                    "against the one-sided kernel over all bodies before the timed steps: max relative error %.1e"
                    % selfcheck)
            inner_barrier = barrier

            def barrier():  # noqa: F811  (the system's own streams first)
                csys.synchronize()
                inner_barrier()
        else:
            sysm = ShardedDirectSystem(ic, G, eps, backend=HipBackend(ctx))
            sysm.initial_forces()
            step = lambda: sysm.step(dt)  # noqa: E731
            path = ("torch.distributed host: index-range shards; per step one RCCL all-gather of float4 positions and one "
                    "reduce-scatter of float4 accelerations; each shard pair evaluated once (mode %s)%s" % (sysm.mode, why))

    for _ in range(a.warmup):
        step()
    barrier()
    sampler = (ClockSampler(torch.cuda.current_device()).start()
               if (rank == 0 and world == 1 and not a.no_clock and not ClockSampler.under_profiler()) else None)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    clock = sampler.stop() if sampler else None
    if sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    out = None
    if rank == 0:
        pairs = float(n) * float(n) * a.steps
        value = pairs / elapsed
        out = {
            "metric": "pair_interactions_per_s", "value": value, "unit": "pair-interactions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps, "steps_per_s": a.steps / elapsed,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"plummer_N{n}_direct_n2_velocity_verlet", "bodies": n,
                       "eps": eps, "dt": dt, "G": G, "seed": 42, "path": path, "deterministic": not a.atomics,
                       "deterministic_mode": ctx.directInfo()["deterministic_mode"],
                       "sharding": f"targets_by_index_range_x{world}"},
        }
        # --- roofline of the dominant kernel, timed live with HIP events on the launch stream
        p = torch.from_numpy(np.ascontiguousarray(
            np.stack([ic["pos_x"], ic["pos_y"], ic["pos_z"], ic["mass"]], 1))).cuda()
        eps2 = float(np.float32(eps) * np.float32(eps))
        if world == 1:
            # the step's force evaluation: all pairs of one body set -> nbh::direct_sym_kernel
            # which kernel that is comes from the library itself (nbody_hip_direct_info), not from a copy of its rules
            info = ctx.directInfo(n, eps2)
            det = info["kernel"] == 2
            eqm = float(np.ptp(ic['mass'])) == 0.0
            r_run = info["bodies_per_lane_equal"] if eqm else info["bodies_per_lane_general"]
            kname = (f"nbh::direct_sym_kernel<{r_run},false,{'true' if eqm else 'false'},{'true' if det else 'false'}>"
                     if info["kernel"] > 0 else "nbh::direct_kernel (one-sided)")
            ms = nb.time_direct_packed(ctx, p, p, G, eps2, a.kernel_iters)
            pairs = float(n) * n
            alg_bytes = 16.0 * n + 16.0 * n
        else:
            # the step's dominant launch: own shard x one remote shard, action + reaction
            S = (n + world - 1) // world
            r = a.tpl if a.tpl in (2, 4, 6, 8, 16) else (16 if S >= 49152 else 8 if S >= 16384 else 4)
            kname = f"nbh::direct_sym_kernel<{r},true,...,{'true' if not a.atomics else 'false'}> (shard pair)"
            A, B = p[:S].contiguous(), p[S:2 * S].contiguous()
            accA, accB = torch.zeros_like(A), torch.zeros_like(B)
            nb.direct_forces_pair_packed(ctx, A, B, G, eps2, accA, accB)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()  # ctx launches on torch's current stream, so these events bracket the kernels
            for _ in range(a.kernel_iters):
                nb.direct_forces_pair_packed(ctx, A, B, G, eps2, accA, accB)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / a.kernel_iters
            pairs = 2.0 * float(A.shape[0]) * float(B.shape[0])  # ordered pairs covered per launch
            alg_bytes = 32.0 * (A.shape[0] + B.shape[0])
        flops = FLOP_PER_PAIR * pairs
        achieved = flops / (ms * 1e-3) / 1e12
        out["roofline"] = {
            "kernel": kname, "bound": "valu", "achieved": achieved,
            "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VALU_TFLOPS,
            "traffic": ((_PMC_MEASURED["hbm_bytes_per_launch"] if _PMC_MEASURED else read_pmc_traffic()) if world == 1 else None),
            "traffic_source": (_PMC_MEASURED["source"] if _PMC_MEASURED else
                               "profiles/pmc_summary.json: FETCH_SIZE x 2 + WRITE_SIZE of separate rocprofv3 --pmc passes over "
                               "this command (committed; `bench.py --pmc` measures it in the run itself)"),
            "launch_ms": ms, "pairs_per_launch": pairs, "flop_per_pair": FLOP_PER_PAIR,
            "direct_info": (ctx.directInfo(n, eps2) if world == 1 else None),
            "pair_interactions_per_s_kernel": pairs / (ms * 1e-3),
            "hbm": {"algorithmic_bytes": alg_bytes, "achieved": alg_bytes / (ms * 1e-3) / 1e9,
                    "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": alg_bytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS},
            "note": "FP32 VALU-issue bound, not HBM, no MFMA. `achieved` counts the ALGORITHMIC 20 flop "
                    "per ordered pair interaction (SURVEY 8d); the kernel evaluates each unordered pair "
                    "once (action = -reaction: 10 VALU issue slots per ordered pair instead of 16), so "
                    "its executed flop count is lower.  launch_ms = event mean over the kernel + its "
                    "finalize epilogue",
        }
        if world == 1 and info["kernel"] > 0:
            # what the kernel EXECUTES beside the algorithmic figure: each unordered pair once, 18 (equal masses) / 20
            # (general masses) fp32 lane-slots per pair -- a packed instruction is two lane-slots -- against the packed
            # issue rate 1024 SIMDs x 64 lanes x 2 slots / 4 cycles at the clock sampled beside the timed steps
            slots = 18.0 if eqm else 20.0
            clk_hz = (clock["sclk_mhz_median"] if clock else 2400.0) * 1e6
            rate = 1024.0 * 64.0 * 2.0 / 4.0 * clk_hz
            out["roofline"]["executed"] = {
                "lane_slots_per_unordered_pair": slots, "unordered_pairs_per_launch": pairs / 2.0,
                "lane_slots_per_s": slots * pairs / 2.0 / (ms * 1e-3), "packed_issue_rate_lane_slots_per_s": rate,
                "frac": slots * pairs / 2.0 / (ms * 1e-3) / rate,
                "clock_mhz": clk_hz / 1e6, "clock_source": "sampled beside the timed steps" if clock else "nominal 2400 MHz (no sample)"}
        if clock:
            # the same fraction against the FP32 peak at the clock the chip actually sustained (peak is quoted at 2.4 GHz)
            clock["frac_at_sustained_clock"] = out["roofline"]["frac"] * 2400.0 / clock["sclk_mhz_median"]
            clock["step_frac_at_sustained_clock"] = (FLOP_PER_PAIR * float(n) * n / (elapsed / a.steps) / 1e12
                                                     / PEAK_FP32_VALU_TFLOPS * 2400.0 / clock["sclk_mhz_median"])
            out["roofline"]["clock"] = clock
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ic, eps, a.cpu_seconds)
        if world == 1 and not a.no_extra and n >= 12288 and a.variant in (-1, 3):
            # the same bodies with masses spread over +-25 %: the general-mass instantiation
            # <R,false,false> (two more multiplies per pair than the equal-mass one above)
            rng = np.random.default_rng(7)
            pg = p.clone()
            pg[:, 3] *= torch.from_numpy((0.75 + 0.5 * rng.random(n)).astype(np.float32)).cuda()
            ms_g = nb.time_direct_packed(ctx, pg, pg, G, eps2, a.kernel_iters)
            ach_g = FLOP_PER_PAIR * float(n) * n / (ms_g * 1e-3) / 1e12
            out["roofline"]["general_mass"] = {
                "kernel": (f"nbh::direct_sym_kernel<{info['bodies_per_lane_general']},false,false,"
                           f"{'true' if det else 'false'}>"), "launch_ms": ms_g, "achieved": ach_g,
                "frac": ach_g / PEAK_FP32_VALU_TFLOPS, "pair_interactions_per_s_kernel": float(n) * n / (ms_g * 1e-3),
                "note": "same positions, masses multiplied by U[0.75, 1.25)"}
            if clock:  # (the clock sampled beside the timed steps: this launch runs at the same power limit)
                out["roofline"]["general_mass"]["frac_at_sustained_clock"] = \
                    ach_g / PEAK_FP32_VALU_TFLOPS * 2400.0 / clock["sclk_mhz_median"]
            del pg
        del p
    if not sharded and not a.no_extra and a.n == (1 << 20):
        # BASELINE configs 4 and 5 in the driver-run record (outside the Direct timed region):
        # what `--workload bh` / `--workload hash` print, each with its roofline and CPU baseline
        torch.cuda.empty_cache()
        extra = {}
        # (the hash workload is timed over its first steps: the uniform box clumps under its own short-range gravity --
        # free-fall time ~0.25 = 250 steps of dt 1e-3 -- and a step of the clumped state costs more)
        for key, wl, nn, st, wu in (("barnes_hut", "bh", 1 << 20, 200, 20), ("spatial_hash", "hash", 4194304, 50, 5)):
            try:
                extra[key] = run_other_workload(a, nb, ctx, world, rank, False, dist, torch, workload=wl,
                                                n_bodies=nn, steps=st, warmup=wu)
            except Exception as e:  # the headline line must survive a failure here
                extra[key] = {"error": f"{type(e).__name__}: {e}"}
        if out is not None:
            out["extra"] = extra
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()
