#!/usr/bin/env python3
"""bench.py — ALS sweeps/s of the MI355X-native CP engine on BASELINE.json's headline problem.

    python bench.py [--gpus N --steps K --warmup W]           (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE exact dimension-tree ALS sweep (the body of alsCP_DT's loop incl. Normalize,
reference als_CP.cxx:215-303) of the order-4 s=200 R=10 CP problem of BASELINE.json configs[1]
(`-tensor r`: V = [[W_true]], W_true and W0 ~ U(0,1)), tensor resident in HBM before the timed
region. With N > 1 the SAME tensor is block-partitioned along its leading mode over the N ranks
(strong scaling), one process per GPU, factor-matrix partials reduce-scattered over RCCL.

One JSON line is printed by rank 0; see DESIGN.md "Measurement" for every field.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pairwise-perturbation_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (lens, R)   — BASELINE.json configs[1] and configs[3]
    "cp4_s200_r10": ([200, 200, 200, 200], 10),
    "cp4_s400_r20": ([400, 400, 400, 400], 20),
    "cp4_s64_r10": ([64, 64, 64, 64], 10),  # smoke-sized
    "cp4_s12_r3": ([12, 12, 12, 12], 3),    # the CPU rehearsal of the N > 1 path (tests/test_bench_hostsim.py)
    "cp4_s16_r3": ([16, 12, 12, 12], 3),    # the same for 8 ranks (2 rows of the leading mode each)
    # diagnostics (not BASELINE configs): which of size and rank costs cfg4 its 8 % against cfg2
    "cp4_s400_r10": ([400, 400, 400, 400], 10),
    "cp4_s200_r20": ([200, 200, 200, 200], 20),
}
# the reference's real-data runs (test_ALS.cxx:287-326,366-379; script/script_real.py:42-56): the
# EXTENTS of coil-100 / time-lapse with synthetic values (`--workload coil100|timelapse`; not a
# BASELINE config, never the driver's line): name -> (lens, CP rank, Tucker core ranks)
REAL_SHAPES = {
    "coil100": ([3, 128, 128, 7200], 10, [3, 10, 10, 70]),
    "timelapse": ([33, 1344, 1024, 9], 10, [10, 100, 100, 5]),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)


def sweep_flops(lens, R, schedule="msdt"):
    """flops one exact N=4 sweep executes.
    dt   (alsCP_DT's two-node tree): 4 s^4 R + 4 s^3 R + 8 s^2 R            
    msdt (one first-level contraction per N-1 = 3 mode updates, 4/3 per sweep):
         4/3 * (2 s^4 R  +  2*2 s^3 R  +  3*2 s^2 R)"""
    s = lens[0]
    if schedule == "dt":
        return 4.0 * s ** 4 * R + 4.0 * s ** 3 * R + 8.0 * s ** 2 * R
    return (4.0 / 3.0) * (2.0 * s ** 4 * R + 4.0 * s ** 3 * R + 6.0 * s ** 2 * R)


def _mem_available_bytes():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) * 1024
    except OSError:
        pass
    return 0


def _usable_cpus():
    """cores this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU
    box exposes all 256 hardware threads to a container that is scheduled on 16 of them)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def _blas_sweeps(V, W0, R, ncpu, budget_s=8.0):
    """exact DT sweeps of an order-4 CP problem with the reference's TTM order (mttkrp_map_DT,
    common.cxx:56,83: the first-level node contracts its sibling's modes one at a time), the two
    2 s^4 R products per sweep through numpy's BLAS dgemm (OpenBLAS), no copy of the tensor; the
    s^3 R and smaller contractions (Hadamard index, not GEMM-shaped) through einsum. Returns
    sweeps/s on `ncpu` BLAS threads."""
    import numpy as np
    try:
        from threadpoolctl import threadpool_limits, threadpool_info
        lim = threadpool_limits(limits=ncpu, user_api="blas")
        libs = sorted({(i.get("internal_api") or "?") + " " + (i.get("version") or "")
                       for i in threadpool_info() if i.get("user_api") == "blas"})
    except Exception:
        lim, libs = None, ["numpy's BLAS"]
    s0, s1, s2, s3 = V.shape
    W = [w.copy() for w in W0]
    V3 = V.reshape(s0 * s1, s2, s3, order="F")         # [(ab), c, d]
    Va = V.reshape(s0, s1 * s2 * s3, order="F").T      # [(bcd), a]  (a view)
    X = np.empty((s0 * s1, s3, R))

    def sweep():
        for i in range(4):
            if i == 0:     # node "ab": V x_c W_c (one dgemm per d), then the Hadamard contraction of d
                for d in range(s3):
                    X[:, d, :] = V3[:, :, d] @ W[2]
                T = np.einsum("mdr,dr->mr", X, W[3]).reshape(s0, s1, R, order="F")
            if i == 2:     # node "cd": V x_a W_a (one dgemm), then b
                Y = (Va @ W[0]).reshape(s2 * s3, s1, R)
                T = np.einsum("xbr,br->xr", Y, W[1]).reshape(s2, s3, R, order="F")
            j = i ^ 1      # the sibling leaf
            M = np.einsum("xyr,yr->xr", T, W[j]) if i % 2 == 0 else np.einsum("xyr,xr->yr", T, W[j])
            S = np.ones((R, R))
            for k in range(4):
                if k != i:
                    S = S * (W[k].T @ W[k])
            U, sv, Vt = np.linalg.svd(S)               # SVD_solve, common.cxx:710-725
            W[i] = M @ (Vt.T @ np.diag(1.0 / sv) @ U.T)
        nrm = [np.linalg.norm(w) for w in W]           # Normalize, common.cxx:680-688
        c = float(np.prod(nrm)) ** 0.25
        for k in range(4):
            W[k] *= c / nrm[k]

    t0 = time.time()
    sweep()
    t_one = time.time() - t0
    k = max(1, min(10, int(budget_s / max(t_one, 1e-3))))
    t0 = time.time()
    for _ in range(k):
        sweep()
    per = (time.time() - t0) / k
    if lim is not None:
        lim.restore_original_limits() if hasattr(lim, "restore_original_limits") else None
    return {"value": 1.0 / per, "unit": "sweeps/s", "cores": ncpu, "kind": "port+blas",
            "sample": f"{k} timed sweeps (after 1 warm-up) of the same problem: the two 2 s^4 R "
                      f"TTMs of a sweep as dgemm ({', '.join(libs)}), {per:.3f} s/sweep on {ncpu} "
                      "BLAS threads; no print blocks inside"}


def cpu_baseline(lens, R, budget_s=25.0):
    """the fp64 oracle (the reference's TTM-by-TTM contraction sequence, OpenMP) timed on this
    host's cores ON THE SAME PROBLEM AT FULL SIZE (cfg2: 12.8 GB of fp64 tensor in host memory,
    SURVEY §8d) — a bounded sample of 2-3 sweeps. Only when the host cannot hold the tensor (or it
    is not cfg2-sized) is the mode size reduced and the s^4 scaling law applied; `sample` says which."""
    import numpy as np
    import oracle_lib as O
    ncpu = _usable_cpus()
    s_full = lens[0]
    need = 8.0 * float(np.prod(lens)) * 1.2 + (4 << 30)
    s = s_full if (need < _mem_available_bytes() and np.prod(lens) <= 2e9) else min(s_full, 96)
    small = [s] * len(lens)
    Wt = O.init_factors(small, R, 1000)
    V = O.build_V(Wt)
    W = O.init_factors(small, R, 2000)
    G = O.init_factors(small, R, 3000)
    # OpenMP team size: the fastest of a few candidates on a small instance of the same problem
    # (what `cores` reports); an oversubscribed team loses badly
    cands = sorted({c for c in (8, 16, 32, 64, ncpu) if c <= max(ncpu, 8)})
    sp = [48] * len(lens)
    Vp = O.build_V(O.init_factors(sp, R, 1000))
    Wp, Gp = O.init_factors(sp, R, 2000), O.init_factors(sp, R, 3000)
    best = None
    for nt in cands:
        O.lib().ppo_set_num_threads(nt)
        O.als_cp_dt(Vp, Wp, Gp, tol=0.0, maxiter=0, resprint=10 ** 9)
        t0 = time.time()
        O.als_cp_dt(Vp, Wp, Gp, tol=0.0, maxiter=2, resprint=10 ** 9)
        dt = time.time() - t0
        if best is None or dt < best[0]:
            best = (dt, nt)
    nthreads = best[1]
    del Vp
    O.lib().ppo_set_num_threads(nthreads)
    # the print blocks (untimed in the reference, als_CP.cxx:167,189) are measured and subtracted
    t0 = time.time()
    O.residual(V, W)
    t_print = time.time() - t0
    t0 = time.time()
    O.als_cp_dt(V, W, G, tol=0.0, maxiter=0, resprint=10 ** 9)  # warm-up: 1 sweep + 2 print blocks
    t_one = max(time.time() - t0 - 2 * t_print, 1e-3)
    k = max(1, min(10, int((budget_s - 2 * t_print) / t_one)))
    t0 = time.time()
    O.als_cp_dt(V, W, G, tol=0.0, maxiter=k - 1, resprint=10 ** 9)
    t = time.time() - t0
    per_sweep = max((t - 2 * t_print) / k, 1e-9)
    scale = (s_full / s) ** 4
    blas = None
    try:
        blas = _blas_sweeps(V, W, R, ncpu, budget_s=8.0) if len(lens) == 4 else None
    except Exception as e:  # reported, never required
        blas = {"value": None, "kind": "port+blas", "sample": f"failed: {e}"}
    if blas and blas.get("value"):
        blas["value"] /= scale
    what = (f"{k} timed sweeps (after 1 warm-up sweep) of the SAME problem at full size s={s_full} "
            f"(fp64 tensor, {8e-9 * float(np.prod(lens)):.1f} GB in host memory)" if s == s_full else
            f"{k} sweeps at reduced size s={s}, scaled by (s/{s})^4 = {scale:.1f} to s={s_full} "
            "(host memory cannot hold the full tensor)")
    port = {
        "value": 1.0 / (per_sweep * scale),
        "unit": "sweeps/s",
        "cores": nthreads,
        "kind": "port",
        "sample": what + f": fp64 OpenMP oracle with the reference's TTM-by-TTM contraction order, "
                         f"{per_sweep:.3f} s/sweep on {nthreads} threads, print blocks "
                         f"({t_print:.2f} s each) subtracted as in als_CP.cxx:167,189",
    }
    # `value` is the STRONGEST CPU number measured here: the same contraction sequence with its two big
    # TTMs on the host's BLAS (the reference runs them as CTF contractions over MKL dgemm,
    # common.cxx:56,83) when that beats the plain OpenMP port, which stays beside it under `openmp_port`
    if blas and blas.get("value") and blas["value"] > port["value"]:
        out = dict(blas)
        out["sample"] = (f"s={s_full}" if s == s_full else f"s={s} scaled by (s/{s})^4 to s={s_full}") + \
            ": " + blas["sample"]
        out["openmp_port"] = port
        return out
    port["blas"] = blas
    return port


def _csv_rows(path):
    rows = []
    for ln in open(path).read().splitlines()[1:]:
        if ln.strip():
            rows.append([float(x) for x in ln.split(",")])
    return rows


def _bench_lines(path):
    out = {}
    for ln in open(path).read().splitlines():
        if "," in ln and ln.strip().startswith("["):
            k, v = ln.split(",")[:2]
            try:
                out.setdefault(k.strip(), []).append(float(v))
            except ValueError:
                pass
    return out


def _median(x):
    x = sorted(x)
    return x[len(x) // 2] if x else None


def live_pmc_traffic(args, timeout_s=240):
    """HBM bytes per launch of the headline's tensor-scan kernel, counted in THIS run: two child runs
    of this script under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in separate passes, no trace
    domain beside them; FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md's HBM section prescribes; both
    in KiB), started BEFORE this process touches the GPU. A few sweeps of the same workload, schedule
    and storage type; every launch of k_scan_suffix* in the child counts (the placement measurement
    of session set-up launches the same kernel on the same bytes). Returns (bytes per launch or None,
    a sentence saying where the figure comes from or why there is none)."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if not rp:
        return None, "no rocprofv3 on PATH"
    got = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rp, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1",
                   "--workload", args.workload, "--dtype", args.dtype,
                   "--schedule", args.schedule or "msdt", "--no-cpu-baseline", "--no-pmc", "--pmc-child"]
            env = dict(os.environ, TMPDIR="/tmp")
            # (its own session / process group: on a timeout the profiled python grandchild is killed
            # with the launcher — it would otherwise keep the GPU busy under the headline's timed region)
            try:
                pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, start_new_session=True)
            except Exception as e:
                return None, f"rocprofv3 --pmc {counter} child: {e}"
            try:
                pr.communicate(timeout=timeout_s)
            except Exception as e:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except Exception:
                    pass
                try:
                    pr.communicate(timeout=30)
                except Exception:
                    pass
                return None, f"rocprofv3 --pmc {counter} child: {e}"
            if pr.returncode != 0:
                return None, f"rocprofv3 --pmc {counter} child exited {pr.returncode}"
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k_scan_suffix" in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                        vals.append(float(r["Counter_Value"]))
            if not vals:
                return None, f"rocprofv3 --pmc {counter}: no k_scan_suffix rows"
            got[counter] = (sum(vals) / len(vals) * 1024.0 * (2.0 if counter == "FETCH_SIZE" else 1.0),
                            len(vals))
    src = ("counted in this run: child passes `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` of this "
           f"command (1 warm-up + 2 sweeps, nothing else), mean over {got['FETCH_SIZE'][1]} / {got['WRITE_SIZE'][1]} launches of "
           f"k_scan_suffix*: fetch {got['FETCH_SIZE'][0]:.4e} B (FETCH_SIZE KiB x 2, gfx950) + write "
           f"{got['WRITE_SIZE'][0]:.4e} B")
    return got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0], src


def _replayed_traffic(key):
    """HBM bytes per launch of the scan kernel from the committed rocprofv3 --pmc passes
    (profiles/pmc_traffic.json): replayed, not observed in this run"""
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        ent = pm.get(key)
        ks = {k: v for k, v in ent.items() if k.startswith("k_scan")}
        return sum(v["fetch_bytes"] + v["write_bytes"] for v in ks.values()) / len(ks)
    except Exception:
        return None


def cfg3_pp_record(ppals, ctx, V, R, W0, G0, vnorm, tmpdir):
    """BASELINE configs[2]: `-pp 1` against `-pp 0` on the resident cfg2 tensor, measured the way
    the reference measures it — the [dtime] column of alsCP_DT / alsCP_PP (print blocks excluded,
    als_CP.cxx:167,189-190) and the [DTtime] / [PPfirst] / [PPsecond] lines of their `bench` mode
    (als_CP.cxx:204-208,736-747; pp_bench.cxx:295-314: every repetition from the same factors)."""
    cp = ppals.CP(ctx, V, R)
    kw = dict(tol=1e-10 * vnorm, maxiter=300, resprint=10)
    dt_csv, pp_csv, pp1_csv, b_csv = (os.path.join(tmpdir, n) for n in
                                      ("dt.csv", "pp.csv", "pp1.csv", "bench.csv"))
    cp.set_factors(W0, G0)
    cp.run_dt(csv=dt_csv, **kw)
    cp.set_factors(W0, G0)
    cp.run_pp(csv=pp_csv, tol_init=0.01, **kw)
    dt, pp = _csv_rows(dt_csv), _csv_rows(pp_csv)
    floor = 1.02 * max(min(r[5] for r in dt), min(r[5] for r in pp))

    def first_at(rows):
        for r in rows:
            if r[5] <= floor:
                return {"iter": int(r[1]), "dtime_s": r[6]}
        return None
    # per-sweep split by [pp_update]: the same run with a row per iteration (each row costs a
    # stream sync, so these are upper bounds of the in-flight sweep times)
    cp.set_factors(W0, G0)
    cp.pp_build_stats(+1)            # time the operator builds of this run (a sync on both sides)
    cp.run_pp(csv=pp1_csv, tol_init=0.01, tol=1e-10 * vnorm, maxiter=300, resprint=1)
    n_builds, build_s = cp.pp_build_stats(-1)
    r1 = _csv_rows(pp1_csv)
    split = {"exact": [], "exact_plus_build": [], "pp_first_sweep": [], "pp": []}
    # A row is printed BEFORE the sweep of its iteration, so the interval to the next row is the
    # sweep of the row's own kind. The exact phase returns WITHOUT counting its last sweep
    # (als_CP.cxx:594-605), and the PP phase builds its operators BEFORE it prints its first row
    # (als_CP.cxx:667-697; engine.cpp pp_sub): the interval (last exact row) -> (first PP row) is
    # that exact sweep PLUS the operator build; every later interval that starts at a PP row is one
    # approximate sweep (the first of a phase listed separately).
    for k in range(1, len(r1)):
        a, b = r1[k - 1], r1[k]
        if a[4] == 0:
            kind = "exact" if b[4] == 0 else "exact_plus_build"
        else:
            kind = "pp_first_sweep" if (k >= 2 and r1[k - 2][4] == 0) else "pp"
        split[kind].append(b[6] - a[6])
    with open(b_csv, "w") as f:
        f.write("[timetype],[dtime]\n")
    reps = 5
    for _ in range(reps):
        cp.set_factors(W0, G0)
        cp.run_dt(tol=1e-10 * vnorm, maxiter=1, resprint=1, bench=1, csv=b_csv, csv_append=1)
    for _ in range(reps):
        cp.set_factors(W0, G0)
        cp.run_pp(tol=1e-10 * vnorm, tol_init=0.01, maxiter=1, resprint=1, bench=1, csv=b_csv,
                  csv_append=1)
    bl = _bench_lines(b_csv)
    cp.close()
    return {
        "config": "BASELINE configs[2]: CP order-4 s=200 R=10 -tensor r, -pp 1 -pp_res_tol 0.01 vs "
                  "-pp 0, 300 iterations each, -resprint 10, fp32 tensor storage",
        "dt_total_dtime_s": dt[-1][6], "pp_total_dtime_s": pp[-1][6],
        "pp_rows_flagged_pp_update": sum(1 for r in pp if r[4] == 1), "rows": len(pp),
        "diffV_floor": floor, "dt_first_at_floor": first_at(dt), "pp_first_at_floor": first_at(pp),
        "final_gradnorm": {"dt": dt[-1][2], "pp": pp[-1][2]},
        "intervals_per_s_by_kind": {
            k: (len(v) / sum(v) if v and sum(v) > 0 else None) for k, v in split.items()},
        "intervals_counted_by_kind": {k: len(v) for k, v in split.items()},
        "interval_kinds": "row-per-iteration run (each row costs a stream sync: upper bounds). exact = one "
                          "exact sweep; exact_plus_build = the uncounted last exact sweep of a DT phase + "
                          "the PP operator build, which runs before the first PP row is printed; "
                          "pp_first_sweep / pp = one approximate sweep",
        "pp_builds": n_builds,
        "pp_build_ms": (1e3 * build_s / n_builds) if n_builds else None,
        "pp_build_note": "host-timed with the stream synchronised on both sides of every build "
                         "(ppals_cp_pp_build_stats); inside a run one of the three level-1 tensors is the "
                         "multi-sweep intermediate the exact sweep just left: 2 tensor scans + the pair / "
                         "single operators",
        "bench_mode_ms": {"[DTtime]": 1e3 * _median(bl.get("[DTtime]", [])),
                          "[PPfirst]": 1e3 * _median(bl.get("[PPfirst]", [])),
                          "[PPsecond]": 1e3 * _median(bl.get("[PPsecond]", [])), "repetitions": reps,
                          "note": "medians; every line a cold, synchronised measurement from the "
                                  "same factors, as pp_bench.cxx:299-314"},
    }


def cfg5_tucker_record(ppals, ctx, tmpdir):
    """BASELINE configs[4]: Tucker order-3 s=400 core 20^3 on `-tensor r2` (test_ALS.cxx:272), HOSVD
    initialisation + 40 HOOI sweeps of alsTucker_DT (als_Tucker.cxx:240-424), fp32 storage."""
    lens, ranks = [400, 400, 400], [20, 20, 20]
    V = ppals.Tensor(ctx, lens, ppals.F32).fill_uniform(7)
    tk = ppals.Tucker(ctx, V, ranks)
    tk.hosvd()                       # warm-up of the one-off paths (workspaces, kernels' first launch)
    ctx.sync()
    t0 = time.perf_counter()
    tk.hosvd()                       # same session: the eigen-steps start from the first call's state
    ctx.sync()
    hosvd_warm_ms = 1e3 * (time.perf_counter() - t0)
    tk.close()
    tk = ppals.Tucker(ctx, V, ranks)  # a NEW session: nothing known about the three Grams
    ctx.sync()
    t0 = time.perf_counter()
    tk.hosvd()
    ctx.sync()
    hosvd_ms = 1e3 * (time.perf_counter() - t0)
    csv = os.path.join(tmpdir, "tucker.csv")
    tk.run_dt(tol=0.0, maxiter=40, resprint=10, csv=csv)
    rows = _csv_rows(csv)
    # steady state: sweeps 10..40 of the driver's own [dtime] column
    r10 = [r for r in rows if r[1] == 10][0]
    steady_ms = 1e3 * (rows[-1][6] - r10[6]) / (rows[-1][1] - r10[1])
    # kernel split of a sweep, HIP events on the engine's stream (untimed pass)
    n = 10
    ctx.profile_reset()
    ctx.profile_enable(2)
    tk.sweeps_dt(n)
    ctx.sync()
    ctx.profile_enable(0)
    ls, scan_ms, scan_bytes = ctx.profile_read(0)
    lg, gram_ms, _ = ctx.profile_read(1)
    rec = {
        "config": "BASELINE configs[4]: Tucker order-3 s=400 core 20x20x20, -tensor r2, hosvd + "
                  "alsTucker_DT 40 sweeps, fp32 tensor storage, 1 GPU",
        "hosvd_ms": hosvd_ms, "hosvd_repeated_in_one_session_ms": hosvd_warm_ms,
        "ms_per_hooi_sweep": steady_ms,
        "ms_per_hooi_sweep_incl_first_10": 1e3 * rows[-1][6] / rows[-1][1],
        "total_dtime_s_40_sweeps": rows[-1][6],
        "final_diffnorm": rows[-1][2], "final_diffV": rows[-1][5],
        "scan_ms_per_sweep": scan_ms / n, "scan_launches_per_sweep": ls / n,
        "gram_ms_per_sweep": gram_ms / n,
        "eigen_step_us": 1e3 * max(steady_ms - scan_ms / n - gram_ms / n, 0.0) / 3.0,
        "eigen_step_note": "(sweep - tensor scans - Grams) / 3 modes: the spectral-projector step "
                           "incl. its launches' gaps",
    }
    if ls > 0 and scan_ms > 0:
        ach = scan_bytes / (scan_ms * 1e-3) / 1e9
        rec["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": ach / HBM_PEAK_GBS, "launches": ls,
                           "avg_launch_ms": scan_ms / ls,
                           "algorithmic_bytes_per_launch": scan_bytes / ls,
                           "kernel": "tensor scans of the TTMc chain (K11)"}
    tk.close()
    V.close()
    return rec


def shard_probe_record(ppals, torch, local_rank):
    """Per-rank compute of the strong-scaled runs, measured on THIS one GPU: a [s/8, s, s, s] tensor is
    what each of 8 ranks holds of configs[1] / configs[3] (leading-mode block). The session runs the
    SHARDED code path — shard-aware multi-sweep schedule, packed partials, every collective issued on a
    one-rank RCCL communicator — so its sweep time is the floor of the 8-GPU run before a byte crosses
    xGMI (tools/shard_probe.py). A context of its own: PPALS_FORCE_COMM is read when sessions are made."""
    old = os.environ.get("PPALS_FORCE_COMM")
    os.environ["PPALS_FORCE_COMM"] = "1"
    out = {"note": "one GPU, one-rank RCCL communicator, PPALS_FORCE_COMM=1: the rank-local sweep of a P = 8 "
                   "run (no exchange partner); projected efficiency = (P=1 sweep / 8) / this"}
    try:
        ctx = ppals.Context(local_rank)
        ctx.init_comm(0, 1, ppals.Context.unique_id())
        for name, s, R, K in (("cfg2_P8", 200, 10, 24), ("cfg4_P8", 400, 20, 8)):
            lens = [s // 8, s, s, s]
            V = ppals.Tensor(ctx, lens, ppals.F32).fill_cp(ppals.init_factors(lens, R, 1000))
            cp = ppals.CP(ctx, V, R)
            cp.set_factors(ppals.init_factors(lens, R, 2000), ppals.init_factors(lens, R, 3000))
            cp.sweeps_dt(4)
            ctx.sync()
            t0 = time.perf_counter()
            cp.sweeps_dt(K)          # (a multiple of 2 sweeps: the sharded cycle is 3 scans / 2 sweeps)
            ctx.sync()
            dt = (time.perf_counter() - t0) / K
            out[name] = {"shard": lens, "rank": R, "ms_per_sweep_per_rank": 1e3 * dt, "sweeps": K,
                         "schedule": cp.schedule, "rccl_ranks": ctx.nranks}
            cp.close()
            V.close()
        ctx.close()
    except Exception as e:  # reported, never required
        out["error"] = str(e)
    if old is None:
        del os.environ["PPALS_FORCE_COMM"]
    else:
        os.environ["PPALS_FORCE_COMM"] = old
    return out


def sharded_config_records(ppals, ctx, torch, dist, dev, hostsim, world, measure, barrier):
    """N > 1: BASELINE configs[3] (CP order-4 s=400 R=20 block-partitioned over the N ranks, both
    collective plans: script/script_strongscaling.py:10,45-46 is the reference's analogue) and the
    N-GPU leg of configs[4] (Tucker order-3 s=400 core 20^3 sharded: hosvd + alsTucker_DT sweeps,
    als_Tucker.cxx:12-70,240-424). Times are max over ranks between barriers; the scan roofline is
    per rank (its own shard's bytes; the slowest rank's average launch). The CPU rehearsal
    (PPALS_BENCH_BACKEND=hostsim) runs the same code on shrunken shapes."""
    out = {}

    def max_over_ranks(x, op=None):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=op or dist.ReduceOp.MAX)
        return float(t.item())

    def timed(fn):
        barrier()
        t0 = time.perf_counter()
        fn()
        ctx.sync()
        if not hostsim:
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return max_over_ranks(dt)

    def agree(ok):
        """every rank learns whether ALL ranks got through the stage just ended (a stage that can fail
        — an allocation, a session set-up — holds no collective; the sweeps that follow do, so a rank
        that failed must not leave the others inside them): all ranks skip the rest of a record together"""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    def guarded(make):
        """(object, error) of a collective-free construction, agreed over the ranks"""
        obj, err = None, None
        try:
            obj = make()
        except Exception as e:  # reported, never required
            err = str(e)
        if not agree(err is None):
            if obj is not None:
                obj.close()
            return None, err or "another rank failed this stage"
        return obj, None

    # ---- configs[3]
    lens4, R4 = ([16, 12, 12, 12], 3) if hostsim else WORKLOADS["cp4_s400_r20"]
    f32 = ppals.F64 if hostsim else ppals.F32
    V4, err = guarded(lambda: ppals.Tensor(ctx, lens4, f32).fill_cp(ppals.init_factors(lens4, R4, 1000)))
    if V4 is None:
        out["cfg4_sharded"] = {"error": err}
    else:
        try:
            W4, G4 = ppals.init_factors(lens4, R4, 2000), ppals.init_factors(lens4, R4, 3000)
            rec = {"config": f"BASELINE configs[3]: CP order-4 s={lens4[0]} R={R4} -tensor r, leading-mode blocks "
                             f"over {world} ranks, 2 warm-up + 5 timed exact sweeps per collective plan",
                   "rccl_ranks": ctx.nranks, "dtype": "f64" if hostsim else "f32"}
            for plan, small in (("allreduce_plan", None), ("reduce_scatter_plan", "0")):
                old = os.environ.get("PPALS_COMM_SMALL_BYTES")
                if small is not None:
                    os.environ["PPALS_COMM_SMALL_BYTES"] = small
                cp4 = None
                try:
                    cp4, err = guarded(lambda: ppals.CP(ctx, V4, R4))
                    if cp4 is None:
                        rec[plan] = {"error": err}
                        continue
                    r = measure(cp4, 5, 2, W4, G4)
                    r["sweep_flops"] = sweep_flops(lens4, R4, cp4.schedule)
                    r["mttkrp_tflops"] = r["sweep_flops"] * r["value"] / 1e12
                    r["comm_plan"] = ("one all-reduce of the s x R partials per mode + redundant fused update"
                                      if small is None else
                                      "reduce-scatter + row-block update + all-gather per mode (the plan north_star names)")
                    if "roofline" in r:
                        rl = r["roofline"]
                        rl["avg_launch_ms_slowest_rank"] = max_over_ranks(rl["avg_launch_ms"])
                        rl["avg_launch_ms_fastest_rank"] = max_over_ranks(rl["avg_launch_ms"], dist.ReduceOp.MIN)
                        rl["frac_slowest_rank"] = (rl["algorithmic_bytes_per_launch"] /
                                                   (rl["avg_launch_ms_slowest_rank"] * 1e-3) / 1e9 / HBM_PEAK_GBS)
                        rl["note"] = "rank 0's shard: bytes of ITS scans / ITS average launch; *_slowest_rank over all ranks"
                    r["final_gradnorm"] = cp4.gradnorm()
                    rec[plan] = r
                finally:
                    if cp4 is not None:
                        cp4.close()
                    if small is not None:
                        if old is None:
                            del os.environ["PPALS_COMM_SMALL_BYTES"]
                        else:
                            os.environ["PPALS_COMM_SMALL_BYTES"] = old
            out["cfg4_sharded"] = rec
        finally:
            V4.close()

    # ---- configs[4], N-GPU leg
    lens5, ranks5, nsw = ([16, 12, 10], [3, 3, 3], 4) if hostsim else ([400, 400, 400], [20, 20, 20], 20)
    V5, err = guarded(lambda: ppals.Tensor(ctx, lens5, f32).fill_uniform(7))
    tk = None
    if V5 is None:
        out["cfg5_tucker_sharded"] = {"error": err}
    else:
        try:
            tk, err = guarded(lambda: ppals.Tucker(ctx, V5, ranks5))
            if tk is None:
                raise RuntimeError(err)
            tk.hosvd()                      # warm-up of the one-off paths
            tk.close()
            tk, err = guarded(lambda: ppals.Tucker(ctx, V5, ranks5))   # a new session: nothing known about the Grams
            if tk is None:
                raise RuntimeError(err)
            hosvd_s = timed(tk.hosvd)
            tk.sweeps_dt(2)
            ctx.profile_reset()
            ctx.profile_enable(1)
            sweeps_s = timed(lambda: tk.sweeps_dt(nsw))
            ctx.profile_enable(0)
            ls, scan_ms, scan_bytes = ctx.profile_read(0)
            rec = {"config": f"BASELINE configs[4] on {world} ranks: Tucker order-3 s={lens5[0]} core "
                             f"{'x'.join(map(str, ranks5))}, -tensor r2, leading-mode blocks; hosvd + 2 warm-up + "
                             f"{nsw} timed HOOI sweeps (alsTucker_DT)",
                   "rccl_ranks": ctx.nranks, "dtype": "f64" if hostsim else "f32",
                   "hosvd_ms": 1e3 * hosvd_s, "ms_per_hooi_sweep": 1e3 * sweeps_s / nsw, "sweeps": nsw}
            if ls > 0 and scan_ms > 0:
                avg = scan_ms / ls
                rec["roofline"] = {"bound": "hbm", "achieved": scan_bytes / ls / (avg * 1e-3) / 1e9,
                                   "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": scan_bytes / ls / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "launches": ls, "avg_launch_ms": avg,
                                   "avg_launch_ms_slowest_rank": max_over_ranks(avg),
                                   "algorithmic_bytes_per_launch": scan_bytes / ls,
                                   "kernel": "tensor scans of the TTMc chain (K11), rank 0's shard"}
            out["cfg5_tucker_sharded"] = rec
        except Exception as e:  # reported, never required (a failure agreed by all ranks)
            out["cfg5_tucker_sharded"] = {"error": str(e)}
        finally:
            if tk is not None:
                tk.close()
            V5.close()
    return out


def real_shape_main(args):
    """`--workload coil100|timelapse`: the reference's real-data EXTENTS (3 x 128 x 128 x 7200 and
    33 x 1344 x 1024 x 9) filled with synthetic image-like values (a decaying rank-100 model), one GPU: CP R = 10 `-pp 0` (sweeps/s,
    tensor-scan roofline, which scans the schedule ran), `-pp 1 -pp_res_tol 0.05` against `-pp 0` over
    the scripts' 250 iterations, Tucker with the ranks of test_ALS.cxx:366-379 (HOSVD + HOOI sweeps).
    One JSON line; per-root scan times come from a kernel trace of this command (tools/runs/r05_real.sh)."""
    import tempfile
    import torch  # noqa: F401  (first: libppals.so then shares torch's HIP runtime)
    import ppals
    lens, R, tranks = REAL_SHAPES[args.workload]
    ppals.preload_eigensolver()
    ctx = ppals.Context(int(os.environ.get("LOCAL_RANK", "0")))
    dtype = ppals.F32 if args.dtype == "f32" else ppals.F64
    esz = 4 if args.dtype == "f32" else 8
    # image-like synthetic values: a rank-100 CP model whose weights decay geometrically (every unfolding
    # has a decaying spectrum with a gap below any rank, as photographs do; U(0.5,1) noise — `-tensor r2`
    # — has a flat spectrum below its mean component, the worst case for the Tucker eigen-steps and not
    # what these extents hold in the reference's runs)
    import numpy as np
    Wt = ppals.init_factors(lens, 100, 1000)
    decay = (0.93 ** np.arange(100)) ** (1.0 / len(lens))
    Wt = [np.asfortranarray(w * decay[None, :]) for w in Wt]
    V = ppals.Tensor(ctx, lens, dtype).fill_cp(Wt)
    vnorm = V.norm()
    W0, G0 = ppals.init_factors(lens, R, 2000), ppals.init_factors(lens, R, 3000)
    out = {"metric": f"ALS sweeps/sec (exact sweep, CP order-4 {'x'.join(map(str, lens))} R={R})",
           "unit": "sweeps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": args.dtype,
           "data": "synthetic (rank-100 CP model with geometrically decaying weights, in the reference's "
                   "real-data extents)",
           "config": {"workload": f"{args.workload}: extents of test_ALS.cxx:287-326 ({lens}), CP R={R} -pp 0 exact "
                                  "sweeps incl. Normalize; not a BASELINE config", "lens": lens, "rank": R}}
    sub = {}
    with tempfile.TemporaryDirectory() as tmp:
        trace = os.path.join(tmp, "steps.txt")
        os.environ["PPALS_TRACE_STEPS"] = trace
        cp = ppals.CP(ctx, V, R)
        if args.schedule:
            cp.set_schedule(args.schedule)
        cp.set_factors(W0, G0)
        cp.sweeps_dt(args.warmup)
        ctx.sync()
        open(trace, "w").close()
        ctx.profile_reset()
        ctx.profile_enable(1)
        t0 = time.perf_counter()
        cp.sweeps_dt(args.steps)
        ctx.sync()
        el = time.perf_counter() - t0
        ctx.profile_enable(0)
        del os.environ["PPALS_TRACE_STEPS"]
        launches, scan_ms, scan_bytes = ctx.profile_read(0)
        out["value"] = args.steps / el
        out["ms_per_step"] = 1e3 * el / args.steps
        out["schedule"] = cp.schedule
        # which first-level scans the schedule ran in the timed sweeps (root set, layout, shape)
        scans = {}
        for ln in open(trace).read().splitlines():
            kv = dict(t.split("=") for t in ln.split() if "=" in t)
            key = (kv["root"], kv["k"], kv["layout"], kv["L"], kv["J"], kv["T"])
            scans[key] = scans.get(key, 0) + 1
        out["scans"] = [{"root": int(k[0]), "modes_contracted": int(k[1]), "layout": k[2], "L": int(k[3]),
                         "J": int(k[4]), "T": int(k[5]), "launches_in_timed_region": n,
                         "algorithmic_bytes": int(k[3]) * int(k[4]) * int(k[5]) * esz
                                              + int(k[3]) * int(k[5]) * R * esz}
                        for k, n in sorted(scans.items())]
        if launches > 0 and scan_ms > 0:
            ach = scan_bytes / (scan_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "traffic": None, "launches": launches,
                               "avg_launch_ms": scan_ms / launches,
                               "algorithmic_bytes_per_launch": scan_bytes / launches,
                               "scan_ms_per_step": scan_ms / args.steps,
                               "note": "all first-level tensor scans of the timed sweeps together (HIP events on the "
                                       "engine's stream); per root: the kernel trace of this command"}
        out["final_gradnorm"] = cp.gradnorm()
        # per root: the online placement choice times every root's own scans (PPALS_PLACE_MIN_MB lowers its
        # size threshold below these 1.4 / 1.6 GB tensors) — run until it has, then price each root
        try:
            for _ in range(15):
                rep = cp.placement_report()
                if rep.get("mode") != "online" or not rep["roots"] or all(r["settled"] for r in rep["roots"]):
                    break
                cp.sweeps_dt(8)
            ctx.sync()
            rep = cp.placement_report()
            out["placement"] = rep
            for sc in out["scans"]:
                for r in rep.get("roots", []):
                    if r["root"] == sc["root"] and r["best_ms"] > 0:
                        sc["best_ms"], sc["worst_ms"] = r["best_ms"], r["worst_ms"]
                        sc["frac_of_hbm_peak"] = sc["algorithmic_bytes"] / (r["best_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        except Exception as e:
            out["placement"] = {"error": str(e)}
        if args.no_config_records:    # (the exact sweeps only: what a kernel trace of this command should show)
            cp.close()
            print(json.dumps(out), flush=True)
            V.close()
            ctx.close()
            return
        # -pp 1 against -pp 0 over the scripts' 250 iterations (script/script_real.py:42-47)
        try:
            kw = dict(tol=1e-10 * vnorm, maxiter=250, resprint=10)
            dt_csv, pp_csv = os.path.join(tmp, "dt.csv"), os.path.join(tmp, "pp.csv")
            cp.set_factors(W0, G0)
            cp.run_dt(csv=dt_csv, **kw)
            cp.set_factors(W0, G0)
            cp.run_pp(csv=pp_csv, tol_init=0.05, **kw)
            dt, pp = _csv_rows(dt_csv), _csv_rows(pp_csv)
            sub["cp_pp1_vs_pp0"] = {
                "config": "-model CP -pp 0 / -pp 1 -pp_res_tol 0.05 -rank 10 -maxiter 250 -resprint 10",
                "dt_total_dtime_s": dt[-1][6], "pp_total_dtime_s": pp[-1][6],
                "dt_iters": int(dt[-1][1]), "pp_iters": int(pp[-1][1]),
                "pp_rows_flagged_pp_update": sum(1 for r in pp if r[4] == 1), "rows": len(pp),
                "final_diffV": {"dt": dt[-1][5], "pp": pp[-1][5]}, "vnorm": vnorm}
        except Exception as e:
            sub["cp_pp1_vs_pp0"] = {"error": str(e)}
        cp.close()
        # Tucker with the reference's ranks for this data set
        try:
            tk = ppals.Tucker(ctx, V, tranks)
            ctx.sync()
            t0 = time.perf_counter()
            tk.hosvd()
            ctx.sync()
            hosvd_ms = 1e3 * (time.perf_counter() - t0)
            csv = os.path.join(tmp, "tucker.csv")
            tk.run_dt(tol=0.0, maxiter=20, resprint=5, csv=csv)
            rows = _csv_rows(csv)
            r5 = [r for r in rows if r[1] == 5][0]
            ctx.profile_reset()
            ctx.profile_enable(2)
            tk.sweeps_dt(5)
            ctx.sync()
            ctx.profile_enable(0)
            ls, sms, sby = ctx.profile_read(0)
            rec = {"config": f"-model Tucker -tensor o? extents {lens}, core {tranks} (test_ALS.cxx:366-379), "
                             "hosvd + alsTucker_DT 20 sweeps",
                   "hosvd_ms": hosvd_ms,
                   "ms_per_hooi_sweep": 1e3 * (rows[-1][6] - r5[6]) / (rows[-1][1] - r5[1]),
                   "scan_launches_per_sweep": ls / 5.0, "scan_ms_per_sweep": sms / 5.0,
                   "final_diffV": rows[-1][5]}
            if ls > 0 and sms > 0:
                rec["roofline"] = {"bound": "hbm", "achieved": sby / (sms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": sby / (sms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                   "launches": ls, "avg_launch_ms": sms / ls}
            sub["tucker"] = rec
            tk.close()
        except Exception as e:
            sub["tucker"] = {"error": str(e)}
    out["sub_records"] = sub
    print(json.dumps(out), flush=True)
    V.close()
    ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cp4_s200_r10", choices=sorted(WORKLOADS) + sorted(REAL_SHAPES))
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"],
                    help="storage type of the tensor in HBM (all factor math is fp64)")
    ap.add_argument("--schedule", default=None, choices=["dt", "msdt"],
                    help="sweep schedule (default: the engine's, msdt); same ALS iterates either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config-records", action="store_true",
                    help="skip the cfg3 (PP) / cfg4 (s=400) / cfg5 (Tucker) sub_records (N = 1) and the sharded "
                         "cfg4 / cfg5 sub_records (N > 1)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # live_pmc_traffic's child runs
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not count HBM bytes with rocprofv3 --pmc child passes (roofline.traffic = null)")
    args = ap.parse_args()

    if args.workload in REAL_SHAPES:
        real_shape_main(args)
        return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    # HBM traffic of the scan kernel, counted by child passes under rocprofv3 --pmc BEFORE this
    # process initialises the GPU (N = 1, the real engine only)
    live_traffic, live_traffic_src = None, "not collected"
    # (never from inside a profiled run: a profiler's preloaded library has initialised the GPU
    # already and its environment would be inherited by the children)
    profiled = ("rocprof" in os.environ.get("LD_PRELOAD", "").lower()
                or any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ))
    if profiled:
        live_traffic_src = "not collected: this run is itself under a profiler"
    elif (world == 1 and not args.no_pmc and os.environ.get("PPALS_BENCH_BACKEND") != "hostsim"
            and os.environ.get("PPALS_FORCE_COMM", "0") != "1"):
        live_traffic, live_traffic_src = live_pmc_traffic(args)

    import numpy as np
    import torch  # first: libppals.so then shares torch's libamdhip64 / librccl
    # PPALS_BENCH_BACKEND=hostsim (tests/test_bench_hostsim.py, CPU only): THIS script's N > 1 code path
    # — process group, communicator id hand-over, init_comm, measure(), the reduce-scatter sub-record —
    # over the engine's host stand-in and gloo. Test infrastructure: the line it prints says so and
    # is never a measurement.
    hostsim = os.environ.get("PPALS_BENCH_BACKEND") == "hostsim"
    if hostsim:
        import hostsim_util
        ppals = hostsim_util.load()
    else:
        import ppals
    dev = "cpu" if hostsim else "cuda"
    config_records = (int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_config_records
                      and args.workload == "cp4_s200_r10" and args.dtype == "f32"
                      and not args.schedule)
    # N > 1: the sharded records of configs[3] / configs[4] (the driver's scaling runs use the default flags)
    # (PPALS_FORCE_COMM=1: the same records on a one-rank communicator — the rehearsal on a 1-GPU box)
    sharded_records = ((int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("PPALS_FORCE_COMM", "0") == "1")
                       and not args.no_config_records
                       and not args.schedule
                       and ((args.workload == "cp4_s200_r10" and args.dtype == "f32") or hostsim))
    if (config_records or sharded_records) and not hostsim:
        # the Tucker record's HOSVD uses rocSOLVER: its libraries must enter the process before the
        # HIP runtime is up (include/ppals.h, ppals_preload_eigensolver)
        ppals.preload_eigensolver()

    if not hostsim:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
        torch.cuda.set_device(local_rank)
    dist = None
    # PPALS_FORCE_COMM=1 (tests): take the N > 1 route — process group, unique-id broadcast, RCCL
    # communicator, sharded engine paths — even with a single rank
    use_comm = world > 1 or os.environ.get("PPALS_FORCE_COMM", "0") == "1"
    if use_comm:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import torch.distributed as dist
        if hostsim:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    lens, R = WORKLOADS[args.workload]
    dtype = ppals.F32 if args.dtype == "f32" else ppals.F64
    ctx = ppals.Context(0 if hostsim else local_rank)
    comm_keepalive = None
    if use_comm:
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(ppals.Context.unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)   # every rank joins the communicator rank 0 named
        if hostsim:
            # (the stand-in's "id" is a set of callbacks into this process's gloo group: made locally)
            local_uid, comm_keepalive, _ = hostsim_util.gloo_comm_uid(rank, world)
            ctx.init_comm(rank, world, local_uid)
        else:
            ctx.init_comm(rank, world, bytes(uid.cpu().numpy().tobytes()))

    # ppals_ctx_nranks: the communicator's own answer (RCCL: ncclCommCount, checked against the
    # requested world size when the communicator is made, rccl_comm.cpp); 1 without a communicator
    nranks_seen = ctx.nranks
    Wtrue = ppals.init_factors(lens, R, 1000)
    W0 = ppals.init_factors(lens, R, 2000)
    G0 = ppals.init_factors(lens, R, 3000)

    def barrier():
        ctx.sync()
        if not hostsim:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def _clocks_ns():
        out = {"monotonic": time.monotonic_ns(), "realtime": time.time_ns()}
        for name in ("CLOCK_BOOTTIME", "CLOCK_MONOTONIC_RAW"):
            if hasattr(time, name):
                out[name.lower()[6:]] = time.clock_gettime_ns(getattr(time, name))
        return out

    def measure(cp, steps, warmup, W=None, G=None):
        """W untimed sweeps, then EXACTLY `steps` timed ones between barrier + synchronize on both
        sides (max over ranks); HIP events on the engine's stream around the tensor scans only"""
        cp.set_factors(W if W is not None else W0, G if G is not None else G0)
        cp.sweeps_dt(warmup)
        barrier()
        ctx.profile_reset()
        ctx.profile_enable(1)
        clk0 = _clocks_ns()
        t0 = time.perf_counter()
        cp.sweeps_dt(steps)
        ctx.sync()
        if not hostsim:
            torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        clk1 = _clocks_ns()
        ctx.profile_enable(0)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        barrier()
        launches, scan_ms, scan_bytes = ctx.profile_read(0)
        rec = {"ms_per_step": 1e3 * elapsed / steps, "value": steps / elapsed, "steps": steps,
               "schedule": cp.schedule,
               # host clocks at both ends of the timed region: tools/trace_timed_launches.py picks
               # this region's launches out of a rocprofv3 kernel trace of the same command
               "timed_region_ns": {k: [clk0[k], clk1[k]] for k in clk0}}
        if launches > 0 and scan_ms > 0:
            avg_ms = scan_ms / launches
            achieved = (scan_bytes / launches) / (avg_ms * 1e-3) / 1e9
            rec["roofline"] = {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "launches": launches, "avg_launch_ms": avg_ms,
                "algorithmic_bytes_per_launch": scan_bytes / launches,
                "scan_ms_per_step": scan_ms / steps, "scan_launches_per_step": launches / steps}
        return rec

    V = ppals.Tensor(ctx, lens, dtype).fill_cp(Wtrue)
    t_setup0 = time.perf_counter()
    cp = ppals.CP(ctx, V, R)
    ctx.sync()
    session_setup_s = time.perf_counter() - t_setup0
    if args.schedule:
        cp.set_schedule(args.schedule)
    schedule = cp.schedule
    # The engine chooses where each root's first-level intermediate lies ONLINE: the first ~20 visits
    # of a root run the sweep's own scan at a different candidate (engine.h, PlaceExplore). The
    # headline is the steady state, so the session is run until every root has settled — untimed, like
    # the warm-up, and reported: `placement.settle_sweeps` / `settle_s`; what the choice is worth on a
    # run of the reference's length, exploration included, is sub_records.time_to_250_sweeps.
    settle_sweeps, settle_s = 0, 0.0
    if args.pmc_child:
        # the counted pass: the sweeps of the workload and nothing else, so that the counter rows are
        # the scan launches of exactly (warmup + steps) sweeps
        measure(cp, args.steps, args.warmup)
        cp.close()
        V.close()
        ctx.close()
        return
    # (on real factors: the session's own are still zero, and sweeps on a singular S would time the
    # solve's fallback path; at most 24 sweeps — the engine's evidence gate ends an exploration that
    # finds nothing after 8 samples per root, and whatever is still exploring then simply goes on inside
    # the timed region: an exploring visit IS the sweep's own scan)
    cp.set_factors(W0, G0)
    t0 = time.perf_counter()
    while settle_sweeps < 24:
        try:
            rep = cp.placement_report()
            done = (rep.get("mode") != "online" or (rep["roots"] and all(r["settled"] for r in rep["roots"]))
                    or (not rep["roots"] and settle_sweeps >= 8))   # (nothing to choose for this shape)
        except Exception:
            done = True
        if dist is not None:
            # every rank runs the SAME number of sweeps (they contain collectives): go on until all are done
            flag = torch.tensor([1 if done else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            done = bool(flag.item())
        if done:
            break
        cp.sweeps_dt(8)
        settle_sweeps += 8
    ctx.sync()
    settle_s = time.perf_counter() - t0
    head = measure(cp, args.steps, args.warmup)
    try:
        placement = cp.placement_report()
        placement["session_setup_s"] = session_setup_s
        placement["settle_sweeps"] = settle_sweeps
        placement["settle_s"] = settle_s
    except Exception as e:  # reported, never required
        placement = {"error": str(e)}
    gradnorm = cp.gradnorm()
    resid = cp.residual()
    vnorm = V.norm()
    # untimed breakdown pass: a few more sweeps with every bracketed kernel under events (each
    # event pair idles the stream for ~10 us, which is why the timed region brackets scans only)
    extra = 6
    ctx.profile_reset()
    ctx.profile_enable(2)
    cp.sweeps_dt(extra)
    barrier()
    ctx.profile_enable(0)
    _, other_ms, _ = ctx.profile_read(1)
    other_ms_per_step = other_ms / extra

    # ---- sub-records beside the headline (same JSON line, never `value`): the reference's own
    # two-scan schedule (SURVEY §8d counts ITS bytes: 2 s^4 sizeof per sweep) and the reference's
    # own precision (fp64 storage) at N = 1; the other shard plan at N > 1
    sub = {}
    sub_steps = max(4, min(args.steps, 10))
    if world == 1 and not args.schedule and not args.workload.startswith("cp4_s400"):
        other = "dt" if schedule == "msdt" else "msdt"
        cp.set_schedule(other)
        r = measure(cp, sub_steps, 2)
        r["dtype"] = args.dtype
        r["sweep_flops"] = sweep_flops(lens, R, other)
        r["mttkrp_tflops"] = r["sweep_flops"] * r["value"] / 1e12
        sub[f"{other}_schedule_{args.dtype}"] = r
        cp.set_schedule(schedule)
    if world == 1 and not args.schedule and schedule == "msdt":
        # the SAME measurement in this process with the placement measurement switched off
        # (PPALS_PLACE_TUNE is read when a session is created): does the set-up's choice of result
        # blocks / offsets / store kinds pay on THIS box?
        old = os.environ.get("PPALS_PLACE_TUNE")
        os.environ["PPALS_PLACE_TUNE"] = "0"
        try:
            t0 = time.perf_counter()
            cpu_ = ppals.CP(ctx, V, R)
            ctx.sync()
            r = measure(cpu_, args.steps, args.warmup)
            r["session_setup_s"] = time.perf_counter() - t0 - r["ms_per_step"] * 1e-3 * (args.steps + args.warmup)
            r["note"] = ("PPALS_PLACE_TUNE=0: first-level intermediates at offset 0 of one block, store kind "
                         "by size; same tensor, same process, measured right after the headline")
            sub["untuned"] = r
            cpu_.close()
        except Exception as e:
            sub["untuned"] = {"error": str(e)}
        if old is None:
            del os.environ["PPALS_PLACE_TUNE"]
        else:
            os.environ["PPALS_PLACE_TUNE"] = old
    if world == 1 and not args.schedule:
        # cold (first sweep) beside the steady state — a fresh session on the resident
        # tensor: session creation (second resident layout, workspaces), then the very first exact sweep
        try:
            t0 = time.perf_counter()
            cpc = ppals.CP(ctx, V, R)
            cpc.set_factors(W0, G0)
            ctx.sync()
            t1 = time.perf_counter()
            cpc.sweeps_dt(1)
            ctx.sync()
            t2 = time.perf_counter()
            cpc.sweeps_dt(1)
            ctx.sync()
            t3 = time.perf_counter()
            sub["cold"] = {"session_create_ms": 1e3 * (t1 - t0), "first_sweep_ms": 1e3 * (t2 - t1),
                           "second_sweep_ms": 1e3 * (t3 - t2), "steady_ms_per_step": head["ms_per_step"],
                           "note": "fresh session on the resident tensor; the first sweep allocates the first-level "
                                   "intermediate and packs its operands for the first time"}
            cpc.close()
        except Exception as e:  # reported, never required
            sub["cold"] = {"error": str(e)}
    if world == 1 and not args.schedule and schedule == "msdt" and not hostsim:
        # wall time from session creation to 250 sweeps done (the reference's scripts run 250,
        # script/script_synthetic.py:18) with the online placement choice and without, same tensor,
        # same process, alternating, the faster of two runs each
        try:
            t250 = {"tuned_s": [], "untuned_s": []}
            old = os.environ.get("PPALS_PLACE_TUNE")
            for rep_ in range(2):
                for name, env in (("untuned_s", "0"), ("tuned_s", "1")):
                    os.environ["PPALS_PLACE_TUNE"] = env
                    ctx.sync()
                    t0 = time.perf_counter()
                    c_ = ppals.CP(ctx, V, R)
                    c_.set_factors(W0, G0)
                    c_.sweeps_dt(250)
                    ctx.sync()
                    t250[name].append(time.perf_counter() - t0)
                    c_.close()
            if old is None:
                del os.environ["PPALS_PLACE_TUNE"]
            else:
                os.environ["PPALS_PLACE_TUNE"] = old
            sub["time_to_250_sweeps"] = {
                "tuned_s": min(t250["tuned_s"]), "untuned_s": min(t250["untuned_s"]),
                "all_runs": t250,
                "note": "ppals_cp_create (second resident layout included) + set_factors + 250 exact sweeps + "
                        "sync, wall clock; tuned = online placement choice (explores during the first ~60 "
                        "sweeps), untuned = PPALS_PLACE_TUNE=0"}
        except Exception as e:
            sub["time_to_250_sweeps"] = {"error": str(e)}
    if world > 1:
        # the plan north_star names (reduce-scatter of the s x R partials + row-block solve +
        # all-gather) beside the default for these message sizes (one all-reduce + redundant fused
        # update): PPALS_COMM_SMALL_BYTES moves the switch, read when a session is created
        cp.close()
        old = os.environ.get("PPALS_COMM_SMALL_BYTES")
        os.environ["PPALS_COMM_SMALL_BYTES"] = "0"
        cp = ppals.CP(ctx, V, R)
        if args.schedule:
            cp.set_schedule(args.schedule)
        r = measure(cp, sub_steps, 2)
        r["comm_plan"] = "reduce-scatter + row-block update + all-gather per mode (the plan north_star names; the headline runs the default for these message sizes: one all-reduce + redundant fused update)"
        sub["reduce_scatter_plan"] = r
        if old is None:
            del os.environ["PPALS_COMM_SMALL_BYTES"]
        else:
            os.environ["PPALS_COMM_SMALL_BYTES"] = old
    cp.close()
    if sharded_records:
        if rank == 0:   # the optional records contain collectives: what is already measured is not lost with them
            print("[bench] headline before the sharded config records: " +
                  json.dumps({"value": head["value"], "ms_per_step": head["ms_per_step"], "n_gpus": world}),
                  file=sys.stderr, flush=True)
        sub.update(sharded_config_records(ppals, ctx, torch, dist, dev, hostsim, world, measure, barrier))
    if config_records:
        import tempfile
        with tempfile.TemporaryDirectory() as tmpdir:
            try:
                sub["cfg3_pp"] = cfg3_pp_record(ppals, ctx, V, R, W0, G0, vnorm, tmpdir)
            except Exception as e:  # reported, never required
                sub["cfg3_pp"] = {"error": str(e)}
    if config_records and world == 1 and args.dtype == "f32" and not args.schedule and args.workload == "cp4_s200_r10":
        # the reference CLI's DEFAULT rank is s/2 (test_ALS.cxx:119-125): past 64 columns the scan is
        # bound by the fp32 matrix cores, not by HBM — one pass per 128 columns (k_scan_wide)
        try:
            R100 = 100
            cpw = ppals.CP(ctx, V, R100)
            cpw.set_factors(ppals.init_factors(lens, R100, 2000), ppals.init_factors(lens, R100, 3000))
            cpw.sweeps_dt(2)
            ctx.sync()
            t0w = time.perf_counter()
            cpw.sweeps_dt(6)
            ctx.sync()
            ms_w = 1e3 * (time.perf_counter() - t0w) / 6
            ctx.profile_reset()
            ctx.profile_enable(1)
            cpw.sweeps_dt(4)
            ctx.sync()
            ctx.profile_enable(0)
            lw, scan_ms_w, _ = ctx.profile_read(0)
            cols = 16 * ((R100 + 15) // 16)
            rec = {"config": "CP order-4 s=200 R=100 (-rank s/2, the reference CLI's default), -tensor values of "
                             "the headline tensor, fp32 storage, exact sweeps",
                   "ms_per_sweep": ms_w, "scan_launches_per_sweep": lw / 4, "schedule": cpw.schedule}
            if lw > 0 and scan_ms_w > 0:
                avg = scan_ms_w / lw
                tf = 2.0 * float(np.prod(lens)) * cols / (avg * 1e-3) / 1e12
                rec["roofline"] = {"bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s",
                                   "frac": tf / 157.3, "avg_launch_ms": avg, "launches": lw,
                                   "kernel": "k_scan_wide<7> (one pass for 7 n-tiles of 16 columns; executed flops "
                                             "= 2 x elements x 112 columns, 100 of them useful)",
                                   "useful_tflops": 2.0 * float(np.prod(lens)) * R100 / (avg * 1e-3) / 1e12}
            cpw.close()
            sub["rank100"] = rec
        except Exception as e:  # reported, never required
            sub["rank100"] = {"error": str(e)}
    V.close()
    if world == 1 and args.dtype == "f32" and not args.schedule and args.workload == "cp4_s200_r10":
        V64 = ppals.Tensor(ctx, lens, ppals.F64).fill_cp(Wtrue)
        cp64 = ppals.CP(ctx, V64, R)
        r = measure(cp64, sub_steps, 2)
        r["dtype"] = "f64"
        if "roofline" in r:
            r["roofline"]["traffic"] = None
            r["roofline"]["traffic_replayed"] = _replayed_traffic("cp4_s200_r10/f64/1")
        r["sweep_flops"] = sweep_flops(lens, R, cp64.schedule)
        r["mttkrp_tflops"] = r["sweep_flops"] * r["value"] / 1e12
        sub[f"{cp64.schedule}_schedule_f64"] = r
        cp64.close()
        V64.close()

    if config_records:
        import tempfile
        with tempfile.TemporaryDirectory() as tmpdir:
            for name, fn in (("cfg5_tucker", lambda: cfg5_tucker_record(ppals, ctx, tmpdir)),):
                try:
                    sub[name] = fn()
                except Exception as e:  # reported, never required
                    sub[name] = {"error": str(e)}
        # configs[3] on ONE GPU (the 8-GPU curve is the driver's --gpus runs of this script)
        try:
            lens4, R4 = WORKLOADS["cp4_s400_r20"]
            V4 = ppals.Tensor(ctx, lens4, ppals.F32).fill_cp(ppals.init_factors(lens4, R4, 1000))
            cp4 = ppals.CP(ctx, V4, R4)
            r = measure(cp4, 5, 2, ppals.init_factors(lens4, R4, 2000),
                        ppals.init_factors(lens4, R4, 3000))
            r["config"] = ("BASELINE configs[3] on 1 GPU: CP order-4 s=400 R=20 -tensor r, 102 GB "
                           "fp32 + the second resident layout")
            r["dtype"] = "f32"
            r["sweep_flops"] = sweep_flops(lens4, R4, cp4.schedule)
            r["mttkrp_tflops"] = r["sweep_flops"] * r["value"] / 1e12
            if "roofline" in r:
                r["roofline"]["kernel"] = "k_scan_suffix_fast<float,2,5> (two n-tiles, non-temporal result stores)"
                r["roofline"]["traffic"] = None
                r["roofline"]["traffic_replayed"] = _replayed_traffic("cp4_s400_r20/f32/1")
            sub["cfg4_1gpu"] = r
            cp4.close()
            V4.close()
        except Exception as e:
            sub["cfg4_1gpu"] = {"error": str(e)}
        sub["shard_probe"] = shard_probe_record(ppals, torch, local_rank)
        for key, single in (("cfg2_P8", head["ms_per_step"]),
                            ("cfg4_P8", sub.get("cfg4_1gpu", {}).get("ms_per_step"))):
            ent = sub["shard_probe"].get(key)
            if ent and single:
                ent["projected_efficiency_at_8"] = (single / 8.0) / ent["ms_per_sweep_per_rank"]

    if rank == 0:
        ms_per_step = head["ms_per_step"]
        sweeps_s = head["value"]
        flops = sweep_flops(lens, R, schedule)
        esz = 4 if args.dtype == "f32" else 8
        out = {
            "metric": "ALS sweeps/sec (exact dimension-tree sweep, CP order-4 "
                      f"s={lens[0]} R={R})",
            "value": sweeps_s,
            "unit": "sweeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic" if not hostsim else "synthetic — CPU REHEARSAL on the host stand-in, not a measurement",
            "config": {"workload": f"CP order-4 s={lens[0]} R={R} dense `-tensor r` (V=[[W_true]], "
                                   "U(0,1) factors), -pp 0 exact DT sweep incl. Normalize; tensor "
                                   f"stored {args.dtype} in HBM, factor/Gram/solve math fp64; "
                                   f"sweep schedule {schedule} (same ALS iterates either way)",
                       "lens": lens, "rank": R, "sharding": f"leading-mode block x{world}",
                       "comm": ("rccl: one all-reduce of the s x R partials per mode + redundant "
                                "fused update (default below 1 MiB); sub_records.reduce_scatter_plan "
                                "= the reduce-scatter / all-gather plan") if use_comm else "none"},
            "mttkrp_tflops": flops * sweeps_s / 1e12,
            "sweep_flops": flops,
            "rccl_ranks": nranks_seen,
            "final_gradnorm": gradnorm,
            "final_rel_residual": resid / vnorm,
        }
        # HBM traffic per launch of the scan kernels comes from separate rocprofv3 --pmc passes of
        # this same command (it cannot be read from inside the process): profiles/pmc_traffic.json
        traffic, traffic_src = None, None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            key = f"{args.workload}/{args.dtype}/{world}" + ("/dt" if schedule == "dt" else "")
            ent = pm.get(key)
            if ent:
                ks = {k: v for k, v in ent.items() if k.startswith("k_scan")}
                traffic = sum(v["fetch_bytes"] + v["write_bytes"] for v in ks.values()) / len(ks)
                traffic_src = (f"profiles/pmc_traffic.json[{key}]: rocprofv3 --pmc FETCH_SIZE / "
                               "WRITE_SIZE, separate passes, FETCH_SIZE x2 (gfx950), per launch of "
                               + " / ".join(sorted(ks)))
        except Exception:
            pass
        if "roofline" in head:
            rl = dict(head["roofline"])
            rl.update({
                # `traffic`: PMC counters of THIS run (child passes under rocprofv3 --pmc, see
                # live_pmc_traffic) or null; `traffic_replayed` is the committed figure of earlier
                # rocprofv3 --pmc passes of this same command
                "traffic": live_traffic, "traffic_source": live_traffic_src,
                "traffic_replayed": traffic, "traffic_replayed_source": traffic_src,
                "kernel": ("k_scan_suffix_buf" if (R <= 16 or esz == 8) else "k_scan_suffix_fast")
                          + " (tensor scan: one mode contracted per launch under msdt, a mode half "
                            "under dt; _buf = persistent buffer-load form, one n-tile or fp64 storage; "
                            "_fast = global-load form, fp32 with 16 < R)",
                "other_profiled_ms_per_step": other_ms_per_step,
                "note": f"algorithmic bytes = one read of the local tensor shard "
                        f"({esz} B/elem) + the result the launch must write, per scan launch; "
                        f"launches per sweep: 2 (dt) or N/(N-1) (msdt: one first-level "
                        f"contraction serves N-1 mode updates)"})
            out["roofline"] = rl
        out["placement"] = placement
        out["timed_region_ns"] = head.get("timed_region_ns")
        if sub:
            out["sub_records"] = sub
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(lens, R)
            except Exception as e:  # the baseline is reported, never required
                out["cpu_baseline"] = {"value": None, "unit": "sweeps/s", "cores": 0,
                                       "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
