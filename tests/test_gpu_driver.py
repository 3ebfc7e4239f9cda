"""GPU tests of the C++ drivers (bin/test_ALS, bin/pp_bench): CLI compatibility with the reference
(test_ALS.cxx:64-217), CSV format, and numeric agreement of the whole run with the oracle driven
from the same counter-based initialisation."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "pairwise-perturbation_amd", "bin")


def run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


@pytest.mark.parametrize("pp,prec", [(0, 64), (1, 64), (0, 32)])
def test_test_ALS_matches_oracle(tmp_path, pp, prec):
    s, R, N = 12, 3, 4
    csv = str(tmp_path / "out.csv")
    out = run([os.path.join(BIN, "test_ALS"), "-model", "CP", "-tensor", "r", "-dim", str(N),
               "-size", str(s), "-rank", str(R), "-pp", str(pp), "-maxiter", "30", "-resprint", "1",
               "-pp_res_tol", "0.1", "-tol", "1e-6", "-filename", csv, "-prec", str(prec)])
    lines = out.splitlines()
    # echo block, test_ALS.cxx:203-217
    assert lines[0] == f"  model=  CP  tensor=  r  pp=  {pp}"
    assert lines[1] == f"  dim=  {N}  size=  {s}  rank=  {R}"
    assert lines[2] == "  issparse=  0  tolerance=  1e-06  restarttol=  0.1"
    assert lines[5] == "  timelimit=  5000  maxiter=  30  resprint=  1"
    assert lines[6] == "  tensorfile=  test  update_percentage_pp=  1"
    assert lines[7].startswith("Vnorm= ")
    assert any(ln.startswith("Iter = ") for ln in lines)
    assert any(ln.startswith("tf took ") for ln in lines)
    assert lines[-1].startswith("experiment took ")
    lens = [s] * N
    V = O.build_V(O.init_factors(lens, R, 1000))
    W, G = O.init_factors(lens, R, 2000), O.init_factors(lens, R, 3000)
    Vn = np.linalg.norm(V)
    assert abs(float(lines[7].split()[1]) - Vn) < 1e-5 * Vn
    ref = str(tmp_path / "ref.csv")
    if pp == 0:
        O.als_cp_dt(V, W, G, tol=1e-6 * Vn, maxiter=30, csv=ref, resprint=1)
    else:
        O.als_cp_pp(V, W, G, tol=1e-6 * Vn, tol_init=0.1, maxiter=30, csv=ref, resprint=1)
    h1, r1 = O.read_csv(ref)
    h2, r2 = O.read_csv(csv)
    assert h1 == h2 == ["[dim]", "[iter]", "[gradnorm]", "[tol]", "[pp_update]", "[diffV]", "[dtime]"]
    n = min(len(r1), len(r2))
    assert n >= 5
    tol = 1e-4 if prec == 64 else 5e-3
    for a, b in zip(r1[:n], r2[:n]):
        if a[5] < 1e-4 * Vn:
            break
        assert a[:2] == b[:2] and a[4] == b[4]
        assert abs(a[2] - b[2]) <= tol * abs(a[2]) and abs(a[5] - b[5]) <= tol * abs(a[5])


def test_cli_defaults_and_silent_resets(tmp_path):
    """out-of-range values are silently reset to the defaults (test_ALS.cxx:76-146)"""
    csv = str(tmp_path / "o.csv")
    out = run([os.path.join(BIN, "test_ALS"), "-model", "X", "-tensor", "r", "-dim", "3", "-size",
               "8", "-rank", "99", "-pp", "7", "-tol", "5", "-maxiter", "2", "-bogus", "1",
               "-filename", csv])
    lines = out.splitlines()
    assert lines[0] == "  model=  CP  tensor=  r  pp=  0"
    assert lines[1] == "  dim=  3  size=  8  rank=  4"      # rank reset to s/2
    assert "tolerance=  1e-10" in lines[2]                   # tol > 1 reset
    assert "resprint=  10" in lines[5]


def test_pp_bench_lines(tmp_path):
    csv = str(tmp_path / "b.csv")
    run([os.path.join(BIN, "pp_bench"), "-model", "CP", "-tensor", "r", "-dim", "4", "-size", "16",
         "-rank", "3", "-maxiter", "3", "-filename", csv])
    text = open(csv).read().splitlines()
    assert text[0] == "[timetype],[dtime]"
    assert sum(ln.startswith("[DTtime],") for ln in text) == 3
    assert sum(ln.startswith("  [PPfirst]  ,") for ln in text) == 3
    assert sum(ln.startswith("  [PPsecond]  ,") for ln in text) == 3
    for ln in text:
        if "," in ln and not ln.startswith("[timetype]"):
            assert float(ln.split(",")[1]) > 0


@pytest.mark.parametrize("args", [
    ["-tensor", "p", "-dim", "6", "-size", "4", "-rank", "3"],        # folded Poisson, order 3
    ["-tensor", "p2", "-dim", "4", "-size", "6", "-rank", "3"],       # Poisson operator, order 4
    ["-tensor", "c", "-dim", "4", "-size", "10", "-rank", "3", "-pp", "1", "-pp_res_tol", "0.1"],
    ["-tensor", "r", "-dim", "4", "-size", "10", "-rank", "2", "-pp", "2",
     "-update_percentage_pp", "0.5", "-pp_res_tol", "0.1"],
    ["-model", "Tucker", "-tensor", "r2", "-dim", "3", "-size", "12", "-rank", "3", "-pp", "1",
     "-pp_res_tol", "0.1"],
])
def test_every_tensor_source_and_pp_mode_runs(tmp_path, args):
    """the whole -tensor / -pp flag space of test_ALS.cxx:222-326,352-396 is accepted and decreases
    the residual"""
    csv = str(tmp_path / "o.csv")
    run([os.path.join(BIN, "test_ALS"), "-maxiter", "25", "-resprint", "1", "-filename", csv,
         "-prec", "64"] + args)
    _, rows = O.read_csv(csv)
    assert len(rows) >= 5
    assert rows[-1][5] <= rows[0][5] * (1 + 1e-9)


@pytest.mark.parametrize("pp,kind", [(0, 1), (1, 2), (4, 0)])
def test_run_driver_class_api(tmp_path, pp, kind):
    """bin/run: the class-API front end (run.cxx:47-472). Echo block of run.cxx:222-240 (with the
    updaterank / randomsvd fields), dispatch of run.cxx:387-414, console rows with [sweeps] and
    [residual], CSV rows against the oracle's CPD::als restatement."""
    s, R, N = 10, 3, 4
    csv = str(tmp_path / "out.csv")
    out = run([os.path.join(BIN, "run"), "-model", "CP", "-tensor", "r", "-dim", str(N), "-size",
               str(s), "-rank", str(R), "-pp", str(pp), "-maxiter", "6", "-resprint", "1", "-tol",
               "1e-9", "-filename", csv, "-prec", "64"])
    lines = out.splitlines()
    assert lines[0] == f"  model=  CP  tensor=  r  pp=  {pp}"
    assert lines[1] == f"  dim=  {N}  size=  {s}  rank=  {R}  updaterank=  {s // 2}"
    assert lines[6] == "  tensorfile=  test  update_percentage_pp=  1"
    assert lines[7] == "  randomsvd=  0"
    assert lines[8].startswith("Vnorm= ")
    assert any("  [sweeps]=  " in ln and "  [residual]  " in ln for ln in lines)
    assert any(ln.startswith("Iters = ") for ln in lines)
    lens = [s] * N
    V = O.build_V(O.init_factors(lens, R, 1000))
    W, G = O.init_factors(lens, R, 2000), O.init_factors(lens, R, 3000)
    Vn = np.linalg.norm(V)
    ref = str(tmp_path / "ref.csv")
    O.cpd_als(V, W, G, kind, tol=1e-9 * Vn, maxsweep=6, csv=ref, resprint=1)
    h1, r1 = O.read_csv(ref)
    h2, r2 = O.read_csv(csv)
    assert h1 == h2 and len(r1) == len(r2) >= 7
    for a, b in zip(r1, r2):
        assert a[:2] == b[:2] and a[4] == b[4] == 0
        assert abs(a[2] - b[2]) <= 1e-4 * abs(a[2]) + 1e-7
        assert abs(a[5] - b[5]) <= 1e-4 * abs(a[5]) + 1e-7 * Vn


def test_run_driver_rejects_low_rank_optimizers(tmp_path):
    p = subprocess.run([os.path.join(BIN, "run"), "-tensor", "r", "-dim", "4", "-size", "8", "-pp",
                        "2", "-filename", str(tmp_path / "o.csv")], capture_output=True, text=True,
                       timeout=120)
    assert p.returncode == 2 and "low-rank" in p.stderr
