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


@pytest.fixture(scope="module")
def BIN():
    """directory of the driver binaries under test (tests/test_drivers_hostsim.py re-runs these
    tests with the same drivers linked against the host stand-in)"""
    return os.path.join(ROOT, "pairwise-perturbation_amd", "bin")


@pytest.fixture(scope="module")
def full_size():
    return True


def run(cmd):
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return p.stdout


@pytest.mark.parametrize("pp,prec", [(0, 64), (1, 64), (0, 32)])
def test_test_ALS_matches_oracle(BIN, tmp_path, pp, prec):
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


def test_cli_defaults_and_silent_resets(BIN, tmp_path):
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


def test_pp_bench_lines(BIN, tmp_path):
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
def test_every_tensor_source_and_pp_mode_matches_oracle(BIN, tmp_path, args):
    """the rest of the -tensor / -pp flag space of test_ALS.cxx:222-326,352-396 at the CLI, against the
    oracle's CSV: -tensor p | p2 | c, CP -pp 2 (alsCP_PP_partupdate) and Tucker -pp 1 (alsTucker_PP).
    The driver dumps the tensor it built and the factors it started from (-dumpV / -dumpW0); the
    oracle is driven from those files with the CLI's options; compared row by row: iteration numbers,
    [pp_update] flags, [gradnorm] and [diffV] to 1e-4 (as test_test_ALS_matches_oracle does for r)."""
    csv, vdump, w0 = (str(tmp_path / n) for n in ("o.csv", "v.bin", "w0.bin"))
    maxiter = 25
    run([os.path.join(BIN, "test_ALS"), "-maxiter", str(maxiter), "-resprint", "1", "-filename", csv,
         "-prec", "64", "-dumpV", vdump, "-dumpW0", w0] + args)
    opt = dict(zip(args[0::2], args[1::2]))
    model, tensor = opt.get("-model", "CP"), opt["-tensor"]
    dim, s, R, ppmode = int(opt["-dim"]), int(opt["-size"]), int(opt["-rank"]), int(opt.get("-pp", "0"))
    lens = [s * s] * (dim // 2) if tensor == "p" else [s] * dim
    V = _read_doubles(vdump).reshape(lens, order="F")
    Vn = np.linalg.norm(V)
    ref = str(tmp_path / "ref.csv")
    kw = dict(tol=1e-10 * Vn, maxiter=maxiter, csv=ref, resprint=1)     # (-tol default 1e-10, x ||V||: test_ALS.cxx:354)
    if model == "CP":
        flat = _read_doubles(w0)
        W0, o = _split_flat(flat, lens, [R] * len(lens))
        G0, _ = _split_flat(flat[o:], lens, [R] * len(lens))
        if ppmode == 0:
            O.als_cp_dt(V, W0, G0, **kw)
        elif ppmode == 1:
            O.als_cp_pp(V, W0, G0, tol_init=float(opt["-pp_res_tol"]), **kw)
        else:
            O.als_cp_pp_partupdate(V, W0, G0, tol_init=float(opt["-pp_res_tol"]),
                                   update_percentage=float(opt["-update_percentage_pp"]), **kw)
    else:
        W0, _ = _split_flat(_read_doubles(w0), lens, [R] * len(lens))
        O.als_tucker_pp(V, W0, O.ttmc(V, W0, -1), tol_init=float(opt["-pp_res_tol"]), **kw)
    h1, r1 = O.read_csv(ref)
    h2, r2 = O.read_csv(csv)
    assert h1 == h2
    n = min(len(r1), len(r2))
    assert n >= 5
    compared = 0
    for a, b in zip(r1[:n], r2[:n]):
        if model == "CP" and a[5] < 1e-4 * Vn:
            break        # (converged: what is left of the rows is rounding)
        assert a[:2] == b[:2] and a[4] == b[4], (a, b)              # [dim], [iter], [pp_update]
        for col in (2, 5):                                          # [gradnorm] / metric, [diffV]
            assert abs(a[col] - b[col]) <= 1e-4 * abs(a[col]) + 1e-9 * Vn, (col, a, b)
        compared += 1
    assert compared >= 5
    if ppmode:
        assert any(r[4] == 1 for r in r2[:n]), "no PP sweep in the run: the case does not test the PP path"
    assert r2[-1][5] <= r2[0][5] * (1 + 1e-9)


@pytest.mark.parametrize("pp,kind", [(0, 1), (1, 2), (4, 0)])
def test_run_driver_class_api(BIN, tmp_path, pp, kind):
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


@pytest.mark.parametrize("pp,kind,ur", [(2, 3, 2), (3, 4, 1)])
def test_run_driver_low_rank_optimizers(BIN, tmp_path, pp, kind, ur):
    """bin/run -pp 2 / -pp 3 (run.cxx:401-411): CPD<double, CPDTLROptimizer / CPMSDTLROptimizer> with
    -updaterank (and once more with -randomsvd 1), CSV rows against the oracle's restatement; an
    update rank above the CP rank is refused with exit code 2"""
    s, R, N = 10, 3, 4
    csv = str(tmp_path / "out.csv")
    out = run([os.path.join(BIN, "run"), "-model", "CP", "-tensor", "r", "-dim", str(N), "-size",
               str(s), "-rank", str(R), "-pp", str(pp), "-updaterank", str(ur), "-maxiter", "8",
               "-resprint", "1", "-tol", "1e-9", "-filename", csv, "-prec", "64"])
    lines = out.splitlines()
    assert lines[1] == f"  dim=  {N}  size=  {s}  rank=  {R}  updaterank=  {ur}"
    lens = [s] * N
    V = O.build_V(O.init_factors(lens, R, 1000))
    W, G = O.init_factors(lens, R, 2000), O.init_factors(lens, R, 3000)
    Vn = np.linalg.norm(V)
    ref = str(tmp_path / "ref.csv")
    O.cpd_als_lr(V, W, G, kind, ur, tol=1e-9 * Vn, maxsweep=8, csv=ref, resprint=1)
    h1, r1 = O.read_csv(ref)
    h2, r2 = O.read_csv(csv)
    assert h1 == h2 and len(r1) == len(r2) >= 9
    for a, b in zip(r1, r2):
        assert a[:2] == b[:2] and a[4] == b[4] == 0
        assert abs(a[2] - b[2]) <= 1e-4 * abs(a[2]) + 1e-7
        assert abs(a[5] - b[5]) <= 1e-4 * abs(a[5]) + 1e-7 * Vn
    csv2, ref2 = str(tmp_path / "out_rnd.csv"), str(tmp_path / "ref_rnd.csv")
    out = run([os.path.join(BIN, "run"), "-model", "CP", "-tensor", "r", "-dim", str(N), "-size",
               str(s), "-rank", str(R), "-pp", str(pp), "-updaterank", str(ur), "-randomsvd", "1",
               "-maxiter", "8", "-resprint", "1", "-tol", "1e-9", "-filename", csv2, "-prec", "64"])
    assert "  randomsvd=  1" in out.splitlines()
    O.cpd_als_lr(V, W, G, kind, ur, tol=1e-9 * Vn, maxsweep=8, csv=ref2, resprint=1, randomsvd=1)
    h1, r1 = O.read_csv(ref2)
    h2, r2 = O.read_csv(csv2)
    assert h1 == h2 and len(r1) == len(r2) >= 9
    for a, b in zip(r1, r2):
        assert a[:2] == b[:2] and a[4] == b[4] == 0
        assert abs(a[2] - b[2]) <= 1e-4 * abs(a[2]) + 1e-7
        assert abs(a[5] - b[5]) <= 1e-4 * abs(a[5]) + 1e-7 * Vn
    for extra, msg in ((["-updaterank", "9"], "updaterank"),):
        p = subprocess.run([os.path.join(BIN, "run"), "-tensor", "r", "-dim", "4", "-size", "8",
                            "-rank", "3", "-pp", str(pp), "-filename", str(tmp_path / "o.csv")] + extra,
                           capture_output=True, text=True, timeout=120)
        assert p.returncode == 2 and msg in p.stderr, (p.returncode, p.stderr)


def _read_doubles(path):
    return np.fromfile(path, dtype="<f8")


def _split_flat(flat, lens, ranks):
    out, o = [], 0
    for s, r in zip(lens, ranks):
        out.append(flat[o:o + s * r].reshape((s, r), order="F"))
        o += s * r
    return out, o


def relerr(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("prec", [64, 32])
def test_cfg1_cli_matches_oracle(BIN, full_size, tmp_path, prec):
    """BASELINE.json configs[0], verbatim: `test_ALS -model CP -tensor r -dim 3 -size 64 -rank 5
    -pp 0` (order 3 = the generalised tree, SURVEY §8a note). Default flags otherwise: -tol 1e-10,
    -maxiter 5000, -resprint 10; the run ends by the reference's stop criterion in fp64 storage
    (the default). CSV rows and the FINAL FACTORS (-dumpW) against the oracle driven from the
    same counter-based initialisation; fp32 storage (-prec 32, 200 sweeps): factors within 1e-5."""
    s, R, N = (64 if full_size else 20), 5, 3
    csv, dump = str(tmp_path / "out.csv"), str(tmp_path / "w.bin")
    cmd = [os.path.join(BIN, "test_ALS"), "-model", "CP", "-tensor", "r", "-dim", str(N), "-size",
           str(s), "-rank", str(R), "-pp", "0", "-filename", csv, "-dumpW", dump]
    maxiter = 5000
    if prec == 32:
        maxiter = 200
        cmd += ["-prec", "32", "-maxiter", str(maxiter)]
    out = run(cmd)
    assert out.splitlines()[0] == "  model=  CP  tensor=  r  pp=  0"
    assert out.splitlines()[1] == f"  dim=  {N}  size=  {s}  rank=  {R}"
    lens = [s] * N
    V = O.build_V(O.init_factors(lens, R, 1000))
    W, G = O.init_factors(lens, R, 2000), O.init_factors(lens, R, 3000)
    Vn = np.linalg.norm(V)
    ref = str(tmp_path / "ref.csv")
    rc_ref, it_ref, W_ref, G_ref = O.als_cp_dt(V, W, G, tol=1e-10 * Vn, maxiter=maxiter, csv=ref,
                                               resprint=10)
    h1, r1 = O.read_csv(ref)
    h2, r2 = O.read_csv(csv)
    assert h1 == h2
    floor = (1e-7 if prec == 64 else 1e-4) * Vn     # compare rows above the rounding floor
    n = 0
    for a, b in zip(r1, r2):
        if a[2] < floor or a[5] < floor:
            break
        assert a[:2] == b[:2] and a[4] == b[4] == 0
        assert abs(a[2] - b[2]) <= (1e-4 if prec == 64 else 1e-2) * a[2], (a, b)
        assert abs(a[5] - b[5]) <= (1e-4 if prec == 64 else 1e-2) * a[5], (a, b)
        n += 1
    assert n >= 5
    if prec == 64:
        # both runs end by `gradnorm < tol` (an exact-rank problem), within a print period
        assert rc_ref == 1 and r2[-1][2] < 1e-10 * Vn * 1.0001
        assert abs(r2[-1][1] - r1[-1][1]) <= 10
    flat = _read_doubles(dump)
    Wd, o = _split_flat(flat, lens, [R] * N)
    assert flat.size == 2 * o
    for a, b in zip(Wd, W_ref):
        assert relerr(a, b) < (1e-7 if prec == 64 else 1e-5), (prec, relerr(a, b))


@pytest.mark.parametrize("model", ["CP", "Tucker"])
def test_file_exchange_and_o_path(BIN, tmp_path, model):
    """-dumpV / -dumpW0 / -loadW0 / -dumpW (raw fp64, first index fastest: the layout
    V.read_dense_from_file reads, test_ALS.cxx:289-325) and the `-tensor o1` file path
    (test_ALS.cxx:287-305) on a small file through the -lens hook: the dumps hold what the oracle
    builds, a run fed back from the files reproduces the first run bit for bit, and the oracle
    driven from the SAME FILES agrees — the route by which a CTF run made elsewhere can be compared."""
    lens, R = [6, 8, 7, 9], 3
    ranks = [2, 3, 3, 2]
    vbin, w0, w1, w2 = (str(tmp_path / n) for n in ("v.bin", "w0.bin", "w1.bin", "w2.bin"))
    c1, c2 = str(tmp_path / "a.csv"), str(tmp_path / "b.csv")
    # the source tensor: an `r`-type tensor of non-cubic shape, written the way imageloader.py does
    V = O.build_V(O.init_factors(lens, R, 1000))
    np.asfortranarray(V).ravel(order="F").astype("<f8").tofile(vbin)
    common = ["-model", model, "-tensor", "o1", "-tensorfile", vbin, "-lens",
              ",".join(map(str, lens)), "-rank", str(R), "-maxiter", "12", "-resprint", "1", "-tol",
              "1e-12", "-dim", "4"]
    if model == "Tucker":
        common += ["-ranks", ",".join(map(str, ranks))]
    vdump = str(tmp_path / "vdump.bin")
    out = run([os.path.join(BIN, "test_ALS")] + common +
              ["-filename", c1, "-dumpV", vdump, "-dumpW0", w0, "-dumpW", w1])
    assert "Read the tensor from file" in out and "Read dataset finished" in out
    assert np.array_equal(_read_doubles(vdump), _read_doubles(vbin))   # fp64 storage: exact
    run([os.path.join(BIN, "test_ALS")] + common + ["-filename", c2, "-loadW0", w0, "-dumpW", w2])
    assert np.array_equal(_read_doubles(w1), _read_doubles(w2))
    rows1, rows2 = O.read_csv(c1)[1], O.read_csv(c2)[1]
    assert [r[:6] for r in rows1] == [r[:6] for r in rows2]
    Vf = _read_doubles(vbin).reshape(lens, order="F")
    Vn = np.linalg.norm(Vf)
    if model == "CP":
        W0, o = _split_flat(_read_doubles(w0), lens, [R] * 4)
        G0, _ = _split_flat(_read_doubles(w0)[o:], lens, [R] * 4)
        for a, b in zip(W0, O.init_factors(lens, R, 2000)):
            assert np.array_equal(a, b)
        _, _, W_ref, G_ref = O.als_cp_dt(Vf, W0, G0, tol=1e-12 * Vn, maxiter=12, resprint=1)
        W1, o = _split_flat(_read_doubles(w1), lens, [R] * 4)
        G1, _ = _split_flat(_read_doubles(w1)[o:], lens, [R] * 4)
        for a, b in zip(W1, W_ref):
            assert relerr(a, b) < 1e-8
        for a, b in zip(G1, G_ref):
            assert np.linalg.norm(a - b) < 1e-7 * (1 + np.linalg.norm(b))
    else:
        W0, o = _split_flat(_read_doubles(w0), lens, ranks)
        Wh, core_h = O.hosvd(Vf, ranks)
        for a, b in zip(W0, Wh):
            assert relerr(a @ a.T, b @ b.T) < 1e-8
        _, _, W_ref, core_ref = O.als_tucker_dt(Vf, W0, O.ttmc(Vf, W0, -1), tol=1e-12 * Vn,
                                                maxiter=12, resprint=1)
        flat = _read_doubles(w1)
        W1, o = _split_flat(flat, lens, ranks)
        assert flat.size == o + int(np.prod(ranks))
        for a, b in zip(W1, W_ref):
            assert relerr(a @ a.T, b @ b.T) < 1e-7
        assert abs(np.linalg.norm(flat[o:]) - np.linalg.norm(core_ref)) < 1e-8 * np.linalg.norm(core_ref)


def test_o_path_rejects_short_file(BIN, tmp_path):
    vbin = str(tmp_path / "v.bin")
    np.zeros(100).tofile(vbin)
    p = subprocess.run([os.path.join(BIN, "test_ALS"), "-tensor", "o1", "-tensorfile", vbin, "-lens",
                        "4,5,6,7", "-dim", "4", "-rank", "2", "-filename", str(tmp_path / "o.csv")],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "does not hold exactly 840 doubles" in p.stderr


def test_pp_bench_tucker_lines(BIN, tmp_path):
    """pp_bench -model Tucker (pp_bench.cxx:321-345; script_weakscaling.py:41-46 runs it)"""
    csv = str(tmp_path / "b.csv")
    out = run([os.path.join(BIN, "pp_bench"), "-model", "Tucker", "-tensor", "r2", "-dim", "3",
               "-size", "14", "-rank", "3", "-maxiter", "3", "-filename", csv])
    text = open(csv).read().splitlines()
    assert text[0] == "[timetype],[dtime]"
    assert sum(ln.startswith("[DTtime],") for ln in text) == 3
    assert sum(ln.startswith("  [PPfirst]  ,") for ln in text) == 3
    assert sum(ln.startswith("  [PPsecond]  ,") for ln in text) == 3
    assert out.count("pairwise perturbation starts from 0") == 3
    assert out.count("Iter = 2 Final Diff norm") == 6   # DT: loop end at maxiter+1; PP: iter++ on exit
