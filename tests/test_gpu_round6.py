"""GPU, round 6: the on-disk contract carried by a GPU run, the division-form re-close of the raw
kernels, grids beyond 2050 points, the nearest-sigma flag of the drop-in."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def ctx():
    import ibs_amd
    c = ibs_amd.Context(0)
    yield c
    c.close()


def test_F4_history_files_written_from_gpu_adjoint_steps_and_read_like_the_optimizer_driver(ctx, tmp_path):
    """Rows A8 + F4 with the GPU in the loop: three optimizer iterations, each ONE AdjointStep.run() on the device (3 equilibria
    x 5 surfaces: base + 2 DOF-perturbed, N = 969), every equilibrium's rows appended to save_n_load-style files exactly as
    ball_scan.py:359-384 does (0-d placeholders of arr_create2.py:87-97 first), then read back exactly as
    sims_runner_NCSX.py reads them: fobj takes `ball_gam0.npy` whole at iteration 0 and its last row afterwards (:300-306),
    dfobj takes `np.load(ball_gam{i}.npy)[-1]` of every dof (:198-199) into f0_arr / df0_arr (:249-261).  The files must hold
    the oracle pipeline's numbers (1e-8) and the objective / gradient rebuilt from the FILES must equal what the step returned
    in memory."""
    import torch
    import ibs_amd
    import bench
    from tests.helpers import oracle_surface_pipeline
    dev = torch.device("cuda:0")
    wout0 = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    wouts, steps, _ = bench.emulated_equilibria(wout0)
    ns, na, nt0, ndof = 5, 24, 15, 2
    svals = np.linspace(0.5, 0.95, ns)                                   # ball_scan.py:197
    th = ibs_amd.theta_grid_for(11, 11)                                  # ball_scan.py:201-208: 969 points
    thresh, prefac = -2.0e-4, 50.0                                       # sims_runner_NCSX.py:56-57
    path0 = str(tmp_path)
    ibs_amd.create_history_placeholders(path0, ndof)                     # arr_create2.py:87-97
    assert np.load(os.path.join(path0, "ball_gam0.npy")).shape == ()
    step = ibs_amd.AdjointStep(ctx, th, svals, dev, nalpha=na, ntheta0=nt0, gamma_thresh=thresh, prefac=prefac)
    picks = [(0, 0, 1), (1, 2, 4), (2, 1, 0)]                            # (iteration, dof, surface) re-done by the oracle pipeline
    for iter0 in range(3):
        # each iteration has its own equilibria (the optimizer moved): three of the emulated NCSX_op set, the first acting as base
        eq = [3 * iter0, 3 * iter0 + 1, 3 * iter0 + 2]
        w = [wouts[q] for q in eq]
        st = np.array([1.0, steps[eq[1]], steps[eq[2]]])              # (entry 0 unused: sims_runner_NCSX.py:258)
        f_other = np.array([0.8, 0.81, 0.82]) + 0.1 * iter0              # (f{i}.npy: the non-ballooning part, out of scope)
        out = step.run(w, f_other, st)
        for dof in range(ndof + 1):                                      # what every `srun ball_scan.py iter dof ngroups` does at its end
            ibs_amd.append_history(path0, dof, iter0, out["gam"][dof], out["theta0"][dof], out["alpha"][dof])
        # ---- consumer side, sims_runner_NCSX.py written out
        for name in ("ball_gam", "ball_theta0", "ball_alpha"):
            arr = np.load(os.path.join(path0, "%s1.npy" % name), allow_pickle=True)
            assert arr.shape == ((ns,) if iter0 == 0 else (iter0 + 1, ns)), (name, arr.shape)    # ball_scan.py:369-379
        if iter0 == 0:
            gamma_ball = np.load(path0 + "/ball_gam{0}.npy".format(0), allow_pickle=True)        # :300-303
        else:
            gamma_ball = np.load(path0 + "/ball_gam{0}.npy".format(0), allow_pickle=True)[-1]    # :304-306
        f0 = f_other[0] + prefac * np.sum(np.maximum(gamma_ball - thresh, 0.0))                   # :311-313
        assert abs(np.sqrt(f0) - out["fobj"]) < 1e-15                                             # :318
        if iter0 > 0:                                                    # dfobj is "never called at the first itern" (:192)
            gamma_ball2 = np.zeros((ndof + 1, ns)); f0_arr = np.zeros(ndof + 1); df0 = np.zeros(ndof)
            for i in range(ndof + 1):
                gamma_ball2[i] = np.load(path0 + "/ball_gam{0}.npy".format(i))[-1]               # :198-199
                f0_arr[i] = f_other[i] + prefac * np.sum(np.maximum(gamma_ball2[i] - thresh, 0.0))   # :252-257
                if i > 0:
                    df0[i - 1] = (f0_arr[i] - f0_arr[0]) / st[i] * 0.5 * 1 / np.sqrt(f0_arr[0])  # :258-261
            assert np.array_equal(gamma_ball2, out["gam"]) and np.array_equal(f0_arr, out["f0"])
            assert np.abs(df0 - out["dfobj"]).max() <= 1e-15 * max(1.0, np.abs(df0).max())
        # ---- the numbers in the files against the oracle pipeline (numpy geometry, C-oracle scan, scipy L-BFGS-B, oracle final solve)
        for (it, dof, js) in picks:
            if it != iter0:
                continue
            ref = oracle_surface_pipeline(w[dof], float(svals[js]), th, na, nt0, step.del_alpha)
            row = lambda name: np.atleast_2d(np.load(os.path.join(path0, "%s%d.npy" % (name, dof)), allow_pickle=True))[-1]
            assert abs(row("ball_gam")[js] - ref["gam"]) < 1e-8, (it, dof, js, row("ball_gam")[js], ref["gam"])
            assert abs(row("ball_alpha")[js] - ref["x_opt"][0]) < 1e-5 and abs(row("ball_theta0")[js] - ref["x_opt"][1]) < 1e-5
    hist = np.load(os.path.join(path0, "ball_gam2.npy"))
    assert hist.shape == (3, ns) and np.all(np.isfinite(hist)) and len(np.unique(hist)) == 3 * ns    # three different iterations


# ---------------------------------------------------------------------------------------------- world > 1 on one GPU
def _build_fake_rccl(tmp_path):
    """tests/cabi/fake_rccl.c: RCCL's five entry points over shared memory + stream-ordered copies (test infrastructure)"""
    import subprocess
    so = str(tmp_path / "libfake_rccl.so")
    inc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")
    r = subprocess.run(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-I", inc, os.path.join(ROOT, "tests", "cabi", "fake_rccl.c"),
                        "-o", so, "-lrt", "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return so


@pytest.mark.parametrize("world", [2, 4])
def test_library_collectives_with_more_than_one_rank(tmp_path, world):
    """ibs_comm_init -> ibs_comm_allgather_f64 -> ibs_comm_allgather_start_f64 over all 16 slots (then_wait_slot and host waits)
    -> ibs_comm_wait -> ibs_comm_destroy with 2 and 4 ranks: fresh processes that share the box's one GPU and meet through the
    shared-memory stand-in for librccl (RCCL refuses two ranks on one device).  Every rank checks every rank's rows of every
    gather; the stand-in delays each gather by 3 ms, and a control inside the worker shows that a consumer which does NOT wait
    sees the old buffer contents -- so a missing ordering in the library's slot logic fails these checks."""
    import json
    import subprocess
    fake = _build_fake_rccl(tmp_path)
    env = dict(os.environ, FAKE_RCCL_DELAY_US="3000", FAKE_RCCL_TIMEOUT_S="60", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "cabi", "comm_worker.py"), str(r), str(world), fake,
                               str(tmp_path / "id.bin"), str(tmp_path / ("rank%d.json" % r))], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=420))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        fn = tmp_path / ("rank%d.json" % r)
        res = json.load(open(fn)) if fn.exists() else {"error": "no report", "stderr": outs[r][1][-2000:] if r < len(outs) else ""}
        assert res.get("ok") is True and p.returncode == 0, (r, res, outs[r][1][-1500:] if r < len(outs) else "")
        assert res["checks"]["slots_gathers_checked"] >= 3 * 16 and res["checks"]["control_without_wait_sees_old_contents"] is True


def test_bench_two_ranks_with_the_librarys_own_gathers(tmp_path):
    """`python bench.py --gpus 2` on the one-GPU box with the stand-in bound as the library's RCCL (IBS_RCCL_LIB): beside the
    torch.distributed legs of tests/test_gpu_round5.py the per-step gather now also runs through ibs_comm_allgather_f64 (in the
    step's stream) and through the overlapped 3-slot rotation of ibs_comm_allgather_start_f64 with host waits -- the product's
    default form -- each with its round trip checked on both ranks, and configs[2] sharded is gathered by the library."""
    import json
    import subprocess
    import time
    fake = _build_fake_rccl(tmp_path)
    detail = tmp_path / "detail.json"
    env = dict(os.environ, IBS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", IBS_RCCL_LIB=fake, IBS_BENCH_DETAIL=str(detail),
               FAKE_RCCL_TIMEOUT_S="60")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--no-stress"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 6000
    full = json.load(open(detail))
    modes = full["gather_modes"]
    assert "native_error" not in modes, modes
    for m in ("torch_in_stream", "native_in_stream", "native_overlapped"):
        assert modes[m]["allgather_roundtrip_ok"] is True and modes[m]["ms_per_step"] > 0, (m, modes[m])
    assert modes["native_in_stream"]["ranks_in_collective"] == 2
    assert full["config"]["headline_gather"] in modes and full["value"] == modes[full["config"]["headline_gather"]]["solves_per_s"]
    nat = full["ncsx_c2_sharded_native"]
    assert nat["checks_passed"] is True and nat["gathered_equals_one_gpu_bitwise"] is True
    print("2 ranks, library's own gathers through the stand-in: %.0f s; ms per step %s; headline = %s" % (
        time.time() - t0, {k: round(v["ms_per_step"], 4) for k, v in modes.items() if isinstance(v, dict)}, full["config"]["headline_gather"]))


# ---------------------------------------------------------------------------------------------- grids beyond 2050 points
@pytest.mark.parametrize("N", [2561, 4097, 8193, 65537])
def test_large_grid_salpha_against_the_oracle(ctx, N):
    """utils.py:1556-1624 accepts any grid length and the reference's own rule N = 2 mpol ntor 4 + 1 (ball_scan.py:201-208) exceeds
    2050 points from mpol ntor > 256 on (N = 2561 is mpol = 20, ntor = 16).  Beyond the register-resident kernels the library runs the
    generic division-form path (csrc/ibs_long.hip): s-alpha systems (the reference's own test family,
    bishop_ball_s-alpha.py:30-45) through the raw entry point and through the drop-in gamma_ball_full against the oracle: gam
    1e-8 (the stated FP64 tolerance), lam to 4 N eps ||A||, X / dX, the Sturm count at 0 and around lam exact, invalid data
    flagged, host and device pointers."""
    import torch
    import ibs_amd
    from oracle import ballooning_oracle as bo
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    th = np.linspace(-4 * np.pi, 4 * np.pi, N)
    h = float(th[1] - th[0])
    cases = [(1.0, 0.8, 0.0), (0.5, 0.6, 0.1), (1.5, 1.1, 0.3), (0.3, 0.3, 0.0), (1.0, 0.3, 0.2)]     # (shat, alpha, theta0) of G1
    g = np.empty((len(cases), N)); c = np.empty_like(g)
    for k, (sh, al, t0) in enumerate(cases):
        g[k], c[k] = bo.salpha_gc(th, sh, al, t0)
    f = g.copy()
    gam_c, lam_c, _ = co.solve_gcf_batch(h, g, c, f)
    ee = (g[:, :-2] + 2 * g[:, 1:-1] + g[:, 2:]) / (2 * h * h)                     # e_lo + e_hi of every row (utils.py:1574-1592)
    nA = ((np.abs(c[:, 1:-1] - ee) + ee) / f[:, 1:-1]).max(axis=1)                 # the solver's ||A|| bound
    r = ctx.solve_gcf(h, torch.from_numpy(g).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev), want_X=True, want_info=True)
    assert "k_solve_gcf_long<double>" in ctx.last_launch()[0], ctx.last_launch()
    assert int(((r["info"] >> 16) != 0).sum()) == 0
    assert np.abs(r["gam"].cpu().numpy() - gam_c).max() < 1e-8, np.abs(r["gam"].cpu().numpy() - gam_c).max()
    assert (np.abs(r["lam"].cpu().numpy() - lam_c) / nA).max() < 4 * N * 2.220446049250313e-16
    for k in range(len(cases)):
        ref = co.solve_gcf(h, g[k], c[k], f[k])
        X = r["X"][k].cpu().numpy(); dX = r["dX"][k].cpu().numpy()
        sgn = np.sign(X[np.argmax(np.abs(X))]) * np.sign(ref[2][np.argmax(np.abs(ref[2]))])
        assert np.abs(sgn * X - ref[2]).max() < 1e-6 and np.abs(sgn * dX - ref[3]).max() < 1e-5 * max(1.0, np.abs(ref[3]).max())
        assert abs(np.abs(X).max() - 1.0) < 1e-15 and X[0] == 0.0 and X[-1] == 0.0           # utils.py:1605-1608
    # host pointers, eigenvalues only, FP32 arrays
    rh = ctx.solve_gcf(h, g, c, f, want_info=True)
    assert np.array_equal(rh["lam"], r["lam"].cpu().numpy()) and np.array_equal(rh["gam"], r["gam"].cpu().numpy()) and rh["nbad"] == 0
    rl = ctx.solve_gcf(h, torch.from_numpy(g).to(dev), torch.from_numpy(c).to(dev), torch.from_numpy(f).to(dev), want_gam=False)
    assert torch.equal(rl["lam"], r["lam"]) and rl["gam"] is None
    r32 = ctx.solve_gcf(h, torch.from_numpy(g).to(dev).float(), torch.from_numpy(c).to(dev).float(), torch.from_numpy(f).to(dev).float(),
                        dtype=np.float32)
    assert "k_solve_gcf_long<float>" in ctx.last_launch()[0] and r32["gam"].dtype == torch.float32
    assert np.abs(r32["gam"].double().cpu().numpy() - gam_c).max() < 1e-4      # (the systems the FP32 arrays define, solved in FP64)
    # Sturm counts (division form): the stability verdict of bishop_ball_s-alpha.py:110-115 and the eigenvalue's own neighbourhood
    z = np.zeros(len(cases))
    assert np.array_equal(ctx.sturm_count(h, g, c, f, z), co.count_above_batch(h, g, c, f, z))
    assert "k_sturm_count_long" in ctx.last_launch()[0]
    lam = r["lam"].cpu().numpy()
    # (the multisection closes to a bracket of 2 eps ||A|| and returns its midpoint: ||A|| ~ 2 g / h^2 is 3e6 at N = 2,561, 2e9 at 65,537)
    mg = np.maximum(1e-9, 3 * 2.220446049250313e-16 * nA)
    assert (ctx.sturm_count(h, g, c, f, lam + mg) == 0).all() and (ctx.sturm_count(h, g, c, f, lam - mg) == 1).all()
    # the drop-in on the long grid (geometry-fed path: B = gradpar = 1, dPdrho = -1: bishop_ball_s-alpha.py fed to gamma_ball_full)
    sh, al, t0 = cases[0]
    out = ibs_amd.gamma_ball_full(-1.0, th, np.ones(N), np.ones(N), c[0], g[0], ctx=ctx)
    assert abs(out[0] - gam_c[0]) < 1e-8 and out[1].shape == (N,) and np.array_equal(out[3], g[0])
    # ... and on a NON-uniform long grid (regridded like utils.py:1567-1576: caller-supplied half-grid g)
    th_nu = th + 0.3 * h * np.sin(3 * th)
    gn, cn = bo.salpha_gc(th_nu, sh, al, t0)
    ref = bo.gamma_ball_full(-1.0, th_nu, np.ones(N), np.ones(N), cn, gn)
    out = ibs_amd.gamma_ball_full(-1.0, th_nu, np.ones(N), np.ones(N), cn, gn, ctx=ctx)
    assert abs(out[0] - ref[0]) < 1e-8, (out[0], ref[0])
    # invalid data is flagged (status 2), its neighbours untouched
    g2 = g.copy(); g2[1, N // 3] = -1.0
    rb = ctx.solve_gcf(h, g2, c, f, want_info=True)
    assert rb["nbad"] == 1 and (rb["info"][1] >> 16) == 2 and rb["lam"][0] == rh["lam"][0] and rb["gam"][4] == rh["gam"][4]
    # even N is still refused by the growth-rate entry points (Simpson rule restated for odd N), accepted by the count
    with pytest.raises(ibs_amd.IbsError):
        ctx.solve_gcf(h, g[:, :-1], c[:, :-1], f[:, :-1])
    assert ctx.sturm_count(h, g[:, :-1], c[:, :-1], f[:, :-1], z).shape == (len(cases),)


def test_large_grid_rough_coefficients_against_the_oracle(ctx):
    """The long-grid path on the rough family of configs[4] (iid coefficients inside the NCSX envelopes: the systems on which the
    prefix-product counts of the short-grid kernels needed the closing checks of round 6): division form throughout, so every
    eigenvalue within 4 N eps ||A|| of the oracle's division-form bisection without any re-close, the Sturm count around it exact,
    the growth rate of utils.py:1601-1621 on the twisted-factorisation eigenvector next to the oracle's."""
    import torch
    import bench
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    EPS = 2.220446049250313e-16
    for N, n in ((2561, 96), (4097, 64), (16385, 8)):
        h, g, c, f = bench.c5_family(dev, "rough", n, N, seed=31 + N)
        nA = bench.norm_a(h, g, c, f).cpu().numpy()
        r = ctx.solve_gcf(h, g, c, f, want_info=True)
        assert "k_solve_gcf_long<double>" in ctx.last_launch()[0] and int(((r["info"] >> 16) != 0).sum()) == 0
        gn, cn, fn = g.cpu().numpy(), c.cpu().numpy(), f.cpu().numpy()
        lam = r["lam"].cpu().numpy()
        lam_c = co.lam_batch(h, gn, cn, fn)
        assert (np.abs(lam - lam_c) / nA).max() <= 4 * N * EPS, (N, (np.abs(lam - lam_c) / nA).max() / (N * EPS))
        tol = 4 * N * EPS * nA
        assert np.array_equal(ctx.sturm_count(h, gn, cn, fn, lam + tol), np.zeros(n, dtype=np.int32))
        assert (ctx.sturm_count(h, gn, cn, fn, lam - tol) >= 1).all()
        gam_c, lam_s, _ = co.solve_gcf_batch(h, gn, cn, fn)
        # ||A|| is 5e6 ... 3e8 here against lam ~ 1e-3: the growth rate of the two division-form pipelines agrees to 1e-9 ... 1e-8
        # absolute (tools/experiments/probe_long_rough.py); the 1e-8 of the physical configurations is not claimed on this family
        assert np.abs(r["gam"].cpu().numpy() - gam_c).max() < 1e-7, (N, np.abs(r["gam"].cpu().numpy() - gam_c).max())


def test_division_form_count_with_rows_at_any_alignment(ctx):
    """k_sturm_count_div walks every system from its own 128-byte line boundary (the phase of its row's first element): rows with a
    leading dimension beyond N, arrays that start 8, 40 and 120 bytes into a line, g / c / f at DIFFERENT phases, batches that do not
    fill the last wave -- through the C ABI with device pointers, against the oracle's division-form count; N around the chunk
    boundaries (multiples of 16 +- 1), even N included (ibs.h: accepted here)."""
    import ctypes as C
    import torch
    from ibs_amd._lib import check, lib
    from oracle import c_oracle as co
    L = lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    ctx.set_option("sturm_form", 2)
    try:
        for N, n, ld, offs in ((513, 200, 513, (1, 5, 15)), (512, 67, 520, (0, 3, 9)), (81, 130, 96, (15, 15, 2)), (2049, 70, 2051, (7, 0, 11)),
                                (66, 64, 66, (1, 2, 3)), (97, 1, 97, (13, 6, 0))):
            h = 8 * np.pi / (N - 1)
            g = np.exp(rng.uniform(-1, 2, (n, N))); c = rng.uniform(-2.5, 3.5, (n, N)); f = np.exp(rng.uniform(0, 3, (n, N)))
            sh = rng.uniform(-3.0, 1.0, n)
            bufs, ptrs = [], []
            for arr, off in zip((g, c, f), offs):
                flat = torch.zeros(off + n * ld + 16, dtype=torch.float64, device=dev)
                flat[off:off + n * ld].view(n, ld)[:, :N] = torch.from_numpy(arr).to(dev)
                bufs.append(flat); ptrs.append(C.c_void_p(flat.data_ptr() + 8 * off))
            d_sh = torch.from_numpy(sh).to(dev); d_cnt = torch.full((n,), -1, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            check(L.ibs_sturm_count_f64(ctx._h, n, N, float(h), ptrs[0], ptrs[1], ptrs[2], ld, C.c_void_p(d_sh.data_ptr()),
                                        C.c_void_p(d_cnt.data_ptr()), 0), "ibs_sturm_count_f64")      # (0 = IBS_MEM_DEVICE)
            check(L.ibs_synchronize(ctx._h), "ibs_synchronize")
            assert "k_sturm_count_div" in ctx.last_launch()[0]
            want = co.count_above_batch(h, g, c, f, sh)
            assert np.array_equal(d_cnt.cpu().numpy(), want), (N, n, ld, offs, int((d_cnt.cpu().numpy() != want).sum()))
    finally:
        ctx.set_option("sturm_form", None)


def test_large_grid_geometry_fed_scan_with_theta0_derivative(ctx):
    """ibs_gamma_scan_f64 on a 4097-point grid: two field lines of the tests' smooth geometry family x 3 theta0 against the oracle --
    gam, lam, and the Hellmann-Feynman d gam / d theta0 of utils.py:1666-1680."""
    import torch
    from oracle import ballooning_oracle as bo
    from oracle import c_oracle as co
    N = 4097
    th = np.linspace(-4 * np.pi, 4 * np.pi, N)
    # smooth synthetic field-line geometry of the tests' own family (tests/helpers.py), two lines
    from tests.helpers import synthetic_fieldlines
    lines = synthetic_fieldlines(th)(0.6, np.array([0.3, 1.7]))            # (2, 8, N)
    dP = np.array([bo.dPdrho_of(lines[i, 2], lines[i, 7], lines[i, 0]) for i in range(2)])
    t0 = np.array([0.0, 0.4, 1.1])
    h = float(th[1] - th[0])
    r = ctx.gamma_scan(h, *[lines[:, k, :] for k in range(7)], dP, t0, want_X=True, want_dtheta0=True, want_info=True)
    assert "k_solve_gcf_long<double>" in ctx.last_launch()[0] and r["nbad"] == 0
    gam_c, lam_c, _ = co.gamma_scan(h, *[lines[:, k, :] for k in range(7)], dP, t0)
    assert np.abs(r["gam"] - gam_c).max() < 1e-8 and np.abs(r["lam"] - lam_c).max() < 1e-8      # (lam: both close to ~eps ||A||, ||A|| ~ 1e6-1e7 here)
    for i in range(2):
        for j in range(3):
            cv, gd = bo.fold_theta0(t0[j], lines[i, 2], lines[i, 3], lines[i, 4], lines[i, 5], lines[i, 6])
            gam, X, dX, gg, cc, ff = bo.gamma_ball_full(dP[i], th, lines[i, 0], lines[i, 1], cv, gd)
            gp = np.abs(lines[i, 1]); B = lines[i, 0]
            gdp = 2 * lines[i, 5] + 2 * t0[j] * lines[i, 6]
            jac = bo.hf_derivative(gam, X, dX, ff, gp * gdp / B, -dP[i] * lines[i, 3] / (gp * B), gdp / B ** 3 / gp)    # utils.py:1669-1680
            assert abs(r["dgam_dtheta0"][i, j] - jac) < 1e-7 * max(1.0, abs(jac)), (i, j, r["dgam_dtheta0"][i, j], jac)


def test_large_grid_scan_variants_agree_with_the_plain_scan(ctx):
    """The other geometry-fed entry points on a 2561-point grid (ibs.h: all of them accept grids up to 65,537 points): the scan with
    the per-surface first maximum (ibs_gamma_scan_argmax_f64; ball_scan.py:283-288), one (line, theta0) pair per point
    (ibs_gamma_points_f64; the final solve of ball_scan.py:322-339), and the warm-started scan (a guess is accepted and, on this path,
    not needed) return what ibs_gamma_scan_f64 returns, which the test above pins to the oracle."""
    import torch
    from oracle import ballooning_oracle as bo
    from tests.helpers import synthetic_fieldlines
    dev = torch.device("cuda:0")
    N = 2561
    th = np.linspace(-4 * np.pi, 4 * np.pi, N)
    h = float(th[1] - th[0])
    al = np.array([0.2, 0.9, 1.6, 2.4])
    lines = np.concatenate([synthetic_fieldlines(th)(0.5, al), synthetic_fieldlines(th)(0.7, al)])        # 2 surfaces x 4 lines
    dP = np.array([bo.dPdrho_of(lines[i, 2], lines[i, 7], lines[i, 0]) for i in range(8)])
    t0 = np.array([0.0, 0.5, 1.2])
    geo = [torch.from_numpy(np.ascontiguousarray(lines[:, k, :])).to(dev) for k in range(7)]
    dP_d, t0_d = torch.from_numpy(dP).to(dev), torch.from_numpy(t0).to(dev)
    base = ctx.gamma_scan(h, *geo, dP_d, t0_d, want_info=True)
    assert "k_solve_gcf_long<double>" in ctx.last_launch()[0] and int(((base["info"] >> 16) & 3).sum()) == 0
    am = ctx.gamma_scan_argmax(h, geo, dP_d, t0_d, 2)
    assert torch.equal(am["gam"], base["gam"]) and torch.equal(am["lam"], base["lam"])
    for s in range(2):
        blk = base["gam"][4 * s:4 * s + 4].reshape(-1)
        k = int(torch.argmax(blk))                                               # (first maximum: torch.argmax returns the first)
        assert float(am["pack"][s, 0]) == float(blk[k]) and int(am["pack"][s, 1]) == k
    pts = ctx.gamma_points(h, *geo, dP_d, torch.from_numpy(np.full(8, 0.5)).to(dev))
    assert torch.equal(pts["gam"], base["gam"][:, 1]) and torch.equal(pts["lam"], base["lam"][:, 1])
    warm = ctx.gamma_scan(h, *geo, dP_d, t0_d, lam_guess=base["lam"], guess_width=1e-3)
    assert torch.equal(warm["gam"], base["gam"])


def test_large_grid_obj_w_grad_against_the_oracle(ctx):
    """utils.py:1632-1728 (objective + Hellmann-Feynman gradient of one (alpha, theta0) point from the field lines at alpha - d/2,
    alpha, alpha + d/2) takes any grid length: on 2561 and 4097 points ibs_obj_w_grad_f64 is composed from the long-grid pieces
    (csrc/ibs_api.hip: launch_grad_long) -- batched call on host and on device arrays, and the literal drop-in obj_w_grad(x0, vs,
    rho, theta, vguess, sigma) that scipy's minimize would call (ball_scan.py:307-314) -- against the oracle's restatement."""
    import torch
    import ibs_amd
    from oracle import ballooning_oracle as bo
    from tests.helpers import synthetic_fieldlines
    dev = torch.device("cuda:0")
    for N in (2561, 4097):
        th = np.linspace(-4 * np.pi, 4 * np.pi, N)
        fl = synthetic_fieldlines(th)
        pts = [(0.6, 1.0, 0.4), (0.8, 2.2, 0.0), (0.5, 0.3, 1.1)]                       # (s, alpha, theta0)
        d = 0.004
        geo = np.stack([fl(s_, np.array([a_ - d / 2, a_, a_ + d / 2])) for s_, a_, _ in pts])     # (n_pts, 3, 8, N)
        t0 = np.array([p[2] for p in pts])
        val, jac, info = ctx.obj_w_grad(float(th[1] - th[0]), geo, t0, d, want_info=True)
        assert (((info >> 16) & 3) == 0).all()
        for k, (s_, a_, t_) in enumerate(pts):
            vo, jo = bo.obj_w_grad_lines(th, t_, geo[k, 0], geo[k, 1], geo[k, 2], d)
            assert abs(val[k] - vo) < 1e-8, (N, k, val[k], vo)
            assert np.abs(jac[k] - jo).max() < 1e-6 * max(1.0, np.abs(jo).max()), (N, k, jac[k], jo)
        v2, j2 = ctx.obj_w_grad(float(th[1] - th[0]), torch.from_numpy(geo).to(dev), torch.from_numpy(t0).to(dev), d)
        assert np.array_equal(v2.cpu().numpy(), val) and np.array_equal(j2.cpu().numpy(), jac)
    drop = ibs_amd.make_obj_w_grad(lambda vs, s, al, theta: fl(s, al), ctx=ctx)
    v, j = drop((1.0, 0.4), None, 0.6, th, None, 0.42)
    assert abs(v - val[0]) < 1e-11 and np.abs(j - jac[0]).max() < 1e-9        # (the drop-in takes h from the whole grid, the calls above from its first step)


def test_large_grid_scan_driver_end_to_end(ctx):
    """ball_scan.py:248-339 on a 2561-point grid (mpol = 20, ntor = 16 in the reference's rule): coarse scan, first maximum,
    L-BFGS-B refinement under scipy's minimize with the drop-in obj_w_grad (the reference's own optimizer call, ball_scan.py:305-314),
    final solve -- the GPU-backed driver against the same driver on the oracle."""
    import ibs_amd
    from oracle import ballooning_oracle as bo
    from tests.helpers import OracleContext, synthetic_fieldlines
    N = 2561
    th = np.linspace(-4 * np.pi, 4 * np.pi, N)
    fl = synthetic_fieldlines(th)
    rho = np.array([0.55, 0.8])
    gpu = ibs_amd.BallooningScan(ctx, fl, th, rho, nalpha=4, ntheta0=3)
    cpu = ibs_amd.BallooningScan(OracleContext(), fl, th, rho, nalpha=4, ntheta0=3)
    assert np.abs(gpu.coarse() - cpu.coarse()).max() < 1e-8
    tg, ag, gg = gpu.run(refine=True)
    tc, ac, gc = cpu.run(refine=True)
    assert np.abs(gg - gc).max() < 1e-8, (gg, gc)
    assert np.abs(tg - tc).max() < 1e-4 and np.abs(ag - ac).max() < 1e-4


def test_large_grid_resident_scan_from_wout_tables(ctx):
    """The HBM-resident worker (geometry from the wout tables of tests/golden/G8, coarse scan with the fused first maximum, start
    points, refinement, final solve) on a 2561-point grid: the refinement runs the library's L-BFGS-B state machines on the host
    with ONE batched geometry + ibs_obj_w_grad_f64 launch per round (ibs_refine_f64 holds the register-resident evaluation
    kernel, N <= 2050) and reaches the stopping point of the per-surface scipy path of ball_scan.py:307-314."""
    import torch
    import ibs_amd
    G = os.path.join(ROOT, "tests", "golden")
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    svals = np.array([0.6, 0.9])
    tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
    th = np.linspace(-4 * np.pi, 4 * np.pi, 2561)
    scan = ibs_amd.BallooningScan(ctx, None, th, svals, nalpha=8, ntheta0=5, tables=tabs, device=torch.device("cuda:0"))
    t0, al, gam = scan.run()
    assert np.isfinite(gam).all() and scan.last_refine["rounds"] >= 2
    tabs_c = scan.coarse()
    for k, s_ in enumerate(svals):
        a0, th0 = ibs_amd.pick_start(tabs_c[k], scan.alpha_scan, scan.theta0_scan)[:2]
        t_opt, a_opt, gam_opt, res = scan.refine(s_, a0, th0)
        assert gam[k] >= tabs_c[k].max() - 1e-9
        assert abs(gam[k] - gam_opt) < 1e-8, (k, gam[k], gam_opt)


def test_large_grid_adjoint_step_against_the_oracle_pipeline(ctx):
    """One optimizer iteration of the reference's workflow (sims_runner_NCSX.py:249-261: equilibria x surfaces -> refined maxima ->
    objective and gradient) on a 2561-point grid: AdjointStep.run() with two emulated equilibria, every (equilibrium, surface)
    pair against the oracle on every link (tests/helpers.py: numpy geometry, C-oracle scan, scipy's L-BFGS-B on the oracle
    objective, oracle final solve)."""
    import torch
    import ibs_amd
    import bench
    from tests.helpers import oracle_surface_pipeline
    dev = torch.device("cuda:0")
    wout0 = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
    wouts, steps, x0 = bench.emulated_equilibria(wout0)
    wouts, steps = wouts[:2], steps[:1]
    svals = np.array([0.6, 0.9])
    na, nt0 = 6, 4
    th = np.linspace(-4 * np.pi, 4 * np.pi, 2561)
    step = ibs_amd.AdjointStep(ctx, th, svals, dev, nalpha=na, ntheta0=nt0, gamma_thresh=-2.0e-4, prefac=50.0)
    out = step.run(wouts, 0.8 + 0.01 * np.arange(2), steps)
    gam = out["gam"]
    assert gam.shape == (2, 2) and np.isfinite(gam).all() and np.isfinite(out["dfobj"]).all()
    for q in range(2):
        for js in range(2):
            ref = oracle_surface_pipeline(wouts[q], float(svals[js]), th, na, nt0, step.del_alpha)
            assert abs(ref["gam"] - gam[q, js]) < 1e-8, (q, js, ref["gam"], gam[q, js])


# ---------------------------------------------------------------------------------------------- nearest-sigma report
def test_nearest_sigma_divergence_is_reported(ctx):
    """utils.py:1597 takes the eigenpair NEAREST sigma0 (ARPACK shift-invert); the drop-in always takes lam_max.  A strongly driven
    s-alpha line (dPdrho = -4: lam_max = 2.09 > 0.42) is the one kind of case where the two differ: there the reference formulation
    (oracle: dense matrix + eigs(sigma=sigma0), utils.py:1582-1624 restated) returns ANOTHER eigenpair's growth rate, the drop-in
    returns lam_max's, says so (NearestSigmaWarning, info["above_sigma0"], status bit 4 at the C ABI), and with sigma0 above
    lam_max -- or on the ordinary weakly driven line -- the two agree to the stated 1e-8 and nothing is flagged."""
    import warnings
    import ibs_amd
    from oracle import ballooning_oracle as bo
    N = 257
    th = bo.theta_grid(N)
    g, c0 = bo.salpha_gc(th, 1.0, 0.8, 0.0)
    one = np.ones(N)
    # (a) lam_max >= sigma0: flagged, and genuinely different from the nearest-sigma eigenpair
    info = {}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = ibs_amd.gamma_ball_full(-4.0, th, one, one, c0, g, sigma0=0.42, ctx=ctx, info=info)
    assert any(issubclass(x.category, ibs_amd.NearestSigmaWarning) for x in w)
    assert info["above_sigma0"] is True and info["lam"] > 2.0 and info["status"] == 0
    ref_near = bo.gamma_ball_full_dense_arpack(-4.0, th, one, one, c0, g, sigma0=0.42)      # what upstream returns
    ref_max = bo.gamma_ball_full(-4.0, th, one, one, c0, g)                                  # exact top eigenpair
    assert abs(out[0] - ref_max[0]) < 1e-8 and abs(out[0] - ref_near[0]) > 0.1, (out[0], ref_max[0], ref_near[0])
    # (b) the same line with sigma0 above lam_max: upstream's eigenpair IS lam_max's; nothing flagged
    info = {}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = ibs_amd.gamma_ball_full(-4.0, th, one, one, c0, g, sigma0=2.5, ctx=ctx, info=info)
    assert not w and info["above_sigma0"] is False
    assert abs(out[0] - bo.gamma_ball_full_dense_arpack(-4.0, th, one, one, c0, g, sigma0=2.5)[0]) < 1e-8
    # (c) the reference's own regime (dPdrho = -1, lam_max = 0.109 < 0.42): agree, not flagged
    info = {}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = ibs_amd.gamma_ball_full(-1.0, th, one, one, c0, g, ctx=ctx, info=info)
    assert not w and info["above_sigma0"] is False and abs(info["lam"] - 0.10876115422233879) < 1e-10
    assert abs(out[0] - bo.gamma_ball_full_dense_arpack(-1.0, th, one, one, c0, g)[0]) < 1e-8
    # (d) the flag at the C ABI, batched: option "sigma0" + lam and info outputs; the option is per context and leaves no trace
    G2 = np.stack([g, g]); C2 = np.stack([4.0 * c0, c0])
    ctx.set_option("sigma0", 0.42)
    r = ctx.solve_gcf(th[1] - th[0], G2, C2, G2, want_info=True)
    ctx.set_option("sigma0", None)
    assert ((r["info"] >> 16) & 16).tolist() == [16, 0] and r["nbad"] == 0
    r = ctx.solve_gcf(th[1] - th[0], G2, C2, G2, want_info=True)
    assert ((r["info"] >> 16) & 16).tolist() == [0, 0]


def test_G10_top_pair_1p5e10_apart_is_not_mistaken(ctx):
    """tests/golden/G10_rough_pair_1025.npz: system 245,944 of the config-5 rough family (seed 20240 + 1024, 10^6 systems; captured by
    tools/experiments/find_bad_system.py), whose two largest eigenvalues lie 1.5e-10 ||A|| apart.  In the round-5 kernels a bisection
    midpoint landed within 2 tol of lam_2, the prefix-product sweep's count there read 0 instead of 1, that shift became the
    bracket's upper end and the solve closed on lam_2 -- consistently: the Rayleigh polish agreed (tools/probe_direct.hip shows the
    25 sweeps).  The twisted factorisation at the last shift counts the eigenvalue the sweeps lost (WaveSolver::twisted<true>:
    extra_above), the system is closed again in division form, and every raw-kernel form must return lam_max."""
    import torch
    from oracle import c_oracle as co
    d = np.load(os.path.join(G, "G10_rough_pair_1025.npz"))
    g, c, f = d["g"], d["c"], d["f"]
    N = len(g); h = 8 * np.pi / (N - 1)
    lam_c = co.lam_batch(h, g[None], c[None], f[None])[0]
    assert abs(lam_c - float(d["lam_max"])) < 1e-15 and lam_c - float(d["lam_returned_round5"]) > 8e-5
    e = 0.5 * (g[:-1] + g[1:]) / h ** 2
    nA = float(((np.abs(c[1:-1] - (e[:-1] + e[1:])) + e[:-1] + e[1:]) / f[1:-1]).max())
    tol = 4 * N * 2.220446049250313e-16 * nA
    dev = torch.device("cuda:0")
    n = 4096                                              # (a batch large enough for the library to pick the big-batch forms itself)
    G_, C_, F_ = (torch.from_numpy(np.tile(a, (n, 1))).to(dev) for a in (g, c, f))
    seen = set()
    for direct, want_gam in ((1, True), (1, False), (0, True), (0, False), (None, True)):
        ctx.set_option("gcf_direct", direct)
        r = ctx.solve_gcf(h, G_, C_, F_, want_info=True, want_gam=want_gam)
        seen.add(ctx.last_launch()[0])
        lam = r["lam"].cpu().numpy()
        assert np.abs(lam - lam_c).max() < tol, (direct, want_gam, ctx.last_launch()[0], float(np.abs(lam - lam_c).max()), tol)
        assert int((((r["info"] >> 16) & 3) != 0).sum()) == 0
        if want_gam:                                      # the growth rate of a re-closed system comes from sweeps AT lam_max
            gam_c = co.solve_gcf(h, g, c, f)[0]
            assert np.abs(r["gam"].cpu().numpy() - gam_c).max() < 1e-6 * max(1.0, abs(gam_c))
    ctx.set_option("gcf_direct", None)
    assert any("k_solve_gcf_direct" in k for k in seen) and any(k.endswith("k_solve_gcf<double, 16>") for k in seen), seen
    ctx.set_option("reclose", 0)
    ctx.set_option("gcf_direct", 1)
    r0 = ctx.solve_gcf(h, G_[:8], C_[:8], F_[:8])["lam"].cpu().numpy()
    ctx.set_option("reclose", None); ctx.set_option("gcf_direct", None)
    print("G10 with the checks off: lam - lam_max = %.3e (round 5: %.3e = lam_2 - lam_max)" % (float(r0[0] - lam_c), float(d["lam_returned_round5"]) - lam_c))
