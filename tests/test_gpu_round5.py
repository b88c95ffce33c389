"""GPU, round 5: the mode-row limit of the geometry row kernels at the C ABI, and the multi-rank code of the
FINAL tree under the driver's `pytest -m gpu`: `bench.py --gpus 2` as fresh child processes sharing the one
GPU of the box -- the sharded legs, the gather, the compact line -- and the watchdog when a rank dies."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
from bench import c5_family, norm_a  # noqa: E402  (the synthetic inputs of configs[4] live with the bench)


@pytest.fixture(scope="module")
def ctx():
    import ibs_amd
    c = ibs_amd.Context(0)
    yield c
    c.close()


def _wout_with_a_long_row(nmax):
    """the shipped equilibrium with its m = 0 row of the first mode list extended to n = 0 .. nmax nfp (small coefficients
    on the added modes: a data manipulation for shape coverage, both sides evaluate the same formulas)"""
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    xm, xn = wout["xm"], wout["xn"]
    nfp = int(np.min(np.abs(xn[xn != 0])))
    have = set(zip(xm.astype(int).tolist(), xn.astype(int).tolist()))
    extra = [(0, n * nfp) for n in range(nmax + 1) if (0, n * nfp) not in have]
    em = np.array([e[0] for e in extra], dtype=float); en = np.array([e[1] for e in extra], dtype=float)
    order = np.lexsort((np.concatenate([xn, en]), np.concatenate([xm, em])))
    rng = np.random.default_rng(3)
    prof = np.linspace(0, 1, wout["rmnc"].shape[1]) ** 2
    w = dict(wout)
    for k in ("rmnc", "zmns", "lmns"):
        add = 1e-5 * np.abs(wout[k]).max() * rng.standard_normal((len(extra), 1)) * prof[None, :] / (1.0 + np.abs(en[:, None]) / nfp)
        w[k] = np.concatenate([wout[k], add])[order]
    w["xm"], w["xn"] = np.concatenate([xm, em])[order], np.concatenate([xn, en])[order]
    return w


def test_geometry_rows_longer_than_the_table_image(ctx):
    """the one-lane-per-point image holds 64 pair indices for the root solve; a caller's row with more
    (here: m = 0 with n = 0 .. 70 nfp as ONE row of 71 modes) used to write past the image.  Now: host tables and
    device-resident tables with such a row run on the one-sincos-per-mode kernel and agree with the numpy oracle; the same
    tables with the row split (mode_rows' default) stay on the row kernels; rows outside the mode list are an argument error;
    and device-resident rows REWRITTEN IN PLACE behind the library's per-table check give NaN lines, not a fault."""
    import torch
    import ibs_amd
    from ibs_amd.geometry import mode_rows
    from oracle import geometry_oracle as go
    w = _wout_with_a_long_row(70)
    svals = np.array([0.4, 0.85])
    otab = go.surface_tables_from_wout(w, svals)
    dev = torch.device("cuda:0")
    N = 513; th = ibs_amd.theta_grid(N)
    surf = np.array([0, 1, 1, 0], dtype=np.int32); al = np.array([0.2, 1.4, 2.6, 3.1])
    ref = np.stack([go.fieldline_geometry(otab, int(s), np.array([a]), th)[0] for s, a in zip(surf, al)])      # (lines, 8, N)
    scale = np.abs(ref).max(axis=2, keepdims=True)
    nl = 600                                                               # enough lines for a one-lane-per-point form
    s2 = (np.arange(nl) % 2).astype(np.int32); a2 = np.linspace(0, np.pi, nl)
    s2[:4] = surf; a2[:4] = al

    split = ibs_amd.SurfaceTables.from_wout(w, svals)                      # rows of at most 64 modes: the fast path
    assert split.rows_mn[:, 1].max() <= 64
    r = ctx.fieldline_geometry(split, s2, a2, th, device=dev)
    assert "k_geo_rows" in ctx.last_launch()[0], ctx.last_launch()
    assert (np.abs(r["geo"][:, :4].cpu().numpy().transpose(1, 0, 2) - ref) / scale).max() < 1e-10

    long_ = ibs_amd.SurfaceTables.from_wout(w, svals)
    long_.rows_mn, long_.dn_mn = mode_rows(long_.xm, long_.xn, max_len=1000)
    assert long_.rows_mn[:, 1].max() == 71
    for device in (None, dev):                                             # host pointers, device pointers
        r = ctx.fieldline_geometry(long_, surf, al, th, device=device)
        assert ctx.last_launch()[0] == "ibs::k_fieldline_geometry", ctx.last_launch()
        got = (r["geo"].cpu().numpy() if device is not None else r["geo"]).transpose(1, 0, 2)
        assert (np.abs(got - ref) / scale).max() < 1e-10
    # the refinement takes the same tables (its geometry goes through the same check)
    xo, fo, ne, rounds = ctx.refine(long_, np.array([0, 1], dtype=np.int32), np.array([[1.0, 0.5], [2.0, 0.3]]), th, device=dev)
    xs, fs, _, _ = ctx.refine(split, np.array([0, 1], dtype=np.int32), np.array([[1.0, 0.5], [2.0, 0.3]]), th, device=dev)
    assert np.all(np.isfinite(fo)) and np.abs(fo - fs).max() < 1e-9

    bad = ibs_amd.SurfaceTables.from_wout(w, svals)
    bad.rows_mn = bad.rows_mn.copy(); bad.rows_mn[-1, 1] += 5              # runs past the end of the mode list
    with pytest.raises(ibs_amd.IbsError, match="outside"):
        ctx.fieldline_geometry(bad, surf, al, th)
    with pytest.raises(ibs_amd.IbsError, match="outside"):
        ctx.fieldline_geometry(bad, surf, al, th, device=dev)

    # second line of defence: rows changed in place after the library has checked this table set
    sneak = ibs_amd.SurfaceTables.from_wout(w, svals)
    r = ctx.fieldline_geometry(sneak, s2, a2, th, device=dev)
    name = ctx.last_launch()[0]
    assert "k_geo_rows<2, 1," in name or "k_geo_rows<1, 1," in name, name
    assert torch.isfinite(r["geo"]).all()
    d_rows = ctx._device_tables(sneak, dev)[7]
    merged = torch.from_numpy(np.ascontiguousarray(mode_rows(sneak.xm, sneak.xn, max_len=1000)[0]))
    assert len(merged) < len(sneak.rows_mn)
    keep = d_rows.clone()
    d_rows[:len(merged)] = merged.to(dev)                                  # first row now 71 modes; the library is not told
    d_rows[len(merged):, 0] = 0; d_rows[len(merged):, 1] = 1               # (surplus rows: valid one-mode rows)
    r = ctx.fieldline_geometry(sneak, s2, a2, th, device=dev)
    torch.cuda.synchronize()
    assert torch.isnan(r["geo"][:, :, :512]).all()                        # flagged by k_geo_prepare, NaN from k_geo_rows
    d_rows.copy_(keep)
    r = ctx.fieldline_geometry(sneak, s2, a2, th, device=dev)
    assert torch.isfinite(r["geo"]).all()


def _bench(args, env_extra, timeout):
    env = dict(os.environ, IBS_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=timeout)
    return p, time.time() - t0


def test_two_ranks_share_the_gpu_through_bench(tmp_path):
    """`python bench.py --gpus 2` on the one-GPU box (IBS_BENCH_SHARE_GPU=1: both ranks on device 0, collectives through
    gloo -- RCCL refuses two ranks on one device): the parent spawns fresh rank processes (this pytest process only waits),
    the headline's per-step gather and the three sharded legs run on two ranks and check themselves against one rank, the
    ONE stdout line parses, is under 6,000 bytes and says so."""
    detail = tmp_path / "detail.json"
    p, dt = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-stress"], {"IBS_BENCH_DETAIL": str(detail)}, 900)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert len(lines[0]) < 6000
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["config"]["ranks_in_collective"] == 2 and o["config"]["allgather_roundtrip_ok"] is True
    for leg in ("ncsx_c2_sharded", "ncsx_c2_sharded_refined", "c4_adjoint_step_sharded"):
        assert o[leg]["checks_passed"] is True, (leg, o[leg])
    assert o["ncsx_c2_sharded"]["gathered_equals_one_gpu_bitwise"] is True
    assert o["value"] > 0 and "roofline" in o and "dropped_for_length" not in o
    full = json.load(open(detail))
    assert full["value"] == o["value"] and full["c4_adjoint_step_sharded"]["checks_passed"] is True
    print("2 ranks on one GPU: %.0f s; c2 sharded %.2f ms / pass, c4 sharded %.2f ms / step" % (
        dt, o["ncsx_c2_sharded"]["ms_per_pass"], o["c4_adjoint_step_sharded"]["ms_per_step"]))


def test_a_dying_rank_ends_the_run_nonzero():
    """rank 1 kills itself (IBS_BENCH_DIE_RANK / _AFTER: a test hook in bench.py) inside the first sharded leg, while rank 0
    waits in that leg's collective: the parent notices the dead child, stops the other rank and leaves non-zero -- well
    inside the watchdog period (300 s), not after a hang."""
    p, dt = _bench(["--gpus", "2", "--steps", "20", "--warmup", "5", "--no-stress"],
                   {"IBS_BENCH_DIE_RANK": "1", "IBS_BENCH_DIE_AFTER": "headline"}, 600)
    assert p.returncode != 0, p.stdout[-1000:]
    assert dt < 300, dt
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) <= 1                         # (rank 0 may or may not have got its headline line out; never two)


@pytest.mark.parametrize("N", [969, 1025, 2049, 643, 1409, 1793, 1921])   # M = 16, 16, 32, 11 and the two-waves-per-SIMD forms 22, 28, 30
def test_rows_straight_from_global_memory_agree_with_the_staged_kernels(ctx, N):
    """k_solve_gcf_direct (round 5: one wave per system, every lane reads its chunk of g, c, f from global memory; no LDS, so the
    registers set the occupancy) against the LDS-staged kernels it replaces for big batches on long grids: the same solver on
    the same numbers -- lam agrees to the certified bracket, gam / X / dX of the smooth family to 1e-10 / 1e-6 / 1e-5, FP64 and
    FP32-with-growth-rate, eigenvalue-only calls, a flagged system included;
    a sample against the C oracle (utils.py:1550-1624 restated)."""
    import torch
    from oracle import c_oracle as co
    dev = torch.device("cuda:0")
    n = 3000
    h, g, c, f = c5_family(dev, "rough", n, N, seed=77 + N)
    hs, gs, cs, fs = c5_family(dev, "smooth", n, N, seed=78 + N)
    g = torch.cat([g, gs]); c = torch.cat([c, cs]); f = torch.cat([f, fs])
    f[17, N // 2] = -1.0                                                   # an invalid system: flagged (status 2), neighbours untouched
    out = {}
    for direct in (0, 1):
        ctx.set_option("gcf_direct", direct)
        out[direct] = ctx.solve_gcf(h, g, c, f, want_X=True, want_info=True)
        name = ctx.last_launch()[0]
        assert ("k_solve_gcf_direct" in name) == (direct == 1), name
        out[direct, "f32"] = ctx.solve_gcf(h, g.float(), c.float(), f.float(), want_info=True, dtype=np.float32)
        assert ("k_solve_gcf_direct" in ctx.last_launch()[0]) == (direct == 1), ctx.last_launch()
        out[direct, "lam"] = ctx.solve_gcf(h, g, c, f, want_gam=False)      # eigenvalues only
        ctx.set_option("f32_lam", 1)                                        # FP32 eigenvalues only: all-FP32 iteration + FP64 certificate
        out[direct, "lam32"] = ctx.solve_gcf(h, g.float(), c.float(), f.float(), want_gam=False, want_info=True, dtype=np.float32)
        assert ("k_solve_gcf_f32lam_direct" in ctx.last_launch()[0]) == (direct == 1), ctx.last_launch()
        ctx.set_option("f32_lam", None)
    ctx.set_option("gcf_direct", None)
    a, b = out[0], out[1]
    ok = torch.ones(2 * n, dtype=torch.bool, device=dev); ok[17] = False
    assert int((b["info"][17] >> 16)) == 2 and int((((b["info"][ok] >> 16) & 3) != 0).sum()) == 0
    assert int((a["info"][17] >> 16)) == 2 and int((((a["info"][ok] >> 16) & 3) != 0).sum()) == 0
    # Two compilations of one solver: the shift iterations agree to the certified bracket (4 x 64 eps ||A||), not bit for bit (the
    # compiler contracts multiply-adds of the set-up differently around LDS reads and around global loads), and on the rough
    # family the counts of the scan-form recurrence are certificates for a matrix perturbed by ~N eps (tests/test_gpu_configs.py).
    nA_all = norm_a(h, g, c, torch.where(f > 0, f, torch.ones_like(f)))
    smooth_rows = torch.arange(2 * n, device=dev) >= n
    rel = (a["lam"] - b["lam"]).abs() / nA_all
    print("N = %d: lam bitwise equal on %.1f %% of the systems, max |dlam| / ||A|| %.1e (smooth %.1e)" % (
        N, 100.0 * float((a["lam"][ok] == b["lam"][ok]).double().mean()), float(rel[ok].max()), float(rel[ok & smooth_rows].max())))
    assert float(rel[ok & smooth_rows].max()) < 1e-13 and float(rel[ok].max()) < max(1e-11, 1e-13 * N) * 2
    assert float(((out[0, "lam"]["lam"] - b["lam"]).abs() / nA_all)[ok].max()) < max(1e-11, 1e-13 * N) * 2
    assert float(((out[1, "lam"]["lam"] - b["lam"]).abs() / nA_all)[ok].max()) < max(1e-11, 1e-13 * N) * 2
    sm = ok & smooth_rows                                                  # growth rate / eigenfunction: pinned on the smooth family
    assert float((a["gam"][sm] - b["gam"][sm]).abs().max()) < 1e-10
    # (eigenfunctions of two certified-equal eigenvalues differ by ~|dlam| / gap: 1e-7 seen at N = 1793)
    assert float((a["X"][sm] - b["X"][sm]).abs().max()) < 1e-6 and float((a["dX"][sm] - b["dX"][sm]).abs().max()) < 1e-5
    a32, b32 = out[0, "f32"], out[1, "f32"]
    assert float(((a32["lam"].double() - b32["lam"].double()).abs() / nA_all)[ok].max()) < 1e-6
    assert float((a32["gam"][sm].double() - b32["gam"][sm].double()).abs().max()) < 1e-6
    # FP32 eigenvalues alone: every result of both forms within (N_zeta + 4) eps32 ||A|| of the FP64 solve of the same FP32-valued systems
    g32w, c32w, f32w = g.float().double(), c.float().double(), f.float().double()
    r64 = ctx.solve_gcf(h, g32w, c32w, f32w, want_gam=False)["lam"]
    nA32 = norm_a(h, g32w, c32w, torch.where(f32w > 0, f32w, torch.ones_like(f32w)))
    for d in (0, 1):
        el = ((out[d, "lam32"]["lam"].double() - r64).abs() / nA32 / 1.1920929e-07)[ok]
        assert float(el.max()) <= N - 1 + 4, (d, float(el.max()))
        assert int((((out[d, "lam32"]["info"][ok] >> 16) & 3) != 0).sum()) == 0
    pick = np.array([0, 1, 5, 100, n - 1, n, n + 3, 2 * n - 1])
    pk = torch.from_numpy(pick).to(dev)
    gam_c, lam_c, _ = co.solve_gcf_batch(h, g[pk].cpu().numpy(), c[pk].cpu().numpy(), f[pk].cpu().numpy())
    nA = norm_a(h, g[pk], c[pk], f[pk]).cpu().numpy()
    assert (np.abs(b["lam"][pk].cpu().numpy() - lam_c) / nA).max() < 1e-11
    smooth = pick >= n                                                     # (the growth rate is pinned on the smooth family: SURVEY H4)
    assert np.abs(b["gam"][pk].cpu().numpy() - gam_c)[smooth].max() < 1e-8


def test_direct_form_is_what_big_batches_on_long_grids_get(ctx):
    """the dispatch rule (ibs_api.hip use_direct): N = 1025, one wave per system -- a batch beyond what the three-row staging holds
    in flight (5 waves per CU) runs k_solve_gcf_direct, a small one keeps the staged kernel (coalesced staging = lower latency)"""
    import torch
    dev = torch.device("cuda:0")
    N = 1025
    h, g, c, f = c5_family(dev, "smooth", 8192, N, seed=5)
    ctx.solve_gcf(h, g, c, f)
    assert "k_solve_gcf_direct<double, 16, double>" in ctx.last_launch()[0], ctx.last_launch()
    ctx.solve_gcf(h, g[:512], c[:512], f[:512])
    assert "k_solve_gcf<double, 16>" in ctx.last_launch()[0], ctx.last_launch()
    ctx.solve_gcf(h, g.float(), c.float(), f.float(), dtype=np.float32)
    assert "k_solve_gcf_direct<double, 16, float>" in ctx.last_launch()[0], ctx.last_launch()
    # ragged batch: a system count that is not a multiple of the waves per block -- the last block's spare waves must not write,
    # and every system's result must not depend on the batch it is solved in (same kernel: bit for bit)
    full = ctx.solve_gcf(h, g, c, f, want_X=True, want_info=True)
    for n_r in (8191, 4097, 1283):
        part = ctx.solve_gcf(h, g[:n_r], c[:n_r], f[:n_r], want_X=True, want_info=True)
        assert "k_solve_gcf_direct" in ctx.last_launch()[0]
        for k in ("lam", "gam", "X", "dX", "info"):
            assert torch.equal(part[k], full[k][:n_r]), (n_r, k)
    lam32 = ctx.solve_gcf(h, g.float()[:4097], c.float()[:4097], f.float()[:4097], dtype=np.float32, want_gam=False)["lam"]
    ctx.set_option("f32_lam", 1)
    lam32b = ctx.solve_gcf(h, g.float()[:4097], c.float()[:4097], f.float()[:4097], dtype=np.float32, want_gam=False)["lam"]
    assert "k_solve_gcf_f32lam_direct" in ctx.last_launch()[0] and lam32b.shape == (4097,) and torch.isfinite(lam32b).all()
    ctx.set_option("f32_lam", None)
    assert float((lam32.double() - lam32b.double()).abs().max()) < 1e-4


@pytest.mark.parametrize("N", [131, 321, 513, 641])
def test_fp32_eigenvalues_alone_on_short_grids(ctx, N):
    """FP32 eigenvalue-only requests where the sub-wave forms exist (ibs_api.hip, `wide_lam`): a batch that would get the 32-lane
    form runs the all-FP32 iteration + FP64 certificate with its rows read from global memory (k_solve_gcf_f32lam_direct, built
    from 3 rows per lane), the 16-lane regime (N <= 258, big batches) keeps the FP64 sub-wave solver; both within
    (N_zeta + 4) eps32 ||A|| of the FP64 solve on EVERY system, the only status the informational bit 2; ragged batches included."""
    import torch
    dev = torch.device("cuda:0")
    n = 40003                                                              # (not a multiple of the four waves of a block)
    M = (N - 2 + 63) // 64
    for family in ("smooth", "rough"):
        h, g, c, f = c5_family(dev, family, n, N, seed=77 + N)
        r64 = ctx.solve_gcf(h, g, c, f)
        nA = norm_a(h, g, c, f)
        g32, c32, f32 = g.float(), c.float(), f.float()
        r = ctx.solve_gcf(h, g32, c32, f32, want_info=True, dtype=np.float32, want_gam=False)
        name = ctx.last_launch()[0]
        if N <= 258:
            assert "k_solve_gcf_g<double, %d, 16, float>" % ((N - 2 + 15) // 16) in name, name
        else:
            assert "k_solve_gcf_f32lam_direct<%d>" % M in name, name
        st = r["info"] >> 16
        assert int(((st & ~(4 | 8)) != 0).sum()) == 0      # (informational: bit 2 re-solved in FP64, bit 3 re-closed in division form)
        el = (r["lam"].double() - r64["lam"]).abs() / nA
        assert float(el.max()) <= (N + 3) * 1.1920929e-07, (family, float(el.max()) / 1.1920929e-07)
        # forced: the all-FP32 direct form in the 16-lane regime as well; a small batch keeps the staged all-FP32 kernel
        ctx.set_option("f32_lam", 1)
        r1 = ctx.solve_gcf(h, g32, c32, f32, want_info=True, dtype=np.float32, want_gam=False)
        assert "k_solve_gcf_f32lam_direct<%d>" % M in ctx.last_launch()[0], ctx.last_launch()
        rs = ctx.solve_gcf(h, g32[:300], c32[:300], f32[:300], want_info=True, dtype=np.float32, want_gam=False)
        assert "k_solve_gcf<float, %d>" % M in ctx.last_launch()[0], ctx.last_launch()
        ctx.set_option("f32_lam", None)
        e1 = (r1["lam"].double() - r64["lam"]).abs() / nA
        assert float(e1.max()) <= (N + 3) * 1.1920929e-07 and int((((r1["info"] >> 16) & ~4) != 0).sum()) == 0
        # the same all-FP32 arithmetic from LDS and from global memory: within the certificate's tolerance of each other
        es = (rs["lam"].double() - r1["lam"][:300].double()).abs() / nA[:300]
        assert float(es.max()) <= 2 * (N + 3) * 1.1920929e-07
