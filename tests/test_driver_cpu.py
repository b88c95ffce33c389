"""CPU: host driver logic (sharding, argmax rule, refinement plumbing, gather, on-disk contract) with the
oracle standing in for the GPU, including a world_size-2 gloo run."""
import json
import os
import socket
import sys

import numpy as np
import pytest

import ibs_amd
from oracle import ballooning_oracle as bo
from tests.helpers import OracleContext, synthetic_fieldlines

G = os.path.join(os.path.dirname(__file__), "golden")


def test_shard_and_gather_single_rank():
    assert ibs_amd.shard_surfaces(5, 0, 2) == [0, 2, 4] and ibs_amd.shard_surfaces(5, 1, 2) == [1, 3]
    loc = np.arange(12.0).reshape(4, 3)
    assert np.array_equal(ibs_amd.gather_surfaces(loc, 4, 0, 1), loc)


def test_pick_start_rule_matches_reference_trace():
    g5 = np.load(os.path.join(G, "G5_scan_trace.npz"))
    a0, t0, sig, ij = ibs_amd.pick_start(g5["gam_table"], g5["alpha_scan"], g5["theta0_scan"])
    assert ij == tuple(int(v) for v in g5["argmax"])
    assert abs(sig - float(g5["sigma0"])) < 1e-15
    assert abs(a0 - g5["trace"][0, 0]) < 1e-15 and abs(t0 - g5["trace"][0, 1]) < 1e-15
    assert ibs_amd.pick_start(np.zeros((3, 4)), np.arange(3.0), np.arange(4.0)) == (0.0, 0.0, 0.05, None)
    tie = np.array([[1.0, 2.0], [2.0, 0.0]])
    assert ibs_amd.pick_start(tie, np.array([5.0, 6.0]), np.array([7.0, 8.0]))[3] == (0, 1)


def test_coarse_scan_on_golden_geometry():
    g5 = np.load(os.path.join(G, "G5_scan_trace.npz"))
    geo8 = np.concatenate([g5["geo"], np.zeros((24, 1, 513))], axis=1)
    # gbdrift chosen so that the driver's dPdrho formula reproduces the stored dPdrho
    geo8[:, 7] = geo8[:, 2] + 2 * g5["dPdrho"][:, None] / geo8[:, 0] ** 2
    rows = [0, 9, 15]
    th = bo.theta_grid(513)
    scan = ibs_amd.BallooningScan(OracleContext(), lambda s, al: geo8[rows], th, [float(g5["s"])], nalpha=3, ntheta0=15)
    tab = scan.coarse()[0]
    assert np.abs(tab - g5["gam_table"][rows]).max() < 1e-8


def test_refinement_improves_and_matches_bruteforce():
    N = 257
    th = bo.theta_grid(N)
    fl = synthetic_fieldlines(th)
    scan = ibs_amd.BallooningScan(OracleContext(), fl, th, [0.5], nalpha=6, ntheta0=5)
    t0, al, gam = scan.run(refine=True)
    tab = scan.coarse()[0]
    assert gam[0] >= tab.max() - 1e-10                        # L-BFGS-B never ends below its start
    assert 0 <= al[0] <= np.pi and 0 <= t0[0] <= np.pi / 2
    v, j = scan.obj_w_grad((al[0], t0[0]), 0.5)
    assert abs(-v - gam[0]) < 1e-10


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N = 129
    th = bo.theta_grid(N)
    scan = ibs_amd.BallooningScan(OracleContext(), synthetic_fieldlines(th), th, np.linspace(0.5, 0.95, 5),
                                  nalpha=4, ntheta0=3, rank=rank, world=world, dist=dist)
    out = scan.run(refine=False)
    q.put((rank, [o.tolist() for o in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gather_equals_single_rank():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    N = 129
    th = bo.theta_grid(N)
    single = ibs_amd.BallooningScan(OracleContext(), synthetic_fieldlines(th), th, np.linspace(0.5, 0.95, 5),
                                    nalpha=4, ntheta0=3).run(refine=False)
    for r in (0, 1):
        for a, b in zip(res[r], single):
            assert np.allclose(a, b, rtol=0, atol=0)


def _worker_run_edge(rank, world, port, q, case):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N = 129
    th = bo.theta_grid(N)
    fl = synthetic_fieldlines(th)
    if case == "empty":                 # one surface, two ranks: rank 1 owns nothing and still takes part in the gather
        svals = [0.7]
    else:                               # rank 1's only surface fails (its geometry producer raises): carried through the gather
        svals = [0.6, 0.8]

        def fl_bad(s, alphas, fl=fl):
            if s > 0.75:
                if case == "fail_other":          # not an IbsError: a numpy / framework error on one rank
                    raise ValueError("injected failure on the surface s = %g" % s)
                raise ibs_amd.IbsError("injected failure on the surface s = %g" % s)
            return fl(s, alphas)
        fl = fl_bad
    scan = ibs_amd.BallooningScan(OracleContext(), fl, th, svals, nalpha=4, ntheta0=3, rank=rank, world=world, dist=dist)
    try:
        out = ("ok", [o.tolist() for o in scan.run(refine=False)])
    except (ibs_amd.IbsError, ValueError) as e:
        out = ("raised", str(e))
    q.put((rank, out))
    dist.barrier()                      # (both ranks are still in step: nobody was left inside the gather)
    dist.destroy_process_group()


@pytest.mark.parametrize("case", ["empty", "fail", "fail_other"])
def test_two_rank_run_with_empty_shard_and_with_a_failing_rank(case):
    """BallooningScan.run(): a rank without surfaces takes part in the one gather; a rank whose shard fails -- with an
    IbsError or with any other exception -- sends NaN rows and EVERY rank raises after the collective
    (no rank may leave the others waiting in it)."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_run_edge, args=(r, 2, port, q, case)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    if case == "empty":
        N = 129
        th = bo.theta_grid(N)
        single = ibs_amd.BallooningScan(OracleContext(), synthetic_fieldlines(th), th, [0.7], nalpha=4, ntheta0=3).run(refine=False)
        for r in (0, 1):
            assert res[r][0] == "ok"
            for a, b in zip(res[r][1], single):
                assert np.array_equal(a, b)
    else:
        assert res[0][0] == "raised" and res[1][0] == "raised"
        assert "injected failure" in res[1][1] and "not finite" in res[0][1]


def test_history_files_follow_reference_append_semantics(tmp_path):
    # arr_create2.py creates 1-element placeholders; ball_scan.py:369-379 replaces them, then stacks rows
    for name in ("ball_gam", "ball_theta0", "ball_alpha"):
        np.save(tmp_path / ("%s3.npy" % name), np.zeros((1,)))
    g = np.array([1.0, 2.0, 3.0])
    out = ibs_amd.append_history(str(tmp_path), 3, 0, g, g + 10, g + 20)
    assert out["ball_gam"].shape == (3,) and np.array_equal(out["ball_gam"], g)
    out = ibs_amd.append_history(str(tmp_path), 3, 1, 2 * g, g, g)
    assert out["ball_gam"].shape == (2, 3) and np.array_equal(np.load(tmp_path / "ball_gam3.npy")[1], 2 * g)
    out = ibs_amd.append_history(str(tmp_path), 3, 2, 3 * g, g, g)
    assert out["ball_theta0"].shape == (3, 3)


def test_objective_and_dof_gradient_follow_reference_formulas():
    rng = np.random.default_rng(0)
    ndof, nsurf = 6, 5
    gam = rng.uniform(-6e-4, 4e-4, size=(ndof + 1, nsurf))
    f_other = rng.uniform(0.5, 1.0, size=ndof + 1)
    f0 = ibs_amd.ballooning_objective(f_other, gam, gamma_thresh=-2e-4, prefac=50.0)
    for i in range(ndof + 1):      # sims_runner_NCSX.py:254-257 written out
        assert abs(f0[i] - (f_other[i] + 50.0 * sum(max(g + 2e-4, 0.0) for g in gam[i]))) < 1e-15
    x0 = rng.uniform(-0.2, 0.2, size=ndof)
    isabs = np.array([0, 1, 0, 0, 1, 0, 0])
    steps = ibs_amd.dof_steps(x0, isabs)
    assert steps[1] == 1e-3 and abs(steps[2] - 2e-3 * x0[1]) < 1e-18
    df = ibs_amd.dof_fd_gradient(f0, steps)
    assert df.shape == (ndof,)
    assert abs(df[2] - (f0[3] - f0[0]) / steps[3] * 0.5 / np.sqrt(f0[0])) < 1e-12


def _worker_dof(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    own = ibs_amd.shard_dofs(7, rank, world)
    full = ibs_amd.allreduce_dof_vector([10.0 + k for k in own], own, 7, world, dist)
    q.put((rank, full.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_dof_allreduce():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_dof, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0] == res[1] == [10.0 + k for k in range(7)]


def test_surface_tables_weight_matrices_match_reference_splines():
    """host logic of row F1: SurfaceTables.from_wout applies the radial not-a-knot splines as cached weight matrices;
    its tables against the values the reference's own splines produced (G8), and the theta grid helper (A0)"""
    import ibs_amd
    G = os.path.join(os.path.dirname(__file__), "golden")
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    ref = np.load(os.path.join(G, "G8_surface_tables.npz"))
    tab = ibs_amd.SurfaceTables.from_wout(wout, ref["s"])
    from ibs_amd.geometry import NAMES_MN, NAMES_NYQ
    for q, k in enumerate(NAMES_MN):
        assert np.abs(tab.tab_mn[:, q] - ref[k]).max() <= 1e-12 * max(1.0, np.abs(ref[k]).max()), k
    for q, k in enumerate(NAMES_NYQ):
        assert np.abs(tab.tab_nyq[:, q] - ref[k]).max() <= 1e-12 * max(1.0, np.abs(ref[k]).max()), k
    for col, k in ((1, "iota"), (2, "d_iota_d_s"), (3, "d_pressure_d_s")):
        assert np.abs(tab.scal[:, col] - ref[k]).max() <= 1e-12 * max(1.0, np.abs(ref[k]).max()), k
    again = ibs_amd.SurfaceTables.from_wout(wout, ref["s"])            # cached weights: identical tables
    assert np.array_equal(again.tab_mn, tab.tab_mn) and np.array_equal(again.tab_nyq, tab.tab_nyq)
    th = ibs_amd.theta_grid(969)
    assert len(th) == 969 and th[0] == -4 * np.pi and th[-1] == 4 * np.pi and abs((th[1] - th[0]) - 8 * np.pi / 968) < 1e-15


def test_history_files_with_the_reference_0d_placeholder(tmp_path):
    # arr_create2.py:87-97 creates np.empty([]) (0-d) placeholders; ball_scan.py:369-375 replaces them at iteration 0
    ibs_amd.create_history_placeholders(str(tmp_path), 2)
    assert np.load(tmp_path / "ball_gam2.npy").shape == ()
    g = np.array([1.0, 2.0, 3.0, 4.0, 5.0])
    out = ibs_amd.append_history(str(tmp_path), 2, 0, g, g + 10, g + 20)
    for k, off in (("ball_gam", 0), ("ball_theta0", 10), ("ball_alpha", 20)):
        assert out[k].shape == (5,) and np.array_equal(np.load(tmp_path / ("%s2.npy" % k)), g + off)
    out = ibs_amd.append_history(str(tmp_path), 2, 1, 2 * g, g, g)
    assert out["ball_gam"].shape == (2, 5)
    # what sims_runner_NCSX.py:198 reads: the last row
    assert np.array_equal(np.load(tmp_path / "ball_gam2.npy")[-1], 2 * g)


def test_params_dict_reader_and_grid_rule(tmp_path):
    import pickle
    # the dict create_dict.py:143-160 writes for eqbm_option = 1 (NCSX): 72 boundary DOFs (create_dict.py:29-34, 47-55)
    d = dict(maxf=100, eqbm_option=1, pol_idxs=np.array([0, 1, 2, 3, 4, 5, 6]), tor_idxs=np.array([4, 3, 3, 2, 2, 2, 1]),
             iotaidxs=np.array([]), isphifree=0, totalndofs=72, nsurfs=5, abs_step=1.0e-3, rel_step=2.0e-3,
             username="u", nprocspernode=96, nodesperball=8, totalnexecball=73, njobsball=2, nodespersimsopt=1)
    assert set(d) == set(ibs_amd.PARAMS_KEYS)
    with open(tmp_path / "params_dict.pkl", "wb") as f:
        pickle.dump(d, f)
    cfg = ibs_amd.load_params_dict(str(tmp_path / "params_dict.pkl"))
    assert cfg.n_equilibria == 73 and cfg.nsurfs == 5 and (cfg.nalpha, cfg.ntheta0) == (24, 15)
    assert np.array_equal(cfg.rho_arr, np.linspace(0.5, 0.95, 5))                       # ball_scan.py:197
    assert (cfg.gamma_thresh, cfg.prefac) == (-2.0e-4, 50.0)                            # sims_runner_NCSX.py:56-57
    assert cfg.dof_step(5e-3) == 1e-3 and abs(cfg.dof_step(0.5) - 1e-3) < 1e-18        # ball_scan.py:129-139
    d3d = dict(d, eqbm_option=0, pol_idxs=np.array([1, 2, 3]), tor_idxs=np.array([1, 3, 5]), totalndofs=6)
    assert ibs_amd.ScanConfig(d3d).n_equilibria == 7 and ibs_amd.ScanConfig(d3d).prefac == 2.0
    with pytest.raises(ibs_amd.IbsError):
        ibs_amd.ScanConfig(dict(d, totalndofs=70))
    # ball_scan.py:201-208: D3D (mpol 80, ntor 0) -> 641 points, NCSX / HBERG (mpol = ntor = 11) -> 969
    assert len(ibs_amd.theta_grid_for(80, 0)) == 641 and len(ibs_amd.theta_grid_for(11, 11)) == 969
    th = ibs_amd.theta_grid_for(11, 11)
    assert th[0] == -4 * np.pi and th[-1] == 4 * np.pi and np.array_equal(th, ibs_amd.theta_grid(969))


def _worker_bench(rank, world, port, q, n_surf=7):
    """bench.py's own sharded pass (sharded_surface_pass + gather_rows_tensor) under gloo with CPU tensors"""
    import torch
    import torch.distributed as dist
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local_rows(own):     # what C2Sharded.local_rows returns on a GPU: (gam_max, alpha*, theta0*) per own surface
        return torch.tensor([[100.0 + s, 0.5 * s, 0.25 * s] for s in own], dtype=torch.float64).reshape(len(own), 3)

    full = bench.sharded_surface_pass(local_rows, n_surf, rank, world, dist)
    q.put((rank, full.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_surf", [(2, 7), (3, 7), (3, 64), (8, 64), (8, 7)])
def test_bench_sharded_pass_gloo(world, n_surf):
    """C2Sharded-shaped rows with uneven shards (64 surfaces over 3 ranks) and with empty ones (7 surfaces over 8 ranks)"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_bench, args=(r, world, port, q, n_surf)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = [[100.0 + s, 0.5 * s, 0.25 * s] for s in range(n_surf)]
    for r in range(world):
        assert res[r] == want


def test_bench_refuses_mismatched_world_and_spawns_before_gpu():
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not match --gpus" in r.stderr
    # without a launcher --gpus 2 starts two rank processes (which fail here: no GPU), and the parent reports it
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


class _StubLib:
    """the four entry points Context.comm_init touches, with a failure injected on one rank"""
    def __init__(self, fail_load):
        self.fail_load = fail_load
        self.inits = 0
    def ibs_comm_load(self, path):
        return -5 if self.fail_load else 0
    def ibs_comm_unique_id(self, buf):
        buf.raw = bytes(range(128))
        return 0
    def ibs_comm_init(self, h, ident, rank, world):
        self.inits += 1
        return 0


def _worker_comm_init(rank, world, port, q, failing_rank):
    import types
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    me = types.SimpleNamespace(_lib=_StubLib(rank == failing_rank), _h=None)
    try:
        ibs_amd.Context.comm_init(me, dist, rank, world)
        out = "ok"
    except ibs_amd.IbsError:
        out = "refused"
    q.put((rank, (out, me._lib.inits)))
    dist.barrier()                      # (both ranks are still in step: nobody was left inside a collective)
    dist.destroy_process_group()


@pytest.mark.parametrize("failing_rank", [-1, 0, 1])
def test_two_rank_comm_init_agrees_before_the_collective_init(failing_rank):
    """Context.comm_init: whatever fails on ONE rank before ncclCommInitRank (librccl missing, no unique id), every
    rank takes the same collectives of the bootstrap backend and every rank refuses -- none enters the communicator's
    collective initialisation alone.  (The library calls are stubbed: the agreement logic is what runs here.)"""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_comm_init, args=(r, 2, port, q, failing_rank)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    want = ("ok", 1) if failing_rank < 0 else ("refused", 0)
    assert res[0] == want and res[1] == want


def test_bench_watchdog_prints_the_headline_once_and_leaves_with_its_own_status():
    """bench.py's N > 1 phases run under a watchdog: a phase that overruns must end with rank 0 printing the JSON line it
    has (the headline is measured before any such phase) and the watchdog's own NON-ZERO status (a hung run must not look like
    a healthy one) -- and after the regular line has been printed, a late overrun must not print a second one."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent('''
        import sys, time
        sys.path.insert(0, %r)
        import bench
        out = {"metric": "m", "value": 1.0, "config": {"a": 1}}
        dog = bench.Watchdog(int(sys.argv[1]), lambda: out)
        if sys.argv[2] == "printed":
            dog.line_printed()
        dog.arm("test phase", 1)
        time.sleep(20)
        print("not reached")
    ''' % root)
    for rank, state, want in ((0, "pending", 1), (0, "printed", 0), (1, "pending", 0)):
        r = subprocess.run([sys.executable, "-c", code, str(rank), state], capture_output=True, text=True, timeout=60)
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert r.returncode == 5 and "not reached" not in r.stdout and len(lines) == want, (rank, state, r.stdout, r.stderr)
        if want:
            d = json.loads(lines[0])
            assert d["value"] == 1.0 and "test phase" in d["aborted"]
        assert "overran" in r.stderr


def test_surface_tables_of_many_equilibria_native_host_routine():
    """SurfaceTables.from_wouts (ibs_surface_tables_f64: the radial spline step of ALL equilibria of an optimizer step,
    threaded, on the host): (a) against the values the reference's own splines produced (G8, 1e-12); (b) against
    concat(from_wout) on the 73 emulated equilibria of configs[3] (surface index = i_eq * n_s + i_s); (c) the same bits
    whatever the thread count."""
    import ibs_amd
    from ibs_amd.geometry import NAMES_MN, NAMES_NYQ
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import emulated_equilibria
    G = os.path.join(os.path.dirname(__file__), "golden")
    wout = dict(np.load(os.path.join(G, "G8_wout_ncsx_op.npz")))
    ref = np.load(os.path.join(G, "G8_surface_tables.npz"))
    tab = ibs_amd.SurfaceTables.from_wouts([wout], ref["s"])
    for q, k in enumerate(NAMES_MN):
        assert np.abs(tab.tab_mn[:, q] - ref[k]).max() <= 1e-12 * max(1.0, np.abs(ref[k]).max()), k
    for q, k in enumerate(NAMES_NYQ):
        assert np.abs(tab.tab_nyq[:, q] - ref[k]).max() <= 1e-12 * max(1.0, np.abs(ref[k]).max()), k
    for col, k in ((1, "iota"), (2, "d_iota_d_s"), (3, "d_pressure_d_s")):
        assert np.abs(tab.scal[:, col] - ref[k]).max() <= 1e-12 * max(1.0, np.abs(ref[k]).max()), k
    wouts, steps, x0 = emulated_equilibria(wout)
    sv = np.linspace(0.5, 0.95, 5)
    one = ibs_amd.SurfaceTables.concat([ibs_amd.SurfaceTables.from_wout(w, sv) for w in wouts[:9]])
    many = ibs_amd.SurfaceTables.from_wouts(wouts[:9], sv)
    for k in ("s", "tab_mn", "tab_nyq", "scal"):
        a, b = getattr(one, k), getattr(many, k)
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-13 * np.abs(a).max(), k
    assert np.array_equal(one.rows_mn, many.rows_mn) and one.dn_nyq == many.dn_nyq
    for nt in (1, 3):
        again = ibs_amd.SurfaceTables.from_wouts(wouts[:9], sv, n_threads=nt)
        assert np.array_equal(again.tab_mn, many.tab_mn) and np.array_equal(again.tab_nyq, many.tab_nyq)
    bad = dict(wouts[1]); bad["xm"] = wouts[1]["xm"][::-1].copy()
    with pytest.raises(ValueError):
        ibs_amd.SurfaceTables.from_wouts([wouts[0], bad], sv)


class _FakeScan:
    """stands in for BallooningScan.device_rows in the CPU test of AdjointStep: rows are a function of the tables alone"""

    def __init__(self, tables, fail):
        self.tables, self.fail = tables, fail

    def device_rows(self, refine=True, phases=None, chunks=None, fill=None):
        import torch
        if self.fail:
            raise ValueError("injected failure")
        t = self.tables
        gam = 1e-3 * (t.tab_mn[:, 0, :].sum(axis=1) - 1.6) + 2e-4 * t.s            # depends on the (perturbed) rmnc row sums
        rows = np.stack([0.5 * t.s, 1.0 + t.s, gam], axis=1)
        return torch.from_numpy(rows), torch.zeros((), dtype=torch.float64)


def _worker_adjoint(rank, world, port, q, fail_rank):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import emulated_equilibria
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    wout = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "G8_wout_ncsx_op.npz")))
    wouts, steps, x0 = emulated_equilibria(wout)
    wouts, steps = wouts[:7], steps[:7]
    step = ibs_amd.AdjointStep(None, bo.theta_grid(129), np.linspace(0.5, 0.95, 5), torch.device("cpu"), rank=rank, world=world,
                               dist=dist if world > 1 else None, n_threads=2)
    step._scan_for = lambda tables, n: _FakeScan(tables, rank == fail_rank)
    try:
        r = step.run(wouts, 0.8 + 0.01 * np.arange(7), steps)
        out = ("ok", {k: np.asarray(v).tolist() for k, v in r.items()})
    except (ibs_amd.IbsError, ValueError) as e:
        out = ("raised", str(e))
    q.put((rank, out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_two_rank_gloo_adjoint_step_equals_single_rank(fail_rank):
    """AdjointStep: the equilibria of an optimizer step dealt round-robin over two ranks (the reference's one-srun-per-DOF
    level, ball_submit.py:64-95), ONE gather of the rows, objective and forward-difference gradient
    (sims_runner_NCSX.py:249-261) identical on both ranks and equal to the one-rank result; a failing rank is raised
    everywhere after the collective."""
    import queue
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_adjoint, args=(r, 2, port, q, fail_rank)) for r in range(2)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    if fail_rank >= 0:
        assert res[0][0] == "raised" and res[1][0] == "raised" and "injected failure" in res[fail_rank][1]
        return
    q1 = queue.Queue()
    _worker_adjoint(0, 1, 0, q1, -1)
    single = q1.get()[1]
    assert res[0][0] == "ok" and res[1][0] == "ok" and single[0] == "ok"
    for k in ("gam", "theta0", "alpha", "f0", "dfobj", "fobj"):
        assert np.array_equal(np.asarray(res[0][1][k]), np.asarray(res[1][1][k])), k
        assert np.array_equal(np.asarray(res[0][1][k]), np.asarray(single[1][k])), k
    f0 = np.asarray(single[1]["f0"]); gam = np.asarray(single[1]["gam"])
    assert gam.shape == (7, 5) and abs(f0[2] - (0.82 + 50.0 * np.maximum(gam[2] + 2e-4, 0).sum())) < 1e-14


def test_adjoint_step_coarse_runs_cover_every_equilibrium_once():
    """chunk_cuts: whatever the number of runs and the growth factor, the runs are non-empty, in order, and cover 0 .. n exactly;
    growth 1 gives the equal cuts, growth > 1 a first run no longer than any later one."""
    from ibs_amd.objective import chunk_cuts
    for n in (1, 2, 5, 36, 37, 73):
        for nch in (1, 2, 4, 6, 9, 100):
            for gr in (1.0, 1.25, 1.5, 2.0, 0.8):
                c = chunk_cuts(n, nch, gr)
                assert c[0] == 0 and c[-1] == n and all(b > a for a, b in zip(c, c[1:])) and len(c) - 1 <= min(nch, n), (n, nch, gr, c)
                if gr > 1.0 and len(c) > 2:
                    sizes = np.diff(c)
                    assert sizes[0] <= sizes[1:].min() + 1, (n, nch, gr, c)
    assert chunk_cuts(73, 4, 1.0) == [0, 18, 36, 55, 73]
