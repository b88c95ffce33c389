#!/usr/bin/env python3
"""Benchmark of the ballooning hot path on MI355X (contract: task statement, section 4).

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N ...` the ranks come from the
environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); run directly (`python bench.py --gpus N`) the parent spawns the
N rank processes itself BEFORE anything touches the GPU and exits with their status.  WORLD_SIZE must equal --gpus.

One step = one pass of the hot path over one batch: the D3D-shape configuration of
BASELINE.json (configs[1]): 16 flux surfaces x 8 alpha x 8 theta0 = 1,024 field-line eigen-solves
on N_zeta=512 (513-point) grids, FP64, geometry resident in HBM.  A step is the geometry-fed scan
kernel, whose epilogue also reduces every completed surface to its first maximum (+ for N>1 one RCCL all-gather of
the per-surface maxima).
Weak scaling: every rank processes its own 16-surface batch.  With N > 1 the line also carries `ncsx_c2_sharded`:
BASELINE.json configs[2] (64 surfaces x 32 alpha x 16 theta0, N_zeta = 1024) with the surfaces sharded round-robin over
the ranks (geometry -> scan -> argmax on each rank's own surfaces, ONE all-gather of the per-surface rows), checked
bitwise against the table rank 0 computes alone.

Prints ONE JSON line on rank 0 (metric/unit from BASELINE.json) with `roofline` (dominant kernel
k_gamma_scan, HIP-event timed) and `cpu_baseline` (C oracle on the host cores, bounded sample);
the extra `stress` object is config 5 (raw (g,c,f) systems) where the batch is large enough for a
roofline figure to mean something.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PMC_FILE = os.path.join(ROOT, "profiles", "pmc_current.json")
ISSUE_PEAK = 256 * 4 * 2.4e9 / 4      # wave64 VALU instructions per second: 256 CUs x 4 SIMDs, one per 4 clocks at 2.4 GHz


def _pmc_file():
    if not hasattr(_pmc_file, "cache"):
        _pmc_file.cache = json.load(open(PMC_FILE)) if os.path.exists(PMC_FILE) else None
    return _pmc_file.cache


def pmc_set_name():
    """name of the profile set profiles/pmc_current.json was summarised from (its "_set" entry), or 'unnamed'"""
    d = _pmc_file() or {}
    return d["_set"] if isinstance(d.get("_set"), str) else "unnamed"


CSRC = os.path.join(ROOT, "ideal-ballooning-solver_amd", "csrc")
# which sources a kernel's instruction stream comes from: the geometry kernels live in one translation unit, every solver /
# scan / refinement kernel is built from the wave + group solver headers (ibs_api.hip only CHOOSES kernels, and an entry is
# looked up by the exact kernel name and launch size the library reports, so a changed choice cannot quote wrong counters)
# (the Makefile carries the compiler flags and the -D knobs that change the code objects; ibs_api.hip holds a few kernels of its own)
SRC_GROUPS = {"geometry": ("ibs_geometry.hip", "ibs_launch.hpp", "Makefile"),
              "solver": ("ibs_kernels.hip", "ibs_kernels_group.hip", "ibs_long.hip", "ibs_wave.hpp", "ibs_group.hpp", "ibs_refine.hpp",
                         "ibs_lbfgsb2.hpp", "ibs_launch.hpp", "ibs_api.hip", "Makefile")}


def kernel_group(kernel):
    # (every kernel of ibs_geometry.hip: the row kernels, the one-sincos-per-mode fallback, the dPdrho helper)
    return "geometry" if any(t in kernel for t in ("k_geo_", "k_fieldline_geometry", "k_line_dPdrho")) else "solver"


def src_sha(csrc=None):
    """{"geometry": sha, "solver": sha}: sha256 (16 hex digits) over the names and bytes of the sources each kernel group is
    compiled from.  tools/pmc_summary.py stores it as "_src_sha" in the PMC set; pmc_entry() refuses an entry whose group's
    sources have changed since ("stale": a kernel edited after the last PMC pass must not quote the old counters)."""
    import hashlib
    csrc = csrc or CSRC
    out = {}
    for grp, files in SRC_GROUPS.items():
        hh = hashlib.sha256()
        for f in files:
            hh.update(f.encode() + b"\0")
            try:
                hh.update(open(os.path.join(csrc, f), "rb").read())
            except OSError:
                hh.update(b"<absent>")
        out[grp] = hh.hexdigest()[:16]
    return out


def pmc_entry(kernel, waves, work_class=0):
    """The committed rocprofv3 --pmc figures of ONE leg (tools/profile_run.py -> tools/pmc_summary.py ->
    profiles/pmc_current.json): the entry whose kernel name is EXACTLY `kernel` ("ibs::k_gamma_scan<double, 8>"), whose
    launches held `waves` waves (SQ_WAVES, i.e. the leg's own batch size) and which is the `work_class`-th class of equal work
    among the launches of that kernel at that size, in order of first appearance (the PMC passes run the bench's own legs in the
    bench's order: configs[4] runs the smooth family before the rough one through the same kernels).  Returns (entry, None) or
    (None, reason): a leg never quotes another kernel's, another batch size's or another data set's counters."""
    d = _pmc_file()
    if d is None:
        return None, "profiles/pmc_current.json is absent"
    grp = kernel_group(kernel)
    if not hasattr(src_sha, "tree"):
        src_sha.tree = src_sha()
    have = (d.get("_src_sha") or {}).get(grp) if isinstance(d.get("_src_sha"), dict) else None
    if have != src_sha.tree[grp]:
        return None, "stale: the %s sources have changed since the PMC set %s was taken (set %s, tree %s)" % (
            grp, pmc_set_name(), have, src_sha.tree[grp])
    cands = [v for k, v in d.items() if isinstance(v, dict) and v.get("kernel", k.split(" @")[0]) == kernel]
    if not cands:
        return None, "no PMC entry for the kernel '%s'" % kernel
    fit = sorted((v for v in cands if "SQ_WAVES" in v and abs(v["SQ_WAVES"]["mean"] - waves) < 0.5), key=lambda v: v.get("work_class", 0))
    if len(fit) <= work_class:
        return None, "%d PMC entries of '%s' hold %d waves per launch, class %d asked for (waves found: %s)" % (
            len(fit), kernel, waves, work_class, sorted(set(int(v["SQ_WAVES"]["mean"]) for v in cands if "SQ_WAVES" in v)))
    e = fit[work_class]
    if e.get("valu_spread", 1.0) > 1.05:
        return None, "the PMC entry of '%s' mixes launches of different work (instruction counts spread %.2fx)" % (kernel, e["valu_spread"])
    return e, None


def pmc_fields(kernel, waves, ms, alg_bytes=None, work_class=0):
    """roofline fields every leg with a PMC entry reports: `traffic` (HBM bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE in
    KiB, the gfx950 correction of MI355X_MICROARCH.md), its ratio to the algorithmic bytes, VALU instructions per wave and
    the bound that binds: `valu_issue` = wave64 VALU instructions issued per second (count from the PMC pass, time from
    THIS run) against one instruction per SIMD per 4 clocks."""
    e, why = pmc_entry(kernel, waves, work_class)
    src = "profiles/pmc_current.json (committed rocprofv3 --pmc passes of this leg at this size, set %s; replayed, not " \
          "measured by this run)" % pmc_set_name()
    if e is None:
        return dict(traffic=None, counters_error=why, counters_kernel=kernel, counters_waves_per_launch=waves)
    unmatched = (_pmc_file() or {}).get("_unmatched_launches_of_the_byte_passes", 0)
    if unmatched:       # the FETCH / WRITE passes could not be matched launch by launch to the SQ pass's work classes:
        e = dict(e)     # byte counts may belong to another data set -> no traffic figure (the instruction counts stand)
        e.pop("hbm_bytes_per_launch", None)
    out = dict(traffic=e.get("hbm_bytes_per_launch"), counters_kernel=kernel, counters_waves_per_launch=waves,
               counters_source=src, valu_insts_per_wave=e.get("valu_insts_per_wave"),
               valu_busy_frac=e.get("valu_busy_frac_of_wave_lifetime"))
    if unmatched:
        out["counters_error"] = "unmatched: %d launches of the byte passes could not be attributed; traffic withheld" % unmatched
    if alg_bytes and out["traffic"]:
        out["traffic_over_algorithmic"] = out["traffic"] / alg_bytes
    if "SQ_INSTS_VALU" in e and ms:
        ach = e["SQ_INSTS_VALU"]["mean"] / (ms * 1e-3)
        out["valu_issue"] = dict(achieved=ach, peak=ISSUE_PEAK, unit="wave-instructions/s", frac=ach / ISSUE_PEAK)
    return out


def hbm_roofline(alg_bytes, ms, bound, kernel, waves, work_class=0, **extra):
    """the `roofline` object of a leg: HBM figures on ALGORITHMIC bytes (`achieved`, `peak`, `frac` = `hbm_frac`, the
    task's definition) whatever binds, `bound` = what does bind, and the PMC fields of the leg's kernel"""
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    out = dict(bound=bound, achieved=gbs, peak=HBM_PEAK_GBS, unit="GB/s", frac=gbs / HBM_PEAK_GBS, hbm_frac=gbs / HBM_PEAK_GBS,
               kernel=kernel.replace("ibs::", ""), algorithmic_bytes_per_launch=alg_bytes)
    out.update(pmc_fields(kernel, waves, ms, alg_bytes, work_class))
    out.update(extra)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# The line the driver keeps.  The driver's record holds an ~8 KB tail of stdout; the round-4 line was 40.5 KB and was cut
# (BENCH_r04.json: parsed = null).  stdout therefore carries ONE line of at most LINE_LIMIT bytes -- the contract's keys,
# `roofline`, `cpu_baseline` and a few numbers per extra leg -- and everything else goes to bench_detail.json (next to
# bench.py, or $IBS_BENCH_DETAIL) and, as one line, to stderr.  tests/test_bench_cpu.py holds the length under the limit.
LINE_LIMIT = 6000
DETAIL_FILE = os.environ.get("IBS_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))


def _sig(x, n=5):
    """floats to n significant digits, through lists and dicts (ints, bools, strings, None untouched)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float("%.*g" % (n, x)) if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return _sig(float(x), n)


def _counters_tag(r):
    """where a roofline's counters come from, in a few bytes: "pmc:<set>" (replayed from profiles/pmc_current.json: DESIGN.md 5)
    or "none:<reason class>" """
    if r.get("counters_error"):
        e = r["counters_error"]
        if e.startswith("unmatched"):
            return "pmc:%s(traffic withheld)" % pmc_set_name()
        return "none:" + ("stale" if e.startswith("stale") else "absent")
    return "pmc:" + pmc_set_name()


def compact_roofline(r, head=False):
    """a leg's roofline object without the prose: the contract's six keys + the kernel, the VALU-issue fraction (the bound that
    binds: DESIGN.md 4), instructions per wave and the counters' source tag"""
    if not isinstance(r, dict):
        return r
    vi = r.get("valu_issue") or {}
    out = dict(bound=r.get("bound"), achieved=r.get("achieved"), peak=r.get("peak"), unit=r.get("unit"), frac=r.get("frac"),
               traffic=r.get("traffic"), kernel=r.get("kernel"), traffic_over_algorithmic=r.get("traffic_over_algorithmic"),
               valu_issue_frac=vi.get("frac"), valu_insts_per_wave=r.get("valu_insts_per_wave"), counters=_counters_tag(r))
    if head:
        out.update(algorithmic_bytes_per_launch=r.get("algorithmic_bytes_per_launch"), kernel_ms=r.get("kernel_ms"),
                   kernel_ms_rocprof=r.get("kernel_ms_rocprof"), kernel_ms_rocprof_source=r.get("kernel_ms_rocprof_source"),
                   waves_per_launch=r.get("counters_waves_per_launch"))
    return out


def _leg4(leg, rate_key):
    """[rate, hbm_frac, valu_issue_frac, traffic_over_algorithmic] of a throughput leg"""
    r = leg.get("roofline") or {}
    return [leg.get(rate_key), r.get("hbm_frac"), (r.get("valu_issue") or {}).get("frac"), r.get("traffic_over_algorithmic")]


C5_COLS = ["n_zeta", "family", "mode", "solves_per_s", "hbm_frac", "valu_issue_frac", "traffic_over_algorithmic", "kernel"]


def compact_c5(m):
    """configs[4]: one array per (N_zeta, family, mode) row (C5_COLS); kernel names once, rows point into the list"""
    kernels, rows = [], []
    for r in m.get("rows", []):
        if "mode" not in r:
            rows.append([r.get("n_zeta"), r.get("family", "?")[0], "skipped"])
            continue
        k = (r.get("roofline") or {}).get("kernel")
        if k not in kernels:
            kernels.append(k)
        rows.append([r["n_zeta"], r["family"][0], r["mode"].replace("f32_", "f32")] + _leg4(r, "solves_per_s")[0:4] + [kernels.index(k)])
    return dict(systems_per_row=next((r.get("systems") for r in m.get("rows", []) if "systems" in r), None), cols=C5_COLS,
                rows=rows, kernels=kernels, f32_gam_over_f64_min=min([v for v in (m.get("f32_gam_over_f64") or {}).values() if v] or [None]),
                f32_results_outside_tolerance=m.get("f32_results_outside_tolerance"),
                # FP64 rows: worst |lam - oracle| / ||A|| of the sampled systems in units of N eps (stated tolerance: 4), systems re-closed
                f64_max_dlam_over_N_eps_normA=max([r["max_abs_dlam_over_normA"] / ((r["n_zeta"] + 1) * 2.220446049250313e-16)
                                                   for r in m.get("rows", []) if "max_abs_dlam_over_normA" in r] or [None]),
                f64_rows_outside_4N_eps=m.get("f64_rows_outside_4N_eps"),
                f64_reclosed=sum(r.get("reclosed_in_division_form", 0) for r in m.get("rows", [])),
                flagged=sum(r.get("flagged", 0) for r in m.get("rows", [])), seconds=m.get("seconds"))


def _strip_prose(d, limit=90):
    """an extra leg without its long strings (workload descriptions, notes, *_how): for legs with no schema of their own below"""
    if not isinstance(d, dict):
        return d
    return {k: (_strip_prose(v, limit) if isinstance(v, dict) else v) for k, v in d.items()
            if not (isinstance(v, str) and len(v) > limit) and not k.endswith("_how") and k != "note"}


# what may be dropped, first to last, if a line is still over the limit (it never is with today's legs: the test pins 6,000)
_DROP_ORDER = ("batch_scaling", "warm_rescan", "gather_modes", "dropin_call_us", "long_grid", "sturm_sweep", "scan_large", "stress_rough",
               "reference_batch", "ncsx_c3", "c5_matrix", "ncsx_c2_sharded_native", "cpu_reference_cost")


def compact_line(out, limit=LINE_LIMIT):
    """the ONE stdout line: `out` (everything the run measured) reduced to the contract's keys + 3-8 numbers per leg, floats at
    5 significant digits except value / ms_per_step; at most `limit` bytes whatever the legs hold"""
    o = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                             "vs_baseline", "dtype", "data") if k in out}
    cfg = dict(out.get("config", {}))
    if len(cfg.get("workload", "")) > 340:
        cfg["workload"] = cfg["workload"][:337] + "..."
    o["config"] = _sig(cfg)
    o["solves_per_s_kernel"] = _sig(out.get("solves_per_s_kernel"))
    o["roofline"] = _sig(compact_roofline(out.get("roofline"), head=True))
    for k in ("cpu_baseline", "cpu_reference_cost"):
        if k in out:
            o[k] = _sig({a: b for a, b in out[k].items() if a != "sample" or k == "cpu_baseline"})
            if isinstance(o[k].get("sample"), str) and len(o[k]["sample"]) > 120:
                o[k]["sample"] = o[k]["sample"][:117] + "..."
    for k in ("parity_ok", "max_abs_dgam_vs_oracle"):
        if k in out:
            o[k] = _sig(out[k])
    four = "[rate_per_s, hbm_frac, valu_issue_frac, traffic_over_algorithmic]"
    for k, rate in (("stress", "solves_per_s"), ("stress_rough", "solves_per_s"), ("sturm_sweep", "sweeps_per_s"), ("scan_large", "solves_per_s")):
        if k in out:
            o.setdefault("legs_cols", four)
            o[k] = _sig(_leg4(out[k], rate))
    if "warm_rescan" in out:
        w = out["warm_rescan"]
        o["warm_rescan"] = _sig(dict(solves_per_s=w.get("solves_per_s"), sweeps_cold=w.get("sweeps_cold"), sweeps_warm=w.get("sweeps_warm")))
    if "batch_scaling" in out:
        b = out["batch_scaling"]
        o["batch_scaling"] = _sig(dict(solves_per_launch=[r["solves_per_launch"] for r in b.get("one_launch", [])],
                                       us_per_launch=[r["us_per_launch"] for r in b.get("one_launch", [])],
                                       two_streams_us=(b.get("two_streams") or {}).get("us_per_launch")))
    for k in ("ncsx_c3", "reference_batch"):
        if k in out:
            L = out[k]
            gr, sr = L.get("geometry_roofline") or {}, L.get("roofline") or {}
            o[k] = _sig(dict(geometry_ms=L.get("geometry_ms"), scan_ms=L.get("scan_ms"), argmax_ms=L.get("argmax_ms"),
                             refine_ms=L.get("refine_ms"), scan_solves_per_s=L.get("scan_solves_per_s"),
                             geometry_points_per_s=L.get("geometry_points_per_s"), nonconverged=L.get("nonconverged"),
                             geometry_kernel=gr.get("kernel"), geometry_valu_issue_frac=(gr.get("valu_issue") or {}).get("frac"),
                             scan_kernel=sr.get("kernel"), scan_valu_issue_frac=(sr.get("valu_issue") or {}).get("frac")))
            o[k] = {a: b for a, b in o[k].items() if b is not None}
    if "dropin_call_us" in out:
        o["dropin_call_us"] = _sig(out["dropin_call_us"].get("us_per_call") if isinstance(out["dropin_call_us"], dict) else out["dropin_call_us"])
    if "c4_adjoint_step" in out:
        c = out["c4_adjoint_step"]
        o["c4_adjoint_step"] = _sig({k: c.get(k) for k in ("total_ms", "total_ms_runs", "phases_ms", "refine_evaluations", "refine_rounds",
                                                            "coarse_solves", "fobj", "max_abs_dgam_vs_oracle", "oracle_pairs", "parity_ok", "error")
                                     if c.get(k) is not None})
    if "c5_matrix" in out:
        o["c5_matrix"] = _sig(compact_c5(out["c5_matrix"]), 4)
    known = set(o) | {"config", "roofline"}
    extra = []
    for k, v in out.items():              # legs without a schema above (the N > 1 legs): numbers and flags, no prose
        if k not in known:
            o[k] = _sig(_strip_prose(v))
            extra.append(k)
    o["detail"] = os.path.basename(DETAIL_FILE) + " (+ one line on stderr)"
    line = json.dumps(o, separators=(",", ":"))
    dropped = []
    while len(line) > limit and extra:    # over the limit: first the largest of the legs that have no schema here ...
        k = max(extra, key=lambda k: len(json.dumps(o[k])))
        if len(json.dumps(o[k])) < 600:
            break
        extra.remove(k)
        del o[k]
        dropped.append(k)
        o["dropped_for_length"] = dropped
        line = json.dumps(o, separators=(",", ":"))
    for k in _DROP_ORDER:                 # ... then the side legs in a fixed order
        if len(line) <= limit:
            break
        if k in o:
            del o[k]
            dropped.append(k)
            o["dropped_for_length"] = dropped
            line = json.dumps(o, separators=(",", ":"))
    keep = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline", "cpu_baseline", "parity_ok", "max_abs_dgam_vs_oracle", "detail", "dropped_for_length"}
    while len(line) > limit and set(o) - keep:      # then whatever is largest among the keys the contract does not name
        k = max(set(o) - keep, key=lambda k: len(json.dumps(o[k])))
        del o[k]
        dropped.append(k)
        o["dropped_for_length"] = dropped
        line = json.dumps(o, separators=(",", ":"))
    if len(line) > limit:                 # last resort: the contract's keys, roofline and cpu_baseline only
        o = {k: o[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                               "vs_baseline", "dtype", "data", "roofline", "cpu_baseline", "parity_ok") if k in o}
        o["config"] = {"workload": cfg.get("workload")}
        o["dropped_for_length"] = "everything but the contract's keys"
        line = json.dumps(o, separators=(",", ":"))
    return line


def emit(out):
    """rank 0's output: the full record to DETAIL_FILE and (one line) to stderr, the compact line to stdout -- last, alone"""
    full = json.dumps(out)
    try:
        with open(DETAIL_FILE, "w") as fh:
            fh.write(full + "\n")
    except OSError as e:
        print("bench.py: could not write %s (%s)" % (DETAIL_FILE, e), file=sys.stderr)
    print("bench.py detail: " + full, file=sys.stderr, flush=True)
    print(compact_line(out), flush=True)


HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
N_SURF, N_ALPHA, N_THETA0, NPTS = 16, 8, 8, 513


def build_workload(rank, device):
    """D3D-shape batch from the NCSX_op-derived golden lines (true D3D geometry needs VMEC: SURVEY H6):
    128 lines = 16 surfaces x 8 alpha, each a golden line with a smooth per-line perturbation."""
    import torch
    g3 = np.load(os.path.join(ROOT, "tests", "golden", "G3_ncsx_lines.npz"))
    geo = g3["geo_513"]                      # (16, 8, 513)
    nl = N_SURF * N_ALPHA
    rng = np.random.default_rng(1000 + rank)
    base = geo[np.arange(nl) % len(geo)].copy()
    eps = rng.uniform(-0.03, 0.03, size=(nl, 2))
    base[:, 4:7, :] *= (1 + eps[:, 0])[:, None, None]      # gds2, gds21, gds22
    base[:, 2:4, :] *= (1 + eps[:, 1])[:, None, None]      # cvdrift, cvdrift0
    base[:, 7, :] *= (1 + eps[:, 1])[:, None]              # gbdrift (keeps cvdrift-gbdrift consistent)
    dP = -0.5 * np.mean((base[:, 2] - base[:, 7]) * base[:, 0] ** 2, axis=1)   # ball_scan.py:262
    theta0 = np.linspace(0.0, 0.5 * np.pi, N_THETA0)                           # ball_scan.py:225
    h = 8 * np.pi / (NPTS - 1)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    geo7 = [t(base[:, k, :]) for k in range(7)]
    return h, geo7, t(dP), t(theta0), base, dP, theta0


def ncsx_boundary_dofs(xm, xn, nfp):
    """the 72 boundary DOFs of the NCSX / HBERG set-up (create_dict.py:29-34, 47-55): RBC(m, n) and ZBS(m, n) for
    m = 0..6 with |n| <= (4, 3, 3, 2, 2, 2, 1)[m] (m = 0: n = 1..tor only) -> list of (table name, row in the wout mode list)"""
    pol = [0, 1, 2, 3, 4, 5, 6]; tor = [4, 3, 3, 2, 2, 2, 1]
    dofs = []
    for name in ("rmnc", "zmns"):
        for m, t in zip(pol, tor):
            for n in (range(1, t + 1) if m == 0 else range(-t, t + 1)):
                row = np.nonzero((xm == m) & (xn == n * nfp))[0]
                assert len(row) == 1, (m, n)
                dofs.append((name, int(row[0])))
    return dofs


def emulated_equilibria(wout0):
    """without VMEC the 72 DOF-perturbed equilibria are emulated on the shipped NCSX_op tables: the boundary
    value x of DOF k is stepped by abs 1e-3 if |x| <= 1e-2 else rel 2e-3 (create_dict.py:67, 70; ball_scan.py:129-139)
    and the change carried inward with an s^2 profile.  Identical arithmetic, not a consistent equilibrium."""
    nfp = int(wout0["nfp"])
    dofs = ncsx_boundary_dofs(np.asarray(wout0["xm"]), np.asarray(wout0["xn"]), nfp)
    assert len(dofs) == 72
    prof = np.linspace(0, 1, wout0["rmnc"].shape[1]) ** 2
    wouts, steps, x0 = [wout0], [1.0], []
    for name, row in dofs:
        w = dict(wout0)
        w[name] = wout0[name].copy()
        x = w[name][row, -1]
        step = 1.0e-3 if abs(x) <= 1.0e-2 else 2.0e-3 * x
        w[name][row, :] += step * prof
        wouts.append(w); steps.append(step); x0.append(x)
    return wouts, np.array(steps), np.array(x0)


def cpu_baseline(h, base, dP, theta0, budget_s=12.0):
    """C oracle (oracle/ibs_oracle.c, 'port') on the same D3D-shape batch, all host cores."""
    from oracle import c_oracle as co
    arrs = [base[:, k, :] for k in range(7)]
    cores = min(len(os.sched_getaffinity(0)), 16)     # the 1-GPU box's CPU share
    co.gamma_scan(h, *[a[:8] for a in arrs], dP[:8], theta0, nthreads=cores)      # warm the pool
    n = 0
    t0 = time.time()
    used = cores
    while time.time() - t0 < budget_s:
        gam, lam, used = co.gamma_scan(h, *arrs, dP, theta0, nthreads=cores)
        n += gam.size
    dt = time.time() - t0
    # "1 core and all cores" -- the same code on ONE thread (OMP_NUM_THREADS=1 is how the reference itself runs,
    # slurm_ball_scan_template.sl:10), a quarter of the batch per pass for about 4 s
    q = max(1, len(dP) // 4)
    n1 = 0
    t1 = time.time()
    while time.time() - t1 < 4.0:
        g1, _, _ = co.gamma_scan(h, *[a[:q] for a in arrs], dP[:q], theta0, nthreads=1)
        n1 += g1.size
    dt1 = time.time() - t1
    return dict(value=n / dt, unit="solves/s", cores=int(used), kind="port", value_1core=n1 / dt1,
                sample="%d passes of the same 1,024-solve D3D-shape batch (%.1f s) on %d threads, then %d solves on ONE thread "
                       "(%.1f s): Sturm bisection + inverse iteration in C/OpenMP" % (n // gam.size, dt, int(used), n1, dt1)), gam


def cpu_reference_cost(h, base, dP, theta0, nsolve=24):
    """the reference's own formulation (dense (N-2)^2 matrix + ARPACK shift-invert) restated in the oracle, 1 core"""
    from oracle import ballooning_oracle as bo
    th = bo.theta_grid(NPTS)
    vg = bo.vguess(th)
    t0 = time.time()
    for k in range(nsolve):
        line = base[k % len(base)]
        cv, gd = bo.fold_theta0(theta0[k % len(theta0)], line[2], line[3], line[4], line[5], line[6])
        bo.gamma_ball_full_dense_arpack(dP[k % len(base)], th, line[0], line[1], cv, gd, vg, 1.0)
    dt = time.time() - t0
    return dict(value=nsolve / dt, unit="solves/s", cores=1, kind="reference-formulation restated (dense LU + ARPACK)",
                sample="%d solves" % nsolve)


def stress(ctx, device, n_sys, family, reps=3):
    """config 5: raw (g, c, f) systems, N_zeta=512, FP64; returns solves/s and the roofline of k_solve_gcf"""
    import torch
    N = NPTS
    th = torch.linspace(-4 * np.pi, 4 * np.pi, N, dtype=torch.float64, device=device)
    gen = torch.Generator(device=device)
    gen.manual_seed(20240 + 512)
    u = lambda lo, hi, shape: lo + (hi - lo) * torch.rand(shape, dtype=torch.float64, device=device, generator=gen)
    if family == "smooth":      # s-alpha coefficients (bishop_ball_s-alpha.py:30-45), f = g
        sh, al, t0 = u(0.1, 2.0, (n_sys, 1)), u(0.0, 1.2, (n_sys, 1)), u(0.0, np.pi / 2, (n_sys, 1))
        lam = sh * (th[None] - t0) - al * (torch.sin(th)[None] - torch.sin(t0))
        g = 1 + lam ** 2
        c = al * (torch.cos(th)[None] + torch.sin(th)[None] * lam)
        f = g.clone()          # its own array: the three rows of a system are three streams of HBM traffic
        del lam
    else:                       # iid per point inside the measured NCSX_op envelopes
        g = torch.exp(u(np.log(0.01), np.log(50.0), (n_sys, N)))
        c = u(-2.5, 3.5, (n_sys, N))
        f = torch.exp(u(np.log(0.2), np.log(3e3), (n_sys, N)))
    h = 8 * np.pi / (N - 1)
    r = ctx.solve_gcf(h, g, c, f, want_info=True)                  # warm-up + correctness word
    torch.cuda.synchronize()
    nbad = int((((r["info"] >> 16) & 3) != 0).sum().item())
    sweeps = float((r["info"] & 0xffff).double().mean().item())
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        ctx.solve_gcf(h, g, c, f)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    bytes_per = (3 * N + 1) * 8
    kern, waves = ctx.last_launch()
    return dict(workload="config 5 raw (g,c,f), %s family, %d systems, N_zeta=512, f64" % (family, n_sys),
                solves_per_s=n_sys / (ms * 1e-3), ms_per_launch=ms, mean_sweeps=sweeps, nonconverged=nbad,
                roofline=hbm_roofline(n_sys * bytes_per, ms, "valu_issue", kern, waves, bytes_per_solve=bytes_per))


def sturm_sweep(ctx, device, n_sys, reps=5):
    """the bandwidth kernel of the path: ONE Sturm-count sweep per system (k_sturm_count), N_zeta=512, f64"""
    import torch
    N = NPTS
    gen = torch.Generator(device=device)
    gen.manual_seed(7)
    g = torch.exp(torch.rand((n_sys, N), dtype=torch.float64, device=device, generator=gen) * 3 - 1)
    c = torch.rand((n_sys, N), dtype=torch.float64, device=device, generator=gen) * 6 - 2.5
    f = torch.exp(torch.rand((n_sys, N), dtype=torch.float64, device=device, generator=gen) * 3)
    sh = torch.zeros(n_sys, dtype=torch.float64, device=device)
    h = 8 * np.pi / (N - 1)
    ctx.sturm_count(h, g, c, f, sh)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        ctx.sturm_count(h, g, c, f, sh)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
    bytes_per = (3 * N + 1) * 8 + 4
    kern, waves = ctx.last_launch()
    # the same counts in DIVISION form (lanes as systems, rows through an LDS transpose: exact for a pencil a few ulp away, any N):
    # what the 10^6-system tests certify with; the prefix-product sweep above is the bandwidth kernel
    cnt_p = ctx.sturm_count(h, g, c, f, sh)
    cnt_d = ctx.sturm_count(h, g, c, f, sh, exact=True)
    kern_d = ctx.last_launch()[0]
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        ctx.sturm_count(h, g, c, f, sh, exact=True)
        b.record()
    torch.cuda.synchronize()
    ms_d = float(np.median([a.elapsed_time(b) for a, b in evs]))
    return dict(workload="one Sturm-count sweep per system, %d random systems, N_zeta=512, f64" % n_sys,
                sweeps_per_s=n_sys / (ms * 1e-3), ms_per_launch=ms,
                roofline=hbm_roofline(n_sys * bytes_per, ms, "hbm", kern, waves, bytes_per_sweep=bytes_per),
                division_form=dict(kernel=kern_d, ms_per_launch=ms_d, gb_per_s=n_sys * bytes_per / (ms_d * 1e6),
                                   frac_of_8_tb_s=n_sys * bytes_per / (ms_d * 1e6) / 8000.0,
                                   counts_equal_the_sweeps=bool(torch.equal(cnt_p, cnt_d))))


def long_grid(ctx, device, N=4097, n_batch=2048, oracle_check=True):
    """grids beyond the register-resident kernels (csrc/ibs_long.hip; the reference's grid rule passes 2,050 points from
    mpol ntor > 256 on): s-alpha systems at N = 4,097 -- one system with its growth rate (latency), a batch (rate), and a sample
    against the C oracle"""
    import torch
    th = np.linspace(-4 * np.pi, 4 * np.pi, N); h = float(th[1] - th[0])
    rng = np.random.default_rng(N)
    sh, al, t0 = rng.uniform(0.1, 2.0, (n_batch, 1)), rng.uniform(0.0, 1.2, (n_batch, 1)), rng.uniform(0.0, np.pi / 2, (n_batch, 1))
    lam = sh * (th[None] - t0) - al * (np.sin(th)[None] - np.sin(t0))
    g = torch.from_numpy(1 + lam ** 2).to(device); c = torch.from_numpy(al * (np.cos(th)[None] + np.sin(th)[None] * lam)).to(device)
    f = g.clone()
    out = dict(workload="s-alpha systems, N = %d, f64: one system / %d systems per call, growth rate wanted" % (N, n_batch))
    for name, sl in (("one", slice(0, 1)), ("batch", slice(0, n_batch))):
        gg, cc, ff = g[sl], c[sl], f[sl]
        r = ctx.solve_gcf(h, gg, cc, ff, want_info=True); torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in evs:
            a.record(); ctx.solve_gcf(h, gg, cc, ff); b.record()
        torch.cuda.synchronize()
        ms = float(min(a.elapsed_time(b) for a, b in evs))
        if name == "one":
            out["one_system_ms"] = ms
        else:
            out["batch_systems_per_s"] = n_batch / (ms * 1e-3); out["batch_ms"] = ms
            out["kernel"] = ctx.last_launch()[0]
            out["passes_mean"] = float((r["info"] & 0xffff).double().mean().item())
            out["flagged"] = int(((r["info"] >> 16) != 0).sum().item())
            if oracle_check:
                from oracle import c_oracle as co
                gam_c, lam_c, _ = co.solve_gcf_batch(h, gg[:8].cpu().numpy(), cc[:8].cpu().numpy(), ff[:8].cpu().numpy())
                out["max_abs_dgam_vs_oracle"] = float(np.abs(r["gam"][:8].cpu().numpy() - gam_c).max())
                out["parity_ok"] = bool(out["max_abs_dgam_vs_oracle"] < 1e-8 and out["flagged"] == 0)
    return out


def warm_rescan(ctx, device, h, geo7, dP_d, th0_d, reps=20):
    """config-4 pattern (FD gradient over boundary DOFs, sims_runner_NCSX.py:151-276): the batch of a
    DOF-perturbed equilibrium (geometry changed by rel 2e-3, create_dict.py:70) re-scanned with the base
    equilibrium's eigenvalues as warm start."""
    import torch
    base = ctx.gamma_scan(h, *geo7, dP_d, th0_d)
    pert = [g.clone() for g in geo7]
    for k in (4, 5, 6):
        pert[k] *= 1.002
    pert[2] *= 0.999
    cold = ctx.gamma_scan(h, *pert, dP_d, th0_d, want_info=True)
    width = float(3 * (cold["lam"] - base["lam"]).abs().max().item())
    warm = ctx.gamma_scan(h, *pert, dP_d, th0_d, want_info=True, lam_guess=base["lam"], guess_width=width)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        ctx.gamma_scan(h, *pert, dP_d, th0_d, lam_guess=base["lam"], guess_width=width)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.median([a.elapsed_time(b) for a, b in evs]))
    n = cold["lam"].numel()
    return dict(workload="re-scan of a DOF-perturbed batch (rel 2e-3) warm-started from the base scan, same shape as the step",
                solves_per_s=n / (ms * 1e-3), ms_per_launch_incl_host=ms,
                sweeps_cold=float((cold["info"] & 0xffff).double().mean().item()),
                sweeps_warm=float((warm["info"] & 0xffff).double().mean().item()),
                max_abs_dgam_warm_vs_cold=float((warm["gam"] - cold["gam"]).abs().max().item()), guess_width=width)


def scan_large(ctx, device, geo7, dP_d, reps=3):
    """the geometry-fed scan far above the chip size: the bench step's 128 lines tiled to 8,192 lines x 16 theta0 =
    131,072 solves at N_zeta = 512 (sub-wave kernels, theta0 chained through the groups)"""
    import torch
    rep = 64
    g7 = [g.repeat(rep, 1) * (1 + 0.0005 * (torch.arange(g.shape[0] * rep, device=device) % 41).double())[:, None]
          if k in (4, 5, 6) else g.repeat(rep, 1) for k, g in enumerate(geo7)]
    dP = dP_d.repeat(rep)
    t0 = torch.linspace(0, np.pi / 2, 16, dtype=torch.float64, device=device)
    h = 8 * np.pi / (NPTS - 1)
    out = ctx.gamma_scan(h, *g7, dP, t0, want_info=True)
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        ctx.gamma_scan(h, *g7, dP, t0)
        b.record()
    torch.cuda.synchronize()
    ms = float(np.min([a.elapsed_time(b) for a, b in evs]))
    n = out["gam"].numel()
    bytes_per = (7 * NPTS * 8 + 8) / 16 + 8
    kern, waves = ctx.last_launch()
    return dict(workload="%d lines x 16 theta0 = %d solves, N_zeta=512, f64" % (g7[0].shape[0], n),
                solves_per_s=n / (ms * 1e-3), ms_per_launch_incl_host=ms,
                mean_sweeps=float((out["info"] & 0xffff).double().mean().item()),
                nonconverged=int(((out["info"] >> 16) != 0).sum().item()),
                roofline=hbm_roofline(n * bytes_per, ms, "valu_issue", kern, waves, bytes_per_solve=bytes_per))


def batch_scaling(ctx, device, h, geo7, dP_d, th0_d):
    """where the latency-bound regime of the headline step ends: the step's batch replicated 1-8x in ONE launch, and two
    step-sized launches in flight on two streams (two contexts: stream and arrival counters are per context)."""
    import torch
    import ibs_amd

    def timed(fn, n):
        for _ in range(max(20, n // 10)):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    rows = []
    for rep in (1, 2, 4, 8):
        plan = ibs_amd.ScanPlan(ctx, h, [x.repeat(rep, 1) for x in geo7], dP_d.repeat(rep), th0_d, N_SURF * rep)
        dt = timed(plan.scan_argmax, 600 // rep + 50)
        n = plan.n_lines * plan.n_t0
        rows.append(dict(solves_per_launch=n, us_per_launch=dt * 1e6, solves_per_s=n / dt))
    ctx2 = ibs_amd.Context(ctx.device)
    plans = [ibs_amd.ScanPlan(c, h, geo7, dP_d, th0_d, N_SURF) for c in (ctx, ctx2)]
    streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)]
    k = [0]

    def two():
        i = k[0] & 1
        k[0] += 1
        with torch.cuda.stream(streams[i]):
            plans[i].scan_argmax()

    dt = timed(two, 2000)
    torch.cuda.synchronize()
    ctx2.close()
    n1 = N_SURF * N_ALPHA * N_THETA0
    return dict(workload="scan + per-surface argmax of the step's batch replicated in one launch; two step-sized launches in "
                         "flight on two streams", one_launch=rows,
                two_streams=dict(solves_per_launch=n1, us_per_launch=dt * 1e6, solves_per_s=n1 / dt),
                note="the headline step keeps one 1,024-solve launch at a time (BASELINE configs[1]); a caller with several "
                     "independent equilibria should batch them into one launch")


def ncsx_pipeline(ctx, device):
    """configs[2] shape on one GPU (64 surfaces x 32 alpha x 16 theta0, N_zeta=1024) from the shipped NCSX equilibrium's
    wout tables: field-line geometry kernel (row F1) -> geometry-fed scan -> per-surface argmax, and the reference's
    own batch (5 surfaces x 24 x 15, N=969) with its refinement (ball_scan.py:305-339) on the device (row F2)."""
    import torch
    import ibs_amd
    wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
    out = {}
    for tag, ns, na, nt0, N, svals in (("ncsx_c3", 64, 32, 16, 1025, np.linspace(0.1, 0.95, 64)),
                                       ("reference_batch", 5, 24, 15, 969, np.linspace(0.5, 0.95, 5))):
        tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
        th = ibs_amd.theta_grid(N)
        alphas = np.linspace(0, np.pi, na)
        t0 = torch.from_numpy(np.linspace(0, np.pi / 2, nt0)).to(device)
        # (line tables and the theta grid resident in HBM, as a driver that scans every optimizer iteration keeps them)
        surf = torch.from_numpy(np.repeat(np.arange(ns), na).astype(np.int32)).to(device)
        al = torch.from_numpy(np.tile(alphas, ns)).to(device)
        th_d = torch.from_numpy(th).to(device)
        best = None
        for rep in range(6):
            e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            e[0].record()
            r = ctx.fieldline_geometry(tabs, surf, al, th_d, device=device)
            k_geo = ctx.last_launch()
            e[1].record()
            sc = ctx.gamma_scan(th[1] - th[0], *[r["geo"][k] for k in range(7)], r["dPdrho"], t0, want_info=True)
            k_scan = ctx.last_launch()
            e[2].record()
            idx, val = ctx.surface_argmax(sc["gam"].reshape(ns, -1))
            e[3].record()
            torch.cuda.synchronize()
            t = [e[k].elapsed_time(e[k + 1]) for k in range(3)]
            if best is None or sum(t) < sum(best):
                best = t
        n = ns * na * nt0
        bytes_per = (7 * N * 8 + 8) / nt0 + 8
        leg = dict(workload="%d surfaces x %d alpha x %d theta0, N=%d, NCSX_op wout tables" % (ns, na, nt0, N),
                   geometry_ms=best[0], geometry_points_per_s=ns * na * N / (best[0] * 1e-3),
                   scan_ms=best[1], scan_solves_per_s=n / (best[1] * 1e-3), argmax_ms=best[2],
                   mean_sweeps=float((sc["info"] & 0xffff).double().mean().item()),
                   nonconverged=int(((sc["info"] >> 16) != 0).sum().item()),
                   roofline=hbm_roofline(n * bytes_per, best[1], "valu_issue", *k_scan, bytes_per_solve=bytes_per),
                   # the geometry kernel writes 8 arrays per grid point (its algorithmic bytes); what binds is FP64 issue
                   geometry_roofline=hbm_roofline(ns * na * N * 64.0, best[0], "valu_issue", *k_geo, bytes_per_point=64))
        if tag == "reference_batch":
            scan = ibs_amd.BallooningScan(ctx, None, th, svals, nalpha=na, ntheta0=nt0, tables=tabs, device=device)
            tab = sc["gam"].reshape(ns, na, nt0).cpu().numpy()
            starts = np.array([ibs_amd.pick_start(t_, scan.alpha_scan, scan.theta0_scan)[:2] for t_ in tab])
            scan.refine_device(starts)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            xo, fo, ne = scan.refine_device(starts)
            leg["refine_ms"] = (time.perf_counter() - t1) * 1e3
            leg["refine_evaluations_per_surface"] = [int(v) for v in ne]
            leg["gam_coarse_max"] = [float(v) for v in tab.reshape(ns, -1).max(axis=1)]
            leg["gam_refined"] = [float(-v) for v in fo]
        out[tag] = leg
    return out


def c5_family(dev, family, n, N, seed):
    """'smooth' = s-alpha coefficients (bishop_ball_s-alpha.py:30-45, f = g), shat ~ U(0.1, 2), alpha ~ U(0, 1.2),
    theta0 ~ U(0, pi/2); 'rough' = iid per point inside the measured NCSX_op envelopes."""
    import torch
    gen = torch.Generator(device=dev); gen.manual_seed(seed)
    u = lambda lo, hi, shape: lo + (hi - lo) * torch.rand(shape, dtype=torch.float64, device=dev, generator=gen)
    h = 8 * np.pi / (N - 1)
    if family == "smooth":
        th = torch.linspace(-4 * np.pi, 4 * np.pi, N, dtype=torch.float64, device=dev)
        sh, al, t0 = u(0.1, 2.0, (n, 1)), u(0.0, 1.2, (n, 1)), u(0.0, np.pi / 2, (n, 1))
        lam = sh * (th[None] - t0) - al * (torch.sin(th)[None] - torch.sin(t0))
        g = 1 + lam ** 2
        c = al * (torch.cos(th)[None] + torch.sin(th)[None] * lam)
        del lam
        return h, g, c, g.clone()      # f = g in value, its own array in HBM (priced at (3 N + 1) w bytes: all three rows move)
    g = torch.exp(u(np.log(0.01), np.log(50.0), (n, N)))
    c = u(-2.5, 3.5, (n, N))
    f = torch.exp(u(np.log(0.2), np.log(3e3), (n, N)))
    return h, g, c, f


def norm_a(h, g, c, f, chunk=65536):
    """the solver's ||A|| bound per system: max_r (|d_r| + e_r + e_{r+1}) / f_r   (utils.py:1584-1592 rows)"""
    import torch
    out = torch.empty(g.shape[0], dtype=torch.float64, device=g.device)
    for a in range(0, g.shape[0], chunk):
        gg, cc, ff = g[a:a + chunk], c[a:a + chunk], f[a:a + chunk]
        e = 0.5 * (gg[:, :-1] + gg[:, 1:]) / h ** 2
        d = cc[:, 1:-1] - (e[:, :-1] + e[:, 1:])
        out[a:a + chunk] = ((d.abs() + e[:, :-1] + e[:, 1:]) / ff[:, 1:-1]).amax(dim=1)
    return out


def c4_adjoint_step(ctx, device, n_oracle=4, reps=3):
    """BASELINE configs[3] END TO END on one GPU (the reference: 73 x `srun ball_scan.py`, ball_scan.py:248-347, consumed by
    sims_runner_NCSX.py:249-261): 73 emulated equilibria (base + the 72 NCSX boundary DOFs stepped by create_dict.py:67-70 on
    the shipped tables; every equilibrium holds ITS OWN arrays: 115 MB of wout tables) x 5 surfaces x 24 alpha x 15 theta0,
    N = 969 -- ibs_amd.AdjointStep.run(): host tables (radial splines of all equilibria, native threaded routine) -> geometry
    -> coarse scan + per-surface argmax -> L-BFGS-B refinement of all 365 maxima -> final solve -> objective + 72-gradient.
    `total_ms` is the wall time of run() (it ends with the one copy of the rows to the host); the phases come from a separate
    pass with HIP events between them.  n_oracle (equilibrium, surface) pairs are re-done by the oracle pipeline."""
    import ibs_amd
    wout0 = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
    wouts, steps, x0 = emulated_equilibria(wout0)
    wouts = [{k: (np.array(v, copy=True) if isinstance(v, np.ndarray) else v) for k, v in w.items()} for w in wouts]
    n_eq, ns, na, nt0 = len(wouts), 5, 24, 15
    svals = np.linspace(0.5, 0.95, ns)                                   # ball_scan.py:197
    th = ibs_amd.theta_grid_for(11, 11)                                  # ball_scan.py:201-208: 969 points
    f_other = 0.8 + 0.01 * np.arange(n_eq)
    step = ibs_amd.AdjointStep(ctx, th, svals, device, nalpha=na, ntheta0=nt0, gamma_thresh=-2.0e-4, prefac=50.0)
    step.run(wouts, f_other, steps)                                       # warm-up (spline weights, workspaces, resident inputs)
    times = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = step.run(wouts, f_other, steps)
        times.append((time.perf_counter() - t0) * 1e3)
    ev = ctx.refine_stats()
    ph = {}
    step.run(wouts, f_other, steps, phases=ph)
    leg = dict(workload="configs[3]: %d emulated equilibria x %d surfaces x %d alpha x %d theta0 = %d coarse solves, N = %d, + "
                        "refinement of the %d maxima + final solve + objective and %d-gradient; one AdjointStep.run()" % (
                            n_eq, ns, na, nt0, n_eq * ns * na * nt0, len(th), n_eq * ns, n_eq - 1),
               total_ms=float(np.median(times)), total_ms_runs=[float(t) for t in times],
               phases_ms={k: float(v) for k, v in ph.items()}, phases_how="separate pass, HIP events between the phases "
               "(one extra synchronisation); host_tables and gather_copy_objective are host wall times",
               refine_evaluations=ev[0], refine_rounds=ev[2], coarse_solves=n_eq * ns * na * nt0,
               fobj=out["fobj"], dfobj_norm=float(np.linalg.norm(out["dfobj"])),
               gam_min=float(out["gam"].min()), gam_max=float(out["gam"].max()))
    if n_oracle:
        from oracle.pipeline import oracle_surface_pipeline
        rng = np.random.default_rng(41)
        errs = []
        t0 = time.perf_counter()
        for k in rng.choice(n_eq * ns, size=n_oracle, replace=False):
            q, js = divmod(int(k), ns)
            ref = oracle_surface_pipeline(wouts[q], float(svals[js]), th, na, nt0, step.del_alpha)
            errs.append(abs(ref["gam"] - out["gam"][q, js]))
        leg.update(max_abs_dgam_vs_oracle=float(max(errs)), oracle_pairs=int(n_oracle), parity_ok=bool(max(errs) < 1e-8),
                   oracle_seconds=time.perf_counter() - t0)
    return leg


def c5_matrix(ctx, device, n_sys=1 << 20, budget_s=75.0, oracle_check=True):
    """BASELINE configs[4] as stated: 10^6 (2^20) random (g, c, f) systems at N_zeta in {256, 512, 1024, 2048}, smooth and
    rough families, as FP64, FP32 eigenvalues only (all-FP32 solver, every result certified by an FP64 count
    pair) and FP32 with the growth rate (FP32 in HBM, FP64 in the solver).  Per row: solves/s, the HBM fraction on algorithmic
    bytes (3 N + 1) w, the kernel that ran with its PMC fields, sweeps per solve, flagged systems; for the FP32 rows the
    distance to the FP64 solve of the SAME (FP32-valued) systems in units of eps32 ||A||; for the FP64 rows the systems closed
    again in division form (informational status bit 3) and -- the checker, not the thing measured -- the distance of a sample
    (every re-closed system up to 256, and 2,048 random ones) to the C oracle's division-form bisection in units of ||A||, against
    the stated 4 N eps.  Rows are skipped (and say so) once the leg's time budget is spent."""
    import torch
    rows = []
    seen = {}                       # (kernel, waves) -> rows so far: the work class of the next one in the PMC summary
    t_leg = time.perf_counter()
    for nz in (256, 512, 1024, 2048):
        N = nz + 1
        for family in ("smooth", "rough"):
            if time.perf_counter() - t_leg > budget_s:
                rows.append(dict(n_zeta=nz, family=family, skipped="time budget of the leg spent"))
                continue
            h, g, c, f = c5_family(device, family, n_sys, N, seed=20240 + nz)           # rng seed 20240 + N_zeta
            g32, c32 = g.float(), c.float()
            f32 = f.float()
            g64w, c64w, f64w = g32.double(), c32.double(), f32.double()
            r64 = ctx.solve_gcf(h, g64w, c64w, f64w)        # the FP32-valued systems solved in FP64: the FP32 rows' reference
            nA = norm_a(h, g64w, c64w, f64w)
            del g64w, c64w, f64w
            for mode in ("f64", "f32_lam", "f32_gam"):
                if mode == "f64":
                    call = lambda want_info=False: ctx.solve_gcf(h, g, c, f, want_info=want_info)
                    w = 8
                elif mode == "f32_lam":
                    call = lambda want_info=False: ctx.solve_gcf(h, g32, c32, f32, want_info=want_info, dtype=np.float32, want_gam=False)
                    w = 4
                else:
                    call = lambda want_info=False: ctx.solve_gcf(h, g32, c32, f32, want_info=want_info, dtype=np.float32)
                    w = 4
                r = call(True)
                kern, waves = ctx.last_launch()
                torch.cuda.synchronize()
                evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
                for a, b in evs:
                    a.record(); call(); b.record()
                torch.cuda.synchronize()
                ms = float(min(a.elapsed_time(b) for a, b in evs))
                row = dict(n_zeta=nz, family=family, mode=mode, systems=n_sys, solves_per_s=n_sys / (ms * 1e-3), ms_per_launch=ms,
                           mean_sweeps=float((r["info"] & 0xffff).double().mean().item()),
                           flagged=int((((r["info"] >> 16) & 3) != 0).sum().item()),
                           roofline=hbm_roofline(n_sys * (3 * N + 1) * w, ms, "valu_issue", kern, waves, work_class=seen.get((kern, waves), 0),
                                                 bytes_per_solve=(3 * N + 1) * w))
                seen[(kern, waves)] = seen.get((kern, waves), 0) + 1
                if mode == "f64":
                    rc = torch.nonzero(((r["info"] >> 16) & 8) != 0).flatten()
                    row["reclosed_in_division_form"] = int(rc.numel())
                    checked = (row, rc, r["lam"])      # (the oracle runs after the three modes are timed: seconds of host work
                    #                                     between two timed modes let the GPU's clocks fall and cost the next mode 4-10 %)
                if mode == "f32_lam":     # (informational status bit 2: the all-FP32 result failed its FP64 certificate, solved in FP64)
                    row["resolved_in_f64"] = int((((r["info"] >> 16) & 4) != 0).sum().item())
                if mode != "f64":
                    el = (r["lam"].double() - r64["lam"]).abs() / nA / 1.1920929e-07
                    row["max_abs_dlam_over_eps32_normA"] = float(el.max().item())
                    row["tolerance_n_eps32_normA"] = float(nz + 4)
                    row["within_tolerance"] = bool(el.max().item() <= nz + 4)
                    if mode == "f32_gam" and family == "smooth":
                        row["max_abs_dgam_vs_f64"] = float((r["gam"].double() - r64["gam"]).abs().max().item())
                rows.append(row)
                del r
            if oracle_check:
                from oracle import c_oracle as co
                row, rc, lam64 = checked
                gen = torch.Generator(device=device); gen.manual_seed(7 + nz)
                pk = torch.unique(torch.cat([rc[:256], torch.randint(0, n_sys, (2048,), device=device, generator=gen)]))
                lam_c = co.lam_batch(h, g[pk].cpu().numpy(), c[pk].cpu().numpy(), f[pk].cpu().numpy())
                nA64 = norm_a(h, g[pk], c[pk], f[pk]).cpu().numpy()
                row["max_abs_dlam_over_normA"] = float((np.abs(lam64[pk].cpu().numpy() - lam_c) / nA64).max())
                row["tolerance_4N_eps"] = 4 * N * 2.220446049250313e-16
                row["within_tolerance"] = bool(row["max_abs_dlam_over_normA"] <= row["tolerance_4N_eps"])
                row["oracle_sample"] = int(pk.numel())
            del g, c, f, g32, c32, f32, r64, nA, checked
            torch.cuda.empty_cache()
    done = [r for r in rows if "mode" in r]
    by = lambda nz, fam, mode: next((r["solves_per_s"] for r in done if (r["n_zeta"], r["family"], r["mode"]) == (nz, fam, mode)), None)
    ratio = {"%d_%s" % (nz, fam): (by(nz, fam, "f32_gam") / by(nz, fam, "f64")) if by(nz, fam, "f32_gam") and by(nz, fam, "f64") else None
             for nz in (256, 512, 1024, 2048) for fam in ("smooth", "rough")}
    return dict(workload="configs[4]: %d random (g, c, f) systems per row, N_zeta x {f64, f32 eigenvalues only, f32 with growth rate} x "
                         "{smooth, rough}" % n_sys, rows=rows, f32_gam_over_f64=ratio,
                f32_results_outside_tolerance=int(sum(1 for r in done if r.get("within_tolerance") is False and r["mode"] != "f64")),
                f64_rows_outside_4N_eps=int(sum(1 for r in done if r.get("within_tolerance") is False and r["mode"] == "f64")),
                seconds=time.perf_counter() - t_leg)


def dropin_call(ctx, reps=200):
    """the literal two-line integration of INTEGRATION.md 2: ONE `ibs_amd.gamma_ball_full(dPdrho, theta, B, gradpar, cvdrift,
    gds2)` call on host numpy arrays (utils.py:1550-1552), N = 969 (the reference's NCSX grid): upload, one-wave solve with
    eigenfunction output, download -- the per-call latency a ball_scan.py loop would see (reference: ~55 ms per call)."""
    import ibs_amd
    g3 = np.load(os.path.join(ROOT, "tests", "golden", "G3_ncsx_lines.npz"))
    line = g3["geo_969"][1]
    th = ibs_amd.theta_grid(969)
    dP = float(-0.5 * np.mean((line[2] - line[7]) * line[0] ** 2))
    cv = line[2] + 0.3 * line[3]; gd = line[4] + 2 * 0.3 * line[5] + 0.09 * line[6]       # ball_scan.py:267-268 at theta0 = 0.3
    for _ in range(10):
        out = ibs_amd.gamma_ball_full(dP, th, line[0], line[1], cv, gd, ctx=ctx)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = ibs_amd.gamma_ball_full(dP, th, line[0], line[1], cv, gd, ctx=ctx)
    us = (time.perf_counter() - t0) / reps * 1e6
    return dict(workload="one gamma_ball_full(...) call on host arrays, N = 969 (reference signature, 6-tuple back)",
                us_per_call=us, calls_per_s=1e6 / us, gam=float(out[0]))


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (fresh interpreters, so nothing has
    touched the GPU before they start), wait for all of them, return the worst exit code.  If one rank dies, or this
    process is told to stop (SIGTERM / SIGINT) or raises, the ranks are stopped -- each by the process group it was
    started in, first SIGTERM, after a grace period SIGKILL -- instead of sitting in a collective for ever."""
    import signal
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    procs = []

    def stop_all(grace=5.0):
        for sig in (signal.SIGTERM, signal.SIGKILL):
            live = [p for p in procs if p.poll() is None]
            if not live:
                return
            for p in live:
                try:
                    os.killpg(p.pid, sig)           # (start_new_session: the rank's pid is its group id)
                except (ProcessLookupError, PermissionError):
                    pass
            t_end = time.time() + grace
            while time.time() < t_end and any(p.poll() is None for p in live):
                time.sleep(0.05)

    def on_signal(signum, frame):
        raise KeyboardInterrupt("signal %d" % signum)

    old = {sg: signal.signal(sg, on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          start_new_session=True))
        alive = list(procs)
        while alive:
            time.sleep(0.2)
            for p in list(alive):
                r = p.poll()
                if r is None:
                    continue
                alive.remove(p)
                if r != 0:
                    rc = rc or (r if r > 0 else 1)
                    stop_all()
                    alive = []
                    break
    except KeyboardInterrupt:
        rc = rc or 130
    finally:
        stop_all()
        for sg, h in old.items():
            signal.signal(sg, h)
    return rc


def sharded_surface_pass(local_rows, n_surf, rank, world, dist, ctx=None):
    """one pass of a surface-sharded scan (ball_scan.py:172, 251-252: one process group per surface): this rank's
    surfaces -> local_rows(own) = (len(own), k) tensor -> ONE all-gather -> (n_surf, k) in surface order everywhere.
    No other collective is on the data path."""
    import ibs_amd
    own = ibs_amd.shard_surfaces(n_surf, rank, world)
    return ibs_amd.gather_rows_tensor(local_rows(own), n_surf, rank, world, dist, ctx)


class C2Sharded:
    """BASELINE.json configs[2]: NCSX, 64 surfaces x 32 alpha x 16 theta0, N_zeta = 1024 (1,025 points), from the shipped
    equilibrium's wout tables; rows = (gam_max, alpha*, theta0*) of the coarse scan per surface (ball_scan.py:279-295)."""
    NS, NA, NT0, N = 64, 32, 16, 1025

    def __init__(self, ctx, device):
        import torch
        import ibs_amd
        wout = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
        self.ctx, self.device = ctx, device
        self.tabs = ibs_amd.SurfaceTables.from_wout(wout, np.linspace(0.1, 0.95, self.NS))
        self.th = ibs_amd.theta_grid(self.N)
        self.h = float(self.th[1] - self.th[0])
        self.alphas = np.linspace(0, np.pi, self.NA)
        self.alphas_d = torch.from_numpy(self.alphas).to(device)
        self.t0_d = torch.from_numpy(np.linspace(0, np.pi / 2, self.NT0)).to(device)
        self.th_d = torch.from_numpy(self.th).to(device)
        # (alpha, theta0) of every entry of a surface's row-major table: one gather turns the argmax index into the row
        self.lut = torch.stack([self.alphas_d.repeat_interleave(self.NT0), self.t0_d.repeat(self.NA)], dim=1)
        self._lines = {}                                    # own surfaces -> (line_surf, line_alpha) resident in HBM

    def local_rows(self, own):
        import torch
        if len(own) == 0:
            return torch.empty((0, 3), dtype=torch.float64, device=self.device)
        key = tuple(own)
        if key not in self._lines:
            self._lines[key] = (torch.from_numpy(np.repeat(np.asarray(own), self.NA).astype(np.int32)).to(self.device),
                                torch.from_numpy(np.tile(self.alphas, len(own))).to(self.device))
        surf, al = self._lines[key]
        r = self.ctx.fieldline_geometry(self.tabs, surf, al, self.th_d, device=self.device)
        sc = self.ctx.gamma_scan(self.h, *[r["geo"][k] for k in range(7)], r["dPdrho"], self.t0_d)
        idx, val = self.ctx.surface_argmax(sc["gam"].reshape(len(own), -1))
        return torch.cat([val[:, None], self.lut[idx.long()]], dim=1)


def c2_sharded_leg(ctx, device, rank, world, dist, fence, passes=20, native=False):
    """the north-star multi-GPU leg: strong scaling of configs[2] over the ranks.  Returns the leg's dict on rank 0."""
    import torch
    import ibs_amd
    job = C2Sharded(ctx, device)
    cc = ctx if native else None
    full = sharded_surface_pass(job.local_rows, job.NS, rank, world, dist, cc)    # warm-up + the table to check
    fence()
    t0 = time.perf_counter()
    for _ in range(passes):
        full = sharded_surface_pass(job.local_rows, job.NS, rank, world, dist, cc)
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    out = None
    if rank == 0:
        # the same surfaces on ONE GPU: (a) with the launches the ranks made (one per shard) -- must agree bitwise, the
        # gather only moves rows; (b) as a single 64-surface launch -- other theta0 chaining / lanes per system, i.e.
        # other shift sequences: agreement to rounding of the solver (1e-10 asserted), not bitwise
        per_shard = torch.empty_like(full)
        for r in range(world):
            own = ibs_amd.shard_surfaces(job.NS, r, world)
            per_shard[own] = job.local_rows(own)
        one_launch = job.local_rows(list(range(job.NS)))
        torch.cuda.synchronize()
        bitwise = bool(torch.equal(per_shard, full))
        dmax = float((one_launch[:, 0] - full[:, 0]).abs().max().item())
        same_arg = bool(torch.equal(one_launch[:, 1:], full[:, 1:]))
        # (reported, not raised: a rank that dies here would leave the others waiting in the final barrier)
        ok = bitwise and dmax < 1e-10
        if not ok:
            print("bench.py: ncsx_c2_sharded check FAILED (bitwise=%s, max|dgam| vs single launch=%g)" % (bitwise, dmax),
                  file=sys.stderr, flush=True)
        n = job.NS * job.NA * job.NT0
        out = dict(workload="configs[2]: 64 surfaces x 32 alpha x 16 theta0 = %d solves, N_zeta=1024, NCSX_op wout tables; "
                            "surfaces round-robin over %d ranks, geometry -> scan -> argmax per rank, ONE all-gather of "
                            "[n_surf_local, 3] (%s)" % (n, world, "ncclAllGather issued by the library" if native else "torch.distributed"),
                   scaling="strong", n_gpus=world, passes=passes, ms_per_pass=dt / passes * 1e3,
                   solves_per_s=n * passes / dt, checks_passed=ok, gathered_equals_one_gpu_bitwise=bitwise,
                   max_abs_dgam_vs_single_launch=dmax, same_argmax_as_single_launch=same_arg,
                   gam_max_min=float(full[:, 0].min().item()), gam_max_max=float(full[:, 0].max().item()))
    return out


def c2_refined_leg(ctx, device, rank, world, dist, fence, passes=3):
    """configs[2] as the reference's worker runs it (ball_scan.py:248-347): coarse scan -> argmax -> L-BFGS-B refinement on
    the device -> final solve per surface, surfaces round-robin over the ranks, ONE gather of the REFINED rows
    (theta0*, alpha*, gam) -- BallooningScan.run(), the product's own driver."""
    import torch
    import ibs_amd
    job = C2Sharded(ctx, device)
    svals = np.linspace(0.1, 0.95, job.NS)

    def scan_of(r):
        return ibs_amd.BallooningScan(ctx, None, job.th, svals, nalpha=job.NA, ntheta0=job.NT0, tables=job.tabs, device=device,
                                      rank=r, world=world, dist=dist, gather_device=device)
    scan = scan_of(rank)
    full = scan.run()                                        # warm-up + the rows to check
    fence()
    t0 = time.perf_counter()
    for _ in range(passes):
        full = scan.run()
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    out = None
    if rank == 0:
        full = np.stack(full, axis=1)                        # (n_surf, 3): theta0*, alpha*, gam
        per_shard = np.empty_like(full)
        for r in range(world):                               # the same shards, computed here: the gather only moves rows
            per_shard[ibs_amd.shard_surfaces(job.NS, r, world)] = scan_of(r).local_rows()
        bitwise = bool(np.array_equal(per_shard, full))
        coarse_max = job.local_rows(list(range(job.NS)))[:, 0].cpu().numpy()
        never_below = bool(np.all(full[:, 2] >= coarse_max - 1e-9))          # L-BFGS-B never ends below its start
        ev = ctx.refine_stats()
        out = dict(workload="configs[2] with the refinement: 64 surfaces x (32 alpha x 16 theta0 coarse scan + L-BFGS-B on the device "
                            "+ final solve), N_zeta=1024, surfaces round-robin over %d ranks, ONE gather of the refined rows" % world,
                   scaling="strong", n_gpus=world, passes=passes, ms_per_pass=dt / passes * 1e3,
                   checks_passed=bitwise and never_below, gathered_equals_one_gpu_bitwise=bitwise,
                   refined_never_below_coarse_max=never_below,
                   gam_refined_min=float(full[:, 2].min()), gam_refined_max=float(full[:, 2].max()),
                   last_refine_on_rank0={"evaluations": ev[0], "forward_sweeps": ev[1], "rounds": ev[2]})
    return out


def c4_sharded_leg(ctx, device, rank, world, dist, fence, passes=2):
    """BASELINE configs[3] over the ranks: the 73 equilibria of an optimizer step dealt round-robin (the reference's one srun per
    DOF, ball_submit.py:64-95), every rank running its share as ONE batch (AdjointStep), ONE all-gather of the per-equilibrium
    rows, objective and 72-gradient formed everywhere.  Checked on rank 0 against the same step computed there alone: growth rates
    to 1e-10 (the refinement's geometry form follows the batch size, i.e. other summation orders), gradient to 1e-8 relative."""
    import torch
    import ibs_amd
    wout0 = dict(np.load(os.path.join(ROOT, "tests", "golden", "G8_wout_ncsx_op.npz")))
    wouts, steps, x0 = emulated_equilibria(wout0)
    n_eq, ns = len(wouts), 5
    svals = np.linspace(0.5, 0.95, ns)
    th = ibs_amd.theta_grid_for(11, 11)
    f_other = 0.8 + 0.01 * np.arange(n_eq)
    step = ibs_amd.AdjointStep(ctx, th, svals, device, rank=rank, world=world, dist=dist,
                               gather_device=None if dist.get_backend() == "nccl" else "cpu")
    out = step.run(wouts, f_other, steps)                    # warm-up + the result to check
    fence()
    t0 = time.perf_counter()
    for _ in range(passes):
        out = step.run(wouts, f_other, steps)
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else torch.device("cpu"))
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    res = None
    if rank == 0:
        alone = ibs_amd.AdjointStep(ctx, th, svals, device).run(wouts, f_other, steps)
        dg = float(np.abs(alone["gam"] - out["gam"]).max())
        dd = float(np.abs(alone["dfobj"] - out["dfobj"]).max() / max(1.0, np.abs(alone["dfobj"]).max()))
        ok = dg < 1e-10 and dd < 1e-8
        if not ok:
            print("bench.py: c4_adjoint_step_sharded check FAILED (max|dgam| %g, relative gradient difference %g)" % (dg, dd), file=sys.stderr, flush=True)
        res = dict(workload="configs[3]: %d equilibria round-robin over %d ranks, each rank's share as one AdjointStep batch, ONE all-gather of "
                            "[n_eq_local, 3 n_surf + 1]" % (n_eq, world), scaling="strong", n_gpus=world, passes=passes,
                   ms_per_step=dt / passes * 1e3, checks_passed=ok, max_abs_dgam_vs_one_rank=dg, rel_dgradient_vs_one_rank=dd,
                   fobj=out["fobj"])
    return res


class Watchdog:
    """Bounds the phases that can hang for ever inside a collective (a second RCCL communicator next to torch's, the first
    multi-rank gathers of a new build).  When a phase overruns, rank 0 prints the JSON line it has so far -- the headline
    number was measured BEFORE any such phase starts, so the measurement is not lost -- and every rank leaves with status
    EXIT_CODE (5): a hung run must not look like a healthy one.  (Nothing is re-executed: a process that has touched the
    GPU must not exec.)"""
    EXIT_CODE = 5

    def __init__(self, rank, get_line):
        import threading
        self.rank, self.get_line, self.timer, self.threading = rank, get_line, None, threading

    def line_printed(self):
        self.get_line = lambda: None

    def arm(self, phase, seconds):
        self.disarm()

        def fire():
            try:
                if self.rank == 0:
                    line = self.get_line()
                    if line is not None:
                        for _ in range(3):          # (the main thread may be adding a key to the line right now)
                            try:
                                text = compact_line(dict(line, aborted="phase '%s' did not finish within %d s" % (phase, seconds)))
                                break
                            except RuntimeError:
                                time.sleep(0.05)
                        else:
                            text = json.dumps({k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in line})
                        print(text, flush=True)
                print("bench.py: rank %d: phase '%s' overran %d s, leaving" % (self.rank, phase, seconds), file=sys.stderr, flush=True)
            finally:
                os._exit(self.EXIT_CODE)
        self.timer = self.threading.Timer(seconds, fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def rocprof_kernel_ms(prefix):
    """average duration of the named kernel in the newest committed rocprofv3 kernel-trace summary of this bench
    (profiles/*_kernel_stats_bench.csv): (ms, file name) or (None, None)"""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats_bench.csv")))
    for f in reversed(files):
        try:
            for row in csv.DictReader(open(f)):
                if prefix in row.get("Name", ""):
                    return float(row["AverageNs"]) * 1e-6, os.path.basename(f)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-stress", action="store_true", help="skip the config-5 stress leg")
    ap.add_argument("--stress-systems", type=int, default=262144)
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: become the launcher.  Nothing GPU-related has been imported or called in this process.
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    import torch
    import torch.distributed as dist
    import ibs_amd

    force_dist = os.environ.get("IBS_BENCH_FORCE_DIST") == "1"   # rehearsal of the N > 1 path with a 1-rank RCCL group
    # IBS_BENCH_SHARE_GPU=1: rehearsal of N ranks on a one-GPU box (every rank on device 0; RCCL refuses two ranks on one
    # device, so the collectives then go through gloo on host copies).  Never set by the driver.
    share = os.environ.get("IBS_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    backend = "gloo" if share else "nccl"
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    ctx = ibs_amd.Context(local)

    h, geo7, dP_d, th0_d, base, dP, theta0 = build_workload(rank, device)
    plan = ibs_amd.ScanPlan(ctx, h, geo7, dP_d, th0_d, N_SURF, n_pack=3)
    n_solves = N_SURF * N_ALPHA * N_THETA0
    use_dist = dist.is_available() and dist.is_initialized()
    n_ranks = dist.get_world_size() if use_dist else 1
    coll_dev = device if backend == "nccl" else torch.device("cpu")
    # N > 1: the per-surface maxima of every step are all-gathered (replaces comm_lead.Gather x3, ball_scan.py:345-347;
    # 256 B per rank, latency-bound).  THE HEADLINE is timed first and with the plainest form -- torch.distributed's
    # all_gather_into_tensor in the step's stream: nothing that has never run on more than one GPU stands between the
    # start of the process and the number.  The library's own ncclAllGather (in-stream, and overlapped with the next scans
    # on the communicator's stream) is timed AFTERWARDS as extra legs, under a watchdog (`gather_modes`).
    gathered = torch.empty((n_ranks * N_SURF, 2), dtype=torch.float64, device=coll_dev) if use_dist else None
    # (the library's own gathers work on device buffers whatever torch.distributed's backend is: with the shared-memory stand-in
    #  for librccl, IBS_RCCL_LIB, they also run in the one-GPU rehearsal, where torch.distributed goes through gloo)
    gath_nat = (gathered if backend == "nccl" else torch.empty((n_ranks * N_SURF, 2), dtype=torch.float64, device=device)) if use_dist else None
    gathered3 = [gath_nat, torch.empty_like(gath_nat), torch.empty_like(gath_nat)] if use_dist else None
    n_issued = [0]                       # steps issued so far (the slot rotation must not depend on the caller's k)
    mode = ["torch" if use_dist else "none"]       # "none" | "torch" | "native" | "overlap"

    def step(k=0, ev=None):
        # scan + per-surface first maximum: ONE kernel (the block that completes a surface reduces it)
        ov = mode[0] == "overlap"
        slot = n_issued[0] % 3 if ov else 0
        if ev is not None:
            ev[0].record()
        plan.scan_argmax(slot)
        if ev is not None:
            ev[1].record()
        if ov:
            # step k's gather runs on the communicator's own stream while the next steps scan; pack / receive buffers rotate
            # through three slots, and before a slot's next scan is launched the HOST makes sure the gather that last used it
            # has finished (an event query; it blocks only if the ranks have fallen two steps behind)
            ctx.allgather_start(plan.packs[slot], gathered3[slot], slot, same_stream=True, host_wait=(n_issued[0] + 1) % 3)
            n_issued[0] += 1
        elif mode[0] == "native":
            ctx.allgather(plan.pack, gath_nat)
        elif mode[0] == "torch":
            dist.all_gather_into_tensor(gathered, plan.pack if backend == "nccl" else plan.pack.cpu())

    def fence():
        if mode[0] == "overlap":
            ctx.comm_wait(-1)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(n_steps, with_events):
        """K steps between two fences; max over the ranks.  Returns (seconds, live kernel ms or None)."""
        # the dominant kernel is bracketed by a HIP-event pair on eight steps of the region (at least ten steps apart): a pair costs
        # 6.5 us of stream time (tools/experiments/step_fixed_cost.py) -- on every 8th step that was 0.7 us per step, 2.8 % of `value`
        EV = max(10, n_steps // 8)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range((n_steps + EV - 1) // EV)] if with_events else None
        if with_events:               # (torch creates the HIP event at its first record: that is not the region's work)
            for a, b in evs:
                a.record(); b.record()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_steps):
            step(k, evs[k // EV] if (with_events and k % EV == 0) else None)
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, (float(np.mean([a.elapsed_time(b) for a, b in evs])) if with_events else None)

    def roundtrip_ok():
        """every rank holds every rank's maxima of the last step; its own row must be what it sent"""
        last = (n_issued[0] - 1) % 3 if mode[0] == "overlap" else 0
        buf = gathered3[last] if mode[0] == "overlap" else (gath_nat if mode[0] == "native" else gathered)
        got = buf[rank * N_SURF:(rank + 1) * N_SURF].to(device)
        okt = torch.tensor([1.0 if torch.equal(got, plan.packs[last]) else 0.0], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)       # (reported, not raised: no rank may leave the others in a collective)
        return bool(okt.item() == 1.0)

    def ranks_seen(native):
        """the number of distinct ranks counted INSIDE the collective: every rank contributes its id to one all-gather"""
        mine = torch.full((1,), float(rank), dtype=torch.float64, device=device if native else coll_dev)
        allr = torch.empty((n_ranks,), dtype=torch.float64, device=device if native else coll_dev)
        if native:
            ctx.allgather(mine, allr)
        else:
            dist.all_gather_into_tensor(allr, mine)
        torch.cuda.synchronize()
        return int(len(set(int(v) for v in allr.cpu().tolist())))

    def spin_on(mine):
        """N > 1: every step holds a collective, so every rank must take the SAME number of them -- a rank that read its own clock
        a few microseconds past the mark while its neighbour did not would leave the others inside 64 all-gathers it never joins.
        The ranks agree (all continue only while all want to)."""
        if not use_dist:
            return mine
        t = torch.tensor([1.0 if mine else 0.0], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() == 1.0)

    for k in range(args.warmup):
        step(k)
    # clocks: whatever --warmup says, at least 150 ms of untimed steps run before the timed region (a 20-step driver
    # run is 0.7 ms long: without this it would be measured on an idle chip's clocks)
    torch.cuda.synchronize()
    t_spin = time.perf_counter()
    n_spin = 0
    while spin_on(time.perf_counter() - t_spin < 0.15):
        for k in range(64):
            step(k)
        torch.cuda.synchronize()
        n_spin += 64
    fence()
    dt, kern_ms_live = timed_steps(args.steps, True)
    # `kernel_ms`: an UNTIMED pass of 25 brackets, each around 8 back-to-back launches of the step's one kernel, divided
    # by 8.  A bracket around a single launch over-reads rocprofv3's kernel-trace average by ~3 us (the event records
    # are stream work themselves: `event_bracket_ms` is an empty bracket), which is how the figure of a short run could
    # exceed ms_per_step; spread over 8 launches that is < 0.5 us, and the launch gap it adds instead is part of every
    # step as well.  The sparse single-launch brackets inside the timed region are reported as `kernel_ms_live_raw`.
    n_un, per = 25, 8
    un = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_un)]
    emp = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_un)]
    for (a, b), (c, d) in zip(un, emp):
        a.record()
        for _ in range(per):
            plan.scan_argmax()
        b.record()
        c.record(); d.record()
    torch.cuda.synchronize()
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in un])) / per
    head_kernel, head_waves = ctx.last_launch()          # the step's one kernel: ibs::k_gamma_scan<double, 8>, 1,024 waves
    empty_ms = float(np.median([c.elapsed_time(d) for c, d in emp]))
    gather_ok = roundtrip_ok() if use_dist else None
    seen = ranks_seen(False) if use_dist else 1
    info = plan.info.cpu().numpy()
    nbad = int(((info >> 16) != 0).sum())
    sweeps = float((info & 0xffff).mean())

    out = None
    if rank == 0:
        bytes_per_solve = (7 * NPTS * 8 + 8) / N_THETA0 + 8            # geometry-fed path
        alg_bytes = n_solves * bytes_per_solve
        rp_ms, rp_file = rocprof_kernel_ms(head_kernel.replace("ibs::", ""))
        out = {
            "metric": "field-line eigenvalue solves/sec (N_zeta=512)",
            "value": n_ranks * n_solves * args.steps / dt,
            "unit": "solves/s",
            "n_gpus": n_ranks, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (NCSX_op-derived field-line geometry, perturbed per line)",
            # the same workload priced on the kernel alone (n_ranks x solves per launch / kernel_ms): what a run of any
            # length converges to; `value` of a 20-step run carries the two fences of a 0.6 ms timed region
            "solves_per_s_kernel": n_ranks * n_solves / (kern_ms * 1e-3),
            "config": {"workload": "configs[1] D3D-shape: 16 surfaces x 8 alpha x 8 theta0 = 1024 solves/step/GPU, "
                                   "N_zeta=512 (513 points), geometry-fed scan with fused per-surface argmax (one launch)"
                                   + (" + all-gather (%s, %d ranks, in the step's stream) of the per-surface maxima" % (
                                       "RCCL through torch.distributed" if backend == "nccl" else "gloo rehearsal", n_ranks) if use_dist else ""),
                       "solves_per_step_per_gpu": n_solves, "mean_sweeps_per_solve": sweeps,
                       "nonconverged": nbad, "ranks_in_collective": seen, "allgather_roundtrip_ok": gather_ok,
                       "untimed_spinup_steps": args.warmup + n_spin,
                       "headline_gather": "torch_in_stream" if use_dist else None},
            # `bound` = what binds this launch (FP64 VALU issue: 1,024 waves on 1,024 SIMDs); achieved / peak /
            # frac (= hbm_frac) are the HBM figures on algorithmic bytes the task defines; `valu_issue` prices the launch
            # against the binding bound.  Counters: the PMC entry of EXACTLY this kernel at this launch size.
            "roofline": hbm_roofline(
                alg_bytes, kern_ms, "valu_issue", head_kernel, head_waves,
                kernel_role="scan + fused per-surface argmax, one launch", kernel_ms=kern_ms,
                kernel_ms_how="untimed pass after the timed region: 25 HIP-event brackets around 8 back-to-back launches "
                              "each, / 8 (includes the launch gap; rocprofv3's kernel-trace average of the same kernel: "
                              "kernel_ms_rocprof)",
                kernel_ms_rocprof=rp_ms, kernel_ms_rocprof_source=("profiles/" + rp_file) if rp_file else None,
                event_bracket_ms=empty_ms, kernel_ms_live_raw=kern_ms_live,
                note="FP64-VALU-issue bound, not HBM bound (DESIGN.md 4); every step re-scans the same 3.7 MB of geometry, "
                     "which stays in L2 / Infinity Cache: the HBM figures are nominal for this leg"),
        }

    # ---- N > 1: everything after the headline runs under a watchdog -- `out` already holds the headline, and a phase that
    # overruns (a rank that left a leg early while the others wait in its collective, a communicator that never comes up)
    # ends with rank 0 printing what it has and every rank leaving with status 5 (Watchdog.EXIT_CODE)
    dog = Watchdog(rank, lambda: out)
    if use_dist and os.environ.get("IBS_BENCH_DIE_RANK") == str(rank) and os.environ.get("IBS_BENCH_DIE_AFTER") == "headline":
        os._exit(17)           # test hook (tests/test_gpu_round5.py): a rank that dies while the others enter a collective
    if use_dist:
        for key, leg, label in (("ncsx_c2_sharded", lambda: c2_sharded_leg(ctx, device, rank, world, dist, fence, native=False),
                                 "configs[2] sharded, torch.distributed gather"),
                                ("ncsx_c2_sharded_refined", lambda: c2_refined_leg(ctx, device, rank, world, dist, fence),
                                 "configs[2] sharded with the refinement"),
                                ("c4_adjoint_step_sharded", lambda: c4_sharded_leg(ctx, device, rank, world, dist, fence),
                                 "configs[3] sharded over the equilibria")):
            dog.arm(label, 300)
            try:
                res = leg()
            except Exception as e:      # (every rank runs the same collectives inside the leg; an error is reported, not raised)
                res = dict(error="%s: %s" % (type(e).__name__, e)) if rank == 0 else None
            if out is not None and res is not None:
                out[key] = res
        dog.disarm()

    # ---- the library's own collective, as extra legs
    if use_dist and (backend == "nccl" or os.environ.get("IBS_RCCL_LIB")) and os.environ.get("IBS_BENCH_NATIVE_COLL", "1") != "0":
        modes = {"torch_in_stream": {"ms_per_step": dt / args.steps * 1e3, "solves_per_s": n_ranks * n_solves * args.steps / dt,
                                     "allgather_roundtrip_ok": gather_ok, "ranks_in_collective": seen}}
        if out is not None:
            out["gather_modes"] = modes
        dog.arm("ncclCommInitRank of the library's communicator", 90)
        flag = torch.ones(1, dtype=torch.float64, device=coll_dev)
        try:
            ctx.comm_init(dist, rank, n_ranks)
        except Exception as e:
            print("bench.py: native RCCL communicator not available (%s)" % e, file=sys.stderr)
            modes["native_error"] = "%s: %s" % (type(e).__name__, e)
            flag.zero_()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank or none
        native = bool(flag.item() == 1.0)
        if native:
            for name in ("native", "overlap"):
                dog.arm("per-step gather, mode '%s'" % name, 120)
                mode[0] = name
                n_issued[0] = 0
                for k in range(max(args.warmup, 16)):
                    step(k)
                fence()
                dtm, _ = timed_steps(args.steps, False)
                modes["native_in_stream" if name == "native" else "native_overlapped"] = {
                    "ms_per_step": dtm / args.steps * 1e3, "solves_per_s": n_ranks * n_solves * args.steps / dtm,
                    "allgather_roundtrip_ok": roundtrip_ok(),
                    "ranks_in_collective": ranks_seen(True) if name == "native" else None}
            # The headline: every mode above timed the same K steps of the same job between the same fences; the line reports
            # the fastest one whose round trip checked out and names it (`config.headline_gather`); the others stay in
            # `gather_modes`.  (Until this point -- and if a watchdog fires before it -- the line carries the torch.distributed figure.)
            if out is not None:
                valid = {k: v for k, v in modes.items() if isinstance(v, dict) and v.get("allgather_roundtrip_ok")}
                if valid:
                    best = min(valid, key=lambda k: valid[k]["ms_per_step"])
                    out["value"] = valid[best]["solves_per_s"]
                    out["ms_per_step"] = valid[best]["ms_per_step"]
                    out["config"]["headline_gather"] = best
            mode[0] = "native"
            dog.arm("configs[2] sharded with the library's gather", 180)
            try:
                c2n = c2_sharded_leg(ctx, device, rank, world, dist, fence, native=True)
            except Exception as e:
                c2n = dict(error="%s: %s" % (type(e).__name__, e)) if rank == 0 else None
            if out is not None and c2n is not None:
                out["ncsx_c2_sharded_native"] = c2n
            dog.arm("ncclCommDestroy", 60)
            ctx.comm_destroy()
        mode[0] = "torch"
        dog.disarm()

    rc = 0
    if rank == 0:
        if world == 1 and not use_dist and not args.no_cpu:
            cb, gam_cpu = cpu_baseline(h, base, dP, theta0)
            out["cpu_baseline"] = cb
            out["max_abs_dgam_vs_oracle"] = float(np.abs(plan.gam.cpu().numpy() - gam_cpu).max())
            # the bench's own workload against the C oracle: 1e-8 is the stated FP64 tolerance
            out["parity_ok"] = bool(out["max_abs_dgam_vs_oracle"] < 1e-8 and nbad == 0)
            if not out["parity_ok"]:
                rc = 3
            out["cpu_reference_cost"] = cpu_reference_cost(h, base, dP, theta0)
        if world == 1 and not use_dist and not args.no_stress:
            out["stress"] = stress(ctx, device, args.stress_systems, "smooth")
            out["stress_rough"] = stress(ctx, device, max(args.stress_systems // 4, 1024), "rough")
            out["sturm_sweep"] = sturm_sweep(ctx, device, args.stress_systems)
            out["warm_rescan"] = warm_rescan(ctx, device, h, geo7, dP_d, th0_d)
            out["scan_large"] = scan_large(ctx, device, geo7, dP_d)
            out["batch_scaling"] = batch_scaling(ctx, device, h, geo7, dP_d, th0_d)
            out.update(ncsx_pipeline(ctx, device))
            out["dropin_call_us"] = dropin_call(ctx)
            out["long_grid"] = long_grid(ctx, device, oracle_check=not args.no_cpu)
            if out["long_grid"].get("parity_ok") is False:
                rc = 3
            out["c4_adjoint_step"] = c4_adjoint_step(ctx, device, n_oracle=0 if args.no_cpu else 4)
            if out["c4_adjoint_step"].get("parity_ok") is False:
                rc = 3
            out["c5_matrix"] = c5_matrix(ctx, device, oracle_check=not args.no_cpu)
        emit(out)
        if rc:
            print("bench.py: parity check FAILED: max |gam - oracle| = %g (bar 1e-8), flagged solves %d" % (
                out["max_abs_dgam_vs_oracle"], nbad), file=sys.stderr, flush=True)
    if use_dist:
        dog.line_printed()               # (the one JSON line is out: a late overrun must not print a second one)
        dog.arm("final barrier / destroy_process_group", 120)
        dist.barrier()
        dist.destroy_process_group()
        dog.disarm()
    if rc:
        sys.exit(rc)


if __name__ == "__main__":
    main()
