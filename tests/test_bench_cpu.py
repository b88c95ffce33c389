"""CPU: the parts of bench.py that decide what the driver's JSON line may quote (no GPU involved)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _entry(kernel, waves, valu, spread=1.0, fetch_kib=100.0, write_kib=10.0):
    return dict(kernel=kernel, SQ_WAVES=dict(mean=float(waves), n=3), SQ_INSTS_VALU=dict(mean=float(valu), n=3),
                valu_spread=spread, FETCH_SIZE=dict(mean=fetch_kib, n=3), WRITE_SIZE=dict(mean=write_kib, n=3),
                hbm_bytes_per_launch=(2 * fetch_kib + write_kib) * 1024, valu_insts_per_wave=valu / waves,
                valu_busy_frac_of_wave_lifetime=0.5)


def test_pmc_counters_are_quoted_for_the_exact_kernel_and_batch_size_only(monkeypatch):
    """the headline quoted the counters of k_gamma_scan<double, 16> (first key with the prefix).  A leg may
    only quote the entry of EXACTLY its kernel whose launches held the leg's number of waves and were all alike."""
    import bench
    table = {"_set": "test", "_src_sha": bench.src_sha(),
             "ibs::k_gamma_scan<double, 16> @460800x256": _entry("ibs::k_gamma_scan<double, 16>", 1800, 1.35e7),
             "ibs::k_gamma_scan<double, 8> @65536x256": _entry("ibs::k_gamma_scan<double, 8>", 1024, 6.0e6),
             "ibs::k_gamma_scan<double, 8> @131072x256": _entry("ibs::k_gamma_scan<double, 8>", 2048, 1.2e7),
             "ibs::k_geo_rows<2, 1, 12> @131072x512": _entry("ibs::k_geo_rows<2, 1, 12>", 2048, 1.6e8, spread=9.3)}
    monkeypatch.setattr(bench._pmc_file, "cache", table, raising=False)
    e, why = bench.pmc_entry("ibs::k_gamma_scan<double, 8>", 1024)
    assert why is None and e["SQ_INSTS_VALU"]["mean"] == 6.0e6
    e, why = bench.pmc_entry("ibs::k_gamma_scan<double, 8>", 4096)          # another batch size: refused
    assert e is None and "4096" in why
    e, why = bench.pmc_entry("ibs::k_gamma_scan<double", 1024)              # a prefix is not a kernel
    assert e is None
    e, why = bench.pmc_entry("ibs::k_geo_rows<2, 1, 12>", 2048)             # launches of mixed work under one entry
    assert e is None and "mixes" in why
    table["ibs::k_solve_gcf<double, 16> @67108864x64 #0"] = dict(_entry("ibs::k_solve_gcf<double, 16>", 1048576, 7.0e9), work_class=0)
    table["ibs::k_solve_gcf<double, 16> @67108864x64 #1"] = dict(_entry("ibs::k_solve_gcf<double, 16>", 1048576, 9.6e9), work_class=1)
    e0, _ = bench.pmc_entry("ibs::k_solve_gcf<double, 16>", 1048576, 0)       # smooth family first, rough second: by class
    e1, _ = bench.pmc_entry("ibs::k_solve_gcf<double, 16>", 1048576, 1)
    assert e0["SQ_INSTS_VALU"]["mean"] == 7.0e9 and e1["SQ_INSTS_VALU"]["mean"] == 9.6e9
    assert bench.pmc_entry("ibs::k_solve_gcf<double, 16>", 1048576, 2)[0] is None
    r = bench.hbm_roofline(3686400, 0.025, "valu_issue", "ibs::k_gamma_scan<double, 8>", 1024)
    assert r["bound"] == "valu_issue" and r["frac"] == r["hbm_frac"] and abs(r["achieved"] - 147.456) < 1e-9
    assert r["traffic"] == (2 * 100.0 + 10.0) * 1024 and r["valu_insts_per_wave"] == 6.0e6 / 1024
    assert abs(r["valu_issue"]["frac"] - 6.0e6 / 0.025e-3 / bench.ISSUE_PEAK) < 1e-12
    r = bench.hbm_roofline(3686400, 0.025, "valu_issue", "ibs::k_gamma_scan<double, 8>", 512)
    assert r["traffic"] is None and "counters_error" in r and "valu_issue" not in r


def test_committed_pmc_file_serves_the_headline_kernel():
    """profiles/pmc_current.json as committed: if it was taken on THIS tree's kernel sources the headline kernel's entry is
    found by exact name at 1,024 waves and its traffic is within 1.5x of the algorithmic 3.69 MB (the round-3 line quoted
    7.3 MB from another kernel); if the sources have changed since, the lookup must say "stale" instead of serving it."""
    import json
    import bench
    for attr in ("cache",):
        if hasattr(bench._pmc_file, attr):
            delattr(bench._pmc_file, attr)
    if hasattr(bench.src_sha, "tree"):
        del bench.src_sha.tree
    d = json.load(open(bench.PMC_FILE))
    e, why = bench.pmc_entry("ibs::k_gamma_scan<double, 8>", 1024)
    if (d.get("_src_sha") or {}).get("solver") == bench.src_sha()["solver"]:
        assert why is None, why
        assert 3.6e6 < e["hbm_bytes_per_launch"] < 5.5e6
    else:
        assert e is None and why.startswith("stale"), why


def test_stale_or_unattributed_counters_are_not_replayed(monkeypatch):
    """VERDICT r4 weak 8 / ADVICE r4: a PMC set taken before the kernel sources changed is refused (`counters_error` "stale...",
    `traffic` null -- per kernel group: a geometry edit does not stale the solver's entries), and a set whose byte passes could
    not be matched launch by launch keeps its instruction counts but withholds `traffic`."""
    import bench
    sha = bench.src_sha()
    assert set(sha) == {"geometry", "solver"} and all(len(v) == 16 for v in sha.values())
    monkeypatch.setattr(bench.src_sha, "tree", sha, raising=False)
    table = {"_set": "t", "_src_sha": dict(sha, geometry="0" * 16),
             "ibs::k_gamma_scan<double, 8> @65536x256 #0": _entry("ibs::k_gamma_scan<double, 8>", 1024, 6.0e6),
             "ibs::k_geo_rows<2, 1, 12> @131072x512 #0": _entry("ibs::k_geo_rows<2, 1, 12>", 2048, 1.6e8)}
    monkeypatch.setattr(bench._pmc_file, "cache", table, raising=False)
    assert bench.pmc_entry("ibs::k_gamma_scan<double, 8>", 1024)[1] is None
    e, why = bench.pmc_entry("ibs::k_geo_rows<2, 1, 12>", 2048)
    assert e is None and why.startswith("stale")
    r = bench.hbm_roofline(1.0e6, 0.5, "valu_issue", "ibs::k_geo_rows<2, 1, 12>", 2048)
    assert r["traffic"] is None and r["counters_error"].startswith("stale") and "valu_issue" not in r
    assert bench.compact_roofline(r)["counters"] == "none:stale"
    table["_src_sha"] = None                                   # a set from before the guard existed: stale as well
    assert bench.pmc_entry("ibs::k_gamma_scan<double, 8>", 1024)[1].startswith("stale")
    table["_src_sha"] = sha
    table["_unmatched_launches_of_the_byte_passes"] = 3
    r = bench.hbm_roofline(3686400, 0.025, "valu_issue", "ibs::k_gamma_scan<double, 8>", 1024)
    assert r["traffic"] is None and "traffic_over_algorithmic" not in r and r["counters_error"].startswith("unmatched")
    assert r["valu_issue"]["frac"] > 0 and "withheld" in bench.compact_roofline(r)["counters"]


def test_the_stdout_line_fits_the_drivers_capture():
    """the round-4 line was 40.5 KB, the driver keeps ~8 KB and BENCH_r04.json.parsed was null.  The line
    assembled from a full recorded run (profiles/r04c_bench.json: every N = 1 leg) and from a 2-rank run stays under 6,000
    bytes, is one line, parses, and still holds the contract's keys, `roofline`, `cpu_baseline` and one summary per leg."""
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04c_bench.json")))
    assert len(json.dumps(full)) > 30000
    line = bench.compact_line(full)
    assert len(line) < bench.LINE_LIMIT == 6000 and "\n" not in line
    o = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "cpu_reference_cost", "parity_ok", "max_abs_dgam_vs_oracle",
              "stress", "stress_rough", "sturm_sweep", "scan_large", "ncsx_c3", "reference_batch", "c4_adjoint_step", "c5_matrix",
              "dropin_call_us"):
        assert k in o, k
    assert o["value"] == full["value"] and o["ms_per_step"] == full["ms_per_step"]            # the headline is not rounded
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "valu_issue_frac", "counters"):
        assert k in o["roofline"], k
    assert set(o["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    assert "dropped_for_length" not in o
    c5 = o["c5_matrix"]
    assert len(c5["rows"]) == 24 and all(len(r) == len(c5["cols"]) for r in c5["rows"])
    r0 = full["c5_matrix"]["rows"][0]
    assert c5["rows"][0][:3] == [256, "s", "f64"] and abs(c5["rows"][0][3] / r0["solves_per_s"] - 1) < 1e-3
    assert c5["kernels"][c5["rows"][0][7]] == r0["roofline"]["kernel"]
    assert o["c4_adjoint_step"]["parity_ok"] is True and abs(o["c4_adjoint_step"]["total_ms"] / full["c4_adjoint_step"]["total_ms"] - 1) < 1e-4
    # this round's full record (profiles/r05b_bench_detail.json: the direct / two-wave kernels' names in c5_matrix)
    full5 = json.load(open(os.path.join(ROOT, "profiles", "r05b_bench_detail.json")))
    line5 = bench.compact_line(full5)
    o5 = json.loads(line5)
    assert len(line5) < 6000 and "dropped_for_length" not in o5 and o5["value"] == full5["value"]
    assert len(o5["c5_matrix"]["rows"]) == 24 and any("k_solve_gcf_direct" in k for k in o5["c5_matrix"]["kernels"])
    # round 6 (profiles/r06a_bench_detail.json: the per-row accuracy fields of c5_matrix, the division-form count beside the sweep) and
    # the 2-rank run through the stand-in for librccl (three gather modes with two ranks in the collective, the native sharded leg)
    full6 = json.load(open(os.path.join(ROOT, "profiles", "r06a_bench_detail.json")))
    line6 = bench.compact_line(full6)
    o6 = json.loads(line6)
    assert len(line6) < 6000 and "dropped_for_length" not in o6 and o6["value"] == full6["value"]
    assert o6["roofline"]["counters"] == "pmc:r06a" and o6["cpu_baseline"]["value_1core"] > 0
    rec6 = json.loads(open(os.path.join(ROOT, "profiles", "r06a_bench_line.json")).read())       # the line that run printed
    assert {k: v for k, v in rec6.items() if k != "detail"} == {k: v for k, v in o6.items() if k != "detail"}
    assert all(r["within_tolerance"] for r in full6["c5_matrix"]["rows"]) and full6["c5_matrix"]["f64_rows_outside_4N_eps"] == 0
    two6 = json.load(open(os.path.join(ROOT, "profiles", "r06a_rehearsal_standin_2ranks.json")))
    l26 = bench.compact_line(two6)
    assert len(l26) < 6000 and set(json.loads(l26)["gather_modes"]) >= {"torch_in_stream", "native_in_stream", "native_overlapped"}
    assert two6["gather_modes"]["native_in_stream"]["ranks_in_collective"] == 2 and two6["ncsx_c2_sharded_native"]["checks_passed"] is True
    # a 2-rank line (profiles/r04_rehearsal_2rank.json): the sharded legs keep their flags
    two = next(json.loads(l) for l in open(os.path.join(ROOT, "profiles", "r04_rehearsal_2rank.json")) if l.startswith("{"))
    o2 = json.loads(bench.compact_line(two))
    assert o2["config"]["ranks_in_collective"] == 2 and o2["ncsx_c2_sharded"]["checks_passed"] is True
    assert o2["c4_adjoint_step_sharded"]["checks_passed"] is True and len(bench.compact_line(two)) < 6000
    # the line of an N = 8 RCCL run as the driver's SCALE tier would get it: the three gather modes and the native sharded leg on top
    eight = dict(two, n_gpus=8, gather_modes={m: dict(ms_per_step=0.05, solves_per_s=1.6e8, allgather_roundtrip_ok=True, ranks_in_collective=8)
                                              for m in ("torch_in_stream", "native_in_stream", "native_overlapped")},
                 ncsx_c2_sharded_native=dict(two["ncsx_c2_sharded"], workload="x" * 300))
    l8 = bench.compact_line(eight)
    o8 = json.loads(l8)
    assert len(l8) < 6000 and "dropped_for_length" not in o8 and set(o8["gather_modes"]) == {"torch_in_stream", "native_in_stream", "native_overlapped"}
    assert "workload" not in o8["ncsx_c2_sharded_native"] and o8["ncsx_c2_sharded_native"]["checks_passed"] is True
    # whatever a future leg adds, the bound holds: legs are dropped in a stated order, the contract's keys never
    fat = dict(full, future_leg={"rows": [{"x": float(i), "name": "k" * 40} for i in range(400)]})
    fl = bench.compact_line(fat)
    assert len(fl) <= 6000 and {"value", "roofline", "cpu_baseline", "c4_adjoint_step"} <= set(json.loads(fl))
    assert json.loads(fl)["dropped_for_length"] == ["future_leg"] and "c5_matrix" in json.loads(fl)


def test_emit_writes_detail_file_and_one_stdout_line(tmp_path, monkeypatch, capsys):
    import json
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04c_bench.json")))
    monkeypatch.setattr(bench, "DETAIL_FILE", str(tmp_path / "bench_detail.json"))
    bench.emit(full)
    cap = capsys.readouterr()
    assert cap.out.count("\n") == 1 and json.loads(cap.out)["value"] == full["value"]
    assert json.load(open(tmp_path / "bench_detail.json")) == full
    assert cap.err.startswith("bench.py detail: ") and json.loads(cap.err[len("bench.py detail: "):]) == full


def test_pmc_summary_splits_launches_into_work_classes(tmp_path):
    """tools/pmc_summary.py: launches of one kernel at one launch size on different data get separate entries (work classes by
    VALU instruction count, in order of first appearance); the byte passes are matched to them by launch ordinal."""
    import json
    import subprocess
    hdr = '"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",' \
          '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value",' \
          '"Start_Timestamp","End_Timestamp"\n'
    kern = '"void ibs::k_solve_gcf<double, 16>(long, int, double)"'
    other = '"void at::native::vectorized_elementwise_kernel<4>(int)"'

    def rows(did, name, counters):
        return "".join('%d,%d,"Agent 2",1,1,1,4096,7,%s,64,0,0,8,0,32,"%s",%f,1,2\n' % (did, did, name, c, v) for c, v in counters)
    # dispatch order: smooth, smooth, (a torch kernel), rough, rough, smooth
    valu = [7000.0, 7001.0, None, 9600.0, 9590.0, 7000.5]
    sq = hdr; fetch = hdr; write = hdr
    for i, v in enumerate(valu, start=1):
        if v is None:
            sq += rows(i, other, [("SQ_WAVES", 1.0)]); fetch += rows(i, other, [("FETCH_SIZE", 1.0)]); write += rows(i, other, [("WRITE_SIZE", 1.0)])
            continue
        sq += rows(i, kern, [("SQ_WAVES", 64.0), ("SQ_INSTS_VALU", v), ("SQ_ACTIVE_INST_VALU", 10.0), ("SQ_WAVE_CYCLES", 20.0)])
        fetch += rows(i, kern, [("FETCH_SIZE", 100.0 if v < 8000 else 300.0)])
        write += rows(i, kern, [("WRITE_SIZE", 10.0)])
    for tag, text in (("sq", sq), ("fetch", fetch), ("write", write)):
        (tmp_path / ("%s_counter_collection.csv" % tag)).write_text(text)
    out = tmp_path / "x_pmc.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(tmp_path), str(out), "test"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.load(open(out))
    k0, k1 = "ibs::k_solve_gcf<double, 16> @4096x64 #0", "ibs::k_solve_gcf<double, 16> @4096x64 #1"
    assert d[k0]["SQ_WAVES"]["n"] == 3 and d[k1]["SQ_WAVES"]["n"] == 2 and d["_unmatched_launches_of_the_byte_passes"] == 0
    assert d[k0]["hbm_bytes_per_launch"] == (2 * 100.0 + 10.0) * 1024 and d[k1]["hbm_bytes_per_launch"] == (2 * 300.0 + 10.0) * 1024
    assert d[k0]["work_class"] == 0 and d[k1]["work_class"] == 1 and d[k0]["valu_spread"] < 1.01


def test_every_tool_is_in_the_tools_index():
    """tools/ holds the measurement scripts of five rounds: each one is named in tools/README.md (what it measures, which round)"""
    tdir = os.path.join(ROOT, "tools")
    text = open(os.path.join(tdir, "README.md")).read()
    missing = [f for f in sorted(os.listdir(tdir)) if os.path.isfile(os.path.join(tdir, f)) and f != "README.md" and f not in text]
    assert not missing, missing
    assert not [f for f in os.listdir(tdir) if ".bin" in f or f.endswith((".o", ".so"))], "build products in tools/"


def test_docs_state_the_limits_the_code_has():
    """the numbers a reader meets in INTEGRATION.md / include/ibs.h are the ones the library enforces (a stale '4 slots' survived two
    rounds): slots of the overlapped gather, the grid limits of the register-resident and the long-grid paths"""
    import re
    api = open(os.path.join(ROOT, "ideal-ballooning-solver_amd", "csrc", "ibs_api.hip")).read()
    launch = open(os.path.join(ROOT, "ideal-ballooning-solver_amd", "csrc", "ibs_launch.hpp")).read()
    hdr = open(os.path.join(ROOT, "include", "ibs.h")).read()
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    slots = int(re.search(r"kCommSlots = (\d+);", api).group(1))
    assert "slot in [0, %d)" % slots in hdr
    assert "up to %d gathers (slots 0 … %d)" % (slots, slots - 1) in integ
    max_m = int(re.search(r"constexpr int kMaxM = (\d+);", launch).group(1))
    max_long = int(re.search(r"constexpr int kMaxLongN = (\d+);", launch).group(1))
    n_reg = 64 * max_m + 2
    assert "Up to N = %d the register-resident kernels run" % n_reg in hdr and "%d < N <= %d" % (n_reg, max_long) in hdr
    assert "{:,}".format(max_long) in integ and "{:,}".format(n_reg) in integ
    assert "{:,}".format(max_long) in design
