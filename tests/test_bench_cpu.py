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
    table = {"_set": "test",
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
    """profiles/pmc_current.json as committed: the headline kernel's entry is found by exact name at 1,024 waves, and its
    traffic is within 2x of the algorithmic 3.69 MB (the round-3 line quoted 7.3 MB from another kernel)."""
    import bench
    if hasattr(bench._pmc_file, "cache"):
        del bench._pmc_file.cache
    e, why = bench.pmc_entry("ibs::k_gamma_scan<double, 8>", 1024)
    assert why is None, why
    assert 3.6e6 < e["hbm_bytes_per_launch"] < 5.5e6
