"""configs[4] as bench.py's c5_matrix leg runs it, once per eigenvalue-only FP32 form (option f32_lam: 1 = all-FP32 iteration + FP64
certificate, 2 = FP64 solver on the FP32 arrays, 0 = the library's choice)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch, ibs_amd, bench
ctx = ibs_amd.Context(0); dev = torch.device("cuda", 0)
modes = [int(a) for a in sys.argv[1:]] or [1, 2, 0]
for m in modes:
    ctx.set_option("f32_lam", m)
    r = bench.c5_matrix(ctx, dev, budget_s=600.0)
    print("f32_lam = %d: f32_gam / f64 = %s ; outside tolerance %d ; %.1f s" % (m, {k: round(v, 3) for k, v in r["f32_gam_over_f64"].items()}, r["f32_results_outside_tolerance"], r["seconds"]))
    for row in r["rows"]:
        if m == modes[0] or row.get("mode") == "f32_lam":
            print("   N_zeta %4d %-6s %-7s %.3e solves/s  %5.1f sweeps  hbm %.3f  %-44s flagged %d %s %s" % (
                row["n_zeta"], row["family"], row["mode"], row["solves_per_s"], row["mean_sweeps"], row["roofline"]["hbm_frac"], row["roofline"]["kernel"],
                row["flagged"], ("resolved %d" % row["resolved_in_f64"]) if "resolved_in_f64" in row else "",
                ("max err %.2f eps32|A| (tol %d)" % (row["max_abs_dlam_over_eps32_normA"], row["tolerance_n_eps32_normA"])) if "max_abs_dlam_over_eps32_normA" in row else ""), flush=True)
