#!/usr/bin/env python3
"""Round 6: where the division-form Sturm count with lanes as systems (k_sturm_count_div, 64 systems per wave) passes the prefix-product
sweep (k_sturm_count, one wave per system) -- grid length x batch size.     python tools/experiments/sturm_crossover.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device('cuda', 0)
for N in (513, 641, 769, 1025, 1537, 2049):
    for n in (8192, 16384, 32768, 65536, 131072, 262144):
        if n * N > (1 << 28):
            continue
        gen = torch.Generator(device=dev); gen.manual_seed(1)
        g = torch.exp(torch.rand((n, N), dtype=torch.float64, device=dev, generator=gen) * 3 - 1)
        c = torch.rand((n, N), dtype=torch.float64, device=dev, generator=gen) * 6 - 2.5
        f = torch.exp(torch.rand((n, N), dtype=torch.float64, device=dev, generator=gen) * 3)
        sh = torch.zeros(n, dtype=torch.float64, device=dev)
        h = 8 * np.pi / (N - 1)
        ms = {}
        for form in (1, 2):
            ctx.set_option("sturm_form", form)
            ctx.sturm_count(h, g, c, f, sh); torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
            for a, b in ev:
                a.record(); ctx.sturm_count(h, g, c, f, sh); b.record()
            torch.cuda.synchronize()
            ms[form] = float(np.median([a.elapsed_time(b) for a, b in ev]))
        ctx.set_option("sturm_form", None)
        byts = n * (3 * N + 1) * 8 + n * 4
        print("N=%5d n=%7d  sweep %.3f ms %.0f GB/s | division form, lanes as systems %.3f ms %.0f GB/s  (x %.2f)" % (
            N, n, ms[1], byts / ms[1] / 1e6, ms[2], byts / ms[2] / 1e6, ms[1] / ms[2]), flush=True)
        del g, c, f
