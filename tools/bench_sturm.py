"""ibs_sturm_count_f64 by form (round 6): the prefix-product sweep (one wave per system: the bandwidth kernel of rounds 1-5), the
division-form count with lanes as systems, the division-form count with one wave per system -- GB/s on algorithmic bytes, and where
the counts differ from the C oracle's division-form count at shifts next to an eigenvalue.     python tools/bench_sturm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch, ibs_amd
ctx = ibs_amd.Context(0); dev = torch.device('cuda', 0)
for N in (257, 513, 1025, 2049, 4097):
    n = (1 << 18) * 513 // N
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    g = torch.exp(torch.rand((n, N), dtype=torch.float64, device=dev, generator=gen) * 3 - 1)
    c = torch.rand((n, N), dtype=torch.float64, device=dev, generator=gen) * 6 - 2.5
    f = torch.exp(torch.rand((n, N), dtype=torch.float64, device=dev, generator=gen) * 3)
    sh = torch.zeros(n, dtype=torch.float64, device=dev)
    h = 8 * np.pi / (N - 1)
    ref = None
    for form in (1, 2, 3):
        if form == 1 and N > 2050:
            continue
        if form == 3 and n > 40000:
            continue
        ctx.set_option("sturm_form", form)
        cnt = ctx.sturm_count(h, g, c, f, sh); torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in ev:
            a.record(); ctx.sturm_count(h, g, c, f, sh); b.record()
        torch.cuda.synchronize()
        ms = np.median([a.elapsed_time(b) for a, b in ev])
        byts = n * (3 * N + 1) * 8 + n * 4
        if ref is None:
            ref = cnt
        print('N=%d n=%d form %d %-28s %.3f ms  %.1f GB/s  (%.1f%% of 8 TB/s)  %.3g sweeps/s   counts differing from the first form: %d' % (
            N, n, form, ctx.last_launch()[0], ms, byts / ms / 1e6, byts / ms / 1e6 / 80, n / ms * 1e3, int((cnt != ref).sum())), flush=True)
    ctx.set_option("sturm_form", None)
