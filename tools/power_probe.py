import os, sys, time, subprocess, threading
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ibs_amd
dev = torch.device("cuda:0"); ctx = ibs_amd.Context(0)
wout = dict(np.load("tests/golden/G8_wout_ncsx_op.npz"))
svals = np.linspace(0.1, 0.95, 64)
tabs = ibs_amd.SurfaceTables.from_wout(wout, svals)
th = torch.from_numpy(ibs_amd.theta_grid(1025)).to(dev)
surf = torch.from_numpy(np.repeat(np.arange(64), 32).astype(np.int32)).to(dev)
al = torch.from_numpy(np.tile(np.linspace(0, np.pi, 32), 64)).to(dev)
samples = []
stop = [False]
def sampler():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), out))
        except Exception as e:
            samples.append((time.perf_counter(), "ERR %s" % e))
        time.sleep(0.05)
t = threading.Thread(target=sampler); t.start()
time.sleep(0.5)
t0 = time.perf_counter()
times = []
for k in range(40):
    torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(25): r = ctx.fieldline_geometry(tabs, surf, al, th, device=dev)
    torch.cuda.synchronize(); times.append((time.perf_counter() - t0, (time.perf_counter() - a) / 25 * 1e3))
stop[0] = True; t.join()
print("ms per call over time:", ["%.2fs:%.3f" % x for x in times[::4]])
import json
for ts, out in samples[::6]:
    try:
        d = json.loads(out); c = d[list(d.keys())[0]]
        print("%.2fs" % (ts - t0), {k: v for k, v in c.items() if "ower" in k or "sclk" in k.lower()})
    except Exception as e:
        print("%.2fs" % (ts - t0), out[:200].replace("\n", " "))
