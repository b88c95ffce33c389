#!/usr/bin/env python3
"""One rank of the library's collective code with world > 1 on ONE GPU (test infrastructure; started by tests/test_gpu_round6.py as a
fresh process per rank, through the C ABI only: ibs_comm_load(<stand-in>) -> ibs_comm_unique_id / ibs_comm_init ->
ibs_comm_allgather_f64, ibs_comm_allgather_start_f64 over all 16 slots with then_wait_slot / host waits -> ibs_comm_wait ->
ibs_comm_destroy).  The transport is tests/cabi/fake_rccl.c (shared memory + copies on the stream it is handed, with a delay that
makes a missing wait visible).  Replaces nothing upstream: what it exercises replaces ball_scan.py:341-347.

    comm_worker.py <rank> <world> <fake_lib> <id_file> <out_json>
"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

rank, world = int(sys.argv[1]), int(sys.argv[2])
fake, id_file, out_file = sys.argv[3], sys.argv[4], sys.argv[5]

import torch                                   # noqa: E402
import ibs_amd                                 # noqa: E402
from ibs_amd._lib import check, lib            # noqa: E402

res = {"rank": rank, "ok": False, "checks": {}}
try:
    L = lib()
    check(L.ibs_comm_load(fake.encode()), "ibs_comm_load")
    ident = C.create_string_buffer(128)
    if rank == 0:
        check(L.ibs_comm_unique_id(ident), "ibs_comm_unique_id")
        with open(id_file + ".tmp", "wb") as fh:
            fh.write(ident.raw)
        os.replace(id_file + ".tmp", id_file)
    else:
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 120:
                raise RuntimeError("no unique id from rank 0")
            time.sleep(0.01)
        ident = C.create_string_buffer(open(id_file, "rb").read(), 128)
    dev = torch.device("cuda", 0)
    ctx = ibs_amd.Context(0)
    check(L.ibs_comm_init(ctx._h, ident, rank, world), "ibs_comm_init")
    ctx._comm_world = world
    ck = res["checks"]
    NS = 16                                                         # slots of the library (include/ibs.h)
    n = 24                                                          # doubles per rank and gather: 8 surfaces x (theta0*, alpha*, gam)

    def payload(r, t):                                              # what rank r sends in gather number t
        return (1000.0 * (r + 1) + t) + 0.001 * torch.arange(n, dtype=torch.float64, device=dev)

    def expected(t):
        return torch.cat([payload(r, t) for r in range(world)])

    # (a) in-stream gather, ordered after the kernel that produces `send` on the same stream
    send = torch.empty(n, dtype=torch.float64, device=dev); recv = torch.full((world * n,), -1.0, dtype=torch.float64, device=dev)
    big = torch.ones(1 << 22, dtype=torch.float64, device=dev)
    for t in range(5):
        send.copy_(payload(rank, t) + 0.0 * big.sum())              # a long producer: the gather must wait for it
        ctx.allgather(send, recv)
        got = recv.clone()                                          # consumer on the same stream
        torch.cuda.synchronize()
        ck["in_stream_%d" % t] = bool(torch.equal(got, expected(t)))
    # (b) control: the transport really is asynchronous -- without ibs_comm_wait a consumer on the context's stream sees the
    # buffer's previous contents (otherwise (c) below could not detect a missing wait)
    sends = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(NS)]
    recvs = [torch.full((world * n,), -7.0, dtype=torch.float64, device=dev) for _ in range(NS)]
    sends[0].copy_(payload(rank, 100))
    ctx.allgather_start(sends[0], recvs[0], slot=0)
    early = recvs[0].clone()
    ctx.comm_wait(0)
    late = recvs[0].clone()
    torch.cuda.synchronize()
    ck["control_without_wait_sees_old_contents"] = bool((early == -7.0).all())
    ck["after_wait"] = bool(torch.equal(late, expected(100)))
    # (c) the 16 slots, running NS - 1 gathers ahead: step t starts the gather of slot t % NS and, in the same call, orders the
    # context's stream after the gather that used the NEXT slot NS - 1 steps ago (then_wait_slot); the consumer then reads it
    T = 3 * NS + 5
    seen = []
    for t in range(T):
        s = t % NS
        sends[s].copy_(payload(rank, 200 + t) + 0.0 * big[: 1 << 16].sum())
        nxt = (t + 1) % NS
        ctx.allgather_start(sends[s], recvs[s], slot=s, then_wait=nxt if t >= NS - 1 else -1)
        if t >= NS - 1:
            seen.append((t - (NS - 1), recvs[nxt].clone()))       # gather number t - 15 lives in slot (t + 1) % 16
    ctx.comm_wait(-1)
    tail = [(t, recvs[t % NS].clone()) for t in range(T - (NS - 1), T)]
    torch.cuda.synchronize()
    ck["slots_then_wait"] = all(bool(torch.equal(v, expected(200 + t))) for t, v in seen)
    ck["slots_wait_all"] = all(bool(torch.equal(v, expected(200 + t))) for t, v in tail)
    ck["slots_gathers_checked"] = len(seen) + len(tail)
    # (d) host-side wait (then_wait_slot = -2 - s): no wait on the context's stream; after the call returns the slot's gather is
    # complete, whatever the stream is doing
    for t in range(2 * NS):
        s = t % NS
        sends[s].copy_(payload(rank, 400 + t))
        hw = (t + 1) % NS if t >= NS - 1 else None
        ctx.allgather_start(sends[s], recvs[s], slot=s, host_wait=hw)
        if hw is not None:
            # the gather is complete on the HOST's clock: a copy on a DIFFERENT stream must already see it
            with torch.cuda.stream(torch.cuda.Stream(dev)):
                side = recvs[hw].clone()
            torch.cuda.synchronize()
            ck["host_wait_%d" % t] = bool(torch.equal(side, expected(400 + t - (NS - 1))))
    ctx.comm_wait(-1)
    torch.cuda.synchronize()
    # (e) argument checks of the slot interface
    rc = L.ibs_comm_allgather_start_f64(ctx._h, C.c_void_p(sends[0].data_ptr()), C.c_void_p(recvs[0].data_ptr()), n, NS, -1)
    ck["slot_16_refused"] = rc < 0
    rc = L.ibs_comm_allgather_start_f64(ctx._h, C.c_void_p(sends[0].data_ptr()), C.c_void_p(recvs[0].data_ptr()), n, 3, 3)
    ck["wait_on_own_slot_refused"] = rc < 0
    # (f) the product's gather of per-surface rows through the library's communicator (gather_rows_tensor, uneven shards)
    n_surf = 4 * world + 1
    own = ibs_amd.shard_surfaces(n_surf, rank, world)
    local = torch.tensor([[j, 10.0 * j, 100.0 * j] for j in own], dtype=torch.float64, device=dev).reshape(len(own), 3)
    full = ibs_amd.gather_rows_tensor(local, n_surf, rank, world, None, ctx)
    torch.cuda.synchronize()
    want = torch.tensor([[j, 10.0 * j, 100.0 * j] for j in range(n_surf)], dtype=torch.float64, device=dev)
    ck["gather_rows_tensor_native"] = bool(torch.equal(full, want))
    ctx.comm_destroy()
    res["ok"] = all(v is True or (k == "slots_gathers_checked" and v > 0) for k, v in ck.items())
except Exception as e:                          # (reported through the file: the parent compares the ranks)
    res["error"] = "%s: %s" % (type(e).__name__, e)
with open(out_file, "w") as fh:
    json.dump(res, fh)
sys.exit(0 if res["ok"] else 1)
