#!/usr/bin/env python3
"""Soak run of the work-queue pairing kernels: many launches of odd sizes back to back, results compared with the first
launch of the same inputs (a scheduling-dependent error or a stuck spin would show up as a mismatch or a timeout)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402
from tools.prof_driver import G1, G2, sc  # noqa: E402

c = Context(0)
dev = torch.device("cuda", 0)
s = torch.cuda.Stream(device=dev)
c.set_stream(s.cuda_stream)
base = 1024
p = c.g1_mul(G1 * base, sc(3, base), 96)
q = c.g2_mul(G2 * base, sc(4, base), 192)
t_start = time.time()
launches = 0
for n in (43009, 50000, 65536, 65541, 100003, 131072, 262144):
    rep = (n + base - 1) // base
    dp = torch.frombuffer(bytearray((p * rep)[:96 * n]), dtype=torch.uint8).to(dev)
    dq = torch.frombuffer(bytearray((q * rep)[:192 * n]), dtype=torch.uint8).to(dev)
    ref = torch.empty(576 * n, dtype=torch.uint8, device=dev)
    out = torch.empty(576 * n, dtype=torch.uint8, device=dev)
    ok0 = torch.empty(n, dtype=torch.uint8, device=dev)
    ok1 = torch.empty(n, dtype=torch.uint8, device=dev)
    c.pair_dev(n, dp.data_ptr(), dq.data_ptr(), ref.data_ptr()); c.sync()
    c.pair_eq_dev(n, dp.data_ptr(), dq.data_ptr(), dp.data_ptr(), dq.data_ptr(), ok0.data_ptr()); c.sync()
    assert bool((ok0 == 1).all())
    for it in range(12):
        c.pair_dev(n, dp.data_ptr(), dq.data_ptr(), out.data_ptr())
        c.pair_eq_dev(n, dp.data_ptr(), dq.data_ptr(), dp.data_ptr(), dq.data_ptr(), ok1.data_ptr())
        assert c.sync() == 0
        assert torch.equal(out, ref) and torch.equal(ok0, ok1), (n, it)
        launches += 2
    # the split forms through the same queue (miller3_queue_kernel / fexp3_queue_kernel): fexp(miller(P, Q)) == pair(P, Q), repeatedly
    mil = torch.empty(576 * n, dtype=torch.uint8, device=dev)
    for it in range(4):
        c.miller_dev(n, dp.data_ptr(), dq.data_ptr(), mil.data_ptr())
        c.gt_op_dev("fexp", n, mil.data_ptr(), None, out.data_ptr())
        assert c.sync() == 0
        assert torch.equal(out, ref), ("split", n, it)
        launches += 2
    # first row against the small plain-kernel launch
    small = torch.empty(576 * 21, dtype=torch.uint8, device=dev)
    c.pair_dev(21, dp.data_ptr(), dq.data_ptr(), small.data_ptr()); c.sync()
    assert torch.equal(small, ref[:576 * 21])
    print("n=%7d ok  (%.1f s elapsed)" % (n, time.time() - t_start), flush=True)
print("soak ok: %d queue-kernel launches" % launches)

# the table-driven kernels (one G2 argument for the batch, normalised line tables): repeated launches at odd sizes, every output
# equal to the generic pairing of the same inputs; then the same with the G2 argument at infinity (raw tables)
for qfix, label in ((q[:192], "point"), (bytes(192), "infinity")):
    dq1 = torch.frombuffer(bytearray(qfix), dtype=torch.uint8).to(dev)
    for n in (43009, 65541, 131072):
        rep = (n + base - 1) // base
        dp = torch.frombuffer(bytearray((p * rep)[:96 * n]), dtype=torch.uint8).to(dev)
        dqn = dq1.repeat(n)
        ref = torch.empty(576 * n, dtype=torch.uint8, device=dev)
        out = torch.empty(576 * n, dtype=torch.uint8, device=dev)
        c.pair_dev(n, dp.data_ptr(), dqn.data_ptr(), ref.data_ptr()); c.sync()
        for it in range(6):
            c.pair_fixed_g2_dev(n, dp.data_ptr(), dq1.data_ptr(), out.data_ptr())
            assert c.sync() == 0
            assert torch.equal(out, ref), (label, n, it)
            launches += 1
        print("fixed G2 %-8s n=%7d ok  (%.1f s elapsed)" % (label, n, time.time() - t_start), flush=True)
print("soak ok incl. table-driven kernels: %d launches" % launches)

# two contexts driving the same GPU at once: each queue grid is only partly resident, tasks still only ever wait for
# tasks already claimed by resident wavefronts
import threading  # noqa: E402

c2 = Context(0)
n = 65536
rep = n // base
hp, hq = (p * rep)[:96 * n], (q * rep)[:192 * n]
want = c.pair(hp[:96 * 2100 * 21 // 21], hq[:192 * 2100 * 21 // 21])[:576]
res = {}


def run(ctx, tag):
    for _ in range(4):
        res[tag] = ctx.pair(hp, hq)


ths = [threading.Thread(target=run, args=(c, "a")), threading.Thread(target=run, args=(c2, "b"))]
t0 = time.time()
for t in ths:
    t.start()
for t in ths:
    t.join()
assert res["a"] == res["b"] and res["a"][:576] == want
print("two concurrent contexts ok (%.1f s)" % (time.time() - t0))

# streamed launches: two contexts on two streams, no host wait between the launches, DIFFERENT queue kernels and odd sizes overlapping on
# the device (the next launch's wavefronts enter as the previous one's leave): every output equals the serial result of the same inputs
cs = []
for ctx in (c, c2):
    st = torch.cuda.Stream(device=dev)
    ctx.set_stream(st.cuda_stream)
    cs.append((ctx, st))
sizes = (43009, 65536, 100003)
data = {}
for n in sizes:
    rep = (n + base - 1) // base
    dp = torch.frombuffer(bytearray((p * rep)[:96 * n]), dtype=torch.uint8).to(dev)
    dq = torch.frombuffer(bytearray((q * rep)[:192 * n]), dtype=torch.uint8).to(dev)
    ref = torch.empty(576 * n, dtype=torch.uint8, device=dev)
    mref = torch.empty(576 * n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    c.pair_dev(n, dp.data_ptr(), dq.data_ptr(), ref.data_ptr())
    c.miller_dev(n, dp.data_ptr(), dq.data_ptr(), mref.data_ptr())
    assert c.sync() == 0
    data[n] = (dp, dq, ref, mref)
t0 = time.time()
streamed = 0
for rnd in range(40):
    outs = []
    bufs = [torch.zeros(576 * sizes[(i + rnd) % 3], dtype=torch.uint8, device=dev) for i in range(12)]
    torch.cuda.synchronize(dev)                    # the zero fills run on torch's stream: complete before the library's streams write
    for i in range(12):
        n = sizes[(i + rnd) % 3]
        dp, dq, ref, mref = data[n]
        ctx = cs[i & 1][0]
        o = bufs[i]
        kind = (i // 2 + rnd) % 3
        if kind == 0:
            ctx.pair_dev(n, dp.data_ptr(), dq.data_ptr(), o.data_ptr()); want_t = ref
        elif kind == 1:
            ctx.miller_dev(n, dp.data_ptr(), dq.data_ptr(), o.data_ptr()); want_t = mref
        else:
            ctx.gt_op_dev("fexp", n, mref.data_ptr(), None, o.data_ptr()); want_t = ref
        outs.append((o, want_t, n, kind))
        streamed += 1
    for ctx, _ in cs:
        assert ctx.sync() == 0
    for o, want_t, n, kind in outs:
        assert torch.equal(o, want_t), ("streamed", rnd, n, kind)
print("streamed over two contexts ok: %d overlapping queue launches of three kinds and sizes (%.1f s)" % (streamed, time.time() - t0))
