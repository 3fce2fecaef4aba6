#!/usr/bin/env python3
"""Rates of the SURVEY.md 8(f) rows — the callers and data formats either side of the hot path — measured like the five
BASELINE configs: inputs resident in HBM, the `_dev` entry points on one stream, K timed launches between synchronisations,
a sample of the outputs compared with the CPU oracle, and the oracle's own rate on a bounded sample (one host thread).

  f1  point decompression        c12381_g1/g2_decompress_batch_dev         ECP_fromOctet / ECP2_fromOctet
  f2  BBS+ verify from wire      c12381_bbs_plus_verify_wire_batch_dev     examples/bbs-plus/src/bbs+.cpp:57-73 on serialized input
  f3  hash-to-G1, GT operators   c12381_g1_from_hash_batch_dev, c12381_gt_op_batch_dev (mul, pow), c12381_gt_is_unity_batch_dev
  f4  Zp helpers                 c12381_zp_op_batch_dev (mul, inv), c12381_zp_inner_product_dev

usage (GPU box): python3 tools/next_rows_bench.py [out.json]"""
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from crypto12381_amd import Context  # noqa: E402
from oracle.bindings import Oracle, build, have_reference  # noqa: E402
from tools.prof_driver import G1, G2  # noqa: E402

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def scalars(seed, n, reduce_=True):
    a = np.random.Generator(np.random.PCG64(seed)).integers(0, 256, size=(n, 32), dtype=np.uint8)
    if reduce_:
        a[:, 0] &= 0x3f
    return a


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "next_rows.json")
    build()
    orc = Oracle("reference" if have_reference() else "port")
    dev = torch.device("cuda", 0)
    ctx = Context(0)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    rows = {}

    def t(b):
        return torch.from_numpy(np.frombuffer(bytes(b), dtype=np.uint8).copy()).to(dev)

    def timed(fn, steps=3, warmup=1):
        for _ in range(warmup):
            fn()
        ctx.sync()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        ctx.sync()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / steps

    def cpu_rate(fn, units):
        t0 = time.perf_counter()
        fn()
        return units / (time.perf_counter() - t0)

    def row(name, entry, n, secs, unit, checked, cpu, cpu_sample, note=None):
        rows[name] = {"entry": entry, "batch": n, "ms_per_batch": secs * 1e3, "value": n / secs, "unit": unit,
                      "parity": {"checked": checked, "against": "oracle/%s" % orc.kind, "ok": True},
                      "cpu_baseline": {"value": cpu, "unit": unit, "cores": 1, "kind": orc.kind, "sample": cpu_sample},
                      "gpu_over_one_core": (n / secs) / cpu}
        if note:
            rows[name]["note"] = note
        print("%-22s %9d  %8.2f ms  %.3e %s   cpu(1 thread) %.3e   x%.0f" % (name, n, secs * 1e3, n / secs, unit, cpu, (n / secs) / cpu), flush=True)

    g1 = t(G1); g2 = t(G2)
    # ---------------------------------------------------------------- f1: decompression
    n = 1 << 20
    sc = torch.from_numpy(scalars(11, n)).to(dev)
    c49 = torch.empty(49 * n, dtype=torch.uint8, device=dev)
    ctx.g1_mul_fixed_dev(n, g1.data_ptr(), sc.data_ptr(), c49.data_ptr(), 49)
    ctx.sync()
    o96 = torch.empty(96 * n, dtype=torch.uint8, device=dev); st = torch.empty(n, dtype=torch.uint8, device=dev)
    secs = timed(lambda: ctx.g1_decompress_dev(n, c49.data_ptr(), o96.data_ptr(), st.data_ptr()))
    k = 512
    ch = c49[:49 * k].cpu().numpy().tobytes()
    exp, est = orc.g1_decompress(ch)
    assert o96[:96 * k].cpu().numpy().tobytes() == exp and st[:k].cpu().numpy().tobytes() == est and bool((st == 1).all().item())
    cpu = cpu_rate(lambda: orc.g1_decompress(c49[:49 * 8192].cpu().numpy().tobytes()), 8192)
    row("g1_decompress", "c12381_g1_decompress_batch_dev", n, secs, "points/s", k, cpu, "8192 points")

    n2 = 1 << 18
    c97 = torch.empty(97 * n2, dtype=torch.uint8, device=dev)
    ctx.g2_mul_fixed_dev(n2, g2.data_ptr(), sc.data_ptr(), c97.data_ptr(), 97)
    ctx.sync()
    o192 = torch.empty(192 * n2, dtype=torch.uint8, device=dev); st2 = torch.empty(n2, dtype=torch.uint8, device=dev)
    secs = timed(lambda: ctx.g2_decompress_dev(n2, c97.data_ptr(), o192.data_ptr(), st2.data_ptr()))
    ch = c97[:97 * k].cpu().numpy().tobytes()
    exp, est = orc.g2_decompress(ch)
    assert o192[:192 * k].cpu().numpy().tobytes() == exp and st2[:k].cpu().numpy().tobytes() == est
    cpu = cpu_rate(lambda: orc.g2_decompress(c97[:97 * 4096].cpu().numpy().tobytes()), 4096)
    row("g2_decompress", "c12381_g2_decompress_batch_dev", n2, secs, "points/s", k, cpu, "4096 points")

    # ---------------------------------------------------------------- f3: hash-to-G1
    dg = torch.from_numpy(scalars(12, 2 * n, False).reshape(-1)).to(dev)              # n digests of 64 bytes
    secs = timed(lambda: ctx.g1_from_hash_dev(n, dg.data_ptr(), o96.data_ptr(), 96))
    exp = orc.g1_from_hash(dg[:64 * k].cpu().numpy().tobytes(), 96)
    assert o96[:96 * k].cpu().numpy().tobytes() == exp
    cpu = cpu_rate(lambda: orc.g1_from_hash(dg[:64 * 4096].cpu().numpy().tobytes(), 96), 4096)
    row("g1_from_hash", "c12381_g1_from_hash_batch_dev", n, secs, "points/s", k, cpu, "4096 digests")

    # ---------------------------------------------------------------- f3: GT operators on pairing values
    ng = 1 << 16
    P = torch.empty(96 * ng, dtype=torch.uint8, device=dev); Q = torch.empty(192 * ng, dtype=torch.uint8, device=dev)
    ctx.g1_mul_fixed_dev(ng, g1.data_ptr(), sc.data_ptr(), P.data_ptr(), 96)
    ctx.g2_mul_fixed_dev(ng, g2.data_ptr(), sc[ng:].data_ptr(), Q.data_ptr(), 192)
    ga = torch.empty(576 * ng, dtype=torch.uint8, device=dev)
    go = torch.empty(576 * ng, dtype=torch.uint8, device=dev)
    ctx.pair_dev(ng, P.data_ptr(), Q.data_ptr(), ga.data_ptr())
    ctx.sync()
    gb = torch.roll(ga.view(ng, 576), 1, 0).contiguous().view(-1)                  # b_i = a_{i-1}
    secs = timed(lambda: ctx.gt_op_dev("mul", ng, ga.data_ptr(), gb.data_ptr(), go.data_ptr()))
    kk = 128
    exp = orc.gt_op("mul", ga[:576 * kk].cpu().numpy().tobytes(), gb[:576 * kk].cpu().numpy().tobytes())
    assert go[:576 * kk].cpu().numpy().tobytes() == exp
    cpu = cpu_rate(lambda: orc.gt_op("mul", ga[:576 * 4096].cpu().numpy().tobytes(), gb[:576 * 4096].cpu().numpy().tobytes()), 4096)
    row("gt_mul", "c12381_gt_op_batch_dev(op 0)", ng, secs, "products/s", kk, cpu, "4096 products")

    ex = torch.from_numpy(scalars(13, ng).reshape(-1)).to(dev)
    secs = timed(lambda: ctx.gt_op_dev("pow", ng, ga.data_ptr(), ex.data_ptr(), go.data_ptr()), 2, 1)
    exp = orc.gt_op("pow", ga[:576 * kk].cpu().numpy().tobytes(), ex[:32 * kk].cpu().numpy().tobytes())
    assert go[:576 * kk].cpu().numpy().tobytes() == exp
    cpu = cpu_rate(lambda: orc.gt_op("pow", ga[:576 * 512].cpu().numpy().tobytes(), ex[:32 * 512].cpu().numpy().tobytes()), 512)
    row("gt_pow", "c12381_gt_op_batch_dev(op 2)", ng, secs, "exponentiations/s", kk, cpu, "512 exponentiations (254-bit exponents)")

    un = torch.empty(ng, dtype=torch.uint8, device=dev)
    secs = timed(lambda: ctx.gt_is_unity_dev(ng, ga.data_ptr(), un.data_ptr()))
    assert not bool(un.any().item())
    rows["gt_is_unity"] = {"entry": "c12381_gt_is_unity_batch_dev", "batch": ng, "ms_per_batch": secs * 1e3, "value": ng / secs, "unit": "tests/s"}

    # ---------------------------------------------------------------- f4: Zp helpers
    za = torch.from_numpy(scalars(14, n).reshape(-1)).to(dev); zb = torch.from_numpy(scalars(15, n).reshape(-1)).to(dev)
    zo = torch.empty(32 * n, dtype=torch.uint8, device=dev)
    for op, sample in (("mul", 1 << 18), ("inv", 1 << 14)):
        secs = timed(lambda: ctx.zp_op_dev(op, n, za.data_ptr(), zb.data_ptr() if op == "mul" else None, zo.data_ptr()))
        a_h = za[:32 * k].cpu().numpy().tobytes(); b_h = zb[:32 * k].cpu().numpy().tobytes()
        assert zo[:32 * k].cpu().numpy().tobytes() == orc.zp_op(op, a_h, b_h if op == "mul" else None)
        A = za[:32 * sample].cpu().numpy().tobytes(); B = zb[:32 * sample].cpu().numpy().tobytes()
        cpu = cpu_rate(lambda: orc.zp_op(op, A, B if op == "mul" else None), sample)
        row("zp_" + op, "c12381_zp_op_batch_dev(%s)" % op, n, secs, "ops/s", k, cpu, "%d elements" % sample)
    nz = 1 << 22
    ia = torch.from_numpy(scalars(16, nz)).to(dev); ib = torch.from_numpy(scalars(17, nz)).to(dev)
    io = torch.empty(32, dtype=torch.uint8, device=dev)
    secs = timed(lambda: ctx.zp_inner_product_dev(nz, ia.data_ptr(), ib.data_ptr(), io.data_ptr()))
    av = ia.cpu().numpy().reshape(nz, 32); bv = ib.cpu().numpy().reshape(nz, 32)
    full = 0
    step = 1 << 12
    for i in range(0, nz, step):
        aa = [int.from_bytes(x.tobytes(), "big") for x in av[i:i + step]]
        bb = [int.from_bytes(x.tobytes(), "big") for x in bv[i:i + step]]
        full = (full + sum(x * y for x, y in zip(aa, bb))) % R
    assert int.from_bytes(io.cpu().numpy().tobytes(), "big") == full
    rows["zp_inner_product"] = {"entry": "c12381_zp_inner_product_dev", "batch": nz, "ms_per_batch": secs * 1e3, "value": nz / secs, "unit": "terms/s",
                                "parity": {"checked": nz, "against": "python integers", "ok": True}}
    print("%-22s %9d  %8.2f ms  %.3e terms/s" % ("zp_inner_product", nz, secs * 1e3, nz / secs), flush=True)

    # ---------------------------------------------------------------- f2: BBS+ verification from the wire formats
    from test_gpu_bbs import _setup                                              # the reference's setup(), restated for the tests
    nb = 1 << 16
    msg_len, nh = 12, 2
    G1p, G2p, h0, h, gamma, w = _setup(orc, nh)
    pp = orc.g1_compress(G1p) + orc.g2_compress(G2p) + orc.g1_compress(h0)
    h49 = orc.g1_compress(h); pk = orc.g2_compress(w)
    distinct = 256
    msgs_d = [hashlib.sha256(b"msg|%d" % j).digest()[:msg_len] for j in range(distinct)]
    enc = [orc.encode_to_zp(m) for m in msgs_d]
    msgs = b"".join(msgs_d[j % distinct] for j in range(nb))
    m32 = b"".join(enc[j % distinct] for j in range(nb))
    xs = scalars(18, nb); rs = scalars(19, nb)
    A96 = ctx.bbs_plus_sign(G1p, h0, h[:96], gamma.to_bytes(32, "big") if isinstance(gamma, int) else gamma, xs.tobytes(), rs.tobytes(), m32)
    one = (1).to_bytes(32, "big") * nb
    A49 = np.frombuffer(ctx.g1_mul(A96, one, 49), dtype=np.uint8).reshape(nb, 49)
    sig = np.zeros((nb, 145), dtype=np.uint8)
    sig[:, :49] = A49; sig[:, 65:97] = xs; sig[:, 113:145] = rs
    bad = np.arange(0, nb, 1009)
    sig[bad, 144] ^= 1                                                             # every 1009th signature is wrong
    dsig = torch.from_numpy(sig.reshape(-1)).to(dev); dmsg = t(msgs); dpp = t(pp); dh = t(h49); dpk = t(pk)
    ok = torch.empty(nb, dtype=torch.uint8, device=dev)
    secs = timed(lambda: ctx.bbs_plus_verify_wire_dev(nb, nh, msg_len, dpp.data_ptr(), dh.data_ptr(), dpk.data_ptr(), dsig.data_ptr(), dmsg.data_ptr(), ok.data_ptr()), 2, 1)
    okh = ok.cpu().numpy()
    expect = np.ones(nb, dtype=np.uint8); expect[bad] = 0
    assert (okh == expect).all(), "wire verdicts: %d differ" % int((okh != expect).sum())
    ks = 256
    got = orc.bbs_plus_verify_wire(pp, h49, pk, sig[:ks].tobytes(), msgs[:msg_len * ks], msg_len, 1)
    assert got == okh[:ks].tobytes()
    cpu = cpu_rate(lambda: orc.bbs_plus_verify_wire(pp, h49, pk, sig[:1024].tobytes(), msgs[:msg_len * 1024], msg_len, 1), 1024)
    row("bbs_plus_verify_wire", "c12381_bbs_plus_verify_wire_batch_dev", nb, secs, "signatures/s", ks, cpu, "1024 signatures",
        note="every verdict of the batch also checked against the construction (every 1009th signature corrupted)")

    ctx.close()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    json.dump({"device": torch.cuda.get_device_name(0), "rows": rows}, open(out_path, "w"), indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
