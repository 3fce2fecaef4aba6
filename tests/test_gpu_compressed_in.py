"""C12381_F_COMPRESSED_IN: the batch entry points on serialized points — from_bytes -> multiply / pair / product in one call
(the reference: g1_point.hpp:87-111, g2_point.hpp:73-77 in front of from_bytes(point1& / point2&, bytes_view&), src/miracl_core_interface.cpp:
109-112, 187-190 -> ECP_fromOctet ecp_BLS12381.cpp:495-545, ECP2_fromOctet ecp2_BLS12381.cpp:225-266), against the compiled reference's
from_bytes followed by multiply / pair_ate + pair_final_exponentiation on the same bytes."""
import pytest

from util import P, R, cat, golden, scalars

pytestmark = pytest.mark.gpu


def _encodings(oracle_ref, n):
    """n G1 and n G2 encodings: valid ones, then infinity (leading 0x00, arbitrary tail), x + p (accepted: x is taken mod p), and the rejects —
    tag 0x04 in the short form, an unknown tag (G1 only: G2 accepts any tag but 0x04), an x with no point above it"""
    g1, g2 = oracle_ref.g1_generator(), oracle_ref.g2_generator()
    one = b"".join((1).to_bytes(32, "big") for _ in range(n))
    p96 = oracle_ref.g1_mul(g1 * n, scalars(1201, n), 96, 4)
    q192 = oracle_ref.g2_mul(g2 * n, scalars(1202, n), 192, 4)
    c1 = bytearray(oracle_ref.g1_mul(p96, one, 49, 4))
    c2 = bytearray(oracle_ref.g2_mul(q192, one, 97, 4))
    c1[0:49] = b"\x00" + bytes(range(1, 49))                                   # infinity
    c2[0:97] = b"\x00" + bytes(range(1, 97))
    x = int.from_bytes(c1[49 + 1:49 + 49], "big")
    c1[49 + 1:49 + 49] = (x + P).to_bytes(48, "big")                           # lane 1: x + p, same point
    xa = int.from_bytes(c2[97 + 49:97 + 97], "big")
    c2[97 + 49:97 + 97] = (xa + P).to_bytes(48, "big")
    c1[2 * 49] = 0x04                                                          # lane 2: rejected tag
    c2[2 * 97] = 0x04
    c1[3 * 49] = 0x07                                                          # lane 3: unknown tag (G1 rejects; for G2 "compressed, sign 1")
    c2[3 * 97] = 0x07
    # lane 4: an x without a point: search from the encoded x upwards with the reference's decoder
    for off, buf, ln, dec in ((4 * 49, c1, 49, oracle_ref.g1_decompress), (4 * 97, c2, 97, oracle_ref.g2_decompress)):
        v = int.from_bytes(buf[off + ln - 48:off + ln], "big")
        while True:
            v += 1
            cand = bytes(buf[off:off + ln - 48]) + v.to_bytes(48, "big")
            if dec(cand)[1][0] == 0:
                buf[off:off + ln] = cand
                break
    return bytes(c1), bytes(c2)


def _ref_status(oracle_ref, c1, c2):
    d1, s1 = oracle_ref.g1_decompress(c1)
    d2, s2 = oracle_ref.g2_decompress(c2)
    return d1, list(s1), d2, list(s2)


def test_compressed_inputs_vs_reference(oracle_ref):
    from crypto12381_amd import Context
    from crypto12381_amd.capi import C12381Error, E_ARG, E_POINT, F_COMPRESSED_IN, F_IN_SUBGROUP
    n = 40
    c1, c2 = _encodings(oracle_ref, n)
    d1, s1, d2, s2 = _ref_status(oracle_ref, c1, c2)
    assert s1[:5] == [1, 1, 0, 0, 0] and s2[:5] == [1, 1, 0, 1, 0] and all(s1[5:]) and all(s2[5:])
    sc = scalars(1203, n, 1 << 256)
    ctx = Context(0)
    # ---- G1: from_bytes -> multiply -> to_bytes
    for fmt in (96, 49):
        got = ctx.g1_mul_flags(c1, sc, fmt, F_COMPRESSED_IN, strict=False)
        exp = oracle_ref.g1_mul(d1, sc, fmt, 4)
        for i in range(n):
            assert got[fmt * i:fmt * i + fmt] == (exp[fmt * i:fmt * i + fmt] if s1[i] else b"\xff" * fmt), ("g1", fmt, i)
    with pytest.raises(C12381Error) as e:                                      # the status is reported
        ctx.g1_mul_flags(c1, sc, 96, F_COMPRESSED_IN)
    assert e.value.code == E_POINT
    ok1 = [i for i in range(n) if s1[i]]
    sel = lambda b, w, idx: b"".join(b[w * i:w * i + w] for i in idx)
    assert ctx.g1_mul_flags(sel(c1, 49, ok1), sel(sc, 32, ok1), 49, F_COMPRESSED_IN) == oracle_ref.g1_mul(sel(d1, 96, ok1), sel(sc, 32, ok1), 49, 4)
    assert ctx.g1_mul_flags(sel(c1, 49, ok1), sel(sc, 32, ok1), 96, F_COMPRESSED_IN | F_IN_SUBGROUP) == oracle_ref.g1_mul(sel(d1, 96, ok1), sel(sc, 32, ok1), 96, 4)
    # ---- G2
    for fmt in (192, 97):
        got = ctx.g2_mul_flags(c2, sc, fmt, F_COMPRESSED_IN, strict=False)
        exp = oracle_ref.g2_mul(d2, sc, fmt, 4)
        for i in range(n):
            assert got[fmt * i:fmt * i + fmt] == (exp[fmt * i:fmt * i + fmt] if s2[i] else b"\xff" * fmt), ("g2", fmt, i)
    # ---- product of powers: the accepted terms as the reference evaluates PI (multiply + add chain); a rejected term is reported
    assert ctx.g1_msm_flags(sel(c1, 49, ok1), sel(sc, 32, ok1), 49, F_COMPRESSED_IN) == oracle_ref.g1_msm(sel(d1, 96, ok1), sel(sc, 32, ok1), 49, 4)
    assert ctx.g1_msm_flags(c1[:49 * 2], sc[:64], 96, F_COMPRESSED_IN) == oracle_ref.g1_msm(d1[:192], sc[:64], 96, 1)          # 2 terms, one at infinity
    assert ctx.g1_msm_flags(c1[49:98], sc[32:64], 96, F_COMPRESSED_IN) == oracle_ref.g1_mul(d1[96:192], sc[32:64], 96, 1)       # a single term
    with pytest.raises(C12381Error) as e:
        ctx.g1_msm_flags(c1, sc, 49, F_COMPRESSED_IN)
    assert e.value.code == E_POINT
    assert ctx.g1_msm_flags(c1, sc, 49, F_COMPRESSED_IN, strict=False) == oracle_ref.g1_msm(sel(d1, 96, ok1), sel(sc, 32, ok1), 49, 4)
    # ---- pairing: from_bytes x 2 -> pair_ate -> pair_final_exponentiation -> to_bytes
    got = ctx.pair_flags(c1, c2, F_COMPRESSED_IN, strict=False)
    exp = oracle_ref.pair(d1, d2, 4)
    for i in range(n):
        assert got[576 * i:576 * i + 576] == (exp[576 * i:576 * i + 576] if (s1[i] and s2[i]) else b"\xff" * 576), ("pair", i)
    both = [i for i in range(n) if s1[i] and s2[i]]
    assert ctx.pair_flags(sel(c1, 49, both), sel(c2, 97, both), F_COMPRESSED_IN) == oracle_ref.pair(sel(d1, 96, both), sel(d2, 192, both), 4)
    # without the flag the same entry points are the 96 / 192-byte forms
    assert ctx.pair_flags(sel(d1, 96, both), sel(d2, 192, both), 0) == ctx.pair(sel(d1, 96, both), sel(d2, 192, both))
    # an unknown flag is an argument error
    with pytest.raises(C12381Error) as e:
        ctx.pair_flags(c1, c2, 64)
    assert e.value.code == E_ARG
    ctx.close()


def test_compressed_inputs_device_pointers_full_chunks(oracle_ref):
    """_dev forms on a batch that spans two scalar-multiplication launches (2^20 + 77 G1 elements: a launch takes 2^20) and the
    bucket MSM at 2^14 compressed terms, sampled against the reference"""
    import torch
    from crypto12381_amd import Context
    from crypto12381_amd.capi import F_COMPRESSED_IN
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    g1 = oracle_ref.g1_generator()
    m = 1 << 10
    base = ctx.g1_mul(g1 * m, scalars(1211, m), 49)
    n = (1 << 20) + 77
    c1 = (base * (n // m + 1))[:49 * n]
    sc = scalars(1212, 512, 1 << 256) * (n // 512 + 1)
    sc = sc[:32 * n]
    dc, ds = torch.frombuffer(bytearray(c1), dtype=torch.uint8).to(dev), torch.frombuffer(bytearray(sc), dtype=torch.uint8).to(dev)
    out = torch.empty(96 * n, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(device=dev)
    ctx.set_stream(s.cuda_stream)
    ctx.g1_mul_flags_dev(n, dc.data_ptr(), ds.data_ptr(), out.data_ptr(), 96, F_COMPRESSED_IN)
    assert ctx.sync() == 0
    got = out.cpu().numpy().tobytes()
    idx = [0, 1, 2, 511, 512, 1023, 1024, (1 << 17) - 1, 1 << 17, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, n - 1]
    pts96 = oracle_ref.g1_decompress(b"".join(c1[49 * i:49 * i + 49] for i in idx))[0]
    assert b"".join(got[96 * i:96 * i + 96] for i in idx) == oracle_ref.g1_mul(pts96, b"".join(sc[32 * i:32 * i + 32] for i in idx), 96, 4)
    nm = 1 << 14
    o1 = torch.empty(96, dtype=torch.uint8, device=dev)
    ctx.lib.c12381_g1_msm_flags_dev(ctx.h, nm, dc.data_ptr(), ds.data_ptr(), o1.data_ptr(), 96, F_COMPRESSED_IN)
    assert ctx.sync() == 0
    d96 = ctx.g1_decompress(c1[:49 * nm])[0]
    assert o1.cpu().numpy().tobytes() == ctx.g1_msm(d96, sc[:32 * nm], 96)
    ctx.close()


def test_g2_batch_spanning_two_launches(oracle_ref):
    """2^19 + 33 G2 multiplications through the _dev entry (a launch takes 2^19 points): lanes either side of the launch boundary and
    at both ends against the reference"""
    import torch
    from crypto12381_amd import Context
    ctx = Context(0)
    dev = torch.device("cuda", 0)
    g2 = oracle_ref.g2_generator()
    m = 1 << 10
    base = ctx.g2_mul(g2 * m, scalars(1311, m), 192)
    n = (1 << 19) + 33
    pts = (base * (n // m + 1))[:192 * n]
    sc = (scalars(1312, 512, 1 << 256) * (n // 512 + 1))[:32 * n]
    dp, ds = torch.frombuffer(bytearray(pts), dtype=torch.uint8).to(dev), torch.frombuffer(bytearray(sc), dtype=torch.uint8).to(dev)
    out = torch.empty(192 * n, dtype=torch.uint8, device=dev)
    s = torch.cuda.Stream(device=dev)
    ctx.set_stream(s.cuda_stream)
    ctx.g2_mul_dev(n, dp.data_ptr(), ds.data_ptr(), out.data_ptr(), 192)
    assert ctx.sync() == 0
    got = out.cpu().numpy().tobytes()
    idx = [0, 1, 63, 64, (1 << 17) - 1, 1 << 17, (1 << 19) - 2, (1 << 19) - 1, 1 << 19, (1 << 19) + 1, n - 2, n - 1]
    want = oracle_ref.g2_mul(b"".join(pts[192 * i:192 * i + 192] for i in idx), b"".join(sc[32 * i:32 * i + 32] for i in idx), 192, 4)
    assert b"".join(got[192 * i:192 * i + 192] for i in idx) == want
    ctx.close()
