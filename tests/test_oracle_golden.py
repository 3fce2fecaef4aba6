"""CPU tests: pin the oracle (our C restatement, and the compiled reference when present)
to the golden vectors generated from the reference (oracle/gen_golden.py)."""
import hashlib

import pytest

from util import P, R, cat, golden


@pytest.fixture(params=["port", "reference"])
def orc(request, oracle_port):
    if request.param == "port":
        return oracle_port
    return request.getfixturevalue("oracle_ref")


def test_anchor_values(orc):
    # SURVEY.md §8(c) anchors captured from the compiled reference
    g1 = orc.g1_generator()
    assert orc.g1_compress(g1).hex() == ("0317f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
                                         "6c55e83ff97a1aeffb3af00adb22c6bb")
    g2c = orc.g2_compress(orc.g2_generator())
    assert hashlib.sha256(g2c).hexdigest() == "db334b743fa2679e10f00105e29074a4a3a34fda7e4d3e5cbc790e145c07df68"
    gt = orc.pair(g1, orc.g2_generator())
    assert gt.hex().startswith("0f41e58663bf08cf068672cbd01a7ec7")
    assert hashlib.sha256(gt).hexdigest() == "8d47ab0b1283a6346b774d68d578b1c29b8e9e922e1105eca27a7126b4a6676b"


def test_fp_golden(orc):
    g = golden("fp")
    a, b = cat(g["a"]), cat(g["b"])
    for op in ("mul", "add", "sub", "sqr", "neg", "inv"):
        out, _ = orc.fp_op(op, a, b)
        assert out == cat(g[op]), op
    out, ok = orc.fp_op("sqrt", a, b)
    assert list(ok) == g["sqrt_is_qr"]
    for i, q in enumerate(ok):
        if q:
            s = int.from_bytes(out[48 * i:48 * i + 48], "big")
            assert (s * s - int(g["a"][i], 16)) % P == 0


def test_fp_against_python_ints(orc):
    g = golden("fp")
    a, b = cat(g["a"]), cat(g["b"])
    out, _ = orc.fp_op("mul", a, b)
    for i in range(len(g["a"])):
        assert int.from_bytes(out[48 * i:48 * i + 48], "big") == int(g["a"][i], 16) * int(g["b"][i], 16) % P


def test_g1_golden(orc):
    g = golden("g1")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    assert orc.g1_mul(pts, sc, 49) == cat(g["mul49"])
    assert orc.g1_mul(pts, sc, 96, nthreads=3) == cat(g["mul96"])
    assert orc.g1_add(cat(g["add_a"]), cat(g["add_b"]), 96) == cat(g["add96"])
    assert orc.g1_add(cat(g["add_a"]), cat(g["add_b"]), 49) == cat(g["add49"])
    dec, st = orc.g1_decompress(cat(g["compressed"]))
    assert list(st) == g["decompress_status"]
    assert dec == cat(g["decompressed"])
    assert orc.g1_msm(pts, sc, 49, 1).hex() == g["msm49"]
    assert orc.g1_msm(pts, sc, 49, 5).hex() == g["msm49"]
    assert orc.g1_generator().hex() == g["generator"]


def test_points_outside_the_subgroup(orc):
    """The reference never checks subgroup membership; its endomorphism-based multiplications are then a different
    (well-defined) function of the input, which the oracle — and the GPU path — must reproduce."""
    g = golden("g1")
    pts, sc = cat(g["offsubgroup_points"]), cat(g["offsubgroup_scalars"])
    assert orc.g1_mul(pts, sc, 96) == cat(g["offsubgroup_mul96"])
    assert orc.g1_msm(pts, sc, 49, 2).hex() == g["offsubgroup_msm49"]
    g = golden("g2")
    pts, sc = cat(g["offsubgroup_points"]), cat(g["offsubgroup_scalars"])
    assert orc.g2_mul(pts, sc, 192) == cat(g["offsubgroup_mul192"])
    assert orc.g2_mul(cat(g["offsubgroup_small_points"]), cat(g["offsubgroup_small_scalars"]), 192) == cat(g["offsubgroup_small_mul192"])
    g = golden("g1")
    assert orc.g1_mul(cat(g["offsubgroup_small_points"]), cat(g["offsubgroup_small_scalars"]), 96) == cat(g["offsubgroup_small_mul96"])


def test_g1_edge_semantics(orc):
    """Edge cases the reference's unit tests pin as laws (unit-tests/g1_point.cpp:51-78)."""
    g = golden("g1")
    p0 = bytes.fromhex(g["points"][0])
    inf = bytes(96)
    k = lambda v: (v % (1 << 256)).to_bytes(32, "big")
    assert orc.g1_mul(p0, k(0), 96) == inf                      # scalar 0 -> identity
    assert orc.g1_mul(p0, k(R), 96) == inf                      # scalar = r -> identity
    assert orc.g1_mul(p0, k(1), 96) == p0
    assert orc.g1_mul(p0, k(R + 1), 96) == p0                   # reduced mod r first
    assert orc.g1_mul(inf, k(12345), 96) == inf                 # infinity in -> infinity out
    m = orc.g1_mul(p0, k(R - 1), 96)                            # -P
    assert orc.g1_add(p0, m, 96) == inf
    assert orc.g1_compress(inf) == bytes(49)


def test_g2_golden(orc):
    g = golden("g2")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    assert orc.g2_mul(pts, sc, 97) == cat(g["mul97"])
    assert orc.g2_mul(pts, sc, 192, nthreads=2) == cat(g["mul192"])
    assert orc.g2_add(cat(g["add_a"]), cat(g["add_b"]), 192) == cat(g["add192"])
    dec, st = orc.g2_decompress(cat(g["compressed"]))
    assert list(st) == g["decompress_status"]
    assert dec == cat(g["decompressed"])
    assert orc.g2_generator().hex() == g["generator"]


def test_pairing_golden(orc):
    g = golden("pairing")
    g1, g2 = cat(g["g1"]), cat(g["g2"])
    gt = orc.pair(g1, g2, nthreads=2)
    assert gt == cat(g["gt"])
    one = bytes(575) + b"\x01"
    # infinity in either argument gives the identity of GT (unit-tests/liner_pair.cpp:28-40)
    assert gt[576 * 6:576 * 7] != gt[576 * 5:576 * 6]
    assert orc.pair(bytes(96), g2[:192]) == orc.pair(g1[:96], bytes(192))
    a1, a2, b1, b2 = g1[:96 * 4], g2[:192 * 4], g1[96 * 4:], g2[192 * 4:]
    assert orc.pair2(a1, a2, b1, b2) == cat(g["pair2"])
    assert list(orc.pair_eq(cat(g["eq_a1"]), cat(g["eq_a2"]), cat(g["eq_b1"]), cat(g["eq_b2"]))) == g["eq"]
    assert list(orc.pair_eq(cat(g["eq2_a1"]), cat(g["eq2_a2"]), cat(g["eq2_b1"]), cat(g["eq2_b2"]))) == g["eq2"]
    gta, gtb = gt[:576 * 4], gt[576 * 4:]
    assert orc.gt_op("mul", gta, gtb) == cat(g["gt_mul"])
    assert orc.gt_op("conj", gta) == cat(g["gt_conj"])
    assert orc.gt_op("pow", gta, cat(g["gt_pow_exp"])) == cat(g["gt_pow"])
    del one


def test_config1_bilinearity(orc):
    """BASELINE.json configs[0]: pair(g1^x, g2^y) == pair(g1, g2)^(xy) on the reference's own seed."""
    g = golden("config1_bilinearity")
    sc = cat(g["scalars"])
    x, y = sc[64:96], sc[96:128]
    Pp, Qq = bytes.fromhex(g["P"]), bytes.fromhex(g["Q"])
    lhs = orc.pair(orc.g1_mul(Pp, x, 96), orc.g2_mul(Qq, y, 192))
    assert lhs.hex() == g["pair_Px_Qy"]
    xy = ((int.from_bytes(x, "big") * int.from_bytes(y, "big")) % R).to_bytes(32, "big")
    assert xy.hex() == g["xy"]
    assert orc.gt_op("pow", orc.pair(Pp, Qq), xy) == lhs


def test_reference_seeded_scalars(oracle_ref):
    g = golden("config1_bilinearity")
    assert oracle_ref.random_scalars(g["seed"].encode(), 4) == cat(g["scalars"])


def test_port_matches_reference_on_fresh_inputs(oracle_port, oracle_ref):
    """Beyond the committed fixtures: the C restatement (the checker most -m gpu tests use) against the compiled reference on fresh
    seeded inputs — 384 G1 and 256 G2 multiplications with scalars up to 2^256 and the edge scalars, additions, (de)compression,
    64 pairings with their Miller values and final exponentiations, GT operations, the bucket product and sum_of_products."""
    from util import scalars
    n1, n2, npair = 384, 256, 64
    g1, g2 = oracle_ref.g1_generator(), oracle_ref.g2_generator()
    edge = b"".join(int(k).to_bytes(32, "big") for k in (0, 1, 2, R - 1, R, R + 1, (1 << 128) - 1, (1 << 256) - 1))
    pts = oracle_ref.g1_mul(g1 * n1, scalars(901, n1), 96)
    q = oracle_ref.g2_mul(g2 * n2, scalars(902, n2), 192, 4)
    sc1 = scalars(903, n1 - 8, 1 << 256) + edge
    sc2 = scalars(904, n2 - 8, 1 << 256) + edge
    for fmt in (49, 96):
        assert oracle_port.g1_mul(pts, sc1, fmt, 4) == oracle_ref.g1_mul(pts, sc1, fmt, 4)
    for fmt in (97, 192):
        assert oracle_port.g2_mul(q, sc2, fmt, 4) == oracle_ref.g2_mul(q, sc2, fmt, 4)
    # additions incl. P + P, P + (-P) (the golden files hold the infinity cases), compression round trips
    half = 96 * (n1 // 2)
    assert oracle_port.g1_add(pts[:half], pts[half:], 96) == oracle_ref.g1_add(pts[:half], pts[half:], 96)
    assert oracle_port.g1_add(pts[:half], pts[:half], 49) == oracle_ref.g1_add(pts[:half], pts[:half], 49)
    h2 = 192 * (n2 // 2)
    assert oracle_port.g2_add(q[:h2], q[h2:], 192) == oracle_ref.g2_add(q[:h2], q[h2:], 192)
    c1 = oracle_ref.g1_mul(pts, b"".join((1).to_bytes(32, "big") for _ in range(n1)), 49, 4)
    assert oracle_port.g1_decompress(c1) == oracle_ref.g1_decompress(c1)
    c2 = oracle_ref.g2_mul(q, b"".join((1).to_bytes(32, "big") for _ in range(n2)), 97, 4)
    assert oracle_port.g2_decompress(c2) == oracle_ref.g2_decompress(c2)
    # pairings: whole value, Miller value, final exponentiation of the Miller value; one lane with each argument at infinity
    p1 = pts[:96 * (npair - 2)] + bytes(96) + pts[:96]
    q2 = q[:192 * (npair - 2)] + q[:192] + bytes(192)
    gt_ref = oracle_ref.pair(p1, q2, 4)
    assert oracle_port.pair(p1, q2, 4) == gt_ref
    mil = oracle_ref.miller(p1, q2)
    assert oracle_port.miller(p1, q2) == mil
    assert oracle_port.fexp(mil) == oracle_ref.fexp(mil) == gt_ref
    gta, gtb = gt_ref[:576 * 16], gt_ref[576 * 16:576 * 32]
    assert oracle_port.gt_op("mul", gta, gtb) == oracle_ref.gt_op("mul", gta, gtb)
    assert oracle_port.gt_op("pow", gta, sc1[:32 * 16]) == oracle_ref.gt_op("pow", gta, sc1[:32 * 16])
    assert oracle_port.gt_op("pow", mil[:576 * 8], sc1[-32 * 8:]) == oracle_ref.gt_op("pow", mil[:576 * 8], sc1[-32 * 8:])    # non-unitary bases, edge exponents
    # products: the header-level chain (g1_msm) and the boundary's sum_of_products
    assert oracle_port.g1_msm(pts[:96 * 64], sc1[-32 * 64:], 49, 4) == oracle_ref.g1_msm(pts[:96 * 64], sc1[-32 * 64:], 49, 4)


def test_hash_to_g1_and_zp_golden(orc):
    g = golden("hash_zp")
    d = cat(g["digests"])
    assert orc.g1_from_hash(d, 96) == cat(g["g1_from_hash_96"])
    assert orc.g1_from_hash(d, 49) == cat(g["g1_from_hash_49"])
    assert orc.zp_from_hash(d) == cat(g["zp_from_hash"])
    a, b = cat(g["zp_a"]), cat(g["zp_b"])
    for op in ("mul", "add", "sub", "neg", "inv"):
        assert orc.zp_op(op, a, b if op in ("mul", "add", "sub") else None) == cat(g["zp_" + op]), op


def test_zp_golden_against_python_ints():
    g = golden("hash_zp")
    for i, (x, y) in enumerate(zip(g["zp_a"], g["zp_b"])):
        x, y = int(x, 16) % R, int(y, 16) % R
        assert int(g["zp_mul"][i], 16) == x * y % R
        assert int(g["zp_sub"][i], 16) == (x - y) % R
        assert int(g["zp_inv"][i], 16) == pow(x, R - 2, R)
    for dg, z in zip(g["digests"], g["zp_from_hash"]):
        assert int(z, 16) == int(dg, 16) % R


def test_bbs_wire_restatement_matches_reference(oracle_ref, oracle_port):
    """encode_to<Zp> (zp_number.hpp:1011-1037) and the wire-format BBS+ verification: the C restatement against the compiled
    reference on valid, forged and malformed signatures (CPU only)."""
    from util import R, golden, prng, scalars
    g1 = bytes.fromhex(golden("g1")["generator"]); g2 = bytes.fromhex(golden("g2")["generator"])
    for msg in (b"", b"a", b"Hello, BBS+!", bytes(range(31)), bytes(range(32)), bytes(range(100))):
        assert oracle_ref.encode_to_zp(msg) == oracle_port.encode_to_zp(msg)
    assert oracle_port.encode_to_zp(b"Hello, BBS+!").hex() == "01" + b"Hello, BBS+!".hex() + "00" * 19
    gs = oracle_ref.g1_mul(g1 * 4, scalars(801, 4), 96)
    G1p, h0, h = gs[:96], gs[96:192], gs[192:]
    G2p = oracle_ref.g2_mul(g2, scalars(802, 1), 192)
    gamma = prng(803, 0) % R
    w = oracle_ref.g2_mul(G2p, gamma.to_bytes(32, "big"), 192)
    pp = oracle_ref.g1_compress(G1p) + oracle_ref.g2_compress(G2p) + oracle_ref.g1_compress(h0)
    h49, pk = oracle_ref.g1_compress(h), oracle_ref.g2_compress(w)
    msg_len, n = 40, 8
    sigs, msgs = b"", b""
    for j in range(n):
        msg = bytes(prng(804, j * 64 + b, 1) for b in range(msg_len))
        u = oracle_ref.encode_to_zp(msg)
        x, r = prng(805, j) % R, prng(806, j) % R
        B = oracle_ref.g1_msm(G1p + h0 + h, (1).to_bytes(32, "big") + r.to_bytes(32, "big") + u, 96, 1)
        a = oracle_ref.g1_mul(B, pow((gamma + x) % R, -1, R).to_bytes(32, "big"), 96)
        sig = bytearray(oracle_ref.g1_compress(a) + bytes(16) + x.to_bytes(32, "big") + bytes(16) + r.to_bytes(32, "big"))
        if j == 2:
            msg = bytes([msg[0] ^ 0x80]) + msg[1:]
        if j == 3:
            sig[49 + 16:49 + 48] = R.to_bytes(32, "big")
        if j == 4:
            sig[0] = 0x07
        if j == 5:
            sig[:49] = bytes(49)
        sigs += bytes(sig); msgs += msg
    a = oracle_ref.bbs_plus_verify_wire(pp, h49, pk, sigs, msgs, msg_len, 4)
    b = oracle_port.bbs_plus_verify_wire(pp, h49, pk, sigs, msgs, msg_len, 4)
    assert a == b and list(a) == [1, 1, 0, 0xff, 0xff, 0, 1, 1]
