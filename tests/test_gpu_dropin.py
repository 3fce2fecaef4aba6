"""GPU test of the DROP-IN boundary: oracle/_ref/libc12381_shimtest.so is the reference's own boundary file +
MIRACL for hash/big/random, with crypto12381_amd/csrc/miracl_core_interface_hip.cpp linked in front so that
every G1 / G2 / GT boundary function runs on the GPU through the C ABI (size-1 batches).  The same wrapper
(oracle/ref_wrap.cpp) drives both libraries, so equality of outputs is equality at the reference's seam."""
import pytest

from util import R, cat, golden, scalars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shim():
    from oracle.bindings import Oracle, have_shim
    if not have_shim():
        pytest.skip("oracle/_ref/libc12381_shimtest.so not built (needs /root/reference at build time)")
    return Oracle("shim")


def test_dropin_g1(shim, oracle_ref):
    g = golden("g1")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    assert shim.g1_generator() == oracle_ref.g1_generator()
    assert shim.g1_mul(pts, sc, 49) == cat(g["mul49"])                      # multiply + to_bytes(compressed)
    assert shim.g1_mul(pts, sc, 96) == cat(g["mul96"])
    assert shim.g1_add(cat(g["add_a"]), cat(g["add_b"]), 96) == cat(g["add96"])
    dec, st = shim.g1_decompress(cat(g["compressed"]))                       # from_bytes (sqrt on the device)
    assert list(st) == g["decompress_status"] and dec == cat(g["decompressed"])
    assert shim.g1_msm(pts, sc, 49, 1).hex() == g["msm49"]                  # multiply + add chain


def test_dropin_g1_sum_of_products_and_double_multiply(shim, oracle_ref):
    import ctypes
    g = golden("g1")
    pts, sc = cat(g["points"])[:96 * 9], cat(g["scalars"])[:32 * 9]
    for lib in (shim, oracle_ref):
        o = ctypes.create_string_buffer(49)
        assert lib.lib.ref_g1_sum_of_products(9, pts, sc, o, 49) == 0
        lib._sop = o.raw[:49]
        o2 = ctypes.create_string_buffer(96 * 4)
        assert lib.lib.ref_g1_mul2_batch(ctypes.c_size_t(4), pts[:96 * 4], pts[96 * 4:96 * 8], sc[:32 * 4], sc[32 * 4:32 * 8], o2, 96) == 0
        lib._mul2 = o2.raw[:96 * 4]
    assert shim._sop == oracle_ref._sop == oracle_ref.g1_msm(pts, sc, 49, 1)
    assert shim._mul2 == oracle_ref._mul2


def test_dropin_hash_to_g1(shim, oracle_ref):
    """G1Point::from_hash through the seam: residue, map_to_point, multiply_cofactor on the GPU (fixed_time_mod and the
    hashing stay with the reference's big/hash functions)."""
    g = golden("hash_zp")
    d = cat(g["digests"][7:])                      # the degenerate digests (u = 0 mod p) are covered at the C ABI level
    exp = cat(g["g1_from_hash_96"][7:])
    assert shim.g1_from_hash(d, 96) == exp
    assert shim.g1_from_hash(d, 49) == cat(g["g1_from_hash_49"][7:])
    u = b"".join((int(x, 16) % (1 << 384)).to_bytes(48, "big") for x in g["digests"][7:15])
    assert shim.g1_map_to_point(u) == oracle_ref.g1_map_to_point(u)


def test_dropin_g2(shim, oracle_ref):
    g = golden("g2")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    assert shim.g2_generator() == oracle_ref.g2_generator()
    assert shim.g2_mul(pts, sc, 97) == cat(g["mul97"])
    assert shim.g2_mul(pts, sc, 192) == cat(g["mul192"])
    assert shim.g2_add(cat(g["add_a"]), cat(g["add_b"]), 192) == cat(g["add192"])
    dec, st = shim.g2_decompress(cat(g["compressed"]))
    assert list(st) == g["decompress_status"] and dec == cat(g["decompressed"])


def test_dropin_pairing_and_gt(shim, oracle_ref):
    g = golden("pairing")
    g1, g2 = cat(g["g1"]), cat(g["g2"])
    gt = shim.pair(g1, g2)                                                   # pair_ate + pair_final_exponentiation + to_bytes
    assert gt == cat(g["gt"])
    a1, a2, b1, b2 = g1[:96 * 4], g2[:192 * 4], g1[96 * 4:], g2[192 * 4:]
    assert shim.pair2(a1, a2, b1, b2) == cat(g["pair2"])                    # pair_double_ate
    assert list(shim.pair_eq(cat(g["eq_a1"]), cat(g["eq_a2"]), cat(g["eq_b1"]), cat(g["eq_b2"]))) == g["eq"]
    gta, gtb = gt[:576 * 4], gt[576 * 4:]
    assert shim.gt_op("mul", gta, gtb) == cat(g["gt_mul"])
    assert shim.gt_op("conj", gta) == cat(g["gt_conj"])
    assert shim.gt_op("pow", gta, cat(g["gt_pow_exp"])) == cat(g["gt_pow"])
    assert shim.miller(g1, g2) == oracle_ref.miller(g1, g2)


def test_dropin_config1_bilinearity(shim):
    """BASELINE configs[0] at the boundary, on the GPU: pair(g1^x, g2^y) == pair(g1, g2)^(xy)."""
    g = golden("config1_bilinearity")
    sc = cat(g["scalars"])
    x, y = sc[64:96], sc[96:128]
    Pp, Qq = bytes.fromhex(g["P"]), bytes.fromhex(g["Q"])
    lhs = shim.pair(shim.g1_mul(Pp, x, 96), shim.g2_mul(Qq, y, 192))
    assert lhs.hex() == g["pair_Px_Qy"]
    assert shim.gt_op("pow", shim.pair(Pp, Qq), bytes.fromhex(g["xy"])) == lhs


def test_dropin_negate_sub_equal(shim, oracle_ref):
    """The seven seam functions no other test reaches — negate / sub / equal on point1 and point2, equal on fp12
    (/root/reference/src/miracl_core_interface.cpp:124-147, 207-226, 266-269: ECP_neg, ECP_sub, ECP_equals, ECP2_*, FP12_equals) —
    through the shim against the unmodified reference: ordinary points, P - P, infinity on either side, equal and unequal
    operands, and values that are results of seam arithmetic rather than fresh from the decoder.  (A point with y = 0 does
    not exist on either curve: #E(Fp) and #E'(Fp2) are odd, so there is no point of order two.)"""
    import ctypes
    sz = ctypes.c_size_t
    g = golden("g1")
    P = cat(g["points"])[:96 * 6]
    inf1 = bytes(96)
    A1 = P + P[:96 * 2] + inf1 + P[:96] + inf1                  # a
    B1 = P[96:] + P[:96] + P[:96 * 2] + P[:96] + inf1 + inf1    # b: different points, then P - P twice, inf - P, P - inf, inf - inf
    n1 = len(A1) // 96
    g2v = golden("g2")
    Q = cat(g2v["points"])[:192 * 4]
    inf2 = bytes(192)
    A2 = Q + Q[:192 * 2] + inf2 + Q[:192] + inf2
    B2 = Q[192:] + Q[:192] + Q[:192 * 2] + Q[:192] + inf2 + inf2
    n2 = len(A2) // 192
    gp = golden("pairing")
    gt = cat(gp["gt"])
    gta, gtb, gtm = gt[:576 * 4], gt[576 * 4:576 * 8], cat(gp["gt_mul"])
    res = {}
    for name, lib in (("shim", shim), ("ref", oracle_ref)):
        L = lib.lib
        r = {}
        for fmt in (96, 49):
            o = ctypes.create_string_buffer(fmt * n1)
            assert L.ref_g1_neg_batch(sz(n1), A1, o, fmt) == 0
            r["g1_neg%d" % fmt] = o.raw
            o = ctypes.create_string_buffer(fmt * n1)
            assert L.ref_g1_sub_batch(sz(n1), A1, B1, o, fmt) == 0
            r["g1_sub%d" % fmt] = o.raw
        o = ctypes.create_string_buffer(n1)
        assert L.ref_g1_equal_batch(sz(n1), A1, B1, o) == 0
        r["g1_eq"] = o.raw
        o = ctypes.create_string_buffer(n1)
        assert L.ref_g1_equal_batch(sz(n1), A1, A1, o) == 0
        r["g1_eq_self"] = o.raw
        # (a + b) == c with c = the golden sum (equal) and with c = a (unequal unless b is infinity)
        ga, gb, gs = cat(g["add_a"]), cat(g["add_b"]), cat(g["add96"])
        m = len(ga) // 96
        o = ctypes.create_string_buffer(m)
        assert L.ref_g1_equal_sum_batch(sz(m), ga, gb, gs, o) == 0
        r["g1_eq_sum"] = o.raw
        o = ctypes.create_string_buffer(m)
        assert L.ref_g1_equal_sum_batch(sz(m), ga, gb, ga, o) == 0
        r["g1_eq_sum_ne"] = o.raw
        for fmt in (192, 97):
            o = ctypes.create_string_buffer(fmt * n2)
            assert L.ref_g2_neg_batch(sz(n2), A2, o, fmt) == 0
            r["g2_neg%d" % fmt] = o.raw
            o = ctypes.create_string_buffer(fmt * n2)
            assert L.ref_g2_sub_batch(sz(n2), A2, B2, o, fmt) == 0
            r["g2_sub%d" % fmt] = o.raw
        o = ctypes.create_string_buffer(n2)
        assert L.ref_g2_equal_batch(sz(n2), A2, B2, o) == 0
        r["g2_eq"] = o.raw
        o = ctypes.create_string_buffer(n2)
        assert L.ref_g2_equal_batch(sz(n2), A2, A2, o) == 0
        r["g2_eq_self"] = o.raw
        ga2, gb2, gs2 = cat(g2v["add_a"]), cat(g2v["add_b"]), cat(g2v["add192"])
        m2 = len(ga2) // 192
        o = ctypes.create_string_buffer(m2)
        assert L.ref_g2_equal_sum_batch(sz(m2), ga2, gb2, gs2, o) == 0
        r["g2_eq_sum"] = o.raw
        o = ctypes.create_string_buffer(4)
        assert L.ref_gt_equal_batch(sz(4), gta, gtb, None, o) == 0
        r["gt_eq_ne"] = o.raw
        o = ctypes.create_string_buffer(4)
        assert L.ref_gt_equal_batch(sz(4), gta, gta, None, o) == 0
        r["gt_eq_self"] = o.raw
        o = ctypes.create_string_buffer(4)
        assert L.ref_gt_equal_batch(sz(4), gta, gtb, gtm, o) == 0             # a * b == the golden product
        r["gt_eq_prod"] = o.raw
        o = ctypes.create_string_buffer(4)
        assert L.ref_gt_equal_batch(sz(4), gta, gtb, gta, o) == 0             # a * b != a
        r["gt_eq_prod_ne"] = o.raw
        res[name] = r
    assert res["shim"].keys() == res["ref"].keys()
    for k in res["ref"]:
        assert res["shim"][k] == res["ref"][k], k
    # and the values are what the group laws say: P - P = infinity, inf - P = -P, P - inf = P
    s96 = res["ref"]["g1_sub96"]
    assert s96[96 * 6:96 * 8] == bytes(192) and s96[96 * 8:96 * 9] == res["ref"]["g1_neg96"][:96] and s96[96 * 9:96 * 10] == P[:96] and s96[96 * 10:] == inf1
    assert res["ref"]["g1_eq_self"] == b"\x01" * n1 and res["ref"]["g1_eq"][:6] == bytes(6) and res["ref"]["g1_eq"][6:8] == b"\x01\x01"
    assert res["ref"]["g1_eq_sum"] == b"\x01" * (len(cat(g["add_a"])) // 96)
    assert res["ref"]["gt_eq_self"] == b"\x01" * 4 and res["ref"]["gt_eq_ne"] == bytes(4) and res["ref"]["gt_eq_prod"] == b"\x01" * 4
