"""CPU tests of the DEVICE algorithms: tests/host_sim/sim.cpp compiles crypto12381_amd/csrc/*.hpp
for the host with C12381_CHECK_BOUNDS (every limb/value bound of fp.hpp asserted at run time) and the
results are compared with the golden vectors / the oracle.  This is how the HIP code is validated in
the build container, which has no GPU; it is not a product path."""
import ctypes
import os
import subprocess

import pytest

from util import P, R, cat, golden, scalars

HERE = os.path.dirname(os.path.abspath(__file__))
SIM_DIR = os.path.join(HERE, "host_sim")
CSRC = os.path.join(os.path.dirname(HERE), "crypto12381_amd", "csrc")
sz = ctypes.c_size_t


@pytest.fixture(scope="module")
def sim():
    so = os.path.join(SIM_DIR, "libsim.so")
    srcs = [os.path.join(SIM_DIR, "sim.cpp")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["g++", "-O1", "-std=c++17", "-DC12381_CHECK_BOUNDS", "-fPIC", "-shared", "-pthread", "-o", so,
                        os.path.join(SIM_DIR, "sim.cpp")], check=True)
    return ctypes.CDLL(so)


def _fp(sim, op, a, b):
    n = len(a) // 48
    out = ctypes.create_string_buffer(48 * n)
    assert sim.sim_fp_op_batch(op, sz(n), a, b, out) == 0
    return out.raw


def test_sim_fp_ops(sim):
    g = golden("fp")
    a, b = cat(g["a"]), cat(g["b"])
    for op, name in ((0, "mul"), (1, "add"), (2, "sub"), (3, "sqr"), (4, "neg"), (5, "inv")):
        assert _fp(sim, op, a, b) == cat(g[name]), name
    out = _fp(sim, 7, a, b)
    assert all(int.from_bytes(out[48 * i:48 * i + 48], "big") == int(g["a"][i], 16) * 12 % P for i in range(len(g["a"])))
    out = _fp(sim, 8, a, b)
    assert all(int.from_bytes(out[48 * i:48 * i + 48], "big") == int(g["a"][i], 16) % P for i in range(len(g["a"])))
    out = _fp(sim, 6, a, b)      # sqrt candidate: squares back when a is a residue
    for i, qr in enumerate(g["sqrt_is_qr"]):
        if qr:
            assert pow(int.from_bytes(out[48 * i:48 * i + 48], "big"), 2, P) == int(g["a"][i], 16) % P


def test_sim_inversion_by_divsteps(sim):
    """fp_inv (safegcd divsteps, fp.hpp) against Python's modular inverse and against the Fermat ladder it replaced: edge values,
    values the iteration treats specially (powers of two, p - small, small), a few thousand random ones; 0 -> 0 like FP_inv."""
    from util import prng
    vals = [0, 1, 2, 3, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, P, P + 1, (1 << 384) - 1, (1 << 381) - 1, 0xd201000000010000]
    vals += [1 << b for b in range(0, 384, 7)] + [(1 << b) - 1 for b in range(1, 384, 11)] + [P - (1 << b) for b in range(0, 380, 13)]
    vals += [prng(4242, i, 48) for i in range(3000)]
    a = b"".join((v % (1 << 384)).to_bytes(48, "big") for v in vals)
    got = _fp(sim, 5, a, None)
    fermat = _fp(sim, 9, a, None)
    assert got == fermat
    for i, v in enumerate(vals):
        v %= 1 << 384
        exp = pow(v % P, -1, P) if v % P else 0
        assert int.from_bytes(got[48 * i:48 * i + 48], "big") == exp, hex(v)


def test_sim_glv_split(sim):
    x2 = 0xd201000000010000 ** 2
    ks = [0, 1, R - 1, R, R + 5, (1 << 256) - 1, x2, x2 - 1, x2 + 1, 3 * x2 - 1] + \
         [int.from_bytes(scalars(77, 20, 1 << 256)[32 * i:32 * i + 32], "big") for i in range(20)]
    # the split is a Barrett division by x^2 with two correction steps: walk the multiples of x^2 (quotient estimate
    # off by 0, 1, 2), powers of two and a few thousand random values
    qs = [int.from_bytes(scalars(78, 400)[32 * i:32 * i + 32], "big") >> (128 + (i % 120)) for i in range(400)] + [(R // x2) - j for j in range(3)]
    ks += [q * x2 + d for q in qs for d in (-2, -1, 0, 1, 2) if 0 <= q * x2 + d < R]
    ks += [(1 << b) + d for b in range(255) for d in (-1, 0, 1) if 0 <= (1 << b) + d < R]
    ks += [int.from_bytes(scalars(79, 3000)[32 * i:32 * i + 32], "big") for i in range(3000)]
    for k in ks:
        k0 = (ctypes.c_uint32 * 4)()
        k1 = (ctypes.c_uint32 * 4)()
        sim.sim_glv_split((k % (1 << 256)).to_bytes(32, "big"), k0, k1)
        a0 = sum(k0[i] << (32 * i) for i in range(4))
        a1 = sum(k1[i] << (32 * i) for i in range(4))
        assert a0 + a1 * x2 == (k % (1 << 256)) % R and a0 < x2


def test_sim_gs_split(sim):
    """G2 decomposition k = u0 + u1|x| + u2|x|^2 + u3|x|^3 by word-wise division with a precomputed reciprocal: digits
    against Python integers for boundary values (multiples of |x|^j +- 1, powers of two) and random scalars."""
    x = 0xd201000000010000
    ks = [0, 1, x - 1, x, x + 1, x * x - 1, x * x, x ** 3 - 1, x ** 3, x ** 3 + 1, R - 1, R, R + 7, (1 << 256) - 1]
    ks += [q * x ** j + d for j in (1, 2, 3) for q in (1, 2, x - 1, 0xffffffff, 1 << 63) for d in (-1, 0, 1) if 0 <= q * x ** j + d < R]
    ks += [(1 << b) + d for b in range(0, 255, 3) for d in (-1, 0, 1) if (1 << b) + d >= 0]
    ks += [int.from_bytes(scalars(88, 3000, 1 << 256)[32 * i:32 * i + 32], "big") for i in range(3000)]
    for k in ks:
        u = (ctypes.c_uint32 * 8)()
        sim.sim_gs_split((k % (1 << 256)).to_bytes(32, "big"), u)
        v = (k % (1 << 256)) % R
        for i in range(4):
            d = u[2 * i] | (u[2 * i + 1] << 32)
            exp = v % x if i < 3 else v
            assert d == exp, (hex(k), i)
            v //= x


def test_sim_g1_mul_golden(sim):
    g = golden("g1")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    n = len(pts) // 96
    for fmt, key in ((49, "mul49"), (96, "mul96")):
        out = ctypes.create_string_buffer(fmt * n)
        assert sim.sim_g1_mul_batch(sz(n), pts, sc, out, fmt) == 0
        assert out.raw == cat(g[key])


def test_sim_g1_mul_random_vs_oracle(sim, oracle_port):
    n = 24
    g1 = bytes.fromhex(golden("g1")["generator"])
    pts = oracle_port.g1_mul(g1 * n, scalars(501, n), 96, 4)
    sc = scalars(502, n, 1 << 256)
    out = ctypes.create_string_buffer(49 * n)
    assert sim.sim_g1_mul_batch(sz(n), pts, sc, out, 49) == 0
    assert out.raw == oracle_port.g1_mul(pts, sc, 49, 4)


G2_FORMS = ("sim_g2_mul_batch", "sim_g2h_mul_batch")      # one lane per point (fp2) / two lanes per point (fp2h, k_g2h.hip)


@pytest.mark.parametrize("form", G2_FORMS)
def test_sim_g2_mul_golden(sim, form):
    g = golden("g2")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    n = len(pts) // 192
    for fmt, key in ((97, "mul97"), (192, "mul192")):
        out = ctypes.create_string_buffer(fmt * n)
        assert getattr(sim, form)(sz(n), pts, sc, out, fmt) == 0
        assert out.raw == cat(g[key])


@pytest.mark.parametrize("form", G2_FORMS)
def test_sim_g2_mul_random_vs_oracle(sim, oracle_port, form):
    """random points of G2, scalars up to 2^256 and the edge scalars 0, 1, r - 1, r, r + 1, 2^256 - 1 against the oracle"""
    gen = bytes.fromhex(golden("g2")["generator"])
    n = 24
    pts = oracle_port.g2_mul(gen * n, scalars(511, n), 192) + bytes(192)
    edge = b"".join(int(k).to_bytes(32, "big") for k in (0, 1, R - 1, R, R + 1, (1 << 256) - 1))
    sc = scalars(512, n - 6, 1 << 256) + edge + scalars(513, 1)
    out = ctypes.create_string_buffer(192 * (n + 1))
    assert getattr(sim, form)(sz(n + 1), pts, sc, out, 192) == 0
    assert out.raw == oracle_port.g2_mul(pts, sc, 192, 4)


def test_sim_points_outside_the_subgroup(sim):
    """GLV (G1) and GS (G2) on the device reproduce the reference's results for curve points outside G1/G2."""
    g = golden("g1")
    pts, sc = cat(g["offsubgroup_points"]), cat(g["offsubgroup_scalars"])
    out = ctypes.create_string_buffer(96 * 6)
    assert sim.sim_g1_mul_batch(sz(6), pts, sc, out, 96) == 0 and out.raw == cat(g["offsubgroup_mul96"])
    g = golden("g2")
    pts, sc = cat(g["offsubgroup_points"]), cat(g["offsubgroup_scalars"])
    for form in G2_FORMS:
        out = ctypes.create_string_buffer(192 * 6)
        assert getattr(sim, form)(sz(6), pts, sc, out, 192) == 0 and out.raw == cat(g["offsubgroup_mul192"])


def test_sim_small_scalars_outside_the_subgroup(sim):
    """k < x^2 (G1) / a zero odd base-|x| digit (G2): the reference's decomposition keeps r as a sub-scalar and adds
    [r]phi(P) / [r]psi^i(Q); the device code reproduces that term (it vanishes on the subgroup)."""
    g = golden("g1")
    pts, sc = cat(g["offsubgroup_small_points"]), cat(g["offsubgroup_small_scalars"])
    n = len(sc) // 32
    out = ctypes.create_string_buffer(96 * n)
    assert sim.sim_g1_mul_batch(sz(n), pts, sc, out, 96) == 0
    exp = cat(g["offsubgroup_small_mul96"])
    assert [out.raw[96 * i:96 * i + 96] == exp[96 * i:96 * i + 96] for i in range(n)] == [True] * n
    gen = bytes.fromhex(g["generator"])
    out2 = ctypes.create_string_buffer(96 * n)           # on the subgroup the same scalars give plain multiples
    assert sim.sim_g1_mul_batch(sz(n), gen * n, sc, out2, 96) == 0
    assert out2.raw[96:192] == gen and out2.raw[:96] == bytes(96)
    g = golden("g2")
    pts, sc = cat(g["offsubgroup_small_points"]), cat(g["offsubgroup_small_scalars"])
    n = len(sc) // 32
    exp = cat(g["offsubgroup_small_mul192"])
    gen = bytes.fromhex(g["generator"])
    for form in G2_FORMS:
        out = ctypes.create_string_buffer(192 * n)
        assert getattr(sim, form)(sz(n), pts, sc, out, 192) == 0
        assert [out.raw[192 * i:192 * i + 192] == exp[192 * i:192 * i + 192] for i in range(n)] == [True] * n
        out2 = ctypes.create_string_buffer(192 * n)
        assert getattr(sim, form)(sz(n), gen * n, sc, out2, 192) == 0
        assert out2.raw[192:384] == gen and out2.raw[:192] == bytes(192)


def test_sim_pairing_golden(sim):
    """Miller loop + final exponentiation of the device headers, incl. infinity arguments."""
    g = golden("pairing")
    g1, g2 = cat(g["g1"]), cat(g["g2"])
    n = len(g1) // 96
    out = ctypes.create_string_buffer(576 * n)
    assert sim.sim_pair_batch(sz(n), g1, g2, out) == 0
    assert out.raw == cat(g["gt"])
    a1, a2, b1, b2 = cat(g["eq_a1"]), cat(g["eq_a2"]), cat(g["eq_b1"]), cat(g["eq_b2"])
    m = len(a1) // 96
    ok = ctypes.create_string_buffer(m)
    assert sim.sim_pair_eq_batch(sz(m), a1, a2, b1, b2, ok) == 0
    assert list(ok.raw[:m]) == g["eq"]
    a1, a2, b1, b2 = cat(g["eq2_a1"]), cat(g["eq2_a2"]), cat(g["eq2_b1"]), cat(g["eq2_b2"])
    m = len(a1) // 96
    ok = ctypes.create_string_buffer(m)
    assert sim.sim_pair_eq_batch(sz(m), a1, a2, b1, b2, ok) == 0
    assert list(ok.raw[:m]) == g["eq2"]


def test_sim_decompress(sim):
    g = golden("g1")
    cin = cat(g["compressed"])
    n = len(cin) // 49
    out, st = ctypes.create_string_buffer(96 * n), ctypes.create_string_buffer(n)
    assert sim.sim_g1_decompress_batch(sz(n), cin, out, st) == 0
    assert list(st.raw[:n]) == g["decompress_status"]
    assert out.raw == cat(g["decompressed"])
    g = golden("g2")
    cin = cat(g["compressed"])
    n = len(cin) // 97
    out, st = ctypes.create_string_buffer(192 * n), ctypes.create_string_buffer(n)
    assert sim.sim_g2_decompress_batch(sz(n), cin, out, st) == 0
    assert list(st.raw[:n]) == g["decompress_status"]
    assert out.raw == cat(g["decompressed"])


def test_sim_gt_ops_and_split_pairing(sim, oracle_port):
    g = golden("pairing")
    gt = cat(g["gt"])
    gta, gtb = gt[:576 * 4], gt[576 * 4:]
    out = ctypes.create_string_buffer(576 * 4)
    assert sim.sim_gt_op_batch(0, sz(4), gta, gtb, out) == 0 and out.raw == cat(g["gt_mul"])
    assert sim.sim_gt_op_batch(1, sz(4), gta, None, out) == 0 and out.raw == cat(g["gt_conj"])
    assert sim.sim_gt_op_batch(2, sz(4), gta, cat(g["gt_pow_exp"]), out) == 0 and out.raw == cat(g["gt_pow"])
    # Miller value and final exponentiation separately, against the oracle's PAIR_ate / PAIR_fexp
    g1, g2 = cat(g["g1"])[:96 * 3], cat(g["g2"])[:192 * 3]
    m = ctypes.create_string_buffer(576 * 3)
    assert sim.sim_miller_batch(sz(3), g1, g2, m) == 0
    assert m.raw == oracle_port.miller(g1, g2)
    f = ctypes.create_string_buffer(576 * 3)
    assert sim.sim_gt_op_batch(3, sz(3), m.raw, None, f) == 0
    assert f.raw == gt[:576 * 3] == oracle_port.fexp(m.raw)


def test_sim_pairing_three_lanes(sim):
    """pairing3.hpp: three lanes per pairing, cross-lane shuffles emulated by three host threads and a barrier."""
    g = golden("pairing")
    g1, g2 = cat(g["g1"]), cat(g["g2"])
    n = len(g1) // 96
    out = ctypes.create_string_buffer(576 * n)
    assert sim.sim_pair3_batch(sz(n), g1, g2, out) == 0
    assert out.raw == cat(g["gt"])
    a1, a2, b1, b2 = cat(g["eq_a1"]), cat(g["eq_a2"]), cat(g["eq_b1"]), cat(g["eq_b2"])
    m = len(a1) // 96
    ok = ctypes.create_string_buffer(m)
    assert sim.sim_pair3_eq_batch(sz(m), a1, a2, b1, b2, ok) == 0
    assert list(ok.raw[:m]) == g["eq"]
    # degenerate / adversarial rows through the joint Miller loop (one-sided infinity, negated and off-subgroup arguments)
    a1, a2, b1, b2 = cat(g["eq2_a1"]), cat(g["eq2_a2"]), cat(g["eq2_b1"]), cat(g["eq2_b2"])
    m = len(a1) // 96
    ok = ctypes.create_string_buffer(m)
    assert sim.sim_pair3_eq_batch(sz(m), a1, a2, b1, b2, ok) == 0
    assert list(ok.raw[:m]) == g["eq2"]


def test_sim_msm_pippenger(sim, oracle_port):
    """Bucket-method MSM of msm.hpp (device routines run sequentially, std::stable_sort instead of hipCUB)."""
    g = golden("g1")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    n = len(pts) // 96
    for c in (4, 7, 16):
        out = ctypes.create_string_buffer(49)
        assert sim.sim_g1_msm_pippenger(sz(n), pts, sc, out, 49, c) == 0
        assert out.raw.hex() == g["msm49"], c
    pts, sc = cat(g["offsubgroup_points"]), cat(g["offsubgroup_scalars"])
    out = ctypes.create_string_buffer(49)
    assert sim.sim_g1_msm_pippenger(sz(6), pts, sc, out, 49, 5) == 0
    assert out.raw.hex() == g["offsubgroup_msm49"]
    # scalars below x^2 on points outside G1: the [r]phi(P) terms of multiply() arrive through the extra bucket
    spts, ssc = cat(g["offsubgroup_small_points"]), cat(g["offsubgroup_small_scalars"])
    mixed_p, mixed_s = spts + pts + cat(g["points"])[:96 * 5], ssc + sc + cat(g["scalars"])[:32 * 5]
    out = ctypes.create_string_buffer(96)
    assert sim.sim_g1_msm_pippenger(sz(len(mixed_p) // 96), mixed_p, mixed_s, out, 96, 5) == 0
    assert out.raw == oracle_port.g1_msm(mixed_p, mixed_s, 96, 1)
    m = 300
    g1 = bytes.fromhex(g["generator"])
    p = oracle_port.g1_mul(g1 * m, scalars(801, m), 96, 4)
    k = scalars(802, m, 1 << 256)
    out = ctypes.create_string_buffer(96)
    assert sim.sim_g1_msm_pippenger(sz(m), p, k, out, 96, 0) == 0
    assert out.raw == oracle_port.g1_msm(p, k, 96, 4)


def test_sim_hash_to_g1_and_zp(sim):
    """h2c.hpp / fr.hpp as the kernels run them, against the reference's vectors (degenerate digests included)"""
    g = golden("hash_zp")
    d = cat(g["digests"])
    n = len(d) // 64
    for fmt in (96, 49):
        out = ctypes.create_string_buffer(fmt * n)
        assert sim.sim_g1_from_hash_batch(sz(n), d, out, fmt) == 0
        assert out.raw == cat(g["g1_from_hash_%d" % fmt])
    out = ctypes.create_string_buffer(32 * n)
    assert sim.sim_zp_from_hash_batch(sz(n), d, out) == 0
    assert out.raw == cat(g["zp_from_hash"])
    a, b = cat(g["zp_a"]), cat(g["zp_b"])
    m = len(a) // 32
    for op, name in enumerate(("mul", "add", "sub", "neg", "inv")):
        out = ctypes.create_string_buffer(32 * m)
        assert sim.sim_zp_op_batch(op, sz(m), a, b if op <= 2 else None, out) == 0
        assert out.raw == cat(g["zp_" + name]), name


def test_sim_fixed_base_tables(sim, oracle_port):
    """fixed_base.hpp: table entries, table-driven evaluation and the subgroup tests, against the generic results"""
    g = golden("g1")
    gen = bytes.fromhex(g["generator"])
    sc = cat(g["scalars"])[:32 * 10] + cat(g["offsubgroup_small_scalars"])
    n = len(sc) // 32
    base = oracle_port.g1_mul(gen, scalars(901, 1), 96)
    out = ctypes.create_string_buffer(96 * n)
    assert sim.sim_g1_fixed_mul_batch(sz(n), base, sc, out) == 0
    assert out.raw == oracle_port.g1_mul(base * n, sc, 96)
    assert sim.sim_g1_fixed_mul_batch(sz(1), cat(g["offsubgroup_points"])[:96], sc, out) == -2      # not a subgroup point
    assert sim.sim_g1_fixed_mul_batch(sz(1), bytes(96), sc, out) == -2
    g = golden("g2")
    gen2 = bytes.fromhex(g["generator"])
    sc2 = cat(g["scalars"])[:32 * 6] + cat(g["offsubgroup_small_scalars"])[:32 * 8]
    n = len(sc2) // 32
    base2 = oracle_port.g2_mul(gen2, scalars(902, 1), 192)
    out = ctypes.create_string_buffer(192 * n)
    assert sim.sim_g2_fixed_mul_batch(sz(n), base2, sc2, out) == 0
    assert out.raw == oracle_port.g2_mul(base2 * n, sc2, 192)
    assert sim.sim_g2_fixed_mul_batch(sz(1), cat(g["offsubgroup_points"])[:192], sc2, out) == -2


def test_sim_fixed_g2_lines(sim, oracle_port):
    """precomputed line coefficients of a fixed G2 argument + the table-driven joint Miller loop (pairing3.hpp): the
    product of two pairings must be the reference's pair_double_ate value after the final exponentiation"""
    g = golden("pairing")
    g1s = cat(g["g1"])
    a, c = g1s[:96 * 3] + bytes(96), g1s[96 * 3:96 * 6] + g1s[:96]          # one G1 argument at infinity
    w, q = cat(g["g2"])[:192], cat(g["g2"])[192:384]
    n = 4
    out = ctypes.create_string_buffer(576 * n)
    assert sim.sim_pair2_fixed_batch(sz(n), a, w, c, q, out) == 0
    assert out.raw == oracle_port.pair2(a, w * n, c, q * n)


def test_sim_gt_power_windowed_and_generic_routes(sim, oracle_port):
    """gt3_op_kernel's power: pairing values (cyclotomic subgroup) take the 4-bit windowed ladder, anything else the reference's own
    digit sequence; both against the oracle, edge exponents included, with the bounds checker on (C12381_CHECK_BOUNDS build)"""
    from util import prng
    g = golden("pairing")
    gt = cat(g["gt"])                                  # pairing values
    mil = oracle_port.miller(cat(g["g1"])[:96 * 2], cat(g["g2"])[:192 * 2])          # Miller values: NOT in the subgroup, not even unitary
    R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
    exps = [0, 1, 2, 15, 16, 17, R - 1, R, R + 1, (1 << 256) - 1, 1 << 255, 0x8000000000000000000000000000000080000000000000000000000000000001,
            0xf0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0, prng(77, 1, 32), prng(77, 2, 32)]
    n = len(exps)
    a = b"".join(gt[576 * (i % (len(gt) // 576)):576 * (i % (len(gt) // 576)) + 576] for i in range(n))
    e = b"".join(x.to_bytes(32, "big") for x in exps)
    want = oracle_port.gt_op("pow", a, e)
    from oracle.bindings import Oracle, have_reference
    if have_reference():                               # the compiled reference itself, where it is present
        ref = Oracle("reference")
        assert ref.gt_op("pow", a, e) == want and ref.gt_op("pow", mil, e[32 * 3:32 * 5]) == oracle_port.gt_op("pow", mil, e[32 * 3:32 * 5])
    out = ctypes.create_string_buffer(576 * n)
    sim.sim_gt3_pow_route(0)
    assert sim.sim_gt3_op_batch(2, sz(n), a, e, out) == 0 and out.raw == want
    assert sim.sim_gt3_pow_route(1) == n               # every pairing value took the windows
    assert sim.sim_gt3_op_batch(2, sz(n), a, e, out) == 0 and out.raw == want      # the generic ladder agrees on them
    assert sim.sim_gt3_pow_route(2) == 0
    assert sim.sim_gt3_op_batch(2, sz(n), a, e, out) == 0 and out.raw == want      # ... and the ladder cut into the work-queue kernel's five tasks
    assert sim.sim_gt3_pow_route(2) == n
    e2q = e[32 * 3:32 * 5]
    o2q = ctypes.create_string_buffer(576 * 2)
    assert sim.sim_gt3_op_batch(2, sz(2), mil, e2q, o2q) == 0 and o2q.raw == oracle_port.gt_op("pow", mil, e2q)     # the generic ladder in four ranges
    assert sim.sim_gt3_pow_route(0) == 0
    # outside the subgroup: the kernel's choice must be the generic ladder, value as the reference's sequence gives it
    e2 = e[32 * 3:32 * 5]
    o2 = ctypes.create_string_buffer(576 * 2)
    assert sim.sim_gt3_op_batch(2, sz(2), mil, e2, o2) == 0 and o2.raw == oracle_port.gt_op("pow", mil, e2)
    # zero passes the membership test and is 0 (e != 0) / 1 (e = 0) on both routes
    z = bytes(576 * 2)
    ez = (5).to_bytes(32, "big") + bytes(32)
    assert sim.sim_gt3_op_batch(2, sz(2), z, ez, o2) == 0 and o2.raw == oracle_port.gt_op("pow", z, ez)
    assert sim.sim_gt3_pow_route(0) == 2


def test_sim_three_lane_miller_and_gt_ops(sim, oracle_port):
    """miller3_kernel / gt3_op_kernel bodies: the Miller VALUE (not only the pairing) and the GT operators on triples"""
    g = golden("pairing")
    g1, g2 = cat(g["g1"]), cat(g["g2"])
    n = len(g1) // 96
    out = ctypes.create_string_buffer(576 * n)
    assert sim.sim_miller3_batch(sz(n), g1, g2, out) == 0
    mil = oracle_port.miller(g1, g2)
    assert out.raw == mil
    gt = cat(g["gt"])
    a, b = gt[:576 * 4], gt[576 * 4:]
    o4 = ctypes.create_string_buffer(576 * 4)
    assert sim.sim_gt3_op_batch(0, sz(4), a, b, o4) == 0 and o4.raw == cat(g["gt_mul"])
    assert sim.sim_gt3_op_batch(1, sz(4), a, None, o4) == 0 and o4.raw == cat(g["gt_conj"])
    assert sim.sim_gt3_op_batch(2, sz(4), a, cat(g["gt_pow_exp"]), o4) == 0 and o4.raw == cat(g["gt_pow"])
    o3 = ctypes.create_string_buffer(576 * 3)
    assert sim.sim_gt3_op_batch(3, sz(3), mil[:576 * 3], None, o3) == 0 and o3.raw == gt[:576 * 3]
    one = oracle_port.gt_op("mul", a[:576], oracle_port.gt_op("conj", a[:576]))
    u = ctypes.create_string_buffer(2)
    assert sim.sim_gt3_op_batch(4, sz(2), one + a[:576], None, u) == 0 and u.raw == b"\x01\x00"
