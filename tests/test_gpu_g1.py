"""GPU parity tests (run on the MI355X box with -m gpu): HIP path through the C ABI vs the
golden vectors generated from the reference and vs the CPU oracle on fresh seeded inputs."""
import pytest

from util import P, R, cat, golden, prng, scalars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from crypto12381_amd import Context
    c = Context(0)
    yield c
    c.close()


def test_fp_ops_golden(ctx):
    g = golden("fp")
    a, b = cat(g["a"]), cat(g["b"])
    for op in ("mul", "add", "sub", "sqr", "neg", "inv"):
        assert ctx.fp_op(op, a, b if op in ("mul", "add", "sub") else None) == cat(g[op]), op


def test_fp_mul_random_vs_python(ctx):
    n = 4096
    a = [prng(101, i, 48) % P for i in range(n)]
    b = [prng(102, i, 48) % P for i in range(n)]
    out = ctx.fp_op("mul", b"".join(x.to_bytes(48, "big") for x in a), b"".join(x.to_bytes(48, "big") for x in b))
    for i in range(n):
        assert int.from_bytes(out[48 * i:48 * i + 48], "big") == a[i] * b[i] % P


def test_g1_mul_golden(ctx):
    g = golden("g1")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    assert ctx.g1_mul(pts, sc, 49) == cat(g["mul49"])
    assert ctx.g1_mul(pts, sc, 96) == cat(g["mul96"])


def test_g1_add_golden(ctx):
    g = golden("g1")
    assert ctx.g1_add(cat(g["add_a"]), cat(g["add_b"]), 96) == cat(g["add96"])
    assert ctx.g1_add(cat(g["add_a"]), cat(g["add_b"]), 49) == cat(g["add49"])


def test_g1_msm_golden(ctx):
    g = golden("g1")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    assert ctx.g1_msm(pts, sc, 49).hex() == g["msm49"]
    assert ctx.g1_msm(pts[:96], sc[:32], 49) == ctx.g1_mul(pts[:96], sc[:32], 49)


def test_g1_mul_vs_oracle_random(ctx, oracle_port):
    """Fresh seeded batch (ragged size, not a multiple of the block) against the CPU oracle."""
    n = 1000
    g1 = bytes.fromhex(golden("g1")["generator"])
    pts = ctx.g1_mul(g1 * n, scalars(201, n), 96)            # P_i = G^{s_i} built by the HIP path ...
    assert pts[:96 * 64] == oracle_port.g1_mul(g1 * 64, scalars(201, 64), 96, 4)   # ... and spot-checked
    sc = scalars(202, n, 1 << 256)                           # unreduced 256-bit scalars
    got = ctx.g1_mul(pts, sc, 49)
    assert got == oracle_port.g1_mul(pts, sc, 49, 16)


def test_g1_invalid_point_is_flagged(ctx):
    from crypto12381_amd import C12381Error
    g = golden("g1")
    good = bytes.fromhex(g["points"][0])
    bad = good[:95] + bytes([good[95] ^ 1])                  # y perturbed: not on the curve
    with pytest.raises(C12381Error):
        ctx.g1_mul(good + bad, scalars(5, 2), 49)
    out = ctx.g1_mul(good + bad, scalars(5, 2), 49, strict=False)
    assert out[49:] == b"\xff" * 49
    assert out[:49] == ctx.g1_mul(good, scalars(5, 1), 49)


def test_g1_full_size_properties(ctx, oracle_port):
    """BASELINE configs[1] size (2^20): size-independent checks — linearity through the MSM and
    a sampled oracle comparison."""
    n = 1 << 20
    g1 = bytes.fromhex(golden("g1")["generator"])
    m = 1 << 12
    base = ctx.g1_mul(g1 * m, scalars(301, m), 96)
    pts = base * (n // m)                                    # 2^20 points (period 2^12)
    sc = scalars(302, n)
    out = ctx.g1_mul(pts, sc, 96)
    idx = [0, 1, m - 1, m, n // 2 + 17, n - 1] + [prng(303, i, 4) % n for i in range(58)]
    sp = b"".join(pts[96 * i:96 * i + 96] for i in idx)
    ss = b"".join(sc[32 * i:32 * i + 32] for i in idx)
    exp = oracle_port.g1_mul(sp, ss, 96, 16)
    assert b"".join(out[96 * i:96 * i + 96] for i in idx) == exp
    # sum of all outputs == MSM of the inputs (both on the GPU), and == g^(sum s_i k_i) via one oracle mul
    one = (1).to_bytes(32, "big")
    lhs = ctx.g1_msm(out, one * n, 49)
    assert lhs == ctx.g1_msm(pts, sc, 49)
    s_base = [prng(301, i) % R for i in range(m)]
    tot = 0
    for i in range(n):
        tot += s_base[i % m] * int.from_bytes(sc[32 * i:32 * i + 32], "big")
    assert lhs == oracle_port.g1_mul(g1, (tot % R).to_bytes(32, "big"), 49)


def test_points_outside_the_subgroup(ctx):
    """No subgroup check in the reference: GLV/GS results on such points are pinned by golden vectors."""
    g = golden("g1")
    pts, sc = cat(g["offsubgroup_points"]), cat(g["offsubgroup_scalars"])
    assert ctx.g1_mul(pts, sc, 96) == cat(g["offsubgroup_mul96"])
    assert ctx.g1_msm(pts, sc, 49).hex() == g["offsubgroup_msm49"]
    g = golden("g2")
    pts, sc = cat(g["offsubgroup_points"]), cat(g["offsubgroup_scalars"])
    assert ctx.g2_mul(pts, sc, 192) == cat(g["offsubgroup_mul192"])
    # scalars below x^2 / zero odd base-|x| digits: the reference adds [r]phi(P) / [r]psi^i(Q) there
    assert ctx.g2_mul(cat(g["offsubgroup_small_points"]), cat(g["offsubgroup_small_scalars"]), 192) == cat(g["offsubgroup_small_mul192"])
    g = golden("g1")
    assert ctx.g1_mul(cat(g["offsubgroup_small_points"]), cat(g["offsubgroup_small_scalars"]), 96) == cat(g["offsubgroup_small_mul96"])


def test_msm_bucket_method_vs_naive_and_oracle(ctx, oracle_port):
    """The bucket-method MSM (>= 2^12 terms) against the oracle and against the n-scalar-muls path."""
    n = 5000
    g1 = bytes.fromhex(golden("g1")["generator"])
    pts = ctx.g1_mul(g1 * n, scalars(611, n), 96)
    pts = pts[:96 * 7] + bytes(96) + pts[96 * 8:]            # one point at infinity
    sc = scalars(612, n, 1 << 256)
    sc = (0).to_bytes(32, "big") + sc[32:]                   # one zero scalar
    got = ctx.g1_msm(pts, sc, 96)
    assert got == oracle_port.g1_msm(pts, sc, 96, 16)
    # sum of the individual products (GPU) folded with unit scalars must agree too
    prods = ctx.g1_mul(pts, sc, 96)
    assert ctx.g1_msm(prods, (1).to_bytes(32, "big") * n, 96) == got
    # off-subgroup points: every term must follow the reference's GLV evaluation
    g = golden("g1")
    off, osc = cat(g["offsubgroup_points"]), cat(g["offsubgroup_scalars"])
    reps = 700
    assert ctx.g1_msm(off * reps, osc * reps, 96) == oracle_port.g1_msm(off * reps, osc * reps, 96, 16)


def test_msm_large_product_path_vs_oracle(ctx, oracle_port):
    """From 2^15 terms on the product sorts 16-bit digit keys per window segment with positional values (c12381_hip.hip, msm.hpp
    msm_entry_value, k_g1.hip msm_ranges16_kernel); below that one sort over 32-bit keys.  Both sides of the switch against the oracle's chain of
    multiply() results, with the inputs that take the special routes: infinity, zero and small scalars (the [r]phi(S) bucket), points
    outside the subgroup, a run of equal scalars long enough to be cut into overflow segments."""
    g = golden("g1")
    g1 = bytes.fromhex(g["generator"])
    for n in ((1 << 15) - 1, (1 << 15) + 3):
        base = ctx.g1_mul(g1 * 4096, scalars(641, 4096), 96)
        pts = bytearray((base * (n // 4096 + 1))[:96 * n])
        sc = bytearray(scalars(642 + n, n, 1 << 256))
        pts[96 * 5:96 * 6] = bytes(96)                                        # infinity
        sc[32 * 9:32 * 10] = bytes(32)                                        # zero scalar
        off, osc = cat(g["offsubgroup_small_points"]), cat(g["offsubgroup_small_scalars"])
        m = len(off) // 96
        pts[96 * 100:96 * (100 + m)] = off; sc[32 * 100:32 * (100 + m)] = osc  # small scalars on points outside G1
        eq = scalars(643, 1)
        sc[32 * 2000:32 * 3000] = eq * 1000                                   # 1000 equal scalars: one long run per window
        got = ctx.g1_msm(bytes(pts), bytes(sc), 96)
        assert got == oracle_port.g1_msm(bytes(pts), bytes(sc), 96, 16), n


def test_msm_skewed_scalars(ctx, oracle_port):
    """Scalars that put thousands of terms into one bucket (all equal, all small, 128-bit, one giant value among zeros):
    runs longer than the per-lane cap are cut into overflow segments and recombined — same point as the oracle's sum."""
    n = 6000
    g1 = bytes.fromhex(golden("g1")["generator"])
    pts = ctx.g1_mul(g1 * n, scalars(621, n), 96)
    k = scalars(622, 1)
    x2 = 0xd201000000010000 ** 2
    cases = {
        "equal": k * n,
        "ones": (1).to_bytes(32, "big") * n,
        "128-bit": b"".join((int.from_bytes(scalars(623, n)[32 * j:32 * j + 32], "big") % (1 << 128)).to_bytes(32, "big") for j in range(n)),
        "two values": b"".join((k if j % 3 else (x2 + 7).to_bytes(32, "big")) for j in range(n)),
        "run of 33": k * 33 + scalars(624, n - 33),                 # cap 32 at this size: one overflow segment holding a single entry
        "run of 64": k * 64 + scalars(625, n - 64),                 # the cap and exactly two full segments of 16
        "zeros": bytes(32) * (n - 1) + k,
    }
    for name, sc in cases.items():
        assert ctx.g1_msm(pts, sc, 96) == oracle_port.g1_msm(pts, sc, 96, 16), name
    # equal scalars: k * (sum of the points)
    total = ctx.g1_msm(pts, (1).to_bytes(32, "big") * n, 96)
    assert ctx.g1_msm(pts, k * n, 96) == ctx.g1_mul(total, k, 96)


def test_msm_small_scalars_outside_the_subgroup(ctx, oracle_port):
    """The reference's product of powers is a chain of multiply() calls, and multiply() adds [r]phi(P) for scalars below x^2 —
    a cofactor point when P is outside G1.  The bucket method collects those terms separately; with them the product
    equals the oracle's chain for EVERY input (4200 such terms among ordinary ones, bucket path)."""
    g = golden("g1")
    off, osc = cat(g["offsubgroup_small_points"]), cat(g["offsubgroup_small_scalars"])
    reps = 300
    n_norm = 1000
    g1 = bytes.fromhex(g["generator"])
    norm_pts = ctx.g1_mul(g1 * n_norm, scalars(631, n_norm), 96)
    pts = off * reps + norm_pts + cat(g["offsubgroup_points"]) * 5
    sc = osc * reps + scalars(632, n_norm, 1 << 256) + cat(g["offsubgroup_scalars"]) * 5
    assert len(pts) // 96 >= 4096
    got = ctx.g1_msm(pts, sc, 96)
    assert got == oracle_port.g1_msm(pts, sc, 96, 16)
    # the same terms one by one (the n-scalar-muls path below 2^12 terms is exact by construction)
    small = ctx.g1_msm(off, osc, 96)
    assert small == oracle_port.g1_msm(off, osc, 96, 1)
    # unit scalars on points outside G1: every term owes [r]phi(P)
    ones = (1).to_bytes(32, "big") * (len(off) // 96) * reps
    assert ctx.g1_msm(off * reps, ones, 96) == oracle_port.g1_msm(off * reps, ones, 96, 16)
    # the small-scalar bucket is summed by a wavefront of its own in front of the bucket kernel up to 4 096 entries (msm_small_early_kernel)
    # and by the bucket kernel beyond: both sides of the switch, a count that leaves lanes of that wavefront empty, and a single entry
    m = len(off) // 96
    for small in (1, 100, 4096, 4097):
        sp = (off * (small // m + 1))[:96 * small]
        ss = (osc * (small // m + 1))[:32 * small]
        p2, s2 = sp + norm_pts, ss + scalars(633 + small, n_norm, 1 << 256)
        assert ctx.g1_msm(p2, s2, 96) == oracle_port.g1_msm(p2, s2, 96, 16), small


def test_fixed_base_entry_points(ctx, oracle_port):
    """g^x_i with one base: table path for subgroup bases, generic path otherwise — always equal to the generic batch."""
    g = golden("g1")
    gen = bytes.fromhex(g["generator"])
    n = 700
    sc = scalars(951, n - 14, 1 << 256) + cat(g["offsubgroup_small_scalars"])
    base = oracle_port.g1_mul(gen, scalars(952, 1), 96)
    assert ctx.g1_mul_fixed(base, sc, 49) == ctx.g1_mul(base * n, sc, 49)
    assert ctx.g1_mul_fixed(base, sc[:32 * 40], 96) == oracle_port.g1_mul(base * 40, sc[:32 * 40], 96, 4)
    off = cat(g["offsubgroup_points"])[:96]                 # not a subgroup point: generic path, still the reference's result
    assert ctx.g1_mul_fixed(off, sc[-32 * 14:], 96) == oracle_port.g1_mul(off * 14, sc[-32 * 14:], 96)
    assert ctx.g1_mul_fixed(gen, sc[:32 * 5], 96) == oracle_port.g1_mul(gen * 5, sc[:32 * 5], 96)     # base changed: table rebuilt
    assert ctx.g1_mul_fixed(bytes(96), sc[:64], 96) == bytes(192)
    g = golden("g2")
    gen2 = bytes.fromhex(g["generator"])
    sc2 = scalars(953, 90, 1 << 256) + cat(g["offsubgroup_small_scalars"])
    m = len(sc2) // 32
    assert ctx.g2_mul_fixed(gen2, sc2, 97) == ctx.g2_mul(gen2 * m, sc2, 97)
    assert ctx.g2_mul_fixed(gen2, sc2[:32 * 12], 192) == oracle_port.g2_mul(gen2 * 12, sc2[:32 * 12], 192, 4)
    off2 = cat(g["offsubgroup_points"])[:192]
    assert ctx.g2_mul_fixed(off2, sc2[-32 * 14:], 192) == oracle_port.g2_mul(off2 * 14, sc2[-32 * 14:], 192)


def test_in_subgroup_flag(ctx, oracle_port):
    """C12381_F_IN_SUBGROUP skips the [r]phi(P) / [r]psi^i(Q) side paths (pair_BLS12381.cpp:793-805, 896-914, 868-871):
    identical results on subgroup points for short scalars, the default entry stays exact off the subgroup, and the
    flagged entry returns the plain endomorphism combination there (documented divergence)."""
    from crypto12381_amd.capi import F_IN_SUBGROUP, E_ARG
    g = golden("g1")
    g1 = bytes.fromhex(g["generator"])
    n = 300
    pts = ctx.g1_mul(g1 * n, scalars(3101, n), 96)
    short = b"".join((prng(3102, i) % (1 << (8 * (1 + i % 16)))).to_bytes(32, "big") for i in range(n))      # 8 .. 128-bit scalars
    exp = oracle_port.g1_mul(pts, short, 96, 8)
    assert ctx.g1_mul(pts, short, 96) == exp
    assert ctx.g1_mul_flags(pts, short, 96, F_IN_SUBGROUP) == exp
    assert ctx.g1_mul_flags(pts, short, 49, F_IN_SUBGROUP) == oracle_port.g1_mul(pts, short, 49, 8)
    full = scalars(3103, n)
    assert ctx.g1_mul_flags(pts, full, 96, F_IN_SUBGROUP) == oracle_port.g1_mul(pts, full, 96, 8)
    g2g = golden("g2")
    g2 = bytes.fromhex(g2g["generator"])
    m = 120
    q = ctx.g2_mul(g2 * m, scalars(3104, m), 192)
    short2 = short[:32 * m]
    e2 = oracle_port.g2_mul(q, short2, 192, 8)
    assert ctx.g2_mul(q, short2, 192) == e2 and ctx.g2_mul_flags(q, short2, 192, F_IN_SUBGROUP) == e2
    assert ctx.g2_mul_flags(q, short2, 97, F_IN_SUBGROUP) == oracle_port.g2_mul(q, short2, 97, 8)
    # off the subgroup: the default entry equals the reference (golden), the flagged one drops exactly the [r]-terms
    op, osc = cat(g["offsubgroup_small_points"]), cat(g["offsubgroup_small_scalars"])
    assert ctx.g1_mul(op, osc, 96) == cat(g["offsubgroup_small_mul96"])
    assert ctx.g1_mul_flags(op, osc, 96, F_IN_SUBGROUP) != cat(g["offsubgroup_small_mul96"])
    # unknown flag bits are argument errors
    import ctypes
    from crypto12381_amd.capi import _p
    out = ctypes.create_string_buffer(96)
    assert ctx.lib.c12381_g1_mul_batch_flags(ctx.h, 1, _p(pts[:96]), _p(short[:32]), _p(out), 96, 2) == E_ARG


def test_sum_of_products_is_ecp_muln_on_every_point(ctx, oracle_port):
    """The seam's sum_of_products (-> ECP_muln ecp_BLS12381.cpp:1112-1148) sums TRUE multiples; the header-level product
    (c12381_g1_msm) goes through multiply()'s GLV form.  Equal on G1, different off it — both pinned by reference vectors."""
    g = golden("g1")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    n = g["msm_n"]
    assert ctx.g1_sum_of_products(pts[:96 * n], sc[:32 * n], 49).hex() == g["sum_of_products49"] == g["msm49"]
    off, osc = cat(g["offsubgroup_points"]), cat(g["offsubgroup_scalars"])
    assert ctx.g1_sum_of_products(off, osc, 49).hex() == g["offsubgroup_sum_of_products49"]
    assert ctx.g1_msm(off, osc, 49).hex() == g["offsubgroup_msm49"] != g["offsubgroup_sum_of_products49"]
    assert ctx.g1_sum_of_products(off, osc, 96) == oracle_port.g1_sum_of_products(off, osc, 96)
    # sizes across the tree-sum levels, infinity terms, zero scalars
    m = 150
    g1 = bytes.fromhex(g["generator"])
    P = bytearray(ctx.g1_mul(g1 * m, scalars(3201, m), 96)); P[96 * 3:96 * 4] = bytes(96)
    K = bytearray(scalars(3202, m)); K[32 * 5:32 * 6] = bytes(32)
    assert ctx.g1_sum_of_products(bytes(P), bytes(K), 49) == oracle_port.g1_msm(bytes(P), bytes(K), 49, 8)
    assert ctx.g1_sum_of_products(b"", b"", 49) == bytes(49)


def test_g2_product(ctx, oracle_port):
    """c12381_g2_msm: the header's product over G2 points (g2_point.hpp:225-236, a chain of add) and Π q_i^{x_i}."""
    g2 = bytes.fromhex(golden("g2")["generator"])
    m = 131
    Q = bytearray(ctx.g2_mul(g2 * m, scalars(3301, m), 192)); Q[192 * 2:192 * 3] = bytes(192)
    Q = bytes(Q)
    acc = bytes(192)
    for i in range(m):
        acc = oracle_port.g2_add(acc, Q[192 * i:192 * i + 192], 192)
    assert ctx.g2_msm(Q, None, 192) == acc
    assert ctx.g2_msm(Q, None, 97) == oracle_port.g2_compress(acc)
    K = scalars(3302, m)
    T = oracle_port.g2_mul(Q, K, 192, 8)
    acc = bytes(192)
    for i in range(m):
        acc = oracle_port.g2_add(acc, T[192 * i:192 * i + 192], 192)
    assert ctx.g2_msm(Q, K, 192) == acc
    assert ctx.g2_msm(b"", None, 97) == bytes(97)
    assert ctx.g2_msm(Q[:192], K[:32], 192) == T[:192]


def test_g1_sum_of_points(ctx, oracle_port):
    """c12381_g1_sum: the plain sum of affine points (a chain of add(point1&, point1&); the combine step of a sharded product) against
    the oracle's product with unit scalars — ragged sizes around the reduction levels, infinity terms, P + (-P), the empty sum, and a
    point that is not on the curve (left out, reported)."""
    from crypto12381_amd.capi import C12381Error, E_POINT
    g = golden("g1")
    gen = bytes.fromhex(g["generator"])
    m = 4200
    P = ctx.g1_mul(gen * m, scalars(1301, m), 96)
    one = (1).to_bytes(32, "big")
    for n in (1, 2, 3, 8, 63, 64, 65, 4095, 4096, 4097, m):
        assert ctx.g1_sum(P[:96 * n], 96) == oracle_port.g1_msm(P[:96 * n], one * n, 96, 8), n
    assert ctx.g1_sum(b"", 96) == bytes(96) and ctx.g1_sum(b"", 49) == bytes(49)
    neg = oracle_port.g1_mul(P[:96], (R - 1).to_bytes(32, "big"), 96)
    assert ctx.g1_sum(P[:96] + neg, 96) == bytes(96)                                    # P + (-P)
    assert ctx.g1_sum(P[:96] + bytes(96) + P[96:192] + bytes(96), 49) == oracle_port.g1_msm(P[:192], one * 2, 49)
    bad = bytearray(P[:96 * 5]); bad[96 * 2 + 95] ^= 1
    with pytest.raises(C12381Error) as e:
        ctx.g1_sum(bytes(bad), 96)
    assert e.value.code == E_POINT
