"""The reference's own unit tests are algebraic-law / round-trip property tests on seeded random inputs.
This file restates them against the batched C ABI on the GPU, file by file:
  unit-tests/liner_pair.cpp  non-degeneracy :28-40, bilinearity :42-64, double pairing :66-79, pair == pair :81-90,
                             triple product :92-103, inverse laws :105-115, GT group / exponent laws :129-160,
                             GT byte round trips :162-181
  unit-tests/g1_point.cpp    group laws :18-49, scalar laws incl. 0, 1, -x :51-78, g^x h^y fused vs separate :80-98,
                             byte round trips incl. the identity :113-130, all-0xff is rejected :132-138
  unit-tests/g2_point.cpp    the same structure :18-118
  unit-tests/zp_number.cpp   field laws :35-212 (the batched helpers)
Every law is evaluated on a whole batch per call."""
import pytest

from util import R, golden, prng, scalars

pytestmark = pytest.mark.gpu
N = 24


@pytest.fixture(scope="module")
def ctx():
    from crypto12381_amd import Context
    c = Context(0)
    yield c
    c.close()


def _sc(seed, n=N):
    return scalars(seed, n)


def _ints(b):
    return [int.from_bytes(b[32 * i:32 * i + 32], "big") for i in range(len(b) // 32)]


def _cat_ints(v):
    return b"".join((x % R).to_bytes(32, "big") for x in v)


def _neg1(p):
    from util import P
    out = b""
    for i in range(len(p) // 96):
        q = p[96 * i:96 * i + 96]
        y = int.from_bytes(q[48:], "big")
        out += q if q == bytes(96) else q[:48] + ((P - y) % P).to_bytes(48, "big")
    return out


@pytest.fixture(scope="module")
def pts(ctx):
    g1 = bytes.fromhex(golden("g1")["generator"])
    g2 = bytes.fromhex(golden("g2")["generator"])
    return {"g1": g1, "g2": g2, "P": ctx.g1_mul_fixed(g1, _sc(1001), 96), "Pb": ctx.g1_mul_fixed(g1, _sc(1002), 96),
            "Q": ctx.g2_mul_fixed(g2, _sc(1003), 192), "Qb": ctx.g2_mul_fixed(g2, _sc(1004), 192)}


# ---------------------------------------------------------------- unit-tests/g1_point.cpp, g2_point.cpp
def test_g1_group_and_scalar_laws(ctx, pts):
    P, Pb = pts["P"], pts["Pb"]
    inf = bytes(96) * N
    assert ctx.g1_add(P, Pb, 96) == ctx.g1_add(Pb, P, 96)                                     # commutative
    Pc = ctx.g1_mul_fixed(pts["g1"], _sc(1005), 96)
    assert ctx.g1_add(ctx.g1_add(P, Pb, 96), Pc, 96) == ctx.g1_add(P, ctx.g1_add(Pb, Pc, 96), 96)   # associative
    assert ctx.g1_add(P, inf, 96) == P and ctx.g1_add(inf, P, 96) == P                         # identity
    assert ctx.g1_add(P, _neg1(P), 96) == inf                                                 # inverse
    x, y = _sc(1006), _sc(1007)
    xi, yi = _ints(x), _ints(y)
    assert ctx.g1_mul(P, bytes(32) * N, 96) == inf                                            # g^0
    assert ctx.g1_mul(P, (1).to_bytes(32, "big") * N, 96) == P                                # g^1
    assert ctx.g1_mul(P, _cat_ints([-a for a in xi]), 96) == _neg1(ctx.g1_mul(P, x, 96))       # g^(-x) = (g^x)^-1
    assert ctx.g1_mul(ctx.g1_mul(P, x, 96), y, 96) == ctx.g1_mul(P, _cat_ints([a * b for a, b in zip(xi, yi)]), 96)
    assert ctx.g1_add(ctx.g1_mul(P, x, 96), ctx.g1_mul(P, y, 96), 96) == ctx.g1_mul(P, _cat_ints([a + b for a, b in zip(xi, yi)]), 96)
    # g^x h^y fused (double_multiply / sum_of_products) vs separate
    for i in range(8):
        fused = ctx.g1_msm(P[96 * i:96 * i + 96] + Pb[96 * i:96 * i + 96], x[32 * i:32 * i + 32] + y[32 * i:32 * i + 32], 96)
        assert fused == ctx.g1_add(ctx.g1_mul(P[96 * i:96 * i + 96], x[32 * i:32 * i + 32], 96), ctx.g1_mul(Pb[96 * i:96 * i + 96], y[32 * i:32 * i + 32], 96), 96)
    # byte round trips, identity included; all-0xff is rejected
    comp = ctx.g1_mul(P + bytes(96), x + bytes(32), 49)
    dec, st = ctx.g1_decompress(comp)
    assert set(st) == {1} and dec == ctx.g1_mul(P, x, 96) + bytes(96)
    assert comp[-49:] == bytes(49)
    _, st = ctx.g1_decompress(b"\xff" * 49)
    assert st == b"\x00"


def test_g2_group_and_scalar_laws(ctx, pts):
    Q, Qb = pts["Q"], pts["Qb"]
    inf = bytes(192) * N
    assert ctx.g2_add(Q, Qb, 192) == ctx.g2_add(Qb, Q, 192)
    Qc = ctx.g2_mul_fixed(pts["g2"], _sc(1008), 192)
    assert ctx.g2_add(ctx.g2_add(Q, Qb, 192), Qc, 192) == ctx.g2_add(Q, ctx.g2_add(Qb, Qc, 192), 192)
    assert ctx.g2_add(Q, inf, 192) == Q
    x, y = _sc(1009), _sc(1010)
    xi, yi = _ints(x), _ints(y)
    assert ctx.g2_mul(Q, bytes(32) * N, 192) == inf
    assert ctx.g2_mul(Q, (1).to_bytes(32, "big") * N, 192) == Q
    assert ctx.g2_add(ctx.g2_mul(Q, x, 192), ctx.g2_mul(Q, _cat_ints([-a for a in xi]), 192), 192) == inf
    assert ctx.g2_mul(ctx.g2_mul(Q, x, 192), y, 192) == ctx.g2_mul(Q, _cat_ints([a * b for a, b in zip(xi, yi)]), 192)
    assert ctx.g2_add(ctx.g2_mul(Q, x, 192), ctx.g2_mul(Q, y, 192), 192) == ctx.g2_mul(Q, _cat_ints([a + b for a, b in zip(xi, yi)]), 192)
    comp = ctx.g2_mul(Q + bytes(192), x + bytes(32), 97)
    dec, st = ctx.g2_decompress(comp)
    assert set(st) == {1} and dec == ctx.g2_mul(Q, x, 192) + bytes(192)


# ---------------------------------------------------------------- unit-tests/liner_pair.cpp
def test_pairing_laws(ctx, pts):
    P, Pb, Q, Qb = pts["P"], pts["Pb"], pts["Q"], pts["Qb"]
    one = (ctx.gt_op("mul", ctx.pair(P[:96], Q[:192]), ctx.gt_op("conj", ctx.pair(P[:96], Q[:192]))))
    e = ctx.pair(P, Q)
    assert ctx.gt_is_unity(e) == bytes(N)                                                     # non-degenerate
    assert ctx.gt_is_unity(ctx.pair(bytes(96) * N, Q)) == b"\x01" * N                         # e(1, Q) = 1
    assert ctx.gt_is_unity(ctx.pair(P, bytes(192) * N)) == b"\x01" * N                        # e(P, 1) = 1
    x, y = _sc(1011), _sc(1012)
    xi, yi = _ints(x), _ints(y)
    ex = ctx.gt_op("pow", e, x)
    assert ctx.pair(ctx.g1_mul(P, x, 96), Q) == ex                                            # bilinear in the G1 exponent
    assert ctx.pair(P, ctx.g2_mul(Q, x, 192)) == ex                                           # ... in the G2 exponent
    assert ctx.pair(ctx.g1_mul(P, x, 96), ctx.g2_mul(Q, y, 192)) == ctx.gt_op("pow", e, _cat_ints([a * b for a, b in zip(xi, yi)]))
    # product of two pairings: one shared final exponentiation vs two single pairings; pair == pair
    m1, m2 = ctx.miller(P, Q), ctx.miller(Pb, Qb)
    assert ctx.fexp(ctx.gt_op("mul", m1, m2)) == ctx.gt_op("mul", e, ctx.pair(Pb, Qb))
    assert ctx.pair_eq(ctx.g1_mul(P, x, 96), Q, P, ctx.g2_mul(Q, x, 192)) == b"\x01" * N
    assert ctx.pair_eq(P, Q, Pb, Q) == bytes(N)
    # triple product and inverse laws
    e3 = ctx.gt_op("mul", ctx.gt_op("mul", e, ctx.pair(Pb, Q)), ctx.pair(P, Qb))
    assert e3 == ctx.gt_op("mul", ctx.pair(ctx.g1_add(P, Pb, 96), Q), ctx.pair(P, Qb))
    assert ctx.pair(_neg1(P), Q) == ctx.gt_op("conj", e)                                      # e(P^-1, Q) = e(P, Q)^-1
    assert ctx.gt_is_unity(ctx.gt_op("mul", e, ctx.gt_op("conj", e))) == b"\x01" * N
    # GT exponent laws incl. exponents 0 and 1; byte round trip of the identity
    assert ctx.gt_op("pow", e, bytes(32) * N) == one * N
    assert ctx.gt_op("pow", e, (1).to_bytes(32, "big") * N) == e
    assert ctx.gt_op("mul", ex, ctx.gt_op("pow", e, y)) == ctx.gt_op("pow", e, _cat_ints([a + b for a, b in zip(xi, yi)]))
    assert ctx.gt_op("pow", ex, y) == ctx.gt_op("pow", e, _cat_ints([a * b for a, b in zip(xi, yi)]))
    assert ctx.gt_is_unity(one) == b"\x01" and one == ctx.gt_op("mul", one, one)


# ---------------------------------------------------------------- unit-tests/zp_number.cpp (batched helpers)
def test_zp_field_laws(ctx):
    a, b, c = _sc(1013, 64), _sc(1014, 64), _sc(1015, 64)
    mul, add, sub = (lambda u, v: ctx.zp_op("mul", u, v)), (lambda u, v: ctx.zp_op("add", u, v)), (lambda u, v: ctx.zp_op("sub", u, v))
    assert mul(a, b) == mul(b, a) and add(a, b) == add(b, a)
    assert mul(mul(a, b), c) == mul(a, mul(b, c))
    assert mul(a, add(b, c)) == add(mul(a, b), mul(a, c))                                     # distributive
    assert sub(add(a, b), b) == a
    assert add(a, ctx.zp_op("neg", a)) == bytes(32) * 64
    inv = ctx.zp_op("inv", a)
    assert mul(a, inv) == (1).to_bytes(32, "big") * 64
    assert ctx.zp_op("inv", bytes(32)) == bytes(32)                                           # inverse(0) = 0 (zp_number.cpp:76)
    assert ctx.zp_inner_product(a, b) == _cat_ints([sum(x * y for x, y in zip(_ints(a), _ints(b)))])
