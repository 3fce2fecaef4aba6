"""GPU parity, SURVEY.md 8(f) rows 3 and 4: hash-to-G1 (G1Point::from_hash from the digest on) and the Zp batch
helpers, through the C ABI, against the reference's golden vectors, the oracle and Python integers."""
import hashlib

import pytest

from util import R, cat, golden, prng, scalars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from crypto12381_amd import Context
    c = Context(0)
    yield c
    c.close()


def test_hash_to_g1_golden(ctx):
    g = golden("hash_zp")
    d = cat(g["digests"])
    assert ctx.g1_from_hash(d, 96) == cat(g["g1_from_hash_96"])
    assert ctx.g1_from_hash(d, 49) == cat(g["g1_from_hash_49"])
    assert ctx.g1_from_hash(b"", 96) == b""


def test_hash_to_g1_batch_vs_oracle(ctx, oracle_port):
    n = 3000                                   # ragged: not a multiple of the block or the finish chunk
    d = b"".join(hashlib.sha3_512(b"gpu h2c|%d" % i).digest() for i in range(n))
    got = ctx.g1_from_hash(d, 49)
    m = 96
    assert got[:49 * m] == oracle_port.g1_from_hash(d[:64 * m], 49)
    assert got[-49 * 8:] == oracle_port.g1_from_hash(d[-64 * 8:], 49)
    # the images are in the r-torsion: [r]P = infinity, i.e. [r - 1]P + P = infinity
    pts = ctx.g1_from_hash(d[:64 * 64], 96)
    rm1 = (R - 1).to_bytes(32, "big") * 64
    assert ctx.g1_add(ctx.g1_mul(pts, rm1, 96), pts, 96) == bytes(96 * 64)


def test_map_to_point_vs_reference(ctx, oracle_ref):
    """map_to_point alone (no cofactor): points of E, generally outside the r-torsion; [1 - x] of them is from_hash"""
    u = b"".join((prng(51, i) % (1 << 384)).to_bytes(48, "big") for i in range(20)) + (5).to_bytes(48, "big")
    got = ctx.g1_map_to_point(u)
    assert got == oracle_ref.g1_map_to_point(u)
    # multiply_cofactor alone: the plain multiple [1 - x]P; composed with map_to_point it is from_hash
    assert ctx.g1_clear_cofactor(got[:96 * 20]) == oracle_ref.g1_from_hash(b"".join(bytes(16) + u[48 * i:48 * i + 48] for i in range(20)), 96)
    # `multiply` by the same scalar is PAIR_G1mul: off the subgroup it is NOT the plain multiple, and we match that too
    cof = (0xd201000000010001).to_bytes(32, "big") * 21
    assert ctx.g1_mul(got, cof, 96) == oracle_ref.g1_mul(got, cof, 96)
    assert ctx.g1_clear_cofactor(bytes(96)) == bytes(96)


def test_zp_golden_and_ints(ctx):
    g = golden("hash_zp")
    a, b = cat(g["zp_a"]), cat(g["zp_b"])
    for op in ("mul", "add", "sub", "neg", "inv"):
        assert ctx.zp_op(op, a, b if op in ("mul", "add", "sub") else None) == cat(g["zp_" + op]), op
    assert ctx.zp_from_hash(cat(g["digests"])) == cat(g["zp_from_hash"])
    n = 5001
    x, y = scalars(41, n, 1 << 256), scalars(42, n, 1 << 256)
    xi = [int.from_bytes(x[32 * i:32 * i + 32], "big") % R for i in range(n)]
    yi = [int.from_bytes(y[32 * i:32 * i + 32], "big") % R for i in range(n)]
    assert ctx.zp_op("mul", x, y) == b"".join((p * q % R).to_bytes(32, "big") for p, q in zip(xi, yi))
    inv = ctx.zp_op("inv", x)
    assert ctx.zp_op("mul", inv, x) == (1).to_bytes(32, "big") * n
    assert ctx.zp_inner_product(x, y) == (sum(p * q for p, q in zip(xi, yi)) % R).to_bytes(32, "big")
    assert ctx.zp_inner_product(x) == (sum(xi) % R).to_bytes(32, "big")
    assert ctx.zp_inner_product(x[:32], y[:32]) == (xi[0] * yi[0] % R).to_bytes(32, "big")
    assert ctx.zp_inner_product(b"", b"") == bytes(32)


def test_zp_batch_inverse_with_zeros(ctx):
    """The inversion entry runs a simultaneous inversion over runs of 16 strided elements: zeros (inverse(0) = 0, also
    as r and 2r) anywhere in a run — first, last, several, a whole run — must not disturb their neighbours; sizes that
    leave the last runs short; exact values against Python."""
    for n in (1, 15, 16, 17, 1000, 4099):
        x = bytearray(scalars(61, n, 1 << 256))
        T = (n + 15) // 16
        zero_at = {0, n - 1, n // 2, T, 2 * T, min(n - 1, 3)} | ({j * T + 1 for j in range(16)} if n > 400 else set())
        for k, i in enumerate(sorted(z for z in zero_at if 0 <= z < n)):
            x[32 * i:32 * i + 32] = ((0, R, 2 * R)[k % 3]).to_bytes(32, "big")
        x = bytes(x)
        exp = b"".join(pow(int.from_bytes(x[32 * i:32 * i + 32], "big") % R, R - 2, R).to_bytes(32, "big") for i in range(n))
        assert ctx.zp_op("inv", x) == exp, n
