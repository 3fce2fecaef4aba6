"""GPU parity tests for G2 and the pairing (run with -m gpu): HIP path through the C ABI vs golden
vectors from the reference and vs the CPU oracle on fresh inputs."""
import time

import pytest

from util import R, cat, golden, prng, scalars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from crypto12381_amd import Context
    c = Context(0)
    yield c
    c.close()


def test_g2_golden(ctx):
    g = golden("g2")
    pts, sc = cat(g["points"]), cat(g["scalars"])
    assert ctx.g2_mul(pts, sc, 97) == cat(g["mul97"])
    assert ctx.g2_mul(pts, sc, 192) == cat(g["mul192"])
    assert ctx.g2_add(cat(g["add_a"]), cat(g["add_b"]), 192) == cat(g["add192"])


def test_g2_mul_vs_oracle_random(ctx, oracle_port):
    n = 300
    g2 = bytes.fromhex(golden("g2")["generator"])
    pts = ctx.g2_mul(g2 * n, scalars(401, n), 192)
    sc = scalars(402, n, 1 << 256)
    assert ctx.g2_mul(pts, sc, 97) == oracle_port.g2_mul(pts, sc, 97, 16)


def test_pairing_golden(ctx):
    g = golden("pairing")
    g1, g2 = cat(g["g1"]), cat(g["g2"])
    gt = ctx.pair(g1, g2)
    assert gt == cat(g["gt"])
    assert list(ctx.pair_eq(cat(g["eq_a1"]), cat(g["eq_a2"]), cat(g["eq_b1"]), cat(g["eq_b2"]))) == g["eq"]
    assert list(ctx.pair_eq(cat(g["eq2_a1"]), cat(g["eq2_a2"]), cat(g["eq2_b1"]), cat(g["eq2_b2"]))) == g["eq2"]


def test_config1_bilinearity_on_gpu(ctx):
    """BASELINE configs[0] with the reference's own seeded inputs: e(P^x, Q^y) computed on the GPU."""
    g = golden("config1_bilinearity")
    sc = cat(g["scalars"])
    x, y = sc[64:96], sc[96:128]
    Pp, Qq = bytes.fromhex(g["P"]), bytes.fromhex(g["Q"])
    lhs = ctx.pair(ctx.g1_mul(Pp, x, 96), ctx.g2_mul(Qq, y, 192))
    assert lhs.hex() == g["pair_Px_Qy"]


def test_pairing_vs_oracle_and_bilinearity(ctx, oracle_port):
    n = 512
    g1 = bytes.fromhex(golden("g1")["generator"])
    g2 = bytes.fromhex(golden("g2")["generator"])
    P = ctx.g1_mul(g1 * n, scalars(411, n), 96)
    Q = ctx.g2_mul(g2 * n, scalars(412, n), 192)
    gt = ctx.pair(P, Q)
    m = 48
    assert gt[:576 * m] == oracle_port.pair(P[:96 * m], Q[:192 * m], 16)
    # bilinearity over the whole batch: e(xP, Q) == e(P, xQ)
    xs = scalars(413, n)
    ok = ctx.pair_eq(ctx.g1_mul(P, xs, 96), Q, P, ctx.g2_mul(Q, xs, 192))
    assert ok == b"\x01" * n
    # and a perturbed batch is rejected everywhere
    ok = ctx.pair_eq(ctx.g1_mul(P, xs, 96), Q, P, ctx.g2_mul(Q, scalars(414, n), 192))
    assert ok == b"\x00" * n


def test_pairing_full_size_2_16(ctx, oracle_port):
    """BASELINE configs[2] size: 2^16 pairings; sampled lanes vs the oracle + the pairing-product check
    prod e(P_i, Q)^(k_i) == e(sum k_i P_i, Q) evaluated through the equality kernel on a folded batch."""
    n = 1 << 16
    g1 = bytes.fromhex(golden("g1")["generator"])
    g2 = bytes.fromhex(golden("g2")["generator"])
    m = 1 << 10
    P = ctx.g1_mul(g1 * m, scalars(421, m), 96) * (n // m)
    Q = ctx.g2_mul(g2 * m, scalars(422, m), 192)
    Q = b"".join(Q[192 * ((7 * i) % m):192 * ((7 * i) % m) + 192] for i in range(n))
    t0 = time.time()
    gt = ctx.pair(P, Q)
    dt = time.time() - t0
    print("2^16 pairings incl. PCIe: %.3f s" % dt)
    idx = [0, 1, n - 1] + [prng(423, i, 4) % n for i in range(29)]
    sp = b"".join(P[96 * i:96 * i + 96] for i in idx)
    sq = b"".join(Q[192 * i:192 * i + 192] for i in idx)
    exp = oracle_port.pair(sp, sq, 16)
    assert b"".join(gt[576 * i:576 * i + 576] for i in idx) == exp


def test_pairing_of_curve_points_outside_the_subgroups(ctx, oracle_port):
    """from_bytes checks the curve equation only (ecp_BLS12381.cpp:495-545, ecp2_BLS12381.cpp:225-266), and PAIR_ate / PAIR_fexp run on
    whatever they are given: Miller values and pairings of random curve points (decoded from random x, almost never in G1 / G2)
    have to be the reference's — this is where a formula that is only right on the subgroup would show (addition steps, lines)."""
    m = 96
    c1 = b"".join(bytes([2 + (i & 1)]) + (prng(451, i, 48) % (1 << 381)).to_bytes(48, "big") for i in range(4 * m))
    c2 = b"".join(bytes([2 + (i & 1)]) + (prng(452, i, 48) % (1 << 381)).to_bytes(48, "big") + (prng(453, i, 48) % (1 << 381)).to_bytes(48, "big")
                  for i in range(4 * m))
    p1, s1 = ctx.g1_decompress(c1)
    p2, s2 = ctx.g2_decompress(c2)
    P = b"".join(p1[96 * i:96 * i + 96] for i in range(4 * m) if s1[i] == 1)[:96 * m]
    Q = b"".join(p2[192 * i:192 * i + 192] for i in range(4 * m) if s2[i] == 1)[:192 * m]
    n = min(len(P) // 96, len(Q) // 192)
    assert n >= 64
    P, Q = P[:96 * n], Q[:192 * n]
    assert ctx.miller(P, Q) == oracle_port.miller(P, Q)
    gt = ctx.pair(P, Q)
    assert gt == oracle_port.pair(P, Q, 16)
    # the table-driven kernels (normalised lines) and the product kernel on the same points
    for k in (0, 1):
        q = Q[192 * k:192 * k + 192]
        assert ctx.pair_fixed_g2(P, q) == oracle_port.pair(P, q * n, 16)
    prod = ctx.pair_product(P + P[96:] + P[:96], Q + Q, 2)
    assert prod == ctx.gt_op("mul", gt, oracle_port.pair(P[96:] + P[:96], Q, 16))


def test_decompress_golden(ctx):
    g = golden("g1")
    out, st = ctx.g1_decompress(cat(g["compressed"]))
    assert list(st) == g["decompress_status"] and out == cat(g["decompressed"])
    g = golden("g2")
    out, st = ctx.g2_decompress(cat(g["compressed"]))
    assert list(st) == g["decompress_status"] and out == cat(g["decompressed"])


def test_decompress_roundtrip_vs_oracle(ctx, oracle_port):
    n = 2000
    g1 = bytes.fromhex(golden("g1")["generator"])
    pts = ctx.g1_mul(g1 * n, scalars(431, n), 96)
    comp = ctx.g1_mul(g1 * n, scalars(431, n), 49)
    assert comp == oracle_port.g1_compress(pts)
    out, st = ctx.g1_decompress(comp)
    assert st == b"\x01" * n and out == pts
    # random x: about half are not on the curve; statuses and points must match the oracle lane by lane
    rnd = b"".join(bytes([2 + (i & 1)]) + (prng(432, i, 48) % (1 << 381)).to_bytes(48, "big") for i in range(n))
    out, st = ctx.g1_decompress(rnd)
    eo, es = oracle_port.g1_decompress(rnd)
    assert st == es and out == eo and 0 < sum(st) < n
    g2 = bytes.fromhex(golden("g2")["generator"])
    m = 500
    q = ctx.g2_mul(g2 * m, scalars(433, m), 192)
    qc = ctx.g2_mul(g2 * m, scalars(433, m), 97)
    out, st = ctx.g2_decompress(qc)
    assert st == b"\x01" * m and out == q
    rnd2 = b"".join(bytes([2 + (i & 1)]) + (prng(434, i, 48) % (1 << 381)).to_bytes(48, "big") + (prng(435, i, 48) % (1 << 381)).to_bytes(48, "big")
                    for i in range(m))
    out, st = ctx.g2_decompress(rnd2)
    eo, es = oracle_port.g2_decompress(rnd2)
    assert st == es and out == eo and 0 < sum(st) < m


def test_gt_ops_and_split_pairing(ctx, oracle_port):
    g = golden("pairing")
    gt = cat(g["gt"])
    gta, gtb = gt[:576 * 4], gt[576 * 4:]
    assert ctx.gt_op("mul", gta, gtb) == cat(g["gt_mul"])
    assert ctx.gt_op("conj", gta) == cat(g["gt_conj"])
    assert ctx.gt_op("pow", gta, cat(g["gt_pow_exp"])) == cat(g["gt_pow"])
    g1, g2 = cat(g["g1"]), cat(g["g2"])
    m = ctx.miller(g1, g2)
    assert m == oracle_port.miller(g1, g2)
    assert ctx.fexp(m) == gt
    one = ctx.gt_op("mul", gta, ctx.gt_op("conj", gta))     # unitary: a * conj(a) = 1
    assert ctx.gt_is_unity(one) == b"\x01" * 4 and ctx.gt_is_unity(gta)[:3] == b"\x00" * 3
    # pow() on inputs that are NOT unitary (Miller values): the reference's unitary squarings make the result a function of the
    # exact operation sequence (fp12_BLS12381.cpp:736-774) — it has to be reproduced, not "corrected"; exponents of every
    # length class incl. 0, 1, 2, 3 and the top bit set (a wavefront holds 21 different ones)
    nm = len(m) // 576
    reps = 6
    exps = [0, 1, 2, 3, 4, 5, (1 << 255) + 12345, (1 << 256) - 1, R - 1, R, 1 << 64, (1 << 128) - 1] + [prng(436, i, 32) for i in range(nm * reps - 12)]
    eb = b"".join(int(v).to_bytes(32, "big") for v in exps)
    assert ctx.gt_op("pow", m * reps, eb) == oracle_port.gt_op("pow", m * reps, eb)
    assert ctx.gt_op("pow", gt * reps, eb) == oracle_port.gt_op("pow", gt * reps, eb)


def test_gt_power_routes(ctx, oracle_port):
    """gt3_op_kernel takes the 4-bit windowed ladder for a wavefront (21 elements) whose bases are all in the cyclotomic subgroup and the
    reference's own digit sequence otherwise: wavefronts of either kind and mixed ones in one batch, zero bases, edge exponents — every element
    against the oracle; then the same mixture as a batch of 4097 groups, which takes gt3_pow_queue_kernel (more than one machine round:
    whole groups on the grid's wavefronts, the rest as five queued tasks per group), both ladders in queued and in whole groups."""
    g = golden("pairing")
    gt = cat(g["gt"])
    ng = len(gt) // 576
    mil = ctx.miller(cat(g["g1"]), cat(g["g2"]))           # not in the subgroup
    exps = [0, 1, 2, 15, 16, 17, 255, R - 1, R, R + 1, (1 << 256) - 1, 1 << 255, (1 << 255) + 1, 0x0f << 252, 0xf0f0f0f0 << 100] + [prng(437, i, 32) for i in range(48)]
    bases = []
    for i in range(63):                                    # three wavefronts: members only | one Miller value in the middle | zero and members
        if i == 21 + 9:
            bases.append(mil[:576])
        elif i == 42 + 4:
            bases.append(bytes(576))
        else:
            bases.append(gt[576 * (i % ng):576 * (i % ng) + 576])
    a = b"".join(bases)
    e = b"".join(int(v).to_bytes(32, "big") for v in exps)
    want = oracle_port.gt_op("pow", a, e)
    assert ctx.gt_op("pow", a, e) == want
    big = 4096 * 21 + 100                                  # 4097 groups: 2048 of them whole, the others through the queue
    reps = big // 63 + 1
    got = ctx.gt_op("pow", (a * reps)[:576 * big], (e * reps)[:32 * big])
    assert got == (want * reps)[:576 * big]


def test_pair_fixed_g2(ctx, oracle_port):
    """One G2 argument for the batch: table-driven Miller loop, identical GT bytes to the general entry point."""
    g = golden("pairing")
    g1s, g2s = cat(g["g1"]), cat(g["g2"])
    n = len(g1s) // 96
    for k in (0, 5, 7):                                     # an ordinary point, the generator, infinity (row 7 of the golden set)
        q = g2s[192 * k:192 * k + 192]
        assert ctx.pair_fixed_g2(g1s, q) == ctx.pair(g1s, q * n)
    assert ctx.pair_fixed_g2(g1s, g2s[:192]) == oracle_port.pair(g1s, g2s[:192] * n, 4)
    off = cat(golden("g2")["offsubgroup_points"])[:192]     # on the twist, outside G2: still the same lines
    assert ctx.pair_fixed_g2(g1s, off) == ctx.pair(g1s, off * n)
    m = 5000                                                # ragged size, several work-queue groups
    p = (g1s * (m // n + 1))[:96 * m]
    got = ctx.pair_fixed_g2(p, g2s[192:384])
    assert got[:576 * n] == ctx.pair(g1s, g2s[192:384] * n) and got[-576:] == ctx.pair(p[-96:], g2s[192:384])
    bad = g2s[:191] + bytes([g2s[191] ^ 1])
    assert ctx.pair_fixed_g2(g1s, bad, strict=False) == b"\xff" * (576 * n)
    assert ctx.pair_fixed_g2(b"", g2s[:192]) == b""


def test_pair_product_batch(ctx, oracle_port):
    """c12381_pair_product_batch: pair(a,b) * pair(c,d) [* pair(e,f)] with one joint Miller loop and one final exponentiation
    (liner_pair.hpp:291-303 -> pair_double_ate pair_BLS12381.cpp:508-626): equals the reference's pair2 golden, the product of
    single pairings through gt_op, and the reference's law tests (unit-tests/liner_pair.cpp:66-79, 92-103)."""
    from crypto12381_amd.capi import F_MILLER_ONLY
    g = golden("pairing")
    a1, a2, b1, b2 = cat(g["eq_a1"]), cat(g["eq_a2"]), cat(g["eq_b1"]), cat(g["eq_b2"])
    n = len(a1) // 96
    got = ctx.pair_product(a1 + b1, a2 + b2, 2)
    assert got == oracle_port.pair2(a1, a2, b1, b2)
    assert got == ctx.gt_op("mul", ctx.pair(a1, a2), ctx.pair(b1, b2))
    if "pair2" in g and len(cat(g["pair2"])) == 576 * n:
        assert got == cat(g["pair2"])
    # Miller-only form: the reference's pair_double_ate value (product of the two Miller values), then fexp gives the same GT
    m2 = ctx.pair_product(a1 + b1, a2 + b2, 2, F_MILLER_ONLY)
    assert m2 == ctx.gt_op("mul", ctx.miller(a1, a2), ctx.miller(b1, b2))
    assert ctx.fexp(m2) == got
    # k = 1 degenerates to the plain pairing, k = 3 is the triple product
    assert ctx.pair_product(a1, a2, 1) == ctx.pair(a1, a2)
    c1, c2 = cat(g["g1"])[:96 * n] if len(cat(g["g1"])) >= 96 * n else (cat(g["g1"]) * n)[:96 * n], (cat(g["g2"]) * n)[:192 * n]
    tri = ctx.pair_product(a1 + b1 + c1, a2 + b2 + c2, 3)
    assert tri == ctx.gt_op("mul", got, ctx.pair(c1, c2))
    # infinity arguments contribute 1 (pair_BLS12381.cpp:532-541; G2 infinity: unit-tests/liner_pair.cpp:28-40)
    z1, z2 = bytes(96 * n), bytes(192 * n)
    assert ctx.pair_product(a1 + z1, a2 + b2, 2) == ctx.pair(a1, a2)
    assert ctx.pair_product(a1 + b1, a2 + z2, 2) == ctx.pair(a1, a2)
    # a ragged batch across wavefront groups against the oracle
    m = 47
    g1 = bytes.fromhex(golden("g1")["generator"]); g2 = bytes.fromhex(golden("g2")["generator"])
    P = ctx.g1_mul(g1 * (2 * m), scalars(4401, 2 * m), 96)
    Q = ctx.g2_mul(g2 * (2 * m), scalars(4402, 2 * m), 192)
    assert ctx.pair_product(P, Q, 2) == oracle_port.pair2(P[:96 * m], Q[:192 * m], P[96 * m:], Q[192 * m:])
