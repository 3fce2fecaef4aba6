"""BASELINE configs[3] and configs[4] at their full sizes through size-independent properties (the oracle cannot redo
2^22 terms or 2^18 verifications in seconds): every P_i is a known multiple of the generator, so the MSM must equal
g^(sum s_i k_i); signatures produced by the batch signer must all verify, corrupted lanes must fail exactly where they
were corrupted, and sampled lanes are re-evaluated by the oracle."""
import numpy as np
import pytest

from util import R, golden, scalars

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from crypto12381_amd import Context
    c = Context(0)
    yield c
    c.close()


def _rand_scalars(seed, n):
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    a[:, 0] &= 0x3f                                          # < 2^254 < r: already reduced, sums stay simple
    return a


def test_msm_full_size_2_22(ctx, oracle_port):
    n = 1 << 22
    gen = bytes.fromhex(golden("g1")["generator"])
    s = _rand_scalars(11, n)
    k = _rand_scalars(12, n)
    pts = ctx.g1_mul_fixed(gen, s.tobytes(), 96)             # P_i = g^{s_i}
    got = ctx.g1_msm(pts, k.tobytes(), 49)
    # sum s_i k_i mod r with exact integer arithmetic on 64-bit limbs (object arrays stay exact)
    tot = 0
    sb, kb = s.tobytes(), k.tobytes()
    for i in range(n):
        tot += int.from_bytes(sb[32 * i:32 * i + 32], "big") * int.from_bytes(kb[32 * i:32 * i + 32], "big")
    assert got == oracle_port.g1_mul(gen, (tot % R).to_bytes(32, "big"), 49)
    # the same product in two halves (linearity) and with a permutation of the terms
    half = n // 2
    a = ctx.g1_msm(pts[:96 * half], kb[:32 * half], 96)
    b = ctx.g1_msm(pts[96 * half:], kb[32 * half:], 96)
    assert ctx.g1_add(a, b, 49) == got


def test_bbs_plus_full_size_2_18(ctx, oracle_port):
    orc = oracle_port
    n, nmsg = 1 << 18, 1
    g1 = bytes.fromhex(golden("g1")["generator"])
    g2 = bytes.fromhex(golden("g2")["generator"])
    gs = orc.g1_mul(g1 * 3, scalars(771, 3), 96)
    G1p, h0, h = gs[:96], gs[96:192], gs[192:]
    G2p = orc.g2_mul(g2, scalars(772, 1), 192)
    gamma = int.from_bytes(scalars(773, 1), "big")
    w = orc.g2_mul(G2p, gamma.to_bytes(32, "big"), 192)
    X, Rr, Mm = _rand_scalars(21, n).tobytes(), _rand_scalars(22, n).tobytes(), _rand_scalars(23, n).tobytes()
    A = ctx.bbs_plus_sign(G1p, h0, h, gamma.to_bytes(32, "big"), X, Rr, Mm)
    ok = ctx.bbs_plus_verify(G1p, G2p, h0, h, w, A, X, Rr, Mm)
    assert ok == b"\x01" * n
    # corrupt every 1000th message and every 777th x: exactly those lanes must fail
    m2, x2 = bytearray(Mm), bytearray(X)
    bad = set()
    for j in range(0, n, 1000):
        m2[32 * j + 31] ^= 1; bad.add(j)
    for j in range(5, n, 777):
        x2[32 * j + 31] ^= 1; bad.add(j)
    ok2 = ctx.bbs_plus_verify(G1p, G2p, h0, h, w, A, bytes(x2), Rr, bytes(m2))
    assert [j for j in range(n) if ok2[j] != 1] == sorted(bad)
    assert set(ok2) <= {0, 1}
    # sampled lanes against the oracle's evaluation of sign()
    for j in (0, 1, n // 2, n - 1):
        x = int.from_bytes(X[32 * j:32 * j + 32], "big")
        e = pow((gamma + x) % R, R - 2, R)
        B = orc.g1_msm(G1p + h0 + h, (1).to_bytes(32, "big") + Rr[32 * j:32 * j + 32] + Mm[32 * j:32 * j + 32], 96, 1)
        assert A[96 * j:96 * j + 96] == orc.g1_mul(B, e.to_bytes(32, "big"), 96)


def test_batches_streamed_over_two_contexts_overlap_and_stay_equal(ctx):
    """A caller that streams batches alternates two contexts (two HIP streams, own workspaces): the work-queue launches of
    consecutive batches then overlap on the device (the next batch's wavefronts enter as the previous one's leave).  2^16
    pairings (the queued route: 3 121 groups on 2 048 resident wavefronts), six launches in flight over two streams, no host
    wait between them: every output equals the output of the same batch run alone (which test_gpu_full_batch.py compares
    lane by lane with the compiled reference)."""
    import torch
    from crypto12381_amd import Context
    n = 1 << 16
    dev = torch.device("cuda", 0)
    g1 = bytes.fromhex(golden("g1")["generator"])
    g2 = bytes.fromhex(golden("g2")["generator"])

    def up(b):
        return torch.frombuffer(bytearray(b), dtype=torch.uint8).to(dev)
    p1 = up(ctx.g1_mul_fixed(g1, _rand_scalars(31, n).tobytes(), 96))
    q2 = up(ctx.g2_mul_fixed(g2, _rand_scalars(32, n).tobytes(), 192))
    alone = torch.empty(n * 576, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    ctx.pair_dev(n, p1.data_ptr(), q2.data_ptr(), alone.data_ptr())
    assert ctx.sync() == 0
    pairs = []
    for _ in range(2):
        c = Context(0)
        s = torch.cuda.Stream(device=dev)
        c.set_stream(s.cuda_stream)
        pairs.append((c, s))
    outs = [torch.zeros(n * 576, dtype=torch.uint8, device=dev) for _ in range(6)]
    torch.cuda.synchronize(dev)
    for i, o in enumerate(outs):
        pairs[i & 1][0].pair_dev(n, p1.data_ptr(), q2.data_ptr(), o.data_ptr())
    for c, _ in pairs:
        assert c.sync() == 0
    for i, o in enumerate(outs):
        assert torch.equal(o, alone), "streamed launch %d differs from the batch run alone" % i
    # Miller values and final exponentiations through the same two contexts, chained per context (stream order within a context)
    mil = [torch.zeros(n * 576, dtype=torch.uint8, device=dev) for _ in range(2)]
    fex = [torch.zeros(n * 576, dtype=torch.uint8, device=dev) for _ in range(2)]
    for i in range(2):
        pairs[i][0].miller_dev(n, p1.data_ptr(), q2.data_ptr(), mil[i].data_ptr())
    for i in range(2):
        pairs[i][0].gt_op_dev("fexp", n, mil[i].data_ptr(), None, fex[i].data_ptr())
    for c, _ in pairs:
        assert c.sync() == 0
    assert torch.equal(mil[0], mil[1]) and torch.equal(fex[0], alone) and torch.equal(fex[1], alone)
    for c, _ in pairs:
        c.close()
