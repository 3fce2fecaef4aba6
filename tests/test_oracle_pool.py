"""oracle/pool.py (the CPU baseline's process pool, test infrastructure): every pooled call returns the rows the
single-threaded oracle returns, on ragged splits (units not a multiple of the workers), incl. the message-major
chunking of BBS+ blocks and the signature-major raw messages of the wire form."""
import sys
import os

sys.path.insert(0, os.path.dirname(__file__))
from util import R, golden, prng, scalars
from test_gpu_bbs import _setup, _sign


def test_pooled_calls_equal_the_single_threaded_oracle(oracle_port):
    from oracle.pool import OraclePool, OracleThreads
    orc = oracle_port
    n = 7
    g1 = bytes.fromhex(golden("g1")["generator"])
    g2 = bytes.fromhex(golden("g2")["generator"])
    sc = scalars(811, 3 * n)
    p1 = orc.g1_mul(g1 * n, sc[:32 * n], 96)
    q2 = orc.g2_mul(g2 * n, sc[32 * n:64 * n], 192)
    k = sc[64 * n:]
    with OraclePool("port", 3) as pool:
        assert pool.bounds(7) == [(0, 2), (2, 4), (4, 7)] and pool.bounds(2) == [(0, 1), (1, 2)]
        for w in (pool, OracleThreads(orc, 3)):
            assert w.g1_mul(p1, k, 96) == orc.g1_mul(p1, k, 96)
            assert w.g1_mul(p1, k, 49) == orc.g1_mul(p1, k, 49)
            assert w.g2_mul(q2, k, 192) == orc.g2_mul(q2, k, 192)
            assert w.pair(p1, q2) == orc.pair(p1, q2)
            m = orc.miller_t(p1, q2)
            assert w.miller_t(p1, q2) == m
            assert w.fexp_t(m) == orc.fexp_t(m) == orc.pair(p1, q2)
            assert w.g1_msm(p1, k, 96) == orc.g1_msm(p1, k, 96)
            assert w.g1_msm(p1, k, 49) == orc.g1_msm(p1, k, 49)
            # BBS+ from parsed values: 5 signatures over 2 blocks, lanes 1 and 3 carry another first block
            nmsg, ns = 2, 5
            G1p, G2p, h0, h, gamma, wk = _setup(orc, nmsg)
            A, X, Rr, M = b"", b"", b"", [b"", b""]
            for j in range(ns):
                msgs = [prng(820 + i, j) % R for i in range(nmsg)]
                x, r = prng(830, j) % R, prng(831, j) % R
                A += _sign(orc, G1p, h0, h, gamma, msgs, x, r)
                if j in (1, 3):
                    msgs[0] = (msgs[0] + 1) % R
                X += x.to_bytes(32, "big"); Rr += r.to_bytes(32, "big")
                for i in range(nmsg):
                    M[i] += msgs[i].to_bytes(32, "big")
            got = w.bbs_plus_verify(G1p, G2p, h0, h, wk, A, X, Rr, M[0] + M[1])
            assert got == orc.bbs_plus_verify(G1p, G2p, h0, h, wk, A, X, Rr, M[0] + M[1]) == b"\x01\x00\x01\x00\x01"
            # the wire form: 45-byte messages (two blocks), lane 2 altered
            msg_len = 45
            pp = orc.g1_compress(G1p) + orc.g2_compress(G2p) + orc.g1_compress(h0)
            h49, pk = orc.g1_compress(h), orc.g2_compress(wk)
            sigs, raw = b"", b""
            for j in range(ns):
                msg = bytes(prng(840, j * 64 + b, 1) for b in range(msg_len))
                units = orc.encode_to_zp(msg)
                ms = [int.from_bytes(units[32 * i:32 * i + 32], "big") for i in range(2)]
                x, r = prng(841, j) % R, prng(842, j) % R
                a = _sign(orc, G1p, h0, h, gamma, ms, x, r)
                sigs += orc.g1_compress(a) + bytes(16) + x.to_bytes(32, "big") + bytes(16) + r.to_bytes(32, "big")
                raw += (bytes([msg[0] ^ 1]) + msg[1:]) if j == 2 else msg
            got = w.bbs_plus_verify_wire(pp, h49, pk, sigs, raw, msg_len)
            assert got == orc.bbs_plus_verify_wire(pp, h49, pk, sigs, raw, msg_len) == b"\x01\x01\x00\x01\x01"
