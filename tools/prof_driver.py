#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 counter passes: one 2^18 G1 scalar-mul launch and one 2^16 pairing
launch through the C ABI (host-pointer entry points).  Usage: python3 tools/prof_driver.py [g1|pair|both|split]
(split: the Miller loops alone (miller3_kernel), the final exponentiations alone (fexp3_kernel) and the queue pairing, 2^16 each —
where the pairing's private-memory traffic comes from; C12381_PAIR_QUEUE=0 selects the plain-grid pairing kernel)"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tools.libsel  # noqa: E402,F401  (C12381_LIB -> capi.use_library)
from crypto12381_amd import Context  # noqa: E402

G1 = bytes.fromhex("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
                   "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1")
G2 = bytes.fromhex("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
                   "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
                   "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"
                   "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801")


def sc(seed, n):
    out = bytearray()
    i = 0
    while len(out) < 32 * n:
        out += hashlib.sha512(b"prof|%d|%d" % (seed, i)).digest()
        i += 1
    return bytes(out[:32 * n])


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    c = Context(0)
    if what in ("g1", "both"):
        n = 1 << 18
        pts = c.g1_mul(G1 * 4096, sc(1, 4096), 96) * (n // 4096)
        c.g1_mul(pts, sc(2, n), 96)
    if what in ("pair", "both"):
        n = 1 << 16
        p = c.g1_mul(G1 * 1024, sc(3, 1024), 96) * (n // 1024)
        q = c.g2_mul(G2 * 1024, sc(4, 1024), 192) * (n // 1024)
        c.pair(p, q)
    if what == "split":
        n = 1 << 16
        p = c.g1_mul(G1 * 1024, sc(3, 1024), 96) * (n // 1024)
        q = c.g2_mul(G2 * 1024, sc(4, 1024), 192) * (n // 1024)
        f = c.miller(p, q)
        c.fexp(f)
        c.pair(p, q)
    c.close()


if __name__ == "__main__":
    main()
