"""Shared helpers for the test-suite: golden loader and the deterministic input stream."""
import hashlib
import json
import os

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def cat(hexlist):
    return b"".join(bytes.fromhex(h) for h in hexlist)


def prng(seed, i, nbytes=64):
    out = b""
    ctr = 0
    while len(out) < nbytes:
        out += hashlib.sha256(b"c12381|%d|%d|%d" % (seed, i, ctr)).digest()
        ctr += 1
    return int.from_bytes(out[:nbytes], "big")


def scalars(seed, n, mod=R):
    return b"".join((prng(seed, i) % mod).to_bytes(32, "big") for i in range(n))
