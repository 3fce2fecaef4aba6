"""GPU parity of the code paths that the EXPERIMENTS build selects through the environment (A/B switches kept for measurements):
the one-lane pairing kernels, the forced naive / bucket MSM, the generic path behind the fixed-base entry points, the
one-lane-per-point G2 kernel, raw line tables of a fixed G2 argument, the queue split override.  The product library
(libc12381_hip.so) reads no environment variable and has none of these paths; the switches exist in libc12381_hip_exp.so
(-DC12381_EXPERIMENTS, same sources), which every child process here loads through C12381_LIB.
Each variant runs in a child process (the switches are read once per process) against the golden vectors."""
import os
import subprocess
import sys

import pytest

from util import golden as golden_g

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP_LIB = os.path.join(ROOT, "crypto12381_amd", "lib", "libc12381_hip_exp.so")


def exp_env(extra):
    e = dict(os.environ)
    for k in [k for k in e if k.startswith("C12381_") and k != "C12381_LIB"]:
        del e[k]
    e["C12381_LIB"] = EXP_LIB
    e.update(extra)
    return e

CODE = r"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import tools.libsel  # C12381_LIB -> capi.use_library
from util import cat, golden
from crypto12381_amd import Context
c = Context(0)
g = golden('pairing')
assert c.pair(cat(g['g1']), cat(g['g2'])) == cat(g['gt'])
assert list(c.pair_eq(cat(g['eq_a1']), cat(g['eq_a2']), cat(g['eq_b1']), cat(g['eq_b2']))) == g['eq']
assert list(c.pair_eq(cat(g['eq2_a1']), cat(g['eq2_a2']), cat(g['eq2_b1']), cat(g['eq2_b2']))) == g['eq2']
g1s, g2s = cat(g['g1']), cat(g['g2'])
for k in (0, 5, 7):                                          # table-driven Miller loop (normalised or raw line records) == generic pairing
    q = g2s[192 * k:192 * k + 192]
    assert c.pair_fixed_g2(g1s, q) == c.pair(g1s, q * (len(g1s) // 96))
g = golden('g1')
pts, sc = cat(g['points']), cat(g['scalars'])
assert c.g1_msm(pts, sc, 49).hex() == g['msm49']
assert c.g1_msm(cat(g['offsubgroup_points']), cat(g['offsubgroup_scalars']), 49).hex() == g['offsubgroup_msm49']
gen = bytes.fromhex(g['generator'])
n = len(sc) // 32
assert c.g1_mul_fixed(gen, sc, 96) == c.g1_mul(gen * n, sc, 96)
g2 = golden('g2')
gen2 = bytes.fromhex(g2['generator'])
sc2 = cat(g2['scalars'])
assert c.g2_mul_fixed(gen2, sc2, 192) == c.g2_mul(gen2 * (len(sc2) // 32), sc2, 192)
assert c.g2_mul(cat(g2['points']), sc2, 97) == cat(g2['mul97'])
assert c.g2_mul(cat(g2['offsubgroup_points']), cat(g2['offsubgroup_scalars']), 192) == cat(g2['offsubgroup_mul192'])
assert c.g2_mul(cat(g2['offsubgroup_small_points']), cat(g2['offsubgroup_small_scalars']), 192) == cat(g2['offsubgroup_small_mul192'])
c.close()
print('variant ok')
"""


@pytest.mark.parametrize("env", [{"C12381_PAIR_LANES": "1"}, {"C12381_MSM": "naive"}, {"C12381_MSM": "bucket"}, {"C12381_FIXED_BASE": "0"},
                                 {"C12381_PAIR_QUEUE": "1"}, {"C12381_PAIR_QUEUE": "0"}, {"C12381_G2_LANES": "1"}, {"C12381_FQ_RAW": "1"},
                                 {"C12381_PAIR_QUEUE": "1", "C12381_QUEUE_GROUPS": "1"}, {}],
                         ids=["one-lane-pairing", "msm-naive", "msm-bucket", "fixed-base-off", "pair-queue-on", "pair-queue-off", "g2-one-lane", "raw-line-tables",
                              "queue-groups-override", "defaults"])
def test_environment_selected_paths(env):
    e = exp_env(env)
    r = subprocess.run([sys.executable, "-c", CODE], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


DIGEST_CODE = r"""
import hashlib, sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import tools.libsel  # C12381_LIB -> capi.use_library
from util import golden, scalars
from crypto12381_amd import Context
c = Context(0)
g1 = bytes.fromhex(golden('g1')['generator'])
g2 = bytes.fromhex(golden('g2')['generator'])
m = 1 << 12
P = c.g1_mul(g1 * m, scalars(981, m), 96)
Q = c.g2_mul(g2 * m, scalars(982, m), 192)
n = 1 << 16
Pn = P * (n // m)
Qn = b"".join(Q[192 * ((5 * i + i // m) % m):192 * ((5 * i + i // m) % m) + 192] for i in range(n))
print('pair', hashlib.sha256(c.pair(Pn, Qn)).hexdigest())
print('pair_eq', hashlib.sha256(c.pair_eq(Pn, Qn, Pn[96:] + Pn[:96], Qn)).hexdigest())
n2 = 1 << 15
print('g2_mul', hashlib.sha256(c.g2_mul(Q * (n2 // m), scalars(983, n2, 1 << 256), 192)).hexdigest())
n1 = 1 << 18
sc = scalars(984, n1, 1 << 256)
print('g1_fixed', hashlib.sha256(c.g1_mul_fixed(P[:96], sc, 96)).hexdigest())
print('g1_mul', hashlib.sha256(c.g1_mul(P[:96] * n1, sc, 96)).hexdigest())
c.close()
"""


def test_full_size_digests_across_implementations():
    """EVERY output of full-size batches, compared between independent kernels for the same operation (digests computed in
    child processes, the switches being per process): three-lane work-queue pairing vs the plain grid vs the one-lane
    kernel, two-lane vs one-lane G2 multiplication, table-driven vs generic G1 multiplication.  The sampled-lane
    comparisons against the CPU oracle (test_gpu_full_size.py, test_gpu_pairing.py) cannot see a fault that hits a few
    lanes of a loaded machine; this can."""
    def run(env, product=False):
        e = exp_env(env)
        if product:
            del e["C12381_LIB"]
        r = subprocess.run([sys.executable, "-c", DIGEST_CODE], env=e, cwd=ROOT, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return dict(line.split() for line in r.stdout.strip().splitlines() if " " in line)

    base = run({}, product=True)                                   # the product library
    assert run({}) == base                                         # the experiments build with no switch set is the same path
    assert base["g1_fixed"] == base["g1_mul"]                      # table-driven == generic, all 2^18 outputs
    one = run({"C12381_PAIR_LANES": "1", "C12381_G2_LANES": "1", "C12381_FIXED_BASE": "0"})
    assert one == base
    plain = run({"C12381_PAIR_QUEUE": "0"})
    assert plain["pair"] == base["pair"] and plain["pair_eq"] == base["pair_eq"]


def test_ragged_sizes_vs_oracle(oracle_port):
    """Batch sizes around the lane-grouping boundaries of the kernels (two lanes per G2 point, 21 pairings per wavefront, 64-entry
    MSM stages, the 2^12-term switch of the window width): G2 multiplication, MSM and pairing against the CPU oracle."""
    from crypto12381_amd import Context
    from util import scalars
    ctx = Context(0)
    g1 = bytes.fromhex(golden_g("g1")["generator"])
    g2 = bytes.fromhex(golden_g("g2")["generator"])
    m = 4200
    P = ctx.g1_mul(g1 * m, scalars(961, m), 96)
    Q = ctx.g2_mul(g2 * 300, scalars(962, 300), 192)
    assert Q[:192 * 40] == oracle_port.g2_mul(g2 * 40, scalars(962, 40), 192, 8)
    for n in (1, 2, 3, 31, 32, 33, 63, 64, 65, 127, 129, 300):
        sc = scalars(963 + n, n, 1 << 256)
        assert ctx.g2_mul(Q[:192 * n], sc, 97) == oracle_port.g2_mul(Q[:192 * n], sc, 97, 16), n
    for n in (2, 3, 5, 63, 64, 65, 255, 257, 1023, 4095, 4096, 4097):
        sc = scalars(964 + n, n, 1 << 256)
        assert ctx.g1_msm(P[:96 * n], sc, 96) == oracle_port.g1_msm(P[:96 * n], sc, 96, 16), n
    for n in (1, 2, 20, 21, 22, 41, 42, 43, 63, 64, 85):
        assert ctx.pair(P[:96 * n], Q[:192 * n]) == oracle_port.pair(P[:96 * n], Q[:192 * n], 16), n
    ctx.close()


def test_product_library_ignores_the_switches():
    """libc12381_hip.so reads no tuning variable: with the variable that makes every queue hand-over of the experiments build fail
    (C12381_PAIR_SPIN_LIMIT=-1, forced queue) set in the environment it still returns the golden pairings."""
    e = dict(os.environ)
    e.pop("C12381_LIB", None)
    e.update({"C12381_PAIR_SPIN_LIMIT": "-1", "C12381_PAIR_QUEUE": "1", "C12381_PAIR_LANES": "1", "C12381_MSM": "naive", "C12381_FIXED_BASE": "0"})
    r = subprocess.run([sys.executable, "-c", CODE], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
