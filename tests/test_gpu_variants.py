"""GPU parity of the code paths that are selected through the environment (A/B switches kept for measurements):
the one-lane pairing kernels, the forced naive / bucket MSM, the generic path behind the fixed-base entry points, the
one-lane-per-point G2 kernel.
Each variant runs in a child process (the switches are read once per process) against the golden vectors."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from util import cat, golden
from crypto12381_amd import Context
c = Context(0)
g = golden('pairing')
assert c.pair(cat(g['g1']), cat(g['g2'])) == cat(g['gt'])
assert list(c.pair_eq(cat(g['eq_a1']), cat(g['eq_a2']), cat(g['eq_b1']), cat(g['eq_b2']))) == g['eq']
assert list(c.pair_eq(cat(g['eq2_a1']), cat(g['eq2_a2']), cat(g['eq2_b1']), cat(g['eq2_b2']))) == g['eq2']
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
                                 {"C12381_PAIR_QUEUE": "1"}, {"C12381_PAIR_QUEUE": "0"}, {"C12381_G2_LANES": "1"}, {}],
                         ids=["one-lane-pairing", "msm-naive", "msm-bucket", "fixed-base-off", "pair-queue-on", "pair-queue-off", "g2-one-lane", "defaults"])
def test_environment_selected_paths(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CODE], env=e, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "variant ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
