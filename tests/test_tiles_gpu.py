"""Every tap-GEMM tile instantiation, forced one at a time (the heuristic alone would leave some of them untested)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tile", [256128, 256064, 128128, 128064, 64064, 128032, 128016])
def test_forced_tile(tile):
    env = dict(os.environ, L2S_FORCE_TILE=str(tile))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_gemm.py")], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_phase_staggered_kernel_forced():
    """csrc/phasegemm.hip on every family / ragged shape, whatever the selection heuristic would do.  (The four-wave
    128x128-wave-tile alternative, csrc/widegemm_kernel.h, is no longer part of the product library: build it with
    `make -C lip2speech_unit_amd/csrc WIDE=1` into a variant and run tools/check_phasegemm.py with L2S_WIDEGEMM=1 against it.)"""
    env = dict(os.environ, L2S_PHASEGEMM="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_phasegemm.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
