"""Build-container-only: the committed golden fixtures are what the committed generator scripts produce.

Regenerates all nine ``tests/golden/*.npz`` files into a temporary directory by running the three generator scripts
(they import the reference from /root/reference) and compares them with the committed files array by array, bit for
bit.  Skipped where the reference is absent (the GPU box): the fixtures travel, the reference does not.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
REFERENCE = "/root/reference"

SCRIPTS = {
    "make_golden.py": ["frustum.npz", "geometry.npz", "voxel_pooling.npz", "lift.npz", "input_contract.npz"],
    "make_golden_aux.py": ["losses.npz", "kitti_eval.npz", "result2kitti.npz"],
    "make_golden_modules.py": ["modules.npz"],
}


def _same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="needs the reference checkout (build container only)")
@pytest.mark.parametrize("script", sorted(SCRIPTS))
def test_generator_reproduces_committed_fixture(script, tmp_path):
    env = dict(os.environ, SGV3D_GOLDEN_OUT=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for name in SCRIPTS[script]:
        new, old = np.load(tmp_path / name, allow_pickle=False), np.load(os.path.join(GOLDEN, name), allow_pickle=False)
        assert sorted(new.files) == sorted(old.files), f"{name}: array names differ"
        bad = [k for k in old.files if not _same(new[k], old[k])]
        assert not bad, f"{name}: {len(bad)} of {len(old.files)} arrays differ from the committed fixture, e.g. {bad[:5]}"
