"""The C-ABI from plain C: examples/c_abi_demo.c is compiled with gcc, run on the GPU, and the
checksum of its disparity map is compared with the oracle's on the same generated frames."""
import os
import re
import subprocess

import numpy as np
import pytest

from stereomatch_amd import capi
from oracle import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def demo_frames(W, H):
    s = 12345
    ref = np.empty(W * H, np.uint8)
    for i in range(W * H):                      # the demo's LCG
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        ref[i] = s >> 24
    ref = ref.reshape(H, W)
    idx = np.minimum(np.arange(W) + 5, W - 1)
    return ref, np.ascontiguousarray(ref[:, idx])


def fnv1a(disp):
    h = 1469598103934665603
    for b in disp.astype("<i4").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.mark.parametrize("W,H,D", [(160, 48, 32), (97, 23, 16)])
def test_c_demo_matches_oracle(tmp_path, W, H, D):
    lib_dir = os.path.dirname(capi.LIB_PATH)
    exe = str(tmp_path / "c_abi_demo")
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "c_abi_demo.c"), "-L", lib_dir, "-lsmhip",
                    "-Wl,-rpath," + lib_dir, "-o", exe], check=True)
    r = subprocess.run([exe, str(W), str(H), str(D)], capture_output=True, text=True, timeout=100)
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"sum=(\d+) fnv=([0-9a-f]{16})", r.stdout)
    assert m, r.stdout
    assert "error path:" in r.stdout

    ref, mtc = demo_frames(W, H)
    p = oracle.make_params(W, H, 1, disp_min=0, disp_max=D - 1, match_fn=1, match_interp=3,
                           match_max=1000, match_interval=1, match_interpolated=0)
    assert p.disp_n == D
    cost = oracle.raw_costs(ref, mtc, p)
    cost = oracle.box_filter(cost, 9, True)
    cost = oracle.pad_costs(cost, p)
    disp, _ = oracle.wta(cost)
    assert int(m.group(1)) == int(disp.astype(np.int64).sum())
    assert int(m.group(2), 16) == fnv1a(disp)
