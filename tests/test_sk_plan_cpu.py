"""CPU: invariants of the stream-K work partition (csrc/sk_plan.h, the header the persistent GEMM kernel compiles) checked by a small
C++ program built with g++: every (tile, K-block) unit covered exactly once, at most one contribution per workgroup and always its
first piece, one owner per tile whose gather loop meets exactly the contributing workgroups (lower-numbered, same XCD), tile decode a
bijection."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GEOMS = [
    # n_mt tiles_n nkb W whole_tiles gw
    (194, 16, 72, 128, 0, 4),      # decoder FFN conv forward at the canonical batch (776 of 1024 m-tiles... per 4 XCD pairs)
    (194, 4, 288, 128, 0, 1),      # its data gradient: few tiles, K = 9216
    (16, 36, 388, 128, 0, 9),      # its weight gradient: 576 tiles, ~388 active K-blocks
    (250, 4, 8, 128, 1, 1),        # conformer projection, K = 256: whole tiles
    (250, 16, 8, 128, 1, 4),
    (256, 4, 32, 128, 0, 1),
    (1, 1, 40, 128, 0, 1),         # one tile cut over a whole XCD
    (3, 5, 7, 2, 0, 5),
    (7, 3, 33, 16, 0, 3),
    (5, 2, 100, 128, 0, 2),
    (13, 7, 1, 64, 0, 7),
    (2, 2, 3, 128, 1, 2),          # far fewer units than workgroups
]


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("sk") / "sk_plan_check")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "native", "sk_plan_check.cpp"), "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("geom", GEOMS)
def test_partition_invariants(checker, geom):
    r = subprocess.run([checker] + [str(v) for v in geom], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (geom, r.stdout, r.stderr)
