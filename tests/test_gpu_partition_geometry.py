"""The generation-3 scatter of the partitioned group-by (one-partition-per-thread scan, descriptor copy-out) under every geometry the planner can choose,
not only the benchmark's 256 partitions / 8192-row tiles: 64 to 512 partitions (one to eight scan waves), 2048- to 8192-row tiles, hash and direct
partitions, plain and packed records, with and without the hot-key build.  The knobs are read once per process, so each geometry runs tests/part_geometry_worker.py in its own."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GEOMETRIES = [      # (what the second run -- key range known -- must report; the planner never takes fewer partitions than the group estimate needs: 128 here)
    ("hash", "hot", {"PLX_PART_LOG2_PARTS": "6"}, ["hash,P=128,", "hot=48)"]),
    ("hash", "flat", {"PLX_PART_LOG2_PARTS": "7", "PLX_PART_TILES": "2"}, ["hash,P=128,", "tile=4096,", "hot=0)"]),
    ("hash", "flat", {"PLX_PART_LOG2_PARTS": "9", "PLX_PART_TILES": "1"}, ["hash,P=512,", "tile=2048,"]),
    ("hash", "hot", {"PLX_PART_LOG2_PARTS": "9", "PLX_PART_PACK": "0"}, ["hash,P=512,", "pack=0,"]),
    ("direct", "flat", {"PLX_PART_DIRECT_LOG2_PARTS": "6"}, ["direct,P=64,", "tile=8192,"]),
    ("direct", "hot", {"PLX_PART_DIRECT_LOG2_PARTS": "9", "PLX_PART_TILES": "2"}, ["direct,P=512,", "tile=4096,"]),
    ("direct", "flat", {"PLX_PART_DIRECT_LOG2_PARTS": "8", "PLX_PART_PACK": "1"}, ["direct,P=256,"]),
    ("hash", "flat", {"PLX_PART_LOG2_PARTS": "8", "PLX_PART_TILES": "3", "PLX_PART_PACK": "0"}, ["hash,P=256,", "rec=24B,pack=0,", "tile=4096,"]),
    ("hash", "hot", {"PLX_PART_LOG2_PARTS": "8", "PLX_PART_TILES": "3"}, ["hash,P=256,", "rec=20B,", "tile=4096,", "slots=4606)"]),     # an LDS table that is not a power of two
    ("direct", "hot", {"PLX_PART_DIRECT_LOG2_PARTS": "8", "PLX_PART_INTERLEAVE": "0"}, ["direct,P=256,"]),          # partition = the id's high bits (the join probe's mapping)
    # one f64 value over dense ids: two rows a record (fused::kPackPair), pairs formed in the tile sort, lone halves closed with the absent slot
    ("direct", "flat1", {"PLX_PART_DIRECT_LOG2_PARTS": "8"}, ["direct,P=256,", "rec=10B,pack=4,", "tile=8192,"]),
    ("direct", "hot1", {"PLX_PART_DIRECT_LOG2_PARTS": "9", "PLX_PART_TILES": "2"}, ["direct,P=512,", "rec=10B,pack=4,", "tile=4096,"]),
    ("direct", "hot1", {"PLX_PART_DIRECT_LOG2_PARTS": "6"}, ["direct,P=64,", "pack=4,"]),
    ("direct", "flat1", {"PLX_PART_DIRECT_LOG2_PARTS": "9", "PLX_PART_TILES": "1"}, ["direct,P=512,", "rec=12B,pack=0,", "tile=2048,"]),      # no room for 512 lone halves in a 2048-row tile
    ("direct", "flat1", {"PLX_PART_PAIR": "0"}, ["rec=12B,pack=0,"]),
    # ... and over sparse 64-bit keys once their exact range is known (the first run learns it): 48-bit key offsets, seven dwords for two rows
    ("hash", "flat1", {"PLX_PART_LOG2_PARTS": "8", "PLX_PART_TILES": "3"}, ["hash,P=256,", "rec=14B,pack=4,", "tile=6144,"]),
    ("hash", "hot1", {"PLX_PART_LOG2_PARTS": "7"}, ["hash,P=128,", "rec=14B,pack=4,"]),
    ("hash", "flat1", {"PLX_PART_LOG2_PARTS": "9", "PLX_PART_TILES": "1"}, ["hash,P=512,", "rec=16B,pack=0,", "tile=2048,"]),
    # keys over all 64 bits, an Int64 value spanning 2^41: the VALUE travels as a 48-bit offset (fused::kPackPairV); the first run takes sampled bounds, checked per row
    ("hash", "flatv", {"PLX_PART_LOG2_PARTS": "8", "PLX_PART_TILES": "3"}, ["hash,P=256,", "rec=14B,pack=5,", "tile=6144,"]),
    ("hash", "hotv", {"PLX_PART_LOG2_PARTS": "7"}, ["hash,P=256,", "rec=14B,pack=5,", "hot=48,"]),      # (hot keys: tables planned for 4 x the estimate)
]


@pytest.mark.parametrize("mode,shape,env,want", GEOMETRIES, ids=[f"{m}-{sh}-" + "-".join(f"{k[9:].lower()}{v}" for k, v in e.items()) for m, sh, e, _ in GEOMETRIES])
def test_generation3_scatter_geometries(mode, shape, env, want):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "part_geometry_worker.py"), mode, shape, *want], capture_output=True, text=True, timeout=240, cwd=ROOT, env=e)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-1500:], r.stderr[-2500:])
