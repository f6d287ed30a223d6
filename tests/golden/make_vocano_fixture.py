"""Turns the reference's volcano point cloud (a DATA asset: uneven_map/maps/vocano.pcd, 100 000 points) into the fixture
tests/golden/vocano_xyz.npz (float32 x, y, z -- what pcl::PCDReader::read<pcl::PointXYZ> extracts, uneven_map.cpp:130-131) for
BASELINE.json configs[3] (volcano scene, plane-fit build sharded over x-slabs).  The GPU box has no /root/reference.
Run here:  python tests/golden/make_vocano_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uneven_planner_amd.scenes import read_pcd  # noqa: E402

if __name__ == "__main__":
    src = "/root/reference/src/uneven_planner/uneven_map/maps/vocano.pcd"
    xyz = read_pcd(src)
    assert xyz.shape == (100000, 3) and xyz.dtype == np.float32
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vocano_xyz.npz"), xyz=xyz)
    print("vocano_xyz.npz:", xyz.shape, xyz.min(axis=0), xyz.max(axis=0))
