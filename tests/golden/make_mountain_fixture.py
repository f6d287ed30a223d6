"""Turns the reference's mountain point cloud (a DATA asset: uneven_map/maps/mountain.pcd, fields x y z) into the fixture
tests/golden/mountain_xyz.npz (float32 x, y, z: what pcl::PCDReader::read<pcl::PointXYZ> delivers, uneven_map.cpp:130-131) -- the last of the
reference's five scenes without a fixture (hill.pcd is not shipped by the reference; desert, vocano and forest have theirs).  The GPU box has no
/root/reference.  Run here:  python tests/golden/make_mountain_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uneven_planner_amd.scenes import read_pcd  # noqa: E402

if __name__ == "__main__":
    src = "/root/reference/src/uneven_planner/uneven_map/maps/mountain.pcd"
    xyz = read_pcd(src)
    assert xyz.ndim == 2 and xyz.shape[1] == 3 and xyz.dtype == np.float32
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mountain_xyz.npz"), xyz=xyz)
    print("mountain_xyz.npz:", xyz.shape, xyz.min(axis=0), xyz.max(axis=0))
