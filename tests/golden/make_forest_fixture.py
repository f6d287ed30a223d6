"""Turns the reference's forest point cloud (a DATA asset: uneven_map/maps/forest.pcd, 141 068 points, fields x y z) into the fixture
tests/golden/forest_xyz.npz (float32 x, y, z: what pcl::PCDReader::read<pcl::PointXYZ> delivers, uneven_map.cpp:130-131).  forest.pcd is
the ONE cloud of the reference whose 1 cm voxel filter really merges points (141 068 -> 137 490 voxels, uneven_map.cpp:140-143) and
run_forest.yaml the one parameter file on the unscaled branch (use_scaling: false, alm_traj_opt.cpp:890-893, 929-932).  The GPU box has no
/root/reference.  Run here:  python tests/golden/make_forest_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from uneven_planner_amd.scenes import read_pcd  # noqa: E402

if __name__ == "__main__":
    src = "/root/reference/src/uneven_planner/uneven_map/maps/forest.pcd"
    xyz = read_pcd(src)
    assert xyz.shape == (141068, 3) and xyz.dtype == np.float32
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "forest_xyz.npz"), xyz=xyz)
    print("forest_xyz.npz:", xyz.shape, xyz.min(axis=0), xyz.max(axis=0))
