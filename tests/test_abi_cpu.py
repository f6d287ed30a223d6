"""CPU tier: the C-ABI shared library loads and exports every symbol include/uneven_hip.h declares (no compute calls
without a GPU), the ctypes structs match the header's layout, and the product path refuses to run without a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "uneven_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(uph_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import uneven_planner_amd as U
    L = U._lib.load()
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "libunevenhip.so does not export %s" % n
    # and the binding table covers exactly the header
    assert sorted(U._lib.SYMBOLS.keys()) == names


def test_struct_layouts_match_header():
    import uneven_planner_amd as U
    # uph_map_params: int32 + 10 doubles (8-byte aligned) ; uph_opt_params: 8 d, i32, 9 d, 3 i32
    assert C.sizeof(U._lib.MapParams) == 8 + 10 * 8
    assert C.sizeof(U._lib.OptParams) == 8 * 8 + 8 + 9 * 8 + 16
    assert C.sizeof(U._lib.Problem) == 8 + 18 * 8 + 2 * 8 + 8
    assert U._lib.Problem.inner_xy.offset == 8 + 18 * 8


def test_no_cpu_fallback_without_device():
    """on a box without a GPU the product must fail loudly instead of computing on the host"""
    import uneven_planner_amd as U
    L = U._lib.load()
    if L.uph_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(U._lib.UnevenHipError):
        U.UnevenMap()


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "uneven_planner_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in txt and "liboracle" not in txt and '"../../oracle' not in txt, f


CONSUMER = r"""
// What a goal callback does with the back-end (the call sequence of plan_manager/src/plan_manager.cpp:17-22 and 134-185, written
// against the adapter; Eigen is not installed in this image, so Eigen::Vector2d / VectorXd are aliases of the adapter's stand-in)
#include <cstdio>
#include "uneven_hip_adapter.hpp"
namespace Eigen { using Vector2d = uneven_hip::VecN<2>; using VectorXd = uneven_hip::VecN<1>; }
struct FakeNodeHandle {};
struct FakeFrontend {};
using namespace uneven_hip;

int plan_once(UnevenMapHandle& map, const Mat& init_xy, const Mat& end_xy, const Mat& inner_xy, const Mat& init_yaw, const Mat& end_yaw,
              const Mat& inner_yaw, double total_time, SE2TrajMsg& traj_msg, std::vector<double>& report) {
    FakeNodeHandle nh;
    FakeFrontend* astar = nullptr;
    ALMTrajOpt traj_opt;
    traj_opt.init(nh);
    traj_opt.setFrontend(astar);
    traj_opt.setEnvironment(&map);
    const int rc = traj_opt.optimizeSE2Traj(init_xy, end_xy, inner_xy, init_yaw, end_yaw, inner_yaw, total_time);
    SE2Trajectory back_end_traj = traj_opt.getTraj();
    traj_opt.visSE2Traj(back_end_traj);
    traj_opt.visSE3Traj(back_end_traj);
    std::vector<double> max_terrain_value = traj_opt.getMaxVxAxAyCurAttSig(back_end_traj);
    report = max_terrain_value;
    report.push_back(back_end_traj.getNonHolError());
    std::printf("equal error %g max vx %g min cosxi %g\n", back_end_traj.getNonHolError(), max_terrain_value[0], -max_terrain_value[4]);
    // the message the MPC node receives: piece start points + end point, piece durations
    for (int i = 0; i < back_end_traj.pos_traj.getPieceNum(); i++) {
        Point3 pospt;
        Eigen::Vector2d pos = back_end_traj.pos_traj[i].getValue(0.0);
        pospt.x = pos[0]; pospt.y = pos[1];
        traj_msg.pos_pts.push_back(pospt);
        traj_msg.posT_pts.push_back(back_end_traj.pos_traj[i].getDuration());
    }
    Point3 pospt;
    Eigen::Vector2d pos = back_end_traj.pos_traj.getValue(back_end_traj.pos_traj.getTotalDuration());
    pospt.x = pos[0]; pospt.y = pos[1];
    traj_msg.pos_pts.push_back(pospt);
    for (int i = 0; i < back_end_traj.yaw_traj.getPieceNum(); i++) {
        Point3 anglept;
        Eigen::VectorXd angle = back_end_traj.yaw_traj[i].getValue(0.0);
        anglept.x = angle[0];
        traj_msg.angle_pts.push_back(anglept);
        traj_msg.angleT_pts.push_back(back_end_traj.yaw_traj[i].getDuration());
    }
    Point3 anglept;
    Eigen::VectorXd angle = back_end_traj.yaw_traj.getValue(back_end_traj.yaw_traj.getTotalDuration());
    anglept.x = angle[0];
    traj_msg.angle_pts.push_back(anglept);
    // ... which is what the adapter's own filler produces
    SE2TrajMsg m2;
    fillSE2TrajMsg(back_end_traj, m2);
    if (m2.pos_pts.size() != traj_msg.pos_pts.size() || m2.angleT_pts.size() != traj_msg.angleT_pts.size()) return -100;
    return rc + (traj_opt.getTrajJerkCost() > 0 ? 0 : 10);
}

// many candidate goals at once: the front-end's pose lists go in, one trajectory and return code per goal comes out
ALMTrajOpt::BatchPlan plan_many(UnevenMapHandle& map, const std::vector<std::vector<VecN<3>>>& paths) {
    ALMTrajOpt traj_opt;
    traj_opt.setEnvironment(&map);
    const uph_manager_params mgr = {0.3, 0.5, 1.2, 2.0, 0.05};           // plan_manager/params/run_hill.yaml:57-62
    return traj_opt.optimizeSE2TrajBatch(paths, mgr);
}
"""


def test_cpp_adapter_compiles(tmp_path):
    """the header-only adapter offers every member the goal callback calls on ALMTrajOpt / SE2Trajectory and compiles as plain C++17
    against the C-ABI header (the GPU tier builds and RUNS the same consumer: tests/test_gpu_adapter.py)"""
    import subprocess
    src = tmp_path / "consumer.cpp"
    src.write_text(CONSUMER)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "a.o")])
