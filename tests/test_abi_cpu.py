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

// the front end in front of it (plan_manager.cpp:56-60): kino_astar->plan(start_state, end_state) for one goal and for many
std::vector<std::vector<VecN<3>>> search_many(UnevenMapHandle& map, const std::vector<VecN<3>>& starts, const std::vector<VecN<3>>& goals) {
    KinoAstar kino_astar;
    FakeNodeHandle nh;
    kino_astar.init(nh);
    kino_astar.weight_sigma = 10.0;
    kino_astar.setEnvironment(&map);
    std::vector<VecN<3>> init_path = kino_astar.plan(starts[0], goals[0]);
    if (init_path.empty()) std::printf("front end failed: status %d\n", kino_astar.status[0]);
    return kino_astar.planBatch(starts, goals);
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


# ---- INTEGRATION.md's reference-side recipe, compiled against the reference's ACCESS RULES -------------------------------------------------
# A stand-in for uneven_map/include/uneven_map/uneven_map.h:66-152 with the same member names in the same sections: parameters and buffers
# PRIVATE (:68-110), the query interface public (:112-151).  Types the image lacks (Eigen, PCL, ROS) are minimal stand-ins with just the
# members the snippets touch; what matters here is who may touch what.
INTEGRATION_STUB = r"""
#include <memory>
#include <string>
#include <vector>
%(map_header)s
namespace Eigen {
struct Vector3d { double v[3] = {0, 0, 0}; double& operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } double& operator[](int i) { return v[i]; } };
struct Vector3i { int v[3] = {0, 0, 0}; int& operator()(int i) { return v[i]; } };
}
namespace pcl { struct PointXYZ { float x, y, z; }; template <class P> struct PointCloud { std::vector<P> points; }; }
namespace ros { struct NodeHandle {}; }
namespace uneven_planner {
using std::string; using std::vector; using std::shared_ptr;
struct RXS2 { double z = 0, sigma = 0, zbx = 0, zby = 0; };           // uneven_map.h:36-64: four doubles
class UnevenMap
{
    private:                                                           // uneven_map.h:68-110
        int             iter_num;
        double          ellipsoid_x;
        double          ellipsoid_y;
        double          ellipsoid_z;
        double          xy_resolution, xy_resolution_inv;
        double          yaw_resolution, yaw_resolution_inv;
        double          min_cnormal;
        double          max_rho;
        double          gravity;
        double          mass;
        Eigen::Vector3d map_origin;
        Eigen::Vector3d map_size;
        Eigen::Vector3d min_boundary;
        Eigen::Vector3d max_boundary;
        Eigen::Vector3i min_idx;
        Eigen::Vector3i max_idx;
        Eigen::Vector3i voxel_num;
        string          map_file;
        string          pcd_file;
        vector<RXS2>    map_buffer;
        vector<double>  c_buffer;
        vector<char>    occ_buffer;
        vector<char>    occ_r2_buffer;
        bool            map_ready = false;
    public:                                                            // uneven_map.h:112-151
        UnevenMap() {}
%(map_members)s
        void init(ros::NodeHandle& nh);
        bool constructMapInput() { return false; }
        bool constructMap() { return true; }
        typedef shared_ptr<UnevenMap> Ptr;
};
void UnevenMap::init(ros::NodeHandle& nh)
{
    (void)nh;
    pcl::PointCloud<pcl::PointXYZ> cloudMapOrigin;                     // uneven_map.cpp:127-131
%(map_init_read)s
%(map_init_build)s
    {
%(multi_map_init)s
    }
}
struct PlanManager                                                     // plan_manager.h:26-40: a friend of nobody
{
    UnevenMap::Ptr uneven_map;
    uneven_hip::ALMTrajOpt traj_opt;
    void init(ros::NodeHandle& nh)
    {
        uneven_map.reset(new UnevenMap);
        uneven_map->init(nh);
%(manager_init)s
    }
    void many(const std::vector<std::vector<uneven_hip::VecN<3>>>& paths, const uph_manager_params& manager_params)
    {
%(multi_manager)s
        (void)plans; (void)reports;
    }
};
}
"""


def _integration_snippets():
    """fenced cpp blocks of INTEGRATION.md tagged `<!-- snippet: name -->`: in a diff block the `+` lines (verbatim, marker stripped), else all lines"""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    out = {}
    for m in re.finditer(r"<!-- snippet: (\w+) -->\n```cpp\n(.*?)```", txt, flags=re.S):
        lines = m.group(2).split("\n")
        is_diff = any(ln.startswith("+") for ln in lines)
        out[m.group(1)] = "\n".join((ln[1:] if is_diff else ln) for ln in lines if (ln.startswith("+") if is_diff else True))
    return out


def test_integration_recipe_compiles_against_private_members(tmp_path):
    """the reference-side snippets of INTEGRATION.md 1 and 4(a) are compiled AS WRITTEN against a stand-in UnevenMap whose
    parameters and buffers are private like the reference's -- a recipe that reads them from PlanManager does not build"""
    import subprocess
    sn = _integration_snippets()
    for need in ("map_header", "map_init_read", "map_init_build", "manager_init", "multi_map_init", "multi_manager"):
        assert need in sn and sn[need].strip(), "INTEGRATION.md lost its `%s` snippet" % need
    hdr = [ln for ln in sn["map_header"].split("\n") if ln.strip().startswith("#include")]
    members = [ln for ln in sn["map_header"].split("\n") if not ln.strip().startswith("#include")]
    members.append("        std::vector<std::shared_ptr<uneven_hip::UnevenMapHandle>> gpu_maps;")       # the multi-GPU variant's member (in prose)
    src = INTEGRATION_STUB % dict(map_header="\n".join(hdr), map_members="\n".join(members), map_init_read=sn["map_init_read"], map_init_build=sn["map_init_build"],
                                  manager_init=sn["manager_init"], multi_map_init=sn["multi_map_init"], multi_manager=sn["multi_manager"])
    f = tmp_path / "recipe.cpp"
    f.write_text(src)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wno-unused-variable", "-I", os.path.join(ROOT, "include"), "-c", str(f), "-o", str(tmp_path / "r.o")])
    # ... and the check bites: the round-3 form of the recipe (PlanManager reading the private buffers) must NOT compile
    bad = src.replace("        uneven_map->init(nh);\n", "        uneven_map->init(nh);\n        (void)uneven_map->map_buffer.data();\n", 1)
    g = tmp_path / "bad.cpp"
    g.write_text(bad)
    assert subprocess.call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-c", str(g), "-o", str(tmp_path / "b.o")], stderr=subprocess.DEVNULL) != 0
