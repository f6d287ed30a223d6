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
#include <map>
// a parameter server holding plan_manager/params/run_hill.yaml's kino_astar section (the optimiser's keys are absent: getParam then leaves the
// adapter's members -- run_hill.yaml's values -- as they are); roscpp's getParam / param semantics
struct FakeNodeHandle {
    std::map<std::string, double> server{{"kino_astar/yaw_resolution", 3.15}, {"kino_astar/lambda_heu", 1.0}, {"kino_astar/weight_r2", 1.0}, {"kino_astar/weight_so2", 0.5},
        {"kino_astar/weight_v_change", 0.0}, {"kino_astar/weight_delta_change", 0.0}, {"kino_astar/weight_sigma", 10.0}, {"kino_astar/time_interval", 0.3},
        {"kino_astar/collision_interval", 0.06}, {"kino_astar/oneshot_range", 1.0}, {"kino_astar/wheel_base", 0.26}, {"kino_astar/max_steer", 0.5}, {"kino_astar/max_vel", 0.5}};
    template <class T> bool getParam(const std::string& k, T& v) const { auto it = server.find(k); if (it == server.end()) return false; v = (T)it->second; return true; }
    template <class T> void param(const std::string& k, T& v, const T& dflt) const { if (!getParam(k, v)) v = dflt; }
};
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


# ---- rosparam loading behind the adapter (VERDICT r04 missing 2): PlanManager::init's body (plan_manager/src/plan_manager.cpp:7-22) compiled against
# the adapter with a parameter server that holds the values of one of the reference's YAML files (tests/golden/run_params.json, written by
# tests/golden/make_params_fixture.py).  FakeNodeHandle follows roscpp's semantics: getParam(key, T&) leaves T alone and returns false when the key is
# absent, an integer value converts to a double member (`max_iter: 10` into `double max_iter`), param(key, T&, default) assigns the default when absent.
# The three *_create entry points of the C-ABI are defined in the test program itself (they take precedence over the library's for the header-only
# adapter's calls) and record the parameter blocks they are handed: what reaches the device is what is asserted.
ROSPARAM_MAIN = r"""
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include "uneven_hip_adapter.hpp"
struct Value { int kind = 0; bool b = false; long long i = 0; double d = 0.0; std::string s; };    // 1 bool, 2 int, 3 double, 4 string
struct FakeNodeHandle {
    std::map<std::string, Value> server;
    bool getParam(const std::string& k, bool& v) const { auto it = server.find(k); if (it == server.end() || it->second.kind != 1) return false; v = it->second.b; return true; }
    bool getParam(const std::string& k, int& v) const { auto it = server.find(k); if (it == server.end()) return false;
        if (it->second.kind == 2) { v = (int)it->second.i; return true; } if (it->second.kind == 3) { v = (int)it->second.d; return true; } return false; }
    bool getParam(const std::string& k, double& v) const { auto it = server.find(k); if (it == server.end()) return false;
        if (it->second.kind == 3) { v = it->second.d; return true; } if (it->second.kind == 2) { v = (double)it->second.i; return true; } return false; }
    bool getParam(const std::string& k, std::string& v) const { auto it = server.find(k); if (it == server.end() || it->second.kind != 4) return false; v = it->second.s; return true; }
    template <class T> void param(const std::string& k, T& v, const T& dflt) const { if (!getParam(k, v)) v = dflt; }
};
static uph_map_params g_mp; static uph_opt_params g_op; static uph_kino_params g_kp; static int g_created[3] = {0, 0, 0};
extern "C" {
int uph_map_create(const uph_map_params* mp, int, uph_map** out) { g_mp = *mp; g_created[0]++; *out = (uph_map*)&g_mp; return UPH_OK; }
void uph_map_destroy(uph_map*) {}
int uph_ctx_create(uph_map*, const uph_opt_params* p, uph_ctx** out) { g_op = *p; g_created[1]++; *out = (uph_ctx*)&g_op; return UPH_OK; }
void uph_ctx_destroy(uph_ctx*) {}
int uph_kino_create(uph_map*, const uph_kino_params* kp, int32_t, uph_kino** out) { g_kp = *kp; g_created[2]++; *out = (uph_kino*)&g_kp; return UPH_OK; }
void uph_kino_destroy(uph_kino*) {}
}
namespace ros { using NodeHandle = FakeNodeHandle; }
using std::string;
namespace uneven_planner {
struct UnevenMap {                                              // the host map after INTEGRATION.md 1: init(nh) reads its keys, then owns the device grid
    std::shared_ptr<uneven_hip::UnevenMapHandle> gpu_map;
    void init(ros::NodeHandle& nh) { gpu_map.reset(new uneven_hip::UnevenMapHandle(uneven_hip::loadMapParams(nh), 0)); }
    typedef std::shared_ptr<UnevenMap> Ptr;
};
using KinoAstar = uneven_hip::KinoAstar;
using ALMTrajOpt = uneven_hip::ALMTrajOpt;
struct PlanManager {
    double piece_len = -1, mean_vel = -1, init_time_times = -1, yaw_piece_times = -1, init_sig_vel = -1;
    string bk_dir;
    UnevenMap::Ptr uneven_map;
    std::shared_ptr<KinoAstar> kino_astar;
    ALMTrajOpt traj_opt;
    void init(ros::NodeHandle& nh)
    {
        // ---- plan_manager.cpp:9-22 as it stands, but for the two `->gpu_map.get()` of INTEGRATION.md 1
        nh.getParam("manager/piece_len", piece_len);
        nh.getParam("manager/mean_vel", mean_vel);
        nh.getParam("manager/init_time_times", init_time_times);
        nh.getParam("manager/yaw_piece_times", yaw_piece_times);
        nh.getParam("manager/init_sig_vel", init_sig_vel);
        nh.param<string>("manager/bk_dir", bk_dir, "xxx");

        uneven_map.reset(new UnevenMap);
        kino_astar.reset(new KinoAstar);

        uneven_map->init(nh);
        kino_astar->init(nh);
        kino_astar->setEnvironment(uneven_map->gpu_map.get());
        traj_opt.init(nh);
        traj_opt.setFrontend(kino_astar);
        traj_opt.setEnvironment(uneven_map->gpu_map.get());
    }
};
}
int main() {
    FakeNodeHandle nh;
%(fill)s
    uneven_planner::PlanManager pm;
    pm.init(nh);
    const uph_manager_params mg = uneven_hip::loadManagerParams(nh);
    std::printf("created %%d %%d %%d\n", g_created[0], g_created[1], g_created[2]);
    std::printf("map %%d %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g\n", g_mp.iter_num, g_mp.map_size_x, g_mp.map_size_y, g_mp.ellipsoid_x, g_mp.ellipsoid_y,
                g_mp.ellipsoid_z, g_mp.xy_resolution, g_mp.yaw_resolution, g_mp.min_cnormal, g_mp.max_rho, g_mp.gravity);
    std::printf("opt %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%d %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%d %%d %%d\n", g_op.rho_T, g_op.rho_ter, g_op.max_vel,
                g_op.max_acc_lon, g_op.max_acc_lat, g_op.max_kap, g_op.min_cxi, g_op.max_sig, g_op.use_scaling, g_op.rho, g_op.beta, g_op.gamma, g_op.epsilon_con, g_op.max_iter,
                g_op.g_epsilon, g_op.min_step, g_op.inner_max_iter, g_op.delta, g_op.mem_size, g_op.past, g_op.int_K);
    std::printf("kino %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g %%.17g\n", g_kp.yaw_resolution, g_kp.lambda_heu, g_kp.weight_r2, g_kp.weight_so2,
                g_kp.weight_v_change, g_kp.weight_delta_change, g_kp.weight_sigma, g_kp.time_interval, g_kp.collision_interval, g_kp.oneshot_range, g_kp.wheel_base, g_kp.max_steer, g_kp.max_vel);
    std::printf("manager %%.17g %%.17g %%.17g %%.17g %%.17g | %%.17g %%.17g %%.17g %%.17g %%.17g %%s\n", mg.piece_len, mg.mean_vel, mg.init_time_times, mg.yaw_piece_times, mg.init_sig_vel,
                pm.piece_len, pm.mean_vel, pm.init_time_times, pm.yaw_piece_times, pm.init_sig_vel, pm.bk_dir.c_str());
    std::printf("flags %%d %%d %%d\n", (int)pm.traj_opt.in_test, (int)pm.traj_opt.in_debug, (int)pm.kino_astar->in_test);
    return 0;
}
"""

MAP_KEYS = ["iter_num", "map_size_x", "map_size_y", "ellipsoid_x", "ellipsoid_y", "ellipsoid_z", "xy_resolution", "yaw_resolution", "min_cnormal", "max_rho", "gravity"]
OPT_KEYS = ["rho_T", "rho_ter", "max_vel", "max_acc_lon", "max_acc_lat", "max_kap", "min_cxi", "max_sig", "use_scaling", "rho", "beta", "gamma", "epsilon_con", "max_iter",
            "g_epsilon", "min_step", "inner_max_iter", "delta", "mem_size", "past", "int_K"]
KINO_KEYS = ["yaw_resolution", "lambda_heu", "weight_r2", "weight_so2", "weight_v_change", "weight_delta_change", "weight_sigma", "time_interval", "collision_interval",
             "oneshot_range", "wheel_base", "max_steer", "max_vel"]
# front_end/src/kino_astar.cpp:7-19: the defaults nh.param falls back to (NOT the YAML's values)
KINO_REF_DEFAULTS = [3.15, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0]
# the adapter's member defaults = run_hill.yaml (what a getParam on an absent key leaves in place)
OPT_MEMBER_DEFAULTS = [100000.0, 10.0, 0.5, 5.0, 10.0, 2.1, 0.8, 0.05, 1, 1.0, 1000.0, 1.0, 0.001, 10.0, 1e-3, 1e-32, 10000.0, 1e-4, 256, 3, 16]


def _run_rosparam_program(tmp_path, server, tag):
    import subprocess
    lines = []
    for k, v in sorted(server.items()):
        if isinstance(v, bool):
            lines.append('    { Value x; x.kind = 1; x.b = %s; nh.server["%s"] = x; }' % ("true" if v else "false", k))
        elif isinstance(v, int):
            lines.append('    { Value x; x.kind = 2; x.i = %d; nh.server["%s"] = x; }' % (v, k))
        elif isinstance(v, float):
            lines.append('    { Value x; x.kind = 3; x.d = %s; nh.server["%s"] = x; }' % (float(v).hex(), k))
        else:
            lines.append('    { Value x; x.kind = 4; x.s = "%s"; nh.server["%s"] = x; }' % (v, k))
    src = tmp_path / ("rosparam_%s.cpp" % tag)
    src.write_text(ROSPARAM_MAIN % dict(fill="\n".join(lines)))
    exe = str(tmp_path / ("rosparam_%s" % tag))
    libdir = os.path.join(ROOT, "uneven_planner_amd")     # every other symbol of the C-ABI comes from the real library (none of them is called here)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe, "-L", libdir, "-lunevenhip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([exe]).decode().strip().split("\n")
    return {ln.split()[0]: ln.split()[1:] for ln in out}


def _server_for(scene):
    import json
    doc = json.load(open(os.path.join(ROOT, "tests", "golden", "run_params.json")))[scene]
    return doc, {"%s/%s" % (sec, k): v for sec, kv in doc.items() for k, v in kv.items()}


@pytest.mark.parametrize("scene", ["forest", "vocano", "hill"])
def test_plan_manager_init_loads_the_rosparams(tmp_path, scene):
    """init(nh) of the adapter's ALMTrajOpt / KinoAstar and loadMapParams read the parameter server like the reference's init functions
    (alm_traj_opt.cpp:7-29, kino_astar.cpp:7-20, uneven_map.cpp:75-88): a forest / volcano launch reaches the device with ITS values"""
    doc, server = _server_for(scene)
    r = _run_rosparam_program(tmp_path, server, scene)
    assert r["created"] == ["1", "1", "1"]
    got_map = [float(v) for v in r["map"]]
    assert got_map == [float(doc["uneven_map"][k]) for k in MAP_KEYS]
    got_opt = [float(v) for v in r["opt"]]
    assert got_opt == [float(doc["alm_traj_opt"][k]) for k in OPT_KEYS]
    assert [float(v) for v in r["kino"]] == [float(doc["kino_astar"][k]) for k in KINO_KEYS]
    mg = [float(v) for v in r["manager"][:5]]
    assert mg == [doc["manager"][k] for k in ("piece_len", "mean_vel", "init_time_times", "yaw_piece_times", "init_sig_vel")] == [float(v) for v in r["manager"][6:11]]
    assert r["manager"][11] == "xxx" and r["flags"] == ["0", "0", "0"]
    if scene == "forest":                # the values the review named (run_forest.yaml:12,33,40-41)
        o = dict(zip(OPT_KEYS, got_opt))
        assert o["use_scaling"] == 0 and o["rho_T"] == 500.0 and o["max_sig"] == 0.001 and dict(zip(MAP_KEYS, got_map))["max_rho"] == 0.001
    if scene == "vocano":
        assert dict(zip(OPT_KEYS, got_opt))["max_sig"] == 0.08 and dict(zip(MAP_KEYS, got_map))["max_rho"] == 0.08


def test_rosparam_absent_keys_follow_the_reference(tmp_path):
    """an EMPTY parameter server: getParam leaves the optimiser's and the map's members alone (alm_traj_opt.cpp:7-29: the reference's members are then
    uninitialised; the adapter's start as run_hill.yaml), nh.param gives KinoAstar the reference's OWN defaults (kino_astar.cpp:7-19: weight_so2 1.0,
    weight_sigma 0.0, time_interval 1.0 ... not the YAML's); a partly filled server overrides exactly its keys"""
    r = _run_rosparam_program(tmp_path, {}, "empty")
    assert [float(v) for v in r["opt"]] == [float(v) for v in OPT_MEMBER_DEFAULTS]
    assert [float(v) for v in r["kino"]] == KINO_REF_DEFAULTS
    assert [float(v) for v in r["map"]] == [2.0, 10.0, 10.0, 0.2, 0.1, 0.1, 0.05, 0.1, 0.8, 0.05, 9.81]
    assert [float(v) for v in r["manager"][5 + 1:11]] == [-1.0] * 5          # PlanManager's own members: untouched by getParam
    r = _run_rosparam_program(tmp_path, {"alm_traj_opt/rho_T": 500.0, "alm_traj_opt/use_scaling": False, "alm_traj_opt/max_iter": 7, "alm_traj_opt/mem_size": 16,
                                         "kino_astar/weight_sigma": 10.0, "uneven_map/max_rho": 0.001, "alm_traj_opt/in_debug": True}, "partial")
    want = list(OPT_MEMBER_DEFAULTS)
    want[OPT_KEYS.index("rho_T")] = 500.0; want[OPT_KEYS.index("use_scaling")] = 0; want[OPT_KEYS.index("max_iter")] = 7.0; want[OPT_KEYS.index("mem_size")] = 16
    assert [float(v) for v in r["opt"]] == [float(v) for v in want]
    k = list(KINO_REF_DEFAULTS); k[KINO_KEYS.index("weight_sigma")] = 10.0
    assert [float(v) for v in r["kino"]] == k
    assert float(r["map"][MAP_KEYS.index("max_rho")]) == 0.001 and r["flags"] == ["0", "1", "0"]
