// uneven_hip_adapter.hpp -- header-only C++ adapter that gives libunevenhip.so the member names PlanManager uses on the
// reference's ALMTrajOpt and SE2Trajectory (plan_manager/src/plan_manager.cpp:17-22, 134-185), so that the goal callback compiles
// against it as it stands (INTEGRATION.md shows the swap).
//
// Mirrors (paths under /root/reference/src/uneven_planner):
//   ALMTrajOpt::optimizeSE2Traj   back_end/include/back_end/alm_traj_opt.h:92-98   (same argument list, same 0/1/2 return)
//   ALMTrajOpt::getTraj           alm_traj_opt.h:165-168  -> per-piece durations + D x 6 coefficient matrices, highest order first
//                                                            (MinJerkOpt::getTraj, back_end/include/utils/se2traj.hpp:682-695)
//   ALMTrajOpt::setEnvironment    alm_traj_opt.h:127-130
//   public parameter members      alm_traj_opt.h:29-53
//   getMaxVxAxAyCurAttSig         alm_traj_opt.h:170-229 (device report), init / setFrontend / visSE2Traj / visSE3Traj (no-ops here)
//   Piece / PolyTrajectory / SE2Trajectory   back_end/include/utils/se2traj.hpp:30-150, 253-406, 408-562 (evaluation members, getNonHolError)
//   mpc_controller/SE2Traj filler  plan_manager.cpp:150-182, mpc_controller/msg/SE2Traj.msg:1-9
//   KinoAstar::plan / setEnvironment / init   front_end/include/front_end/kino_astar.h:147-154, front_end/src/kino_astar.cpp:5-43, 67-236
//                                  (the batched device search, uph_kino_plan_batch; planBatch = many goals in one call)
//
// The matrix/vector types are template parameters: anything with data(), rows(), cols()/size() and column-major storage works
// (Eigen::MatrixXd / Eigen::VectorXd in the ROS workspace; the tiny Mat/Vec below where Eigen is not installed, as in this
// repository's image).  No Eigen header is included here.
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "uneven_hip.h"

namespace uneven_hip {

struct Mat {                       // minimal column-major stand-in with the Eigen accessors the adapter touches
    int r = 0, c = 0;
    std::vector<double> v;
    Mat() {}
    Mat(int rows_, int cols_) : r(rows_), c(cols_), v((size_t)rows_ * cols_, 0.0) {}
    const double* data() const { return v.data(); }
    double* data() { return v.data(); }
    int rows() const { return r; }
    int cols() const { return c; }
    int size() const { return r * c; }
    double& operator()(int i, int j) { return v[(size_t)j * r + i]; }
    double operator()(int i, int j) const { return v[(size_t)j * r + i]; }
};

// column vector of D doubles: Eigen's own type where Eigen is installed (so that `Eigen::Vector2d p = traj.pos_traj[i].getValue(0.0)`
// compiles unchanged), a minimal stand-in otherwise
#if __has_include(<Eigen/Core>)
}  // namespace uneven_hip
#include <Eigen/Core>
namespace uneven_hip {
template <int D> using VecN = Eigen::Matrix<double, D, 1>;
template <int D> inline VecN<D> vecZero() { return VecN<D>::Zero(); }
#else
template <int D>
struct VecN {
    double v[D];
    double operator[](int i) const { return v[i]; }
    double& operator[](int i) { return v[i]; }
    double operator()(int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
    double x() const { return v[0]; }
    double y() const { return v[D > 1 ? 1 : 0]; }
};
template <int D> inline VecN<D> vecZero() { VecN<D> z; for (int d = 0; d < D; d++) z.v[d] = 0.0; return z; }
#endif

// Piece<Dim> (se2traj.hpp:30-150): duration + Dim x 6 coefficients, highest order first; the three evaluators follow the reference's loops
template <int Dim>
class Piece {
public:
    double duration = 0.0;
    double coeff[Dim][6];
    int getDim() const { return Dim; }
    int getOrder() const { return 5; }
    double getDuration() const { return duration; }
    VecN<Dim> getValue(const double& t) const {                         // :106-116
        VecN<Dim> value = vecZero<Dim>();
        double tn = 1.0;
        for (int i = 5; i >= 0; i--) { for (int d = 0; d < Dim; d++) value[d] += tn * coeff[d][i]; tn *= t; }
        return value;
    }
    VecN<Dim> getDotValue(const double& t) const {                      // :118-131
        VecN<Dim> value = vecZero<Dim>();
        double tn = 1.0;
        int n = 1;
        for (int i = 4; i >= 0; i--) { for (int d = 0; d < Dim; d++) value[d] += n * tn * coeff[d][i]; tn *= t; n++; }
        return value;
    }
    VecN<Dim> getDDotValue(const double& t) const {                     // :133-150
        VecN<Dim> value = vecZero<Dim>();
        double tn = 1.0;
        int m = 1, n = 2;
        for (int i = 3; i >= 0; i--) { for (int d = 0; d < Dim; d++) value[d] += m * n * tn * coeff[d][i]; tn *= t; m++; n++; }
        return value;
    }
};

// PolyTrajectory<Dim> (se2traj.hpp:253-406): the members PlanManager, the report and the MPC message filler use
template <int Dim>
class PolyTrajectory {
public:
    std::vector<Piece<Dim>> pieces;
    int getPieceNum() const { return (int)pieces.size(); }
    double getTotalDuration() const {                                   // :290-299
        double total = 0.0;
        for (const Piece<Dim>& p : pieces) total += p.getDuration();
        return total;
    }
    const Piece<Dim>& operator[](int i) const { return pieces[i]; }
    Piece<Dim>& operator[](int i) { return pieces[i]; }
    typename std::vector<Piece<Dim>>::const_iterator begin() const { return pieces.begin(); }
    typename std::vector<Piece<Dim>>::const_iterator end() const { return pieces.end(); }
    void clear() { pieces.clear(); }
    void emplace_back(const Piece<Dim>& p) { pieces.emplace_back(p); }
    int locatePieceIdx(double& t) const {                               // :343-361
        const int N = getPieceNum();
        int idx;
        double dur;
        for (idx = 0; idx < N && t > (dur = pieces[idx].getDuration()); idx++) t -= dur;
        if (idx == N) { idx--; t += pieces[idx].getDuration(); }
        return idx;
    }
    VecN<Dim> getValue(double t) const { const int i = locatePieceIdx(t); return pieces[i].getValue(t); }
    VecN<Dim> getDotValue(double t) const { const int i = locatePieceIdx(t); return pieces[i].getDotValue(t); }
    VecN<Dim> getDDotValue(double t) const { const int i = locatePieceIdx(t); return pieces[i].getDDotValue(t); }
};

class SE2Trajectory {              // se2traj.hpp:408-562
public:
    PolyTrajectory<2> pos_traj;
    PolyTrajectory<1> yaw_traj;
    double getTotalDuration() const { const double a = pos_traj.getTotalDuration(), b = yaw_traj.getTotalDuration(); return a < b ? a : b; }
    VecN<2> getPos(double t) const { return pos_traj.getValue(t); }
    VecN<2> getVel(double t) const { return pos_traj.getDotValue(t); }
    VecN<2> getAcc(double t) const { return pos_traj.getDDotValue(t); }
    double getAngle(double t) const { return yaw_traj.getValue(t)[0]; }
    double getAngleRate(double t) const { return yaw_traj.getDotValue(t)[0]; }
    double getNonHolError() const {                                     // :551-561
        double error = 0.0;
        for (double t = 0.0; t < getTotalDuration(); t += 0.01) {
            const VecN<2> v = getVel(t);
            const double yaw = getAngle(t);
            error += std::fabs(v[0] * std::sin(yaw) + v[1] * (-std::cos(yaw)));
        }
        return error;
    }
};

// mpc_controller/SE2Traj (mpc_controller/msg/SE2Traj.msg:1-9) as plain data, and the filler PlanManager runs before publishing
// (plan_manager.cpp:150-182): piece start points + the end point, piece durations, zero init_v / init_a.  fillSE2TrajMsg is a
// template, so the real ROS message type works as well as this stand-in (start_time is left to the caller: ros::Time::now()).
struct Point3 { double x = 0.0, y = 0.0, z = 0.0; };
struct SE2TrajMsg {
    double start_time = 0.0;
    std::vector<Point3> pos_pts, angle_pts;
    Point3 init_v, init_a;
    std::vector<double> posT_pts, angleT_pts;
};
template <class Msg>
inline void fillSE2TrajMsg(const SE2Trajectory& traj, Msg& msg) {
    typedef typename std::remove_reference<decltype(msg.pos_pts[0])>::type Pt;
    msg.init_v.x = 0.0; msg.init_v.y = 0.0; msg.init_v.z = 0.0;
    msg.init_a.x = 0.0; msg.init_a.y = 0.0; msg.init_a.z = 0.0;
    msg.pos_pts.clear(); msg.posT_pts.clear(); msg.angle_pts.clear(); msg.angleT_pts.clear();
    for (int i = 0; i < traj.pos_traj.getPieceNum(); i++) {
        Pt pt{};
        const VecN<2> pos = traj.pos_traj[i].getValue(0.0);
        pt.x = pos[0]; pt.y = pos[1];
        msg.pos_pts.push_back(pt);
        msg.posT_pts.push_back(traj.pos_traj[i].getDuration());
    }
    {
        Pt pt{};
        const VecN<2> pos = traj.pos_traj.getValue(traj.pos_traj.getTotalDuration());
        pt.x = pos[0]; pt.y = pos[1];
        msg.pos_pts.push_back(pt);
    }
    for (int i = 0; i < traj.yaw_traj.getPieceNum(); i++) {
        Pt pt{};
        pt.x = traj.yaw_traj[i].getValue(0.0)[0];
        msg.angle_pts.push_back(pt);
        msg.angleT_pts.push_back(traj.yaw_traj[i].getDuration());
    }
    {
        Pt pt{};
        pt.x = traj.yaw_traj.getValue(traj.yaw_traj.getTotalDuration())[0];
        msg.angle_pts.push_back(pt);
    }
}

// rosparam uneven_map/... as UnevenMap::init reads it (uneven_map/src/uneven_map.cpp:75-88; getParam: absent keys keep the value passed in,
// which starts as run_hill.yaml:2-14) -- for hosts that create the device grid without the reference's UnevenMap object.  INTEGRATION.md 1 builds
// the same block from UnevenMap's own members, which that very init has just loaded.  `mass`, `map_pcd`, `map_file` have no device side.
template <class NodeHandle>
inline uph_map_params loadMapParams(NodeHandle& nh, uph_map_params mp = uph_map_params{2, 10.0, 10.0, 0.2, 0.1, 0.1, 0.05, 0.1, 0.8, 0.05, 9.81}) {
    int iter_num = mp.iter_num;
    nh.getParam("uneven_map/iter_num", iter_num);
    mp.iter_num = iter_num;
    nh.getParam("uneven_map/map_size_x", mp.map_size_x);
    nh.getParam("uneven_map/map_size_y", mp.map_size_y);
    nh.getParam("uneven_map/ellipsoid_x", mp.ellipsoid_x);
    nh.getParam("uneven_map/ellipsoid_y", mp.ellipsoid_y);
    nh.getParam("uneven_map/ellipsoid_z", mp.ellipsoid_z);
    nh.getParam("uneven_map/xy_resolution", mp.xy_resolution);
    nh.getParam("uneven_map/yaw_resolution", mp.yaw_resolution);
    nh.getParam("uneven_map/min_cnormal", mp.min_cnormal);
    nh.getParam("uneven_map/max_rho", mp.max_rho);
    nh.getParam("uneven_map/gravity", mp.gravity);
    return mp;
}
// rosparam manager/... of PlanManager::init (plan_manager/src/plan_manager.cpp:9-13), for optimizeSE2TrajBatch's initial-guess stage
template <class NodeHandle>
inline uph_manager_params loadManagerParams(NodeHandle& nh, uph_manager_params mg = uph_manager_params{0.3, 0.5, 1.2, 2.0, 0.05, 0, 0.5}) {
    nh.getParam("manager/piece_len", mg.piece_len);
    nh.getParam("manager/mean_vel", mg.mean_vel);
    nh.getParam("manager/init_time_times", mg.init_time_times);
    nh.getParam("manager/yaw_piece_times", mg.yaw_piece_times);
    nh.getParam("manager/init_sig_vel", mg.init_sig_vel);
    return mg;
}

class UnevenMapHandle {            // owns a uph_map; what UnevenMap::Ptr is to the reference's optimiser
public:
    // fp32_cells: store the cells as four floats (16 bytes) instead of four doubles -- km^2-scale grids (BASELINE.json configs[4])
    explicit UnevenMapHandle(const uph_map_params& mp, int device = 0, bool fp32_cells = false) {
        if ((fp32_cells ? uph_map_create_f32(&mp, device, &m_) : uph_map_create(&mp, device, &m_)) != UPH_OK)
            throw std::runtime_error(std::string("uph_map_create: ") + uph_last_error());
    }
    ~UnevenMapHandle() { uph_map_destroy(m_); }
    UnevenMapHandle(const UnevenMapHandle&) = delete;
    UnevenMapHandle& operator=(const UnevenMapHandle&) = delete;
    // UnevenMap::constructMap replacement: xyz = the cloud pcl::PCDReader delivered (n x 3 float)
    void constructMap(const float* xyz, long n) {
        int32_t d[3];
        uph_map_dims(m_, d);
        if (uph_map_build(m_, xyz, n, 0, d[0]) != UPH_OK) throw std::runtime_error(std::string("uph_map_build: ") + uph_last_error());
    }
    // UnevenMap::constructMapInput replacement: cells from the `.map` cache (ncell x 4: z, sigma, zb.x, zb.y in the reference's address order)
    void setCells(const double* rxs2) {
        if (uph_map_set_cells(m_, rxs2) != UPH_OK) throw std::runtime_error(std::string("uph_map_set_cells: ") + uph_last_error());
    }
    // the `.map` cache (uneven_map.cpp:166-167, 270-315, 400-412).  loadMap = constructMapInput: false when no cache can be read (build the map
    // then); it prefers the bit-exact side-car `<map_file>.bin` and falls back to the reference's CSV.  saveMap = the "to txt" block at the end of
    // constructMap -- the CSV the reference's own constructMapInput reads (six significant digits) -- plus the side-car.
    bool loadMap(const std::string& map_file) {
        int32_t src = 0;
        const int rc = uph_map_load_cache(m_, map_file.c_str(), (map_file + ".bin").c_str(), &src);
        if (rc == UPH_ERR_NO_CACHE) return false;      // build the map; every other failure surfaces instead of triggering a rebuild that overwrites the cache
        if (rc != UPH_OK) throw std::runtime_error(std::string("uph_map_load_cache: ") + uph_last_error());
        return true;
    }
    void saveMap(const std::string& map_file, bool with_sidecar = true) {
        const std::string bin = map_file + ".bin";
        if (uph_map_save_cache(m_, map_file.c_str(), with_sidecar ? bin.c_str() : nullptr) != UPH_OK) throw std::runtime_error(std::string("uph_map_save_cache: ") + uph_last_error());
    }
    // analytic fractal terrain instead of a cloud (configs[4]; no counterpart in the reference)
    void fillFractal(const uph_fbm_params& fp) {
        if (uph_map_fill_fbm(m_, &fp, 0, 0) != UPH_OK) throw std::runtime_error(std::string("uph_map_fill_fbm: ") + uph_last_error());
    }
    // fills the host members of the reference's UnevenMap (map_buffer as 4 doubles per cell, c_buffer, occ_buffer, occ_r2_buffer)
    void download(double* rxs2, double* c, char* occ, char* occ_r2) { uph_map_get_cells(m_, rxs2, c, occ, occ_r2); }
    uph_map* get() const { return m_; }
    // constructMap over several GPUs of this process: maps[g] lives on device g; x-slab fits on all devices at once, one RCCL all-gather inside
    // the library, every map ends up with the complete grid (uph_map_build_multi)
    static void constructMapMulti(const std::vector<UnevenMapHandle*>& maps, const float* xyz, long n) {
        std::vector<uph_map*> h;
        for (UnevenMapHandle* m : maps) h.push_back(m->get());
        if (uph_map_build_multi(h.data(), (int32_t)h.size(), xyz, n) != UPH_OK) throw std::runtime_error(std::string("uph_map_build_multi: ") + uph_last_error());
    }

private:
    uph_map* m_ = nullptr;
};

class ALMTrajOpt {
public:
    // ---- the reference's public parameter members (alm_traj_opt.h:29-53); defaults = plan_manager/params/run_hill.yaml:32-55
    double rho_T = 100000.0, rho_ter = 10.0, max_vel = 0.5, max_acc_lon = 5.0, max_acc_lat = 10.0, max_kap = 2.1, min_cxi = 0.8, max_sig = 0.05;
    bool use_scaling = true;
    double rho = 1.0, beta = 1000.0, gamma = 1.0, epsilon_con = 0.001, max_iter = 10.0;
    double g_epsilon = 1.0e-3, min_step = 1.0e-32, inner_max_iter = 10000.0, delta = 1.0e-4;
    int mem_size = 256, past = 3, int_K = 16;
    bool in_opt = false;
    bool in_test = false, in_debug = false;   // alm_traj_opt.h:56-57: loaded like the reference does; the test node's topics and the debug drawing stay on the host

    ALMTrajOpt() = default;
    ~ALMTrajOpt() { if (ctx_) uph_ctx_destroy(ctx_); }
    ALMTrajOpt(const ALMTrajOpt&) = delete;               // owns a device context (the reference's object is a value member that is never copied)
    ALMTrajOpt& operator=(const ALMTrajOpt&) = delete;

    // void ALMTrajOpt::init(ros::NodeHandle& nh)  (back_end/src/alm_traj_opt.cpp:5-29): the 21 optimiser keys + in_test / in_debug, read with
    // getParam exactly as the reference does -- a key the parameter server does not hold leaves the member as it was.  NodeHandle is a template
    // parameter (ros::NodeHandle in the workspace; anything with getParam(const std::string&, T&) elsewhere).  The publishers / subscribers of
    // :31-44 (RViz paths, the test node's goal topic) stay with the host.  Call order as in PlanManager::init (plan_manager.cpp:20-22):
    // init, setFrontend, setEnvironment -- setEnvironment hands the members to the device context.
    template <class NodeHandle>
    void init(NodeHandle& nh) {
        nh.getParam("alm_traj_opt/rho_T", rho_T);
        nh.getParam("alm_traj_opt/rho_ter", rho_ter);
        nh.getParam("alm_traj_opt/max_vel", max_vel);
        nh.getParam("alm_traj_opt/max_acc_lon", max_acc_lon);
        nh.getParam("alm_traj_opt/max_acc_lat", max_acc_lat);
        nh.getParam("alm_traj_opt/max_kap", max_kap);
        nh.getParam("alm_traj_opt/min_cxi", min_cxi);
        nh.getParam("alm_traj_opt/max_sig", max_sig);
        nh.getParam("alm_traj_opt/use_scaling", use_scaling);
        nh.getParam("alm_traj_opt/rho", rho);
        nh.getParam("alm_traj_opt/beta", beta);
        nh.getParam("alm_traj_opt/gamma", gamma);
        nh.getParam("alm_traj_opt/epsilon_con", epsilon_con);
        nh.getParam("alm_traj_opt/max_iter", max_iter);
        nh.getParam("alm_traj_opt/g_epsilon", g_epsilon);
        nh.getParam("alm_traj_opt/min_step", min_step);
        nh.getParam("alm_traj_opt/inner_max_iter", inner_max_iter);
        nh.getParam("alm_traj_opt/delta", delta);
        nh.getParam("alm_traj_opt/mem_size", mem_size);
        nh.getParam("alm_traj_opt/past", past);
        nh.getParam("alm_traj_opt/int_K", int_K);
        nh.getParam("alm_traj_opt/in_test", in_test);
        nh.getParam("alm_traj_opt/in_debug", in_debug);
    }

    // the members as the C-ABI's parameter block (what setEnvironment hands to uph_ctx_create)
    uph_opt_params optParams() const {
        uph_opt_params p;
        p.rho_T = rho_T; p.rho_ter = rho_ter; p.max_vel = max_vel; p.max_acc_lon = max_acc_lon; p.max_acc_lat = max_acc_lat;
        p.max_kap = max_kap; p.min_cxi = min_cxi; p.max_sig = max_sig; p.use_scaling = use_scaling ? 1 : 0;
        p.rho = rho; p.beta = beta; p.gamma = gamma; p.epsilon_con = epsilon_con; p.max_iter = max_iter;
        p.g_epsilon = g_epsilon; p.min_step = min_step; p.inner_max_iter = inner_max_iter; p.delta = delta;
        p.mem_size = mem_size; p.past = past; p.int_K = int_K;
        return p;
    }

    void setEnvironment(UnevenMapHandle* env) {          // alm_traj_opt.h:127-130
        env_ = env;
        if (ctx_) { uph_ctx_destroy(ctx_); ctx_ = nullptr; }
        const uph_opt_params p = optParams();
        if (uph_ctx_create(env->get(), &p, &ctx_) != UPH_OK) throw std::runtime_error(std::string("uph_ctx_create: ") + uph_last_error());
    }

    // int ALMTrajOpt::optimizeSE2Traj(initStateXY 2x3, endStateXY 2x3, innerPtsXY 2x(Nxy-1), initYaw 3, endYaw 3, innerPtsYaw, totalTime)
    template <class MatXY, class MatIn, class VecYaw, class VecIn>
    int optimizeSE2Traj(const MatXY& initStateXY, const MatXY& endStateXY, const MatIn& innerPtsXY, const VecYaw& initYaw, const VecYaw& endYaw,
                        const VecIn& innerPtsYaw, const double& totalTime) {
        in_opt = true;
        uph_problem pr;
        pr.n_inner_xy = (int32_t)innerPtsXY.cols();
        pr.n_inner_yaw = (int32_t)innerPtsYaw.size();
        for (int k = 0; k < 6; k++) { pr.init_xy[k] = initStateXY.data()[k]; pr.end_xy[k] = endStateXY.data()[k]; }   // column-major 2x3
        for (int k = 0; k < 3; k++) { pr.init_yaw[k] = initYaw.data()[k]; pr.end_yaw[k] = endYaw.data()[k]; }
        pr.inner_xy = innerPtsXY.data();
        pr.inner_yaw = innerPtsYaw.data();
        pr.total_time = totalTime;
        const int Nxy = pr.n_inner_xy + 1, Nyaw = pr.n_inner_yaw + 1;
        cxy_.assign((size_t)12 * Nxy, 0.0);
        cyaw_.assign((size_t)6 * Nyaw, 0.0);
        x_.assign((size_t)2 * pr.n_inner_xy + pr.n_inner_yaw + 1, 0.0);
        uph_result rs{};
        rs.x_final = x_.data(); rs.c_xy = cxy_.data(); rs.c_yaw = cyaw_.data();
        const int rc = uph_optimize_batch(ctx_, 1, &pr, &rs);
        in_opt = false;
        last_multi_ = false;
        if (rc != UPH_OK) return 1;                       // solver error, as the reference reports a hard L-BFGS failure
        last_ = rs;
        rho = rs.rho_final;                               // rho is a member that persists (alm_traj_opt.h:137)
        return rs.ret_code;
    }

    SE2Trajectory getTraj() const {                        // alm_traj_opt.h:165-168
        return makeTraj(cxy_.data(), (int)cxy_.size() / 12, last_.piece_T_xy, cyaw_.data(), (int)cyaw_.size() / 6, last_.piece_T_yaw);
    }

    // ---- batch form (no counterpart in the reference: many candidate goals in one call).  paths[b] = the poses (x, y, yaw) the front-end
    // returned for goal b (kino_astar->plan, plan_manager.cpp:56-60; any container of things indexable by [0..2], e.g.
    // std::vector<Eigen::Vector3d>).  Runs the stage of plan_manager.cpp:62-132 for all of them (uph_resample_batch), one
    // uph_optimize_batch, and returns one trajectory + return code per path.  ret[b] == UPH_RET_UNSUPPORTED marks a path outside the
    // compiled limits (its trajectory is empty); rho is left as it was (a batch has no "next call").
    struct BatchPlan {
        std::vector<int> ret;
        std::vector<SE2Trajectory> traj;
        std::vector<double> jerk_cost, total_time;
    };
    // peers: optimisers bound to the OTHER GPUs' copies of the map (setEnvironment done); the batch is then split over this object's device
    // and theirs (uph_optimize_batch_multi: one host thread per device, results in the caller's order)
    template <class Path>
    BatchPlan optimizeSE2TrajBatch(const std::vector<Path>& paths, const uph_manager_params& mgr, const std::vector<ALMTrajOpt*>& peers = {}) {
        const int32_t B = (int32_t)paths.size(), cap_xy = 2 * UPH_MAX_PIECE_XY, cap_yaw = 2 * UPH_MAX_PIECE_YAW;
        BatchPlan out;
        if (B == 0) return out;
        std::vector<int64_t> off((size_t)B + 1, 0);
        for (int32_t b = 0; b < B; b++) off[b + 1] = off[b] + (int64_t)paths[b].size();
        std::vector<double> flat((size_t)off[B] * 3);
        for (int32_t b = 0; b < B; b++)
            for (size_t i = 0; i < paths[b].size(); i++)
                for (int k = 0; k < 3; k++) flat[((size_t)off[b] + i) * 3 + k] = paths[b][i][k];
        std::vector<double> ixy((size_t)B * 6), exy((size_t)B * 6), iyw((size_t)B * 3), eyw((size_t)B * 3), oxy((size_t)B * 2 * cap_xy), oyw((size_t)B * cap_yaw), tt(B);
        std::vector<int32_t> nxy(B), nyw(B);
        const int rrc = uph_resample_batch(&mgr, B, flat.data(), off.data(), cap_xy, cap_yaw, ixy.data(), exy.data(), iyw.data(), eyw.data(), oxy.data(), oyw.data(),
                                           nxy.data(), nyw.data(), tt.data(), nullptr);
        if (rrc != UPH_OK && rrc != UPH_ERR_LIMIT) throw std::runtime_error(std::string("uph_resample_batch: ") + uph_last_error());
        // UPH_ERR_LIMIT: some path needs more way-points than the buffers hold (the counts are still reported).  Such a path is far beyond
        // the compiled piece limits anyway: clamp its counts to just above them so that the upload marks exactly that slot UNSUPPORTED and
        // the other goals are solved (the documented contract), instead of failing the whole batch
        for (int32_t b = 0; b < B; b++)
            if (nxy[b] > cap_xy || nyw[b] > cap_yaw) { nxy[b] = UPH_MAX_PIECE_XY; nyw[b] = UPH_MAX_PIECE_YAW; }
        std::vector<uph_problem> pr(B);
        std::vector<uph_result> rs(B);
        std::vector<size_t> ox(B), oc(B), oy(B);
        size_t sx = 0, sc = 0, sy = 0;
        for (int32_t b = 0; b < B; b++) {
            ox[b] = sx; oc[b] = sc; oy[b] = sy;
            sx += (size_t)2 * nxy[b] + nyw[b] + 1; sc += (size_t)12 * (nxy[b] + 1); sy += (size_t)6 * (nyw[b] + 1);
        }
        std::vector<double> xs(sx, 0.0), cx(sc, 0.0), cy(sy, 0.0);
        for (int32_t b = 0; b < B; b++) {
            uph_problem& p = pr[b];
            p.n_inner_xy = nxy[b]; p.n_inner_yaw = nyw[b];
            for (int k = 0; k < 6; k++) { p.init_xy[k] = ixy[(size_t)6 * b + k]; p.end_xy[k] = exy[(size_t)6 * b + k]; }
            for (int k = 0; k < 3; k++) { p.init_yaw[k] = iyw[(size_t)3 * b + k]; p.end_yaw[k] = eyw[(size_t)3 * b + k]; }
            p.inner_xy = oxy.data() + (size_t)2 * cap_xy * b;
            p.inner_yaw = oyw.data() + (size_t)cap_yaw * b;
            p.total_time = tt[b];
            rs[b] = uph_result{};
            rs[b].x_final = xs.data() + ox[b]; rs[b].c_xy = cx.data() + oc[b]; rs[b].c_yaw = cy.data() + oy[b];
        }
        in_opt = true;
        std::vector<uph_ctx*> cs(1, ctx_);
        for (ALMTrajOpt* q : peers) cs.push_back(q->ctx_);
        last_report_.clear(); last_multi_ = false;
        const int rc = uph_optimize_batch_multi(cs.data(), (int32_t)cs.size(), B, pr.data(), rs.data());
        in_opt = false;
        if (rc != UPH_OK) throw std::runtime_error(std::string("uph_optimize_batch_multi: ") + uph_last_error());
        if (cs.size() > 1) {
            // the batch now lives in shares on several devices.  The report rows are fetched HERE, while every context still holds its share
            // of THIS batch (a peer may be destroyed or run another solve before the caller asks), and go back to the caller's order through
            // uph_batch_origin
            last_report_.assign((size_t)7 * B, 0.0);
            for (uph_ctx* c : cs) {
                const int n = uph_batch_count(c);
                if (n <= 0) continue;                                                     // (a device whose share was empty or wholly unsupported)
                std::vector<double> part((size_t)7 * n);
                std::vector<int32_t> idx(n);
                if (uph_report_batch(c, part.data()) != UPH_OK) throw std::runtime_error(std::string("uph_report_batch: ") + uph_last_error());
                if (uph_batch_origin(c, idx.data()) != UPH_OK) throw std::runtime_error(std::string("uph_batch_origin: ") + uph_last_error());
                for (int k = 0; k < n; k++)
                    if (idx[k] >= 0 && idx[k] < B) for (int q = 0; q < 7; q++) last_report_[(size_t)7 * idx[k] + q] = part[(size_t)7 * k + q];
            }
            last_multi_ = true;
        }
        for (int32_t b = 0; b < B; b++) {
            out.ret.push_back(rs[b].ret_code);
            out.jerk_cost.push_back(rs[b].jerk_cost);
            out.total_time.push_back(tt[b]);
            out.traj.push_back(rs[b].ret_code == UPH_RET_UNSUPPORTED ? SE2Trajectory()
                                   : makeTraj(cx.data() + oc[b], nxy[b] + 1, rs[b].piece_T_xy, cy.data() + oy[b], nyw[b] + 1, rs[b].piece_T_yaw));
        }
        return out;
    }
    double getTrajJerkCost() const { return last_.jerk_cost; }   // minco_se2.getTrajJerkCost() (alm_traj_opt.cpp:273)

    // getMaxVxAxAyCurAttSig (alm_traj_opt.h:170-229): max vx, ax, ay, curvature, attitude (-cos xi), sigma sampled every 0.01 s -- evaluated
    // on the device for the trajectory of the last optimizeSE2Traj (the argument is what getTraj() returned for it)
    std::vector<double> getMaxVxAxAyCurAttSig(const SE2Trajectory&) {
        const std::vector<double> all = getMaxVxAxAyCurAttSigBatch();     // (after optimizeSE2TrajBatch the context holds B trajectories: row 0)
        return std::vector<double>(all.begin(), all.begin() + 6);
    }
    // the same report for every trajectory of the last call: [B][7] = max vx, ax, ay, cur, att, sigma, non-holonomic error (rows of
    // UPH_RET_UNSUPPORTED problems describe a placeholder, not a path)
    // After a call with `peers` the rows were collected from every device's share at the end of that call (cached here).
    std::vector<double> getMaxVxAxAyCurAttSigBatch() {
        if (last_multi_) return last_report_;
        const int B = uph_batch_count(ctx_);
        if (B <= 0) throw std::runtime_error("getMaxVxAxAyCurAttSig: no trajectory has been optimised on this object");
        std::vector<double> o((size_t)7 * B, 0.0);
        if (uph_report_batch(ctx_, o.data()) != UPH_OK) throw std::runtime_error(std::string("uph_report_batch: ") + uph_last_error());
        return o;
    }
    // members PlanManager calls that have no device side: the A* handle, RViz output
    template <class FrontendPtr> void setFrontend(const FrontendPtr&) {}
    void visSE2Traj(const SE2Trajectory&) {}
    void visSE3Traj(const SE2Trajectory&) {}

private:
    // coefficient blocks of the C-ABI (lowest order first, xy interleaved per row) -> pieces with the highest order first (se2traj.hpp:682-695)
    static SE2Trajectory makeTraj(const double* cxy, int Nxy, double Txy, const double* cyaw, int Nyaw, double Tyaw) {
        SE2Trajectory t;
        for (int i = 0; i < Nxy; i++) {
            Piece<2> p; p.duration = Txy;
            for (int d = 0; d < 2; d++) for (int k = 0; k < 6; k++) p.coeff[d][5 - k] = cxy[(size_t)(6 * i + k) * 2 + d];
            t.pos_traj.emplace_back(p);
        }
        for (int i = 0; i < Nyaw; i++) {
            Piece<1> p; p.duration = Tyaw;
            for (int k = 0; k < 6; k++) p.coeff[0][5 - k] = cyaw[(size_t)6 * i + k];
            t.yaw_traj.emplace_back(p);
        }
        return t;
    }

    UnevenMapHandle* env_ = nullptr;
    uph_ctx* ctx_ = nullptr;
    uph_result last_{};
    std::vector<double> cxy_, cyaw_, x_;
    std::vector<double> last_report_;       // [B][7] of the last optimizeSE2TrajBatch over several devices, in the caller's order
    bool last_multi_ = false;
};


// KinoAstar (front_end/include/front_end/kino_astar.h:99-168): the front-end PlanManager calls right before the back-end
// (`kino_astar->plan(start_state, end_state)`, plan_manager.cpp:59-60).  Same public parameter members as the reference reads from rosparam
// (kino_astar.cpp:7-20; defaults = run_hill.yaml:16-30), same plan() signature; the search runs on the device, one wave64 per query.
class KinoAstar {
public:
    double yaw_resolution = 3.15, lambda_heu = 1.0, weight_r2 = 1.0, weight_so2 = 0.5, weight_v_change = 0.0, weight_delta_change = 0.0, weight_sigma = 10.0;
    double time_interval = 0.3, collision_interval = 0.06, oneshot_range = 1.0, wheel_base = 0.26, max_steer = 0.5, max_vel = 0.5;
    bool in_test = false;

    KinoAstar() = default;
    ~KinoAstar() { if (k_) uph_kino_destroy(k_); }
    KinoAstar(const KinoAstar&) = delete;
    KinoAstar& operator=(const KinoAstar&) = delete;

    // void KinoAstar::init(ros::NodeHandle& nh)  (front_end/src/kino_astar.cpp:5-20): nh.param with the REFERENCE's defaults -- a key the
    // parameter server does not hold sets the member to that default (weight_so2 1.0, weight_sigma 0.0, time_interval 1.0, ...: not the YAML
    // values the members start with).  The publishers / subscribers of :22-30 and the RViz model of :36-43 stay with the host; the Dubins
    // radius wheel_base / tan(max_steer) (:34) is formed by uph_kino_create from the parameters setEnvironment hands over.
    template <class NodeHandle>
    void init(NodeHandle& nh) {
        nh.param("kino_astar/yaw_resolution", yaw_resolution, 3.15);
        nh.param("kino_astar/lambda_heu", lambda_heu, 1.0);
        nh.param("kino_astar/weight_r2", weight_r2, 1.0);
        nh.param("kino_astar/weight_so2", weight_so2, 1.0);
        nh.param("kino_astar/weight_v_change", weight_v_change, 0.0);
        nh.param("kino_astar/weight_delta_change", weight_delta_change, 0.0);
        nh.param("kino_astar/weight_sigma", weight_sigma, 0.0);
        nh.param("kino_astar/time_interval", time_interval, 1.0);
        nh.param("kino_astar/collision_interval", collision_interval, 1.0);
        nh.param("kino_astar/oneshot_range", oneshot_range, 1.0);
        nh.param("kino_astar/wheel_base", wheel_base, 1.0);
        nh.param("kino_astar/max_steer", max_steer, 1.0);
        nh.param("kino_astar/max_vel", max_vel, 1.0);
        nh.param("kino_astar/in_test", in_test, false);
    }
    uph_kino_params kinoParams() const {
        return uph_kino_params{yaw_resolution, lambda_heu, weight_r2, weight_so2, weight_v_change, weight_delta_change, weight_sigma,
                               time_interval, collision_interval, oneshot_range, wheel_base, max_steer, max_vel};
    }
    void setEnvironment(UnevenMapHandle* env, int slots = 0) {        // kino_astar.h:170-178: binds the map, allocates the node pools
        if (k_) { uph_kino_destroy(k_); k_ = nullptr; }
        const uph_kino_params kp = kinoParams();
        if (uph_kino_create(env->get(), &kp, slots, &k_) != UPH_OK) throw std::runtime_error(std::string("uph_kino_create: ") + uph_last_error());
    }
    // std::vector<Eigen::Vector3d> plan(const Eigen::Vector3d& start_state, const Eigen::Vector3d& end_state)   (kino_astar.cpp:67-236)
    template <class V3>
    std::vector<VecN<3>> plan(const V3& start_state, const V3& end_state) {
        std::vector<std::vector<VecN<3>>> r = planBatch(std::vector<V3>(1, start_state), std::vector<V3>(1, end_state));
        front_end_path = r[0];
        return front_end_path;
    }
    // many goals at once: paths[b] is empty where the reference would return an empty vector (start / goal occupied, no path, pool exhausted);
    // status[b] says which (UPH_KINO_*)
    template <class V3>
    std::vector<std::vector<VecN<3>>> planBatch(const std::vector<V3>& starts, const std::vector<V3>& goals, int path_cap = 2048) {
        if (!k_) throw std::runtime_error("KinoAstar: setEnvironment has not been called");
        const int32_t B = (int32_t)starts.size();
        std::vector<std::vector<VecN<3>>> out((size_t)B);
        status.assign((size_t)B, 0); iter_num.assign((size_t)B, 0);
        if (B == 0 || goals.size() != starts.size()) return out;
        std::vector<double> s((size_t)3 * B), g((size_t)3 * B), paths((size_t)3 * B * path_cap);
        std::vector<int32_t> np(B), use(B);
        for (int32_t b = 0; b < B; b++) for (int k = 0; k < 3; k++) { s[(size_t)3 * b + k] = starts[b][k]; g[(size_t)3 * b + k] = goals[b][k]; }
        if (uph_kino_plan_batch(k_, B, s.data(), g.data(), path_cap, paths.data(), np.data(), status.data(), iter_num.data(), use.data(), 0, 0, nullptr) != UPH_OK)
            throw std::runtime_error(std::string("uph_kino_plan_batch: ") + uph_last_error());
        std::vector<int32_t> longer;                    // queries whose front_end_path has more poses than path_cap: searched again with room for all of them
        int32_t need = 0;                               // (the reference returns the whole path; a clipped one would end short of the goal)
        for (int32_t b = 0; b < B; b++) {
            if (status[b] != UPH_KINO_OK) continue;
            if (np[b] > path_cap) { longer.push_back(b); if (np[b] > need) need = np[b]; continue; }
            out[b].resize((size_t)np[b]);
            for (int i = 0; i < np[b]; i++) for (int k = 0; k < 3; k++) out[b][i][k] = paths[((size_t)b * path_cap + i) * 3 + k];
        }
        if (!longer.empty()) {
            const int32_t L = (int32_t)longer.size();
            std::vector<double> s2((size_t)3 * L), g2((size_t)3 * L), p2((size_t)3 * L * need);
            std::vector<int32_t> np2(L), st2(L), it2(L), us2(L);
            for (int32_t j = 0; j < L; j++) for (int k = 0; k < 3; k++) { s2[(size_t)3 * j + k] = s[(size_t)3 * longer[j] + k]; g2[(size_t)3 * j + k] = g[(size_t)3 * longer[j] + k]; }
            if (uph_kino_plan_batch(k_, L, s2.data(), g2.data(), need, p2.data(), np2.data(), st2.data(), it2.data(), us2.data(), 0, 0, nullptr) != UPH_OK)
                throw std::runtime_error(std::string("uph_kino_plan_batch: ") + uph_last_error());
            for (int32_t j = 0; j < L; j++) {
                const int32_t b = longer[j];
                if (st2[j] != UPH_KINO_OK || np2[j] > need) { status[b] = UPH_KINO_INTERNAL; continue; }      // (the search is deterministic: not expected)
                out[b].resize((size_t)np2[j]);
                for (int i = 0; i < np2[j]; i++) for (int k = 0; k < 3; k++) out[b][i][k] = p2[((size_t)j * need + i) * 3 + k];
            }
        }
        return out;
    }
    std::vector<VecN<3>> front_end_path;
    std::vector<int32_t> status, iter_num;      // of the last plan / planBatch
    void visFrontEnd() {}                        // RViz output stays with the host (no device side)
    void visExpanded() {}

private:
    uph_kino* k_ = nullptr;
};

}  // namespace uneven_hip
