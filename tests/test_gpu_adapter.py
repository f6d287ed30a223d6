"""The C++ boundary, executed: a small C++17 program is built against include/uneven_hip_adapter.hpp + libunevenhip.so, runs the goal
callback's call sequence (tests/test_abi_cpu.py CONSUMER: optimizeSE2Traj -> getTraj -> report -> SE2Traj message) on the hill problem
and must reproduce, bit for bit, what the ctypes mirror gets from the same C-ABI; the message is checked against the trajectory
evaluated independently in numpy (mpc_controller/msg/SE2Traj.msg:1-9, plan_manager.cpp:150-182)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from test_abi_cpu import CONSUMER

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MAIN = r"""
#include <cstdio>
#include <cstdlib>
static std::vector<double> readv(FILE* f, size_t n) { std::vector<double> v(n); if (fread(v.data(), 8, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(3); } return v; }
int main(int argc, char** argv) {
    // in:  header {ncell, n_inner_xy, n_inner_yaw}, cells[ncell*4], init_xy[6], end_xy[6], inner_xy[2*nxy], init_yaw[3], end_yaw[3], inner_yaw[nyaw], total_time
    FILE* f = std::fopen(argv[1], "rb");
    long long hdr[3];
    if (!f || fread(hdr, 8, 3, f) != 3) return 2;
    const long long ncell = hdr[0]; const int nxy = (int)hdr[1], nyaw = (int)hdr[2];
    std::vector<double> cells = readv(f, (size_t)ncell * 4);
    Mat init_xy(2, 3), end_xy(2, 3), inner_xy(2, nxy), init_yaw(3, 1), end_yaw(3, 1), inner_yaw(nyaw, 1);
    init_xy.v = readv(f, 6); end_xy.v = readv(f, 6); inner_xy.v = readv(f, 2 * (size_t)nxy);
    init_yaw.v = readv(f, 3); end_yaw.v = readv(f, 3); inner_yaw.v = readv(f, nyaw);
    const double total_time = readv(f, 1)[0];
    std::fclose(f);
    uph_map_params mp = {2, 10.0, 10.0, 0.2, 0.1, 0.1, 0.05, 0.1, 0.8, 0.05, 9.81};
    UnevenMapHandle map(mp, 0);
    map.setCells(cells.data());
    SE2TrajMsg msg;
    std::vector<double> report;
    const int rc = plan_once(map, init_xy, end_xy, inner_xy, init_yaw, end_yaw, inner_yaw, total_time, msg, report);
    FILE* o = std::fopen(argv[2], "wb");
    double head[4] = {(double)rc, (double)msg.pos_pts.size(), (double)msg.angle_pts.size(), (double)report.size()};
    fwrite(head, 8, 4, o);
    for (const Point3& p : msg.pos_pts) { fwrite(&p.x, 8, 1, o); fwrite(&p.y, 8, 1, o); }
    fwrite(msg.posT_pts.data(), 8, msg.posT_pts.size(), o);
    for (const Point3& p : msg.angle_pts) fwrite(&p.x, 8, 1, o);
    fwrite(msg.angleT_pts.data(), 8, msg.angleT_pts.size(), o);
    fwrite(report.data(), 8, report.size(), o);
    std::fclose(o);
    return 0;
}
"""


MAIN_BATCH = r"""
#include <cstdio>
#include <cstdlib>
static std::vector<double> readv(FILE* f, size_t n) { std::vector<double> v(n); if (fread(v.data(), 8, n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(3); } return v; }
int main(int argc, char** argv) {
    // in: header {ncell, B}, cells[ncell*4], B x {M, poses[M*3]}
    FILE* f = std::fopen(argv[1], "rb");
    long long hdr[2];
    if (!f || fread(hdr, 8, 2, f) != 2) return 2;
    std::vector<double> cells = readv(f, (size_t)hdr[0] * 4);
    std::vector<std::vector<VecN<3>>> paths((size_t)hdr[1]);
    for (auto& p : paths) {
        long long M;
        if (fread(&M, 8, 1, f) != 1) return 2;
        std::vector<double> v = readv(f, (size_t)M * 3);
        p.resize((size_t)M);
        for (long long i = 0; i < M; i++) for (int k = 0; k < 3; k++) p[i][k] = v[3 * i + k];
    }
    std::fclose(f);
    uph_map_params mp = {2, 10.0, 10.0, 0.2, 0.1, 0.1, 0.05, 0.1, 0.8, 0.05, 9.81};
    UnevenMapHandle map(mp, 0);
    map.setCells(cells.data());
    ALMTrajOpt::BatchPlan out = plan_many(map, paths);
    // out: per path {ret, jerk_cost, total_time, npos, nang, pos_pts xy, posT, angle_pts, angleT}
    FILE* o = std::fopen(argv[2], "wb");
    for (size_t b = 0; b < paths.size(); b++) {
        SE2TrajMsg msg;
        if (out.traj[b].pos_traj.getPieceNum() > 0) fillSE2TrajMsg(out.traj[b], msg);
        double head[5] = {(double)out.ret[b], out.jerk_cost[b], out.total_time[b], (double)msg.pos_pts.size(), (double)msg.angle_pts.size()};
        fwrite(head, 8, 5, o);
        for (const Point3& p : msg.pos_pts) { fwrite(&p.x, 8, 1, o); fwrite(&p.y, 8, 1, o); }
        fwrite(msg.posT_pts.data(), 8, msg.posT_pts.size(), o);
        for (const Point3& p : msg.angle_pts) fwrite(&p.x, 8, 1, o);
        fwrite(msg.angleT_pts.data(), 8, msg.angleT_pts.size(), o);
    }
    std::fclose(o);
    return 0;
}
"""


def _build(tmp_path, name, text):
    src = tmp_path / (name + ".cpp")
    src.write_text(text)
    exe = str(tmp_path / name)
    libdir = os.path.join(ROOT, "uneven_planner_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe, "-L", libdir, "-lunevenhip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_batch_plan_matches_ctypes_bit_for_bit(tmp_path, analytic_cells):
    """ALMTrajOpt::optimizeSE2TrajBatch (front-end paths -> uph_resample_batch -> uph_optimize_batch -> trajectories) from C++ against
    resample_batch + optimize_batch through ctypes; one path is a single piece (goal closer than a piece length), one has more pieces than the
    compiled limit and one more way-points than the adapter's resampling buffers hold (2 x the limit: the resampler reports UPH_ERR_LIMIT) -- both
    must come back UNSUPPORTED without disturbing the others"""
    import uneven_planner_amd as U
    from uneven_planner_amd import resample
    rng = np.random.default_rng(21)
    paths = []
    for _ in range(6):
        s = np.array([rng.uniform(-4, -1), rng.uniform(-4, 4), rng.uniform(-1, 1)])
        g = np.array([rng.uniform(1, 4), rng.uniform(-4, 4), rng.uniform(-1, 1)])
        paths.append(resample.hermite_path(s, g))
    paths.insert(2, resample.hermite_path((0.0, 0.0, 0.0), (0.2, 0.0, 0.0)))          # shorter than one piece: a single quintic, solved like the others
    tl = np.linspace(-4.5, 4.5, 1200)
    paths.insert(3, np.column_stack([tl, 4.0 * np.sin(3.0 * tl), np.arctan(12.0 * np.cos(3.0 * tl))]))     # ~70 m: > UPH_MAX_PIECE_XY pieces
    tl2 = np.linspace(-4.5, 4.5, 4000)
    paths.insert(5, np.column_stack([tl2, 4.0 * np.sin(7.0 * tl2), np.arctan(28.0 * np.cos(7.0 * tl2))]))   # ~160 m: > 2 x UPH_MAX_PIECE_XY way-points
    exe = _build(tmp_path, "batch_consumer", CONSUMER + MAIN_BATCH)
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    cells = np.ascontiguousarray(analytic_cells, dtype=np.float64)
    with open(fin, "wb") as f:
        f.write(struct.pack("<2q", cells.shape[0], len(paths)))
        f.write(cells.tobytes())
        for p in paths:
            f.write(struct.pack("<q", p.shape[0]))
            f.write(np.ascontiguousarray(p, dtype=np.float64).tobytes())
    subprocess.check_call([exe, fin, fout])
    raw = np.fromfile(fout, dtype=np.float64)
    m = U.UnevenMap()
    m.set_cells(analytic_cells)
    opt = U.ALMTrajOpt(m)
    probs = resample.resample_batch(paths, cap_xy=2048, cap_yaw=4096)
    assert 128 < probs[3]["inner_xy"].shape[1] + 1 <= 256 and probs[5]["inner_xy"].shape[1] > 256
    out = opt.optimize_batch(probs)
    o = 0
    for b, pr in enumerate(probs):
        ret, jc, tt, npos, nang = raw[o:o + 5]; o += 5
        npos, nang = int(npos), int(nang)
        assert int(ret) == out[b]["ret"] and tt == pr["total_time"]
        if b in (3, 5):
            assert int(ret) == 4 and npos == 0 and nang == 0                     # UPH_RET_UNSUPPORTED
            continue
        msg = opt.getTraj(b).to_msg()
        pos = raw[o:o + 2 * npos].reshape(npos, 2); o += 2 * npos
        posT = raw[o:o + npos - 1]; o += npos - 1
        ang = raw[o:o + nang]; o += nang
        angT = raw[o:o + nang - 1]; o += nang - 1
        assert jc == out[b]["jerk_cost"]
        assert np.array_equal(pos, msg["pos_pts"][:, :2]) and np.array_equal(ang, msg["angle_pts"][:, 0])
        assert np.array_equal(posT, msg["posT_pts"]) and np.array_equal(angT, msg["angleT_pts"])
    assert o == raw.size


def _locate(durs, t):                      # PolyTrajectory::locatePieceIdx (se2traj.hpp:343-361)
    idx = 0
    while idx < len(durs) and t > durs[idx]:
        t -= durs[idx]
        idx += 1
    if idx == len(durs):
        idx -= 1
        t += durs[idx]
    return idx, t


def _value(c_desc, t):                     # Piece::getValue (se2traj.hpp:106-116), coefficients highest order first
    v, tn = 0.0, 1.0
    for i in range(5, -1, -1):
        v += tn * c_desc[i]
        tn *= t
    return v


def test_cpp_consumer_matches_ctypes_bit_for_bit(tmp_path, analytic_cells):
    import uneven_planner_amd as U
    from uneven_planner_amd import scenes
    p = scenes.hill_problem()
    src = tmp_path / "consumer_main.cpp"
    src.write_text(CONSUMER + MAIN)
    exe = str(tmp_path / "consumer")
    libdir = os.path.join(ROOT, "uneven_planner_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), str(src), "-o", exe, "-L", libdir, "-lunevenhip",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    cells = np.ascontiguousarray(analytic_cells, dtype=np.float64)
    nxy, nyaw = p["inner_xy"].shape[1], p["inner_yaw"].shape[0]
    with open(fin, "wb") as f:
        f.write(struct.pack("<3q", cells.shape[0], nxy, nyaw))
        f.write(cells.tobytes())
        for a in (p["init_xy"].T, p["end_xy"].T, p["inner_xy"].T, p["init_yaw"], p["end_yaw"], p["inner_yaw"], np.array([p["total_time"]])):
            f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
    subprocess.check_call([exe, fin, fout])
    raw = np.fromfile(fout, dtype=np.float64)
    rc, npos, nang, nrep = (int(v) for v in raw[:4])
    o = 4
    pos = raw[o:o + 2 * npos].reshape(npos, 2); o += 2 * npos
    posT = raw[o:o + npos - 1]; o += npos - 1
    ang = raw[o:o + nang]; o += nang
    angT = raw[o:o + nang - 1]; o += nang - 1
    rep = raw[o:o + nrep]
    # the same solve through the ctypes mirror
    m = U.UnevenMap()
    m.set_cells(analytic_cells)
    opt = U.ALMTrajOpt(m)
    ret = opt.optimizeSE2Traj(p["init_xy"], p["end_xy"], p["inner_xy"], p["init_yaw"], p["end_yaw"], p["inner_yaw"], p["total_time"])
    tr = opt.getTraj()
    assert rc == ret and npos == nxy + 2 and nang == nyaw + 2
    msg = tr.to_msg()
    assert np.array_equal(pos, msg["pos_pts"][:, :2]) and np.array_equal(ang, msg["angle_pts"][:, 0])        # bit for bit
    assert np.array_equal(posT, msg["posT_pts"]) and np.array_equal(angT, msg["angleT_pts"])
    rep_py = opt.getMaxVxAxAyCurAttSig()[0]
    assert np.array_equal(rep[:6], rep_py[:6]) and abs(rep[6] - rep_py[6]) < 1e-9 * max(1.0, rep_py[6])      # host getNonHolError vs device sum
    # and the message against an independent evaluation of the stored coefficients
    cx = tr.pos_coeffs                                             # (piece, dim, 6) highest order first
    for i in range(nxy + 1):
        assert pos[i, 0] == cx[i, 0, 5] and pos[i, 1] == cx[i, 1, 5]                       # piece start = constant coefficient
    durs = [tr.T_xy] * (nxy + 1)
    idx, tl = _locate(durs, sum(durs))
    assert abs(pos[-1, 0] - _value(cx[idx, 0], tl)) < 1e-12 and abs(pos[-1, 1] - _value(cx[idx, 1], tl)) < 1e-12
    assert abs(pos[-1, 0] - p["end_xy"][0, 0]) < 1e-9 and abs(pos[-1, 1] - p["end_xy"][1, 0]) < 1e-9      # the end state of the problem
    assert abs(ang[-1] - p["end_yaw"][0]) < 1e-9 and abs(ang[0] - p["init_yaw"][0]) < 1e-12


def test_message_matches_the_oracles_trajectory(analytic_cells, oracle, oracle_grid, small_problems):
    """SE2Traj message fields from the device's trajectory vs the oracle's coefficients for the same x (short solve: strict)"""
    import uneven_planner_amd as U
    m = U.UnevenMap()
    m.set_cells(analytic_cells)
    prm = dict(inner_max_iter=3.0)
    opt = U.ALMTrajOpt(m, prm)
    opt.set_rho(1.0)
    out = opt.optimize_batch(small_problems)
    for i, p in enumerate(small_problems):
        a = oracle.OracleALM(oracle_grid, prm)
        ro = a.optimize(p)
        cxy, cyaw, txy, tyaw, _ = a.coeffs()
        msg = opt.getTraj(i).to_msg()
        nxy = cxy.shape[0] // 6
        want = np.vstack([cxy.reshape(nxy, 6, 2)[:, 0, :], [sum(cxy.reshape(nxy, 6, 2)[-1, k] * txy ** k for k in range(6))]])
        assert np.abs(msg["pos_pts"][:, :2] - want).max() < 1e-8
        assert np.allclose(msg["posT_pts"], txy, rtol=1e-9) and np.allclose(msg["angleT_pts"], tyaw, rtol=1e-9)
        nyw = cyaw.shape[0] // 6
        wanty = np.concatenate([cyaw.reshape(nyw, 6)[:, 0], [sum(cyaw.reshape(nyw, 6)[-1, k] * tyaw ** k for k in range(6))]])
        assert np.abs(msg["angle_pts"][:, 0] - wanty).max() < 1e-8
