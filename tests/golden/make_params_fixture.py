"""Writes tests/golden/run_params.json: the VALUES of the reference's five parameter files (plan_manager/params/run_*.yaml, sections uneven_map,
kino_astar, alm_traj_opt, manager) as plain data -- what a ROS parameter server holds after `roslaunch plan_manager run_<scene>.launch`.
Run in the build container (needs /root/reference); the JSON is what travels."""
import json
import os
import yaml

REF = "/root/reference/src/uneven_planner/plan_manager/params"
out = {}
for scene in ("hill", "desert", "vocano", "forest", "mountain"):
    doc = yaml.safe_load(open(os.path.join(REF, "run_%s.yaml" % scene)))["manager_node"]
    out[scene] = {sec: doc[sec] for sec in ("uneven_map", "kino_astar", "alm_traj_opt", "manager")}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "run_params.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print("wrote", path, {k: len(v["alm_traj_opt"]) for k, v in out.items()})
