"""ISA instruction census of one rollout step -> profiles/<tag>_rollout_isa_census.json (runs on the CPU: hipcc -S).

  python scripts/rollout_census.py [out.json]

Compiles csrc/rollout.hip to assembly, takes each model's driver kernel in its straight-line form (FAST, time-blocked,
8 steps per trip of the block loop: the variant bench.py's C2 workload launches; the rolled and returns-only forms of
larger batches run the same arithmetic per step around different loop / addressing code) and counts the instructions
of that loop by category.  bench.py uses `fp64_flops_per_lane_step` for the rollout's compute roofline; the stamp ties the file to the
sources it was made from."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from trajopt_amd import _lib  # noqa: E402

# <SYS, MINOR, KMINOR (the gains' layout in the driver: PROBLEM_MINOR for dx <= 6), FAST, TB, UNR, CO, KL>
KERNELS = {"cartpole": "rollout_kernelINS_7SysPhysILi2EEELb1ELb1ELb1ELb1ELb1ELb0ELb0EEEvNS_11RolloutArgsE",
           "pendulum": "rollout_kernelINS_7SysPhysILi1EEELb1ELb1ELb1ELb1ELb1ELb0ELb0EEEvNS_11RolloutArgsE",
           "double_cartpole": "rollout_kernelINS_7SysPhysILi3EEELb1ELb1ELb1ELb1ELb1ELb0ELb0EEEvNS_11RolloutArgsE",
           # dx + du > 8: half-block stage, gains through LDS (KL), PROBLEM_MAJOR gains
           "quadrotor": "rollout_kernelINS_7SysPhysILi5EEELb1ELb0ELb1ELb1ELb0ELb0ELb1EEEvNS_11RolloutArgsE"}
# steps of the time loop in one trip of the counted loop: 8 for the straight-line form, 4 for the half-block form
STEPS_PER_TRIP = {"quadrotor": 4}


ROLLOUT_SOURCES = _lib.ROLLOUT_SOURCES


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_rollout_isa_census.json")
    res = {"stamp": _lib.source_stamp(), "rollout_stamp": _lib.source_stamp(ROLLOUT_SOURCES), "steps_per_trip": 8, "kernels": {},
           "note": "block loop of the driver's rollout kernel (8 steps per trip, straight-line); counts are "
                   "instructions per wave; fp64 flops per lane: 2 per fma, 1 per mul / add"}
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, "rollout.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                               os.path.join(ROOT, "trajopt_amd", "csrc", "rollout.hip"), "-o", asm], stderr=subprocess.DEVNULL)
        for env, key in KERNELS.items():
            js = os.path.join(d, env + ".json")
            subprocess.check_output([sys.executable, os.path.join(ROOT, "scripts", "isa_census.py"), asm, key, js])
            loops = json.load(open(js))["loops"]
            # the largest loop closed by a conditional branch = the block loop (8 straight-line steps)
            # (the half-block form's 4-step loop closes with an unconditional branch; its tail loop of single steps is smaller)
            big = max((l for l in loops if l["conditional"] or env in STEPS_PER_TRIP), key=lambda l: l["instructions"])
            spt = float(STEPS_PER_TRIP.get(env, 8))
            per_step = {k: v / spt for k, v in big["by_category"].items()}
            res["kernels"][env] = {"mangled": key, "trip_instructions": big["instructions"], "steps_per_trip": spt,
                                   "instructions_per_step": big["instructions"] / spt,
                                   "valu_per_step": big["valu"] / spt,
                                   "fp64_flops_per_lane_step": big["fp64_flops_per_lane"] / spt,
                                   "per_step_by_category": per_step}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: (v["instructions_per_step"], v["fp64_flops_per_lane_step"]) for k, v in res["kernels"].items()}))


if __name__ == "__main__":
    main()
