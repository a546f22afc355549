#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's command next to the bench's own HIP-event figure of the SAME run: the k_phys dispatches of the timed
# region (the last `steps` full launches; the stats file's average also holds the warm-up, reset and pose-pass launches) -> gpurun_out/<tag>/kernel_trace_agree.json
tag=${1:-agree}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag; mkdir -p $o
rocprofv3 --kernel-trace --stats -d $o/trace -o t -f csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $o/bench.json 2> $o/bench.err
python - <<PY
import csv, glob, json
o = "$o"
line = json.loads(open(o + "/bench.json").read().strip().splitlines()[-1])
f = glob.glob(o + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_phys" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
full = [d for d in dur if d > 0.5 * max(dur)]
timed = full[-line["steps"]:]
st = glob.glob(o + "/trace/**/*kernel_stats.csv", recursive=True)[0]
open(o + "/kernel_stats.csv", "w").write(open(st).read())
out = {"command": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras",
       "k_phys_dispatches": len(dur), "k_phys_ms_all_dispatches_mean": sum(dur) / len(dur),
       "k_phys_ms_timed_region": timed, "k_phys_ms_timed_region_mean": sum(timed) / len(timed),
       "bench_kernel_avg_ms_same_run": line["roofline"]["kernel_avg_ms"], "bench_ms_per_step_same_run": line["ms_per_step"], "bench_value_same_run": line["value"],
       "note": "the stats file's AverageNs is over ALL k_phys dispatches of the process (warm-up steps, the reset's forward pass, the trailing pose passes of a few hundred microseconds); the timed region's dispatches are the last --steps full-length ones"}
json.dump(out, open(o + "/kernel_trace_agree.json", "w"), indent=1)
print({k: v for k, v in out.items() if k != "k_phys_ms_timed_region" and k != "note"})
PY
