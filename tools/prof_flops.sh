#!/bin/bash
# The physics kernel's OWN floating-point work per env-step (its sparse row windows and per-tree solves, not the oracle's dense rows):
# SQ instruction counters of k_phys in PMC-only passes (never combined with a trace), then
#   flops per launch = (ADD + MUL + 2 FMA + TRANS wave-instructions) x 64 lanes x lane utilisation
#   lane utilisation = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU x 64)     (rocprofv3's own VALU-utilisation expression)
# next to the hardware's SQ_INSTS_VALU_FLOPS_* counters.  usage: tools/prof_flops.sh <tag>   -> gpurun_out/flops_<tag>/kernel_flops.json
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
o=gpurun_out/flops_$tag
mkdir -p $o
run() {   # name, counters..., -- bench args
  name=$1; shift; ctr=""; while [ "$1" != "--" ]; do ctr="$ctr $1"; shift; done; shift
  rocprofv3 --pmc $ctr -d $o/$name -o p -f csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras "$@" > $o/$name.log 2>&1
}
F32="SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP32_TRANS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"
F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP64_TRANS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"
MISC="SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY"
run c2 $F32 -- 
run c2m $MISC --
run c3 $F32 -- --config 3 --warmup 150
run c4 $F32 -- --config 4 --warmup 30
run c2f64 $F64 -- --f64
run c2f64b $F32 -- --f64
python - <<PY
import csv, glob, json, collections
o = "$o"
def collect(name):
    # the TIMED env-step launches: dispatches in order, without the forward-only launches of reset / FK and the near-empty second
    # passes of the two-tier capacities (small), the last five of the rest -- the same launches for every counter
    out = {}
    for f in glob.glob(o + "/%s/**/*counter_collection.csv" % name, recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "k_phys" in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        if not per:
            continue
        ref = max(next(iter(per.values())).keys(), key=lambda k: max(d.get(k, 0.0) for d in per.values()))
        top = max(d.get(ref, 0.0) for d in per.values())
        ids = [i for i in sorted(per) if per[i].get(ref, 0.0) > 0.1 * top][-5:]
        for k in per[ids[0]]:
            out[k] = sum(per[i].get(k, 0.0) for i in ids) / len(ids)
    return out
N = 4096
res = {"source": "tools/prof_flops.sh $tag: rocprofv3 --pmc passes of bench.py --steps 5 (SQ counters of k_phys, mean over its env-step launches of 4096 envs x 20 substeps); flops = (ADD + MUL + 2 FMA + TRANS) wave-instructions x 64 x lane utilisation (SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)); packed v_pk_* instructions are counted by the SQ as one instruction per wave, so this is a lower bound where they are used"}
for key, name, p in (("config2", "c2", "F32"), ("config3", "c3", "F32"), ("config4", "c4", "F32"), ("config2_f64", "c2f64", "F64")):
    c = collect(name)
    if not c:
        continue
    add, mul, fma, tr = (c.get("SQ_INSTS_VALU_%s_%s" % (k, p), 0.0) for k in ("ADD", "MUL", "FMA", "TRANS"))
    util = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, 64.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0))
    winst = add + mul + 2 * fma + tr
    hw = c.get("SQ_INSTS_VALU_FLOPS_FP%s" % p[1:], 0.0)
    entry = {"wave_instructions": {"add": add, "mul": mul, "fma": fma, "trans": tr}, "lane_utilisation": util,
             "flops_per_launch": winst * 64 * util, "flops_per_env_step": winst * 64 * util / N,
             "hw_flops_counter_per_launch": hw, "hw_flops_counter_trans_per_launch": c.get("SQ_INSTS_VALU_FLOPS_FP%s_TRANS" % p[1:], 0.0),
             "source": res["source"]}
    if key == "config2_f64":
        c32 = collect("c2f64b")
        entry["f32_wave_instructions_in_the_f64_kernel"] = {k: c32.get("SQ_INSTS_VALU_%s_F32" % k, 0.0) for k in ("ADD", "MUL", "FMA", "TRANS")}
    res[key] = entry
m = collect("c2m")
res["config2_instruction_mix_per_launch"] = m
json.dump(res, open(o + "/kernel_flops.json", "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "source"}) for k, v in res.items() if k != "source"}, indent=1))
PY
