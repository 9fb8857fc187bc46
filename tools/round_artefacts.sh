#!/usr/bin/env bash
# The profiles of record of a round, in one GPU-box call:  tools/round_artefacts.sh TAG   -> gpurun_out/prof_TAG/
#   kernel traces + PMC traffic of the four workloads (tools/profile_workloads.py), SQ counters of the cfg2 launch (three passes of
#   tools/pmc_pass.py), phase clocks (ab_libs/libfl_timing.so = tools/build_variant.sh timing -DFL_OBS_TIMING), the default bench line
set -uo pipefail
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
# (PART=traces: only this step; PART=rest: everything after it -- two GPU-box calls when one would not fit the call's time limit)
if [ "${PART:-all}" != "rest" ]; then
python tools/profile_workloads.py $tag cfg2:2 cfg3:3 cfg4:2 cfg5:3:rebuild cfg2:2:distinct10 cfg3:3:distinct10 cfg4:2:distinct4 cfg5:3:rebuild:distinct2 > $out/profile_workloads.log 2>&1 || tail -5 $out/profile_workloads.log
fi
[ "${PART:-all}" = "traces" ] && exit 0
echo "[artefacts] traces + traffic done"
A="--no-extra-workloads --steps 100"
python tools/pmc_pass.py $out/sq1.json "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" $A > /dev/null 2>&1 || echo "sq pass 1 failed"
python tools/pmc_pass.py $out/sq2.json "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $A > /dev/null 2>&1 || echo "sq pass 2 failed"
python tools/pmc_pass.py $out/sq3.json "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM" $A > /dev/null 2>&1 || echo "sq pass 3 failed (counter names)"
python - $out $tag <<'PY'
import json, os, sys
out, tag = sys.argv[1], sys.argv[2]
acc = {}
for k in (1, 2, 3):
    p = os.path.join(out, "sq%d.json" % k)
    if os.path.exists(p):
        for kern, d in json.load(open(p)).items():
            acc.setdefault(kern, {}).update(d)
sys.path.insert(0, os.getcwd())
import bench
acc["kernel_source_sha"] = bench.kernel_source_sha()   # (the bench line quotes the VALU issue fraction only for these sources)
acc["tag"] = tag
json.dump(acc, open(os.path.join(out, "%s_sq_counters_cfg2.json" % tag), "w"), indent=1, sort_keys=True)
PY
cp $out/${tag}_sq_counters_cfg2.json profiles/ 2>/dev/null   # (on the GPU box: the default bench line below quotes it as roofline.valu_issue)
echo "[artefacts] SQ counters done"
# (the timing variant is built HERE, on the box, when it did not travel: no toggling of .gpurunignore)
[ -f ab_libs/libfl_timing.so ] || tools/build_variant.sh timing -DFL_OBS_TIMING > $out/build_timing.log 2>&1 || tail -3 $out/build_timing.log
if [ -f ab_libs/libfl_timing.so ]; then
  : > $out/${tag}_phase_clocks.txt
  # (the clocks are taken in the bench's regime: de-phased replicas, see tools/obs_phase_clocks.py)
  for w in "cfg2 2" "cfg3 3" "cfg4 2" "cfg5 3" "cfg4 2 4" "cfg5 3 2"; do
    python tools/obs_phase_clocks.py ab_libs/libfl_timing.so $w 2>&1 | grep -v amdgpu.ids >> $out/${tag}_phase_clocks.txt
    echo >> $out/${tag}_phase_clocks.txt
  done
fi
echo "[artefacts] phase clocks done"
python bench.py > $out/${tag}_bench_default.json 2> $out/bench_default.err || tail -3 $out/bench_default.err
python -c "
import json,sys; d=json.load(open('$out/${tag}_bench_default.json')); print('headline %.2f M' % (d['value']/1e6), d['roofline'], {k: round(v['value']/1e6,1) for k,v in d.get('workloads',{}).items()})"
# round 5: throughput against the number of envs at the cfg2 shape (one / two workgroups a CU, fixed classes and runtime carving), the
# latency of the drop-in plug-in, the soak of the non-BASELINE shapes
python tools/bsweep.py --bs 256,384,512,768,1024,2048 --modes one_a_cu,two_a_cu,default,nofix,nofix_two_a_cu --out $out/${tag}_cfg2_bsweep.json > $out/bsweep.log 2>&1 || tail -3 $out/bsweep.log
python tools/plugin_latency.py --out $out/${tag}_plugin_latency.json > $out/plugin.log 2>&1 || tail -3 $out/plugin.log
python tools/soak_round2.py 320 1 $out/${tag}_soak_round2.txt > $out/soak.log 2>&1 || tail -3 $out/soak.log
echo "[artefacts] sweep, plug-in latency, soak done"
