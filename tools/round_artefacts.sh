#!/usr/bin/env bash
# The profiles of record of a round, in a few GPU-box calls (a call is limited to 20 minutes):  PART=<part> tools/round_artefacts.sh TAG   -> gpurun_out/prof_TAG/
#   PART=traces1   kernel traces + PMC traffic (tools/profile_workloads.py) of the four BASELINE workloads
#   PART=traces2   ... of the same on distinct generated maps
#   PART=traces3   ... of the consumer's path (the flatland_cutils builder alone, fl_obs_cutils_policy) and of FL_OBS_KEEP_TREE_ROWS
#   PART=sq        SQ counters of the cfg2 launch (three passes of tools/pmc_pass.py) + phase clocks (ab_libs/libfl_timing.so, built on the box when absent)
#   PART=rest      the default bench line (run AFTER tools/install_artefacts.py TAG put the traffic / SQ files into profiles/: the line quotes them),
#                  throughput against the number of envs, the plug-in's latency, the soak and the class gains of the Round-2 table
set -uo pipefail
tag=$1
out=gpurun_out/prof_$tag
mkdir -p $out
part=${PART:-traces1}
export PART=$part
case $part in
traces1) python tools/profile_workloads.py $tag cfg2:2 cfg3:3 cfg4:2 cfg5:3:rebuild > $out/profile_workloads_$part.log 2>&1 || tail -5 $out/profile_workloads_$part.log ;;
traces2) python tools/profile_workloads.py $tag cfg2:2:distinct10 cfg3:3:distinct10 cfg4:2:distinct4 cfg5:3:rebuild:distinct2 > $out/profile_workloads_$part.log 2>&1 || tail -5 $out/profile_workloads_$part.log ;;
traces3) python tools/profile_workloads.py $tag cfg2:0:pack cfg3:0:pack cfg4:0:pack cfg5:0:pack cfg3:3:keeprows cfg5:3:rebuild:keeprows > $out/profile_workloads_$part.log 2>&1 || tail -5 $out/profile_workloads_$part.log ;;
sq)
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
echo "[artefacts] phase clocks done" ;;
rest)
python bench.py --detail $out/${tag}_bench_detail.json > $out/${tag}_bench_default.json 2> $out/bench_default.err || tail -3 $out/bench_default.err
python -c "
import json,sys; d=json.load(open('$out/${tag}_bench_default.json')); print('headline %.2f M' % (d['value']/1e6), d['roofline'], {k: round(v['value']/1e6,1) for k,v in d.get('workloads',{}).items()}); print('line bytes', len(open('$out/${tag}_bench_default.json').read()))"
python tools/bsweep.py --bs 256,512,1024,2048 --modes one_a_cu,two_a_cu,default,nofix --out $out/${tag}_cfg2_bsweep.json > $out/bsweep.log 2>&1 || tail -3 $out/bsweep.log
python tools/bsweep.py --depth 0 --pack 1 --bs 256,512,1024,2048 --modes default,one_a_cu --out $out/${tag}_cfg2_alone_bsweep.json > $out/bsweep_alone.log 2>&1 || tail -3 $out/bsweep_alone.log
python tools/bsweep.py --workload cfg3 --depth 0 --pack 1 --steps 100 --bs 256,512,1024,2048 --modes default,one_a_cu --out $out/${tag}_cfg3_alone_bsweep.json > $out/bsweep_alone3.log 2>&1 || tail -3 $out/bsweep_alone3.log
python tools/plugin_latency.py --out $out/${tag}_plugin_latency.json > $out/plugin.log 2>&1 || tail -3 $out/plugin.log
python tools/soak_round2.py 320 1 $out/${tag}_soak_round2.txt > $out/soak.log 2>&1 || tail -3 $out/soak.log
python tools/round2_class_gain.py $out/${tag}_round2_classes.txt > $out/round2_gain.log 2>&1 || tail -3 $out/round2_gain.log
echo "[artefacts] bench line, sweep, plug-in latency, soak, class gains done" ;;
*) echo "unknown PART $part"; exit 2 ;;
esac
