#!/usr/bin/env bash
# Diagnostic: texture-addresser / L1 (TA, TCP) counters of the observation kernel of one workload, four rocprofv3 --pmc passes.
#   tools/ta_counters.sh OUTDIR bench.py-args...        (on the GPU box)
set -uo pipefail
out=$1; shift
mkdir -p $out
python tools/pmc_pass.py $out/ta1.json "GRBM_GUI_ACTIVE TA_BUSY_avr TA_BUSY_max TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "$@" > /dev/null 2>$out/ta1.err || echo "pass 1 failed"
python tools/pmc_pass.py $out/ta2.json "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "$@" > /dev/null 2>$out/ta2.err || echo "pass 2 failed"
python tools/pmc_pass.py $out/ta3.json "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum" "$@" > /dev/null 2>$out/ta3.err || echo "pass 3 failed"
python tools/pmc_pass.py $out/ta4.json "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "$@" > /dev/null 2>$out/ta4.err || echo "pass 4 failed"
python - <<PY
import json, glob
acc = {}
for p in sorted(glob.glob("$out/ta?.json")):
    for k, v in json.load(open(p)).items():
        if "k_obs" in k or "k_step" in k: acc.setdefault(k, {}).update(v)
json.dump(acc, open("$out/ta_counters.json", "w"), indent=1, sort_keys=True)
print(json.dumps(acc, indent=1, sort_keys=True))
PY
