#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes)
into profiles/pmc_traffic.json, which bench.py reads for roofline.traffic.

usage: parse_pmc.py <fetch_dir> <write_dir> <workload> <envs_per_gpu> <tag>

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE (KiB) reports half the bytes of
wide coalesced reads, WRITE_SIZE (KiB) is exact for 16-B stores.  Our reads are mostly narrow gathers, for which
the factor 2 is an upper bound (uncalibrated); both raw counters are kept in the JSON.
"""
import collections
import csv
import glob
import json
import os
import sys

NAMES = {"k_obs<0": "k_obs<cutils>", "k_obs<1": "k_obs<tree>", "k_obs<2": "k_obs<cutils+tree>", "k_obs<3": "k_obs<cutils+tree>", "k_obs<4": "k_obs<cutils+tree>", "k_step<": "k_step<synth>"}


def agg(d, counter):
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"] != counter:
                continue
            for k, v in NAMES.items():
                if k in row["Kernel_Name"]:
                    acc[v].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


if __name__ == "__main__":
    fdir, wdir, workload, envs, tag = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    f, nf = agg(fdir, "FETCH_SIZE")
    w, nw = agg(wdir, "WRITE_SIZE")
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    out.setdefault(workload, {})
    for k in f:
        out[workload][k] = dict(envs=envs, tag=tag, launches=min(nf[k], nw.get(k, 0)), fetch_size_kib=f[k],
                                write_size_kib=w.get(k, 0.0),
                                hbm_bytes_per_launch=(2 * f[k] + w.get(k, 0.0)) * 1024)
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out[workload], indent=1))
