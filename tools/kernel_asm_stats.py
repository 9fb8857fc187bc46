#!/usr/bin/env python3
"""Static instruction mix of the kernels of one translation unit (device-only compile with the build's flags, disassembled):
total / VALU / SALU / LDS / global instructions, v_readlane + v_writelane (= scalar-register spill traffic), registers and
spills as the compiler reports them.  No GPU needed.
    tools/kernel_asm_stats.py fl_obs_m3 [fl_obs_m4 ...]      (EXTRA_HIPCC_FLAGS is honoured)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "--cuda-device-only", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-disable-machine-licm",
         "-mllvm", "-disable-lsr", "-mllvm", "-amdgpu-atomic-optimizer-strategy=None", "-Wno-unused-result"]


def stats(unit, outdir="/tmp/asm"):
    os.makedirs(outdir, exist_ok=True)
    obj, elf = os.path.join(outdir, unit + ".o"), os.path.join(outdir, unit + ".elf")
    extra = os.environ.get("EXTRA_HIPCC_FLAGS", "").split()
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", "-c",
                        os.path.join(ROOT, "flatland_marl_amd", "csrc", unit + ".hip"), "-o", obj], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    res, cur = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^ ]* *(Function Name|TotalSGPRs|VGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]): (\S+)", line)
        if m:
            if m.group(1) == "Function Name":
                cur = res.setdefault(m.group(2), {})
            elif cur is not None:
                cur[m.group(1)] = m.group(2)
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + obj,
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + elf])
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", elf], capture_output=True, text=True).stdout
    open(os.path.join(outdir, unit + ".s"), "w").write(txt)
    out = []
    for f in re.split(r"\n(?=[0-9a-f]{16} <)", txt):
        m = re.match(r"[0-9a-f]{16} <([^>]+)>", f)
        if not m:
            continue
        ins = [ln.split()[0] for ln in f.splitlines()[1:] if re.match(r"\s+[a-z][a-z_0-9]+\s", ln)]
        c = collections.Counter(ins)
        grp = lambda p: sum(v for k, v in c.items() if k.startswith(p))  # noqa: E731
        r_ = res.get(m.group(1), {})
        out.append(dict(kernel=m.group(1), total=len(ins), valu=grp("v_"), salu=grp("s_"), lds=grp("ds_"), vmem=grp("global_") + grp("flat_") + grp("buffer_"),
                        readlane=c["v_readlane_b32"], writelane=c["v_writelane_b32"], vgpr=r_.get("VGPRs"), sgpr_spill=r_.get("SGPRs Spill"),
                        vgpr_spill=r_.get("VGPRs Spill"), scratch=r_.get("ScratchSize [bytes/lane]")))
    return out


if __name__ == "__main__":
    for unit in sys.argv[1:] or ["fl_obs_m3"]:
        for row in stats(unit):
            print("%-48s total %6d  valu %6d  salu %6d  lds %5d  vmem %4d  readlane %4d  writelane %4d  vgpr %s  sgpr-spill %s  vgpr-spill %s  scratch %s"
                  % (row["kernel"][:48], row["total"], row["valu"], row["salu"], row["lds"], row["vmem"], row["readlane"], row["writelane"],
                     row["vgpr"], row["sgpr_spill"], row["vgpr_spill"], row["scratch"]))
