#!/bin/bash
# registers / scratch / occupancy of every kernel of one source, as the compiler reports them
# usage: tools/kres.sh advection.hip [-DPYRO_FAST=1 -ffp-contract=fast ...]   (developer tool)
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -O3 "$@" -Rpass-analysis=kernel-resource-usage \
  -c pyro2_amd/csrc/$src -o /tmp/kres_$$.o 2>&1 | python3 -c '
import sys, re, subprocess
cur = {}
def flush():
    if cur:
        n = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
        n = re.sub(r"\(.*", "", n).replace("void pyro::", "")
        print("%-60s vgpr %3s agpr %3s sgpr %3s scratch %4s occ %s" % (n[:60], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]")))
for line in sys.stdin:
    m = re.search(r"remark: [^ ]+ +Function Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m and "Name:" in line:
        flush(); cur = {"name": m.group(1)}; continue
    m = re.search(r":\s+([A-Za-z][A-Za-z /\[\]]+): (\d+)", line)
    if m: cur[m.group(1).strip()] = m.group(2)
flush()
'
rm -f /tmp/kres_$$.o
