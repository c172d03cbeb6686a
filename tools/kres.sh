#!/bin/bash
# tools/kres.sh [-DFLAG ...] : compile the kernels for gfx950 (device only) and print registers / spills / occupancy per kernel
# (hipcc -Rpass-analysis=kernel-resource-usage).  KRES_FILTER=regex limits the kernels shown; KRES_ASM=path keeps the assembly.
cd "$(dirname "$0")/../h264bsd_amd/csrc" || exit 1
out=${KRES_ASM:-/tmp/kres_$$.s}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -Wno-unused-value -Wno-unused-command-line-argument -S -Rpass-analysis=kernel-resource-usage "$@" engine.hip -o "$out" 2>&1 |
  python3 -c '
import sys,re,os
flt=os.environ.get("KRES_FILTER","")
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass",l) or re.search(r"remark: (.*?) \[-Rpass",l)
    if not m:
        if "error" in l or "warning" in l: sys.stderr.write(l)
        continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        cur={"name":t.split(":",1)[1].strip()};rows.append(cur)
    elif cur is not None and ":" in t:
        k,v=t.split(":",1);cur[k.strip()]=v.strip()
print("%-58s %5s %5s %6s %6s %7s %4s"%("kernel","VGPR","SGPR","vspill","sspill","scratch","occ"))
for r in rows:
    if flt and not re.search(flt,r["name"]): continue
    print("%-58s %5s %5s %6s %6s %7s %4s"%(r["name"][:58],r.get("VGPRs"),r.get("TotalSGPRs"),r.get("VGPRs Spill"),r.get("SGPRs Spill"),r.get("ScratchSize [bytes/lane]"),r.get("Occupancy [waves/SIMD]")))
'
[ -z "$KRES_ASM" ] && rm -f "$out"
exit 0
