#!/bin/bash
# Where does a step kernel spill?  Compiles ONE unit of the split build (device code only) with line tables and attributes every
# scratch_store / scratch_load of a kernel to the source line it came from; also prints the kernel's resource remark.
#   scripts/spill_where.sh <tag> [extra hipcc flags]      env: NS=16 AV=1 KIND=0 (0 = k_step unit, 1 = front unit), KFN=<mangled kernel name>
# Runs in the build container (no GPU needed).  Output under /tmp/agarcl_spill/.
ROOT=$(cd "$(dirname "$0")/.." && pwd); TAG=$1; shift
NS=${NS:-16}; AV=${AV:-1}; KIND=${KIND:-0}; D=/tmp/agarcl_spill; mkdir -p $D
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math"
[ "$KIND" = "0" ] && FLAGS="$FLAGS -mllvm -disable-machine-licm"
hipcc $FLAGS -gline-tables-only -Rpass-analysis=kernel-resource-usage -DAG_PART_NS=$NS -DAG_PART_AV=$AV -DAG_PART_KIND=$KIND --cuda-device-only "$@" \
  -c $ROOT/agarcl_amd/csrc/agar_engine.hip -o $D/$TAG.o 2> $D/$TAG.remarks
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --unbundle --input=$D/$TAG.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$D/$TAG.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d -l --no-show-raw-insn $D/$TAG.co > $D/$TAG.s 2>/dev/null
python3 - <<PY
import re, collections
k = "${KFN:-_Z6k_stepILi${NS}ELb${AV}ELi0EEvPK7AgStatePKfPKiiiiiiiS6_}"
t = open("$D/$TAG.remarks").read()
for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
    if b.split()[0] == k:
        g = lambda x: re.search(x + r": (\d+)", b).group(1)
        print(k[:48], "VGPR", g("VGPRs"), "SGPR", g("SGPRs"), "scratch", g(r"ScratchSize \[bytes/lane\]"), "spilled VGPRs", g("VGPRs Spill"), "SGPRs", g("SGPRs Spill"), "waves/SIMD", g(r"Occupancy \[waves/SIMD\]"))
fn = cur = None; st = collections.Counter(); ld = collections.Counter(); n = 0
for l in open("$D/$TAG.s"):
    m = re.match(r"^[0-9a-f]+ <(.*)>:$", l)
    if m: fn = m.group(1); continue
    if fn != k: continue
    m = re.match(r"^; (/.*):(\d+)$", l)
    if m: cur = (m.group(1).split("/")[-1], int(m.group(2))); continue
    m2 = re.search(r"scratch_(store|load)_dword(x(\d))?", l)
    if m2:
        (st if m2.group(1) == "store" else ld)[cur] += int(m2.group(3) or 1)
    n += 1
print(n, "lines of", k[:40], "; spilled dwords stored", sum(st.values()), "loaded", sum(ld.values()))
print("stores by line:", " ".join("%s:%d=%d" % (f.replace("agar_", "")[:-4], ln, c) for (f, ln), c in sorted(st.items())))
print("loads by line: ", " ".join("%s:%d=%d" % (f.replace("agar_", "")[:-4], ln, c) for (f, ln), c in sorted(ld.items())))
PY
