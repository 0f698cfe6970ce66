#!/bin/bash
# Experiment builds of the octet kernel with one piece removed each (timing only: results are wrong by construction).
#   bash tools/oct_ablate.sh <N> <waves> -> build/var/abl_<name>.so
set -eu
N=${1:-3}; W=${2:-2}
cd "$(dirname "$0")/.."
SRC=cooperative-search_amd/csrc
mkdir -p build/var
python3 - "$SRC/coopsearch.hip" "$SRC/coop_abl.hip" <<'PY'
import sys
s = open(sys.argv[1]).read()
def rep(a, b):
    global s
    assert a in s, a[:60]
    s = s.replace(a, b)
rep("            if (__ballot(inr)) {   // wave-uniform",
    "#ifdef ABL_NOSLOW\n            if (false) {\n#else\n            if (__ballot(inr)) {\n#endif")
rep("        if (__builtin_expect(need != 0ull, 0)) {   // cold: about one wave-step in 24",
    "#ifdef ABL_NORESET\n        if (false) {\n#else\n        if (__builtin_expect(need != 0ull, 0)) {\n#endif")
rep("        const int reward = oct_detect<N>(p, sh, o, t, sh8, stepping, e, tape);",
    "#ifdef ABL_NODETECT\n        const int reward = -1;\n#else\n        const int reward = oct_detect<N>(p, sh, o, t, sh8, stepping, e, tape);\n#endif")
rep("    double s1, c1, s2, c2;\n    trig_heading(T, yw, s1, c1);\n    trig_heading(T, yr, s2, c2);\n    OctKin k{",
    "    double s1, c1, s2, c2;\n#ifdef ABL_NOTRIG\n    s1 = yw * 0.1; c1 = yw * 0.2; s2 = yr * 0.1; c2 = yr * 0.2;\n#else\n    trig_heading(T, yw, s1, c1);\n    trig_heading(T, yr, s2, c2);\n#endif\n    OctKin k{")
rep("        const unsigned out = oct_kinematics<N>(p, T, sh, o, t, sh8, stepping, act, e);",
    "#ifdef ABL_NOKIN\n        const unsigned out = 0; e.x += 0.001 * act;\n#else\n        const unsigned out = oct_kinematics<N>(p, T, sh, o, t, sh8, stepping, act, e);\n#endif")
open(sys.argv[2], "w").write(s)
PY
for v in BASE NOSLOW NORESET NODETECT NOTRIG NOKIN "NOKIN -DABL_NODETECT -DABL_NORESET"; do
  name=$(echo "$v" | tr -d ' ' | sed 's/-DABL_/_/g')
  (cd $SRC && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DCS_ONLY_N=$N -DCS_OCT_WAVES=$W \
     -DABL_$v -I ../../include coop_abl.hip policy.hip episodes.hip -o ../../build/var/abl${N}_$name.so 2>&1 | grep -v "^$" | grep -iv warning | tail -2) &
done
wait
rm -f $SRC/coop_abl.hip
ls build/var/abl${N}_*.so
