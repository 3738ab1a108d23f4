#!/bin/bash
# Copies a run directory of tools/profile_set.sh (gpurun_out/<tag>/) into profiles/<tag>_* : the set-x traces the shell leaves in
# shapes.txt / split_ab.txt are dropped, *.err files are not copied.     usage: bash tools/install_profile_set.sh r06zz
TAG=${1:-r06zz}
cd "$(dirname "$0")/.."
rm -f profiles/${TAG}_*
( cd gpurun_out/$TAG && for f in *; do case $f in
    *.err) ;;
    shapes.txt) { head -1 $f; grep "clips/s" $f | grep -v "^+"; } > ../../profiles/${TAG}_shapes.txt ;;
    split_ab.txt) ;;
    *) cp $f ../../profiles/${TAG}_$f ;;
  esac; done )
python - "$TAG" <<'PY'
import re, sys
tag = sys.argv[1]
raw = open(f"gpurun_out/{tag}/split_ab.txt").read().split("\n")
out, label = [raw[0]], None
for l in raw[1:]:
    m = re.match(r"^(avenue B=\d+ split=\d+): ", l)
    if m:
        label = m.group(1)
    elif label and re.match(r"^[\d.]+ clips/s", l):
        out.append(f"{label}: {l}"); label = None
open(f"profiles/{tag}_split_ab.txt", "w").write("\n".join(out) + "\n")
PY
sed -i '/amdgpu.ids/d' profiles/${TAG}_e2e_shards.txt
ls profiles/${TAG}_* | wc -l
