#!/bin/bash
# tools/_abl_mkvariant.sh <name> <git ref>: a copy of the tree at <ref> under tools/_abl/<name> (git-ignored), objects seeded from the main
# build so that only what a patch touches is recompiled.  Apply patches to the copy, then: tools/_abl_mkvariant.sh --build <name>
set -e
R=/root/repo
if [ "$1" == "--build" ]; then
  cd $R/tools/_abl/$2
  python - <<'P'
import __graft_entry__ as g, os, subprocess
g_emu = os.path.join(g.ROOT, "tests", "_emu", "libbcp_emu.so")
# (skip the simulator build: the variant only runs on the GPU)
os.makedirs(os.path.dirname(g_emu), exist_ok=True)
open(g_emu, "ab").close(); os.utime(g_emu, (2e9, 2e9))
try:
    print(g.build())
except Exception as e:
    print("build:", e)
P
  exit 0
fi
name=$1; ref=$2
rm -rf $R/tools/_abl/$name; mkdir -p $R/tools/_abl/$name
git -C $R archive $ref bcp_amd tests oracle include __graft_entry__.py tools/emu | tar -x -C $R/tools/_abl/$name
mkdir -p $R/tools/_abl/$name/tools/probe $R/tools/_abl/$name/bcp_amd/csrc/build
cp $R/tools/probe/replay_stress.py $R/tools/_abl/$name/tools/probe/
