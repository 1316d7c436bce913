#!/bin/bash
# Build the native library of another git revision (default HEAD) as slowfast_amd/libsfamd_prev.so, for in-step A/B runs of the
# working tree against it ON THE SAME BOX (boxes differ by +-1.5 %, more than most changes):
#   tools/build_prev_lib.sh [REV]
#   tools/gpu/ab.sh OUT --preset MVITv2_S_16x4 -- "new:X=1" "prev:SFAMD_LIBRARY=$PWD/slowfast_amd/libsfamd_prev.so,SF_ALLOW_STALE_LIBRARY=1"
set -e
REV=${1:-HEAD}; ROOT=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
mkdir -p $T/slowfast_amd/csrc $T/include
for f in $(git -C $ROOT ls-tree --name-only $REV slowfast_amd/csrc/); do git -C $ROOT show $REV:$f > $T/$f; done
git -C $ROOT show $REV:include/sfamd.h > $T/include/sfamd.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -shared -Wno-comment -Wno-pass-failed -I$T/include \
  "-DSF_BUILD_ID=\"$(git -C $ROOT rev-parse --short $REV)\"" $T/slowfast_amd/csrc/sf_api.hip -o $ROOT/slowfast_amd/libsfamd_prev.so
rm -rf $T; ls -la $ROOT/slowfast_amd/libsfamd_prev.so
