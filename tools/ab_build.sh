#!/bin/bash
# Build libirdm_hip.so of several branches side by side (authoring container, no GPU needed):
#   tools/ab_build.sh exp/leader-inline-tails exp/leader-cached-expiry ...
# -> iridium-sniffer_amd/build/ab/<branch with / replaced by _>/libirdm_hip.so  (build/ is git-ignored but travels to the
# GPU box).  Then, on the GPU: gpurun -- 'bash tools/ab_run.sh'  compares every library found there with the in-tree one.
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for br in "$@"; do
    name=${br//\//_}
    dst=$ROOT/iridium-sniffer_amd/build/ab/$name
    rm -rf "$dst" && mkdir -p "$dst/src"
    git -C "$ROOT" archive "$br" iridium-sniffer_amd include | tar -x -C "$dst/src"
    make -s -C "$dst/src/iridium-sniffer_amd" -j8 libirdm_hip.so
    cp "$dst/src/iridium-sniffer_amd/libirdm_hip.so" "$dst/libirdm_hip.so"
    rm -rf "$dst/src"
    echo "built $dst/libirdm_hip.so from $br ($(git -C "$ROOT" rev-parse --short "$br"))"
done
