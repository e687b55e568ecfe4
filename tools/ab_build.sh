#!/bin/bash
# Builds the library of an EARLIER commit as an A/B baseline (run in the build container: the GPU box has no .git):
#   tools/ab_build.sh <commit> <name>   ->  fixed-wing-gym_amd/gym_fixed_wing/_abl/libfwgym_<name>.so
# (csrc/ + include/ + Makefile of that commit in a scratch directory, the commit's own Makefile line; the committed
# csrc/generated/specs.inc of that commit is used as it is).  _abl/ is git-ignored but travels to the GPU box.
set -euo pipefail
commit=${1:?commit}; name=${2:?name}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
trap 'rm -rf "$tmp"' EXIT
git -C "$root" archive "$commit" fixed-wing-gym_amd/csrc fixed-wing-gym_amd/Makefile include | tar -x -C "$tmp"
mkdir -p "$tmp/fixed-wing-gym_amd/gym_fixed_wing" "$root/fixed-wing-gym_amd/gym_fixed_wing/_abl"
make -C "$tmp/fixed-wing-gym_amd" all
out="$root/fixed-wing-gym_amd/gym_fixed_wing/_abl/libfwgym_${name}.so"
cp "$tmp/fixed-wing-gym_amd/gym_fixed_wing/libfwgym.so" "$out"
echo "built $out from $(git -C "$root" rev-parse --short "$commit")"
