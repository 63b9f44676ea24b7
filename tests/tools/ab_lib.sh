#!/bin/bash
# A/B of library builds: default build, then each yak_amd/alt/*.so swapped in
cd $GRAFT_REPO_ROOT
bash tests/tools/ab.sh "$@"
cp yak_amd/libyak_amd.so /tmp/keep.so
for l in yak_amd/alt/*.so; do
  cp $l yak_amd/libyak_amd.so; echo "== $l"
  YAKAMD_VERBOSE=2 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-qv --no-pcie 2>&1 | grep -E "drounds|equals_reference" | tail -6 | sed 's/.*replay2//' | cut -c1-160
done
cp /tmp/keep.so yak_amd/libyak_amd.so
