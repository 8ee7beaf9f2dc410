#!/bin/bash
# round 6, experiment (c): the fused kernel under other workgroup orders / ray tilings — time, FETCH_SIZE, L2 hit rate (profiles/round6_fused_closing.md)
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for o in 0 1 2; do
  CAR_DEV_FLAGS="-DCAR_WG_ORDER=$o -DCAR_ABLATION_NONE" bash tools/pmc_fetch_fused.sh order$o 0
done
CAR_BENCH_TILE=2x4 CAR_DEV_FLAGS="-DCAR_WG_ORDER=0 -DCAR_ABLATION_NONE" bash tools/pmc_fetch_fused.sh order0_tile2x4 0
CAR_BENCH_TILE=2x4 CAR_DEV_FLAGS="-DCAR_WG_ORDER=2 -DCAR_ABLATION_NONE" bash tools/pmc_fetch_fused.sh order2_tile2x4 0
