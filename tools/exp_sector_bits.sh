#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for b in 15 16 17 18; do
 for cap in 6500 14000; do
  echo "== bits $b cap $cap"
  timeout 600 python tools/exp_sector.py 12 5 --sector-only --opt=sector_bits=$b --opt=sector_tile_cap=$cap --opt=sector_profile=1 2>&1 | tail -2
 done
done
