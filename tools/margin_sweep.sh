#!/bin/bash
# Any-hit mismatches (modes 1 and 3 against mode 0) and kernel times on the scenes' own shadow rays, for every
# library variant given (paths relative to the package directory).
cd "$(dirname "$0")/.."
for lib in "$@"; do
  echo "== $lib"
  PTAMD_LIB=$PWD/cuda-path-tracer_amd/$lib timeout -k 10 280 python -u tools/mode_compare.py teapot cornell helmet party_armadillo pbr_metallic --sets shadow --iters 4 2>&1 \
    | grep -v amdgpu.ids | python -c "
import sys, json
for line in sys.stdin:
    if '{' not in line: print(line.rstrip()); continue
    head, js = line.split('{', 1)
    d = json.loads('{' + js)
    print(head, d['rays'], 'mismatch m1', d['any_mismatch_mode1'], 'm3', d['any_mismatch_mode3'], 'missed3', d['any_mode3_missed_occluder'], 'ms m0 %.3f m1 %.3f m2 %.3f m3 %.3f' % (d['any_ms_mode0'], d['any_ms_mode1'], d['any_ms_mode2'], d['any_ms_mode3']))
" || exit 1
done
