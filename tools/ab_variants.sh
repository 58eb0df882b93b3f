#!/bin/bash
# Dev tool (GPU box): correctness (tests/test_gpu_msm.py) + headline bench for each built variant library.
for k in "$@"; do
  export GM_LIB_PATH=$PWD/tools/_build/var${k%%:*}/gemini_amd/libgemini_hip.so
  [[ $k == *:* ]] && export ${k#*:}
  echo "=== variant $k: $GM_LIB_PATH"
  timeout 600 python -m pytest tests/test_gpu_msm.py -x -q -m gpu 2>&1 | tail -3
  timeout 300 python bench.py --headline-only 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
  [[ $k == *:* ]] && v=${k#*:} && unset ${v%%=*}
done
unset GM_LIB_PATH
