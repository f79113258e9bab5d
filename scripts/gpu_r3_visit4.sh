#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3v4
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_interop.py -m gpu -q > gpurun_out/r3v4/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3v4/pytest.log
tail -5 gpurun_out/r3v4/pytest.log
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v" >> gpurun_out/r3v4/kernarg.txt
  HIP_FORCE_DEV_KERNARG=$v timeout 300 python scripts/ab_merged.py 256:1 256:4 512:4 1024:1 1024:4 >> gpurun_out/r3v4/kernarg.txt 2>&1
done
cat gpurun_out/r3v4/kernarg.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/r3v4/bench.json 2> gpurun_out/r3v4/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r3v4/bench.json') if l.startswith('{')][0])
r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['unmerged'])
PY
