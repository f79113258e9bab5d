cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_s3; mkdir -p $O
F="--no-cpu-baseline --no-measure-traffic --no-scene --no-other-configs --no-unmerged --min-time 1.0"
for rep in 1 2; do
for cfg in "512 8" "1024 8"; do set -- $cfg
for ss in "" "--single-stream"; do
python bench.py --map-size $1 --cascades $2 --steps 500 --warmup 100 $F $ss 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1 x $2 $ss', d['ms_per_step'], d['value'], r['frac'], r.get('concurrent_launches'), r['kernel'])"
done; done; done > $O/bench_ab_chains.txt 2>&1
cat $O/bench_ab_chains.txt
