cd $GRAFT_REPO_ROOT
O=gpurun_out
run() { name=$1; shift; python bench.py "$@" 2>$O/r06_bench_$name.err | tail -1 > $O/r06_bench_$name.json; }
run tracking_n60 --horizon 60 --no-others --no-cpu-baseline --steps 10
run tracking_n80 --horizon 80 --no-others --no-cpu-baseline --steps 10
bash scratch/r6/prof.sh n60 --horizon 60 --steps 10 > /dev/null 2>&1
bash scratch/r6/prof.sh n80 --horizon 80 --steps 10 > /dev/null 2>&1
for n in 60 80; do python -c "
import json; d=json.loads(open('gpurun_out/r06_bench_tracking_n$n.json').read()); r=d['roofline']; print($n, d['value'], d['kernels_ms'], r.get('traffic'), r.get('traffic_over_algorithmic'), r.get('issue'))"; done
