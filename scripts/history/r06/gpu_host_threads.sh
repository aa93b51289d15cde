cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r06u
show() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']; print(sys.argv[1], '| q/s', d['value'], '| ms/step', d['ms_per_step'], '| kernel', r['avg_launch_ms'], '| cpus', c.get('host_cpus_busy'), '| p50', c.get('p50_ms'), c.get('latency_ms'))" "$1" 2>/dev/null || echo "$1 FAILED"; }
for rep in 1 2; do for t in 2 3 4; do
python bench.py --steps 100 --warmup 10 --host-threads $t --no-cpu-baseline --closed-loop '' --exhaustive-steps 0 --c4-steps 0 --c2-steps 0 --c5-steps 0 2>/dev/null | tee gpurun_out/r06u/thr${t}_$rep.json | show "threads=$t rep $rep"
done; done | tee gpurun_out/r06u/r06u_host_threads.log
