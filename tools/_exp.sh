cd /root/repo
timeout 600 python -m pytest tests/test_gpu_fused.py tests/test_gpu_resume.py -x -q 2>&1 | tail -2
LEAN="--no-cpu-baseline --no-other-workloads --no-boundary-path"
for v in new two_forks pruned_event adam_main; do
  case $v in new) envs="A=1";; two_forks) envs="NSR_WGRAD_TWO_FORKS=1";; pruned_event) envs="NSR_PRUNED_EVENT=1";; adam_main) envs="NSR_ADAM_ON_MAIN=1";; esac
  env $envs timeout 600 python bench.py --steps 200 --warmup 20 $LEAN 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), round(d['steady_state']['ms_per_step'],4), 'host', round(d['host_enqueue_ms_per_step'],3), {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k.startswith('hashgrid') or k.startswith('mlp_backward')}, d['final_loss'])"
done
