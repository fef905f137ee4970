import os, sys, json, time
sys.path[:0]=[os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "instant-nsr-pl_amd")]
import torch, bench
out={}
for t in (16,32,64,128):
    os.environ["NSR_CPU_BASELINE_THREADS"]=str(t)
    r=bench.cpu_baseline(seconds_budget=4.0)
    out[t]={"reference_formulation": r["value"], "hashgrid_port": r["hashgrid_port"]["value"]}
print(json.dumps({"host_logical_cpus": os.cpu_count(), "samples_per_s_by_threads": out}))
