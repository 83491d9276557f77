"""GPU box: what distributed.pin_to_gpu_numa_node finds in the real sysfs and does to this process (round 6: GPU 0 -> NUMA node 0,
128 of 256 CPUs)."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from jssenv_amd import distributed as D
print("affinity before", len(os.sched_getaffinity(0)))
n = D.gpu_numa_node(0)
print("gpu 0 numa node", n, "cpus", len(D.numa_cpus(n)) if n is not None else None)
print(D.pin_to_gpu_numa_node(0, 0, 1))
print("affinity after", len(os.sched_getaffinity(0)))
print(open("/sys/class/kfd/kfd/topology/nodes/0/properties").read()[:300] if os.path.exists("/sys/class/kfd/kfd/topology/nodes/0/properties") else "no kfd topology")
print(sorted(os.listdir("/sys/class/kfd/kfd/topology/nodes"))[:20] if os.path.exists("/sys/class/kfd/kfd/topology/nodes") else "")
