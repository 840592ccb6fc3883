"""Where does the feeder's pinned memory land?  GPU NUMA node, this process's affinity, H2D rate of a pinned buffer allocated
under each node's CPUs, and the loader's own rate."""
import os, sys, time, glob, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rubiksnet_amd import input_pipeline as ip
dev = torch.device("cuda:0")
props = torch.cuda.get_device_properties(dev)
bdf = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
try: node = open("/sys/bus/pci/devices/%s/numa_node" % bdf).read().strip()
except OSError as e: node = "unreadable (%s)" % e
aff = os.sched_getaffinity(0)
print("gpu", bdf, "numa_node", node, "| affinity: %d cpus, min %d max %d" % (len(aff), min(aff), max(aff)))
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
print("nodes:", [os.path.basename(n) + ":" + open(n + "/cpulist").read().strip() for n in nodes])
print("_gpu_numa_cpus:", None if ip._gpu_numa_cpus(dev) is None else len(ip._gpu_numa_cpus(dev)))
def cpus_of(n):
    s = set()
    for part in open(n + "/cpulist").read().strip().split(","):
        lo, _, hi = part.partition("-"); s.update(range(int(lo), int(hi or lo) + 1))
    return s
d = torch.empty(32, 224, 224, 24, dtype=torch.uint8, device=dev)
def rate(h, n=20):
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); return n * h.numel() / (time.perf_counter() - t0) / 1e9
for n in nodes:
    allowed = aff & cpus_of(n)
    if not allowed: print(os.path.basename(n), "no allowed cpu"); continue
    os.sched_setaffinity(0, allowed)
    h = torch.empty(32, 224, 224, 24, dtype=torch.uint8).pin_memory(); h.random_(0, 256)
    r = rate(h); os.sched_setaffinity(0, aff)
    print(os.path.basename(n), "pinned there: H2D %.1f GB/s" % r)
it = iter(ip.SyntheticClipLoader(batch=32, device=dev))
for _ in range(3): next(it)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): next(it)
torch.cuda.synchronize(); print("loader: %.0f clips/s" % (20 * 32 / (time.perf_counter() - t0)))
