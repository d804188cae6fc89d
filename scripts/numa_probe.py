"""Where does the box put things?  NUMA node of the GPU, of hipHostMalloc'ed memory (first-touched by this thread, by a thread bound
to the other node), of a big numpy array, and the cpus of each node."""
import ctypes as C, glob, os, re, sys, threading
sys.path.insert(0, '.')
import numpy as np
import torch
from pyruhvro_amd import cabi
cabi.lib()
hip = C.CDLL("libamdhip64.so.7")
for p in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")) + sorted(glob.glob("/sys/bus/pci/devices/*/numa_node"))[:0]:
    print(p, open(p).read().strip(), os.path.basename(os.path.realpath(os.path.dirname(p))))
try:
    bus = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
    print("gpu0 pci bus", bus, hex(bus) if isinstance(bus, int) else "")
    buf = C.create_string_buffer(64); hip.hipDeviceGetPCIBusId(buf, 64, 0); print("hipDeviceGetPCIBusId(0):", buf.value.decode())
except Exception as e:
    print("pci:", e)
for n in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
    print(n, open(n).read().strip())
print("affinity size", len(os.sched_getaffinity(0)))

def where(addr, nbytes):
    pat = re.compile(r"^([0-9a-f]+) .*")
    out = {}
    for line in open("/proc/self/numa_maps"):
        a = int(line.split()[0], 16)
        if addr <= a < addr + nbytes or a <= addr < a + (1 << 34):
            m = dict(kv.split("=") for kv in line.split()[1:] if "=" in kv and kv.startswith("N"))
            if a == addr or (a <= addr and m):
                out[hex(a)] = {k: int(v) for k, v in m.items()}
    return out

def alloc_touch(tag):
    p = C.c_void_p()
    n = 256 << 20
    assert hip.hipHostMalloc(C.byref(p), C.c_size_t(n), C.c_uint(0)) == 0
    C.memset(p, 1, n)
    print(tag, "hipHostMalloc 256 MB ->", where(p.value, n))
    return p

alloc_touch("main thread:")
nodes = sorted(glob.glob("/sys/devices/system/node/node*/cpulist"))
def parse(s):
    out = []
    for part in s.strip().split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out
for i, n in enumerate(nodes):
    cpus = parse(open(n).read())
    def run():
        os.sched_setaffinity(0, cpus)
        alloc_touch(f"thread bound to node {i}:")
    t = threading.Thread(target=run); t.start(); t.join()
a = np.ones(256 << 20, dtype=np.uint8)
print("numpy 256 MB ->", where(a.ctypes.data, a.nbytes))
