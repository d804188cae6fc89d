// The engine's process-wide pools (engine_internal.h: Pool, CtrlPool) -- never destroyed: worker threads and Arrow release
// callbacks may outlive static destruction order.
#include "engine_internal.h"

#include <sched.h>
#include <unistd.h>

#include <cctype>
#include <map>

#include <cmath>
#include <fstream>

namespace rhe {

Pool& dev_pool() { static Pool* p = new Pool(false); return *p; }
Pool& pin_pool() { static Pool* p = new Pool(true); return *p; }
CtrlPool& ctrl_pool() { static CtrlPool* p = new CtrlPool(); return *p; }

std::atomic<uint64_t> g_counters[RH_CTR_COUNT];

char* dup_msg(const std::string& s) {
  char* p = (char*)std::malloc(s.size() + 1);
  if (p) std::memcpy(p, s.c_str(), s.size() + 1);
  return p;
}


namespace {
struct Refill {            // the background allocator of pinned result blocks (one thread, started on first use, never joined)
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::pair<uint64_t, int>> queue;
  uint64_t queued_bytes = 0;
  bool started = false;
};
Refill& refill() { static Refill* r = new Refill; return *r; }
}  // namespace

void Pool::prefetch(uint64_t size, int device, uint64_t budget_left) {
  if (!host_) return;
  size = align_up(std::max<uint64_t>(size, 1), 1 << 16);
  {
    const int key = pin_key(device);
    std::lock_guard<std::mutex> g(mu_);
    for (const Block& b : free_)
      if (b.device == key && b.size >= size && b.size <= size * 2 + (1 << 20)) return;       // one is there
  }
  Refill& r = refill();
  std::lock_guard<std::mutex> g(r.mu);
  if (r.queued_bytes + size > budget_left || r.queue.size() >= 64) return;
  r.queue.emplace_back(size, device);
  r.queued_bytes += size;
  if (!r.started) {
    r.started = true;
    std::thread([this] {
      Refill& q = refill();
      for (;;) {
        std::pair<uint64_t, int> job;
        {
          std::unique_lock<std::mutex> l(q.mu);
          q.cv.wait(l, [&] { return !q.queue.empty(); });
          job = q.queue.front();
          q.queue.erase(q.queue.begin());
        }
        Block b;
        b.size = job.first;
        b.device = job.second;
        (void)hipSetDevice(job.second);                  // (placement follows the current device)
        b.device = pin_key(job.second);
        if (hipHostMalloc(&b.p, b.size, kPinnedFlags) == hipSuccess) put(b);
        else (void)hipGetLastError();
        std::lock_guard<std::mutex> l(q.mu);
        q.queued_bytes -= job.first;
      }
    }).detach();
  }
  r.cv.notify_one();
}

// cpus of every NUMA node (from /sys/devices/system/node/node<N>/cpulist); empty when the kernel shows none
const std::vector<std::vector<int>>& numa_node_cpus() {
  static const std::vector<std::vector<int>> cached = [] {
  std::vector<std::vector<int>> nodes;
  for (int nd = 0; nd < 64; nd++) {
    FILE* f = std::fopen(("/sys/devices/system/node/node" + std::to_string(nd) + "/cpulist").c_str(), "r");
    if (!f) break;
    char buf[4096];
    std::vector<int> cpus;
    if (std::fgets(buf, sizeof buf, f)) {
      for (char* p = buf; *p;) {
        char* e = nullptr;
        const long a = std::strtol(p, &e, 10);
        if (e == p) break;
        long z = a;
        p = e;
        if (*p == '-') { z = std::strtol(p + 1, &e, 10); p = e; }
        for (long c = a; c <= z; c++) cpus.push_back((int)c);
        if (*p == ',') p++;
      }
    }
    std::fclose(f);
    nodes.push_back(cpus);
  }
  return nodes;
  }();
  return cached;
}


// NUMA node of a HIP device (from its PCI address: /sys/bus/pci/devices/<id>/numa_node), -1 when the kernel does not say.
// hipHostMalloc places pinned memory on the node of the CURRENT device whatever thread calls it (scripts/numa_probe.py on the
// 8-GPU box: a thread bound to node 0 still gets node-1 pages for a GPU on node 1), so pinned blocks are pooled per node and
// allocated under the device they are for: a multi-GPU call's staging and result blocks then sit next to the GPU that reads /
// writes them.
int device_numa_node(int device) {
  static std::mutex mu;
  static std::map<int, int> cache;
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find(device);
  if (it != cache.end()) return it->second;
  int node = -1;
  char bus[64] = {0};
  if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) == hipSuccess) {
    std::string id(bus);
    for (char& ch : id) ch = (char)std::tolower((unsigned char)ch);
    std::ifstream f("/sys/bus/pci/devices/" + id + "/numa_node");
    if (!(f >> node)) node = -1;
  } else {
    (void)hipGetLastError();
  }
  cache[device] = node;
  return node;
}

// Binds the calling thread to the cpus of `node` (no-op when there is one node, when the node is unknown, or when this thread
// is bound there already).  RUHVRO_HIP_NUMA_BIND=0 turns it off.
void bind_thread_to_node(int node) {
  static const bool on = env_long_early("RUHVRO_HIP_NUMA_BIND", 1, 0, 1) != 0;
  thread_local int tried = -2;           // the node this thread last asked for (bound there, or found no allowed cpu on it)
  const std::vector<std::vector<int>>& nodes = numa_node_cpus();
  if (!on || node < 0 || nodes.size() < 2 || (size_t)node >= nodes.size() || tried == node) return;
  tried = node;
  // only within the cpus the PROCESS was given (taskset / a launcher's binding): the mask of its main thread, read once -- a
  // thread may set its affinity to cpus outside the mask it inherited, and the engine must not undo the user's restriction
  static cpu_set_t allowed;
  static std::once_flag once;
  std::call_once(once, [] {
    CPU_ZERO(&allowed);
    if (sched_getaffinity(getpid(), sizeof allowed, &allowed) != 0)
      for (int c = 0; c < CPU_SETSIZE; c++) CPU_SET(c, &allowed);
  });
  cpu_set_t set;
  CPU_ZERO(&set);
  int cnt = 0;
  for (int c : nodes[(size_t)node])
    if (c < CPU_SETSIZE && CPU_ISSET(c, &allowed)) { CPU_SET(c, &set); cnt++; }
  if (cnt > 0) (void)sched_setaffinity(0, sizeof set, &set);      // (no allowed cpu on that node: the thread stays where it may run)
}

unsigned effective_cpus() {
  static const unsigned v = [] {
    if (const char* e = std::getenv("RUHVRO_HIP_CPUS")) {
      const long n = std::atol(e);
      if (n >= 1 && n <= 4096) return (unsigned)n;
    }
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) hw = std::min<unsigned>(hw, (unsigned)CPU_COUNT(&set));
    auto cut = [&](double quota, double period) {
      if (quota > 0 && period > 0) hw = std::min<unsigned>(hw, std::max(1u, (unsigned)std::ceil(quota / period)));
    };
    {
      std::ifstream f("/sys/fs/cgroup/cpu.max");                 // cgroup v2: "<quota|max> <period>"
      std::string q;
      double period = 0;
      if (f >> q >> period && q != "max") cut(std::atof(q.c_str()), period);
    }
    {
      std::ifstream fq("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), fp("/sys/fs/cgroup/cpu/cpu.cfs_period_us");     // cgroup v1
      double q = 0, per = 0;
      if (fq >> q && fp >> per) cut(q, per);
    }
    return hw;
  }();
  return v;
}

}  // namespace rhe
