// The engine's process-wide pools (engine_internal.h: Pool, CtrlPool) -- never destroyed: worker threads and Arrow release
// callbacks may outlive static destruction order.
#include "engine_internal.h"

#include <sched.h>

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
    std::lock_guard<std::mutex> g(mu_);
    for (const Block& b : free_)
      if (b.size >= size && b.size <= size * 2 + (1 << 20)) return;       // one is there
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
        if (hipHostMalloc(&b.p, b.size, kPinnedFlags) == hipSuccess) put(b);
        else (void)hipGetLastError();
        std::lock_guard<std::mutex> l(q.mu);
        q.queued_bytes -= job.first;
      }
    }).detach();
  }
  r.cv.notify_one();
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
