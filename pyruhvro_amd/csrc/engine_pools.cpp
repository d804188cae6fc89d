// The engine's process-wide pools (engine_internal.h: Pool, CtrlPool) -- never destroyed: worker threads and Arrow release
// callbacks may outlive static destruction order.
#include "engine_internal.h"

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


}  // namespace rhe
