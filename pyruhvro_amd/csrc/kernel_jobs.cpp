// Kernel cache + background compile jobs (see kernel_jobs.h).
#include "kernel_jobs.h"

#include <fcntl.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <mutex>
#include <set>
#include <thread>

#include <dirent.h>

#include "rtc_compile.h"

extern char** environ;

namespace rh {

namespace {

struct Slot {
  ImageState state = IMG_UNKNOWN;
  std::shared_ptr<const std::vector<char>> code;
  std::string why;
  bool from_cache = false;
  double seconds = 0;
};

// Process-wide bookkeeping of running jobs.  Leaked on purpose: job threads outlive static destruction at exit.
struct Registry {
  std::mutex mu;
  std::condition_variable cv;
  std::set<pid_t> children;      // helper processes in flight
  unsigned running = 0;          // jobs holding a slot
  unsigned in_process = 0;       // of which compile inside this process (hiprtc on a job thread)
  bool exiting = false;
  bool hooked = false;
  std::string priv_dir;          // work_dir()'s private directory of this process (read-only kernel cache), removed at exit
  std::set<std::string> scratch; // <object>.<pid>.<n>.src / .err files of helpers in flight: removed at exit when their helper is killed
};
Registry& registry() {
  static Registry* r = new Registry;
  return *r;
}

// (the engine's effective_cpus() lives in engine_pools.cpp; this file is also part of nothing else, so it asks the C ABI)
extern "C" uint32_t rh_effective_cpus(void);
unsigned effective_cpus_for_jobs() { return rh_effective_cpus(); }

unsigned max_jobs() {
  static const unsigned v = [] {
    if (const char* e = std::getenv("RUHVRO_HIP_COMPILE_JOBS")) {
      const long n = std::atol(e);
      if (n >= 1 && n <= 64) return (unsigned)n;
    }
    const unsigned hw = effective_cpus_for_jobs();
    return std::max(1u, std::min(8u, hw ? hw / 2 : 2u));
  }();
  return v;
}

// At exit: helper processes are killed (their output is only ever renamed into place when complete); a compile that runs
// inside this process cannot be interrupted and LLVM must not be torn down under it, so exit waits for those.
// (the killed helpers' scratch files and a private work directory used to stay behind; the wait for an
//  in-process compile -- only without the helper program -- is bounded by RUHVRO_HIP_EXIT_WAIT_S, default 60 s.)
void at_exit() {
  Registry& r = registry();
  std::unique_lock<std::mutex> g(r.mu);
  r.exiting = true;
  for (pid_t p : r.children) ::kill(p, SIGKILL);
  r.cv.notify_all();
  long wait_s = 60;
  if (const char* e = std::getenv("RUHVRO_HIP_EXIT_WAIT_S")) { const long v = std::atol(e); if (v >= 0 && v <= 3600) wait_s = v; }
  r.cv.wait_for(g, std::chrono::seconds(wait_s), [&] { return r.in_process == 0; });
  for (pid_t p : r.children) { int st = 0; (void)::waitpid(p, &st, 0); }      // reaped: nothing writes the files below any more
  for (const std::string& f : r.scratch) { std::remove(f.c_str()); std::remove((f + ".err").c_str()); }
  if (!r.priv_dir.empty()) {
    if (DIR* d = ::opendir(r.priv_dir.c_str())) {
      while (dirent* e = ::readdir(d)) {
        if (!std::strcmp(e->d_name, ".") || !std::strcmp(e->d_name, "..")) continue;
        std::remove((r.priv_dir + "/" + e->d_name).c_str());
      }
      ::closedir(d);
    }
    ::rmdir(r.priv_dir.c_str());
  }
}

bool read_file(const std::string& path, std::vector<char>& out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  out.assign((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  return true;
}

bool write_file_atomic(const std::string& path, const char* data, size_t n) {
  static std::atomic<unsigned> seq{0};
  const std::string tmp = path + "." + std::to_string((long)::getpid()) + "." + std::to_string(seq.fetch_add(1)) + ".tmp";
  {
    std::ofstream f(tmp, std::ios::binary);
    if (!f) return false;
    f.write(data, (std::streamsize)n);
    f.close();
    if (!f) { std::remove(tmp.c_str()); return false; }
  }
  if (std::rename(tmp.c_str(), path.c_str()) != 0) { std::remove(tmp.c_str()); return false; }
  return true;
}

// The compile helper next to the library (RUHVRO_HIP_KCOMPILE: another path; "0" / "off" = compile inside the process).
std::string helper_path() {
  static const std::string v = [] {
    std::string p;
    if (const char* e = std::getenv("RUHVRO_HIP_KCOMPILE")) {
      if (!*e || !std::strcmp(e, "0") || !std::strcmp(e, "off")) return std::string();
      p = e;
    } else {
      p = module_dir() + "/rh_kcompile";
    }
    return ::access(p.c_str(), X_OK) == 0 ? p : std::string();
  }();
  return v;
}

// Where a job's files go: the kernel cache when it can be written (other processes then find the code object there), a
// private directory of this process otherwise (read-only install: compiled once per process, like before).
std::string work_dir() {
  static std::mutex mu;
  static std::string priv;
  const std::string dir = kernel_cache_dir();
  ::mkdir(dir.c_str(), 0755);
  if (::access(dir.c_str(), W_OK | X_OK) == 0) return dir;
  std::lock_guard<std::mutex> g(mu);
  if (priv.empty()) {
    const char* t = std::getenv("TMPDIR");
    std::string tmpl = std::string(t && *t ? t : "/tmp") + "/ruhvro_hip_kcache.XXXXXX";
    std::vector<char> b(tmpl.begin(), tmpl.end());
    b.push_back(0);
    if (::mkdtemp(b.data())) {
      priv = b.data();
      Registry& r = registry();
      std::lock_guard<std::mutex> rg(r.mu);
      r.priv_dir = priv;
      if (!r.hooked) { r.hooked = true; std::atexit(at_exit); }
    }
  }
  return priv;
}

// One compile, out of process.  true = `code` holds the object (also stored at `out_path`).
bool run_helper(const std::string& helper, const std::string& source, const std::string& out_path, std::vector<char>& code, std::string& why) {
  Registry& r = registry();
  static std::atomic<unsigned> seq{0};
  const std::string src_path = out_path + "." + std::to_string((long)::getpid()) + "." + std::to_string(seq.fetch_add(1)) + ".src";
  {
    std::ofstream f(src_path, std::ios::binary);
    f.write(source.data(), (std::streamsize)source.size());
    f.close();
    if (!f) { why = "cannot write " + src_path; std::remove(src_path.c_str()); return false; }
  }
  const std::string err_path = src_path + ".err";
  char* argv[] = {(char*)helper.c_str(), (char*)src_path.c_str(), (char*)out_path.c_str(), (char*)err_path.c_str(), nullptr};
  pid_t pid = -1;
  int rc;
  {
    std::lock_guard<std::mutex> g(r.mu);     // (spawn and registration in one step: at_exit sees every child)
    if (r.exiting) { std::remove(src_path.c_str()); why = "process is exiting"; return false; }
    rc = ::posix_spawn(&pid, helper.c_str(), nullptr, nullptr, argv, environ);
    if (rc == 0) { r.children.insert(pid); r.scratch.insert(src_path); }
  }
  if (rc != 0) {
    std::remove(src_path.c_str());
    why = std::string("posix_spawn(") + helper + "): " + std::strerror(rc);
    return false;
  }
  int status = 0;
  pid_t w;
  do { w = ::waitpid(pid, &status, 0); } while (w < 0 && errno == EINTR);
  {
    std::lock_guard<std::mutex> g(r.mu);
    r.children.erase(pid);
    r.scratch.erase(src_path);
  }
  std::remove(src_path.c_str());
  // (w < 0: somebody else reaped the child -- a host with SIGCHLD ignored, a blanket wait(); its files tell the outcome)
  const bool exited_ok = w < 0 || (WIFEXITED(status) && WEXITSTATUS(status) == 0);
  if (exited_ok && read_file(out_path, code) && code.size() > 64) {
    std::remove(err_path.c_str());
    return true;
  }
  std::vector<char> log;
  if (read_file(err_path, log) && !log.empty()) why.assign(log.begin(), log.end());
  else if (w >= 0 && WIFSIGNALED(status)) why = "compile helper killed by signal " + std::to_string(WTERMSIG(status));
  else why = "compile helper failed without a message";
  std::remove(err_path.c_str());
  return false;
}

}  // namespace

class KernelImages {
 public:
  std::mutex mu;
  std::condition_variable cv;
  Slot slot[KP_COUNT];
};

std::shared_ptr<KernelImages> new_kernel_images() { return std::make_shared<KernelImages>(); }

namespace {

void job_main(std::shared_ptr<KernelImages> im, int part, std::string source, std::string out_path) {
  Registry& r = registry();
  const std::string helper = helper_path();
  bool go = true;
  {
    std::unique_lock<std::mutex> g(r.mu);
    r.cv.wait(g, [&] { return r.exiting || r.running < max_jobs(); });
    if (r.exiting) go = false;
    else {
      r.running++;
      if (helper.empty()) r.in_process++;
    }
  }
  std::vector<char> code;
  std::string why;
  bool ok = false;
  const auto t0 = std::chrono::steady_clock::now();
  if (!go) {
    why = "process is exiting";
  } else {
    if (!helper.empty()) {
      ok = run_helper(helper, source, out_path, code, why);
    } else {
      try {
        std::string log;
        code = compile_kernel(source, log);
        ok = true;
        (void)write_file_atomic(out_path, code.data(), code.size());     // best effort (a read-only install recompiles per process)
      } catch (const std::exception& e) {
        why = e.what();
      }
    }
    std::lock_guard<std::mutex> g(r.mu);
    r.running--;
    if (helper.empty()) r.in_process--;
    r.cv.notify_all();
  }
  const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  {
    std::lock_guard<std::mutex> g(im->mu);
    Slot& s = im->slot[part];
    s.seconds = secs;
    if (ok) {
      s.code = std::make_shared<const std::vector<char>>(std::move(code));
      s.state = IMG_READY;
    } else {
      s.why = why;
      s.state = IMG_FAILED;
    }
  }
  im->cv.notify_all();
}

void fill(const Slot& s, KernelImage& o) {
  o.state = s.state; o.code = s.code; o.why = s.why; o.from_cache = s.from_cache; o.seconds = s.seconds;
}

}  // namespace

unsigned kernel_images(const std::shared_ptr<KernelImages>& im, const CompiledSchema& cs, unsigned parts, CompilePolicy policy,
                       KernelImage out[KP_COUNT]) {
  unsigned started = 0;
  std::unique_lock<std::mutex> g(im->mu);
  for (int part = 0; part < KP_COUNT; part++) {
    if (!(parts & (1u << part))) continue;
    Slot& s = im->slot[part];
    if (s.state == IMG_UNKNOWN || (s.state == IMG_NOT_CACHED && policy != CP_CACHED_ONLY)) {
      const std::string src = generate_part_source(cs, part);
      if (src.empty()) { s.state = IMG_NONE; continue; }
      const std::string key = kernel_cache_key(src, kernel_part_is_encode(part));
      std::vector<char> code;
      if (read_file(kernel_cache_dir() + "/" + key + ".hsaco", code) && code.size() > 64) {
        s.code = std::make_shared<const std::vector<char>>(std::move(code));
        s.from_cache = true;
        s.state = IMG_READY;
      } else if (policy == CP_CACHED_ONLY) {
        s.state = IMG_NOT_CACHED;
      } else {
        const std::string dir = work_dir();
        if (dir.empty() && !helper_path().empty()) { s.state = IMG_FAILED; s.why = "no writable directory for the kernel compile"; continue; }
        {
          Registry& r = registry();
          std::lock_guard<std::mutex> rg(r.mu);
          if (!r.hooked) { r.hooked = true; std::atexit(at_exit); }
        }
        s.state = IMG_COMPILING;
        started++;
        std::thread(job_main, im, part, src, dir + "/" + key + ".hsaco").detach();
      }
    }
  }
  if (policy == CP_BLOCKING)
    im->cv.wait(g, [&] {
      for (int part = 0; part < KP_COUNT; part++)
        if ((parts & (1u << part)) && im->slot[part].state == IMG_COMPILING) return false;
      return true;
    });
  for (int part = 0; part < KP_COUNT; part++)
    if (parts & (1u << part)) fill(im->slot[part], out[part]);
  return started;
}

int kernel_images_wait(const std::shared_ptr<KernelImages>& im, unsigned parts, long timeout_ms, std::string* why) {
  std::unique_lock<std::mutex> g(im->mu);
  auto settled = [&] {
    for (int part = 0; part < KP_COUNT; part++)
      if ((parts & (1u << part)) && im->slot[part].state == IMG_COMPILING) return false;
    return true;
  };
  if (timeout_ms < 0) im->cv.wait(g, settled);
  else im->cv.wait_for(g, std::chrono::milliseconds(timeout_ms), settled);
  int rc = 1;
  for (int part = 0; part < KP_COUNT; part++) {
    if (!(parts & (1u << part))) continue;
    const Slot& s = im->slot[part];
    if (s.state == IMG_FAILED) { if (why) *why = s.why; return -1; }
    if (s.state != IMG_READY && s.state != IMG_NONE) rc = 0;
  }
  return rc;
}

}  // namespace rh
