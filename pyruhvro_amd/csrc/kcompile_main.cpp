// rh_kcompile: the out-of-process half of a background kernel compile (kernel_jobs.cpp).
//
//   rh_kcompile <source file> <code object path> <error file>
//
// Compiles the generated HIP source with hiprtc for gfx950 (no GPU needed) and renames the code object into place.
// Processes that want the same object (eight ranks meeting a new schema together) serialise on a lock file next to it:
// the first compiles, the others find the object there and leave.  Exit 0 = the object is at <code object path>.
#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "rtc_compile.h"

static bool present(const std::string& path) {
  struct stat st;
  return ::stat(path.c_str(), &st) == 0 && st.st_size > 64;
}

static void say(const std::string& err_path, const std::string& msg) {
  std::ofstream f(err_path, std::ios::binary);
  f << msg;
}

int main(int argc, char** argv) {
  if (argc != 4) {
    std::fprintf(stderr, "usage: rh_kcompile <source> <out.hsaco> <errfile>\n");
    return 2;
  }
  const std::string src_path = argv[1], out = argv[2], err = argv[3];
  const std::string lock = out + ".lock";
  // (no lock, e.g. a file system without flock: compile anyway.)  The holder unlinks the lock file when it is done, so a waiter
  // may wake up holding a lock on an inode that is no longer at the path while a later arrival locks a NEW file there: after
  // flock, the descriptor must still be the file at the path, else open again (after a failed compile several
  // helpers used to compile the same object at once).
  int lfd = -1;
  for (int tries = 0; tries < 64; tries++) {
    lfd = ::open(lock.c_str(), O_CREAT | O_RDWR, 0644);
    if (lfd < 0) break;
    if (::flock(lfd, LOCK_EX) != 0) break;
    struct stat a, b;
    if (::fstat(lfd, &a) == 0 && ::stat(lock.c_str(), &b) == 0 && a.st_ino == b.st_ino && a.st_dev == b.st_dev) break;
    ::close(lfd);
    lfd = -1;
  }
  int rc = 0;
  if (!present(out)) {
    try {
      std::ifstream f(src_path, std::ios::binary);
      if (!f) throw std::runtime_error("cannot read " + src_path);
      const std::string source((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
      std::string log;
      const std::vector<char> code = rh::compile_kernel(source, log);
      const std::string tmp = out + "." + std::to_string((long)::getpid()) + ".tmp";
      {
        std::ofstream o(tmp, std::ios::binary);
        o.write(code.data(), (std::streamsize)code.size());
        o.close();
        if (!o) { std::remove(tmp.c_str()); throw std::runtime_error("cannot write " + tmp); }
      }
      if (std::rename(tmp.c_str(), out.c_str()) != 0) { std::remove(tmp.c_str()); throw std::runtime_error("cannot rename to " + out); }
    } catch (const std::exception& e) {
      say(err, e.what());
      rc = 1;
    }
  }
  if (lfd >= 0) {
    ::unlink(lock.c_str());
    ::close(lfd);
  }
  return rc;
}
